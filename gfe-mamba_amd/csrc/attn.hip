// Flash-style fused attention forward for gfx950: O = softmax(Q K^T * scale) V per (batch, head), head dim 64, any sequence
// length (vit_pytorch_diy/vit_3d.py:47-57; the 1729-token synthetic 3-D ViT of SURVEY 8-d).  bf16 MFMA 32x32x16, f32 softmax.
//
// Block = 8 waves, 256 query rows (32 per wave); K/V tiles of 64 keys stream through a 3-deep LDS ring filled by LDS-DMA two tiles
// ahead (issued from asm behind counted vmcnt waits), one barrier per tile; the softmax advances 32 keys at a time (16 score registers)
// and costs ~60 instead of round 2's 122 vector instructions per block: no multiply by the scale (folded into Q), no subtraction of
// the maximum (the S chain starts from -m), deferred rescale, scalar-side key masks.
// Everything a lane owns belongs to ONE query row q = lane & 31 (both products are computed transposed):
//   S^T = K Q^T : A = K tile rows (ds_read_b128 from the swizzled [key][d] image), B = Q fragments kept in registers.
//                 C: lane (q, hi = lane >> 5) holds keys (r&3) + 8(r>>2) + 4hi of each 32-key block -> row max / row sum are
//                 in-lane reductions plus ONE exchange with lane ^ 32.
//   O^T = V^T P^T: A = V^T fragments read straight out of the row-major [key][d] V tile with ds_read_b64_tr_b16 (hardware
//                 transpose), B = P packed to bf16; the two halves of a 16-key slot are exchanged with v_permlane32_swap.
//                 C: lane (q, hi) holds d = (r&3) + 8(r>>2) + 4hi (+32) of ITS row -> the online-softmax rescale is a per-lane
//                 scalar, no cross-lane traffic at all.
#include "common.h"
#include "attn_drop.h"
#include <type_traits>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;
typedef __attribute__((address_space(3))) void* lds_void_t;

namespace {

constexpr int AD = 64;            // head dim
constexpr int QW = 32;            // query rows per wave
#ifndef GFE_ATTN_WAVES
#define GFE_ATTN_WAVES 8       // 256-row blocks, two per CU = four waves per SIMD at <= 128 registers: all 448 blocks of the bench shape resident at once
#endif
#ifndef GFE_ATTN_PREFETCH
#define GFE_ATTN_PREFETCH 0    // 1: fragment reads a block ahead of their MFMAs (+20 registers; measured no gain at three or four waves per SIMD)
#endif
#ifndef GFE_ATTN_MINW
#define GFE_ATTN_MINW 4        // waves per SIMD the register budget is set for (the LDS image is DYNAMIC shared memory: with a static 48 KB array hipcc
#endif                         // derives a lower occupancy from the LDS size and quietly ignores the bound)
constexpr int ANW = GFE_ATTN_WAVES;            // waves per block
#ifndef GFE_ATTN_KT
#define GFE_ATTN_KT 64
#endif
constexpr int KT = GFE_ATTN_KT;   // keys per tile
constexpr int TILE_BYTES = KT * AD * 2;     // 8 KiB
#ifndef GFE_ATTN_RING
#define GFE_ATTN_RING 3        // tile t+2 in flight behind a counted vmcnt wait (possible since the DMA is issued from asm, see dma16)
#endif
constexpr int RING = GFE_ATTN_RING;         // K/V tiles in LDS: tile t+RING-1 is in flight while tile t is multiplied (2: one tile ahead)
constexpr int PIECES_PER_WAVE = 2 * ((KT / 8) / ANW);   // LDS-DMA instructions one wave issues per tile (K + V)

struct AttnParams {
    const bf16_t* q; const bf16_t* k; const bf16_t* v; bf16_t* o;
    int64_t q_batch, q_row, k_batch, k_row, v_batch, v_row, o_batch, o_row;    // element strides
    int H, n;
    int nqb, total;               // query blocks per (batch, head); blocks in the grid
    float c;                      // scale * log2(e): scores are kept in log2 units
    float* nlse;                  // optional (training): -(m + log2 l) per row, [B][H][npad], rows n .. npad-1 = -inf (attn_bwd.hip)
    int npad;
    // dropout on the probabilities (vit_3d.py:56, training): element (bh, q, key) is kept iff attn_drop_hash(seed_bh, q * npad + key) >= drop_thr;
    // kept probabilities are scaled by inv_keep = 1 / (1 - p) (folded into the final normalisation); the row sum is the UNdropped softmax's
    uint32_t drop_thr, seed_lo, seed_hi;
    float inv_keep;
};

__device__ __forceinline__ int crow(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

// K image: 128-B rows (two rows per 64-bank line), 16-B chunk c of row r at slot c ^ ((r >> 1) & 7).  ds_read_b128 is served in four
// 16-lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31} (+32) (MI355X_MICROARCH.md, LDS); a lane reads row = lane & 31, so a group's rows
// are 8 even + 8 odd ones whose (r >> 1) & 7 are all different: 16 distinct (line half, slot) pairs = all 64 banks once.  (Round 2's
// c ^ (r & 7) put rows 0 / 24, 1 / 25, ... of a group on the same banks: every K fragment read was 2-way conflicted:
// SQ_LDS_BANK_CONFLICT 3.2e6 -> 0 of 6.4e6 LDS cycles, profiles/r03/attn_pmc.txt.)
__device__ __forceinline__ int k_swz(int row) { return (row >> 1) & 7; }
__device__ __forceinline__ int k_off(int row, int chunk) { return row * 128 + ((chunk ^ k_swz(row)) * 16); }
// V image: 128-B rows; chunk c of key-row r at slot c ^ (((r >> 1) & 1) << 2): the 4 key-rows x 64 B one half-wave gathers with
// ds_read_b64_tr_b16 then cover 64 distinct banks
__device__ __forceinline__ int v_off(int row, int chunk) { return row * 128 + ((chunk ^ (((row >> 1) & 1) << 2)) * 16); }

// (HIP: the second __launch_bounds__ argument is the minimum number of waves per SIMD: 3 keeps the kernel at <= 168 VGPRs)
template <bool DROP>
__global__ __launch_bounds__(ANW * 64, GFE_ATTN_MINW) void attn_fwd_kernel(const AttnParams p) {
#if defined(__HIP_DEVICE_COMPILE__)       // the host pass only needs the launch stub (the body uses device-only buffer / LDS-DMA builtins)
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];                    // 2 * RING * TILE_BYTES: K[RING], V[RING]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ql = lane & 31, hi = lane >> 5;
    GFE_FUZZ_INIT();
    // static issue priority for the younger half of the block (the arbitration loser of every SIMD pair, MI355X_MICROARCH.md): +1.2 %
    if (ANW == 8 && wave >= ANW / 2) __builtin_amdgcn_s_setprio(1);
    // Workgroups are dealt to the 8 XCDs round-robin.  The query blocks of one (batch, head) stream the same K / V (442 KB at n = 1729): give
    // each XCD a CONTIGUOUS range of (batch, head, query block) items, so that a head's blocks share one 4 MiB L2.
    int item = blockIdx.x;
    if ((p.total & 7) == 0) item = (item & 7) * (p.total >> 3) + (item >> 3);
    const int bh = item / p.nqb, qb = item - bh * p.nqb;
    const int b = bh / p.H, h = bh - b * p.H;
    const int q0 = qb * (ANW * QW) + wave * QW;
    const bf16_t* kb = p.k + (size_t)b * p.k_batch + h * AD;
    const bf16_t* vb = p.v + (size_t)b * p.v_batch + h * AD;

    // Q fragments (B operand of S^T = K Q^T): lane (q, hi) holds d = 16*ds + 8*hi + 0..7
    bf16x8 qf[4];
    {
        const int q = q0 + ql;
        const bf16_t* qp = p.q + (size_t)b * p.q_batch + (size_t)(q < p.n ? q : 0) * p.q_row + h * AD + 8 * hi;
#pragma unroll
        for (int ds = 0; ds < 4; ++ds) {
            uint4 t = *reinterpret_cast<const uint4*>(qp + 16 * ds);
            if (q >= p.n) t = make_uint4(0, 0, 0, 0);
            const uint32_t w[4] = {t.x, t.y, t.z, t.w};              // scale * log2(e) goes into Q ONCE: no multiply per score
            uint32_t r[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) r[i] = pack_bf16x2(bf16lo_to_f32(w[i]) * p.c, bf16hi_to_f32(w[i]) * p.c);
            qf[ds] = __builtin_bit_cast(bf16x8, make_uint4(r[0], r[1], r[2], r[3]));
        }
    }

    f32x16 oacc[2];
    // TRACKED pass: all 16 score registers of a lane belong to ONE query row, so the S chain can start from an accumulator that holds -m: the
    // MFMA delivers S - m and p = exp2(that), no subtraction per score.  m starts at 0 (an arbitrary reference) and is moved at the first block
    // and whenever a block's maximum exceeds it by more than 2^THR (deferred rescale: p <= 2^THR is harmless in f32 / bf16) -- a
    // wave-uniform branch the steady state does not take.
    // UNTRACKED pass (round 5, the one that normally runs): the reference stays at m = 0 for the whole row, p = exp2(S) straight out of an S chain
    // that starts from the constant 0 -- no row maximum, no exchange with lane ^ 32, no ballot, no branch: 13 of the ~60 vector instructions
    // per 32-key block gone (the kernel is bound by vector issue).  exp2 / bf16 / the f32 accumulators carry 8 exponent bits, so
    // this is EXACT arithmetic-wise as long as nothing leaves the exponent range: the row sum must end inside [2^-60, 2^100] (then the row's
    // largest probabilities kept full precision and nothing overflowed); a block in which any row ends outside it -- scores beyond ~ +-60 in
    // log2 units, which a softmax of LayerNorm'ed tokens does not produce -- is redone by the tracked pass (block-uniform decision through LDS).
    f32x16 negm;
    // This lane's share of the row sum.  Without dropout it is summed ON THE MATRIX CORE from the bf16 probabilities the PV product actually
    // multiplies (v_mfma_f32_4x4x4_16b_bf16 with A = ones: D[i][j] = sum_k B[k][j], i.e. every lane gets the exact f32 sum of the four bf16
    // values it holds in a B operand; four such 8-cycle MFMAs per 32-key block instead of sixteen v_add_f32): numerator and denominator of
    // the softmax then carry the SAME rounding -- a row dominated by one key comes out exact (a single-token sequence returns v itself and
    // an exactly-zero score gradient), where summing the unrounded f32 values left the 2^-9 rounding of P in the output.  With dropout the
    // sum is the UNdropped softmax's and stays on the vector unit.
    float lsum;
    f32x4_t lacc;
    constexpr float THR = 6.0f;
    const int ntile = (p.n + KT - 1) / KT;
    // K/V tiles arrive by LDS-DMA (buffer_load ... lds, 1 KiB per wave instruction, no staging registers): lane L of a piece fills
    // LDS slot L&7 of row L>>3, so it fetches the source chunk that the image's swizzle assigns to that slot.  Keys >= n lie beyond
    // num_records of the descriptor -> the hardware writes zeros.
    // The DMA is issued from inline asm: for the builtin (__builtin_amdgcn_raw_ptr_buffer_load_lds) hipcc assumes that the transfer may alias
    // ANY later LDS read and puts an s_waitcnt vmcnt(0) in front of the next ds_read, so every tile paid a full memory round trip (rounds 1-2:
    // "37 % of the wave cycles parked at s_waitcnt").  Here the ordering is explicit: a tile is read only behind its counted vmcnt wait +
    // the workgroup barrier at the bottom of the loop.
    const unsigned row_bytes_k = (unsigned)(p.k_row * 2), row_bytes_v = (unsigned)(p.v_row * 2);
    typedef int v4i_t __attribute__((ext_vector_type(4)));
    auto make_rsrc = [&](const bf16_t* base, unsigned row_bytes) {
        const uint64_t a = (uint64_t)base;
        v4i_t r;
        r.x = __builtin_amdgcn_readfirstlane((int)(uint32_t)a);
        r.y = __builtin_amdgcn_readfirstlane((int)((uint32_t)(a >> 32) & 0xffffu));          // stride 0
        r.z = __builtin_amdgcn_readfirstlane((int)((unsigned)(p.n - 1) * row_bytes + AD * 2));   // num_records: keys >= n read as zeros
        r.w = 0x00020000;
        return r;
    };
    const v4i_t rs_k = make_rsrc(kb, row_bytes_k), rs_v = make_rsrc(vb, row_bytes_v);
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)smem;
    auto dma16 = [&](const v4i_t& rs, unsigned lds_base, unsigned voff) {     // 64 lanes x 16 B -> LDS [lds_base + 16 * lane]
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" :: "s"(lds_base), "v"(voff), "s"(rs));      // (m0 is a reserved register for hipcc: it sets it right before each of its own uses, and cannot be named as a clobber)
    };
    auto dma = [&](int t, int buf) {
        // 16 pieces per tile pair (8 K + 8 V), 2 per wave: piece pc covers rows 8*pc .. 8*pc+7 of K or V
#pragma unroll
        for (int i = 0; i < (KT / 8) / ANW; ++i) {
            const int pc = wave + ANW * i, row = 8 * pc + (lane >> 3), slot = lane & 7;
            const unsigned key = (unsigned)(t * KT + row);
            const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + buf * TILE_BYTES + pc * 1024);
            dma16(rs_k, dst, key * row_bytes_k + ((slot ^ k_swz(row)) * 16));
            dma16(rs_v, dst + RING * TILE_BYTES, key * row_bytes_v + ((slot ^ (((row >> 1) & 1) << 2)) * 16));
        }
    };

    // LDS fragment addresses = a lane-constant offset (computed once, here) + the ring slot and the block's place in the tile, which are
    // compile-time constants because the tile loop is unrolled by RING -> they fold into the ds_read instructions' immediate offsets.
    // (Round 2 recomputed swizzles and a t % RING base per read: 25 of its 122 vector instructions per block.)
    int koff[4], voff[2][2];
#pragma unroll
    for (int ds = 0; ds < 4; ++ds) koff[ds] = k_off(ql, 2 * ds + hi);                 // (k_swz only looks at row bits 1-3: + 32 rows keep it)
    {
        // 16-lane group g = lane >> 4: d columns 32*db + 16*(g & 1) .. +15, keys 16*tt + 8*(g >> 1) + {0..3 | 4..7}
        const int g = lane >> 4, qq = (lane >> 2) & 3, pp = lane & 3;
#pragma unroll
        for (int db = 0; db < 2; ++db) {
            const int chunk = 4 * db + 2 * (g & 1) + (pp >> 1);
            voff[db][0] = v_off(8 * (g >> 1) + qq, chunk) + 8 * (pp & 1);              // (v_off's swizzle only looks at row bit 1: + 4 / + 16 rows keep it)
            voff[db][1] = v_off(8 * (g >> 1) + qq + 4, chunk) + 8 * (pp & 1);
        }
    }
    auto tile = [&](auto slot_c, const bool track, const int t) {       // track: wave-uniform (the same value in every wave of the block)
        constexpr int SLOT = decltype(slot_c)::value;
        const uint8_t* sk = smem + SLOT * TILE_BYTES;
        const uint8_t* sv = smem + (RING + SLOT) * TILE_BYTES;
        const bool ragged = t == ntile - 1 && (p.n & (KT - 1));

        // Fragment reads are issued a block ahead of their MFMAs (their LDS round trip runs under the softmax's vector work instead of in
        // front of every MFMA): K of block 0 at the top of the tile, then -- right behind a block's S chain -- V of that block and K of the next.
        bf16x8 kf[4], vf[2][2];
        auto load_k = [&](int kb2) {
#pragma unroll
            for (int ds = 0; ds < 4; ++ds) kf[ds] = *reinterpret_cast<const bf16x8*>(sk + 32 * kb2 * 128 + koff[ds]);
        };
        auto load_v = [&](int kb2) {
#pragma unroll
            for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                for (int db = 0; db < 2; ++db) {
                    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(sv + (32 * kb2 + 16 * tt) * 128 + voff[db][0]));
                    const s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(sv + (32 * kb2 + 16 * tt) * 128 + voff[db][1]));
                    union { struct { s16x4 a, b; } h; bf16x8 v; } u;
                    u.h.a = lo; u.h.b = hi4;
                    vf[tt][db] = u.v;
                }
        };
#if GFE_ATTN_PREFETCH
        load_k(0);
#endif
#pragma unroll
        for (int kb2 = 0; kb2 < KT / 32; ++kb2) {
            // a 32-key block that lies wholly behind the last key (n = 1729: the second block of the last tile) contributes exact zeros -- p = exp2(-inf),
            // nothing added to the row sums or to O -- so it is not computed at all (wave-uniform; its tile's DMA and barrier stay)
            if (ragged && kb2 > 0 && 32 * (2 * t + kb2) >= p.n) continue;
            // ---- S^T = K Q^T for a 32-key block: four 16-wide d steps
            f32x16 s;
#if !defined(GFE_ATTN_EXP_NODMA)     // NODMA: timing experiment only, K/V tiles are never restaged
            // the tile two ahead: its DMA instructions go out behind the first K fragment reads (nobody reads that ring slot any more)
            if (kb2 == 0 && t + RING - 1 < ntile) { GFE_FUZZ(); dma(t + RING - 1, (SLOT + RING - 1) % RING); }
#endif
#if !GFE_ATTN_PREFETCH
            load_k(kb2);
#if defined(GFE_ATTN_EXP_KFIRST)         // experiment: all four K fragment reads in flight before the first S MFMA (hipcc otherwise issues read, wait, MFMA four times)
            __builtin_amdgcn_sched_barrier(0);
#endif
#endif
#if defined(GFE_ATTN_EXP_NOMFMA)          // timing experiment only: no matrix instructions (wrong results)
            s = negm;
#pragma unroll
            for (int ds = 0; ds < 4; ++ds) { asm volatile("" :: "v"(kf[ds])); s[ds] += 1.0f; }
#else
#if defined(GFE_ATTN_EXP_MFMAPRIO)       // experiment: issue priority for the matrix phases (T5's per-cluster form)
            __builtin_amdgcn_s_setprio(GFE_ATTN_EXP_MFMAPRIO);
#endif
            s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[0], qf[0], negm, 0, 0, 0);         // (untracked pass: negm stays 0)
#pragma unroll
            for (int ds = 1; ds < 4; ++ds) s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[ds], qf[ds], s, 0, 0, 0);
#if defined(GFE_ATTN_EXP_MFMAPRIO)
            if (ANW == 8 && __builtin_amdgcn_readfirstlane(wave) >= ANW / 2) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
#endif
#endif
#if GFE_ATTN_PREFETCH
            __builtin_amdgcn_sched_barrier(0);
            load_v(kb2);
            if (kb2 + 1 < KT / 32) load_k(kb2 + 1);
            __builtin_amdgcn_sched_barrier(0);
#endif
            // ---- online softmax over the block's 32 keys (this lane: 16 of them, lane ^ 32: the others); s = S - m, log2 units
            const int kidx = 2 * t + kb2;
            if (ragged) {                                                // keys >= n do not exist
                const int lim = p.n - 32 * kidx, h4 = 4 * hi;
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (h4 >= lim - crow(r, 0)) s[r] = -INFINITY;        // key index crow(r, 0) + 4 hi >= lim
            }
#if defined(GFE_ATTN_EXP_NOVALU)          // timing experiment only: no softmax arithmetic (wrong results)
            bf16x8 pb[2];
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) pb[tt] = __builtin_bit_cast(bf16x8, make_float4(s[8 * tt], s[8 * tt + 1], s[8 * tt + 2], s[8 * tt + 3]));
#else
            // The row maximum is a plain fmaxf() chain: built with -fno-honor-nans (csrc/Makefile) it compiles to eight v_max3_f32 without the
            // canonicalising v_max_f32 x, x, x that hipcc otherwise puts in front of every MFMA result.  (Rounds 2-3 wrote the chain as inline asm
            // instead -- and the hazard recogniser does not look inside asm: a VALU read of an MFMA result needs 11 software wait states
            // after this 8-pass MFMA, the steady-state path had none, and the chain sometimes read the previous block's p values: spurious,
            // harmless, but run-to-run different moves of m.  Compiler-visible instructions get their s_nops and can be scheduled.)
            if (track) {
                float mx = fmaxf(s[0], s[1]);
    #pragma unroll
                for (int r = 2; r < 16; ++r) mx = fmaxf(mx, s[r]);
                {   // the other half of the row lives in lane ^ 32: one v_permlane32_swap instead of a trip through the LDS crossbar
                    const auto xm = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
                    mx = fmaxf(__uint_as_float(xm[0]), __uint_as_float(xm[1]));
                }
                if (kidx == 0 || __builtin_amdgcn_ballot_w64(mx > THR)) {   // move the maximum (mx = -inf, a fully masked block, moves nothing)
                    const float d = (kidx == 0) ? mx : fmaxf(mx, 0.f);       // per row: new m = m + d
                    const float alpha = (kidx == 0) ? 1.0f : fast_exp2(-d);       // (first block: O and l are still zero -- and exp2(-d) is inf for a first maximum below -128, 0 * inf = NaN: found by the round-5 fallback test)
    #pragma unroll
                    for (int r = 0; r < 16; ++r) { s[r] -= d; negm[r] -= d; oacc[0][r] *= alpha; oacc[1][r] *= alpha; }
                    lsum *= alpha;
                    if constexpr (!DROP) lacc *= alpha;
                }
            }
            if constexpr (DROP) {
                float ps0 = 0.f, ps1 = 0.f;
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    s[r] = fast_exp2(s[r]); s[r + 1] = fast_exp2(s[r + 1]);
                    ps0 += s[r]; ps1 += s[r + 1];
                }
                lsum += ps0 + ps1;
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) s[r] = fast_exp2(s[r]);
            }
            if constexpr (DROP) {                                        // attn = dropout(softmax): the mask is a hash of (seed, head, row, key)
                const uint32_t sb = attn_drop_seed(p.seed_lo, p.seed_hi, (uint32_t)bh);
                const uint32_t e0 = (uint32_t)(q0 + ql) * (uint32_t)p.npad + 32u * (uint32_t)kidx + 4u * (uint32_t)hi;
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (attn_drop_hash(sb, e0 + (uint32_t)crow(r, 0)) < p.drop_thr) s[r] = 0.f;
            }
            // ---- P -> bf16 B operands.  16-key slot tt of the block: lane hi=0 must hold keys 0..7 of the slot, hi=1 keys 8..15;
            // it owns {0..3, 8..11} + 4hi -> one v_permlane32_swap per word pair exchanges the misplaced halves.
            bf16x8 pb[2];
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                const uint32_t w0 = pack_bf16x2(s[8 * tt], s[8 * tt + 1]), w1 = pack_bf16x2(s[8 * tt + 2], s[8 * tt + 3]);
                const uint32_t w2 = pack_bf16x2(s[8 * tt + 4], s[8 * tt + 5]), w3 = pack_bf16x2(s[8 * tt + 6], s[8 * tt + 7]);
                const auto x0 = __builtin_amdgcn_permlane32_swap(w0, w2, false, false);
                const auto x1 = __builtin_amdgcn_permlane32_swap(w1, w3, false, false);
                const uint4 u = make_uint4(x0[0], x1[0], x0[1], x1[1]);
                pb[tt] = __builtin_bit_cast(bf16x8, u);
                if constexpr (!DROP) {
                    const s16x4 ones = {0x3f80, 0x3f80, 0x3f80, 0x3f80};
                    lacc = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(ones, __builtin_bit_cast(s16x4, u32x2_t{x0[0], x1[0]}), lacc, 0, 0, 0);
                    lacc = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(ones, __builtin_bit_cast(s16x4, u32x2_t{x0[1], x1[1]}), lacc, 0, 0, 0);
                }
            }
#endif
            // ---- O^T += V^T P^T: two 32-wide d blocks x the block's two 16-key slots; A fragments by transposing reads of [key][d]
#if !GFE_ATTN_PREFETCH
            load_v(kb2);
#endif
#if defined(GFE_ATTN_EXP_MFMAPRIO)
            __builtin_amdgcn_s_setprio(GFE_ATTN_EXP_MFMAPRIO);
#endif
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
#pragma unroll
                for (int db = 0; db < 2; ++db) {
#if defined(GFE_ATTN_EXP_NOMFMA)
                    asm volatile("" :: "v"(vf[tt][db]), "v"(pb[tt]));
#else
                    oacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[tt][db], pb[tt], oacc[db], 0, 0, 0);
#endif
                }
            }
#if defined(GFE_ATTN_EXP_MFMAPRIO)
            if (ANW == 8 && __builtin_amdgcn_readfirstlane(wave) >= ANW / 2) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
#endif
        }
        // tile t+1 has landed (this wave's pieces; the barrier makes all of it visible) while tile t+2's pieces, issued above, stay in flight
#if defined(GFE_ATTN_EXP_SKEW)            // timing experiment: the younger wave of every SIMD pair leaves the barrier late
        GFE_FUZZ();
        if (RING > 2 && t + RING - 1 < ntile) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" :: "n"(PIECES_PER_WAVE) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        GFE_FUZZ();
        if (wave >= ANW / 2) __builtin_amdgcn_s_sleep(GFE_ATTN_EXP_SKEW);
#elif defined(GFE_ATTN_EXP_DRAIN)
        GFE_FUZZ();
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        GFE_FUZZ();
#elif defined(GFE_ATTN_EXP_2BAR)
        GFE_FUZZ();
        if (RING > 2 && t + RING - 1 < ntile) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier\n\ts_barrier" :: "n"(PIECES_PER_WAVE) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier\n\ts_barrier" ::: "memory");
        GFE_FUZZ();
#else
        GFE_FUZZ();
        if (RING > 2 && t + RING - 1 < ntile) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" :: "n"(PIECES_PER_WAVE) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        GFE_FUZZ();
#endif
    };
    static_assert(RING == 3, "the tile loop is unrolled by the ring depth");
    auto pass = [&](const bool track_c) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[i][r] = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) negm[r] = 0.f;
        lsum = 0.f;
        lacc = f32x4_t{0.f, 0.f, 0.f, 0.f};
        GFE_FUZZ();
        dma(0, 0);
        if constexpr (RING > 2) { if (ntile > 1) dma(1, 1); }
        // vmcnt retires in order: leaving the newest tile's pieces outstanding is a COUNTED wait (a vmcnt(0) here would drain the prefetch)
        GFE_FUZZ();
        if (RING > 2 && ntile > 1) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" :: "n"(PIECES_PER_WAVE) : "memory");   // one tile's pieces stay in flight
        else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        GFE_FUZZ();
        for (int t = 0; t < ntile; t += 3) {
            tile(std::integral_constant<int, 0>{}, track_c, t);
            if (t + 1 < ntile) tile(std::integral_constant<int, 1>{}, track_c, t + 1);
            if (t + 2 < ntile) tile(std::integral_constant<int, 2>{}, track_c, t + 2);
        }
    };
    float l;
#if defined(GFE_ATTN_ALWAYS_TRACK)      // A/B switch: round 4's kernel (the tracked pass only)
    bool track = true;
#else
    bool track = false;
#endif
    for (;;) {                           // ONE instance of the tile loop in the code (two inlined copies cost 12-19 spilled registers at the 128 limit)
        pass(track);
        if constexpr (!DROP) lsum = lacc[0];
        l = lsum + __shfl_xor(lsum, 32, 64);
        if (track) break;
        // every wave has left the last tile's barrier, so nobody reads the ring any more: its first word carries the block's verdict
        volatile int* flag = reinterpret_cast<volatile int*>(smem);
        int lane_c = lane;
        asm volatile("" : "+v"(lane_c));
        // The row sum must lie in [2^-60, 2^100]; NaN / inf / negative are "bad" too.  Tested on the BITS: this file is compiled with
        // -fno-honor-nans, under which a float comparison may be rewritten so that it is false for a NaN.  For a non-negative
        // finite float the bit pattern is monotone in the value; a set sign bit, an all-ones exponent and every NaN fall outside the range.
        const unsigned lbits = __float_as_uint(l);
        const bool bad = (q0 + (lane_c & 31) < p.n) && !(lbits >= 0x21800000u && lbits <= 0x71800000u);      // 2^-60 = 0x21800000, 2^100 = 0x71800000
        if (tid == 0) *flag = 0;
        GFE_FUZZ();
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        GFE_FUZZ();
        if (__builtin_amdgcn_ballot_w64(bad) && lane == 0) *flag = 1;
        GFE_FUZZ();
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        const int redo = __builtin_amdgcn_readfirstlane(*flag);
        GFE_FUZZ();
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");                         // (the flag is read before the ring is refilled)
        GFE_FUZZ();
        if (!redo) break;
        track = true;
    }

    // ---- normalise and store: lane (q, hi) holds d = 32*db + crow(r, hi) of its row
    const float inv = (DROP ? p.inv_keep : 1.0f) / l;
    int lane_o = lane;                            // opaque copy: the row index and the output address are recomputed HERE instead of being kept
    asm volatile("" : "+v"(lane_o));              // alive (or spilled) across the tile loop as common subexpressions of the prologue's Q address
    const int q = q0 + (lane_o & 31);
    if (p.nlse && (lane_o >> 5) == 0 && q < p.npad)        // the backward restarts its score chains from this value: exp2(S + nlse) = the normalised probability
        p.nlse[(size_t)bh * p.npad + q] = q < p.n ? negm[0] - __builtin_amdgcn_logf(l) : -INFINITY;   // (v_log_f32 is log2)
    if (q < p.n) {
        bf16_t* op = p.o + (size_t)b * p.o_batch + (size_t)q * p.o_row + h * AD;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const int d = 32 * db + 8 * r4 + 4 * (lane_o >> 5);
                *reinterpret_cast<uint2*>(op + d) = make_uint2(pack_bf16x2(oacc[db][4 * r4] * inv, oacc[db][4 * r4 + 1] * inv),
                                                               pack_bf16x2(oacc[db][4 * r4 + 2] * inv, oacc[db][4 * r4 + 3] * inv));
            }
    }
#endif
}



}  // namespace

extern "C" {

static int attention_fwd_launch(const void* q, const void* k, const void* v, void* o, float* nlse, int64_t B, int64_t H, int64_t n, int64_t dh,
                                int64_t q_batch, int64_t q_row, int64_t k_batch, int64_t k_row, int64_t v_batch, int64_t v_row,
                                int64_t o_batch, int64_t o_row, float scale, float p_drop, int64_t seed, void* stream) {
    GFE_REQUIRE(q && k && v && o, GFE_ERR_NULL);
    GFE_REQUIRE(B > 0 && H > 0 && n > 0 && dh == AD && B * H <= 65535 && n <= 0x7fffffff, GFE_ERR_SHAPE);
    GFE_REQUIRE(q_row % 8 == 0 && k_row % 8 == 0 && v_row % 8 == 0 && o_row % 4 == 0, GFE_ERR_SHAPE);       // 16-byte loads, 8-byte stores
    GFE_REQUIRE(q_batch % 8 == 0 && k_batch % 8 == 0 && v_batch % 8 == 0 && o_batch % 4 == 0, GFE_ERR_SHAPE);
    AttnParams p;
    p.q = (const bf16_t*)q; p.k = (const bf16_t*)k; p.v = (const bf16_t*)v; p.o = (bf16_t*)o;
    p.q_batch = q_batch; p.q_row = q_row; p.k_batch = k_batch; p.k_row = k_row; p.v_batch = v_batch; p.v_row = v_row;
    p.o_batch = o_batch; p.o_row = o_row; p.H = (int)H; p.n = (int)n; p.c = scale * GFE_LOG2E;
    p.nlse = nlse; p.npad = (int)(ceil_div(n, 64) * 64);
    GFE_REQUIRE(p_drop >= 0.f && p_drop < 1.f && (p_drop == 0.f || n <= 65535), GFE_ERR_SHAPE);          // (the mask's element index q * npad + key is 32-bit)
    p.drop_thr = attn_drop_threshold(p_drop); p.seed_lo = (uint32_t)(uint64_t)seed; p.seed_hi = (uint32_t)((uint64_t)seed >> 32);
    p.inv_keep = 1.0f / (1.0f - p_drop);
    p.nqb = (int)ceil_div(n, ANW * QW);
    const int64_t total = (int64_t)p.nqb * B * H;
    GFE_REQUIRE(total <= 0x7fffffff, GFE_ERR_SHAPE);
    p.total = (int)total;
    if (p.drop_thr) hipLaunchKernelGGL(attn_fwd_kernel<true>, dim3((unsigned)total), dim3(ANW * 64), 2 * RING * TILE_BYTES, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(attn_fwd_kernel<false>, dim3((unsigned)total), dim3(ANW * 64), 2 * RING * TILE_BYTES, (hipStream_t)stream, p);
    return gfe_launch_status();
}

int gfe_attention_fwd(const void* q, const void* k, const void* v, void* o, int64_t B, int64_t H, int64_t n, int64_t dh,
                      int64_t q_batch, int64_t q_row, int64_t k_batch, int64_t k_row, int64_t v_batch, int64_t v_row,
                      int64_t o_batch, int64_t o_row, float scale, void* stream) {
    return attention_fwd_launch(q, k, v, o, nullptr, B, H, n, dh, q_batch, q_row, k_batch, k_row, v_batch, v_row, o_batch, o_row, scale, 0.f, 0, stream);
}

int gfe_attention_fwd_lse(const void* q, const void* k, const void* v, void* o, void* nlse, int64_t B, int64_t H, int64_t n, int64_t dh,
                          int64_t q_batch, int64_t q_row, int64_t k_batch, int64_t k_row, int64_t v_batch, int64_t v_row,
                          int64_t o_batch, int64_t o_row, float scale, float p_drop, int64_t seed, void* stream) {
    GFE_REQUIRE(nlse, GFE_ERR_NULL);
    return attention_fwd_launch(q, k, v, o, (float*)nlse, B, H, n, dh, q_batch, q_row, k_batch, k_row, v_batch, v_row, o_batch, o_row, scale, p_drop, seed, stream);
}

}  // extern "C"
