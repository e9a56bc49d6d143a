// Flash-style attention BACKWARD for gfx950 (the reference trains vit_pytorch_diy/vit_3d.py:47-57 through autograd: softmax(Q K^T * scale) V
// per (batch, head), head dim 64).  Given dO and the forward's row statistic nlse = -(m + log2 l) (gfe_attention_fwd_lse):
//     P = exp2(Qs K^T + nlse)            Qs = bf16(Q * scale * log2 e): the forward's own rounded operand, so P is the forward's P
//     dV = P^T dO      dP = dO V^T      dS = P o (dP - delta),  delta = rowsum(dO o O)
//     dQ = scale * dS K                 dK = scale * dS^T Q = ln 2 * dS^T Qs
// Three launches, no atomics, every output element has one owner and one summation order (deterministic, batch-invariant):
//   attn_bwd_prep_kernel : Qs and ndelta = -delta per row (rows n .. npad-1: zeros), into workspaces
//   attn_bwd_dkdv_kernel : a wave owns 32 KEYS (K, V fragments in registers), the Qs / dO tiles of 64 query rows stream through a 3-deep
//                          LDS-DMA ring;   S, dP, dV^T, dK^T: 16 MFMA 32x32x16 per 32 x 32 block
//   attn_bwd_dq_kernel   : a wave owns 32 QUERY rows (Qs, dO fragments in registers), K / V tiles stream as in the forward;
//                          S^T, dP^T, dQ^T: 12 MFMA per block
// (S and dP are computed twice -- 28 instead of 20 MFMAs per block -- which buys dQ without f32 atomics.)
// Both products of a pair are computed in the orientation in which a lane's 16 accumulator registers belong to ONE key (dkdv) or ONE query
// row (dq), exactly like attn.hip: the row statistics enter as the INITIAL VALUE of the MFMA accumulator chain (no subtraction per
// score: s = S + nlse and dP - delta come straight out of the matrix core), p = exp2(s), dS = p * dP, one v_permlane32_swap per word pair
// turns the products into the next MFMA's B operand.
// A streamed tile is read both by rows (ds_read_b128, the A operand of S / dP) and transposed (ds_read_b64_tr_b16, the A operand of the
// dV^T / dK^T / dQ^T products): ONE swizzled image serves both conflict-free, see u_swz.
#include "common.h"
#include "attn_drop.h"
#include <type_traits>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;

namespace {

constexpr int AD = 64;            // head dim
constexpr int BW = 8;             // waves per block (256 keys / 256 query rows)
constexpr int RT = 64;            // rows per streamed tile
constexpr int TILE_BYTES = RT * AD * 2;
constexpr int RING = 3;          // dq kernel
constexpr int RING_KV = 4;       // dkdv kernel: tile t read, t+1 published, t+2 and t+3 in flight
constexpr int STAT_BYTES = RT * 4;                                   // one row statistic per tile row
constexpr int STAGE_KV = 2 * TILE_BYTES + 2 * STAT_BYTES;            // dkdv ring stage: Qs tile, dO tile, nlse, ndelta
constexpr int STAGE_Q = 2 * TILE_BYTES;                              // dq ring stage: K tile, V tile

struct BwdParams {
    const bf16_t* q; const bf16_t* k; const bf16_t* v; const bf16_t* o; const bf16_t* dout;
    bf16_t* qs; float* ndelta; const float* nlse;      // workspaces [B][H][npad][64], [B][H][npad]; nlse from the forward
    bf16_t* dq; bf16_t* dk; bf16_t* dv;
    int64_t in_batch, in_row, o_batch, o_row, g_batch, g_row;           // q/k/v, o/dout, dq/dk/dv element strides
    int H, n, npad;
    int nblk, total;              // blocks per (batch, head); blocks in the grid
    float scale;
    uint32_t drop_thr, seed_lo, seed_hi; float inv_keep;      // dropout on the probabilities (attn_drop.h), as in the forward
};

__device__ __forceinline__ int crow(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

// Tile image: 128-B rows (two per 64-bank line), 16-B chunk c of row r at slot c ^ u_swz(r), u_swz(r) = bit 1 of r -> slot bit 2, bits 2-3 of r
// -> slot bits 0-1.
//  * by rows (ds_read_b128, lane -> row lane & 31, one logical chunk): a 16-lane service group holds 8 even + 8 odd rows whose bits 1-3 are all
//    different (attn.hip k_swz) and u_swz permutes those bits: 16 distinct (line half, slot) pairs;
//  * transposed (ds_read_b64_tr_b16): a half-wave gathers 4 consecutive rows x one 64-B half; rows r and r+2 share a line half and differ in
//    bit 1 -> their slots differ in bit 2 -> the other 64-B half: 64 banks once.
__device__ __forceinline__ int u_swz(int row) { return (((row >> 1) & 1) << 2) | ((row >> 2) & 3); }
__device__ __forceinline__ int u_off(int row, int chunk) { return row * 128 + ((chunk ^ u_swz(row)) * 16); }

typedef int v4i_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ v4i_t make_rsrc(const void* base, unsigned bytes) {
    const uint64_t a = (uint64_t)base;
    v4i_t r;
    r.x = __builtin_amdgcn_readfirstlane((int)(uint32_t)a);
    r.y = __builtin_amdgcn_readfirstlane((int)((uint32_t)(a >> 32) & 0xffffu));          // stride 0
    r.z = __builtin_amdgcn_readfirstlane((int)bytes);                                        // num_records: beyond it the hardware returns zeros
    r.w = 0x00020000;
    return r;
}
// LDS-DMA issued from asm (attn.hip: hipcc drains vmcnt in front of the next ds_read when its own builtin is used)
__device__ __forceinline__ void dma16(const v4i_t& rs, unsigned lds_base, unsigned voff) {      // 64 lanes x 16 B -> LDS [lds_base + 16 * lane]
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" :: "s"(lds_base), "v"(voff), "s"(rs));
}
__device__ __forceinline__ void dma4(const v4i_t& rs, unsigned lds_base, unsigned voff) {       // 64 lanes x 4 B -> LDS [lds_base + 4 * lane]
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, 0 offen lds" :: "s"(lds_base), "v"(voff), "s"(rs));
}

// lane-constant LDS offsets of the two fragment kinds (row + ring slot + 32-row block are added as immediates: none of them touches row bits 1-3)
struct FragOffsets { int row[4]; int tr[2][2]; };
__device__ __forceinline__ FragOffsets frag_offsets(int lane) {
    FragOffsets f;
    const int ql = lane & 31, hi = lane >> 5;
#pragma unroll
    for (int ds = 0; ds < 4; ++ds) f.row[ds] = u_off(ql, 2 * ds + hi);
    // transposing read: 16-lane group g = lane >> 4 covers d columns 32*db + 16*(g & 1) .. +15 of tile rows 16*tt + 8*(g >> 1) + {0..3 | 4..7}
    const int g = lane >> 4, qq = (lane >> 2) & 3, pp = lane & 3;
#pragma unroll
    for (int db = 0; db < 2; ++db) {
        const int chunk = 4 * db + 2 * (g & 1) + (pp >> 1);
        f.tr[db][0] = u_off(8 * (g >> 1) + qq, chunk) + 8 * (pp & 1);
        f.tr[db][1] = u_off(8 * (g >> 1) + qq + 4, chunk) + 8 * (pp & 1);
    }
    return f;
}
__device__ __forceinline__ void load_rows(bf16x8 (&f)[4], const uint8_t* tile, int blk, const FragOffsets& fo) {
#pragma unroll
    for (int ds = 0; ds < 4; ++ds) f[ds] = *reinterpret_cast<const bf16x8*>(tile + 32 * blk * 128 + fo.row[ds]);
}
__device__ __forceinline__ void load_tr(bf16x8 (&f)[2][2], const uint8_t* tile, int blk, const FragOffsets& fo) {
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int db = 0; db < 2; ++db) {
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(tile + (32 * blk + 16 * tt) * 128 + fo.tr[db][0]));
            const s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(tile + (32 * blk + 16 * tt) * 128 + fo.tr[db][1]));
            union { struct { s16x4 a, b; } h; bf16x8 v; } u;
            u.h.a = lo; u.h.b = hi4;
            f[tt][db] = u.v;
        }
}
// 16 f32 products of a lane (rows crow(r, hi) of a 32-row block) -> the B operands of the two 16-row slots: lane hi = 0 must hold rows 0..7 of
// a slot, hi = 1 rows 8..15; it owns {0..3, 8..11} + 4 hi -> one v_permlane32_swap per word pair exchanges the misplaced halves
__device__ __forceinline__ void pack_b(bf16x8 (&pb)[2], const f32x16& s) {
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
        const uint32_t w0 = pack_bf16x2(s[8 * tt], s[8 * tt + 1]), w1 = pack_bf16x2(s[8 * tt + 2], s[8 * tt + 3]);
        const uint32_t w2 = pack_bf16x2(s[8 * tt + 4], s[8 * tt + 5]), w3 = pack_bf16x2(s[8 * tt + 6], s[8 * tt + 7]);
        const auto x0 = __builtin_amdgcn_permlane32_swap(w0, w2, false, false);
        const auto x1 = __builtin_amdgcn_permlane32_swap(w1, w3, false, false);
        pb[tt] = __builtin_bit_cast(bf16x8, make_uint4(x0[0], x1[0], x0[1], x1[1]));
    }
}
// a lane's row of a [row][64] bf16 matrix as MFMA B fragments: lane (row, hi) holds d = 16*ds + 8*hi + 0..7; rows >= n read as zeros
__device__ __forceinline__ void load_b_frags(bf16x8 (&f)[4], const bf16_t* base, int64_t row_stride, int row, int n, int hi) {
    const bf16_t* rp = base + (size_t)(row < n ? row : 0) * row_stride + 8 * hi;
#pragma unroll
    for (int ds = 0; ds < 4; ++ds) {
        uint4 t = *reinterpret_cast<const uint4*>(rp + 16 * ds);
        if (row >= n) t = make_uint4(0, 0, 0, 0);
        f[ds] = __builtin_bit_cast(bf16x8, t);
    }
}
// lane (row, hi) holds d = 32*db + crow(r, hi) of ITS row: four consecutive d per 8-byte store
__device__ __forceinline__ void store_rows(bf16_t* rp, const f32x16 (&acc)[2], float f, int hi) {
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
            const int d = 32 * db + 8 * r4 + 4 * hi;
            *reinterpret_cast<uint2*>(rp + d) = make_uint2(pack_bf16x2(acc[db][4 * r4] * f, acc[db][4 * r4 + 1] * f),
                                                           pack_bf16x2(acc[db][4 * r4 + 2] * f, acc[db][4 * r4 + 3] * f));
        }
}
__device__ __forceinline__ int xcd_item(int item, int total) {       // attn.hip: a head's blocks share one XCD's L2
    return ((total & 7) == 0) ? (item & 7) * (total >> 3) + (item >> 3) : item;
}

// ---- Qs = bf16(Q * scale * log2 e) (the forward's rounding, attn.hip), ndelta = -sum_d dO O: 8 lanes per row, 8 rows per wave
__global__ __launch_bounds__(256) void attn_bwd_prep_kernel(const BwdParams p, const int64_t rows_total) {
    const int64_t gr = (int64_t)blockIdx.x * 32 + (threadIdx.x >> 3);       // row of [B][H][npad]
    if (gr >= rows_total) return;
    const int part = threadIdx.x & 7;
    const int row = (int)(gr % p.npad);
    const int64_t bh = gr / p.npad;
    const int b = (int)(bh / p.H), h = (int)(bh - (int64_t)b * p.H);
    uint4 qs = make_uint4(0, 0, 0, 0);
    float acc = 0.f;
    if (row < p.n) {
        const uint4 t = *reinterpret_cast<const uint4*>(p.q + (size_t)b * p.in_batch + (size_t)row * p.in_row + h * AD + 8 * part);
        const float c = p.scale * GFE_LOG2E;
        const uint32_t w[4] = {t.x, t.y, t.z, t.w};
        uint32_t r[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) r[i] = pack_bf16x2(bf16lo_to_f32(w[i]) * c, bf16hi_to_f32(w[i]) * c);
        qs = make_uint4(r[0], r[1], r[2], r[3]);
        const size_t oo = (size_t)b * p.o_batch + (size_t)row * p.o_row + h * AD + 8 * part;
        const uint4 a = *reinterpret_cast<const uint4*>(p.o + oo), g = *reinterpret_cast<const uint4*>(p.dout + oo);
        const uint32_t aw[4] = {a.x, a.y, a.z, a.w}, gw[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) acc += bf16lo_to_f32(aw[i]) * bf16lo_to_f32(gw[i]) + bf16hi_to_f32(aw[i]) * bf16hi_to_f32(gw[i]);
    }
    *reinterpret_cast<uint4*>(p.qs + (size_t)gr * AD + 8 * part) = qs;
    acc += __shfl_xor(acc, 1, 64); acc += __shfl_xor(acc, 2, 64); acc += __shfl_xor(acc, 4, 64);
    if (part == 0) p.ndelta[gr] = -acc;
}

// ---- dK, dV: a wave owns 32 keys
// Software pipeline over the 32-row blocks of the streamed tiles (two waves per SIMD cannot hide an LDS round trip in front of every MFMA):
// block i+1's row fragments and row statistics are read into a second register set while block i is multiplied, across tile boundaries too --
// the workgroup barrier that publishes tile t+1 (and retires tile t-1's last readers) sits in FRONT of tile t's last block, and the DMA that
// refills tile t-1's ring slot with tile t+RING-1 is issued right behind it.
template <bool DROP>
__global__ __launch_bounds__(BW * 64, 2) void attn_bwd_dkdv_kernel(const BwdParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];                    // RING_KV stages of STAGE_KV bytes
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ql = lane & 31, hi = lane >> 5;
    GFE_FUZZ_INIT();
#if defined(GFE_ATTNB_PRIO)
    if (wave >= BW / 2) __builtin_amdgcn_s_setprio(GFE_ATTNB_PRIO);
#endif
    const int item = xcd_item(blockIdx.x, p.total);
    const int bh = item / p.nblk, kblk = item - bh * p.nblk;
    const int b = bh / p.H, h = bh - b * p.H;
    const int key = kblk * (BW * 32) + wave * 32 + ql;

    bf16x8 kf[4], vf[4];
    load_b_frags(kf, p.k + (size_t)b * p.in_batch + h * AD, p.in_row, key, p.n, hi);
    load_b_frags(vf, p.v + (size_t)b * p.in_batch + h * AD, p.in_row, key, p.n, hi);

    f32x16 dva[2], dka[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dva[i][r] = 0.f; dka[i][r] = 0.f; }

    const int ntile = p.npad / RT;
    const unsigned do_row_bytes = (unsigned)(p.o_row * 2);
    const v4i_t rs_q = make_rsrc(p.qs + (size_t)bh * p.npad * AD, (unsigned)p.npad * AD * 2);
    const v4i_t rs_do = make_rsrc(p.dout + (size_t)b * p.o_batch + h * AD, (unsigned)(p.n - 1) * do_row_bytes + AD * 2);   // query rows >= n: zeros
    const v4i_t rs_l = make_rsrc(p.nlse + (size_t)bh * p.npad, (unsigned)p.npad * 4);
    const v4i_t rs_d = make_rsrc(p.ndelta + (size_t)bh * p.npad, (unsigned)p.npad * 4);
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)smem;
    constexpr int PIECES = 2 * ((RT / 8) / BW) + 1;        // DMA instructions one wave issues per tile
    auto dma = [&](int t, int buf) {
#pragma unroll
        for (int i = 0; i < (RT / 8) / BW; ++i) {
            const int pc = wave + BW * i, row = 8 * pc + (lane >> 3), slot = lane & 7;
            const unsigned qrow = (unsigned)(t * RT + row);
            const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + buf * STAGE_KV + pc * 1024);
            const int chunk = slot ^ u_swz(row);
            dma16(rs_q, dst, qrow * (AD * 2) + chunk * 16);
            dma16(rs_do, dst + TILE_BYTES, qrow * do_row_bytes + chunk * 16);
        }
        // the tile's 64 + 64 row statistics: one 256-B piece each; even waves carry nlse, odd waves ndelta (the same bytes from four waves each:
        // every wave then has the same number of transfers in flight and one counted wait serves all)
        const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + buf * STAGE_KV + 2 * TILE_BYTES + (wave & 1) * STAT_BYTES);
        if (wave & 1) dma4(rs_d, dst, (unsigned)(t * RT + lane) * 4);
        else dma4(rs_l, dst, (unsigned)(t * RT + lane) * 4);
    };

    const FragOffsets fo = frag_offsets(lane);
    const int stat_off = 16 * hi;                          // rows crow(4*r4 .. 4*r4+3, hi) = 8*r4 + 4*hi + 0..3: one 16-B read per r4
    struct RowSet { bf16x8 q[4], d[4]; };                  // a block's A fragments (Qs rows, dO rows)
    RowSet rs[2];
    f32x16 s, dp;                                          // the chain starts nlse / -delta of the block's 32 rows, then S / dP, then P / dS
    auto load_frags = [&](RowSet& f, const uint8_t* stage, int blk) {
        load_rows(f.q, stage, blk, fo);
        load_rows(f.d, stage + TILE_BYTES, blk, fo);
    };
    auto load_stats = [&](const uint8_t* stage, int blk) {
        const uint8_t* sl = stage + 2 * TILE_BYTES + 32 * blk * 4 + stat_off;
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
            const float4 a = *reinterpret_cast<const float4*>(sl + 32 * r4);
            const float4 d = *reinterpret_cast<const float4*>(sl + STAT_BYTES + 32 * r4);
            s[4 * r4] = a.x; s[4 * r4 + 1] = a.y; s[4 * r4 + 2] = a.z; s[4 * r4 + 3] = a.w;
            if constexpr (DROP) { dp[4 * r4] = 0.f; dp[4 * r4 + 1] = 0.f; dp[4 * r4 + 2] = 0.f; dp[4 * r4 + 3] = 0.f; }     // the mask sits between dP and -delta
            else { dp[4 * r4] = d.x; dp[4 * r4 + 1] = d.y; dp[4 * r4 + 2] = d.z; dp[4 * r4 + 3] = d.w; }
        }
    };

#pragma unroll
    for (int t = 0; t < RING_KV - 1; ++t)
        if (t < ntile) dma(t, t);
    // vmcnt retires in order: tile 0 has landed once at most the younger tiles' pieces are outstanding
    GFE_FUZZ();
    if (ntile >= RING_KV - 1) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" :: "n"((RING_KV - 2) * PIECES) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    GFE_FUZZ();
    load_frags(rs[0], smem, 0);
    load_stats(smem, 0);

    constexpr int NB = RT / 32;                            // blocks per tile (even: the register set of a block is a compile-time choice)
    static_assert(NB % 2 == 0, "ping-pong register sets");
    auto tile = [&](auto slot_c, const int t) {
        constexpr int SLOT = decltype(slot_c)::value;
        const uint8_t* st = smem + SLOT * STAGE_KV;
        const uint8_t* nx = smem + ((SLOT + 1) % RING_KV) * STAGE_KV;
        const bool more = t + 1 < ntile;
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            RowSet& cur = rs[j & 1];
            RowSet& nxt = rs[(j + 1) & 1];
            if (j == NB - 1 && more) {
                // tile t+1 published, tile t-1 retired: (RING_KV - 3) younger tiles may stay in flight
                GFE_FUZZ();
                if (t + RING_KV - 2 < ntile) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" :: "n"((RING_KV - 3) * PIECES) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
                GFE_FUZZ();
#if !defined(GFE_ATTNB_EXP_NODMA)       // timing experiment only
                if (t + RING_KV - 1 < ntile) { GFE_FUZZ(); dma(t + RING_KV - 1, (SLOT + RING_KV - 1) % RING_KV); }
#endif
            }
            // next block's row fragments: in flight under this block's MFMA chains
#if !defined(GFE_ATTNB_EXP_NOLDS)       // timing experiment only
            if (j < NB - 1) load_frags(nxt, st, j + 1);
            else if (more) load_frags(nxt, nx, 0);
#else
#pragma unroll
            for (int i = 0; i < 4; ++i) { asm volatile("" : "+v"(nxt.q[i]), "+v"(nxt.d[i])); }
#endif
            __builtin_amdgcn_sched_barrier(0);
            // ---- S[q][key] + nlse[q]  and  dP[q][key] - delta[q]: the chains start from the row statistics
#pragma unroll
            for (int ds = 0; ds < 4; ++ds) s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cur.q[ds], kf[ds], s, 0, 0, 0);
#pragma unroll
            for (int ds = 0; ds < 4; ++ds) dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cur.d[ds], vf[ds], dp, 0, 0, 0);
            // the transposed fragments of THIS block (into the registers its row fragments leave)
            bf16x8 dot[2][2], qt[2][2];
#if !defined(GFE_ATTNB_EXP_NOLDS)
            load_tr(dot, st + TILE_BYTES, j, fo);
            load_tr(qt, st, j, fo);
#else
#pragma unroll
            for (int i = 0; i < 2; ++i) { dot[i][0] = cur.q[i]; dot[i][1] = cur.d[i]; qt[i][0] = cur.q[i + 2]; qt[i][1] = cur.d[i + 2]; }
#endif
            __builtin_amdgcn_sched_barrier(0);
            // ---- P, dS (query rows >= n: nlse = -inf -> p = 0, dO = 0 and ndelta = 0 -> dS = 0)
            bf16x8 pb[2], dsb[2];
#if defined(GFE_ATTNB_EXP_NOVALU)       // timing experiment only: no exp / product / pack (wrong results)
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                pb[tt] = __builtin_bit_cast(bf16x8, make_float4(s[8 * tt], s[8 * tt + 1], s[8 * tt + 2], s[8 * tt + 3]));
                dsb[tt] = __builtin_bit_cast(bf16x8, make_float4(dp[8 * tt], dp[8 * tt + 1], dp[8 * tt + 2], dp[8 * tt + 3]));
            }
#else
            if constexpr (DROP) {
                // dS = P o (mask / (1 - p) o dP - delta); dV takes the masked P (its 1 / (1 - p) at the very end)
                const uint32_t sb = attn_drop_seed(p.seed_lo, p.seed_hi, (uint32_t)bh);
                const uint32_t e0 = (uint32_t)(RT * t + 32 * j + 4 * hi) * (uint32_t)p.npad + (uint32_t)key;
                const uint8_t* sd = st + 2 * TILE_BYTES + STAT_BYTES + 32 * j * 4 + stat_off;
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    const float4 d4 = *reinterpret_cast<const float4*>(sd + 32 * r4);
                    const float nd[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int r = 4 * r4 + i;
                        const bool keep = attn_drop_hash(sb, e0 + (uint32_t)crow(r, 0) * (uint32_t)p.npad) >= p.drop_thr;
                        const float pr = fast_exp2(s[r]);
                        dp[r] = pr * fmaf(keep ? p.inv_keep : 0.f, dp[r], nd[i]);
                        s[r] = keep ? pr : 0.f;
                    }
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) { s[r] = fast_exp2(s[r]); dp[r] *= s[r]; }
            }
            pack_b(pb, s);
            pack_b(dsb, dp);
#endif
            // the next block's chain starts, into the registers P / dS leave: in flight under the eight MFMAs below
#if !defined(GFE_ATTNB_EXP_NOLDS)
            if (j < NB - 1) load_stats(st, j + 1);
            else if (more) load_stats(nx, 0);
#endif
            // ---- dV^T += dO^T P, dK^T += Qs^T dS
#pragma unroll
            for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                for (int db = 0; db < 2; ++db) dva[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dot[tt][db], pb[tt], dva[db], 0, 0, 0);
#pragma unroll
            for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                for (int db = 0; db < 2; ++db) dka[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qt[tt][db], dsb[tt], dka[db], 0, 0, 0);
        }
    };
    static_assert(RING_KV == 4, "the tile loop is unrolled by the ring depth");
    for (int t = 0; t < ntile; t += 4) {
        tile(std::integral_constant<int, 0>{}, t);
        if (t + 1 < ntile) tile(std::integral_constant<int, 1>{}, t + 1);
        if (t + 2 < ntile) tile(std::integral_constant<int, 2>{}, t + 2);
        if (t + 3 < ntile) tile(std::integral_constant<int, 3>{}, t + 3);
    }

    if (key < p.n) {
        const size_t go = (size_t)b * p.g_batch + (size_t)key * p.g_row + h * AD;
        store_rows(p.dv + go, dva, DROP ? p.inv_keep : 1.0f, hi);
        store_rows(p.dk + go, dka, 0.6931471805599453f, hi);             // dS^T Qs carries scale * log2 e
    }
#endif
}

// ---- dQ: a wave owns 32 query rows
template <bool DROP>
__global__ __launch_bounds__(BW * 64, 2) void attn_bwd_dq_kernel(const BwdParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];                    // RING stages of STAGE_Q bytes
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ql = lane & 31, hi = lane >> 5;
    GFE_FUZZ_INIT();
#if defined(GFE_ATTNB_PRIO)
    if (wave >= BW / 2) __builtin_amdgcn_s_setprio(GFE_ATTNB_PRIO);
#endif
    const int item = xcd_item(blockIdx.x, p.total);
    const int bh = item / p.nblk, qblk = item - bh * p.nblk;
    const int b = bh / p.H, h = bh - b * p.H;
    const int q = qblk * (BW * 32) + wave * 32 + ql;

    bf16x8 qf[4], dof[4];
    load_b_frags(qf, p.qs + (size_t)bh * p.npad * AD, AD, q, p.npad, hi);
    load_b_frags(dof, p.dout + (size_t)b * p.o_batch + h * AD, p.o_row, q, p.n, hi);
    const float nl = q < p.npad ? p.nlse[(size_t)bh * p.npad + q] : -INFINITY;
    const float nd = q < p.npad ? p.ndelta[(size_t)bh * p.npad + q] : 0.f;
    f32x16 nlv, ndv;                 // chain starts (all 16 score registers of a lane belong to its row)
#pragma unroll
    for (int r = 0; r < 16; ++r) { nlv[r] = nl; ndv[r] = nd; }

    f32x16 dqa[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) dqa[i][r] = 0.f;

    const int ntile = (p.n + RT - 1) / RT;
    const unsigned row_bytes = (unsigned)(p.in_row * 2);
    const v4i_t rs_k = make_rsrc(p.k + (size_t)b * p.in_batch + h * AD, (unsigned)(p.n - 1) * row_bytes + AD * 2);    // keys >= n: zeros
    const v4i_t rs_v = make_rsrc(p.v + (size_t)b * p.in_batch + h * AD, (unsigned)(p.n - 1) * row_bytes + AD * 2);
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)smem;
    constexpr int PIECES = 2 * ((RT / 8) / BW);
    auto dma = [&](int t, int buf) {
#pragma unroll
        for (int i = 0; i < (RT / 8) / BW; ++i) {
            const int pc = wave + BW * i, row = 8 * pc + (lane >> 3), slot = lane & 7;
            const unsigned off = (unsigned)(t * RT + row) * row_bytes + (slot ^ u_swz(row)) * 16;
            const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + buf * STAGE_Q + pc * 1024);
            dma16(rs_k, dst, off);
            dma16(rs_v, dst + TILE_BYTES, off);
        }
    };
    dma(0, 0);
    if (ntile > 1) dma(1, 1);
    GFE_FUZZ();
    if (ntile > 1) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" :: "n"(PIECES) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    GFE_FUZZ();

    const FragOffsets fo = frag_offsets(lane);
    auto tile = [&](auto slot_c, const int t) {
        constexpr int SLOT = decltype(slot_c)::value;
        const uint8_t* sk = smem + SLOT * STAGE_Q;
        const uint8_t* sv = sk + TILE_BYTES;
        const bool ragged = t == ntile - 1 && (p.n & (RT - 1));
#pragma unroll
        for (int kb2 = 0; kb2 < RT / 32; ++kb2) {
            if (kb2 == 0 && t + RING - 1 < ntile) { GFE_FUZZ(); dma(t + RING - 1, (SLOT + RING - 1) % RING); }
            if (ragged && kb2 > 0 && RT * t + 32 * kb2 >= p.n) continue;      // a block wholly behind the last key adds exact zeros to dQ (attn.hip: same skip)
            bf16x8 kr[4], vr[4];
            load_rows(kr, sk, kb2, fo);
            load_rows(vr, sv, kb2, fo);
            // ---- S^T[key][q] + nlse[q], dP^T[key][q] - delta[q]
            f32x16 s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kr[0], qf[0], nlv, 0, 0, 0);
#pragma unroll
            for (int ds = 1; ds < 4; ++ds) s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kr[ds], qf[ds], s, 0, 0, 0);
            f32x16 dp;
            if constexpr (DROP) {
                f32x16 z;
#pragma unroll
                for (int r = 0; r < 16; ++r) z[r] = 0.f;
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vr[0], dof[0], z, 0, 0, 0);
            } else {
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vr[0], dof[0], ndv, 0, 0, 0);
            }
#pragma unroll
            for (int ds = 1; ds < 4; ++ds) dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vr[ds], dof[ds], dp, 0, 0, 0);
            if (ragged) {                                                // keys >= n do not exist
                const int lim = p.n - (RT * t + 32 * kb2), h4 = 4 * hi;
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (h4 >= lim - crow(r, 0)) s[r] = -INFINITY;
            }
            if constexpr (DROP) {
                const uint32_t sb = attn_drop_seed(p.seed_lo, p.seed_hi, (uint32_t)bh);
                const uint32_t e0 = (uint32_t)q * (uint32_t)p.npad + (uint32_t)(RT * t + 32 * kb2 + 4 * hi);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const bool keep = attn_drop_hash(sb, e0 + (uint32_t)crow(r, 0)) >= p.drop_thr;
                    dp[r] = fast_exp2(s[r]) * fmaf(keep ? p.inv_keep : 0.f, dp[r], nd);
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) dp[r] *= fast_exp2(s[r]);
            }
            bf16x8 dsb[2];
            pack_b(dsb, dp);
            // ---- dQ^T += K^T dS^T
            bf16x8 kt[2][2];
            load_tr(kt, sk, kb2, fo);
#pragma unroll
            for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                for (int db = 0; db < 2; ++db) dqa[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kt[tt][db], dsb[tt], dqa[db], 0, 0, 0);
        }
        GFE_FUZZ();
        if (t + RING - 1 < ntile) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" :: "n"(PIECES) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        GFE_FUZZ();
    };
    for (int t = 0; t < ntile; t += 3) {
        tile(std::integral_constant<int, 0>{}, t);
        if (t + 1 < ntile) tile(std::integral_constant<int, 1>{}, t + 1);
        if (t + 2 < ntile) tile(std::integral_constant<int, 2>{}, t + 2);
    }
    if (q < p.n) store_rows(p.dq + (size_t)b * p.g_batch + (size_t)q * p.g_row + h * AD, dqa, p.scale, hi);
#endif
}

}  // namespace

extern "C" {

int gfe_attention_bwd(const void* q, const void* k, const void* v, const void* o, const void* dout, const void* nlse,
                      void* dq, void* dk, void* dv, void* qs_ws, void* ndelta_ws, int64_t B, int64_t H, int64_t n, int64_t dh,
                      int64_t in_batch, int64_t in_row, int64_t o_batch, int64_t o_row, int64_t g_batch, int64_t g_row,
                      float scale, float p_drop, int64_t seed, void* stream) {
    GFE_REQUIRE(q && k && v && o && dout && nlse && dq && dk && dv && qs_ws && ndelta_ws, GFE_ERR_NULL);
    GFE_REQUIRE(B > 0 && H > 0 && n > 0 && dh == AD && B * H <= 65535 && n <= (1 << 24), GFE_ERR_SHAPE);
    GFE_REQUIRE(in_row % 8 == 0 && in_batch % 8 == 0 && o_row % 8 == 0 && o_batch % 8 == 0 && g_row % 4 == 0 && g_batch % 4 == 0, GFE_ERR_SHAPE);
    GFE_REQUIRE((uint64_t)(n - 1) * (uint64_t)in_row * 2 + AD * 2 < (1ull << 32) && (uint64_t)(n - 1) * (uint64_t)o_row * 2 + AD * 2 < (1ull << 32), GFE_ERR_SHAPE);
    BwdParams p;
    p.q = (const bf16_t*)q; p.k = (const bf16_t*)k; p.v = (const bf16_t*)v; p.o = (const bf16_t*)o; p.dout = (const bf16_t*)dout;
    p.qs = (bf16_t*)qs_ws; p.ndelta = (float*)ndelta_ws; p.nlse = (const float*)nlse;
    p.dq = (bf16_t*)dq; p.dk = (bf16_t*)dk; p.dv = (bf16_t*)dv;
    p.in_batch = in_batch; p.in_row = in_row; p.o_batch = o_batch; p.o_row = o_row; p.g_batch = g_batch; p.g_row = g_row;
    p.H = (int)H; p.n = (int)n; p.npad = (int)(ceil_div(n, RT) * RT); p.scale = scale;
    GFE_REQUIRE(p_drop >= 0.f && p_drop < 1.f && (p_drop == 0.f || n <= 65535), GFE_ERR_SHAPE);
    p.drop_thr = attn_drop_threshold(p_drop); p.seed_lo = (uint32_t)(uint64_t)seed; p.seed_hi = (uint32_t)((uint64_t)seed >> 32);
    p.inv_keep = 1.0f / (1.0f - p_drop);
    p.nblk = (int)ceil_div(n, BW * 32);
    const int64_t total = (int64_t)p.nblk * B * H, rows = B * H * p.npad;
    GFE_REQUIRE(total <= 0x7fffffff && rows <= 0x7fffffff, GFE_ERR_SHAPE);
    p.total = (int)total;
    hipStream_t st = (hipStream_t)stream;
    static bool attr_set = false;                         // 66 KB of dynamic LDS for the dK/dV ring
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)attn_bwd_dkdv_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, RING_KV * STAGE_KV);
        (void)hipFuncSetAttribute((const void*)attn_bwd_dkdv_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, RING_KV * STAGE_KV);
        (void)hipFuncSetAttribute((const void*)attn_bwd_dq_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, RING * STAGE_Q);
        (void)hipFuncSetAttribute((const void*)attn_bwd_dq_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, RING * STAGE_Q);
        attr_set = true;
    }
    hipLaunchKernelGGL(attn_bwd_prep_kernel, dim3((unsigned)ceil_div(rows, 32)), dim3(256), 0, st, p, rows);
    if (p.drop_thr) {
        hipLaunchKernelGGL(attn_bwd_dkdv_kernel<true>, dim3((unsigned)total), dim3(BW * 64), RING_KV * STAGE_KV, st, p);
        hipLaunchKernelGGL(attn_bwd_dq_kernel<true>, dim3((unsigned)total), dim3(BW * 64), RING * STAGE_Q, st, p);
    } else {
        hipLaunchKernelGGL(attn_bwd_dkdv_kernel<false>, dim3((unsigned)total), dim3(BW * 64), RING_KV * STAGE_KV, st, p);
        hipLaunchKernelGGL(attn_bwd_dq_kernel<false>, dim3((unsigned)total), dim3(BW * 64), RING * STAGE_Q, st, p);
    }
    return gfe_launch_status();
}

}  // extern "C"
