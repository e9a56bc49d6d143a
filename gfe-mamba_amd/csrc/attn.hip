// Flash-style fused attention forward for gfx950: O = softmax(Q K^T * scale) V per (batch, head), head dim 64, any sequence
// length (vit_pytorch_diy/vit_3d.py:47-57; the 1729-token synthetic 3-D ViT of SURVEY 8-d).  bf16 MFMA 32x32x16, f32 softmax.
//
// Block = 8 waves, 256 query rows (32 per wave); K/V tiles of 64 keys stream through a 2-deep LDS ring filled by LDS-DMA (the pieces
// of tile t+1 fly under the MFMAs of tile t), one barrier per tile; the softmax advances 32 keys at a time (16 score registers).
// Everything a lane owns belongs to ONE query row q = lane & 31 (both products are computed transposed):
//   S^T = K Q^T : A = K tile rows (ds_read_b128 from the swizzled [key][d] image), B = Q fragments kept in registers.
//                 C: lane (q, hi = lane >> 5) holds keys (r&3) + 8(r>>2) + 4hi of each 32-key block -> row max / row sum are
//                 in-lane reductions plus ONE exchange with lane ^ 32.
//   O^T = V^T P^T: A = V^T fragments read straight out of the row-major [key][d] V tile with ds_read_b64_tr_b16 (hardware
//                 transpose), B = P packed to bf16; the two halves of a 16-key slot are exchanged with v_permlane32_swap.
//                 C: lane (q, hi) holds d = (r&3) + 8(r>>2) + 4hi (+32) of ITS row -> the online-softmax rescale is a per-lane
//                 scalar, no cross-lane traffic at all.
#include "common.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;
typedef __attribute__((address_space(3))) void* lds_void_t;

namespace {

constexpr int AD = 64;            // head dim
constexpr int QW = 32;            // query rows per wave
#ifndef GFE_ATTN_WAVES
#define GFE_ATTN_WAVES 8
#endif
constexpr int ANW = GFE_ATTN_WAVES;            // waves per block
#ifndef GFE_ATTN_KT
#define GFE_ATTN_KT 64
#endif
constexpr int KT = GFE_ATTN_KT;   // keys per tile
constexpr int TILE_BYTES = KT * AD * 2;     // 8 KiB
#ifndef GFE_ATTN_RING
#define GFE_ATTN_RING 2        // 3 (tile t+2 in flight, counted vmcnt waits) measured 104.6 vs 99-101 us: the DMA's cost is not its latency
#endif
constexpr int RING = GFE_ATTN_RING;         // K/V tiles in LDS: tile t+RING-1 is in flight while tile t is multiplied (2: one tile ahead)
constexpr int PIECES_PER_WAVE = 2 * ((KT / 8) / ANW);   // LDS-DMA instructions one wave issues per tile (K + V)

struct AttnParams {
    const bf16_t* q; const bf16_t* k; const bf16_t* v; bf16_t* o;
    int64_t q_batch, q_row, k_batch, k_row, v_batch, v_row, o_batch, o_row;    // element strides
    int H, n;
    float c;                      // scale * log2(e): scores are kept in log2 units
};

__device__ __forceinline__ int crow(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

// K image: 128-B rows, 16-B chunk c of row r at slot c ^ (r & 7) (conflict-free ds_read_b128 of 32 rows x 2 chunks)
__device__ __forceinline__ int k_off(int row, int chunk) { return row * 128 + ((chunk ^ (row & 7)) * 16); }
// V image: 128-B rows; chunk c of key-row r at slot c ^ (((r >> 1) & 1) << 2): the 4 key-rows x 64 B one half-wave gathers with
// ds_read_b64_tr_b16 then cover 64 distinct banks
__device__ __forceinline__ int v_off(int row, int chunk) { return row * 128 + ((chunk ^ (((row >> 1) & 1) << 2)) * 16); }

// (HIP: the second __launch_bounds__ argument is the minimum number of waves per SIMD: 4 keeps the kernel at <= 128 VGPRs)
__global__ __launch_bounds__(ANW * 64, 4) void attn_fwd_kernel(const AttnParams p) {
#if defined(__HIP_DEVICE_COMPILE__)       // the host pass only needs the launch stub (the body uses device-only buffer / LDS-DMA builtins)
    __shared__ __attribute__((aligned(16))) uint8_t smem[2 * RING * TILE_BYTES];      // K[RING], V[RING]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ql = lane & 31, hi = lane >> 5;
    // static issue priority for the younger half of the block (the arbitration loser of every SIMD pair, MI355X_MICROARCH.md): +1.2 %
    if (wave >= ANW / 2) __builtin_amdgcn_s_setprio(1);
    const int bh = blockIdx.y, b = bh / p.H, h = bh - b * p.H;
    const int q0 = blockIdx.x * (ANW * QW) + wave * QW;
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
            qf[ds] = __builtin_bit_cast(bf16x8, t);
        }
    }

    f32x16 oacc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[i][r] = 0.f;
    float m = -INFINITY, lsum = 0.f;          // running max (log2 units) and this lane's share of the row sum

    const int ntile = (p.n + KT - 1) / KT;
    // K/V tiles arrive by LDS-DMA (buffer_load ... lds, 1 KiB per wave instruction, no staging registers): lane L of a piece fills
    // LDS slot L&7 of row L>>3, so it fetches the source chunk that the image's swizzle assigns to that slot.  Keys >= n lie beyond
    // num_records of the descriptor -> the hardware writes zeros.
    const unsigned row_bytes_k = (unsigned)(p.k_row * 2), row_bytes_v = (unsigned)(p.v_row * 2);
    const __amdgpu_buffer_rsrc_t rs_k = __builtin_amdgcn_make_buffer_rsrc((void*)kb, 0, (int)((unsigned)(p.n - 1) * row_bytes_k + AD * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_v = __builtin_amdgcn_make_buffer_rsrc((void*)vb, 0, (int)((unsigned)(p.n - 1) * row_bytes_v + AD * 2), 0x00020000);
    auto dma = [&](int t, int buf) {
        // 16 pieces per tile pair (8 K + 8 V), 4 per wave: piece pc = wave + 4*i covers rows 8*pc .. 8*pc+7 of K (i < 2) or V
#pragma unroll
        for (int i = 0; i < (KT / 8) / ANW; ++i) {
            const int pc = wave + ANW * i, row = 8 * pc + (lane >> 3), slot = lane & 7;
            const unsigned key = (unsigned)(t * KT + row);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_k, (lds_void_t)(smem + buf * TILE_BYTES + pc * 1024), 16,
                                                     key * row_bytes_k + ((slot ^ (row & 7)) * 16), 0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_v, (lds_void_t)(smem + (RING + buf) * TILE_BYTES + pc * 1024), 16,
                                                     key * row_bytes_v + ((slot ^ (((row >> 1) & 1) << 2)) * 16), 0, 0, 0);
        }
    };

    dma(0, 0);
    if constexpr (RING > 2) { if (ntile > 1) dma(1, 1); }
    // vmcnt retires in order: leaving the newest tile's pieces outstanding is a COUNTED wait (a vmcnt(0) here would drain the prefetch)
    if (RING > 2 && ntile > 1) __builtin_amdgcn_s_waitcnt(0x0f70 | (PIECES_PER_WAVE & 15) | ((PIECES_PER_WAVE >> 4) << 14));
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    for (int t = 0; t < ntile; ++t) {
        const uint8_t* sk = smem + (t % RING) * TILE_BYTES;
        const uint8_t* sv = smem + (RING + (t % RING)) * TILE_BYTES;
        const bool ragged = t == ntile - 1 && (p.n & (KT - 1));

#pragma unroll
        for (int kb2 = 0; kb2 < KT / 32; ++kb2) {
            // ---- S^T = K Q^T for a 32-key block: four 16-wide d steps
            f32x16 s;
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = 0.f;
            bf16x8 kf[4];
#pragma unroll
            for (int ds = 0; ds < 4; ++ds) kf[ds] = *reinterpret_cast<const bf16x8*>(sk + k_off(32 * kb2 + ql, 2 * ds + hi));
#if !defined(GFE_ATTN_EXP_NODMA)     // NODMA: timing experiment only, K/V tiles are never restaged
            // the next tile's DMA instructions go out behind the first K fragment reads (nobody reads that ring slot any more); it lands
            // under the next tiles' MFMAs
            if (kb2 == 0 && t + RING - 1 < ntile) dma(t + RING - 1, (t + RING - 1) % RING);
#endif
#if defined(GFE_ATTN_PRIO)
            __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
            for (int ds = 0; ds < 4; ++ds) s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[ds], qf[ds], s, 0, 0, 0);
#if defined(GFE_ATTN_PRIO)
            __builtin_amdgcn_s_setprio(0);
#endif
            // ---- online softmax over the block's 32 keys (this lane: 16 of them, lane ^ 32: the others)
            if (ragged) {                                                // keys >= n do not exist
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (t * KT + 32 * kb2 + crow(r, hi) >= p.n) s[r] = -INFINITY;
            }
            float mx = s[0];
#pragma unroll
            for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[r]);
            {   // the other half of the row lives in lane ^ 32: one v_permlane32_swap instead of a trip through the LDS crossbar
                const auto xm = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
                mx = fmaxf(__uint_as_float(xm[0]), __uint_as_float(xm[1]));
            }
            const float m_new = fmaxf(m, mx * p.c);                   // -inf only for an all-masked block (then every p is 0)
            const float m_use = m_new == -INFINITY ? 0.f : m_new;
            const float alpha = fast_exp2(m - m_use);
            m = m_new;
            float psum = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float e = fast_exp2(fmaf(s[r], p.c, -m_use));
                s[r] = e; psum += e;
            }
            lsum = fmaf(lsum, alpha, psum);
            if (__builtin_amdgcn_ballot_w64(alpha != 1.0f)) {          // wave-uniform: once the row maxima settle nothing is rescaled
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) oacc[i][r] *= alpha;
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
            }
            // ---- O^T += V^T P^T: two 32-wide d blocks x the block's two 16-key slots; A fragments by transposing reads of [key][d]
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
#pragma unroll
                for (int db = 0; db < 2; ++db) {
                    // 16-lane group g = lane >> 4: d columns 32*db + 16*(g & 1) .. +15, keys 16*ks + 8*(g >> 1) + {0..3 | 4..7}
                    const int g = lane >> 4, qq = (lane >> 2) & 3, pp = lane & 3;
                    const int key0 = 32 * kb2 + 16 * tt + 8 * (g >> 1) + qq;
                    const int chunk = 4 * db + 2 * (g & 1) + (pp >> 1);
                    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(sv + v_off(key0, chunk) + 8 * (pp & 1)));
                    const s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(sv + v_off(key0 + 4, chunk) + 8 * (pp & 1)));
                    union { struct { s16x4 a, b; } h; bf16x8 v; } u;
                    u.h.a = lo; u.h.b = hi4;
                    oacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(u.v, pb[tt], oacc[db], 0, 0, 0);
                }
            }
        }
        // tile t+1 has landed (this wave's pieces; the barrier makes all of it visible) while tile t+2's pieces, issued above, stay in flight
        if (RING > 2 && t + RING - 1 < ntile) __builtin_amdgcn_s_waitcnt(0x0f70 | (PIECES_PER_WAVE & 15) | ((PIECES_PER_WAVE >> 4) << 14));
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    // ---- normalise and store: lane (q, hi) holds d = 32*db + crow(r, hi) of its row
    const float l = lsum + __shfl_xor(lsum, 32, 64);
    const float inv = 1.0f / l;
    const int q = q0 + ql;
    if (q < p.n) {
        bf16_t* op = p.o + (size_t)b * p.o_batch + (size_t)q * p.o_row + h * AD;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const int d = 32 * db + 8 * r4 + 4 * hi;
                *reinterpret_cast<uint2*>(op + d) = make_uint2(pack_bf16x2(oacc[db][4 * r4] * inv, oacc[db][4 * r4 + 1] * inv),
                                                               pack_bf16x2(oacc[db][4 * r4 + 2] * inv, oacc[db][4 * r4 + 3] * inv));
            }
    }
#endif
}

}  // namespace

extern "C" {

int gfe_attention_fwd(const void* q, const void* k, const void* v, void* o, int64_t B, int64_t H, int64_t n, int64_t dh,
                      int64_t q_batch, int64_t q_row, int64_t k_batch, int64_t k_row, int64_t v_batch, int64_t v_row,
                      int64_t o_batch, int64_t o_row, float scale, void* stream) {
    GFE_REQUIRE(q && k && v && o, GFE_ERR_NULL);
    GFE_REQUIRE(B > 0 && H > 0 && n > 0 && dh == AD && B * H <= 65535 && n <= 0x7fffffff, GFE_ERR_SHAPE);
    GFE_REQUIRE(q_row % 8 == 0 && k_row % 8 == 0 && v_row % 8 == 0 && o_row % 4 == 0, GFE_ERR_SHAPE);       // 16-byte loads, 8-byte stores
    GFE_REQUIRE(q_batch % 8 == 0 && k_batch % 8 == 0 && v_batch % 8 == 0 && o_batch % 4 == 0, GFE_ERR_SHAPE);
    AttnParams p;
    p.q = (const bf16_t*)q; p.k = (const bf16_t*)k; p.v = (const bf16_t*)v; p.o = (bf16_t*)o;
    p.q_batch = q_batch; p.q_row = q_row; p.k_batch = k_batch; p.k_row = k_row; p.v_batch = v_batch; p.v_row = v_row;
    p.o_batch = o_batch; p.o_row = o_row; p.H = (int)H; p.n = (int)n; p.c = scale * GFE_LOG2E;
    hipLaunchKernelGGL(attn_fwd_kernel, dim3((unsigned)ceil_div(n, ANW * QW), (unsigned)(B * H)), dim3(ANW * 64), 0, (hipStream_t)stream, p);
    return gfe_launch_status();
}

}  // extern "C"
