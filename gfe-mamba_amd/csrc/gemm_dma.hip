// bf16 MFMA GEMM, persistent LDS-DMA main loop (round 4):  C[M][N] = epilogue(A[M][K] . B[N][K]^T), both operands bf16 and K-contiguous --
// the nn.Linear forward shape y = x W^T of the inference pipelines (the synthetic 3-D ViT's QKV / out / feed-forward projections,
// vit_pytorch_diy/vit_3d.py:41-46, 50, 13 832 token rows; the generator ViT's blocks, vit_pytorch_diy/vit.py:14-63).  gemm.hip keeps the
// operand modes (f32 / reduction-major sources), ragged sizes and split-K and hands this kernel the shapes it takes (gemm_dma_usable).
//
// Why a second main loop: gemm_nt_kernel<128,0,0> ran these shapes at 0.16 of the bf16 MFMA peak (profiles/r03/vit3d_b8_kernel_stats.csv:
// 59.7 % of the 3-D ViT forward) -- register-staged tiles (global_load -> VGPR -> ds_write), a barrier that drains vmcnt every 64-deep
// k-step, and with K = 512 a tile is only 8 k-steps long, so the fill latency of every tile's first loads and its epilogue are paid in
// the open, once per 128 x 128 tile.  Here:
//   * one PERSISTENT block per CU (512 threads = 8 waves, 2 per SIMD, as 4 (M) x 2 (N) wave tiles of 64 x 64) walks its tiles as ONE
//     stream of (tile, k-step) units: the 3-slot LDS ring keeps turning across tile boundaries, so the first two k-steps of tile i+1
//     are already in flight while tile i finishes and stores;
//   * staging is pure LDS-DMA (buffer_load_dwordx4 ... lds, 1 KiB per wave instruction, 6 per wave and unit), issued from inline asm
//     two units ahead and retired by COUNTED s_waitcnt vmcnt(N) in front of ONE raw s_barrier per unit (hipcc would drain vmcnt(0) in
//     front of the first ds_read after a builtin DMA, cdna_hip_programming.md 5 "Pipelining across barriers");
//   * 256 x 128 x 64 units: 384 B of operand traffic per 1 kflop-pair less than the 128 x 128 tile (L2 -> LDS is the next limit), 32 MFMA
//     16x16x32 per wave and unit between barriers;
//   * the LDS image is 128-B rows with the 16-B chunks XOR-swizzled by (row & 7) -- applied on the DMA's SOURCE address, the LDS side of
//     a DMA is lane-linear -- every ds_read_b128 fragment read takes the ideal 4 LDS cycles (tools/lds_bank_model.py);
//   * B rows are staged PERMUTED (source row (r>>2)*16 + j*4 + (r&3) at LDS row j*16 + r of a 64-row group) so that with swapped MFMA
//     operands a lane ends up with 16 CONSECUTIVE output columns of one row: 32-B (bf16) / 64-B (f32) runs per lane, whole 128-B lines
//     per row and instruction pair, instead of the 8-B pieces of the old epilogue;
//   * tiles are dealt XCD-contiguously (blocks b and b + 8 share an L2: each group of 32 blocks walks a contiguous tile range, column
//     tiles innermost), so an A row panel is read from HBM once and from the XCD's L2 by the other column tiles;
//   * epilogue stores are raw buffer stores (rows >= M fall outside num_records and are dropped: no divergence, every wave issues the same
//     number of vector-memory instructions, which is what makes the counted waits exact).
#include "common.h"
#include "gemm_dma.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef int v4i_t __attribute__((ext_vector_type(4)));
typedef unsigned v4u_t __attribute__((ext_vector_type(4)));

constexpr int DBN = 128, DBK = 64;
constexpr int DCOMPUTE = 8, DLOADERS = 4, DTHREADS = (DCOMPUTE + DLOADERS) * 64;               // 8 compute waves + 4 loader waves
constexpr int DB_BYTES = DBN * 128, DB_PIECES = DBN / 8 / DLOADERS;                            // B image of a unit: 16 KB, 4 pieces per loader wave

// Two block shapes, by the number NJ of 16-column MFMA tiles a compute wave owns:
//   NJ = 4: 256 x 128 block, waves 4 (M) x 2 (N), wave tile 64 x 64, 48 KB per unit, 3-slot ring (144 KB)   -- >= 2 tiles per CU
//   NJ = 2: 128 x 128 block, waves 2 (M) x 4 (N), wave tile 64 x 32, 32 KB per unit, 4-slot ring (128 KB)   -- grids of 1.5 .. 4 tiles per
//           CU (the 3-D ViT's N = 512 projections: 436 tiles), where the 256-row block would leave most CUs with ONE tile
template <int NJ> struct Geo {
    static constexpr int WN = DBN / (16 * NJ), WM = DCOMPUTE / WN;
    static constexpr int BM = WM * 64, G = 16 * NJ;                  // rows per block, columns per wave
    static constexpr int RING = NJ == 4 ? 3 : 4, AHEAD = RING - 1;   // ring slots, units in flight
    static constexpr int A_BYTES = BM * 128, STAGE = A_BYTES + DB_BYTES;
    static constexpr int A_PIECES = BM / 8 / DLOADERS, NPW = A_PIECES + DB_PIECES;   // DMA instructions per loader wave and unit
};

struct DmaParams {
    const bf16_t* A; const bf16_t* B; void* C; const float* bias; const void* res;
    unsigned lda2, ldb2;                      // row strides in bytes
    unsigned ldc_b, ldres_b;                  // row strides of C / res in bytes
    unsigned c_bytes, res_bytes;              // M * ld * element size (num_records of the store / residual descriptors)
    int M, N, K, nk, tiles_n, ntiles;
    int res_f32, act;
    // split-K (the generator ViT's patch embedding, vit.py:95-100: 200 x 512 x 147 456 -- a weight-streaming product): work item t = (range z, tile)
    // with the tile innermost, nk = k-steps per RANGE; a range's tile goes raw to part + z * M * N (f32), summed in range order afterwards
    int tiles_mn; float* part; unsigned part_bytes;
};

__device__ __forceinline__ v4i_t make_rsrc(const void* base, unsigned bytes) {
    const uint64_t a = (uint64_t)base;
    v4i_t r;
    r.x = __builtin_amdgcn_readfirstlane((int)(uint32_t)a);
    r.y = __builtin_amdgcn_readfirstlane((int)((uint32_t)(a >> 32) & 0xffffu));          // stride 0: raw buffer
    r.z = __builtin_amdgcn_readfirstlane((int)bytes);
    r.w = 0x00020000;
    return r;
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc_b(const void* base, unsigned bytes) {
    const uint64_t a = (uint64_t)base;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a), hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
    return __builtin_amdgcn_make_buffer_rsrc((void*)(((uint64_t)hi << 32) | lo), 0, (int)__builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}

// 64 lanes x 16 B -> LDS [lds_base + 16 * lane]; voff per lane, soff wave-uniform (m0 is reserved by hipcc: set in the same statement)
__device__ __forceinline__ void dma16(const v4i_t& rs, unsigned lds_base, unsigned voff, unsigned soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" :: "s"(lds_base), "v"(voff), "s"(rs), "s"(soff) : "memory");
}

// exact-erf GELU without the library's branches: erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, i.e. at f32 rounding level of the
// product 0.5 x (1 + erf)): 2 transcendentals + 11 plain operations per value instead of ocml's piecewise erff (the 256 x 128 tile has
// 64 values per lane: the library call cost as much issue time as the tile's MFMAs).
__device__ __forceinline__ float gelu_erf_fast(float x) {
    const float z = fabsf(x) * 0.70710678118654752f;
    const float t = fast_rcp(fmaf(0.3275911f, z, 1.0f));
    float pl = fmaf(1.061405429f, t, -1.453152027f);
    pl = fmaf(pl, t, 1.421413741f);
    pl = fmaf(pl, t, -0.284496736f);
    pl = fmaf(pl, t, 0.254829592f);
    const float e = fast_exp2(-z * z * GFE_LOG2E);
    const float erfa = fmaf(-pl * t, e, 1.0f);                    // erf(|x| / sqrt 2)
    return 0.5f * x + 0.5f * fabsf(x) * erfa;                     // 0.5 x (1 + sign(x) erf(|x|/sqrt2))
}

// Diagnostic build only (-DGFE_GDMA_STAMPS, tools/gemm_stamps.py): waves 0 and 4 of block 0 accumulate s_memtime deltas per phase of a unit.
#ifdef GFE_GDMA_STAMPS
__device__ unsigned long long g_gdma_stamps[32];
#define GD_STAMP_DECL unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long st_last = __builtin_amdgcn_s_memtime();
#define GD_STAMP(i) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); st_acc[i] += now_ - st_last; st_last = now_; }
#define GD_STAMP_FLUSH(w) if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) { for (int i_ = 0; i_ < 8; ++i_) g_gdma_stamps[(w) * 8 + i_] = st_acc[i_]; }
#else
#define GD_STAMP_DECL
#define GD_STAMP(i)
#define GD_STAMP_FLUSH(w)
#endif

// Roles (round 4, third version).  In-kernel stamps of the versions where the eight compute waves staged their own tiles: a wave spent
// 410 cycles per unit issuing its six LDS-DMA instructions (an LDS-DMA instruction costs the issuing wave ~68 cycles, in line with
// MI355X_MICROARCH.md's 60-185), 290 issuing its 16 fragment reads and 512-750 in its 32 MFMAs -- and no arrangement of the three inside
// ONE instruction stream hid the DMA issue (interleaved between MFMA groups: no change; ping-pong with the SIMD partner: the read interval
// became the longer one).  So the DMA instructions moved to waves of their own:
//   * waves 0-7  COMPUTE: 4 (M) x 2 (N) wave tiles of 64 x 64; per unit a READ interval (16 ds_read_b128, and -- in a tile's first unit --
//     the previous tile's epilogue) and an MFMA interval (32 MFMAs), each closed by one workgroup barrier; waves 4-7, the SIMD partners of
//     waves 0-3, run ONE INTERVAL BEHIND (one extra barrier up front, one at the end for the others), so every SIMD always has one wave
//     in its matrix interval while its partner reads (MI355X_MICROARCH.md "Two waves per SIMD");
//   * waves 8-11 LOADERS (one per SIMD): 12 pieces per wave and unit (8 of A, 4 of B), issued two units ahead into the ring slot both
//     compute groups have left, retired by ONE counted s_waitcnt vmcnt(12) per unit -- the loaders issue nothing else, so the count is exact.
// Timeline (tau = barrier intervals after the prologue; unit u): group 0 reads at 2u and multiplies at 2u+1; group 1 reads at 2u+1 and
// multiplies at 2u+2; the loaders issue unit u+2 at 2u (its slot, (u+2) % 3 = (u-1) % 3, was last read at 2u-1) and wait at the end of
// 2u+1 for unit u+1, which group 0 reads at 2u+2.  All twelve waves execute 2 U + 2 barriers.
template <bool OUT_F32, int NJ>
__global__ __launch_bounds__(DTHREADS, 3) void gemm_dma_kernel(const DmaParams p) {
    typedef Geo<NJ> Gm;
    constexpr int DBM = Gm::BM, DRING = Gm::RING, DA_BYTES = Gm::A_BYTES, DSTAGE = Gm::STAGE, DA_PIECES = Gm::A_PIECES, DNPW = Gm::NPW;
    extern __shared__ __attribute__((aligned(1024))) uint8_t smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    GFE_FUZZ_INIT();

    // ---- this block's tiles: XCD-contiguous ranges, interleaved over the blocks that share the XCD
    const int G = gridDim.x, bid = blockIdx.x;
    const int NG = G < 8 ? G : 8;                                  // block groups (b % 8 labels the blocks that share an XCD: a speed matter only)
    const int xcd = bid % NG, jx = bid / NG;
    const int nbx = (G - xcd + NG - 1) / NG;
    const int t_begin = (int)((int64_t)xcd * p.ntiles / NG), t_end = (int)((int64_t)(xcd + 1) * p.ntiles / NG);
    const int nk = p.nk;
    const int t_first = t_begin + jx;
    const int my_tiles = t_first < t_end ? (t_end - t_first + nbx - 1) / nbx : 0;
    const int U = my_tiles * nk;                                   // units of this block

    if (wave >= DCOMPUTE) {
        // =================================================================== loader waves
        const int lw = wave - DCOMPUTE;
        const v4i_t rsA = make_rsrc(p.A, 0xffffffffu), rsB = make_rsrc(p.B, 0xffffffffu);     // rows are clamped: no out-of-range source
        const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)smem;
        // piece pc = LDS rows 8 pc .. 8 pc + 7; lane L fills slot L & 7 of row L >> 3 and therefore fetches source chunk
        // (L & 7) ^ (row & 7) = (L & 7) ^ (L >> 3): the swizzle lives on the SOURCE address (the LDS side of a DMA is lane-linear)
        const unsigned chunk_b = (unsigned)(((lane & 7) ^ (lane >> 3)) * 16);
        int b_row[DB_PIECES];
#pragma unroll
        for (int i = 0; i < DB_PIECES; ++i) {
            const int rho = 8 * (lw + DLOADERS * i) + (lane >> 3);                          // LDS row of the B image
            const int g = rho / Gm::G, j = (rho % Gm::G) >> 4, r = rho & 15;
            b_row[i] = g * Gm::G + (r >> 2) * (4 * NJ) + j * 4 + (r & 3);                    // the source row it holds (epilogue layout, see above)
        }
        unsigned pa[DA_PIECES], pb[DB_PIECES];
        int pf_t = t_first, pf_k = 0, pf_kbase = 0;
        auto issue_unit = [&](int slot) {                              // the cursor's unit -> ring slot, cursor advances; nothing once the block's units are out
            if (pf_t >= t_end) return;
            if (pf_k == 0) {
                const int z = pf_t / p.tiles_mn, tmn = pf_t - z * p.tiles_mn;
                const int tm = tmn / p.tiles_n, tn = tmn - tm * p.tiles_n;
                pf_kbase = z * nk;
#pragma unroll
                for (int i = 0; i < DA_PIECES; ++i) pa[i] = (unsigned)min(tm * DBM + 8 * (lw + DLOADERS * i) + (lane >> 3), p.M - 1) * p.lda2 + chunk_b;
#pragma unroll
                for (int i = 0; i < DB_PIECES; ++i) pb[i] = (unsigned)min(tn * DBN + b_row[i], p.N - 1) * p.ldb2 + chunk_b;
            }
            const unsigned soff = __builtin_amdgcn_readfirstlane((unsigned)(pf_kbase + pf_k) * (DBK * 2));          // ("s" operands must be provably uniform)
            const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)slot * DSTAGE + (unsigned)lw * 1024);
#if !defined(GFE_GDMA_EXP_NODMA)           // timing experiment only (wrong results): no staging traffic at all
#pragma unroll
            for (int i = 0; i < DA_PIECES; ++i) dma16(rsA, dst + i * (DLOADERS * 1024), pa[i], soff);
#pragma unroll
            for (int i = 0; i < DB_PIECES; ++i) dma16(rsB, dst + DA_BYTES + i * (DLOADERS * 1024), pb[i], soff);
#endif
            if (++pf_k == nk) { pf_k = 0; pf_t += nbx; }
        };
        // keep(n): wait until at most n UNITS' pieces of this wave are still in flight (everything older has landed), then the barrier
        auto keep_barrier = [&](int n) {
            GFE_FUZZ();
            if (n >= 2 && Gm::AHEAD >= 3) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" :: "n"(2 * DNPW) : "memory");
            else if (n >= 1) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" :: "n"(DNPW) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        };
#pragma unroll
        for (int u = 0; u < Gm::AHEAD; ++u) issue_unit(u);             // prologue: AHEAD units in flight
        keep_barrier(min(U, Gm::AHEAD) - 1);                           // unit 0 has landed
        int s2 = Gm::AHEAD;
        for (int u = 0; u < U; ++u) {
            GFE_FUZZ();
            issue_unit(s2);                                            // tau = 2u: unit u + AHEAD, into the slot unit u - 1 was read from
            if (++s2 == DRING) s2 = 0;
            GFE_FUZZ();
            asm volatile("s_barrier" ::: "memory");
            // tau = 2u + 1: unit u + 1 must have landed before the barrier; the units behind it (up to AHEAD - 1 of them) stay in flight
            keep_barrier(min(Gm::AHEAD - 1, max(0, U - (u + 2))));
        }
        asm volatile("s_barrier" ::: "memory");                        // tau = 2U: group 1's last matrix interval
        return;
    }

    // ======================================================================= compute waves
    const int lq = lane >> 4, lr = lane & 15;
    const int wm = wave / Gm::WN, wn = wave % Gm::WN;
    // fragment addresses (bytes inside a ring slot): lane constants; the k-half ks flips bit 6 of the chunk term
    const unsigned a_off = (unsigned)(wm * 64 + lr) * 128 + (unsigned)((lq ^ (lr & 7)) * 16);
    const unsigned b_off = DA_BYTES + (unsigned)(wn * Gm::G + lr) * 128 + (unsigned)((lq ^ (lr & 7)) * 16);
    const __amdgpu_buffer_rsrc_t rsC = make_rsrc_b(p.C, p.c_bytes);
    const __amdgpu_buffer_rsrc_t rsR = make_rsrc_b(p.res ? p.res : p.C, p.res ? p.res_bytes : 0u);

    // ---- the epilogue of a finished tile: lane holds C[m = m0 + wm*64 + i*16 + lr][n = n0 + wn*64 + lq*16 + (4 j + reg)], 16 consecutive
    // columns per i; it runs inside the NEXT tile's first read interval, i.e. beside the SIMD partner's matrix interval.
    f32x4 acc[4][NJ];
    auto epilogue = [&](int t) {
        const int z = t / p.tiles_mn, tmn = t - z * p.tiles_mn;
        const int tm = tmn / p.tiles_n, tn = tmn - tm * p.tiles_n;
        constexpr int NV = 4 * NJ;                                   // consecutive output columns of this lane: 16 or 8
        const int nb = tn * DBN + wn * Gm::G + lq * NV;
        if constexpr (OUT_F32) {
            if (p.part) {
                // a K range's raw tile: rows >= M fall outside THIS range's num_records (they would land in the next range's slab otherwise)
                const __amdgpu_buffer_rsrc_t rsP = make_rsrc_b(p.part + (size_t)z * p.M * p.N, p.part_bytes);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const unsigned m = (unsigned)(tm * DBM + wm * 64 + i * 16 + lr);
                    const unsigned co = m * ((unsigned)p.N * 4u) + (unsigned)nb * 4;
#pragma unroll
                    for (int q = 0; q < NJ; ++q) {
                        const v4u_t d = {__float_as_uint(acc[i][q][0]), __float_as_uint(acc[i][q][1]), __float_as_uint(acc[i][q][2]), __float_as_uint(acc[i][q][3])};
                        __builtin_amdgcn_raw_buffer_store_b128(d, rsP, co + 16 * q, 0, 0);
                    }
                }
                return;
            }
        }
        float bv[NV];
        if (p.bias) {
#pragma unroll
            for (int q = 0; q < NJ; ++q) {
                const float4 b4 = *reinterpret_cast<const float4*>(p.bias + nb + 4 * q);
                bv[4 * q] = b4.x; bv[4 * q + 1] = b4.y; bv[4 * q + 2] = b4.z; bv[4 * q + 3] = b4.w;
            }
        } else {
#pragma unroll
            for (int q = 0; q < NV; ++q) bv[q] = 0.f;
        }
        v4u_t rv[2][NJ];                                               // residual rows, requested one row tile ahead
        auto res_load = [&](int i, v4u_t (&dst)[NJ]) {
            const unsigned m = (unsigned)(tm * DBM + wm * 64 + i * 16 + lr);
            const unsigned ro = m * p.ldres_b + (unsigned)nb * (p.res_f32 ? 4u : 2u);
#pragma unroll
            for (int q = 0; q < NJ; ++q)
                if (p.res_f32 || q < NJ / 2) dst[q] = __builtin_amdgcn_raw_buffer_load_b128(rsR, ro + 16 * q, 0, 0);
        };
        if (p.res) res_load(0, rv[0]);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (p.res && i + 1 < 4) res_load(i + 1, rv[(i + 1) & 1]);
            const unsigned m = (unsigned)(tm * DBM + wm * 64 + i * 16 + lr);
            float v[NV];
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) v[4 * j + r] = acc[i][j][r] + bv[4 * j + r];
            if (p.act == 1) {
#pragma unroll
                for (int q = 0; q < NV; ++q) v[q] = gelu_erf_fast(v[q]);
            }
            if (p.res) {
                const v4u_t (&rr)[NJ] = rv[i & 1];
                if (p.res_f32) {
#pragma unroll
                    for (int q = 0; q < NJ; ++q) {
                        v[4 * q] += __uint_as_float(rr[q].x); v[4 * q + 1] += __uint_as_float(rr[q].y);
                        v[4 * q + 2] += __uint_as_float(rr[q].z); v[4 * q + 3] += __uint_as_float(rr[q].w);
                    }
                } else {
#pragma unroll
                    for (int q = 0; q < NJ / 2; ++q) {
                        const uint32_t w[4] = {rr[q].x, rr[q].y, rr[q].z, rr[q].w};
#pragma unroll
                        for (int e = 0; e < 4; ++e) { v[8 * q + 2 * e] += bf16lo_to_f32(w[e]); v[8 * q + 2 * e + 1] += bf16hi_to_f32(w[e]); }
                    }
                }
            }
            if constexpr (OUT_F32) {
                const unsigned co = m * p.ldc_b + (unsigned)nb * 4;
#pragma unroll
                for (int q = 0; q < NJ; ++q) {
                    const v4u_t d = {__float_as_uint(v[4 * q]), __float_as_uint(v[4 * q + 1]), __float_as_uint(v[4 * q + 2]), __float_as_uint(v[4 * q + 3])};
                    __builtin_amdgcn_raw_buffer_store_b128(d, rsC, co + 16 * q, 0, 0);
                }
            } else {
                const unsigned co = m * p.ldc_b + (unsigned)nb * 2;
#pragma unroll
                for (int q = 0; q < NJ / 2; ++q) {
                    const v4u_t d = {pack_bf16x2(v[8 * q], v[8 * q + 1]), pack_bf16x2(v[8 * q + 2], v[8 * q + 3]),
                                     pack_bf16x2(v[8 * q + 4], v[8 * q + 5]), pack_bf16x2(v[8 * q + 6], v[8 * q + 7])};
                    __builtin_amdgcn_raw_buffer_store_b128(d, rsC, co + 16 * q, 0, 0);
                }
            }
        }
    };

    asm volatile("s_barrier" ::: "memory");                            // prologue: unit 0 has landed (the loaders waited for it)
    const int grp = wave >> 2;
    if (grp == 1) asm volatile("s_barrier" ::: "memory");              // group 1 runs one interval behind

    // Measured alternatives of this loop (same box, 4096^3, stamps in tools/gemm_stamps.py): ONE barrier per unit with group 1's loop rotated
    // (MFMA(k-1) then READ(k) while group 0 does READ(k) then MFMA(k)): the two waves of a SIMD drift into multiplying at the same time,
    // 1 900 instead of 1 550 cycles per unit (124 vs 113 us); both loops in one body (one epilogue instance): 137 spilled registers at the 168
    // a wave has with three waves per SIMD.
    int slot = 0;
    int prev_t = -1;
    GD_STAMP_DECL
    for (int t = t_first; t < t_end; t += nbx) {
        for (int kt = 0; kt < nk; ++kt) {
            // ======== READ interval
            GD_STAMP(7)
            if (kt == 0) {
                if (prev_t >= 0) epilogue(prev_t);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            GD_STAMP(0)
            const uint8_t* st = smem + slot * DSTAGE;
            bf16x8 af[2][4], bfr[2][NJ];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
                for (int j = 0; j < NJ; ++j) bfr[ks][j] = *reinterpret_cast<const bf16x8*>(st + (b_off ^ (ks * 64)) + j * 2048);
#pragma unroll
                for (int i = 0; i < 4; ++i) af[ks][i] = *reinterpret_cast<const bf16x8*>(st + (a_off ^ (ks * 64)) + i * 2048);
            }
            GD_STAMP(1)
#ifdef GFE_GDMA_STAMPS
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            GD_STAMP(2)
#endif
            GFE_FUZZ();
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // (the slot may be restaged once BOTH groups are past this barrier of theirs)
            GFE_FUZZ();
            __builtin_amdgcn_sched_barrier(0);
            GD_STAMP(3)
            // ======== MFMA interval
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
#if defined(GFE_GDMA_EXP_NOMFMA)           // timing experiment only (wrong results): fragment reads stay live, no matrix instructions
                        asm volatile("" :: "v"(bfr[ks][j]), "v"(af[ks][i]));
#else
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[ks][j], af[ks][i], acc[i][j], 0, 0, 0);
#endif
                    }
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
#ifdef GFE_GDMA_STAMPS
            asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
            GD_STAMP(4)
#endif
            GFE_FUZZ();
            asm volatile("s_barrier" ::: "memory");
            GD_STAMP(5)
            if (++slot == DRING) slot = 0;
        }
        prev_t = t;
    }
    GD_STAMP_FLUSH(grp)
    if (prev_t >= 0) epilogue(prev_t);
    if (grp == 0) asm volatile("s_barrier" ::: "memory");
}

}  // namespace

static int dma_num_cus() {
    static int ncu = 0;
    if (!ncu) {
        int dev = 0; hipDeviceProp_t pr;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) ncu = pr.multiProcessorCount;
        if (ncu <= 0) ncu = 256;
    }
    return ncu;
}

// Block shape for a problem: 4 = 256 x 128 blocks, 2 = 128 x 128 blocks, 0 = not this kernel.  The persistent unit stream needs tiles to
// pipeline: >= 2 tiles of 256 rows per CU take the big block (fewer operand bytes per flop); grids that would leave most CUs with one
// such tile (the 3-D ViT's N = 512 projections: 220 tiles on 256 CUs ran 34 / 64 us against gemm_nt_kernel's 26 / 54) take the 128-row
// block when that gives >= 1.5 tiles per CU; anything smaller stays on gemm_nt_kernel (two 4-wave blocks per CU hide each other's
// latencies there).  GFE_GEMM_DMA_ALL=1 lifts the size rule (tests / A-B runs), GFE_GEMM_DMA_NJ=2|4 forces a shape.
static int dma_shape(const GemmDmaArgs& a) {
    const int64_t ncu = dma_num_cus();
    const int64_t t256 = ceil_div(a.M, 256) * (a.N / DBN), t128 = ceil_div(a.M, 128) * (a.N / DBN);
    if (const char* e = getenv("GFE_GEMM_DMA_NJ")) { if (e[0] == '2') return 2; if (e[0] == '4') return 4; }
    if (t256 >= 2 * ncu) return 4;
    if (2 * t128 >= 3 * ncu) return 2;
    return getenv("GFE_GEMM_DMA_ALL") ? (t256 >= ncu ? 4 : 2) : 0;
}

bool gemm_dma_usable(const GemmDmaArgs& a) {
    if (getenv("GFE_GEMM_NO_DMA")) return false;                                   // A/B switch for measurements
    // Skinny M: the split-K kernel streams the weights better -- below half a 256-row tile.  From 129 rows up a wide product (dma_shape still
    // wants >= 2 tiles per CU) reads its weight panel once per 256 rows here, twice per 128-row tile there: the generator ViT's un-patchify
    // projection (200 x 147 456 x 512, vit.py:91-95) 104 -> ~50 us, the generator 9.61-9.68 -> 9.56 ms (GFE_GEMM_DMA_MINM=512: the old rule).
    static const int min_m = getenv("GFE_GEMM_DMA_MINM") ? atoi(getenv("GFE_GEMM_DMA_MINM")) : 129;
    if (a.nsplit > 1) {
        // the split form is for callers that ask for it (a fixed cut of K whatever M is: the sum order of a row must not depend on the batch it rides in)
        if (!a.part || !a.out_f32 || a.N % DBN != 0 || a.K % (DBK * a.nsplit) != 0 || (int64_t)a.M * a.N * 4 > 0x7fffffffLL) return false;
        if (a.lda % 8 || a.ldb % 8 || (((uintptr_t)a.A | (uintptr_t)a.B | (uintptr_t)a.part) % 16)) return false;
        return (int64_t)a.M * a.lda * 2 <= 0x7fffffffLL && (int64_t)a.N * a.ldb * 2 <= 0x7fffffffLL;
    }
    if (a.M < min_m || a.N % DBN != 0 || a.K % DBK != 0 || a.K < DBK) return false;
    if (dma_shape(a) == 0) return false;
    if (a.lda % 8 || a.ldb % 8) return false;
    if (((uintptr_t)a.A | (uintptr_t)a.B | (uintptr_t)a.C) % 16) return false;
    const int64_t esz = a.out_f32 ? 4 : 2;
    if ((a.ldc * esz) % 16) return false;
    if (a.bias && ((uintptr_t)a.bias % 16)) return false;
    if (a.res && (((uintptr_t)a.res % 16) || (a.ldres * (a.res_f32 ? 4 : 2)) % 16)) return false;
    // 32-bit buffer offsets
    const int64_t lim = 0x7fffffffLL;
    if ((int64_t)a.M * a.lda * 2 > lim || (int64_t)a.N * a.ldb * 2 > lim || (int64_t)a.M * a.ldc * esz > lim) return false;
    if (a.res && (int64_t)a.M * a.ldres * (a.res_f32 ? 4 : 2) > lim) return false;
    if (a.act != 0 && a.act != 1) return false;
    return true;
}

#ifdef GFE_GDMA_STAMPS
extern "C" int gfe_dbg_gdma_stamps(unsigned long long* host_out) { return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_gdma_stamps), sizeof(unsigned long long) * 32) == hipSuccess ? 0 : -4; }
#endif
static int g_dma_launches = 0;
extern "C" int gfe_gemm_dma_launches(void) { return g_dma_launches; }

template <bool OUT_F32, int NJ>
static int dma_launch_t(DmaParams& p, const GemmDmaArgs& a, hipStream_t st) {
    typedef Geo<NJ> Gm;
    p.tiles_mn = (int)ceil_div(a.M, Gm::BM) * p.tiles_n;
    p.ntiles = p.tiles_mn * (p.part ? a.nsplit : 1);
    const int ncu = dma_num_cus();
    const int grid = p.ntiles < ncu ? p.ntiles : ncu;
    constexpr size_t lds = (size_t)Gm::RING * Gm::STAGE;
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute((const void*)gemm_dma_kernel<OUT_F32, NJ>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr = true; }
    hipLaunchKernelGGL((gemm_dma_kernel<OUT_F32, NJ>), dim3((unsigned)grid), dim3(DTHREADS), lds, st, p);
    return gfe_launch_status();
}

int gemm_dma_launch(const GemmDmaArgs& a, hipStream_t st) {
    ++g_dma_launches;
    DmaParams p;
    p.A = (const bf16_t*)a.A; p.B = (const bf16_t*)a.B; p.C = a.C; p.bias = a.bias; p.res = a.res;
    p.lda2 = (unsigned)(a.lda * 2); p.ldb2 = (unsigned)(a.ldb * 2);
    const unsigned esz = a.out_f32 ? 4u : 2u, rsz = a.res_f32 ? 4u : 2u;
    p.ldc_b = (unsigned)a.ldc * esz; p.ldres_b = (unsigned)a.ldres * rsz;
    // the last row may be shorter than ld: num_records = bytes up to the end of row M - 1's N columns
    p.c_bytes = (unsigned)((int64_t)(a.M - 1) * a.ldc * esz + (int64_t)a.N * esz);
    p.res_bytes = a.res ? (unsigned)((int64_t)(a.M - 1) * a.ldres * rsz + (int64_t)a.N * rsz) : 0u;
    const int nsplit = a.nsplit > 1 ? a.nsplit : 1;
    p.M = a.M; p.N = a.N; p.K = a.K; p.nk = a.K / DBK / nsplit;
    p.tiles_n = a.N / DBN;
    p.res_f32 = a.res_f32; p.act = a.act;
    p.part = nsplit > 1 ? a.part : nullptr; p.part_bytes = (unsigned)((int64_t)a.M * a.N * 4);
    const int nj = nsplit > 1 ? 4 : dma_shape(a);
    if (nj == 4) return a.out_f32 ? dma_launch_t<true, 4>(p, a, st) : dma_launch_t<false, 4>(p, a, st);
    return a.out_f32 ? dma_launch_t<true, 2>(p, a, st) : dma_launch_t<false, 2>(p, a, st);
}
