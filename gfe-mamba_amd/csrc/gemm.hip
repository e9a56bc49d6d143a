// bf16 MFMA GEMM for gfx950: C[M][N] (+)= A[M][K] . B[N][K]^T  (both operands K-contiguous = the nn.Linear forward shape
// y = x W^T).  dgrad / wgrad reuse it after a bf16 transpose (gfe_transpose_bf16) of the operand that is not K-major.
// Serves every Linear of the path: ViT patch-embed / un-embed (vit_pytorch_diy/vit.py:95-110; skinny M = B*24, weights
// streamed once -> split-K over the 147k-wide K), ViT blocks (vit.py:14-63), Mamba projections (cross_atten/mamba.py:204,
// 235-238, 223), CrossAttention q/k/v/out (cross_atten/sd_cross_atten.py:42-45), GEGLU FF (corss_ft_transformer.py:15-22).
//
// Tile BMx128x64, 4 waves as 2(M) x 2(N), MFMA 16x16x32 with swapped operands (B rows = MFMA A operand) so that a lane
// owns 4 consecutive n of one m -> 8-byte stores.  LDS rows are 128 B with the 16-B chunks XOR-swizzled by (row & 7): every
// ds_read_b128 fragment read takes the ideal 4 LDS cycles (tools/lds_bank_model.py; a 144-B padded stride costs 8).  Register-prefetched double buffering (global loads of tile k+1 fly under the MFMAs
// of tile k).  f32 accumulate; epilogue: +bias, exact-erf GELU, +residual, bf16 or f32 store, or f32 atomics for split-K.
#include <cstdlib>
#include "common.h"
#include "gemm_dma.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

namespace {

constexpr int BN = 128, BK = 64, RS = 128;     // RS: LDS row stride in bytes (no padding; 16-B chunk c of row r lives at slot c ^ (r & 7))

struct GemmParams {
    const bf16_t* A; const bf16_t* B; void* C; const float* bias; const void* res;
    int64_t lda, ldb, ldc, ldres;
    int M, N, K, ksplit;      // ksplit: K range per blockIdx.z (multiple of BK)
    int out_f32, res_f32, act, atomic;
    float* part;              // split-K without atomics: range z stores its tile to part[z][M][N], summed in a fixed order afterwards
    int a_mode, b_mode;       // bit 0: f32 source (rounded to bf16 on load); bit 1: reduction-major source,
                              // element (row, k) at base[k * ld + row] (the operand of a dgrad / wgrad, no transpose pass)
};

// 8 operand elements for one register item of a ROWS x 64 tile, kept RAW in registers (an f32 source is rounded when the item is
// written to LDS, so the global loads stay in flight under the MFMAs of the current tile).  K-major source: 8 consecutive k of one
// row (item = row*8 + chunk); reduction-major source: 8 consecutive rows of one k (item = k * ROWS/8 + row chunk), kept in that
// order in LDS and transposed by the fragment read (ds_read_b64_tr_b16).
struct RawItem { uint4 lo, hi; };      // bf16 source: lo only; f32 source: lo = elements 0-3, hi = 4-7

template <int ROWS, int MODE>
__device__ __forceinline__ RawItem gen_load(const void* base, int64_t ld, int row0, int nrows, int k0, int kend, int it) {
    constexpr bool F32 = MODE & 1, TR = MODE & 2;
    RawItem o; o.lo = make_uint4(0, 0, 0, 0); o.hi = make_uint4(0, 0, 0, 0);
    if constexpr (!TR) {
        const int r = row0 + (it >> 3), k = k0 + (it & 7) * 8;
        if (r < nrows && k < kend) {
            if constexpr (F32) {
                const uint4* q = reinterpret_cast<const uint4*>((const float*)base + (size_t)r * ld + k);
                o.lo = q[0]; o.hi = q[1];
            } else {
                o.lo = *reinterpret_cast<const uint4*>((const bf16_t*)base + (size_t)r * ld + k);
            }
        }
    } else {
        constexpr int RC = ROWS / 8;
        const int k = k0 + it / RC, r = row0 + (it % RC) * 8;
        if (k < kend && r < nrows) {
            if (r + 8 <= nrows) {
                if constexpr (F32) {
                    const uint4* q = reinterpret_cast<const uint4*>((const float*)base + (size_t)k * ld + r);
                    o.lo = q[0]; o.hi = q[1];
                } else {
                    o.lo = *reinterpret_cast<const uint4*>((const bf16_t*)base + (size_t)k * ld + r);
                }
            } else {                                   // ragged last row chunk
                uint32_t w[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    if (r + e < nrows) {
                        if constexpr (F32) w[e] = ((const uint32_t*)base)[(size_t)k * ld + r + e];
                        else w[e >> 1] |= (uint32_t)((const bf16_t*)base)[(size_t)k * ld + r + e] << ((e & 1) * 16);
                    }
                }
                o.lo = make_uint4(w[0], w[1], w[2], w[3]); o.hi = make_uint4(w[4], w[5], w[6], w[7]);
            }
        }
    }
    return o;
}

// chunk swizzle of a reduction-major LDS tile (row = k, ROWS*2 bytes per k-row)
template <int ROWS>
__device__ __forceinline__ int tr_swz(int k) {
    if constexpr (ROWS == 128) return ((k & 3) << 1) | (((k >> 3) & 1) << 3);          // 256-B k-rows: 16 chunks, every row starts on bank 0
    else return (((k >> 1) & 1) << 1) | (((k >> 3) & 1) << 2);                          // 128-B k-rows: k & 1 already picks the bank half
}

template <int ROWS, int MODE>
__device__ __forceinline__ void gen_store(uint8_t* s, int it, const RawItem& raw) {
    constexpr bool F32 = MODE & 1, TR = MODE & 2;
    uint4 v = raw.lo;
    if constexpr (F32) {
        v = make_uint4(pack_bf16x2(__uint_as_float(raw.lo.x), __uint_as_float(raw.lo.y)), pack_bf16x2(__uint_as_float(raw.lo.z), __uint_as_float(raw.lo.w)),
                       pack_bf16x2(__uint_as_float(raw.hi.x), __uint_as_float(raw.hi.y)), pack_bf16x2(__uint_as_float(raw.hi.z), __uint_as_float(raw.hi.w)));
    }
    if constexpr (!TR) {
        *reinterpret_cast<uint4*>(s + (it >> 3) * RS + (((it & 7) ^ ((it >> 3) & 7)) * 16)) = v;
    } else {
        // reduction-major tile keeps its memory order in LDS: [64 k][ROWS] bf16, 16-B chunk c of k-row k at slot c ^ tr_swz(k)
        constexpr int RC = ROWS / 8;
        const int k = it / RC, c = it % RC;
        *reinterpret_cast<uint4*>(s + k * (ROWS * 2) + ((c ^ tr_swz<ROWS>(k)) * 16)) = v;
    }
}

// MFMA 16x16x32 operand fragment (lane: row col0 + lr, k = kk0 + 0..7 where kk0 already includes 8*lq) out of a reduction-major
// tile: two ds_read_b64_tr_b16 -- per 16-lane group the hardware gathers a 4(k) x 16(row) block and hands lane i column i.
// Lane 4q+p of the group supplies the address of k-row q, rows 4p..4p+3 (cdna_hip_programming.md T10).  The swizzle puts the
// 8 k-rows x 32 B that one 32-lane half touches on 64 distinct banks.  EXEC must be all ones (it is: no divergence in the main loop).
template <int ROWS>
__device__ __forceinline__ bf16x8 tr_frag(const uint8_t* tile, int col0, int kk0, int lr) {
    typedef __attribute__((ext_vector_type(4))) short s16x4;
    typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;
    const int q = lr >> 2, pp = lr & 3;
    const int c = (col0 >> 3) + (pp >> 1);
    const int k0 = kk0 + q, k1 = k0 + 4;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(tile + k0 * (ROWS * 2) + ((c ^ tr_swz<ROWS>(k0)) * 16) + 8 * (pp & 1)));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(tile + k1 * (ROWS * 2) + ((c ^ tr_swz<ROWS>(k1)) * 16) + 8 * (pp & 1)));
    union { struct { s16x4 a, b; } s; bf16x8 v; } u;
    u.s.a = lo; u.s.b = hi;
    return u.v;
}

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }

template <int BM, int AM, int BMODE>
__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(const GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    constexpr int A_BYTES = BM * RS, B_BYTES = BN * RS, STAGE = A_BYTES + B_BYTES;
    constexpr int MT = BM / 32;                  // 16-row m tiles per wave (wave tile = BM/2 x 64)
    constexpr int A_ITEMS = BM * 8 / 256, B_ITEMS = BN * 8 / 256;   // 16-B chunks per thread per tile
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lq = lane >> 4, lr = lane & 15;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int kbeg = blockIdx.z * p.ksplit;
    const int kend = min(p.K, kbeg + p.ksplit);
    const int nk = (kend - kbeg + BK - 1) / BK;

    f32x4 acc[MT][4];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    RawItem ra[A_ITEMS], rb[B_ITEMS];
    auto gload = [&](int kt) {
        const int k0 = kbeg + kt * BK;
#pragma unroll
        for (int i = 0; i < A_ITEMS; ++i) ra[i] = gen_load<BM, AM>(p.A, p.lda, m0, p.M, k0, kend, tid + i * 256);
#pragma unroll
        for (int i = 0; i < B_ITEMS; ++i) rb[i] = gen_load<BN, BMODE>(p.B, p.ldb, n0, p.N, k0, kend, tid + i * 256);
    };
    auto lstore = [&](int buf) {
        uint8_t* sa = smem + buf * STAGE;
        uint8_t* sb = sa + A_BYTES;
#pragma unroll
        for (int i = 0; i < A_ITEMS; ++i) gen_store<BM, AM>(sa, tid + i * 256, ra[i]);
#pragma unroll
        for (int i = 0; i < B_ITEMS; ++i) gen_store<BN, BMODE>(sb, tid + i * 256, rb[i]);
    };

    if (nk > 0) { gload(0); lstore(0); }
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const bool more = kt + 1 < nk;
        if (more) gload(kt + 1);
        const uint8_t* ta = smem + (kt & 1) * STAGE;
        const uint8_t* tb = ta + A_BYTES;
        const uint8_t* sa = ta + (wm * (BM / 2) + lr) * RS;
        const uint8_t* sb = tb + (wn * 64 + lr) * RS;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 bf[4], af[MT];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if constexpr (BMODE & 2) bf[j] = tr_frag<BN>(tb, wn * 64 + j * 16, ks * 32 + lq * 8, lr);
                else bf[j] = *reinterpret_cast<const bf16x8*>(sb + j * 16 * RS + (((ks * 4 + lq) ^ (lr & 7)) * 16));
            }
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                if constexpr (AM & 2) af[i] = tr_frag<BM>(ta, wm * (BM / 2) + i * 16, ks * 32 + lq * 8, lr);
                else af[i] = *reinterpret_cast<const bf16x8*>(sa + i * 16 * RS + (((ks * 4 + lq) ^ (lr & 7)) * 16));
            }
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[j], af[i], acc[i][j], 0, 0, 0);
        }
        if (more) lstore((kt + 1) & 1);
        __syncthreads();
    }

    // epilogue: lane holds C[m = .. + lr][n = .. + 4*lq + 0..3]
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int m = m0 + wm * (BM / 2) + i * 16 + lr;
        if (m >= p.M) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + wn * 64 + j * 16 + lq * 4;
            if (n >= p.N) continue;                       // N % 4 == 0
            float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
            if (p.atomic && p.part) {
                *reinterpret_cast<float4*>(p.part + ((size_t)blockIdx.z * p.M + m) * p.N + n) = make_float4(v[0], v[1], v[2], v[3]);
                continue;
            }
            if (p.atomic) {
                float* c = (float*)p.C + (size_t)m * p.ldc + n;
                if (p.bias && blockIdx.z == 0) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] += p.bias[n + r];
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) atomicAdd(c + r, v[r]);
                continue;
            }
            if (p.bias) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] += p.bias[n + r];
            }
            if (p.act == 1) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = gelu_erf(v[r]);
            }
            if (p.res) {
                if (p.res_f32) {
                    const float4 rv = *reinterpret_cast<const float4*>((const float*)p.res + (size_t)m * p.ldres + n);
                    v[0] += rv.x; v[1] += rv.y; v[2] += rv.z; v[3] += rv.w;
                } else {
                    const uint2 rv = *reinterpret_cast<const uint2*>((const bf16_t*)p.res + (size_t)m * p.ldres + n);
                    v[0] += bf16lo_to_f32(rv.x); v[1] += bf16hi_to_f32(rv.x); v[2] += bf16lo_to_f32(rv.y); v[3] += bf16hi_to_f32(rv.y);
                }
            }
            if (p.out_f32) *reinterpret_cast<float4*>((float*)p.C + (size_t)m * p.ldc + n) = make_float4(v[0], v[1], v[2], v[3]);
            else *reinterpret_cast<uint2*>((bf16_t*)p.C + (size_t)m * p.ldc + n) = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
        }
    }
}

// ---- skinny M (<= 256 rows): one block per output tile, its waves cut K between them ("ks") -------------------------------------------
// The generator's mid ViT (vit.py:14-63: B*26 token rows against 512 ... 2048-wide weights) is 17 of these per forward.  On the tiled kernel
// above each was a split-K launch over the grid plus a reduction launch (10.7 + 5.3 us at 26 rows): the tile loop's per-step barrier pair and
// exposed round trip on a handful of k-steps per block.  Here a block owns a 32 x 32 output tile and 1 ... 16 waves each take a K range:
// the MFMA 16x16x32 operand of a lane IS a 16-byte load of its row (8 consecutive k at 8 (lane >> 4)), so nothing passes through LDS but
// the waves' partial tiles, summed in a fixed binary tree; wave 0 runs the epilogue (bias, exact-erf GELU, residual, bf16 / f32 store: the
// tiled kernel's).  The K cut depends on K alone: a row's sums are formed in the same order whatever M is -- a volume's tokens come out bit
// for bit the same in a batch of 1 and of 8 (tests/test_configs_gpu.py).
template <int NA, int NB>
__global__ __launch_bounds__(1024) void gemm_ks_kernel(const GemmParams p, const int kw) {
    extern __shared__ __attribute__((aligned(16))) uint8_t ks_smem[];
    f32x4* red = reinterpret_cast<f32x4*>(ks_smem);           // [slot][tile][lane], nw / 2 slots
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int lr = lane & 15, lq = lane >> 4;
    const int m0 = blockIdx.y * (16 * NA), n0 = blockIdx.x * (16 * NB);
    f32x4 acc[NA][NB];
#pragma unroll
    for (int i = 0; i < NA; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bf16_t* ap[NA];
    const bf16_t* bp[NB];
#pragma unroll
    for (int i = 0; i < NA; ++i) ap[i] = p.A + (size_t)min(m0 + 16 * i + lr, p.M - 1) * p.lda + 8 * lq;       // rows / columns past the end: clamped, never stored
#pragma unroll
    for (int j = 0; j < NB; ++j) bp[j] = p.B + (size_t)min(n0 + 16 * j + lr, p.N - 1) * p.ldb + 8 * lq;
    const int kbeg = w * kw, kend = min(p.K, kbeg + kw);
    constexpr int GB = 4;                                      // 32-deep k-steps fetched per batch
    for (int k = kbeg; k < kend; k += 32 * GB) {
        bf16x8 af[GB][NA], bf[GB][NB];
#pragma unroll
        for (int g = 0; g < GB; ++g) {
            const int kg = k + 32 * g;
            if (kg < kend) {                                   // wave-uniform
#pragma unroll
                for (int i = 0; i < NA; ++i) af[g][i] = *reinterpret_cast<const bf16x8*>(ap[i] + kg);
#pragma unroll
                for (int j = 0; j < NB; ++j) bf[g][j] = *reinterpret_cast<const bf16x8*>(bp[j] + kg);
            }
        }
#pragma unroll
        for (int g = 0; g < GB; ++g) {
            if (k + 32 * g < kend) {
#pragma unroll
                for (int i = 0; i < NA; ++i)
#pragma unroll
                    for (int j = 0; j < NB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[g][j], af[g][i], acc[i][j], 0, 0, 0);
            }
        }
    }
    for (int half = nw >> 1; half >= 1; half >>= 1) {          // fixed binary tree over the waves
        if (w >= half && w < 2 * half) {
#pragma unroll
            for (int i = 0; i < NA; ++i)
#pragma unroll
                for (int j = 0; j < NB; ++j) red[((w - half) * NA * NB + i * NB + j) * 64 + lane] = acc[i][j];
        }
        __syncthreads();
        if (w < half) {
#pragma unroll
            for (int i = 0; i < NA; ++i)
#pragma unroll
                for (int j = 0; j < NB; ++j) acc[i][j] += red[(w * NA * NB + i * NB + j) * 64 + lane];
        }
        if (half > 1) __syncthreads();
    }
    if (w != 0) return;
    // lane holds C[m = .. + lr][n = .. + 4 lq + 0..3]
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int m = m0 + 16 * i + lr;
        if (m >= p.M) continue;
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int n = n0 + 16 * j + 4 * lq;
            if (n >= p.N) continue;                           // N % 4 == 0
            float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
            if (p.bias) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] += p.bias[n + r];
            }
            if (p.act == 1) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = gelu_erf(v[r]);
            }
            if (p.res) {
                if (p.res_f32) {
                    const float4 rv = *reinterpret_cast<const float4*>((const float*)p.res + (size_t)m * p.ldres + n);
                    v[0] += rv.x; v[1] += rv.y; v[2] += rv.z; v[3] += rv.w;
                } else {
                    const uint2 rv = *reinterpret_cast<const uint2*>((const bf16_t*)p.res + (size_t)m * p.ldres + n);
                    v[0] += bf16lo_to_f32(rv.x); v[1] += bf16hi_to_f32(rv.x); v[2] += bf16lo_to_f32(rv.y); v[3] += bf16hi_to_f32(rv.y);
                }
            }
            if (p.out_f32) *reinterpret_cast<float4*>((float*)p.C + (size_t)m * p.ldc + n) = make_float4(v[0], v[1], v[2], v[3]);
            else *reinterpret_cast<uint2*>((bf16_t*)p.C + (size_t)m * p.ldc + n) = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
        }
    }
}

// the shape takes gemm_ks_kernel: <= 256 rows, plain bf16 operands, whole 32-deep k-steps, 16-byte rows, a K that the tiled kernel would
// have had to cut over the grid (K >= 256) and not the weight-streaming giants (the patch embedding's K = 147 456 keeps its grid-wide split)
bool gemm_ks_usable(const GemmParams& p) {
    static const bool off = getenv("GFE_GEMM_NO_KS") != nullptr;
    if (off || p.a_mode != 0 || p.b_mode != 0 || p.M > 256 || (p.K & 31) || p.K < 256 || p.K > 8192 || p.N > 8192) return false;   // (wide N: the un-patchify projection streams 151 MB of weights once per 256 rows on the DMA loop)
    if (((uintptr_t)p.A | (uintptr_t)p.B) & 15) return false;
    if (p.act != 0 && p.act != 1) return false;
    const int64_t esz = p.out_f32 ? 4 : 2;
    if ((((uintptr_t)p.C) % (4 * esz)) || ((p.ldc * esz) % (4 * esz))) return false;
    if (p.res && ((((uintptr_t)p.res) % (p.res_f32 ? 16 : 8)) || (p.ldres & 3))) return false;
    return true;
}
int gemm_ks_launch(const GemmParams& p, hipStream_t st) {
    const int groups = p.K / 32;
    int nw = 1;
    while (nw < 16 && 2 * nw <= groups) nw *= 2;               // a function of K alone (see above)
    const int kw = (int)ceil_div(groups, nw) * 32;
    hipLaunchKernelGGL((gemm_ks_kernel<2, 2>), dim3((unsigned)ceil_div(p.N, 32), (unsigned)ceil_div(p.M, 32)), dim3(64 * nw), (size_t)(nw / 2) * 4 * 1024, st, p, kw);
    return gfe_launch_status();
}

// out[c][r] = in[r][c]  (bf16), 64x64 tiles through LDS
__global__ __launch_bounds__(256) void transpose_bf16_kernel(const bf16_t* __restrict__ in, bf16_t* __restrict__ out,
                                                             int R, int Cc, int64_t ldi, int64_t ldo) {
    __shared__ bf16_t t[64][66];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const size_t bi = (size_t)blockIdx.z * R * ldi, bo = (size_t)blockIdx.z * Cc * ldo;
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        const int r = i >> 6, c = i & 63;
        t[r][c] = (r0 + r < R && c0 + c < Cc) ? in[bi + (size_t)(r0 + r) * ldi + c0 + c] : (bf16_t)0;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        const int c = i >> 6, r = i & 63;
        if (r0 + r < R && c0 + c < Cc) out[bo + (size_t)(c0 + c) * ldo + r0 + r] = t[r][c];
    }
}

// f32 -> bf16 (weights / activations) and bf16 -> f32
__global__ __launch_bounds__(256) void cast_kernel(const void* __restrict__ in, void* __restrict__ out, int64_t n, int to_bf16) {
    for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (int64_t)gridDim.x * 1024) {
        if (i + 4 <= n) {
            if (to_bf16) {
                const float4 v = *reinterpret_cast<const float4*>((const float*)in + i);
                *reinterpret_cast<uint2*>((bf16_t*)out + i) = make_uint2(pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w));
            } else {
                const uint2 v = *reinterpret_cast<const uint2*>((const bf16_t*)in + i);
                *reinterpret_cast<float4*>((float*)out + i) = make_float4(bf16lo_to_f32(v.x), bf16hi_to_f32(v.x), bf16lo_to_f32(v.y), bf16hi_to_f32(v.y));
            }
        } else {
            for (int64_t k = i; k < n; ++k) {
                if (to_bf16) ((bf16_t*)out)[k] = f32_to_bf16(((const float*)in)[k]);
                else ((float*)out)[k] = bf16_to_f32(((const bf16_t*)in)[k]);
            }
        }
    }
}

template <int BM, int AM, int BMODE>
int gemm_launch(const GemmParams& p, int nsplit, hipStream_t st) {
    constexpr size_t lds = 2 * (size_t)(BM + BN) * RS;
    static bool attr_set = false;
    if (!attr_set) { (void)hipFuncSetAttribute((const void*)gemm_nt_kernel<BM, AM, BMODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr_set = true; }
    const dim3 grid((unsigned)ceil_div(p.N, BN), (unsigned)ceil_div(p.M, BM), (unsigned)nsplit);
    hipLaunchKernelGGL((gemm_nt_kernel<BM, AM, BMODE>), grid, dim3(256), lds, st, p);
    return gfe_launch_status();
}

template <int AM, int BMODE>
int gemm_launch_bm(const GemmParams& p, int nsplit, hipStream_t st) {
    // weight-streaming shapes (the ViT's patch embedding and its inverse: a 151 MB weight against ~200 rows) re-read B once per row
    // block: 128-row tiles even if the last one is half empty
    const bool small = p.M <= 64 || (p.M % 128 != 0 && p.M % 128 <= 64 && p.M < 1024 && (int64_t)p.N * p.K < (1LL << 25));   // B below 64 MB
    return small ? gemm_launch<64, AM, BMODE>(p, nsplit, st) : gemm_launch<128, AM, BMODE>(p, nsplit, st);
}

template <int AM>
int gemm_launch_b(const GemmParams& p, int nsplit, hipStream_t st) {
    switch (p.b_mode) {
        case 0: return gemm_launch_bm<AM, 0>(p, nsplit, st);
        case 1: return gemm_launch_bm<AM, 1>(p, nsplit, st);
        case 2: return gemm_launch_bm<AM, 2>(p, nsplit, st);
        default: return gemm_launch_bm<AM, 3>(p, nsplit, st);
    }
}

// The split-K ranges summed in a fixed order (bit-reproducible, unlike the atomics it replaces), then the GEMM's epilogue:
//   plain f32 product (no act / res): C[m][n] += bias[n] + sum   (C was initialised by the caller: zeros or an accumulation target)
//   otherwise:                        C[m][n] = act(bias[n] + sum) + res[m][n], stored as f32 or bf16
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ part, void* __restrict__ Cv, const float* __restrict__ bias,
                                                            const void* __restrict__ res, int M, int N, int64_t ldc, int64_t ldres, int nsplit,
                                                            int out_f32, int res_f32, int act, int overwrite) {
    const int n4 = N >> 2;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < (int64_t)M * n4; i += (int64_t)gridDim.x * 256) {
        const int m = (int)(i / n4), n = (int)(i - (int64_t)m * n4) * 4;
        float4 s = bias ? *reinterpret_cast<const float4*>(bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
        // eight ranges' loads in flight at a time, added in range order (the same sums as one load per add: 64 ranges took 19 us as a chain of round trips)
        int z = 0;
        for (; z + 8 <= nsplit; z += 8) {
            float4 v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = *reinterpret_cast<const float4*>(part + ((size_t)(z + j) * M + m) * N + n);
#pragma unroll
            for (int j = 0; j < 8; ++j) { s.x += v[j].x; s.y += v[j].y; s.z += v[j].z; s.w += v[j].w; }
        }
        for (; z < nsplit; ++z) {
            const float4 v = *reinterpret_cast<const float4*>(part + ((size_t)z * M + m) * N + n);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        if (!overwrite) {
            float4* c = reinterpret_cast<float4*>((float*)Cv + (size_t)m * ldc + n);
            float4 o = *c;
            o.x += s.x; o.y += s.y; o.z += s.z; o.w += s.w;
            *c = o;
            continue;
        }
        if (act == 1) { s.x = gelu_erf(s.x); s.y = gelu_erf(s.y); s.z = gelu_erf(s.z); s.w = gelu_erf(s.w); }
        if (res) {
            if (res_f32) {
                const float4 rv = *reinterpret_cast<const float4*>((const float*)res + (size_t)m * ldres + n);
                s.x += rv.x; s.y += rv.y; s.z += rv.z; s.w += rv.w;
            } else {
                const uint2 rv = *reinterpret_cast<const uint2*>((const bf16_t*)res + (size_t)m * ldres + n);
                s.x += bf16lo_to_f32(rv.x); s.y += bf16hi_to_f32(rv.x); s.z += bf16lo_to_f32(rv.y); s.w += bf16hi_to_f32(rv.y);
            }
        }
        if (out_f32) *reinterpret_cast<float4*>((float*)Cv + (size_t)m * ldc + n) = s;
        else *reinterpret_cast<uint2*>((bf16_t*)Cv + (size_t)m * ldc + n) = make_uint2(pack_bf16x2(s.x, s.y), pack_bf16x2(s.z, s.w));
    }
}

int gemm_dispatch(GemmParams& p, int64_t split_k, hipStream_t st) {
    int64_t ks = ceil_div(ceil_div((int64_t)p.K, split_k), BK) * BK;
    const int nsplit = (int)ceil_div((int64_t)p.K, ks);
    p.ksplit = (int)ks; p.atomic = nsplit > 1;
    if (nsplit <= 1) p.part = nullptr;
    if (p.part && (p.ldc & 3)) return GFE_ERR_SHAPE;          // the reduction reads / writes C 16 bytes at a time
    // split_k > 1 with a plain f32 output means "ADD the K ranges into the C the caller initialised" (gfe_hip.h; nn_ops.gemm_ex(accum_into=)):
    // the in-block kernel used to overwrite C there;
    // the in-block kernel keeps that contract by taking C itself as its f32 residual: C = C + A.B (one owner lane per element)
    const bool accumulating = nsplit > 1 && p.out_f32 && p.act == 0 && p.res == nullptr;
    if (gemm_ks_usable(p)) {                                   // skinny M: no grid-wide split, no reduction launch
        if (accumulating) { p.res = p.C; p.res_f32 = 1; p.ldres = p.ldc; }
        p.atomic = 0; p.part = nullptr;
        return gemm_ks_launch(p, st);
    }
    if (nsplit == 1 && p.a_mode == 0 && p.b_mode == 0) {      // plain bf16 x bf16, K-major: the persistent LDS-DMA main loop (gemm_dma.hip)
        GemmDmaArgs d;
        d.A = p.A; d.B = p.B; d.C = p.C; d.bias = p.bias; d.res = p.res;
        d.lda = p.lda; d.ldb = p.ldb; d.ldc = p.ldc; d.ldres = p.ldres;
        d.M = p.M; d.N = p.N; d.K = p.K; d.out_f32 = p.out_f32; d.res_f32 = p.res_f32; d.act = p.act;
        if (gemm_dma_usable(d)) return gemm_dma_launch(d, st);
    }
    int rc;
    bool dma_split = false;
    if (nsplit > 1 && p.part && p.a_mode == 0 && p.b_mode == 0 && p.K >= 16384 && (int64_t)ks * nsplit == p.K) {
        // a weight-streaming product cut along K (the generator ViT's patch embedding: 200 x 512 x 147 456): the K ranges' tiles on the persistent
        // LDS-DMA main loop, one 256-row tile per (range, column tile) -- the weights cross L2 -> LDS once instead of once per 128-row tile through
        // registers; the same fixed-order reduction below.  Taken for ANY M (a row's sum order must not depend on the batch it rides in), and
        // BIT-IDENTICAL to the staged kernel for the same number of ranges (same k order inside a range, same MFMA shape: tools/gemm_split_equal.py,
        // test_gemm_weight_streaming_split_k_on_the_dma_main_loop) -- what moves the generator's rounding noise is the NUMBER of ranges, which the caller keeps.
        GemmDmaArgs d;
        d.A = p.A; d.B = p.B; d.C = p.C; d.bias = nullptr; d.res = nullptr;
        d.lda = p.lda; d.ldb = p.ldb; d.ldc = p.ldc; d.ldres = 0;
        d.M = p.M; d.N = p.N; d.K = p.K; d.out_f32 = 1; d.res_f32 = 0; d.act = 0;
        d.nsplit = nsplit; d.part = p.part;
        if (gemm_dma_usable(d)) { rc = gemm_dma_launch(d, st); dma_split = true; }
    }
    if (!dma_split)
    switch (p.a_mode) {
        case 0: rc = gemm_launch_b<0>(p, nsplit, st); break;
        case 1: rc = gemm_launch_b<1>(p, nsplit, st); break;
        case 2: rc = gemm_launch_b<2>(p, nsplit, st); break;
        default: rc = gemm_launch_b<3>(p, nsplit, st); break;
    }
    if (rc != GFE_OK || !p.part) return rc;
    int64_t blocks = ceil_div((int64_t)p.M * (p.N >> 2), 256); if (blocks > 2048) blocks = 2048;
    const int overwrite = (p.act != 0 || p.res != nullptr || !p.out_f32) ? 1 : 0;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, st, p.part, p.C, p.bias, p.res, p.M, p.N, p.ldc, p.ldres, nsplit,
                       p.out_f32, p.res_f32, p.act, overwrite);
    return gfe_launch_status();
}

}  // namespace

extern "C" {

int gfe_gemm_bf16_nt(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc,
                     int64_t M, int64_t N, int64_t K, const float* bias, const void* res, int64_t ldres, int res_f32,
                     int act, int out_f32, int split_k, float* splitk_ws, void* stream) {
    GFE_REQUIRE(A && B && C, GFE_ERR_NULL);
    GFE_REQUIRE(M > 0 && N > 0 && K > 0 && K % 8 == 0 && N % 4 == 0 && lda % 8 == 0 && ldb % 8 == 0, GFE_ERR_SHAPE);
    GFE_REQUIRE(M <= 0x7fffffff && N <= 0x7fffffff && K <= 0x7fffffff, GFE_ERR_SHAPE);
    GFE_REQUIRE(split_k >= 1 && (split_k == 1 || splitk_ws || (out_f32 && !res && act == 0)), GFE_ERR_SHAPE);
    GemmParams p;
    p.A = (const bf16_t*)A; p.B = (const bf16_t*)B; p.C = C; p.bias = bias; p.res = res;
    p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.ldres = ldres;
    p.M = (int)M; p.N = (int)N; p.K = (int)K;
    p.out_f32 = out_f32; p.res_f32 = res_f32; p.act = act; p.a_mode = p.b_mode = 0; p.part = splitk_ws;
    return gemm_dispatch(p, split_k, (hipStream_t)stream);
}

int gfe_gemm_ex(const void* A, int64_t lda, int a_mode, const void* B, int64_t ldb, int b_mode, void* C, int64_t ldc,
                int64_t M, int64_t N, int64_t K, const float* bias, const void* res, int64_t ldres, int res_f32,
                int act, int out_f32, int split_k, float* splitk_ws, void* stream) {
    GFE_REQUIRE(A && B && C, GFE_ERR_NULL);
    GFE_REQUIRE(M > 0 && N > 0 && K > 0 && N % 4 == 0, GFE_ERR_SHAPE);
    GFE_REQUIRE(M <= 0x7fffffff && N <= 0x7fffffff && K <= 0x7fffffff, GFE_ERR_SHAPE);
    GFE_REQUIRE((unsigned)a_mode < 4 && (unsigned)b_mode < 4, GFE_ERR_SHAPE);
    // 16-byte vector loads: ld a multiple of 8 (bf16) / 4 (f32); a K-major operand also needs K % 8 == 0
    GFE_REQUIRE(lda % ((a_mode & 1) ? 4 : 8) == 0 && ldb % ((b_mode & 1) ? 4 : 8) == 0, GFE_ERR_SHAPE);
    GFE_REQUIRE(((a_mode & 2) && (b_mode & 2)) || K % 8 == 0, GFE_ERR_SHAPE);
    GFE_REQUIRE(split_k >= 1 && (split_k == 1 || splitk_ws || (out_f32 && !res && act == 0)), GFE_ERR_SHAPE);
    GemmParams p;
    p.A = (const bf16_t*)A; p.B = (const bf16_t*)B; p.C = C; p.bias = bias; p.res = res;
    p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.ldres = ldres;
    p.M = (int)M; p.N = (int)N; p.K = (int)K;
    p.out_f32 = out_f32; p.res_f32 = res_f32; p.act = act; p.a_mode = a_mode; p.b_mode = b_mode; p.part = splitk_ws;
    return gemm_dispatch(p, split_k, (hipStream_t)stream);
}

int gfe_transpose_bf16(const void* in, void* out, int64_t batch, int64_t R, int64_t Cc, int64_t ldi, int64_t ldo, void* stream) {
    GFE_REQUIRE(in && out, GFE_ERR_NULL);
    GFE_REQUIRE(batch > 0 && batch <= 65535 && R > 0 && Cc > 0, GFE_ERR_SHAPE);
    hipLaunchKernelGGL(transpose_bf16_kernel, dim3((unsigned)ceil_div(Cc, 64), (unsigned)ceil_div(R, 64), (unsigned)batch), dim3(256), 0,
                       (hipStream_t)stream, (const bf16_t*)in, (bf16_t*)out, (int)R, (int)Cc, ldi, ldo);
    return gfe_launch_status();
}

int gfe_cast(const void* in, void* out, int64_t n, int to_bf16, void* stream) {
    GFE_REQUIRE(in && out, GFE_ERR_NULL);
    GFE_REQUIRE(n > 0, GFE_ERR_SHAPE);
    int64_t g = ceil_div(n, 1024);
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(cast_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, in, out, n, to_bf16);
    return gfe_launch_status();
}

}  // extern "C"
