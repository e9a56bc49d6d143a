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
#include "common.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

namespace {

constexpr int BN = 128, BK = 64, RS = 128;     // RS: LDS row stride in bytes (no padding; 16-B chunk c of row r lives at slot c ^ (r & 7))

struct GemmParams {
    const bf16_t* A; const bf16_t* B; void* C; const float* bias; const void* res;
    int64_t lda, ldb, ldc, ldres;
    int M, N, K, ksplit;      // ksplit: K range per blockIdx.z (multiple of BK)
    int out_f32, res_f32, act, atomic;
};

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }

template <int BM>
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

    uint4 ra[A_ITEMS], rb[B_ITEMS];
    auto gload = [&](int kt) {
        const int k0 = kbeg + kt * BK;
#pragma unroll
        for (int i = 0; i < A_ITEMS; ++i) {
            const int it = tid + i * 256, row = it >> 3, c = it & 7;
            const int m = m0 + row, k = k0 + c * 8;
            ra[i] = (m < p.M && k < kend) ? *reinterpret_cast<const uint4*>(p.A + (size_t)m * p.lda + k) : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < B_ITEMS; ++i) {
            const int it = tid + i * 256, row = it >> 3, c = it & 7;
            const int n = n0 + row, k = k0 + c * 8;
            rb[i] = (n < p.N && k < kend) ? *reinterpret_cast<const uint4*>(p.B + (size_t)n * p.ldb + k) : make_uint4(0, 0, 0, 0);
        }
    };
    auto lstore = [&](int buf) {
        uint8_t* sa = smem + buf * STAGE;
        uint8_t* sb = sa + A_BYTES;
#pragma unroll
        for (int i = 0; i < A_ITEMS; ++i) { const int it = tid + i * 256; *reinterpret_cast<uint4*>(sa + (it >> 3) * RS + (((it & 7) ^ ((it >> 3) & 7)) * 16)) = ra[i]; }
#pragma unroll
        for (int i = 0; i < B_ITEMS; ++i) { const int it = tid + i * 256; *reinterpret_cast<uint4*>(sb + (it >> 3) * RS + (((it & 7) ^ ((it >> 3) & 7)) * 16)) = rb[i]; }
    };

    if (nk > 0) { gload(0); lstore(0); }
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const bool more = kt + 1 < nk;
        if (more) gload(kt + 1);
        const uint8_t* sa = smem + (kt & 1) * STAGE + (wm * (BM / 2) + lr) * RS;
        const uint8_t* sb = smem + (kt & 1) * STAGE + A_BYTES + (wn * 64 + lr) * RS;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 bf[4], af[MT];
#pragma unroll
            for (int j = 0; j < 4; ++j) bf[j] = *reinterpret_cast<const bf16x8*>(sb + j * 16 * RS + (((ks * 4 + lq) ^ (lr & 7)) * 16));
#pragma unroll
            for (int i = 0; i < MT; ++i) af[i] = *reinterpret_cast<const bf16x8*>(sa + i * 16 * RS + (((ks * 4 + lq) ^ (lr & 7)) * 16));
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

template <int BM>
int gemm_launch(const GemmParams& p, int nsplit, hipStream_t st) {
    constexpr size_t lds = 2 * (size_t)(BM + BN) * RS;
    static bool attr_set = false;
    if (!attr_set) { (void)hipFuncSetAttribute((const void*)gemm_nt_kernel<BM>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr_set = true; }
    const dim3 grid((unsigned)ceil_div(p.N, BN), (unsigned)ceil_div(p.M, BM), (unsigned)nsplit);
    hipLaunchKernelGGL((gemm_nt_kernel<BM>), grid, dim3(256), lds, st, p);
    return gfe_launch_status();
}

}  // namespace

extern "C" {

int gfe_gemm_bf16_nt(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc,
                     int64_t M, int64_t N, int64_t K, const float* bias, const void* res, int64_t ldres, int res_f32,
                     int act, int out_f32, int split_k, void* stream) {
    GFE_REQUIRE(A && B && C, GFE_ERR_NULL);
    GFE_REQUIRE(M > 0 && N > 0 && K > 0 && K % 8 == 0 && N % 4 == 0 && lda % 8 == 0 && ldb % 8 == 0, GFE_ERR_SHAPE);
    GFE_REQUIRE(M <= 0x7fffffff && N <= 0x7fffffff && K <= 0x7fffffff, GFE_ERR_SHAPE);
    GFE_REQUIRE(split_k >= 1 && (split_k == 1 || (out_f32 && !res && act == 0)), GFE_ERR_SHAPE);
    GemmParams p;
    p.A = (const bf16_t*)A; p.B = (const bf16_t*)B; p.C = C; p.bias = bias; p.res = res;
    p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.ldres = ldres;
    p.M = (int)M; p.N = (int)N; p.K = (int)K;
    int64_t ks = ceil_div(ceil_div(K, split_k), BK) * BK;
    const int nsplit = (int)ceil_div(K, ks);
    p.ksplit = (int)ks; p.out_f32 = out_f32; p.res_f32 = res_f32; p.act = act; p.atomic = nsplit > 1;
    hipStream_t st = (hipStream_t)stream;
    if (M <= 64 || (M % 128 != 0 && M % 128 <= 64 && M < 1024)) return gemm_launch<64>(p, nsplit, st);
    return gemm_launch<128>(p, nsplit, st);
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
