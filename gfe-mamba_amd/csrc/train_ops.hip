// Training-step kernels that are pure HBM streams:
//   * Combine_classfier_vit_mid (classify/classifier.py:324-333): Linear(H*W -> S) applied per channel to cat(mid_input,
//     mid_output); here directly on the generator's channels-last bf16 mid features (no concat / transpose copy), forward
//     and weight gradient (the mid features carry no gradient: classify_mamba.py:100).
//   * per-PARAMETER gradient clip + Adam (classify_mamba.py:64, 106-108) on flat f32 buffers, one pass for all tensors,
//     also refreshing the bf16 copies of the weights the GEMMs read.
//   * transposing f32 -> bf16 copies that build the image condition (cross_atten/mamba_transformer.py:89-94).
#include "common.h"

namespace {

// ---- head Linear forward: out[b][src*C + c][s] += sum_{hw in chunk} mid_src[b][hw][c] * W[s][hw] --------------------
// grid (chunks, 2*B): blockIdx.y = b*2 + src.  64 lanes x 8 channels = 512 channels per wave pass; 4 waves split the rows.
template <int S>
__global__ __launch_bounds__(256) void mid_linear_fwd_kernel(const bf16_t* __restrict__ mid_in, const bf16_t* __restrict__ mid_out,
                                                             const float* __restrict__ W, float* __restrict__ out,
                                                             int HW, int C, int rows_per_block) {
    extern __shared__ float red[];                       // [4 waves][S][C]
    const int b = blockIdx.y >> 1, src = blockIdx.y & 1;
    const bf16_t* mid = (src ? mid_out : mid_in) + (size_t)b * HW * C;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r0 = blockIdx.x * rows_per_block, r1 = min(HW, r0 + rows_per_block);
    for (int c0 = 0; c0 < C; c0 += 512) {
        const int c = c0 + lane * 8;
        float acc[S][8];
#pragma unroll
        for (int s = 0; s < S; ++s)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[s][j] = 0.f;
        if (c < C) {
#pragma unroll 4
            for (int r = r0 + wave; r < r1; r += 4) {
                const uint4 v = *reinterpret_cast<const uint4*>(mid + (size_t)r * C + c);
                const float f[8] = {bf16lo_to_f32(v.x), bf16hi_to_f32(v.x), bf16lo_to_f32(v.y), bf16hi_to_f32(v.y),
                                    bf16lo_to_f32(v.z), bf16hi_to_f32(v.z), bf16lo_to_f32(v.w), bf16hi_to_f32(v.w)};
#pragma unroll
                for (int s = 0; s < S; ++s) {
                    const float w = W[(size_t)s * HW + r];          // wave-uniform
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[s][j] = fmaf(f[j], w, acc[s][j]);
                }
            }
        }
        __syncthreads();
        if (c < C) {
#pragma unroll
            for (int s = 0; s < S; ++s)
#pragma unroll
                for (int j = 0; j < 8; ++j) red[(wave * S + s) * 512 + lane * 8 + j] = acc[s][j];
        }
        __syncthreads();
        for (int i = threadIdx.x; i < S * 512; i += 256) {
            const int s = i / 512, cc = i - s * 512;
            if (c0 + cc < C) {
                const float t = red[(0 * S + s) * 512 + cc] + red[(1 * S + s) * 512 + cc] + red[(2 * S + s) * 512 + cc] + red[(3 * S + s) * 512 + cc];
                atomicAdd(out + ((size_t)b * 2 * C + src * C + c0 + cc) * S + s, t);
            }
        }
    }
}

// ---- head Linear weight gradient: dW[s][hw] = sum_{b,src,c} dout[b][src*C+c][s] * mid_src[b][hw][c] -------------------
// one wave per hw row; dout of one (b, src) staged in LDS as [S][C].
template <int S>
__global__ __launch_bounds__(256) void mid_linear_wgrad_kernel(const bf16_t* __restrict__ mid_in, const bf16_t* __restrict__ mid_out,
                                                               const float* __restrict__ dout, float* __restrict__ dW,
                                                               int B, int HW, int C, int rows_per_block) {
    extern __shared__ float sd[];                        // [S][C]
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r0 = blockIdx.x * rows_per_block, r1 = min(HW, r0 + rows_per_block);
    constexpr int RPW = 4;                               // rows per wave held in registers
    float acc[RPW][S];
    // rows of this wave: r0 + wave + 4*k
    for (int rb = r0; rb < r1; rb += 4 * RPW) {
#pragma unroll
        for (int k = 0; k < RPW; ++k)
#pragma unroll
            for (int s = 0; s < S; ++s) acc[k][s] = 0.f;
        for (int bs = 0; bs < 2 * B; ++bs) {
            const int b = bs >> 1, src = bs & 1;
            __syncthreads();
            for (int i = threadIdx.x; i < S * C; i += 256) {
                const int c = i / S, s = i - c * S;
                sd[s * C + c] = dout[((size_t)b * 2 * C + src * C + c) * S + s];
            }
            __syncthreads();
            const bf16_t* mid = (src ? mid_out : mid_in) + (size_t)b * HW * C;
#pragma unroll
            for (int k = 0; k < RPW; ++k) {
                const int r = rb + wave + 4 * k;
                if (r < r1) {
                    for (int c = lane * 8; c < C; c += 512) {
                        const uint4 v = *reinterpret_cast<const uint4*>(mid + (size_t)r * C + c);
                        const float f[8] = {bf16lo_to_f32(v.x), bf16hi_to_f32(v.x), bf16lo_to_f32(v.y), bf16hi_to_f32(v.y),
                                            bf16lo_to_f32(v.z), bf16hi_to_f32(v.z), bf16lo_to_f32(v.w), bf16hi_to_f32(v.w)};
#pragma unroll
                        for (int s = 0; s < S; ++s)
#pragma unroll
                            for (int j = 0; j < 8; ++j) acc[k][s] = fmaf(f[j], sd[s * C + c + j], acc[k][s]);
                    }
                }
            }
        }
#pragma unroll
        for (int k = 0; k < RPW; ++k) {
            const int r = rb + wave + 4 * k;
#pragma unroll
            for (int s = 0; s < S; ++s) {
                const float t = wave_sum(acc[k][s]);
                if (lane == 0 && r < r1) dW[(size_t)s * HW + r] = t;
            }
        }
    }
}

// ---- per-tensor clip + Adam on flat buffers -----------------------------------------------------------------------------
// chunk table (host-built, device-resident): {offset, length, tensor id} with chunks never crossing a tensor boundary.
struct Chunk { int64_t off; int len; int tid; };

__global__ __launch_bounds__(256) void sqnorm_kernel(const float* __restrict__ g, const Chunk* __restrict__ chunks, float* __restrict__ norm2, float gscale) {
    __shared__ float red[16];
    const Chunk ck = chunks[blockIdx.x];
    float s = 0.f;
    for (int i = threadIdx.x; i < ck.len; i += 256) { const float v = g[ck.off + i] * gscale; s = fmaf(v, v, s); }
    s = block_sum(s, red);
    if (threadIdx.x == 0) atomicAdd(norm2 + ck.tid, s);
}

__global__ __launch_bounds__(256) void clip_adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                        float* __restrict__ v, bf16_t* __restrict__ p16, const Chunk* __restrict__ chunks,
                                                        const float* __restrict__ norm2, float gscale, float max_norm, float lr, float b1, float b2,
                                                        float eps, float bc1, float bc2) {
    const Chunk ck = chunks[blockIdx.x];
    // torch.nn.utils.clip_grad_norm_ on ONE tensor: coef = max_norm / (norm + 1e-6), clamped to 1 (classify_mamba.py:106-107)
    const float coef = fminf(max_norm / (sqrtf(norm2[ck.tid]) + 1e-6f), 1.0f) * gscale;
    for (int i = threadIdx.x; i < ck.len; i += 256) {
        const int64_t k = ck.off + i;
        const float gi = g[k] * coef;
        const float mi = fmaf(b1, m[k], (1.f - b1) * gi);
        const float vi = fmaf(b2, v[k], (1.f - b2) * gi * gi);
        m[k] = mi; v[k] = vi;
        // torch.optim.Adam: p -= lr/bc1 * m / (sqrt(v)/sqrt(bc2) + eps)
        const float pn = p[k] - (lr / bc1) * mi / (sqrtf(vi) / sqrtf(bc2) + eps);
        p[k] = pn;
        if (p16) p16[k] = f32_to_bf16(pn);
    }
}

// ---- out[b][c][r] (bf16) = in[b][r][c] (f32): image (HW x D3) -> condition rows (D3 x HW) ------------------------------
__global__ __launch_bounds__(256) void transpose_f32_to_bf16_kernel(const float* __restrict__ in, bf16_t* __restrict__ out,
                                                                    int R, int Cc, int64_t in_batch, int64_t out_batch, int64_t ldo) {
    __shared__ float t[64][65];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const float* ib = in + (size_t)blockIdx.z * in_batch;
    bf16_t* ob = out + (size_t)blockIdx.z * out_batch;
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        const int r = i >> 6, c = i & 63;
        t[r][c] = (r0 + r < R && c0 + c < Cc) ? ib[(size_t)(r0 + r) * Cc + c0 + c] : 0.f;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        const int c = i >> 6, r = i & 63;
        if (r0 + r < R && c0 + c < Cc) ob[(size_t)(c0 + c) * ldo + r0 + r] = f32_to_bf16(t[r][c]);
    }
}

// out[r][b*ldb_cols + col_off + c] (bf16) = in[b][r][c] (f32): rows re-interleaved across the batch (cond^T for the wgrad)
__global__ __launch_bounds__(256) void interleave_rows_kernel(const float* __restrict__ in, bf16_t* __restrict__ out, int B, int R, int Cc,
                                                              int64_t ldo, int per_batch_cols, int col_off) {
    const int64_t total = (int64_t)B * R * Cc;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % Cc); const int64_t t = i / Cc;
        const int r = (int)(t % R), b = (int)(t / R);
        out[(size_t)r * ldo + (size_t)b * per_batch_cols + col_off + c] = f32_to_bf16(in[i]);
    }
}

// column sums of a row-major (M, N) f32 matrix (bias gradients): block = 256 threads covers 64 columns x 4 row-strips
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ x, float* __restrict__ out, int M, int N, int64_t ld, int accumulate) {
    __shared__ float red[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), strip = threadIdx.x >> 6;
    float acc = 0.f;
    if (c < N)
        for (int r = strip; r < M; r += 4) acc += x[(size_t)r * ld + c];
    red[strip][threadIdx.x & 63] = acc;
    __syncthreads();
    if (strip == 0 && c < N) {
        const float t = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
        out[c] = accumulate ? out[c] + t : t;
    }
}

}  // namespace

extern "C" {

int gfe_colsum_f32(const float* x, float* out, int64_t M, int64_t N, int64_t ld, int accumulate, void* stream) {
    GFE_REQUIRE(x && out, GFE_ERR_NULL);
    GFE_REQUIRE(M > 0 && N > 0 && M <= 0x7fffffff && N <= 0x7fffffff, GFE_ERR_SHAPE);
    hipLaunchKernelGGL(colsum_kernel, dim3((unsigned)ceil_div(N, 64)), dim3(256), 0, (hipStream_t)stream, x, out, (int)M, (int)N, ld, accumulate);
    return gfe_launch_status();
}

int gfe_mid_linear_fwd(const void* mid_in, const void* mid_out, const float* W, float* out_zeroed,
                       int64_t B, int64_t HW, int64_t C, int64_t S, void* stream) {
    GFE_REQUIRE(mid_in && mid_out && W && out_zeroed, GFE_ERR_NULL);
    GFE_REQUIRE(B > 0 && HW > 0 && C % 8 == 0 && S == 4, GFE_ERR_SHAPE);
    int rpb = (int)ceil_div(HW, 128);
    if (rpb < 16) rpb = 16;
    const dim3 grid((unsigned)ceil_div(HW, rpb), (unsigned)(2 * B));
    hipLaunchKernelGGL((mid_linear_fwd_kernel<4>), grid, dim3(256), 4 * 4 * 512 * sizeof(float), (hipStream_t)stream,
                       (const bf16_t*)mid_in, (const bf16_t*)mid_out, W, out_zeroed, (int)HW, (int)C, rpb);
    return gfe_launch_status();
}

int gfe_mid_linear_wgrad(const void* mid_in, const void* mid_out, const float* dout, float* dW,
                         int64_t B, int64_t HW, int64_t C, int64_t S, void* stream) {
    GFE_REQUIRE(mid_in && mid_out && dout && dW, GFE_ERR_NULL);
    GFE_REQUIRE(B > 0 && HW > 0 && C % 8 == 0 && S == 4 && S * C * sizeof(float) <= 64 * 1024, GFE_ERR_SHAPE);
    const int rpb = 16;
    hipLaunchKernelGGL((mid_linear_wgrad_kernel<4>), dim3((unsigned)ceil_div(HW, rpb)), dim3(256), (size_t)S * C * sizeof(float),
                       (hipStream_t)stream, (const bf16_t*)mid_in, (const bf16_t*)mid_out, dout, dW, (int)B, (int)HW, (int)C, rpb);
    return gfe_launch_status();
}

int gfe_clip_adam(float* p, const float* g, float* m, float* v, void* p_bf16, const void* chunks, int64_t nchunks,
                  float* norm2_zeroed, float grad_scale, float max_norm, float lr, float beta1, float beta2, float eps,
                  int64_t step, void* stream) {
    GFE_REQUIRE(p && g && m && v && chunks && norm2_zeroed, GFE_ERR_NULL);
    GFE_REQUIRE(nchunks > 0 && nchunks <= 0x7fffffff && step >= 1, GFE_ERR_SHAPE);
    const float bc1 = 1.0f - powf(beta1, (float)step), bc2 = 1.0f - powf(beta2, (float)step);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(sqnorm_kernel, dim3((unsigned)nchunks), dim3(256), 0, st, g, (const Chunk*)chunks, norm2_zeroed, grad_scale);
    hipLaunchKernelGGL(clip_adam_kernel, dim3((unsigned)nchunks), dim3(256), 0, st, p, g, m, v, (bf16_t*)p_bf16, (const Chunk*)chunks,
                       norm2_zeroed, grad_scale, max_norm, lr, beta1, beta2, eps, bc1, bc2);
    return gfe_launch_status();
}

int gfe_transpose_f32_to_bf16(const float* in, void* out, int64_t batch, int64_t R, int64_t Cc, int64_t out_batch_stride, int64_t ldo, void* stream) {
    GFE_REQUIRE(in && out, GFE_ERR_NULL);
    GFE_REQUIRE(batch > 0 && batch <= 65535 && R > 0 && Cc > 0, GFE_ERR_SHAPE);
    hipLaunchKernelGGL(transpose_f32_to_bf16_kernel, dim3((unsigned)ceil_div(Cc, 64), (unsigned)ceil_div(R, 64), (unsigned)batch), dim3(256), 0,
                       (hipStream_t)stream, in, (bf16_t*)out, (int)R, (int)Cc, R * Cc, out_batch_stride, ldo);
    return gfe_launch_status();
}

int gfe_interleave_rows_bf16(const float* in, void* out, int64_t B, int64_t R, int64_t Cc, int64_t ldo, int64_t per_batch_cols,
                             int64_t col_off, void* stream) {
    GFE_REQUIRE(in && out, GFE_ERR_NULL);
    GFE_REQUIRE(B > 0 && R > 0 && Cc > 0, GFE_ERR_SHAPE);
    int64_t g = ceil_div(B * R * Cc, 256);
    if (g > 8192) g = 8192;
    hipLaunchKernelGGL(interleave_rows_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, in, (bf16_t*)out, (int)B, (int)R, (int)Cc,
                       ldo, (int)per_batch_cols, (int)col_off);
    return gfe_launch_status();
}

}  // extern "C"
