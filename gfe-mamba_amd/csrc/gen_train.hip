// Backward-pass helpers of the generator (conv3d backward / generator training step, main_gan_vit.py:68-82).
//
// The heavy products of the backward run on the kernels the forward already has -- dgrad of a 3x3x3 conv is the same implicit GEMM with
// flipped taps and transposed weights (conv_igemm), wgrad is a voxel-reduction GEMM (gemm_nt, reduction-major operands) -- so what is
// left are the HBM-bound elementwise / reduction passes around them, on channels-last bf16 tensors (B, V, C), 8 channels (16 B) per lane:
//   gn_apply        x^ = scale[b,c] * x + shift[b,c]                 GroupNorm output materialised (the forward folds it into the weights; the
//                                                                    weight gradient needs it as the conv's input)   buildingblocks.py:55-67
//   mask_relu       dy * (y > 0)                                     ReLU backward from the stored output
//   gn_bwd_sums     S1[b,c] = sum_v dx^, S2[b,c] = sum_v dx^ * xn    xn = (x - mu) * rstd
//   gn_bwd_apply    dx = rstd * (gamma * dx^ - a[b,g] - xn * b[b,g]) (+ a second gradient flowing into the same tensor)
//   maxpool2_bwd    routes dy to the arg-max of every 2x2x2 window   buildingblocks.py:284 (nn.MaxPool3d(2))
//   out1_bwd        final 1x1x1 conv C -> 1: dx = dy * w, dw, db     model.py:123, 162
//   in1_wgrad       first 1x1x1 lift 1 -> C: dw = sum dr * x, db     buildingblocks.py:191-198
//   colsum_bf16     bias gradient of a 1x1x1 lift: sum over voxels
#include "common.h"

namespace {

__device__ __forceinline__ void unpack8(const uint4& v, float (&o)[8]) {
    o[0] = bf16lo_to_f32(v.x); o[1] = bf16hi_to_f32(v.x); o[2] = bf16lo_to_f32(v.y); o[3] = bf16hi_to_f32(v.y);
    o[4] = bf16lo_to_f32(v.z); o[5] = bf16hi_to_f32(v.z); o[6] = bf16lo_to_f32(v.w); o[7] = bf16hi_to_f32(v.w);
}
__device__ __forceinline__ uint4 pack8(const float (&o)[8]) {
    return make_uint4(pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]), pack_bf16x2(o[4], o[5]), pack_bf16x2(o[6], o[7]));
}

// items = B * V * C / 8; item i -> (b, v, c8)
__global__ __launch_bounds__(256) void gn_apply_kernel(const bf16_t* __restrict__ x, const float* __restrict__ scale, const float* __restrict__ shift,
                                                       bf16_t* __restrict__ y, int64_t V, int C, int64_t items) {
    const int c8n = C / 8;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < items; i += (int64_t)gridDim.x * 256) {
        const int c0 = (int)(i % c8n) * 8;
        const int64_t b = i / ((int64_t)c8n * V);
        float f[8];
        unpack8(reinterpret_cast<const uint4*>(x)[i], f);
#pragma unroll
        for (int k = 0; k < 8; ++k) f[k] = fmaf(f[k], scale[b * C + c0 + k], shift[b * C + c0 + k]);
        reinterpret_cast<uint4*>(y)[i] = pack8(f);
    }
}

__global__ __launch_bounds__(256) void mask_relu_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ y, bf16_t* __restrict__ out, int64_t items) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < items; i += (int64_t)gridDim.x * 256) {
        float g[8], o[8];
        unpack8(reinterpret_cast<const uint4*>(dy)[i], g);
        unpack8(reinterpret_cast<const uint4*>(y)[i], o);
#pragma unroll
        for (int k = 0; k < 8; ++k) g[k] = o[k] > 0.f ? g[k] : 0.f;
        reinterpret_cast<uint4*>(out)[i] = pack8(g);
    }
}

// The three reductions of this file (GroupNorm backward sums, the final conv's dw / db, the first lift's dw / db) share one shape: a block's
// threads = (channel octet, row lane) hold 8 + 8 partial sums each; round 5 (the generator-training path was f32 atomics
// throughout, so row f-1 was not run-to-run reproducible): the row lanes park their partials in LDS slots of their own and ONE thread per
// column folds them in lane order; the block's result goes to its own slot of a caller workspace (ws != NULL) and gt_fold_kernel adds the
// blocks in order -- no atomic anywhere.  ws == NULL keeps the old behaviour (f32 atomics onto the zeroed / accumulating target).
__device__ __forceinline__ float gt_block_fold(const float* sm, int ncol, int vlanes, int i) {
    float t = 0.f;
    for (int v = 0; v < vlanes; ++v) t += sm[v * ncol + i];
    return t;
}
// out[b][i] += sum over blocks of ws[(b * nblk + blk) * ncol + i]; columns >= split go to out2 (two target arrays: S1 | S2, dw | db)
__global__ __launch_bounds__(256) void gt_fold_kernel(const float* __restrict__ ws, float* __restrict__ out1, float* __restrict__ out2, int nblk, int ncol, int split,
                                                      int64_t ld1, int64_t ld2) {
    const int b = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    if (i >= ncol) return;
    const float* p = ws + (size_t)b * nblk * ncol + i;
    float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;
    int k = 0;
    for (; k + 3 < nblk; k += 4) { t0 += p[(size_t)k * ncol]; t1 += p[(size_t)(k + 1) * ncol]; t2 += p[(size_t)(k + 2) * ncol]; t3 += p[(size_t)(k + 3) * ncol]; }
    for (; k < nblk; ++k) t0 += p[(size_t)k * ncol];
    const float t = (t0 + t1) + (t2 + t3);
    if (i < split) { if (out1) out1[(size_t)b * ld1 + i] += t; }
    else if (out2) out2[(size_t)b * ld2 + i - split] += t;
}

// grid (voxel chunks, B); block 256: thread -> channel octet tid % (C/8), voxel lane tid / (C/8)
__global__ __launch_bounds__(256) void gn_bwd_sums_kernel(const bf16_t* __restrict__ dxh, const bf16_t* __restrict__ x, const float* __restrict__ mu,
                                                          const float* __restrict__ rstd, float* __restrict__ S1, float* __restrict__ S2,
                                                          int64_t V, int C, int64_t vchunk, float* __restrict__ ws) {
    extern __shared__ float sm[];                 // [vlanes][2 C]
    const int c8n = C / 8, b = blockIdx.y;
    const int oct = threadIdx.x % c8n, vl = threadIdx.x / c8n, vlanes = 256 / c8n;
    if (vl < vlanes) {
        float s1[8] = {0, 0, 0, 0, 0, 0, 0, 0}, s2[8] = {0, 0, 0, 0, 0, 0, 0, 0}, m[8], r[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) { m[k] = mu[(size_t)b * C + oct * 8 + k]; r[k] = rstd[(size_t)b * C + oct * 8 + k]; }
        const int64_t v0 = (int64_t)blockIdx.x * vchunk, v1 = min(V, v0 + vchunk);
        for (int64_t v = v0 + vl; v < v1; v += vlanes) {
            const size_t it = ((size_t)b * V + v) * c8n + oct;
            float g[8], xv[8];
            unpack8(reinterpret_cast<const uint4*>(dxh)[it], g);
            unpack8(reinterpret_cast<const uint4*>(x)[it], xv);
#pragma unroll
            for (int k = 0; k < 8; ++k) { s1[k] += g[k]; s2[k] = fmaf(g[k], (xv[k] - m[k]) * r[k], s2[k]); }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) { sm[vl * 2 * C + oct * 8 + k] = s1[k]; sm[vl * 2 * C + C + oct * 8 + k] = s2[k]; }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * C; i += 256) {
        const float t = gt_block_fold(sm, 2 * C, vlanes, i);
        if (ws) ws[((size_t)b * gridDim.x + blockIdx.x) * 2 * C + i] = t;
        else atomicAdd((i < C ? S1 : S2) + (size_t)b * C + (i < C ? i : i - C), t);
    }
}

__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(const bf16_t* __restrict__ dxh, const bf16_t* __restrict__ x, const float* __restrict__ mu,
                                                           const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                           const float* __restrict__ ca, const float* __restrict__ cb, const bf16_t* __restrict__ add_in,
                                                           bf16_t* __restrict__ dx, int64_t V, int C, int64_t items) {
    const int c8n = C / 8;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < items; i += (int64_t)gridDim.x * 256) {
        const int c0 = (int)(i % c8n) * 8;
        const int64_t b = i / ((int64_t)c8n * V);
        float g[8], xv[8], a[8];
        unpack8(reinterpret_cast<const uint4*>(dxh)[i], g);
        unpack8(reinterpret_cast<const uint4*>(x)[i], xv);
        if (add_in) unpack8(reinterpret_cast<const uint4*>(add_in)[i], a);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const size_t j = (size_t)b * C + c0 + k;
            const float xn = (xv[k] - mu[j]) * rstd[j];
            float d = rstd[j] * (gamma[c0 + k] * g[k] - ca[j] - xn * cb[j]);
            if (add_in) d += a[k];
            g[k] = d;
        }
        reinterpret_cast<uint4*>(dx)[i] = pack8(g);
    }
}

// one thread per (output voxel, channel octet): the first maximum of the window in (d, h, w) scan order gets the gradient (ATen rule)
__global__ __launch_bounds__(256) void maxpool2_bwd_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy, bf16_t* __restrict__ dx,
                                                           int D, int H, int W, int C, int64_t items) {
    const int c8n = C / 8, od = D / 2, oh = H / 2, ow = W / 2;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < items; i += (int64_t)gridDim.x * 256) {
        int64_t t = i;
        const int oct = (int)(t % c8n); t /= c8n;
        const int w = (int)(t % ow); t /= ow;
        const int h = (int)(t % oh); t /= oh;
        const int d = (int)(t % od); const int64_t b = t / od;
        float g[8], best[8], out[8][8];
        int arg[8];
        unpack8(reinterpret_cast<const uint4*>(dy)[i], g);
#pragma unroll
        for (int k = 0; k < 8; ++k) { best[k] = -INFINITY; arg[k] = 0; }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const size_t src = ((((size_t)b * D + 2 * d + (j >> 2)) * H + 2 * h + ((j >> 1) & 1)) * W + 2 * w + (j & 1)) * c8n + oct;
            float xv[8];
            unpack8(reinterpret_cast<const uint4*>(x)[src], xv);
#pragma unroll
            for (int k = 0; k < 8; ++k) if (j == 0 || xv[k] > best[k]) { best[k] = xv[k]; arg[k] = j; }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
#pragma unroll
            for (int k = 0; k < 8; ++k) out[j][k] = arg[k] == j ? g[k] : 0.f;
            const size_t dst = ((((size_t)b * D + 2 * d + (j >> 2)) * H + 2 * h + ((j >> 1) & 1)) * W + 2 * w + (j & 1)) * c8n + oct;
            reinterpret_cast<uint4*>(dx)[dst] = pack8(out[j]);
        }
    }
}

// final conv C -> 1: x (rows, C) bf16, dy (rows) f32, w (C): dx = dy * w (bf16); dw[c] += sum dy * x; db += sum dy
__global__ __launch_bounds__(256) void out1_bwd_kernel(const bf16_t* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ w,
                                                       bf16_t* __restrict__ dx, float* __restrict__ dw, float* __restrict__ db, int64_t rows, int C, int64_t rchunk,
                                                       float* __restrict__ ws) {
    extern __shared__ float sm[];                 // [rlanes][C + 8]  (column C: the row lane's sum of dy)
    const int c8n = C / 8, NC = C + 8;
    const int oct = threadIdx.x % c8n, rl = threadIdx.x / c8n, rlanes = 256 / c8n;
    if (rl < rlanes) {
        float wv[8], acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, sb = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) wv[k] = w[oct * 8 + k];
        const int64_t r0 = (int64_t)blockIdx.x * rchunk, r1 = min(rows, r0 + rchunk);
        for (int64_t r = r0 + rl; r < r1; r += rlanes) {
            const float g = dy[r];
            float xv[8], o[8];
            unpack8(reinterpret_cast<const uint4*>(x)[(size_t)r * c8n + oct], xv);
#pragma unroll
            for (int k = 0; k < 8; ++k) { o[k] = g * wv[k]; acc[k] = fmaf(g, xv[k], acc[k]); }
            reinterpret_cast<uint4*>(dx)[(size_t)r * c8n + oct] = pack8(o);
            if (oct == 0) sb += g;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) sm[rl * NC + oct * 8 + k] = acc[k];
        if (oct == 0) sm[rl * NC + C] = sb;
    }
    __syncthreads();
    for (int i = threadIdx.x; i <= C; i += 256) {
        const float t = gt_block_fold(sm, NC, rlanes, i);
        if (ws) ws[(size_t)blockIdx.x * (C + 1) + i] = t;
        else atomicAdd(i < C ? dw + i : db, t);
    }
}

// first lift 1 -> C: x (rows) f32, dr (rows, C) bf16: dw[c] += sum dr * x, db[c] += sum dr
__global__ __launch_bounds__(256) void in1_wgrad_kernel(const float* __restrict__ x, const bf16_t* __restrict__ dr, float* __restrict__ dw, float* __restrict__ db,
                                                        int64_t rows, int C, int64_t rchunk, float* __restrict__ ws) {
    extern __shared__ float sm[];                 // [rlanes][2 C]
    const int c8n = C / 8;
    const int oct = threadIdx.x % c8n, rl = threadIdx.x / c8n, rlanes = 256 / c8n;
    if (rl < rlanes) {
        float a[8] = {0, 0, 0, 0, 0, 0, 0, 0}, s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        const int64_t r0 = (int64_t)blockIdx.x * rchunk, r1 = min(rows, r0 + rchunk);
        for (int64_t r = r0 + rl; r < r1; r += rlanes) {
            const float xv = x ? x[r] : 1.f;
            float g[8];
            unpack8(reinterpret_cast<const uint4*>(dr)[(size_t)r * c8n + oct], g);
#pragma unroll
            for (int k = 0; k < 8; ++k) { a[k] = fmaf(g[k], xv, a[k]); s[k] += g[k]; }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) { sm[rl * 2 * C + oct * 8 + k] = a[k]; sm[rl * 2 * C + C + oct * 8 + k] = s[k]; }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * C; i += 256) {
        const float t = gt_block_fold(sm, 2 * C, rlanes, i);
        if (ws) ws[(size_t)blockIdx.x * 2 * C + i] = t;
        else if (i < C) { if (dw) atomicAdd(dw + i, t); }
        else atomicAdd(db + i - C, t);
    }
}

static unsigned grid_for(int64_t items) { int64_t g = ceil_div(items, 256); return (unsigned)(g > 8192 ? 8192 : g); }

// nn.L1Loss()(pred, target) (main_gan_vit.py:72 -- the generator's reconstruction loss) with its gradient in one pass: block i sums its 4 096-element chunk
// (every thread its own elements in order, a fixed tree over the block) into part[i] and writes dpred = sign(pred - target) / n; l1_fold_kernel adds the
// blocks' partials in order.  No atomics: the loss is run-to-run bit-identical.
__global__ __launch_bounds__(256) void l1_loss_kernel(const float* __restrict__ pred, const float* __restrict__ target, float* __restrict__ part,
                                                      float* __restrict__ dpred, int64_t n, float inv_n) {
    __shared__ float red[32];
    const int64_t base = (int64_t)blockIdx.x * 4096;
    float s = 0.f;
#pragma unroll 4
    for (int k = 0; k < 16; ++k) {
        const int64_t i = base + threadIdx.x + 256 * k;
        if (i < n) {
            const float d = pred[i] - target[i];
            s += fabsf(d);
            dpred[i] = d > 0.f ? inv_n : (d < 0.f ? -inv_n : 0.f);
        }
    }
    s = block_sum(s, red);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
}
__global__ __launch_bounds__(256) void l1_fold_kernel(const float* __restrict__ part, float* __restrict__ loss, int nblk, float inv_n) {
    __shared__ float red[32];
    float s = 0.f;
    for (int i = threadIdx.x; i < nblk; i += 256) s += part[i];
    s = block_sum(s, red);
    if (threadIdx.x == 0) loss[0] = s * inv_n;
}

}  // namespace

extern "C" {

int gfe_l1_loss_blocks(int64_t n) { return (int)ceil_div(n, 4096); }

int gfe_l1_loss(const float* pred, const float* target, float* loss, float* dpred, float* part_ws, int64_t n, void* stream) {
    GFE_REQUIRE(pred && target && loss && dpred && part_ws, GFE_ERR_NULL);
    GFE_REQUIRE(n > 0 && ceil_div(n, 4096) <= 0x7fffffff, GFE_ERR_SHAPE);
    const int nblk = gfe_l1_loss_blocks(n);
    const float inv_n = 1.0f / (float)n;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(l1_loss_kernel, dim3((unsigned)nblk), dim3(256), 0, st, pred, target, part_ws, dpred, n, inv_n);
    hipLaunchKernelGGL(l1_fold_kernel, dim3(1), dim3(256), 0, st, part_ws, loss, nblk, inv_n);
    return gfe_launch_status();
}

int gfe_gn_apply(const void* x, const float* scale, const float* shift, void* y, int64_t B, int64_t V, int64_t C, void* stream) {
    GFE_REQUIRE(x && scale && shift && y, GFE_ERR_NULL);
    GFE_REQUIRE(B > 0 && V > 0 && C > 0 && C % 8 == 0, GFE_ERR_SHAPE);
    const int64_t items = B * V * C / 8;
    hipLaunchKernelGGL(gn_apply_kernel, dim3(grid_for(items)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, scale, shift, (bf16_t*)y, V, (int)C, items);
    return gfe_launch_status();
}

int gfe_mask_relu_bf16(const void* dy, const void* y, void* out, int64_t n, void* stream) {
    GFE_REQUIRE(dy && y && out, GFE_ERR_NULL);
    GFE_REQUIRE(n > 0 && n % 8 == 0, GFE_ERR_SHAPE);
    hipLaunchKernelGGL(mask_relu_kernel, dim3(grid_for(n / 8)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dy, (const bf16_t*)y, (bf16_t*)out, n / 8);
    return gfe_launch_status();
}

static int64_t gt_blocks(int64_t n, int64_t per, int64_t cap) {
    int64_t chunks = ceil_div(n, per); if (chunks > cap) chunks = cap;
    return ceil_div(n, ceil_div(n, chunks));
}
int gfe_gn_bwd_sums_blocks(int64_t V) { return V > 0 ? (int)gt_blocks(V, 2048, 1024) : 0; }
int gfe_gen_rows_blocks(int64_t rows) { return rows > 0 ? (int)gt_blocks(rows, 4096, 2048) : 0; }

int gfe_gn_bwd_sums(const void* dxhat, const void* x, const float* mu, const float* rstd, float* S1_zeroed, float* S2_zeroed, float* ws,
                    int64_t B, int64_t V, int64_t C, void* stream) {
    GFE_REQUIRE(dxhat && x && mu && rstd && S1_zeroed && S2_zeroed, GFE_ERR_NULL);
    GFE_REQUIRE(B > 0 && B <= 65535 && V > 0 && C > 0 && C % 8 == 0 && C <= 2048 && 256 % (C / 8) == 0, GFE_ERR_SHAPE);
    const int64_t nblk = gt_blocks(V, 2048, 1024), vchunk = ceil_div(V, nblk);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(gn_bwd_sums_kernel, dim3((unsigned)nblk, (unsigned)B), dim3(256), (size_t)(256 / (C / 8)) * 2 * C * sizeof(float), st,
                       (const bf16_t*)dxhat, (const bf16_t*)x, mu, rstd, S1_zeroed, S2_zeroed, V, (int)C, vchunk, ws);
    if (ws) hipLaunchKernelGGL(gt_fold_kernel, dim3((unsigned)ceil_div(2 * C, 256), (unsigned)B), dim3(256), 0, st, ws, S1_zeroed, S2_zeroed, (int)nblk, (int)(2 * C), (int)C, C, C);
    return gfe_launch_status();
}

int gfe_gn_bwd_apply(const void* dxhat, const void* x, const float* mu, const float* rstd, const float* gamma, const float* coef_a, const float* coef_b,
                     const void* add_in, void* dx, int64_t B, int64_t V, int64_t C, void* stream) {
    GFE_REQUIRE(dxhat && x && mu && rstd && gamma && coef_a && coef_b && dx, GFE_ERR_NULL);
    GFE_REQUIRE(B > 0 && V > 0 && C > 0 && C % 8 == 0, GFE_ERR_SHAPE);
    const int64_t items = B * V * C / 8;
    hipLaunchKernelGGL(gn_bwd_apply_kernel, dim3(grid_for(items)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dxhat, (const bf16_t*)x, mu, rstd, gamma,
                       coef_a, coef_b, (const bf16_t*)add_in, (bf16_t*)dx, V, (int)C, items);
    return gfe_launch_status();
}

int gfe_maxpool2_bwd(const void* x, const void* dy, void* dx_zeroed, int64_t B, int64_t D, int64_t H, int64_t W, int64_t C, void* stream) {
    GFE_REQUIRE(x && dy && dx_zeroed, GFE_ERR_NULL);
    GFE_REQUIRE(B > 0 && D >= 2 && H >= 2 && W >= 2 && C > 0 && C % 8 == 0, GFE_ERR_SHAPE);
    const int64_t items = B * (D / 2) * (H / 2) * (W / 2) * (C / 8);
    hipLaunchKernelGGL(maxpool2_bwd_kernel, dim3(grid_for(items)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (const bf16_t*)dy, (bf16_t*)dx_zeroed,
                       (int)D, (int)H, (int)W, (int)C, items);
    return gfe_launch_status();
}

int gfe_conv_out1_bwd(const void* x, const float* dy, const float* w, void* dx, float* dw_accum, float* db_accum, float* ws, int64_t rows, int64_t C, void* stream) {
    GFE_REQUIRE(x && dy && w && dx && dw_accum && db_accum, GFE_ERR_NULL);
    GFE_REQUIRE(rows > 0 && C > 0 && C % 8 == 0 && C <= 2048 && 256 % (C / 8) == 0, GFE_ERR_SHAPE);
    const int64_t nblk = gt_blocks(rows, 4096, 2048), rchunk = ceil_div(rows, nblk);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(out1_bwd_kernel, dim3((unsigned)nblk), dim3(256), (size_t)(256 / (C / 8)) * (C + 8) * sizeof(float), st,
                       (const bf16_t*)x, dy, w, (bf16_t*)dx, dw_accum, db_accum, rows, (int)C, rchunk, ws);
    if (ws) hipLaunchKernelGGL(gt_fold_kernel, dim3((unsigned)ceil_div(C + 1, 256), 1), dim3(256), 0, st, ws, dw_accum, db_accum, (int)nblk, (int)(C + 1), (int)C, 0, 0);
    return gfe_launch_status();
}

int gfe_conv_in1_wgrad(const float* x, const void* dr, float* dw_accum, float* db_accum, float* ws, int64_t rows, int64_t C, void* stream) {
    GFE_REQUIRE(dr && db_accum, GFE_ERR_NULL);
    GFE_REQUIRE(rows > 0 && C > 0 && C % 8 == 0 && C <= 2048 && 256 % (C / 8) == 0, GFE_ERR_SHAPE);
    const int64_t nblk = gt_blocks(rows, 4096, 2048), rchunk = ceil_div(rows, nblk);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(in1_wgrad_kernel, dim3((unsigned)nblk), dim3(256), (size_t)(256 / (C / 8)) * 2 * C * sizeof(float), st,
                       x, (const bf16_t*)dr, dw_accum, db_accum, rows, (int)C, rchunk, ws);
    if (ws) hipLaunchKernelGGL(gt_fold_kernel, dim3((unsigned)ceil_div(2 * C, 256), 1), dim3(256), 0, st, ws, dw_accum, db_accum, (int)nblk, (int)(2 * C), (int)C, 0, 0);
    return gfe_launch_status();
}

}  // extern "C"
