// Fused glue of the Mamba block (cross_atten/mamba.py): RMSNorm (:408-418) and the depthwise causal Conv1d + bias + SiLU
// (:128-131, 208-212), forward and backward.  Tiny tensors ((B*L, 512) / (B, L, 1024) with L = 37): the point is ONE launch
// per op instead of the 6-20 elementwise/reduction launches an eager formulation costs.  f32 I/O.
#include "common.h"

namespace {

// ---- RMSNorm: y = x * rsqrt(mean(x^2) + eps) * w ; one wave per row --------------------------------------------------
__global__ __launch_bounds__(256) void rmsnorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w, float* __restrict__ y,
                                                          float* __restrict__ rstd, float* __restrict__ xcopy, int rows, int dim, float eps) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float* xr = x + (size_t)row * dim;
    float s = 0.f;
    for (int c = lane; c < dim; c += 64) { const float v = xr[c]; s = fmaf(v, v, s); }
    s = wave_sum(s);
    const float r = rsqrtf(s / (float)dim + eps);
    if (lane == 0) rstd[row] = r;
    float* yr = y + (size_t)row * dim;
    if (xcopy) {                                       // the residual branch's starting value for an accumulating out_proj (gfe_hip/mamba_block.py)
        float* cr = xcopy + (size_t)row * dim;
        for (int c = lane; c < dim; c += 64) { const float v = xr[c]; yr[c] = v * r * w[c]; cr[c] = v; }
    } else {
        for (int c = lane; c < dim; c += 64) yr[c] = xr[c] * r * w[c];
    }
}

// dx = rstd * (g - x * rstd^2 * mean(g * x)),  g = dy * w ;  dw += dy * x * rstd.
// Blocks 0 .. nrb-1: one row per wave at a time, RB rows per wave (dx).  Blocks nrb .. nrb + ceil(dim/64) - 1 own 64 COLUMNS of dw each:
// their four waves split the rows, fold through LDS in wave order and add with a plain read-modify-write -- one owner and one summation
// order per element (round 3: per-block partial rows added to dw with 37-way f32 atomics; round 1: one atomic per row and column, 23 us).
// x and dy are read twice (1.2 MB at 296 x 512, L2-resident).
constexpr int RMS_RB = 2;          // rows per wave
constexpr int RMS_CV = 16;         // columns per lane kept in registers (dim <= 1024)
__global__ __launch_bounds__(256) void rmsnorm_bwd_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ rstd,
                                                          const float* __restrict__ dy, float* __restrict__ dx, float* __restrict__ dw,
                                                          const float* __restrict__ dadd, int rows, int dim, int nrb) {
    __shared__ float colred[4][64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if ((int)blockIdx.x >= nrb) {
        const int c = ((int)blockIdx.x - nrb) * 64 + lane;
        // eight independent chains per wave (rows wave, wave + 4, ... in groups of eight): the loop is a string of L2 round trips, and with two
        // chains the 74 rows a wave owns at 296 took 15 us per launch (round 4 trace); fixed fold order, so still one summation order per element
        float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (c < dim) {
            int r = wave;
            for (; r + 28 < rows; r += 32) {
#pragma unroll
                for (int i = 0; i < 8; ++i) a[i] = fmaf(dy[(size_t)(r + 4 * i) * dim + c] * x[(size_t)(r + 4 * i) * dim + c], rstd[r + 4 * i], a[i]);
            }
            for (; r < rows; r += 4) a[0] = fmaf(dy[(size_t)r * dim + c] * x[(size_t)r * dim + c], rstd[r], a[0]);
        }
        colred[wave][lane] = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
        __syncthreads();
        if (wave == 0 && c < dim) dw[c] += (colred[0][lane] + colred[1][lane]) + (colred[2][lane] + colred[3][lane]);
        return;
    }
    float wv[RMS_CV];
#pragma unroll
    for (int i = 0; i < RMS_CV; ++i) { const int c = lane + 64 * i; wv[i] = c < dim ? w[c] : 0.f; }
    for (int k = 0; k < RMS_RB; ++k) {
        const int row = (blockIdx.x * 4 + wave) * RMS_RB + k;
        if (row >= rows) break;
        const float* xr = x + (size_t)row * dim;
        const float* gr = dy + (size_t)row * dim;
        const float r = rstd[row];
        float xv[RMS_CV], gv[RMS_CV], av[RMS_CV];
        const float* ar = dadd ? dadd + (size_t)row * dim : nullptr;          // the residual branch's gradient, added here instead of by a launch of its own
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < RMS_CV; ++i) {
            const int c = lane + 64 * i;
            xv[i] = c < dim ? xr[c] : 0.f; gv[i] = c < dim ? gr[c] : 0.f;
            av[i] = (ar && c < dim) ? ar[c] : 0.f;                             // (fetched with the row, not behind its reduction)
            s = fmaf(gv[i] * wv[i], xv[i], s);
        }
        s = wave_sum(s);
        const float kk = s * r * r / (float)dim;
        float* dxr = dx + (size_t)row * dim;
#pragma unroll
        for (int i = 0; i < RMS_CV; ++i) {
            const int c = lane + 64 * i;
            if (c < dim) {
                float v = r * (gv[i] * wv[i] - xv[i] * kk);
                asm volatile("" : "+v"(v));                    // rounded here: the sum below must not contract into an fma (the same bits as a separate add launch)
                dxr[c] = v + av[i];
            }
        }
    }
}

// ---- depthwise causal conv1d (kernel KS, left padding KS-1) + bias + SiLU on (B, L, ED); lane = channel --------------------
template <int KS>
__global__ __launch_bounds__(256) void dwconv_silu_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                              float* __restrict__ y, int L, int ED, int ldx) {
    // one thread per (sample, step, channel): the KS-wide window is re-read (L1 hits) instead of carried through a sequential loop.
    // ldx: row stride of x in floats (ED for a contiguous tensor; 2 ED when x is the first half of the in_proj output, read in place)
    const int e = blockIdx.x * 256 + threadIdx.x, t = blockIdx.y, b = blockIdx.z;
    if (e >= ED) return;
    float pre = bias ? bias[e] : 0.f;
#pragma unroll
    for (int k = 0; k < KS; ++k) {
        const int tt = t + k - (KS - 1);
        if (tt >= 0) pre = fmaf(w[e * KS + k], x[((size_t)b * L + tt) * ldx + e], pre);
    }
    y[((size_t)b * L + t) * ED + e] = pre * sigmoidf_(pre);
}

// backward: dpre = dy * silu'(pre); dx[t] = sum_k w[k] * dpre[t + (KS-1) - k]; dw[k] += dpre[t] * x[t + k - (KS-1)]; db += dpre.
// Block = 64 channels of one sample; its 4 waves split the time steps: phase 1 writes dpre[t][channel] to LDS and keeps dw / db
// partials, phase 2 forms dx from the LDS copy, then the partials are folded across the waves (fixed order) and stored as the sample's
// partial row part[b][k][e]; dwconv_fold_kernel adds the samples in order to dw / db (round 3: B-way f32 atomics).
template <int KS>
__global__ __launch_bounds__(256) void dwconv_silu_bwd_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                              const float* __restrict__ dy, float* __restrict__ dx, float* __restrict__ partial,
                                                              int L, int ED, int ldx, int lddx) {
    extern __shared__ float sp[];                       // [L + KS - 1][64] dpre (zero tail), then [4][KS + 1][64] partials
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + lane, b = blockIdx.y;
    const bool live = e < ED;
    float wk[KS], dwacc[KS];
#pragma unroll
    for (int k = 0; k < KS; ++k) { wk[k] = live ? w[e * KS + k] : 0.f; dwacc[k] = 0.f; }
    const float bv = (bias && live) ? bias[e] : 0.f;
    float dbacc = 0.f;
    const float* xb = x + (size_t)b * L * ldx + e;
    for (int t = wave; t < L + KS - 1; t += 4) {
        float d = 0.f;
        if (t < L && live) {
            float win[KS];
#pragma unroll
            for (int k = 0; k < KS; ++k) { const int tt = t + k - (KS - 1); win[k] = tt >= 0 ? xb[(size_t)tt * ldx] : 0.f; }
            float pre = bv;
#pragma unroll
            for (int k = 0; k < KS; ++k) pre = fmaf(wk[k], win[k], pre);
            const float sg = sigmoidf_(pre);
            d = dy[((size_t)b * L + t) * ED + e] * sg * (1.f + pre * (1.f - sg));
#pragma unroll
            for (int k = 0; k < KS; ++k) dwacc[k] = fmaf(d, win[k], dwacc[k]);
            dbacc += d;
        }
        sp[t * 64 + lane] = d;                           // rows L .. L+KS-2 stay zero: dpre beyond the sequence
    }
    __syncthreads();
    if (live) {
        for (int t = wave; t < L; t += 4) {
            float acc = 0.f;
#pragma unroll
            for (int k = 0; k < KS; ++k) acc = fmaf(wk[k], sp[(t + (KS - 1) - k) * 64 + lane], acc);
            dx[((size_t)b * L + t) * lddx + e] = acc;
        }
    }
    __syncthreads();
    float* part = sp;                                    // reuse: [4][KS + 1][64]
#pragma unroll
    for (int k = 0; k < KS; ++k) part[(wave * (KS + 1) + k) * 64 + lane] = dwacc[k];
    part[(wave * (KS + 1) + KS) * 64 + lane] = dbacc;
    __syncthreads();
    if (wave == 0 && live) {
#pragma unroll
        for (int k = 0; k <= KS; ++k) {
            const float t = part[(0 * (KS + 1) + k) * 64 + lane] + part[(1 * (KS + 1) + k) * 64 + lane] +
                            part[(2 * (KS + 1) + k) * 64 + lane] + part[(3 * (KS + 1) + k) * 64 + lane];
            partial[((size_t)b * (KS + 1) + k) * ED + e] = t;
        }
    }
}
// dw[e][k] += sum_b part[b][k][e] (k < KS), db[e] += sum_b part[b][KS][e]: samples in order, one owner per element
template <int KS>
__global__ __launch_bounds__(256) void dwconv_fold_kernel(const float* __restrict__ partial, float* __restrict__ dw, float* __restrict__ db, int B, int ED) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= (KS + 1) * ED) return;
    const int k = i / ED, e = i - k * ED;
    float t = 0.f;
    for (int b = 0; b < B; ++b) t += partial[((size_t)b * (KS + 1) + k) * ED + e];
    if (k < KS) dw[e * KS + k] += t;
    else if (db) db[e] += t;
}

}  // namespace

// ---- single-token inference (cross_atten/mamba.py:342-405): one thread per (sample, channel) ----------------------------------------------
// conv step: the window is the cached last KS-1 conv inputs + this token's; the cache shifts by one (mamba.py:354-361, 371-372)
template <int KS>
__global__ __launch_bounds__(256) void step_conv_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ cin, float* __restrict__ cout,
                                                        const float* __restrict__ w, const float* __restrict__ bias, float* __restrict__ xc, int ED) {
    const int e = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    if (e >= ED) return;
    float win[KS];
#pragma unroll
    for (int k = 0; k < KS - 1; ++k) win[k] = cin[((size_t)b * ED + e) * (KS - 1) + k];
    win[KS - 1] = x[(size_t)b * ldx + e];
    float pre = bias ? bias[e] : 0.f;
#pragma unroll
    for (int k = 0; k < KS; ++k) pre = fmaf(w[e * KS + k], win[k], pre);
    xc[(size_t)b * ED + e] = pre * sigmoidf_(pre);
#pragma unroll
    for (int k = 0; k < KS - 1; ++k) cout[((size_t)b * ED + e) * (KS - 1) + k] = win[k + 1];
}

// ssm step (mamba.py:374-405): h <- exp(dt A) h + dt B x;  y = (h . C + D x) * silu(z), dt = softplus(delta + bias), A = -exp(A_log)
__global__ __launch_bounds__(256) void step_ssm_kernel(const float* __restrict__ xc, const float* __restrict__ delta, const float* __restrict__ A_log,
                                                       const float* __restrict__ Bm, const float* __restrict__ Cm, int ld_bc,
                                                       const float* __restrict__ D, const float* __restrict__ dbias, const float* __restrict__ z, int ld_z,
                                                       const float* __restrict__ hin, float* __restrict__ hout, float* __restrict__ y, int ED, int N) {
    const int e = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    if (e >= ED) return;
    const float xv = xc[(size_t)b * ED + e];
    const float dt = softplusf_(delta[(size_t)b * ED + e] + (dbias ? dbias[e] : 0.f));
    const float dtx = dt * xv;
    float acc = 0.f;
    for (int n = 0; n < N; ++n) {
        const float a = fast_exp(dt * -fast_exp(A_log[(size_t)e * N + n]));
        const size_t hi = ((size_t)b * ED + e) * N + n;
        const float h = fmaf(a, hin ? hin[hi] : 0.f, dtx * Bm[(size_t)b * ld_bc + n]);
        hout[hi] = h;
        acc = fmaf(h, Cm[(size_t)b * ld_bc + n], acc);
    }
    acc = fmaf(D ? D[e] : 0.f, xv, acc);
    if (z) { const float zv = z[(size_t)b * ld_z + e]; acc *= siluf_(zv); }
    y[(size_t)b * ED + e] = acc;
}

extern "C" {

int gfe_mamba_step_conv(const float* x, int64_t ldx, const float* cache_in, float* cache_out, const float* w, const float* bias, float* xc,
                        int64_t B, int64_t ED, int64_t KS, void* stream) {
    GFE_REQUIRE(x && cache_in && cache_out && w && xc, GFE_ERR_NULL);
    GFE_REQUIRE(B > 0 && B <= 65535 && ED > 0 && KS == 4 && ldx >= ED && ldx <= 0x7fffffff, GFE_ERR_SHAPE);
    hipLaunchKernelGGL((step_conv_kernel<4>), dim3((unsigned)ceil_div(ED, 256), (unsigned)B), dim3(256), 0, (hipStream_t)stream,
                       x, (int)ldx, cache_in, cache_out, w, bias, xc, (int)ED);
    return gfe_launch_status();
}

int gfe_mamba_step_ssm(const float* xc, const float* delta, const float* A_log, const float* Bm, const float* Cm, int64_t ld_bc,
                       const float* D, const float* delta_bias, const float* z, int64_t ld_z, const float* h_in, float* h_out, float* y,
                       int64_t B, int64_t ED, int64_t N, void* stream) {
    GFE_REQUIRE(xc && delta && A_log && Bm && Cm && h_out && y, GFE_ERR_NULL);
    GFE_REQUIRE(B > 0 && B <= 65535 && ED > 0 && N > 0 && N <= 256 && ld_bc >= N && (!z || ld_z >= ED) && ld_bc <= 0x7fffffff && ld_z <= 0x7fffffff, GFE_ERR_SHAPE);
    hipLaunchKernelGGL(step_ssm_kernel, dim3((unsigned)ceil_div(ED, 256), (unsigned)B), dim3(256), 0, (hipStream_t)stream,
                       xc, delta, A_log, Bm, Cm, (int)ld_bc, D, delta_bias, z, (int)ld_z, h_in, h_out, y, (int)ED, (int)N);
    return gfe_launch_status();
}

int gfe_rmsnorm_fwd(const float* x, const float* w, float* y, float* rstd, float* xcopy, int64_t rows, int64_t dim, float eps, void* stream) {
    GFE_REQUIRE(x && w && y && rstd, GFE_ERR_NULL);
    GFE_REQUIRE(rows > 0 && dim > 0, GFE_ERR_SHAPE);
    hipLaunchKernelGGL(rmsnorm_fwd_kernel, dim3((unsigned)ceil_div(rows, 4)), dim3(256), 0, (hipStream_t)stream, x, w, y, rstd, xcopy, (int)rows, (int)dim, eps);
    return gfe_launch_status();
}

int gfe_rmsnorm_bwd(const float* x, const float* w, const float* rstd, const float* dy, float* dx, float* dw_accum, const float* dadd,
                    int64_t rows, int64_t dim, void* stream) {
    GFE_REQUIRE(x && w && rstd && dy && dx && dw_accum, GFE_ERR_NULL);
    GFE_REQUIRE(rows > 0 && rows <= 0x3fffffff && dim > 0, GFE_ERR_SHAPE);
    GFE_REQUIRE(dim <= 64 * RMS_CV, GFE_ERR_SHAPE);
    const int nrb = (int)ceil_div(rows, 4 * RMS_RB);
    hipLaunchKernelGGL(rmsnorm_bwd_kernel, dim3((unsigned)(nrb + ceil_div(dim, 64))), dim3(256), 0, (hipStream_t)stream,
                       x, w, rstd, dy, dx, dw_accum, dadd, (int)rows, (int)dim, nrb);
    return gfe_launch_status();
}

int gfe_dwconv1d_silu_fwd(const float* x, int64_t ldx, const float* w, const float* bias, float* y, int64_t B, int64_t L, int64_t ED, int64_t KS, void* stream) {
    GFE_REQUIRE(x && w && y, GFE_ERR_NULL);
    GFE_REQUIRE(B > 0 && B <= 65535 && L > 0 && L <= 65535 && ED > 0 && KS == 4 && ldx >= ED && ldx <= 0x7fffffff, GFE_ERR_SHAPE);
    hipLaunchKernelGGL((dwconv_silu_fwd_kernel<4>), dim3((unsigned)ceil_div(ED, 256), (unsigned)L, (unsigned)B), dim3(256), 0, (hipStream_t)stream, x, w, bias, y, (int)L, (int)ED, (int)ldx);
    return gfe_launch_status();
}

int gfe_dwconv1d_silu_bwd(const float* x, int64_t ldx, const float* w, const float* bias, const float* dy, float* dx, int64_t lddx, float* dw_accum, float* db_accum,
                          float* ws, int64_t B, int64_t L, int64_t ED, int64_t KS, void* stream) {
    GFE_REQUIRE(x && w && dy && dx && dw_accum && ws, GFE_ERR_NULL);
    GFE_REQUIRE(B > 0 && B <= 65535 && L > 0 && ED > 0 && KS == 4 && ldx >= ED && lddx >= ED && ldx <= 0x7fffffff && lddx <= 0x7fffffff, GFE_ERR_SHAPE);
    const size_t rows = (size_t)L + 3 > 20 ? (size_t)L + 3 : 20;          // dpre rows, reused for the 4 x 5 partial rows
    GFE_REQUIRE(rows * 64 * sizeof(float) <= 64 * 1024, GFE_ERR_SHAPE);
    hipLaunchKernelGGL((dwconv_silu_bwd_kernel<4>), dim3((unsigned)ceil_div(ED, 64), (unsigned)B), dim3(256), rows * 64 * sizeof(float), (hipStream_t)stream,
                       x, w, bias, dy, dx, ws, (int)L, (int)ED, (int)ldx, (int)lddx);
    hipLaunchKernelGGL((dwconv_fold_kernel<4>), dim3((unsigned)ceil_div(5 * ED, 256)), dim3(256), 0, (hipStream_t)stream, ws, dw_accum, db_accum, (int)B, (int)ED);
    return gfe_launch_status();
}

}  // extern "C"
