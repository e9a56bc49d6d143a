// Input normalisation of the data loader: utils/data_normalization.py:20-48 `adaptive_normal`, applied to every MRI
// volume before it reaches the step (dataloader/pic_table_loader.py:107).
//
// Reference algorithm: sort all voxels >= 0, take the order statistics at ranks int((m-1)*0.001 + 0.5) and int((m-1)*0.999 + 0.5)
// (m = number of such voxels; the arithmetic is Python double), then y = clamp((x - (hi+lo)/2) / ((hi-lo)/2), -1, 1).
// Here the two order statistics come from a 3-pass radix SELECT on the float bit patterns (non-negative floats order like their
// bits): 11 + 11 + 10 bits, one histogram pass over the volume each, both ranks in the same pass -- no sort, 3 reads of the volume
// instead of O(n log n) sorting traffic, bit-exact (an order statistic is a value of the input).  The fourth pass normalises.
// HBM-bound integer work: one float4 per lane, LDS histograms (LDS atomics), non-empty bins flushed with global atomics.
#include "common.h"

namespace {

constexpr int HB = 2048;                        // bins per histogram
constexpr int WS_WORDS = 8 + 2 * HB;            // per volume: m, rank[2], prefix[2], value bits[2], pad; hist[2][HB]
constexpr int HIST_THREADS = 256;

__device__ __forceinline__ bool keyed(float x, uint32_t& key) {
    if (!(x >= 0.f)) return false;              // negatives and NaN are not part of the sorted set (imgArray[imgArray >= 0])
    key = x == 0.f ? 0u : __float_as_uint(x);   // -0.0 >= 0 is true and sorts as 0
    return true;
}

// pass 0: bits 31..21 of every key (one histogram); pass 1 / 2: bits 20..10 / 9..0 of the keys under each target's prefix
template <int PASS>
__global__ __launch_bounds__(HIST_THREADS) void an_hist_kernel(const float* __restrict__ x, uint32_t* __restrict__ ws, int64_t n) {
    __shared__ uint32_t h[2][HB];
    const int b = blockIdx.y, tid = threadIdx.x;
    uint32_t* w = ws + (size_t)b * WS_WORDS;
    for (int i = tid; i < 2 * HB; i += HIST_THREADS) (&h[0][0])[i] = 0;
    const uint32_t p0 = PASS ? w[3] : 0, p1 = PASS ? w[4] : 0;
    __syncthreads();
    const float* xb = x + (size_t)b * n;
    // run-length cache per thread and histogram: MRI volumes are mostly background (exact zeros) and smooth tissue, so consecutive
    // voxels of a thread tend to fall into the same bin -- one LDS atomic per run instead of one per voxel, and no 64-way
    // same-address serialisation on the background bin
    int rb[2] = {-1, -1};
    uint32_t rc[2] = {0, 0};
    auto put = [&](int t, int bin) {
        if (bin == rb[t]) { ++rc[t]; return; }
        if (rc[t]) atomicAdd(&h[t][rb[t]], rc[t]);
        rb[t] = bin; rc[t] = 1;
    };
    auto add = [&](float v) {
        uint32_t k;
        if (!keyed(v, k)) return;
        if (PASS == 0) put(0, (int)(k >> 21));
        else if (PASS == 1) {
            if ((k >> 21) == p0) put(0, (int)((k >> 10) & 2047));
            if ((k >> 21) == p1) put(1, (int)((k >> 10) & 2047));
        } else {
            if ((k >> 10) == p0) put(0, (int)(k & 1023));
            if ((k >> 10) == p1) put(1, (int)(k & 1023));
        }
    };
    const int64_t n4 = ((uintptr_t)xb % 16 == 0) ? n / 4 : 0;                  // float4 body when the volume is 16-B aligned
    for (int64_t i = (int64_t)blockIdx.x * HIST_THREADS + tid; i < n4; i += (int64_t)gridDim.x * HIST_THREADS) {
        const float4 v = reinterpret_cast<const float4*>(xb)[i];
        add(v.x); add(v.y); add(v.z); add(v.w);
    }
    for (int64_t i = n4 * 4 + (int64_t)blockIdx.x * HIST_THREADS + tid; i < n; i += (int64_t)gridDim.x * HIST_THREADS) add(xb[i]);
    for (int t = 0; t < 2; ++t)
        if (rc[t]) atomicAdd(&h[t][rb[t]], rc[t]);
    __syncthreads();
    uint32_t* gh = w + 8;
    for (int i = tid; i < (PASS ? 2 : 1) * HB; i += HIST_THREADS) {
        const uint32_t c = (&h[0][0])[i];
        if (c) atomicAdd(gh + i, c);
    }
}

// one block per volume: finds, for both targets, the bin that holds the wanted rank; narrows prefix / rank; clears the histograms
template <int PASS>
__global__ __launch_bounds__(256) void an_select_kernel(uint32_t* __restrict__ ws) {
    __shared__ uint32_t part[2][256];
    __shared__ uint32_t sel[2][2];               // [target][bin, rank inside the bin]
    const int b = blockIdx.x, tid = threadIdx.x;
    uint32_t* w = ws + (size_t)b * WS_WORDS;
    uint32_t* gh = w + 8;
    constexpr int BINS = PASS == 2 ? 1024 : HB, PER = BINS / 256;
    uint32_t loc[2][PER > 0 ? PER : 1], sum[2] = {0, 0};
    for (int t = 0; t < 2; ++t)
        for (int j = 0; j < PER; ++j) { loc[t][j] = gh[(PASS ? t : 0) * HB + tid * PER + j]; sum[t] += loc[t][j]; }
    part[0][tid] = sum[0]; part[1][tid] = sum[1];
    __syncthreads();
    // inclusive prefix sums of the 256 per-thread totals (Hillis-Steele, 8 steps), both targets at once
    uint32_t inc[2] = {sum[0], sum[1]};
    for (int d = 1; d < 256; d <<= 1) {
        uint32_t a0 = 0, a1 = 0;
        if (tid >= d) { a0 = part[0][tid - d]; a1 = part[1][tid - d]; }
        __syncthreads();
        inc[0] += a0; inc[1] += a1;
        part[0][tid] = inc[0]; part[1][tid] = inc[1];
        __syncthreads();
    }
    if (tid == 0 && PASS == 0) {
        const uint32_t m = part[0][255];
        w[0] = m;
        // int(round(len - 1) * p + 0.5), clipped to [0, len - 1]  (data_normalization.py:28-40; Python float == IEEE double)
        for (int t = 0; t < 2; ++t) {
            long long idx = (long long)((double)((long long)m - 1) * (t ? 0.999 : 0.001) + 0.5);
            if (idx < 0) idx = 0;
            if (idx > (long long)m - 1) idx = (long long)m - 1;
            w[1 + t] = (uint32_t)(idx < 0 ? 0 : idx);
        }
    }
    __syncthreads();
    // the thread chunk that holds rank r: exclusive prefix <= r < inclusive prefix (the last chunk takes anything beyond: m == 0)
    for (int t = 0; t < 2; ++t) {
        const uint32_t r = w[1 + t], exc = inc[t] - sum[t];
        if ((exc <= r && r < inc[t]) || (tid == 255 && r >= inc[t])) { sel[t][0] = (uint32_t)tid; sel[t][1] = r - exc; }
    }
    __syncthreads();
    for (int t = 0; t < 2; ++t) {
        if ((int)sel[t][0] == tid) {
            uint32_t r = sel[t][1];
            int j = 0;
            for (; j < PER - 1 && loc[t][j] <= r; ++j) r -= loc[t][j];
            const uint32_t bin = (uint32_t)(tid * PER + j);
            w[1 + t] = r;
            if (PASS == 0) w[3 + t] = bin;
            else if (PASS == 1) w[3 + t] = (w[3 + t] << 11) | bin;
            else w[5 + t] = (w[3 + t] << 10) | bin;                             // the order statistic's bit pattern
        }
    }
    __syncthreads();
    for (int i = tid; i < 2 * HB; i += 256) gh[i] = 0;
}

__global__ __launch_bounds__(256) void an_apply_kernel(const float* __restrict__ x, float* __restrict__ y, const uint32_t* __restrict__ ws, int64_t n) {
    const int b = blockIdx.y;
    const uint32_t* w = ws + (size_t)b * WS_WORDS;
    const float lo = __uint_as_float(w[5]), hi = __uint_as_float(w[6]);
    const float mean = __fdiv_rn(__fadd_rn(hi, lo), 2.0f), sd = __fdiv_rn(__fsub_rn(hi, lo), 2.0f);       // :42-43
    auto f = [&](float v) {
        float r = __fdiv_rn(__fsub_rn(v, mean), sd);                                                     // :44
        if (r < -1.f) r = -1.f;                                                                          // :45  (NaN stays NaN)
        if (r > 1.f) r = 1.f;                                                                            // :46
        return r;
    };
    const float* xb = x + (size_t)b * n;
    float* yb = y + (size_t)b * n;
    const int64_t n4 = ((uintptr_t)xb % 16 == 0 && (uintptr_t)yb % 16 == 0) ? n / 4 : 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const float4 v = reinterpret_cast<const float4*>(xb)[i];
        reinterpret_cast<float4*>(yb)[i] = make_float4(f(v.x), f(v.y), f(v.z), f(v.w));
    }
    for (int64_t i = n4 * 4 + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) yb[i] = f(xb[i]);
}


// ---- Resized(spatial_size) of the loader (dataloader/pic_table_loader.py:58: monai Resized, default mode "area" = torch
// F.interpolate(mode="area") = adaptive average pooling): out[i] = mean of in[floor(i D/d) .. ceil((i+1) D/d)) per axis. --------
__global__ __launch_bounds__(256) void resize_area_kernel(const float* __restrict__ x, float* __restrict__ y, int D, int H, int W, int d, int h, int w) {
    const int64_t n = (int64_t)d * h * w;
    const float* xb = x + (size_t)blockIdx.y * D * H * W;
    float* yb = y + (size_t)blockIdx.y * n;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int ow = (int)(i % w), oh = (int)((i / w) % h), od = (int)(i / ((int64_t)w * h));
        // start = floor(o * in / out), end = ceil((o + 1) * in / out)   (ATen adaptive pooling index rule)
        const int d0 = (int)(((int64_t)od * D) / d), d1 = (int)((((int64_t)od + 1) * D + d - 1) / d);
        const int h0 = (int)(((int64_t)oh * H) / h), h1 = (int)((((int64_t)oh + 1) * H + h - 1) / h);
        const int w0 = (int)(((int64_t)ow * W) / w), w1 = (int)((((int64_t)ow + 1) * W + w - 1) / w);
        float s = 0.f;
        for (int a = d0; a < d1; ++a)
            for (int b = h0; b < h1; ++b)
                for (int c = w0; c < w1; ++c) s += xb[((size_t)a * H + b) * W + c];
        yb[i] = s / (float)((d1 - d0) * (h1 - h0) * (w1 - w0));
    }
}

}  // namespace

extern "C" {

int gfe_adaptive_normal_ws_words(void) { return WS_WORDS; }

int gfe_adaptive_normal(const float* x, float* y, uint32_t* ws, int64_t B, int64_t n, void* stream) {
    GFE_REQUIRE(x && y && ws, GFE_ERR_NULL);
    GFE_REQUIRE(B > 0 && B <= 65535 && n > 0 && n < (1LL << 32), GFE_ERR_SHAPE);       // counts and ranks are 32-bit
    hipStream_t st = (hipStream_t)stream;
    gfe_zero_async(ws, (size_t)B * WS_WORDS * sizeof(uint32_t), st);
    const int64_t per = (int64_t)HIST_THREADS * 4 * 8;                               // ~8 float4 per thread
    const unsigned gx = (unsigned)(ceil_div(n, per) < 2048 ? ceil_div(n, per) : 2048);
    const dim3 grid(gx, (unsigned)B);
    hipLaunchKernelGGL(an_hist_kernel<0>, grid, dim3(HIST_THREADS), 0, st, x, ws, n);
    hipLaunchKernelGGL(an_select_kernel<0>, dim3((unsigned)B), dim3(256), 0, st, ws);
    hipLaunchKernelGGL(an_hist_kernel<1>, grid, dim3(HIST_THREADS), 0, st, x, ws, n);
    hipLaunchKernelGGL(an_select_kernel<1>, dim3((unsigned)B), dim3(256), 0, st, ws);
    hipLaunchKernelGGL(an_hist_kernel<2>, grid, dim3(HIST_THREADS), 0, st, x, ws, n);
    hipLaunchKernelGGL(an_select_kernel<2>, dim3((unsigned)B), dim3(256), 0, st, ws);
    hipLaunchKernelGGL(an_apply_kernel, grid, dim3(256), 0, st, x, y, ws, n);
    return gfe_launch_status();
}

/* Area (adaptive-average) resize of B single-channel volumes (B, D, H, W) -> (B, d, h, w), f32. */
int gfe_resize_area(const float* x, float* y, int64_t B, int64_t D, int64_t H, int64_t W, int64_t d, int64_t h, int64_t w, void* stream) {
    GFE_REQUIRE(x && y, GFE_ERR_NULL);
    GFE_REQUIRE(B > 0 && B <= 65535 && D > 0 && H > 0 && W > 0 && d > 0 && h > 0 && w > 0 && D * H * W < (1LL << 31) && d * h * w < (1LL << 31), GFE_ERR_SHAPE);
    int64_t g = ceil_div(d * h * w, 256); if (g > 4096) g = 4096;
    hipLaunchKernelGGL(resize_area_kernel, dim3((unsigned)g, (unsigned)B), dim3(256), 0, (hipStream_t)stream, x, y, (int)D, (int)H, (int)W, (int)d, (int)h, (int)w);
    return gfe_launch_status();
}

}  // extern "C"
