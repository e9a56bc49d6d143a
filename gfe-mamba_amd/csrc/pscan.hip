// Materialised linear-recurrence scan H[t] = A[t]*H[t-1] + X[t] -- drop-in for the reference's
// Blelloch `pscan` (cross_atten/pscan.py:36-92 up/down sweep, :151-186 forward, :188-224 backward).
//
// The reference runs 2*log2(L) strided in-place sweeps over (B,D,L,N) copies (pscan.py:54-63, 83-92) and pads L
// to a power of two (pscan.py:20-33).  Here the recurrence is evaluated directly: (B, L, D*N) is streamed once
// with 16-byte lanes over the contiguous D*N axis, sequentially in t, chunked in L with a (prod A, h) carry
// so that B*DN/vec*nchunks lanes fill the chip.  No padding, any L >= 1.  HBM-bound: 3 tensors fwd, 5 bwd.
#include "common.h"

namespace {

template <typename T> struct Vec;
template <> struct Vec<float> {
    static constexpr int W = 4;
    static __device__ __forceinline__ void ld(const float* p, float (&o)[4]) {
        const float4 v = *reinterpret_cast<const float4*>(p);
        o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
    }
    static __device__ __forceinline__ void st(float* p, const float (&o)[4]) {
        *reinterpret_cast<float4*>(p) = make_float4(o[0], o[1], o[2], o[3]);
    }
};
template <> struct Vec<bf16_t> {
    static constexpr int W = 8;
    static __device__ __forceinline__ void ld(const bf16_t* p, float (&o)[8]) {
        const uint4 v = *reinterpret_cast<const uint4*>(p);
        o[0] = bf16lo_to_f32(v.x); o[1] = bf16hi_to_f32(v.x); o[2] = bf16lo_to_f32(v.y); o[3] = bf16hi_to_f32(v.y);
        o[4] = bf16lo_to_f32(v.z); o[5] = bf16hi_to_f32(v.z); o[6] = bf16lo_to_f32(v.w); o[7] = bf16hi_to_f32(v.w);
    }
    static __device__ __forceinline__ void st(bf16_t* p, const float (&o)[8]) {
        uint4 v;
        v.x = pack_bf16x2(o[0], o[1]); v.y = pack_bf16x2(o[2], o[3]); v.z = pack_bf16x2(o[4], o[5]); v.w = pack_bf16x2(o[6], o[7]);
        *reinterpret_cast<uint4*>(p) = v;
    }
};

// MODE 0: chunk summary (P, h) with zero start; MODE 1: final pass from the carried start state.
template <typename T, int MODE>
__global__ __launch_bounds__(256) void pscan_fwd_kernel(const T* __restrict__ A, const T* __restrict__ X, T* __restrict__ H,
                                                        float* __restrict__ ws, int L, int64_t DN, int Tc, int nchunks) {
    constexpr int W = Vec<T>::W;
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * W;
    const int c = blockIdx.y, b = blockIdx.z;
    if (i >= DN) return;
    const int t0 = c * Tc, t1 = min(L, t0 + Tc);
    float h[W], P[W];
    float* wsp = ws + (((size_t)b * nchunks + c) * 2) * DN + i;
#pragma unroll
    for (int k = 0; k < W; ++k) { h[k] = 0.f; P[k] = 1.f; }
    if (MODE == 1 && nchunks > 1) {
#pragma unroll
        for (int k = 0; k < W; ++k) h[k] = wsp[DN + k];
    }
#pragma unroll 4
    for (int t = t0; t < t1; ++t) {
        const size_t off = ((size_t)b * L + t) * DN + i;
        float a[W], x[W];
        Vec<T>::ld(A + off, a);
        Vec<T>::ld(X + off, x);
#pragma unroll
        for (int k = 0; k < W; ++k) { h[k] = fmaf(a[k], h[k], x[k]); if (MODE == 0) P[k] *= a[k]; }
        if (MODE == 1) Vec<T>::st(H + off, h);
    }
    if (MODE == 0) {
#pragma unroll
        for (int k = 0; k < W; ++k) { wsp[k] = P[k]; wsp[DN + k] = h[k]; }
    }
}

template <bool REVERSE>
__global__ __launch_bounds__(256) void pscan_carry_kernel(float* __restrict__ ws, int64_t DN, int nchunks) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int b = blockIdx.y;
    if (i >= DN) return;
    float Hc = 0.f;
#pragma unroll 4
    for (int k = 0; k < nchunks; ++k) {
        const int c = REVERSE ? nchunks - 1 - k : k;
        float* wsp = ws + (((size_t)b * nchunks + c) * 2) * DN + i;
        const float P = wsp[0], loc = wsp[DN];
        wsp[DN] = Hc;
        Hc = fmaf(P, Hc, loc);
    }
}

template <typename T, int MODE>
__global__ __launch_bounds__(256) void pscan_bwd_kernel(const T* __restrict__ A, const T* __restrict__ H, const T* __restrict__ gH,
                                                        T* __restrict__ gA, T* __restrict__ gX, float* __restrict__ ws,
                                                        int L, int64_t DN, int Tc, int nchunks) {
    constexpr int W = Vec<T>::W;
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * W;
    const int c = blockIdx.y, b = blockIdx.z;
    if (i >= DN) return;
    const int t0 = c * Tc, t1 = min(L, t0 + Tc);
    float r[W], P[W];
    float* wsp = ws + (((size_t)b * nchunks + c) * 2) * DN + i;
#pragma unroll
    for (int k = 0; k < W; ++k) { r[k] = 0.f; P[k] = 1.f; }
    if (MODE == 1 && nchunks > 1) {
#pragma unroll
        for (int k = 0; k < W; ++k) r[k] = wsp[DN + k];
    }
#pragma unroll 4
    for (int t = t1 - 1; t >= t0; --t) {
        const size_t off = ((size_t)b * L + t) * DN + i;
        float an[W], g[W];
        if (t + 1 < L) Vec<T>::ld(A + off + DN, an);   // A shifted left by one (pscan.py:216)
        else {
#pragma unroll
            for (int k = 0; k < W; ++k) an[k] = 0.f;
        }
        Vec<T>::ld(gH + off, g);
#pragma unroll
        for (int k = 0; k < W; ++k) { r[k] = fmaf(an[k], r[k], g[k]); if (MODE == 0) P[k] *= an[k]; }
        if (MODE == 1) {
            Vec<T>::st(gX + off, r);
            float hp[W], ga[W];
            if (t > 0) Vec<T>::ld(H + off - DN, hp);
            else {
#pragma unroll
                for (int k = 0; k < W; ++k) hp[k] = 0.f;
            }
#pragma unroll
            for (int k = 0; k < W; ++k) ga[k] = hp[k] * r[k];     // pscan.py:221-222, gA[0] = 0
            Vec<T>::st(gA + off, ga);
        }
    }
    if (MODE == 0) {
#pragma unroll
        for (int k = 0; k < W; ++k) { wsp[k] = P[k]; wsp[DN + k] = r[k]; }
    }
}

template <typename T>
int pscan_fwd_launch(const void* A, const void* X, void* H, float* ws, int64_t B, int64_t L, int64_t DN, int Tc, hipStream_t st) {
    constexpr int W = Vec<T>::W;
    if (DN % W) return GFE_ERR_SHAPE;
    const int nchunks = (int)ceil_div(L, Tc);
    const dim3 blk(256), grid((unsigned)ceil_div(DN / W, 256), nchunks, (unsigned)B);
    if (nchunks > 1) {
        if (!ws) return GFE_ERR_NULL;
        hipLaunchKernelGGL((pscan_fwd_kernel<T, 0>), grid, blk, 0, st, (const T*)A, (const T*)X, (T*)H, ws, (int)L, DN, Tc, nchunks);
        hipLaunchKernelGGL((pscan_carry_kernel<false>), dim3((unsigned)ceil_div(DN, 256), (unsigned)B), blk, 0, st, ws, DN, nchunks);
    }
    hipLaunchKernelGGL((pscan_fwd_kernel<T, 1>), grid, blk, 0, st, (const T*)A, (const T*)X, (T*)H, ws, (int)L, DN, Tc, nchunks);
    return gfe_launch_status();
}

template <typename T>
int pscan_bwd_launch(const void* A, const void* H, const void* gH, void* gA, void* gX, float* ws,
                     int64_t B, int64_t L, int64_t DN, int Tc, hipStream_t st) {
    constexpr int W = Vec<T>::W;
    if (DN % W) return GFE_ERR_SHAPE;
    const int nchunks = (int)ceil_div(L, Tc);
    const dim3 blk(256), grid((unsigned)ceil_div(DN / W, 256), nchunks, (unsigned)B);
    if (nchunks > 1) {
        if (!ws) return GFE_ERR_NULL;
        hipLaunchKernelGGL((pscan_bwd_kernel<T, 0>), grid, blk, 0, st, (const T*)A, (const T*)H, (const T*)gH, (T*)gA, (T*)gX, ws, (int)L, DN, Tc, nchunks);
        hipLaunchKernelGGL((pscan_carry_kernel<true>), dim3((unsigned)ceil_div(DN, 256), (unsigned)B), blk, 0, st, ws, DN, nchunks);
    }
    hipLaunchKernelGGL((pscan_bwd_kernel<T, 1>), grid, blk, 0, st, (const T*)A, (const T*)H, (const T*)gH, (T*)gA, (T*)gX, ws, (int)L, DN, Tc, nchunks);
    return gfe_launch_status();
}

}  // namespace

extern "C" {

int gfe_pscan_fwd(const void* A, const void* X, void* H, float* ws,
                  int64_t B, int64_t L, int64_t DN, int T, int dtype, void* stream) {
    GFE_REQUIRE(A && X && H, GFE_ERR_NULL);
    GFE_REQUIRE(B > 0 && B <= 65535 && L > 0 && DN > 0 && T > 0 && ceil_div(L, T) <= 65535, GFE_ERR_SHAPE);
    if (dtype == GFE_F32) return pscan_fwd_launch<float>(A, X, H, ws, B, L, DN, T, (hipStream_t)stream);
    if (dtype == GFE_BF16) return pscan_fwd_launch<bf16_t>(A, X, H, ws, B, L, DN, T, (hipStream_t)stream);
    return GFE_ERR_DTYPE;
}

int gfe_pscan_bwd(const void* A, const void* H, const void* gH, void* gA, void* gX, float* ws,
                  int64_t B, int64_t L, int64_t DN, int T, int dtype, void* stream) {
    GFE_REQUIRE(A && H && gH && gA && gX, GFE_ERR_NULL);
    GFE_REQUIRE(B > 0 && B <= 65535 && L > 0 && DN > 0 && T > 0 && ceil_div(L, T) <= 65535, GFE_ERR_SHAPE);
    if (dtype == GFE_F32) return pscan_bwd_launch<float>(A, H, gH, gA, gX, ws, B, L, DN, T, (hipStream_t)stream);
    if (dtype == GFE_BF16) return pscan_bwd_launch<bf16_t>(A, H, gH, gA, gX, ws, B, L, DN, T, (hipStream_t)stream);
    return GFE_ERR_DTYPE;
}

}  // extern "C"
