// Shared device/host helpers for the gfx950 kernels behind the C-ABI in include/gfe_hip.h.
// gfx950 (MI355X / CDNA4) only: wave = 64 lanes, no other targets, no compatibility paths.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/gfe_hip.h"
#include "diag_guard.h"
#if defined(GFE_DIAG)   // a diagnostic library identifies itself (tests/test_build_resources.py: the product library must not export this)
extern "C" __attribute__((weak, visibility("default"))) int gfe_diag_build(void) { return 1; }
#endif

#define GFE_WAVE 64

// ---- error plumbing: entry points never throw, never sync, never allocate -------------------
#define GFE_REQUIRE(cond, code) do { if (!(cond)) return (code); } while (0)
static inline int gfe_launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? GFE_OK : GFE_ERR_HIP;
}

// ---- bf16 <-> f32 ------------------------------------------------------------------------------
typedef uint16_t bf16_t;   // raw storage; arithmetic is always f32

__device__ __forceinline__ float bf16_to_f32(bf16_t v) {
    return __uint_as_float(((uint32_t)v) << 16);
}
// plain cast => v_cvt_pk_bf16_f32 (RNE, NaN stays NaN) on gfx950
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
    __bf16 b = (__bf16)f;
    return __builtin_bit_cast(uint16_t, b);
}
// one v_cvt_pk_bf16_f32 for the pair (the scalar casts above cost two converts plus a shift/or to merge)
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    typedef __attribute__((ext_vector_type(2))) float f32x2_t;
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
    const f32x2_t v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ float bf16lo_to_f32(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf16hi_to_f32(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }

template <typename T> struct IO;
template <> struct IO<float> {
    static __device__ __forceinline__ float ld(const float* p) { return *p; }
    static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct IO<bf16_t> {
    static __device__ __forceinline__ float ld(const bf16_t* p) { return bf16_to_f32(*p); }
    static __device__ __forceinline__ void st(bf16_t* p, float v) { *p = f32_to_bf16(v); }
};

// ---- math --------------------------------------------------------------------------------------
#define GFE_LOG2E 1.4426950408889634f
#define GFE_LN2   0.6931471805599453f
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * GFE_LOG2E); }
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float fast_log(float x) { return __builtin_amdgcn_logf(x) * GFE_LN2; }   // v_log_f32 is log2
__device__ __forceinline__ float sigmoidf_(float x) { return fast_rcp(1.0f + fast_exp2(-x * GFE_LOG2E)); }
__device__ __forceinline__ float siluf_(float x) { return x * sigmoidf_(x); }
// torch.nn.functional.softplus(beta=1, threshold=20) = max(x,0) + log1p(exp(-|x|)); short series keeps relative
// accuracy where 1+e would round away e (the reference's dt_proj.bias puts softplus outputs down to 1e-4).
__device__ __forceinline__ float softplusf_(float x) {
    if (x > 20.0f) return x;
    const float e = fast_exp2(-fabsf(x) * GFE_LOG2E);
    const float l = e < 0.0078125f ? e * fmaf(e, fmaf(e, 0.33333333f, -0.5f), 1.0f) : fast_log(1.0f + e);
    return fmaxf(x, 0.0f) + l;
}

// ---- wave / block reductions -----------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
// block-wide sum for blockDim.x <= 1024 (multiple of 64); scratch must hold 16 floats; result in all threads
__device__ __forceinline__ float block_sum(float v, float* scratch) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch[w] = v;
    __syncthreads();
    float r = 0.f;
    for (int i = 0; i < nw; ++i) r += scratch[i];
    return r;
}

static inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// ---- timing fuzz (tools/timing_fuzz.py) -----------------------------------------------------------------------------
// `make fuzz` builds gfe_hip/libgfe_hip_fuzz.so with -DGFE_TIMING_FUZZ: every GFE_FUZZ() site -- in front of the counted / full
// s_waitcnt vmcnt waits, in front of and behind the stage barriers, in front of the LDS-DMA bursts and of every ticket draw / read of the
// persistent kernels -- then parks its WAVE for a pseudo-random time: nothing at 3 of 4 visits, 1..32 x 64 cycles at 1 of 4, and 16k-32k
// cycles (several tiles) at 1 of 128.  The generator state is one SGPR per wave, seeded from the real-time counter, the block and the wave,
// so two launches never see the same schedule.  A kernel whose result depends on the relative timing of its waves (a missing wait, an LDS
// buffer reused one barrier early, a ticket read before it was published) shows up as a bit difference against the product library;
// a correct one is bit-identical under any schedule.  The product library compiles the sites to nothing.
#if defined(GFE_TIMING_FUZZ)
__device__ __forceinline__ unsigned gfe_fuzz_init() {
    const unsigned t = (unsigned)__builtin_amdgcn_s_memrealtime();
    unsigned s = t * 2654435761u ^ (blockIdx.x * 40503u + blockIdx.y * 9973u + 1u) * 2246822519u ^ ((threadIdx.x >> 6) + 1u) * 3266489917u;
    return (unsigned)__builtin_amdgcn_readfirstlane((int)s);
}
__device__ __forceinline__ void gfe_fuzz_point(unsigned& s) {
    s = (unsigned)__builtin_amdgcn_readfirstlane((int)(s * 1664525u + 1013904223u));
    const unsigned r = s >> 20;                                        // 12 bits
#if defined(GFE_TIMING_FUZZ_HEAVY)     // second level (make fuzz FUZZ_EXTRA=-DGFE_TIMING_FUZZ_HEAVY): every other visit sleeps, a long sleep at 1 of 32
    unsigned n = (r & 1u) == 0u ? ((r >> 1) & 63u) + 1u : 0u;
    if ((r >> 7) == 31u) n = 256u + ((r & 127u) << 2);
#else
    unsigned n = (r & 3u) == 0u ? ((r >> 2) & 31u) + 1u : 0u;
    if ((r >> 5) == 127u) n = 256u + ((r & 31u) << 3);
#endif
    for (unsigned i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(1);
}
#define GFE_FUZZ_INIT() unsigned gfe_fz_ = gfe_fuzz_init()
#define GFE_FUZZ() gfe_fuzz_point(gfe_fz_)
#else
#define GFE_FUZZ_INIT() do {} while (0)
#define GFE_FUZZ() do {} while (0)
#endif

// ---- zero fill as a KERNEL --------------------------------------------------------------------------------------------------------
// hipMemsetAsync inside a stream capture becomes a memset NODE, and on this ROCm graph replay does not keep such a node ordered behind
// the kernel nodes in front of it: when the target block of the graph's private pool had an earlier tenant in the same graph, the
// replayed memset clobbers that tenant's live data (tools/graph_gen_bisect.py: every piece of the generator replays bit-exactly on its
// own, the composition returned garbage from the second replay on, and downstream kernels then died with HSA_STATUS_ERROR_EXCEPTION).
// A fill kernel is an ordinary kernel node.  bytes must be a multiple of 4, p 4-byte aligned.
static __global__ __launch_bounds__(256) void gfe_zero_words_kernel(uint32_t* __restrict__ p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = 0u;
}
static inline void gfe_zero_async(void* p, size_t bytes, hipStream_t st) {
    const size_t n = bytes / 4;
    if (n == 0) return;
    size_t blocks = (n + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(gfe_zero_words_kernel, dim3((unsigned)blocks), dim3(256), 0, st, (uint32_t*)p, n);
}
