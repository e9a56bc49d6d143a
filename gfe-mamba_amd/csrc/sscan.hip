// Fused selective scan (Mamba S6 recurrence) for gfx950 -- forward and backward.
//
// Replaces the materialised PyTorch path of the reference:
//   cross_atten/mamba.py:265-286  MambaBlock.selective_scan  (deltaA/deltaB/BX -> pscan -> hs@C + D*x)
//   cross_atten/mamba.py:243-259  softplus(delta + dt_proj.bias) and the y*silu(z) gate that the
//                                 reference's optional `selective_scan_fn` plug-in fuses
//   cross_atten/pscan.py:151-224  PScan.forward/backward (the recurrence and its adjoint)
//
// Layout (token-major, the layout MambaBlock already has before its transposes at mamba.py:245-252):
//   u, delta, z, y : (B, L, ED)   ED contiguous  -> one wave = 64 consecutive channels, 128-B rows (bf16)
//   Bm, Cm         : (B, L, N)    N contiguous   -> wave-uniform, fetched on the scalar path
//   A              : (ED, N) f32, D / delta_bias : (ED) f32
// State is always f32 (mamba.py:232: A is .float()).  No MFMA: ~6 flop per state-step, HBM/VALU bound.
//
// Parallel decomposition: one lane owns one channel and keeps its N states in registers.  L is cut
// into chunks of T steps so that B*ED/64*nchunks waves fill the chip:
//   K1  chunk_state : local end state of every chunk from h=0, plus sum(delta) of the chunk
//                     (prod_t exp(delta_t*A) == exp(A * sum delta) -> the chunk's decay needs no products)
//   K2  carry       : sequential over chunks (tiny): turns local states into chunk-start states
//   K3  full        : re-runs every chunk from its true start state and writes y
// nchunks == 1 skips K1/K2.  The backward uses the same structure mirrored in time (K1' K2' K3').
#include "common.h"

namespace {

struct SScanParams {
    const void* u; const void* delta; const void* z; const void* Bm; const void* Cm;
    const float* A; const float* D; const float* dbias;
    void* y;
    float* hstate;      // (B, nchunks, N, ED)
    float* sdelta;      // (B, nchunks, ED)
    int B, L, ED, T, nchunks, softplus;
};

// wave-uniform row of N values (N contiguous) -> f32 registers (scalar path when the compiler proves uniformity)
template <typename T, int N> struct RowLd;
template <int N> struct RowLd<float, N> {
    static __device__ __forceinline__ void ld(const float* __restrict__ p, float (&o)[N]) {
#pragma unroll
        for (int i = 0; i < N; ++i) o[i] = p[i];
    }
};
template <int N> struct RowLd<bf16_t, N> {
    static __device__ __forceinline__ void ld(const bf16_t* __restrict__ p, float (&o)[N]) {
        const uint32_t* __restrict__ w = reinterpret_cast<const uint32_t*>(p);
#pragma unroll
        for (int i = 0; i < N / 2; ++i) { uint32_t v = w[i]; o[2 * i] = bf16lo_to_f32(v); o[2 * i + 1] = bf16hi_to_f32(v); }
    }
};

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
template <typename T, int N, bool STATE_ONLY>
__global__ __launch_bounds__(256) void sscan_fwd_kernel(const SScanParams p) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    const int c = blockIdx.y, b = blockIdx.z;
    if (e >= p.ED) return;
    const int t0 = c * p.T, t1 = min(p.L, t0 + p.T);
    const T* __restrict__ u = (const T*)p.u;
    const T* __restrict__ dl = (const T*)p.delta;
    const T* __restrict__ z = (const T*)p.z;
    const T* __restrict__ Bm = (const T*)p.Bm;
    const T* __restrict__ Cm = (const T*)p.Cm;
    T* __restrict__ y = (T*)p.y;

    float A2[N], h[N];
#pragma unroll
    for (int n = 0; n < N; ++n) A2[n] = p.A[(size_t)e * N + n] * GFE_LOG2E;
    const size_t sbase = ((size_t)b * p.nchunks + c) * N * p.ED + e;
    if (!STATE_ONLY && p.nchunks > 1) {
#pragma unroll
        for (int n = 0; n < N; ++n) h[n] = p.hstate[sbase + (size_t)n * p.ED];
    } else {
#pragma unroll
        for (int n = 0; n < N; ++n) h[n] = 0.f;
    }
    const float bias = p.dbias ? p.dbias[e] : 0.f;
    const float Dv = (!STATE_ONLY && p.D) ? p.D[e] : 0.f;
    float sd = 0.f;

#pragma unroll 4
    for (int t = t0; t < t1; ++t) {
        const size_t row = (size_t)b * p.L + t;
        const size_t off = row * p.ED + e;
        float dt = IO<T>::ld(dl + off) + bias;
        if (p.softplus) dt = softplusf_(dt);
        const float uu = IO<T>::ld(u + off);
        const float dtu = dt * uu;
        float Bv[N];
        RowLd<T, N>::ld(Bm + row * N, Bv);
        if (STATE_ONLY) {
            sd += dt;
#pragma unroll
            for (int n = 0; n < N; ++n) h[n] = fmaf(fast_exp2(dt * A2[n]), h[n], dtu * Bv[n]);
        } else {
            float Cv[N];
            RowLd<T, N>::ld(Cm + row * N, Cv);
            float acc0 = 0.f, acc1 = 0.f;
#pragma unroll
            for (int n = 0; n < N; n += 2) {
                h[n] = fmaf(fast_exp2(dt * A2[n]), h[n], dtu * Bv[n]);
                h[n + 1] = fmaf(fast_exp2(dt * A2[n + 1]), h[n + 1], dtu * Bv[n + 1]);
                acc0 = fmaf(h[n], Cv[n], acc0);
                acc1 = fmaf(h[n + 1], Cv[n + 1], acc1);
            }
            float yv = fmaf(Dv, uu, acc0 + acc1);
            if (z) yv *= siluf_(IO<T>::ld(z + off));
            IO<T>::st(y + off, yv);
        }
    }
    if (STATE_ONLY) {
#pragma unroll
        for (int n = 0; n < N; ++n) p.hstate[sbase + (size_t)n * p.ED] = h[n];
        p.sdelta[((size_t)b * p.nchunks + c) * p.ED + e] = sd;
    }
}

// K2: hstate[b,c,n,e] (local end state of chunk c) -> start state of chunk c.  REVERSE=true runs the
// adjoint carry (chunk c receives from chunk c+1).  One lane per (n,e); nchunks sequential steps.
template <int N, bool REVERSE>
__global__ __launch_bounds__(256) void sscan_carry_kernel(float* __restrict__ hstate, const float* __restrict__ sdelta,
                                                          const float* __restrict__ A, int ED, int nchunks) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;   // n*ED + e
    const int b = blockIdx.y;
    if (i >= N * ED) return;
    const int n = i / ED, e = i - n * ED;
    const float A2 = A[(size_t)e * N + n] * GFE_LOG2E;
    float H = 0.f;
#pragma unroll 4
    for (int k = 0; k < nchunks; ++k) {
        const int c = REVERSE ? nchunks - 1 - k : k;
        const size_t idx = ((size_t)b * nchunks + c) * N * ED + i;
        const float loc = hstate[idx];
        const float sd = sdelta[((size_t)b * nchunks + c) * ED + e];
        hstate[idx] = H;
        H = fmaf(fast_exp2(A2 * sd), H, loc);
    }
}

// ------------------------------------------------------------------------------------------------
// backward
// ------------------------------------------------------------------------------------------------
struct SScanBwdParams {
    const void* u; const void* delta; const void* z; const void* Bm; const void* Cm; const void* dy;
    const float* A; const float* D; const float* dbias;
    void* du; void* ddelta; void* dz;
    float* dBws; float* dCws;          // (B, L, N) f32, zero-initialised by the caller, atomically accumulated
    float* dAws;                        // (N, ED) f32 (transposed for 256-B atomic rows), zero-initialised
    float* dDws; float* dbiasws;        // (ED) f32, zero-initialised
    const float* hstate;                // (B, nchunks, N, ED) chunk-start states (forward K2 output)
    float* qstate;                      // (B, nchunks, N, ED) adjoint carry
    const float* sdelta;                // (B, nchunks, ED)
    int B, L, ED, T, nchunks, softplus;
};

// K1': local adjoint of each chunk with zero incoming carry: q <- a_t * (C_t g_t + q), t descending.
template <typename T, int N>
__global__ __launch_bounds__(256) void sscan_bwd_state_kernel(const SScanBwdParams p) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    const int c = blockIdx.y, b = blockIdx.z;
    if (e >= p.ED) return;
    const int t0 = c * p.T, t1 = min(p.L, t0 + p.T);
    const T* __restrict__ dl = (const T*)p.delta;
    const T* __restrict__ z = (const T*)p.z;
    const T* __restrict__ Cm = (const T*)p.Cm;
    const T* __restrict__ dy = (const T*)p.dy;
    float A2[N], q[N];
#pragma unroll
    for (int n = 0; n < N; ++n) { A2[n] = p.A[(size_t)e * N + n] * GFE_LOG2E; q[n] = 0.f; }
    const float bias = p.dbias ? p.dbias[e] : 0.f;
#pragma unroll 4
    for (int t = t1 - 1; t >= t0; --t) {
        const size_t row = (size_t)b * p.L + t;
        const size_t off = row * p.ED + e;
        float dt = IO<T>::ld(dl + off) + bias;
        if (p.softplus) dt = softplusf_(dt);
        float g = IO<T>::ld(dy + off);
        if (z) g *= siluf_(IO<T>::ld(z + off));
        float Cv[N];
        RowLd<T, N>::ld(Cm + row * N, Cv);
#pragma unroll
        for (int n = 0; n < N; ++n) q[n] = fast_exp2(dt * A2[n]) * fmaf(Cv[n], g, q[n]);
    }
    const size_t sbase = ((size_t)b * p.nchunks + c) * N * p.ED + e;
#pragma unroll
    for (int n = 0; n < N; ++n) p.qstate[sbase + (size_t)n * p.ED] = q[n];
}

// Sum v[n] over the 64 lanes of the wave; lane l returns the total for n == (l & (N-1)).
// Transposing butterfly: N-1 + log2(64/N) shuffles instead of N*6.
template <int N>
__device__ __forceinline__ float wave_reduce_vec(float (&v)[N]) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int half = N / 2; half >= 1; half >>= 1) {
        const bool up = (lane & half) != 0;
#pragma unroll
        for (int i = 0; i < half; ++i) {
            const float keep = up ? v[half + i] : v[i];
            const float send = up ? v[i] : v[half + i];
            v[i] = keep + __shfl_xor(send, half, 64);
        }
    }
    float r = v[0];
#pragma unroll
    for (int o = N; o < 64; o <<= 1) r += __shfl_xor(r, o, 64);
    return r;
}

// K3': one wave per block (64 channels), chunk of T steps processed as sub-chunks of S steps:
//   sweep 1 (forward): chunk-start state -> state at every sub-chunk start, parked in LDS
//   sweep 2 (sub-chunks in reverse): recompute the S states + decays in registers, then run the adjoint.
template <typename T, int N, int S>
__global__ __launch_bounds__(64) void sscan_bwd_kernel(const SScanBwdParams p) {
    extern __shared__ __attribute__((aligned(16))) float ck[];   // [nsub][N][64]
    const int lane = threadIdx.x;
    const int e = blockIdx.x * 64 + lane;
    const int c = blockIdx.y, b = blockIdx.z;
    const int t0 = c * p.T, t1 = min(p.L, t0 + p.T);
    const int nsub = (t1 - t0 + S - 1) / S;
    const T* __restrict__ u = (const T*)p.u;
    const T* __restrict__ dl = (const T*)p.delta;
    const T* __restrict__ z = (const T*)p.z;
    const T* __restrict__ Bm = (const T*)p.Bm;
    const T* __restrict__ Cm = (const T*)p.Cm;
    const T* __restrict__ dy = (const T*)p.dy;
    T* __restrict__ du = (T*)p.du;
    T* __restrict__ dd = (T*)p.ddelta;
    T* __restrict__ dz = (T*)p.dz;

    float A2[N], h[N];
#pragma unroll
    for (int n = 0; n < N; ++n) A2[n] = p.A[(size_t)e * N + n] * GFE_LOG2E;
    const size_t sbase = ((size_t)b * p.nchunks + c) * N * p.ED + e;
    if (p.nchunks > 1) {
#pragma unroll
        for (int n = 0; n < N; ++n) h[n] = p.hstate[sbase + (size_t)n * p.ED];
    } else {
#pragma unroll
        for (int n = 0; n < N; ++n) h[n] = 0.f;
    }
    const float bias = p.dbias ? p.dbias[e] : 0.f;
    const float Dv = p.D ? p.D[e] : 0.f;

    // sweep 1
    for (int j = 0; j < nsub; ++j) {
#pragma unroll
        for (int n = 0; n < N; ++n) ck[(j * N + n) * 64 + lane] = h[n];
        if (j == nsub - 1) break;
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const int t = t0 + j * S + s;
            const size_t row = (size_t)b * p.L + t;
            const size_t off = row * p.ED + e;
            float dt = IO<T>::ld(dl + off) + bias;
            if (p.softplus) dt = softplusf_(dt);
            const float dtu = dt * IO<T>::ld(u + off);
            float Bv[N];
            RowLd<T, N>::ld(Bm + row * N, Bv);
#pragma unroll
            for (int n = 0; n < N; ++n) h[n] = fmaf(fast_exp2(dt * A2[n]), h[n], dtu * Bv[n]);
        }
    }

    float q[N], dAacc[N];
    if (p.nchunks > 1) {
#pragma unroll
        for (int n = 0; n < N; ++n) q[n] = p.qstate[sbase + (size_t)n * p.ED];
    } else {
#pragma unroll
        for (int n = 0; n < N; ++n) q[n] = 0.f;
    }
#pragma unroll
    for (int n = 0; n < N; ++n) dAacc[n] = 0.f;
    float dDacc = 0.f, dbacc = 0.f;

    // sweep 2
    for (int j = nsub - 1; j >= 0; --j) {
        const int ts = t0 + j * S;
        float hs[S][N], as[S][N], dts[S], us[S], sg[S];
        float h0[N];
#pragma unroll
        for (int n = 0; n < N; ++n) h0[n] = ck[(j * N + n) * 64 + lane];
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const int t = ts + s;
            const bool live = t < t1;
            const size_t row = (size_t)b * p.L + (live ? t : t1 - 1);
            const size_t off = row * p.ED + e;
            const float draw = IO<T>::ld(dl + off) + bias;
            float dt = draw;
            float sgm = 1.f;
            if (p.softplus) { dt = softplusf_(draw); sgm = draw > 20.f ? 1.f : sigmoidf_(draw); }
            dts[s] = dt; sg[s] = sgm;
            us[s] = IO<T>::ld(u + off);
            const float dtu = dt * us[s];
            float Bv[N];
            RowLd<T, N>::ld(Bm + row * N, Bv);
#pragma unroll
            for (int n = 0; n < N; ++n) {
                as[s][n] = fast_exp2(dt * A2[n]);
                const float hp = (s == 0) ? h0[n] : hs[s - 1][n];
                hs[s][n] = fmaf(as[s][n], hp, dtu * Bv[n]);
            }
        }
        float dBrow[S], dCrow[S];
#pragma unroll
        for (int s = S - 1; s >= 0; --s) {
            const int t = ts + s;
            dBrow[s] = 0.f; dCrow[s] = 0.f;
            if (t < t1) {   // wave-uniform
                const size_t row = (size_t)b * p.L + t;
                const size_t off = row * p.ED + e;
                float Bv[N], Cv[N];
                RowLd<T, N>::ld(Bm + row * N, Bv);
                RowLd<T, N>::ld(Cm + row * N, Cv);
                const float dyv = IO<T>::ld(dy + off);
                float g = dyv;
                if (z) {
                    const float zv = IO<T>::ld(z + off);
                    const float sz = sigmoidf_(zv);
                    float yss = Dv * us[s];
#pragma unroll
                    for (int n = 0; n < N; ++n) yss = fmaf(hs[s][n], Cv[n], yss);
                    // d/dz [y*z*sigmoid(z)] = y*sigmoid(z)*(1 + z*(1-sigmoid(z)))
                    IO<T>::st(dz + off, dyv * yss * sz * (1.f + zv * (1.f - sz)));
                    g = dyv * zv * sz;
                }
                float ddt = 0.f, dub = 0.f;
                float dBv[N], dCv[N];
                const float dtu = dts[s] * us[s];
#pragma unroll
                for (int n = 0; n < N; ++n) {
                    const float dh = fmaf(Cv[n], g, q[n]);
                    const float hp = (s == 0) ? h0[n] : hs[s - 1][n];
                    const float da = dh * hp * as[s][n];        // dL/d(dt*A) for this (t,n)
                    dAacc[n] = fmaf(da, dts[s], dAacc[n]);
                    ddt = fmaf(da, A2[n], ddt);                 // A2 = A*log2e, rescaled below
                    dub = fmaf(dh, Bv[n], dub);
                    dBv[n] = dh * dtu;
                    dCv[n] = hs[s][n] * g;
                    q[n] = as[s][n] * dh;
                }
                ddt = ddt * GFE_LN2 + dub * us[s];
                const float ddraw = ddt * sg[s];
                IO<T>::st(dd + off, ddraw);
                IO<T>::st(du + off, fmaf(dub, dts[s], Dv * g));
                dDacc = fmaf(g, us[s], dDacc);
                dbacc += ddraw;
                dBrow[s] = wave_reduce_vec<N>(dBv);
                dCrow[s] = wave_reduce_vec<N>(dCv);
            }
        }
        // lane l holds the (t = ts + s, n = l % N) totals for every s; emit S*N contiguous floats per atomic
        constexpr int STEPS_PER_WAVE = 64 / N;     // timesteps covered by one 64-lane atomic
#pragma unroll
        for (int s0 = 0; s0 < S; s0 += STEPS_PER_WAVE) {
            float vb = 0.f, vc = 0.f;
#pragma unroll
            for (int k = 0; k < STEPS_PER_WAVE; ++k) {
                if (s0 + k < S && (lane / N) == k) { vb = dBrow[s0 + k]; vc = dCrow[s0 + k]; }
            }
            const int t = ts + s0 + lane / N;
            if ((s0 + lane / N) < S && t < t1) {
                const size_t o = ((size_t)b * p.L + t) * N + (lane % N);
                atomicAdd(p.dBws + o, vb);
                atomicAdd(p.dCws + o, vc);
            }
        }
    }
#pragma unroll
    for (int n = 0; n < N; ++n) atomicAdd(p.dAws + (size_t)n * p.ED + e, dAacc[n]);
    if (p.dDws) atomicAdd(p.dDws + e, dDacc);
    if (p.dbiasws) atomicAdd(p.dbiasws + e, dbacc);
}

// chunk length so that the launch has enough waves to fill 256 CUs x 8 waves
static int pick_chunk(int64_t B, int64_t L, int64_t ED, int req) {
    if (req > 0) return (int)(req < L ? req : L);
    const int64_t waves_per_chunk = B * ceil_div(ED, 64);
    int64_t want = ceil_div(2048, waves_per_chunk);           // chunks needed for ~2048 waves
    if (want <= 1) return (int)L;
    int64_t T = ceil_div(L, want);
    const int64_t tmin = L <= 256 ? 8 : 32;      // short sequences (the model's L = 37): favour parallel chunks over state traffic
    if (T < tmin) T = tmin;
    T = ceil_div(T, 8) * 8;
    return (int)(T < L ? T : L);
}

template <typename T, int N>
int sscan_fwd_launch(const SScanParams& p, hipStream_t st) {
    const dim3 blk(256), grid((unsigned)ceil_div(p.ED, 256), p.nchunks, p.B);
    if (p.nchunks > 1) {
        hipLaunchKernelGGL((sscan_fwd_kernel<T, N, true>), grid, blk, 0, st, p);
        hipLaunchKernelGGL((sscan_carry_kernel<N, false>), dim3((unsigned)ceil_div((int64_t)N * p.ED, 256), p.B), blk, 0, st,
                           p.hstate, p.sdelta, p.A, p.ED, p.nchunks);
    }
    hipLaunchKernelGGL((sscan_fwd_kernel<T, N, false>), grid, blk, 0, st, p);
    return gfe_launch_status();
}

template <typename T, int N>
int sscan_bwd_launch(const SScanBwdParams& p, hipStream_t st) {
    constexpr int S = 4;
    if (p.nchunks > 1) {
        const dim3 blk(256), grid((unsigned)ceil_div(p.ED, 256), p.nchunks, p.B);
        hipLaunchKernelGGL((sscan_bwd_state_kernel<T, N>), grid, blk, 0, st, p);
        hipLaunchKernelGGL((sscan_carry_kernel<N, true>), dim3((unsigned)ceil_div((int64_t)N * p.ED, 256), p.B), blk, 0, st,
                           p.qstate, p.sdelta, p.A, p.ED, p.nchunks);
    }
    const int nsub = (p.T + S - 1) / S;
    const size_t lds = (size_t)nsub * N * 64 * sizeof(float);
    if (lds > 160 * 1024) return GFE_ERR_SHAPE;
    static size_t attr_lds = 0;
    if (lds > attr_lds) { (void)hipFuncSetAttribute((const void*)sscan_bwd_kernel<T, N, S>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr_lds = lds; }
    hipLaunchKernelGGL((sscan_bwd_kernel<T, N, S>), dim3(p.ED / 64, p.nchunks, p.B), dim3(64), lds, st, p);
    return gfe_launch_status();
}

}  // namespace

extern "C" {

int gfe_sscan_plan(int64_t B, int64_t L, int64_t ED, int64_t N, int chunk_req, int backward, int* T_out, int* nchunks_out) {
    GFE_REQUIRE(B > 0 && L > 0 && ED > 0 && (N == 4 || N == 8 || N == 16), GFE_ERR_SHAPE);
    int T = pick_chunk(B, L, ED, chunk_req);
    if (backward) {
        // LDS checkpoints: nsub*N*256 B per wave; keep <= 32 KB so >= 4 waves share a CU
        const int tmax = (int)(32 * 1024 / (N * 256)) * 4;
        if (T > tmax) T = tmax;
    }
    *T_out = T;
    *nchunks_out = (int)ceil_div(L, T);
    return GFE_OK;
}

int gfe_selective_scan_fwd(const void* u, const void* delta, const float* A, const void* Bm, const void* Cm,
                           const float* D, const void* z, const float* delta_bias, void* y,
                           float* hstate, float* sdelta,
                           int64_t B, int64_t L, int64_t ED, int64_t N, int T, int delta_softplus,
                           int dtype, void* stream) {
    GFE_REQUIRE(u && delta && A && Bm && Cm && y, GFE_ERR_NULL);
    GFE_REQUIRE(B > 0 && L > 0 && ED > 0 && T > 0, GFE_ERR_SHAPE);
    GFE_REQUIRE(B <= 65535, GFE_ERR_SHAPE);
    SScanParams p;
    p.u = u; p.delta = delta; p.z = z; p.Bm = Bm; p.Cm = Cm; p.A = A; p.D = D; p.dbias = delta_bias; p.y = y;
    p.hstate = hstate; p.sdelta = sdelta;
    p.B = (int)B; p.L = (int)L; p.ED = (int)ED; p.T = T; p.nchunks = (int)ceil_div(L, T); p.softplus = delta_softplus;
    GFE_REQUIRE(p.nchunks <= 65535, GFE_ERR_SHAPE);
    GFE_REQUIRE(p.nchunks == 1 || (hstate && sdelta), GFE_ERR_NULL);
    hipStream_t st = (hipStream_t)stream;
#define GFE_DISPATCH(TT)                                                          \
    switch (N) {                                                                  \
        case 4: return sscan_fwd_launch<TT, 4>(p, st);                            \
        case 8: return sscan_fwd_launch<TT, 8>(p, st);                            \
        case 16: return sscan_fwd_launch<TT, 16>(p, st);                          \
        default: return GFE_ERR_SHAPE;                                            \
    }
    if (dtype == GFE_F32) { GFE_DISPATCH(float) }
    if (dtype == GFE_BF16) { GFE_DISPATCH(bf16_t) }
#undef GFE_DISPATCH
    return GFE_ERR_DTYPE;
}

int gfe_selective_scan_bwd(const void* u, const void* delta, const float* A, const void* Bm, const void* Cm,
                           const float* D, const void* z, const float* delta_bias, const void* dy,
                           void* du, void* ddelta, void* dz,
                           float* dA_ws, float* dB_ws, float* dC_ws, float* dD_ws, float* dbias_ws,
                           const float* hstate, float* qstate, const float* sdelta,
                           int64_t B, int64_t L, int64_t ED, int64_t N, int T, int delta_softplus,
                           int dtype, void* stream) {
    GFE_REQUIRE(u && delta && A && Bm && Cm && dy && du && ddelta && dA_ws && dB_ws && dC_ws, GFE_ERR_NULL);
    GFE_REQUIRE(!z || dz, GFE_ERR_NULL);
    GFE_REQUIRE(B > 0 && L > 0 && ED > 0 && T > 0 && ED % 64 == 0, GFE_ERR_SHAPE);
    GFE_REQUIRE(B <= 65535, GFE_ERR_SHAPE);
    SScanBwdParams p;
    p.u = u; p.delta = delta; p.z = z; p.Bm = Bm; p.Cm = Cm; p.dy = dy; p.A = A; p.D = D; p.dbias = delta_bias;
    p.du = du; p.ddelta = ddelta; p.dz = dz; p.dBws = dB_ws; p.dCws = dC_ws; p.dAws = dA_ws; p.dDws = dD_ws; p.dbiasws = dbias_ws;
    p.hstate = hstate; p.qstate = qstate; p.sdelta = sdelta;
    p.B = (int)B; p.L = (int)L; p.ED = (int)ED; p.T = T; p.nchunks = (int)ceil_div(L, T); p.softplus = delta_softplus;
    GFE_REQUIRE(p.nchunks <= 65535, GFE_ERR_SHAPE);
    GFE_REQUIRE(p.nchunks == 1 || (hstate && qstate && sdelta), GFE_ERR_NULL);
    hipStream_t st = (hipStream_t)stream;
#define GFE_DISPATCH(TT)                                                          \
    switch (N) {                                                                  \
        case 4: return sscan_bwd_launch<TT, 4>(p, st);                            \
        case 8: return sscan_bwd_launch<TT, 8>(p, st);                            \
        case 16: return sscan_bwd_launch<TT, 16>(p, st);                          \
        default: return GFE_ERR_SHAPE;                                            \
    }
    if (dtype == GFE_F32) { GFE_DISPATCH(float) }
    if (dtype == GFE_BF16) { GFE_DISPATCH(bf16_t) }
#undef GFE_DISPATCH
    return GFE_ERR_DTYPE;
}

}  // extern "C"
