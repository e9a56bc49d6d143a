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
//   Bm, Cm         : (B, L, N)    N contiguous f32 -> wave-uniform rows, staged through wave-private LDS tiles (RowTile)
//   A              : (ED, N) f32, D / delta_bias : (ED) f32
// State is always f32 (mamba.py:232: A is .float()), kept as float2 pairs so that the mul/fma chains are v_pk_* instructions.
// No MFMA: ~6 flop and one exp per state-step -- VALU-issue bound, not HBM bound.
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
    const void* u; const void* delta; const void* z; const float* Bm; const float* Cm;
    const float* A; const float* D; const float* dbias;
    void* y;
    float* hstate;      // (B, nchunks, N, ED)
    float* sdelta;      // (B, nchunks, ED)
    int B, L, ED, T, nchunks, softplus;
};

// B / C rows (N f32 each, shared by every channel) reach the lanes through a wave-private LDS tile of TT rows: each lane
// fetches ONE float4 of the (contiguous) TT x N block, parks it in LDS, and every step reads its row back with broadcast
// ds_read_b128.  (Fetching rows on the scalar path looked attractive but the bf16 unpack + addressing cost ~90 SALU
// instructions per step on the CU's single scalar ALU and, in unrolled code, spilled SGPRs into v_readlane/v_writelane chains.)
__device__ __forceinline__ int opaque_zero() { int v; asm volatile("v_mov_b32 %0, 0" : "=v"(v)); return v; }

template <int N, int TT>
struct RowTile {
    static constexpr int F4 = TT * N / 4;                  // float4 per tile (<= 64)
    float4* lds;                                            // this wave's 2 x F4 float4
    const float* g;                                         // (rows, N) f32
    int64_t last_f4;                                        // index of the last valid float4 of the tensor
    int lane;
    int vzero;                                              // opaque per-lane 0: keeps the broadcast rows in VGPRs (the compiler
                                                            // otherwise proves them wave-uniform, moves them to SGPRs and spills)
    __device__ __forceinline__ float4 fetch(int64_t row0) const {   // rows row0 .. row0+TT-1 (clamped at the tensor end)
        if (lane >= F4) return make_float4(0.f, 0.f, 0.f, 0.f);
        int64_t i = row0 * (N / 4) + lane;
        i = i < 0 ? 0 : (i > last_f4 ? last_f4 : i);
        return reinterpret_cast<const float4*>(g)[i];
    }
    __device__ __forceinline__ void park(int buf, const float4& v) const { if (lane < F4) lds[buf * F4 + lane] = v; }
    __device__ __forceinline__ void row(int buf, int r, float (&o)[N]) const {
#pragma unroll
        for (int q = 0; q < N / 4; ++q) {
            const float4 v = lds[buf * F4 + r * (N / 4) + q + vzero];
            o[4 * q] = v.x; o[4 * q + 1] = v.y; o[4 * q + 2] = v.z; o[4 * q + 3] = v.w;
        }
    }
};

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
// One lane = one channel; its u/delta/z values are fetched as register tiles of TT timesteps, the next tile in flight
// (3*TT 128-byte wave loads outstanding) while the current one is consumed: with 2-byte lanes that is what it takes to keep
// ~40 KB per CU in flight (HBM latency x bandwidth); a 4-step unroll left the kernel latency-bound at 8 % of HBM.
template <typename T, int N, bool STATE_ONLY>
__global__ __launch_bounds__(256, STATE_ONLY ? 4 : 3) void sscan_fwd_kernel(const SScanParams p) {
    constexpr int TT = 8;       // the kernel is latency-bound: 8-step tiles keep it under 128 VGPRs (4+ waves per SIMD)
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    const int c = blockIdx.y, b = blockIdx.z;
    if (e >= p.ED) return;
    const int t0 = c * p.T, t1 = min(p.L, t0 + p.T);
    const T* __restrict__ u = (const T*)p.u;
    const T* __restrict__ dl = (const T*)p.delta;
    const T* __restrict__ z = (const T*)p.z;
    T* __restrict__ y = (T*)p.y;
    const bool has_z = !STATE_ONLY && z != nullptr;
    extern __shared__ float4 sm4[];
    constexpr int F4 = RowTile<N, TT>::F4;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int64_t last_f4 = (int64_t)p.B * p.L * (N / 4) - 1;
    const int vz = opaque_zero();
    const RowTile<N, TT> tB{sm4 + wid * 4 * F4, p.Bm, last_f4, lane, vz};
    const RowTile<N, TT> tC{sm4 + wid * 4 * F4 + 2 * F4, p.Cm, last_f4, lane, vz};

    // states as float2 pairs (n = 2k, 2k+1): v_pk_mul_f32 / v_pk_fma_f32 chains without register-pairing moves
    typedef float f2 __attribute__((ext_vector_type(2)));
    constexpr int NP = N / 2;
    f2 A2[NP], h[NP];
#pragma unroll
    for (int k = 0; k < NP; ++k) A2[k] = f2{p.A[(size_t)e * N + 2 * k], p.A[(size_t)e * N + 2 * k + 1]} * GFE_LOG2E;
    const size_t sbase = ((size_t)b * p.nchunks + c) * N * p.ED + e;
    if (!STATE_ONLY && p.nchunks > 1) {
#pragma unroll
        for (int k = 0; k < NP; ++k) h[k] = f2{p.hstate[sbase + (size_t)(2 * k) * p.ED], p.hstate[sbase + (size_t)(2 * k + 1) * p.ED]};
    } else {
#pragma unroll
        for (int k = 0; k < NP; ++k) h[k] = f2{0.f, 0.f};
    }
    auto exp2v = [](f2 x) { return f2{fast_exp2(x.x), fast_exp2(x.y)}; };
    const float bias = p.dbias ? p.dbias[e] : 0.f;
    const float Dv = (!STATE_ONLY && p.D) ? p.D[e] : 0.f;
    float sd = 0.f;

    T ub[TT], db[TT], zb[TT], un[TT], dn[TT], zn[TT];
    auto load_tile = [&](int tb, T (&uu)[TT], T (&dd)[TT], T (&zz)[TT]) {
#pragma unroll
        for (int s2 = 0; s2 < TT; ++s2) {
            const int t = min(tb + s2, t1 - 1);                       // clamped: tail steps are loaded but never used
            const size_t off = ((size_t)b * p.L + t) * p.ED + e;
            uu[s2] = u[off];
            dd[s2] = dl[off];
            if (has_z) zz[s2] = z[off];
        }
    };
    load_tile(t0, ub, db, zb);
    tB.park(0, tB.fetch((int64_t)b * p.L + t0));
    if (!STATE_ONLY) tC.park(0, tC.fetch((int64_t)b * p.L + t0));
    int cur = 0;
    for (int tb = t0; tb < t1; tb += TT, cur ^= 1) {
        const bool more = tb + TT < t1;
        float4 fb, fc;
        if (more) {
            load_tile(tb + TT, un, dn, zn);
            fb = tB.fetch((int64_t)b * p.L + tb + TT);
            if (!STATE_ONLY) fc = tC.fetch((int64_t)b * p.L + tb + TT);
        }
#pragma unroll
        for (int s2 = 0; s2 < TT; ++s2) {
            const int t = tb + s2;
            if (t < t1) {                                               // wave-uniform
                const size_t row = (size_t)b * p.L + t;
                float dt = IO<T>::ld(&db[s2]) + bias;
                if (p.softplus) dt = softplusf_(dt);
                const float uu = IO<T>::ld(&ub[s2]);
                const float dtu = dt * uu;
                float Bv[N];
                tB.row(cur, s2, Bv);
                if (STATE_ONLY) {
                    sd += dt;
#pragma unroll
                    for (int k = 0; k < NP; ++k) h[k] = exp2v(A2[k] * dt) * h[k] + f2{Bv[2 * k], Bv[2 * k + 1]} * dtu;
                } else {
                    float Cv[N];
                    tC.row(cur, s2, Cv);
                    f2 acc = f2{0.f, 0.f};
#pragma unroll
                    for (int k = 0; k < NP; ++k) {
                        h[k] = exp2v(A2[k] * dt) * h[k] + f2{Bv[2 * k], Bv[2 * k + 1]} * dtu;
                        acc += h[k] * f2{Cv[2 * k], Cv[2 * k + 1]};
                    }
                    float yv = fmaf(Dv, uu, acc.x + acc.y);
                    if (has_z) yv *= siluf_(IO<T>::ld(&zb[s2]));
                    IO<T>::st(y + row * p.ED + e, yv);
                }
            }
        }
        if (more) {
            tB.park(cur ^ 1, fb);
            if (!STATE_ONLY) tC.park(cur ^ 1, fc);
        }
#pragma unroll
        for (int s2 = 0; s2 < TT; ++s2) { ub[s2] = un[s2]; db[s2] = dn[s2]; zb[s2] = zn[s2]; }
    }
    if (STATE_ONLY) {
#pragma unroll
        for (int k = 0; k < NP; ++k) { p.hstate[sbase + (size_t)(2 * k) * p.ED] = h[k].x; p.hstate[sbase + (size_t)(2 * k + 1) * p.ED] = h[k].y; }
        p.sdelta[((size_t)b * p.nchunks + c) * p.ED + e] = sd;
    }
}

// K2: hstate[b,c,n,e] (local end state of chunk c) -> start state of chunk c.  REVERSE=true runs the
// adjoint carry (chunk c receives from chunk c+1).  One lane per (n,e); the loads of 8 chunks are batched ahead of the
// 8 dependent fma steps so the serial chain costs one memory round trip per 8 chunks.
template <int N, bool REVERSE>
__global__ __launch_bounds__(256) void sscan_carry_kernel(float* __restrict__ hstate, const float* __restrict__ sdelta,
                                                          const float* __restrict__ A, int ED, int nchunks) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;   // n*ED + e
    const int b = blockIdx.y;
    if (i >= N * ED) return;
    const int n = i / ED, e = i - n * ED;
    const float A2 = A[(size_t)e * N + n] * GFE_LOG2E;
    float H = 0.f;
    constexpr int G = 8;
    for (int k0 = 0; k0 < nchunks; k0 += G) {
        float loc[G], sdv[G];
#pragma unroll
        for (int j = 0; j < G; ++j) {
            const int k = min(k0 + j, nchunks - 1);
            const int c = REVERSE ? nchunks - 1 - k : k;
            loc[j] = hstate[((size_t)b * nchunks + c) * N * ED + i];
            sdv[j] = sdelta[((size_t)b * nchunks + c) * ED + e];
        }
#pragma unroll
        for (int j = 0; j < G; ++j) {
            if (k0 + j < nchunks) {
                const int c = REVERSE ? nchunks - 1 - (k0 + j) : k0 + j;
                hstate[((size_t)b * nchunks + c) * N * ED + i] = H;
                H = fmaf(fast_exp2(A2 * sdv[j]), H, loc[j]);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// backward
// ------------------------------------------------------------------------------------------------
struct SScanBwdParams {
    const void* u; const void* delta; const void* z; const float* Bm; const float* Cm; const void* dy;
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

// K1': local adjoint of each chunk with zero incoming carry: q <- a_t * (C_t g_t + q), t descending (16-step register tiles).
template <typename T, int N>
__global__ __launch_bounds__(256, 4) void sscan_bwd_state_kernel(const SScanBwdParams p) {
    constexpr int TT = 8;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    const int c = blockIdx.y, b = blockIdx.z;
    if (e >= p.ED) return;
    const int t0 = c * p.T, t1 = min(p.L, t0 + p.T);
    const T* __restrict__ dl = (const T*)p.delta;
    const T* __restrict__ z = (const T*)p.z;
    const T* __restrict__ dy = (const T*)p.dy;
    const bool has_z = z != nullptr;
    extern __shared__ float4 sm4[];
    constexpr int F4 = RowTile<N, TT>::F4;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const RowTile<N, TT> tC{sm4 + wid * 2 * F4, p.Cm, (int64_t)p.B * p.L * (N / 4) - 1, lane, opaque_zero()};
    float A2[N], q[N];
#pragma unroll
    for (int n = 0; n < N; ++n) { A2[n] = p.A[(size_t)e * N + n] * GFE_LOG2E; q[n] = 0.f; }
    const float bias = p.dbias ? p.dbias[e] : 0.f;
    T db[TT], gb[TT], zb[TT], dn[TT], gn[TT], zn[TT];
    // tile tb covers steps tb-TT+1 .. tb (descending use)
    auto load_tile = [&](int tb, T (&dv)[TT], T (&gv)[TT], T (&zv)[TT]) {
#pragma unroll
        for (int s2 = 0; s2 < TT; ++s2) {
            const size_t off = ((size_t)b * p.L + max(tb - s2, t0)) * p.ED + e;
            dv[s2] = dl[off]; gv[s2] = dy[off];
            if (has_z) zv[s2] = z[off];
        }
    };
    load_tile(t1 - 1, db, gb, zb);
    tC.park(0, tC.fetch((int64_t)b * p.L + t1 - TT));        // tile = rows tb-TT+1 .. tb
    int cur = 0;
    for (int tb = t1 - 1; tb >= t0; tb -= TT, cur ^= 1) {
        const bool more = tb - TT >= t0;
        float4 fc;
        if (more) { load_tile(tb - TT, dn, gn, zn); fc = tC.fetch((int64_t)b * p.L + tb - 2 * TT + 1); }
#pragma unroll
        for (int s2 = 0; s2 < TT; ++s2) {
            const int t = tb - s2;
            if (t >= t0) {
                float dt = IO<T>::ld(&db[s2]) + bias;
                if (p.softplus) dt = softplusf_(dt);
                float g = IO<T>::ld(&gb[s2]);
                if (has_z) g *= siluf_(IO<T>::ld(&zb[s2]));
                float Cv[N];
                tC.row(cur, TT - 1 - s2, Cv);
#pragma unroll
                for (int n = 0; n < N; ++n) q[n] = fast_exp2(dt * A2[n]) * fmaf(Cv[n], g, q[n]);
            }
        }
        if (more) tC.park(cur ^ 1, fc);
#pragma unroll
        for (int s2 = 0; s2 < TT; ++s2) { db[s2] = dn[s2]; gb[s2] = gn[s2]; zb[s2] = zn[s2]; }
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
            // both operands are pinned in registers first: left as `up ? v[half+i] : v[i]` LLVM folds the select into a
            // dynamically indexed array access and lowers it to 16-way v_cmp/v_cndmask chains with SGPR-mask spills
            float lo = v[i], hi = v[half + i];
            asm volatile("" : "+v"(lo), "+v"(hi));
            const float keep = up ? hi : lo;
            const float send = up ? lo : hi;
            v[i] = keep + __shfl_xor(send, half, 64);
        }
    }
    float r = v[0];
#pragma unroll
    for (int o = N; o < 64; o <<= 1) r += __shfl_xor(r, o, 64);
    return r;
}

// dB / dC rows of one timestep together (N == 16): 32 per-lane partials -> one total per lane, VALU only (no LDS crossbar trips,
// no waits): v_permlane32_swap pairs dB[i] with dC[i] (lanes < 32 end up owning dB, lanes >= 32 dC), v_permlane16_swap halves the
// 16 values by lane bit 4, then three DPP reduce-scatter levels inside the 16-lane rows (row_ror:8, row_half_mirror, reversed
// quads: each pairing flips the split bit) and a final quad xor-1 add.  Lane l returns the total of vector (l >> 5) for
// n = (l >> 1) & 15 (lanes l and l ^ 1 hold the same value).
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float wave_reduce_bc16(const float (&dBv)[16], const float (&dCv)[16]) {
    const int lane = threadIdx.x & 63;
    float v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const auto x = __builtin_amdgcn_permlane32_swap(__float_as_uint(dBv[i]), __float_as_uint(dCv[i]), false, false);
        v[i] = __uint_as_float(x[0]) + __uint_as_float(x[1]);
    }
    float w[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const auto x = __builtin_amdgcn_permlane16_swap(__float_as_uint(v[i]), __float_as_uint(v[8 + i]), false, false);
        w[i] = __uint_as_float(x[0]) + __uint_as_float(x[1]);
    }
    float a4[4], a2[2];
    {
        const bool up = (lane & 8) != 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float lo = w[i], hi = w[4 + i];
            asm volatile("" : "+v"(lo), "+v"(hi));
            a4[i] = (up ? hi : lo) + dpp_mov<0x128>(up ? lo : hi);          // row_ror:8 == lane ^ 8
        }
    }
    {
        const bool up = (lane & 4) != 0;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            float lo = a4[i], hi = a4[2 + i];
            asm volatile("" : "+v"(lo), "+v"(hi));
            a2[i] = (up ? hi : lo) + dpp_mov<0x141>(up ? lo : hi);          // row_half_mirror: i <-> 7 - i
        }
    }
    float r;
    {
        const bool up = (lane & 2) != 0;
        float lo = a2[0], hi = a2[1];
        asm volatile("" : "+v"(lo), "+v"(hi));
        r = (up ? hi : lo) + dpp_mov<0x1b>(up ? lo : hi);                   // quad_perm [3,2,1,0]: i <-> 3 - i
    }
    return r + dpp_mov<0xb1>(r);                                            // quad_perm [1,0,3,2]: i <-> i ^ 1
}

// K3': one wave per block (64 channels), chunk of T steps processed as sub-chunks of S steps:
//   sweep 1 (forward): chunk-start state -> state at every sub-chunk start, parked in LDS (u/delta in 16-step register tiles)
//   sweep 2 (sub-chunks in reverse): recompute the S states in registers, then run the adjoint; the u/delta/z/dy values of
//   the NEXT sub-chunk are already in flight.  The decay exp2(delta*A) is recomputed in the adjoint rather than kept.
template <typename T, int N, int S>
__global__ __launch_bounds__(64) void sscan_bwd_kernel(const SScanBwdParams p) {
    extern __shared__ __attribute__((aligned(16))) float ck[];   // [nsub][N][64]
    constexpr int TT = 16;
    const int lane = threadIdx.x;
    const int e = blockIdx.x * 64 + lane;
    const int c = blockIdx.y, b = blockIdx.z;
    const int t0 = c * p.T, t1 = min(p.L, t0 + p.T);
    const int nsub = (t1 - t0 + S - 1) / S;
    const T* __restrict__ u = (const T*)p.u;
    const T* __restrict__ dl = (const T*)p.delta;
    const T* __restrict__ z = (const T*)p.z;
    const T* __restrict__ dy = (const T*)p.dy;
    T* __restrict__ du = (T*)p.du;
    T* __restrict__ dd = (T*)p.ddelta;
    T* __restrict__ dz = (T*)p.dz;
    const bool has_z = z != nullptr;
    // B / C row tiles (aligned to 16 steps from t0) live behind the checkpoints in LDS
    constexpr int F4 = RowTile<N, TT>::F4;
    float4* tiles = reinterpret_cast<float4*>(ck + (size_t)((p.T + S - 1) / S) * N * 64);
    const int64_t last_f4 = (int64_t)p.B * p.L * (N / 4) - 1;
    const int vz = opaque_zero();
    const RowTile<N, TT> tB{tiles, p.Bm, last_f4, lane, vz};
    const RowTile<N, TT> tC{tiles + 2 * F4, p.Cm, last_f4, lane, vz};
    const int64_t row0 = (int64_t)b * p.L + t0;

    // states travel as float2 pairs (n = 2k, 2k+1): the mul / fma chains become v_pk_mul_f32 / v_pk_fma_f32 without the
    // register-pairing moves the compiler needs when it packs scalar code on its own
    typedef float f2 __attribute__((ext_vector_type(2)));
    constexpr int NP = N / 2;
    f2 A2[NP], h[NP];
#pragma unroll
    for (int k = 0; k < NP; ++k) A2[k] = f2{p.A[(size_t)e * N + 2 * k], p.A[(size_t)e * N + 2 * k + 1]} * GFE_LOG2E;
    const size_t sbase = ((size_t)b * p.nchunks + c) * N * p.ED + e;
    if (p.nchunks > 1) {
#pragma unroll
        for (int k = 0; k < NP; ++k) h[k] = f2{p.hstate[sbase + (size_t)(2 * k) * p.ED], p.hstate[sbase + (size_t)(2 * k + 1) * p.ED]};
    } else {
#pragma unroll
        for (int k = 0; k < NP; ++k) h[k] = f2{0.f, 0.f};
    }
    auto exp2v = [](f2 x) { return f2{fast_exp2(x.x), fast_exp2(x.y)}; };
    const float bias = p.dbias ? p.dbias[e] : 0.f;
    const float Dv = p.D ? p.D[e] : 0.f;

    // ---- sweep 1: checkpoints at every sub-chunk start (the last sub-chunk needs no advance)
    {
        const int tend = t0 + (nsub - 1) * S;           // steps [t0, tend) are advanced
        T ub[TT], db[TT], un[TT], dn[TT];
        auto load_tile = [&](int tb, T (&uu)[TT], T (&dv)[TT]) {
#pragma unroll
            for (int s2 = 0; s2 < TT; ++s2) {
                const size_t off = ((size_t)b * p.L + min(tb + s2, t1 - 1)) * p.ED + e;
                uu[s2] = u[off]; dv[s2] = dl[off];
            }
        };
        if (tend > t0) { load_tile(t0, ub, db); tB.park(0, tB.fetch(row0)); }
        int cur = 0;
        for (int tb = t0; tb < tend; tb += TT, cur ^= 1) {
            const bool more = tb + TT < tend;
            float4 fb;
            if (more) { load_tile(tb + TT, un, dn); fb = tB.fetch((int64_t)b * p.L + tb + TT); }
#pragma unroll
            for (int s2 = 0; s2 < TT; ++s2) {
                const int t = tb + s2;
                if (t < tend) {
                    if ((s2 % S) == 0) {
                        const int j = (t - t0) / S;
#pragma unroll
                        for (int k = 0; k < NP; ++k) { ck[(j * N + 2 * k) * 64 + lane] = h[k].x; ck[(j * N + 2 * k + 1) * 64 + lane] = h[k].y; }
                    }
                    float dt = IO<T>::ld(&db[s2]) + bias;
                    if (p.softplus) dt = softplusf_(dt);
                    const float dtu = dt * IO<T>::ld(&ub[s2]);
                    float Bv[N];
                    tB.row(cur, s2, Bv);
#pragma unroll
                    for (int k = 0; k < NP; ++k) h[k] = exp2v(A2[k] * dt) * h[k] + f2{Bv[2 * k], Bv[2 * k + 1]} * dtu;
                }
            }
            if (more) tB.park(cur ^ 1, fb);
#pragma unroll
            for (int s2 = 0; s2 < TT; ++s2) { ub[s2] = un[s2]; db[s2] = dn[s2]; }
        }
#pragma unroll
        for (int k = 0; k < NP; ++k) { ck[((nsub - 1) * N + 2 * k) * 64 + lane] = h[k].x; ck[((nsub - 1) * N + 2 * k + 1) * 64 + lane] = h[k].y; }
    }

    f2 q[NP], dAacc[NP];
    if (p.nchunks > 1) {
#pragma unroll
        for (int k = 0; k < NP; ++k) q[k] = f2{p.qstate[sbase + (size_t)(2 * k) * p.ED], p.qstate[sbase + (size_t)(2 * k + 1) * p.ED]};
    } else {
#pragma unroll
        for (int k = 0; k < NP; ++k) q[k] = f2{0.f, 0.f};
    }
#pragma unroll
    for (int k = 0; k < NP; ++k) dAacc[k] = f2{0.f, 0.f};
    float dDacc = 0.f, dbacc = 0.f;

    // ---- sweep 2
    T uc[S], dc[S], zc[S], gc[S], un2[S], dn2[S], zn2[S], gn2[S];
    auto load_sub = [&](int j, T (&uu)[S], T (&dv)[S], T (&zz)[S], T (&gg)[S]) {
#pragma unroll
        for (int s2 = 0; s2 < S; ++s2) {
            const size_t off = ((size_t)b * p.L + min(t0 + j * S + s2, t1 - 1)) * p.ED + e;
            uu[s2] = u[off]; dv[s2] = dl[off]; gg[s2] = dy[off];
            if (has_z) zz[s2] = z[off];
        }
    };
    load_sub(nsub - 1, uc, dc, zc, gc);
    constexpr int SPT = TT / S;                       // sub-chunks per row tile
    {
        const int k = (nsub - 1) / SPT;
        tB.park(k & 1, tB.fetch(row0 + (int64_t)k * TT));
        tC.park(k & 1, tC.fetch(row0 + (int64_t)k * TT));
    }
    float4 fb2, fc2;
    for (int j = nsub - 1; j >= 0; --j) {
        if (j > 0) load_sub(j - 1, un2, dn2, zn2, gn2);
        const int k = j / SPT, r0 = (j - k * SPT) * S, tbuf = k & 1;
        if (k > 0 && (j == nsub - 1 || j - k * SPT == SPT - 1)) {      // entering tile k: its predecessor goes out now
            fb2 = tB.fetch(row0 + (int64_t)(k - 1) * TT);
            fc2 = tC.fetch(row0 + (int64_t)(k - 1) * TT);
        }
        const int ts = t0 + j * S;
        f2 hs[S][NP];
        float dts[S], us[S], sg[S];
        f2 h0[NP];
#pragma unroll
        for (int k = 0; k < NP; ++k) h0[k] = f2{ck[(j * N + 2 * k) * 64 + lane], ck[(j * N + 2 * k + 1) * 64 + lane]};
#pragma unroll
        for (int s2 = 0; s2 < S; ++s2) {
            const float draw = IO<T>::ld(&dc[s2]) + bias;
            float dt = draw, sgm = 1.f;
            if (p.softplus) { dt = softplusf_(draw); sgm = draw > 20.f ? 1.f : sigmoidf_(draw); }
            dts[s2] = dt; sg[s2] = sgm;
            us[s2] = IO<T>::ld(&uc[s2]);
            const float dtu = dt * us[s2];
            float Bv[N];
            tB.row(tbuf, r0 + s2, Bv);
#pragma unroll
            for (int k = 0; k < NP; ++k) {
                const f2 hp = (s2 == 0) ? h0[k] : hs[s2 - 1][k];
                hs[s2][k] = exp2v(A2[k] * dt) * hp + f2{Bv[2 * k], Bv[2 * k + 1]} * dtu;
            }
        }
        float dBrow[S], dCrow[S];
#pragma unroll
        for (int s2 = S - 1; s2 >= 0; --s2) {
            const int t = ts + s2;
            dBrow[s2] = 0.f; dCrow[s2] = 0.f;
            if (t < t1) {   // wave-uniform
                const size_t row = (size_t)b * p.L + t;
                const size_t off = row * p.ED + e;
                float Bv[N], Cv[N];
                tB.row(tbuf, r0 + s2, Bv);
                tC.row(tbuf, r0 + s2, Cv);
                const float dyv = IO<T>::ld(&gc[s2]);
                float g = dyv;
                if (has_z) {
                    const float zv = IO<T>::ld(&zc[s2]);
                    const float sz = sigmoidf_(zv);
                    f2 ys2 = f2{Dv * us[s2], 0.f};
#pragma unroll
                    for (int k = 0; k < NP; ++k) ys2 += hs[s2][k] * f2{Cv[2 * k], Cv[2 * k + 1]};
                    const float yss = ys2.x + ys2.y;
                    // d/dz [y*z*sigmoid(z)] = y*sigmoid(z)*(1 + z*(1-sigmoid(z)))
                    IO<T>::st(dz + off, dyv * yss * sz * (1.f + zv * (1.f - sz)));
                    g = dyv * zv * sz;
                }
                float dBv[N], dCv[N];
                const float dtu = dts[s2] * us[s2];
                f2 ddt2 = f2{0.f, 0.f}, dub2 = f2{0.f, 0.f};
#pragma unroll
                for (int k = 0; k < NP; ++k) {
                    const f2 a = exp2v(A2[k] * dts[s2]);
                    const f2 dh = f2{Cv[2 * k], Cv[2 * k + 1]} * g + q[k];
                    const f2 hp = (s2 == 0) ? h0[k] : hs[s2 - 1][k];
                    const f2 da = dh * hp * a;                    // dL/d(dt*A) for this (t, n pair)
                    dAacc[k] += da * dts[s2];
                    ddt2 += da * A2[k];                           // A2 = A*log2e, rescaled below
                    dub2 += dh * f2{Bv[2 * k], Bv[2 * k + 1]};
                    const f2 db2 = dh * dtu, dc2 = hs[s2][k] * g;
                    dBv[2 * k] = db2.x; dBv[2 * k + 1] = db2.y;
                    dCv[2 * k] = dc2.x; dCv[2 * k + 1] = dc2.y;
                    q[k] = a * dh;
                }
                const float dub = dub2.x + dub2.y;
                float ddt = ddt2.x + ddt2.y;
                ddt = ddt * GFE_LN2 + dub * us[s2];
                const float ddraw = ddt * sg[s2];
                IO<T>::st(dd + off, ddraw);
                IO<T>::st(du + off, fmaf(dub, dts[s2], Dv * g));
                dDacc = fmaf(g, us[s2], dDacc);
                dbacc += ddraw;
                if constexpr (N == 16) {
                    dBrow[s2] = wave_reduce_bc16(dBv, dCv);       // both vectors in one VALU-only butterfly (dCrow unused)
                } else {
                    dBrow[s2] = wave_reduce_vec<N>(dBv);
                    dCrow[s2] = wave_reduce_vec<N>(dCv);
                }
            }
        }
        if constexpr (N == 16) {
            // lane l holds, for every step s, the total of vector (l >> 5) for n = (l >> 1) & 15: two steps per 64-lane atomic
#pragma unroll
            for (int s0 = 0; s0 < S; s0 += 2) {
                const int sidx = s0 + (lane & 1);
                float val = dBrow[s0];
                if (s0 + 1 < S && (lane & 1)) val = dBrow[s0 + 1];
                const int t = ts + sidx;
                if (sidx < S && t < t1) {
                    float* base = (lane >> 5) ? p.dCws : p.dBws;
                    atomicAdd(base + ((size_t)b * p.L + t) * N + ((lane >> 1) & 15), val);
                }
            }
        } else {
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
        if (k > 0 && j == k * SPT) { tB.park(tbuf ^ 1, fb2); tC.park(tbuf ^ 1, fc2); }   // leaving tile k
#pragma unroll
        for (int s2 = 0; s2 < S; ++s2) { uc[s2] = un2[s2]; dc[s2] = dn2[s2]; zc[s2] = zn2[s2]; gc[s2] = gn2[s2]; }
    }
#pragma unroll
    for (int k = 0; k < NP; ++k) { atomicAdd(p.dAws + (size_t)(2 * k) * p.ED + e, dAacc[k].x); atomicAdd(p.dAws + (size_t)(2 * k + 1) * p.ED + e, dAacc[k].y); }
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
    const size_t lds = 4 * 4 * (8 * N / 4) * sizeof(float4);         // 4 waves x (B,C) x 2 buffers x TT*N/4 float4 (TT = 8)
    if (p.nchunks > 1) {
        hipLaunchKernelGGL((sscan_fwd_kernel<T, N, true>), grid, blk, lds, st, p);
        hipLaunchKernelGGL((sscan_carry_kernel<N, false>), dim3((unsigned)ceil_div((int64_t)N * p.ED, 256), p.B), blk, 0, st,
                           p.hstate, p.sdelta, p.A, p.ED, p.nchunks);
    }
    hipLaunchKernelGGL((sscan_fwd_kernel<T, N, false>), grid, blk, lds, st, p);
    return gfe_launch_status();
}

template <typename T, int N>
int sscan_bwd_launch(const SScanBwdParams& p, hipStream_t st) {
    constexpr int S = 4;
    if (p.nchunks > 1) {
        const dim3 blk(256), grid((unsigned)ceil_div(p.ED, 256), p.nchunks, p.B);
        hipLaunchKernelGGL((sscan_bwd_state_kernel<T, N>), grid, blk, 4 * 2 * (8 * N / 4) * sizeof(float4), st, p);
        hipLaunchKernelGGL((sscan_carry_kernel<N, true>), dim3((unsigned)ceil_div((int64_t)N * p.ED, 256), p.B), blk, 0, st,
                           p.qstate, p.sdelta, p.A, p.ED, p.nchunks);
    }
    const int nsub = (p.T + S - 1) / S;
    const size_t lds = (size_t)nsub * N * 64 * sizeof(float) + 4 * (16 * N / 4) * sizeof(float4);   // checkpoints + B/C row tiles
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

int gfe_selective_scan_fwd(const void* u, const void* delta, const float* A, const float* Bm, const float* Cm,
                           const float* D, const void* z, const float* delta_bias, void* y,
                           float* hstate, float* sdelta,
                           int64_t B, int64_t L, int64_t ED, int64_t N, int T, int delta_softplus,
                           int dtype, void* stream) {
    GFE_REQUIRE(u && delta && A && Bm && Cm && y, GFE_ERR_NULL);
    GFE_REQUIRE(B > 0 && L > 0 && ED > 0 && T > 0 && ED % 64 == 0, GFE_ERR_SHAPE);
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

int gfe_selective_scan_bwd(const void* u, const void* delta, const float* A, const float* Bm, const float* Cm,
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
