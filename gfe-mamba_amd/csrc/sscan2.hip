// Fused selective scan (Mamba S6 recurrence), N = 16 states -- single-pass formulation for gfx950.
//
// Replaces, like sscan.hip (which stays for N = 4 / 8), the reference's
//   cross_atten/mamba.py:265-286  MambaBlock.selective_scan  (deltaA/deltaB/BX -> pscan -> hs@C + D*x)
//   cross_atten/mamba.py:243-259  softplus(delta + dt_proj.bias) and the y*silu(z) gate of the `selective_scan_fn` plug-in
//   cross_atten/pscan.py:151-224  PScan.forward / backward (the recurrence and its adjoint)
//
// Why a second formulation (instruction prices from tools/probes/valu_rates.hip): the scan is VALU-issue bound --
// one v_exp_f32 (8 cycles) and four f32 operations per state-step -- so the only lever is the number of instructions issued per
// state-step.  sscan.hip keeps a channel's 16 states in one lane; filling the chip then needs chunks along L and TWO passes over
// every chunk (local end state, then the real pass), i.e. two exp per state-step.  Here a lane owns one channel x one PAIR of
// states: B*ED*8 lanes fill 1024 SIMDs from B = 8 up without cutting L, every state-step is computed once, the pair keeps the
// arithmetic in v_pk_* form, and a segment's states fit in registers for the backward (no third exp, no LDS checkpoints).
// L is cut into chunks (two passes + carry) only when B*ED/8 waves cannot fill the chip (B = 1).
//
// Work split: block = 4 waves = 32 channels of one (batch, chunk); wave = 8 channels x 8 state pairs:
//     lane bits {2,3,4} = pair p   (the sum over states = over p runs on DPP row operations + v_permlane16_swap, no selects)
//     lane bits {0,1,5} = channel c within the wave
// u / delta / z rows reach the block as 64-B (bf16) or 128-B (f32) row segments, are turned ONCE per (t, channel) into
// dt = softplus(delta + bias), dt*u, D*u and silu(z) and parked in LDS (double-buffered 32-step tiles, one barrier per tile);
// B / C rows are parked as {B[2p], B[2p+1], C[2p], C[2p+1]} so that one ds_read_b128 feeds a step.  y goes back through LDS so
// that global stores are whole row segments.
#include "common.h"

namespace {

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

constexpr int TT = 32;       // steps per LDS tile == checkpoint interval: the backward recomputes one tile at a time with its states in registers
constexpr int CB = 32;       // channels per block (4 waves x 8)
constexpr int EPS = 36;      // row stride (floats) of the [step][channel] arrays: banks 4*p + c are distinct for the butterfly's owners
constexpr int SEG = TT;

// [channel][step] arrays (dt, dt*u): a lane reads 4 consecutive steps of its channel per ds_read_b128.  The 4-step groups of a row are
// XOR-swizzled by the channel so that both the staging writes (32 lanes = 8 channel quads x 4 steps) and the reads (4 rows per
// 16-lane group) are bank-conflict free without padding.
__device__ __forceinline__ int dts_index(int ch, int t) { return ch * TT + ((((t >> 2) ^ (ch >> 2) ^ ((ch >> 1) & 1)) & 7) << 2) + (t & 3); }

struct S2Fwd {
    const void* u; const void* delta; const void* z; const void* Bm; const void* Cm;      // Bm / Cm: f32 or bf16 rows (bc_bf16)
    const float* A; const float* D; const float* dbias;
    void* y;
    void* yscan;        // (B, L, ED) or NULL: the scan's output BEFORE the gate, hs.C + D*u (the backward's dz = dy * silu'(z) * yscan needs it;
                        // saving it costs one more output row per step and spares the backward a sum over states per step)
    float* hstate;      // (B, nchunks, ED, 16): the state pass writes every chunk's local end state, the full pass folds its predecessors' (chunk_carry)
    float* sdelta;      // (B, nchunks, ED): sum of dt over the chunk (the chunk's decay is exp(A * sum dt): no products)
    float* ckpt;        // (B, nseg, ED, 16) state at the START of every 32-step segment, or NULL (no backward wanted)
    int B, L, ED, T, nchunks, softplus, nseg, bc_bf16;
    int ld_z, ld_bc;    // row strides (elements) of z and of the B / C rows: the fused Mamba block reads them in place out of the in_proj /
                        // x_proj outputs (z = xz[:, ED:], B = dBC[:, R:R+16], C = dBC[:, R+16:]) instead of from contiguous copies
    int a_log;          // A holds A_log (the parameter, mamba.py:160-162): the kernel uses A = -exp(A_log) (mamba.py:232)
};

// Diagnostic build only (-DGFE_S2_STAMPS, tools/scan_stamps.py): wave 0 of block 0 accumulates s_memtime deltas per phase.
#ifdef GFE_S2_STAMPS
__device__ unsigned long long g_s2_stamps[48];
#define S2_STAMP_DECL unsigned long long st_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long st_last = __builtin_amdgcn_s_memtime();
#define S2_STAMP(i) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); st_acc[i] += now_ - st_last; st_last = now_; }
#define S2_STAMP_FLUSH(base) if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) { for (int i_ = 0; i_ < 12; ++i_) g_s2_stamps[base + i_] = st_acc[i_]; }
#else
#define S2_STAMP_DECL
#define S2_STAMP(i)
#define S2_STAMP_FLUSH(base)
#endif

// Workgroup barrier that waits for this wave's LDS traffic only.  __syncthreads() also drains vmcnt: behind a block's global stores and
// (backward) float atomics -- which stay counted for 600-3000 cycles -- that stalls every wave once per tile for nothing: no global
// data is exchanged between the waves of a block here, only LDS.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Staging math, branch-free (the generic helpers in common.h compile to nested exec-masked branches around v_log / the series, which at
// one wave per SIMD cost more than both paths together).  softplus(x) = max(x, 0) + log1p(e), e = exp(-|x|): series below 2^-7 (1 + e
// would round e away; the reference's dt_proj.bias puts softplus outputs down to 1e-4), v_log above; for x > 20 the result equals x
// after rounding, which is torch's threshold rule.  sig (optional) = d softplus / dx = sigmoid(x) from the same e.
__device__ __forceinline__ float softplus_nb(float x, float* sig = nullptr) {
    const float e = fast_exp2(-fabsf(x) * GFE_LOG2E);
    const float w1 = 1.0f + e;
    const float ser = e * fmaf(e, fmaf(e, 0.33333333f, -0.5f), 1.0f);
    const float lg = __builtin_amdgcn_logf(w1) * GFE_LN2;
    const float l = e < 0.0078125f ? ser : lg;
    if (sig) { const float r = fast_rcp(w1); *sig = x >= 0.f ? r : e * r; }
    return __builtin_amdgcn_fmed3f(x, 0.0f, __builtin_huge_valf()) + l;
}

// Sum over the 8 pair-lanes (lane bits 2,3,4) of 8 per-lane values v[0..7]; the lane with pair index p returns the total of v[p].
// Halving butterfly: bit 4 by v_permlane16_swap (rows of 16 lanes trade registers), bits 3 and 2 by pairs of bank-masked
// v_add_f32_dpp (row_ror:8; row_shl:4 / row_shr:4): each pair writes the two halves of one result register, so a level costs two
// instructions per result and needs neither v_cndmask nor copies -- 16 VALU instructions per 8 values.
// (Inline asm because the masked v_add_dpp has no builtin; the leading s_nop 1 is the VALU-write -> DPP-read hazard, which hipcc does
// not pad inside or in front of an asm statement.)
__device__ __forceinline__ float reduce_pairs8(const float (&v)[8]) {
    float w[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const auto x = __builtin_amdgcn_permlane16_swap(__float_as_uint(v[j]), __float_as_uint(v[j + 4]), false, false);
        w[j] = __uint_as_float(x[0]) + __uint_as_float(x[1]);          // rows with bit4 = 0: total of v[j]; bit4 = 1: total of v[j+4]
    }
    float x0, x1, r;
    // lanes 0-7 of a row (bit3 = 0): w[j] + w[j] of lane^8;  lanes 8-15: w[j+2] + w[j+2] of lane^8
    asm volatile("s_nop 1\n\t"
                 "v_add_f32_dpp %0, %2, %2 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
                 "v_add_f32_dpp %1, %3, %3 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
                 "v_add_f32_dpp %0, %4, %4 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
                 "v_add_f32_dpp %1, %5, %5 row_ror:8 row_mask:0xf bank_mask:0xc"
                 : "=&v"(x0), "=&v"(x1) : "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]));
    // banks 0,2 (bit2 = 0): x0 + x0 of lane+4;  banks 1,3 (bit2 = 1): x1 + x1 of lane-4
    asm volatile("s_nop 1\n\t"
                 "v_add_f32_dpp %0, %1, %1 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
                 "v_add_f32_dpp %0, %2, %2 row_shr:4 row_mask:0xf bank_mask:0xa"
                 : "=&v"(r) : "v"(x0), "v"(x1));
    return r;
}

template <typename T> struct Vec4;
template <> struct Vec4<float> {
    typedef f4 type;
    static __device__ __forceinline__ type pack(const f4& v) { return v; }
    static __device__ __forceinline__ void unpack(const type& v, float (&o)[4]) { o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w; }
    static __device__ __forceinline__ type zero() { return type{0.f, 0.f, 0.f, 0.f}; }
};
template <> struct Vec4<bf16_t> {
    typedef uint2 type;
    static __device__ __forceinline__ type pack(const f4& v) {
        return make_uint2((uint32_t)f32_to_bf16(v.x) | ((uint32_t)f32_to_bf16(v.y) << 16), (uint32_t)f32_to_bf16(v.z) | ((uint32_t)f32_to_bf16(v.w) << 16));
    }
    static __device__ __forceinline__ void unpack(const type& v, float (&o)[4]) {
        o[0] = bf16lo_to_f32(v.x); o[1] = bf16hi_to_f32(v.x); o[2] = bf16lo_to_f32(v.y); o[3] = bf16hi_to_f32(v.y);
    }
    static __device__ __forceinline__ type zero() { return make_uint2(0u, 0u); }
};

// LDS image of one 32-step tile
struct Tile {
    float dt[CB * TT];           // [channel][step]  softplus(delta + bias), 0 past the chunk end  (dts_index)
    float dtu[CB * TT];          // [channel][step]  dt * u
    float epu[TT * EPS];         // [step][channel]  D * u
    float epg[TT * EPS];         // [step][channel]  silu(z) (1 without a gate)
    f4 bc[TT * 8];               // [step][pair]     {B[2p], B[2p+1], C[2p], C[2p+1]}
};

// One block = 32 channels x one chunk.  STATE_ONLY (K1 of the chunked plan): end state from h = 0 and sum(dt), no output.
// A block's 32 channels are a 64-byte segment of every (b, t) row of the bf16 tensors: HALF a 128-byte line, the other half belongs to the
// neighbouring channel group.  Workgroups go to the 8 XCDs round-robin, so with the identity mapping the two halves are fetched by two
// different L2s -- every input line crosses the HBM interface twice (FETCH_SIZE 420 MB for 201 MB of forward inputs, profiles/r02).
// Groups 2k and 2k + 1 are therefore given to blocks x and x + 8 (same XCD, dispatched together): the second half is an L2 hit.
__device__ __forceinline__ int xcd_paired_group(int x, int G) {
    if (G & 15) return x;
    const int xcd = x & 7, j = x >> 3;
    return 2 * (xcd + 8 * (j >> 1)) + (j & 1);
}

// The chunked plan's carry, folded into the full pass (round 3; it was a kernel of its own between the two passes: 5 us + two kernel
// boundaries each way at B = 1): the state pass left every chunk's LOCAL end state (from h = 0) and its sum of dt; the state at the
// start of chunk c is the fold of its predecessors' pairs, H <- exp(A * sum dt_j) * H + local_j, j = 0 .. c-1 (REVERSE: the adjoint
// carry, j = nchunks-1 .. c+1) -- at most nchunks - 1 independent 8-byte loads and as many fma per lane.
template <bool REVERSE>
__device__ __forceinline__ f2 chunk_carry(const float* __restrict__ state, const float* __restrict__ sdelta, f2 A2, int b, int c, int e, int pr,
                                          int nchunks, int ED) {
    f2 H = f2{0.f, 0.f};
    const int n = REVERSE ? nchunks - 1 - c : c;
    constexpr int G = 8;
    for (int k0 = 0; k0 < n; k0 += G) {
        f2 loc[G];
        float sdv[G];
#pragma unroll
        for (int j = 0; j < G; ++j) {
            const int k = min(k0 + j, n - 1);
            const int cj = REVERSE ? nchunks - 1 - k : k;
            loc[j] = *reinterpret_cast<const f2*>(state + (((size_t)b * nchunks + cj) * ED + e) * 16 + 2 * pr);
            sdv[j] = sdelta[((size_t)b * nchunks + cj) * ED + e];
        }
#pragma unroll
        for (int j = 0; j < G; ++j) {
            if (k0 + j < n) {
                const f2 x = A2 * sdv[j];
                H = f2{fast_exp2(x.x), fast_exp2(x.y)} * H + loc[j];
            }
        }
    }
    return H;
}

// Wave-specialised block (round 3): 8 waves = 4 SCAN waves (0-3: the recurrence, one wave per SIMD as before) + 4 STAGING waves (4-7,
// the SIMD partners of waves 0-3): fetch the next tile's rows, do the per-(t, channel) math (softplus, silu, products) once, park it in
// the other LDS buffer, and write the previous tile's y rows out.  The scan waves' instruction stream loses a third of its issue slots'
// worth of work (staging 30 + stores 12 of 128 cycles per step) to a partner that fills the slots the dependent h chain leaves idle,
// and plain VALU instructions cost 2.7 instead of 6.5 cycles with two waves on a SIMD (profiles/r02/valu_rates.txt).  One barrier per tile.
template <typename T, bool STATE_ONLY>
__global__ __launch_bounds__(512, 4) void sscan2_fwd_kernel(const S2Fwd p) {      // <= 128 registers: two blocks per CU when the grid has them (chunked plans, B >= 16)
    typedef typename Vec4<T>::type V4;
    __shared__ __attribute__((aligned(16))) Tile tiles[2];
    __shared__ __attribute__((aligned(16))) float ytile[2][TT * EPS];           // hs.C + D*u of the tile (f32: gated and rounded once, on the way out)
    const bool staging = __builtin_amdgcn_readfirstlane(threadIdx.x) >= 256;      // wave-uniform role
    const int tid = threadIdx.x & 255, lane = tid & 63, w = tid >> 6;
    const int e0 = xcd_paired_group(blockIdx.x, gridDim.x) * CB, c = blockIdx.y, b = blockIdx.z;
    const int t0 = c * p.T, t1 = min(p.L, t0 + p.T);
    const int nrows = t1 - t0;
    // staging role of a thread: row sr of the tile, channels 4*sc .. 4*sc+3 of the block
    const int sr = tid >> 3, sc = tid & 7;
    const size_t rowbase = ((size_t)b * p.L + t0) * p.ED + e0 + 4 * sc;

    if (staging) {
        const T* __restrict__ u = (const T*)p.u;
        const T* __restrict__ dl = (const T*)p.delta;
        const T* __restrict__ z = (const T*)p.z;
        T* __restrict__ y = (T*)p.y;
        T* __restrict__ ysc = (T*)p.yscan;
        const bool has_z = !STATE_ONLY && z != nullptr;
        float sbias[4], sD[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            sbias[k] = p.dbias ? p.dbias[e0 + 4 * sc + k] : 0.f;
            sD[k] = (!STATE_ONLY && p.D) ? p.D[e0 + 4 * sc + k] : 0.f;
        }
        // B / C role: threads 0-127 fetch B, 128-255 fetch C: row (tid & 127) >> 2 of the tile, floats 4q .. 4q+3
        const int br = (tid & 127) >> 2, bq = tid & 3;
        const void* bcsrc = (tid < 128) ? p.Bm : p.Cm;
        V4 ru = Vec4<T>::zero(), rd = Vec4<T>::zero(), rz = Vec4<T>::zero();
        f4 rbc = f4{0.f, 0.f, 0.f, 0.f};
        // 64-bit bases once; per tile only a 32-bit row offset (rows past the chunk end: clamped here, masked in park)
        const T* __restrict__ pu = u + rowbase;
        const T* __restrict__ pd = dl + rowbase;
        const T* __restrict__ pz = has_z ? z + ((size_t)b * p.L + t0) * p.ld_z + e0 + 4 * sc : nullptr;
        const size_t bcbase = ((size_t)b * p.L + t0) * p.ld_bc + 4 * bq;
        const float* __restrict__ pbc = (const float*)bcsrc + bcbase;
        const bf16_t* __restrict__ pbc16 = (const bf16_t*)bcsrc + bcbase;
        auto fetch = [&](int tb) {                                    // global -> registers, tile starting at step tb
            const int row = min(tb - t0 + sr, nrows - 1);
            const int off = row * p.ED;
            ru = *reinterpret_cast<const V4*>(pu + off);
            rd = *reinterpret_cast<const V4*>(pd + off);
            if (has_z) rz = *reinterpret_cast<const V4*>(pz + row * p.ld_z);
            if (!STATE_ONLY || tid < 128) {
                const int boff = min(tb - t0 + br, nrows - 1) * p.ld_bc;
                if (p.bc_bf16) { const uint2 r = *reinterpret_cast<const uint2*>(pbc16 + boff); rbc = f4{__uint_as_float(r.x), __uint_as_float(r.y), 0.f, 0.f}; }
                else rbc = *reinterpret_cast<const f4*>(pbc + boff);
            }
        };
        auto bc_rows = [&](int tb) -> f4 {                            // the fetched B / C quarter-row as f32 (zero past the end)
            f4 v = rbc;
            if (p.bc_bf16) {
                const uint32_t lo = __float_as_uint(rbc.x), hi = __float_as_uint(rbc.y);
                v = f4{bf16lo_to_f32(lo), bf16hi_to_f32(lo), bf16lo_to_f32(hi), bf16hi_to_f32(hi)};
            }
            return (tb + br < t1) ? v : f4{0.f, 0.f, 0.f, 0.f};
        };
        auto park = [&](Tile& tl, int tb) {                           // registers -> LDS (the per-(t, channel) math happens once, here)
            float fu[4], fd[4], fz[4];
            Vec4<T>::unpack(ru, fu); Vec4<T>::unpack(rd, fd);
            if (has_z) Vec4<T>::unpack(rz, fz);
            const bool valid = tb + sr < t1;
            f4 vu, vg;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float raw = fd[k] + sbias[k];
                float dt = p.softplus ? softplus_nb(raw) : raw;
                if (!valid) dt = 0.f;                                 // a = exp2(0) = 1, dt*u = 0: steps past the end leave the state alone
                tl.dt[dts_index(4 * sc + k, sr)] = dt;
                tl.dtu[dts_index(4 * sc + k, sr)] = dt * fu[k];
                vu[k] = sD[k] * fu[k];
                vg[k] = has_z ? siluf_(fz[k]) : 1.f;
            }
            if (!STATE_ONLY) {
                *reinterpret_cast<f4*>(&tl.epu[sr * EPS + 4 * sc]) = vu;
                *reinterpret_cast<f4*>(&tl.epg[sr * EPS + 4 * sc]) = vg;
            }
            float* bcp = reinterpret_cast<float*>(&tl.bc[br * 8 + 2 * bq]) + (tid < 128 ? 0 : 2);
            if (!STATE_ONLY || tid < 128) {
                const f4 v = bc_rows(tb);
                *reinterpret_cast<f2*>(bcp) = f2{v.x, v.y};
                *reinterpret_cast<f2*>(bcp + 4) = f2{v.z, v.w};
            }
        };

        // a finished tile's outputs: whole row segments, 4 channels per lane; the gate silu(z) of that tile is still in its staging buffer
        auto rows_out = [&](int buf, int r) {
            const f4 yb = *reinterpret_cast<const f4*>(&ytile[buf][sr * EPS + 4 * sc]);
            const f4 g = *reinterpret_cast<const f4*>(&tiles[buf].epg[sr * EPS + 4 * sc]);
            if (r < nrows) {
                const size_t off = rowbase + (size_t)r * p.ED;
                *reinterpret_cast<V4*>(y + off) = Vec4<T>::pack(yb * g);
                if (ysc) *reinterpret_cast<V4*>(ysc + off) = Vec4<T>::pack(yb);
            }
        };
        fetch(t0);
        park(tiles[0], t0);
        if (t0 + TT < t1) fetch(t0 + TT);                             // the rows of tile k+1 are in flight while tile k-1's outputs go out
        lds_barrier();
        int cur = 0;
        for (int tb = t0; tb < t1; tb += TT, cur ^= 1) {
            const bool more = tb + TT < t1;
            if (!STATE_ONLY && tb > t0) rows_out(cur ^ 1, tb - TT - t0 + sr);   // (before park() reuses that buffer)
            if (more) {
                park(tiles[cur ^ 1], tb + TT);
                if (tb + 2 * TT < t1) fetch(tb + 2 * TT);
            }
            lds_barrier();
        }
        if (!STATE_ONLY) rows_out(cur ^ 1, ((t1 - t0 - 1) / TT) * TT + sr);     // the last tile
        return;
    }

    // ---- scan waves
    __builtin_amdgcn_s_setprio(2);                                   // the dependent chain wins every issue arbitration against its staging partner
    const int pr = (lane >> 2) & 7, cw = (lane & 3) | ((lane >> 5) << 2);
    const int cl = 8 * w + cw;                                   // channel within the block
    const int e = e0 + cl;
    f2 A2 = f2{p.A[(size_t)e * 16 + 2 * pr], p.A[(size_t)e * 16 + 2 * pr + 1]};
    if (p.a_log) A2 = f2{-fast_exp2(A2.x * GFE_LOG2E), -fast_exp2(A2.y * GFE_LOG2E)};       // A = -exp(A_log)
    A2 *= GFE_LOG2E;
    f2 h = f2{0.f, 0.f};
    const size_t sbase = (((size_t)b * p.nchunks + c) * p.ED + e) * 16 + 2 * pr;
    if (!STATE_ONLY && p.nchunks > 1) h = chunk_carry<false>(p.hstate, p.sdelta, A2, b, c, e, pr, p.nchunks, p.ED);
    float sd = 0.f;
    lds_barrier();
    int cur = 0;
    S2_STAMP_DECL
    for (int tb = t0; tb < t1; tb += TT, cur ^= 1) {
        S2_STAMP(0)
        if (!STATE_ONLY && p.ckpt)                                // every tile starts a segment (t0 is a multiple of SEG)
            *reinterpret_cast<f2*>(p.ckpt + ((((size_t)b * p.nseg + tb / SEG) * p.ED + e) * 16 + 2 * pr)) = h;
        const Tile& tl = tiles[cur];
        // Nothing hides an LDS round trip of the dependent chain but the wave's own instruction stream, so reads are issued half a group
        // (4 steps) ahead of their use: a group's second-half B / C rows while its 16 decays are exponentiated, the next group's dt / dt*u
        // quads and first-half rows between its two halves (two register sets for those, pinned with sched_barrier; <= 128 registers).
        struct Grp { f4 dt4[2], du4[2], bc[4]; float epu; };
        auto load_a = [&](Grp& G, int g) {
#pragma unroll
            for (int j4 = 0; j4 < 2; ++j4) {
                G.dt4[j4] = *reinterpret_cast<const f4*>(&tl.dt[dts_index(cl, 8 * g + 4 * j4)]);
                G.du4[j4] = *reinterpret_cast<const f4*>(&tl.dtu[dts_index(cl, 8 * g + 4 * j4)]);
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) G.bc[s] = tl.bc[(8 * g + s) * 8 + pr];
            if (!STATE_ONLY) G.epu = tl.epu[(8 * g + pr) * EPS + cl];
        };
        f4 bcb[4];
        auto load_b = [&](int g) {
#pragma unroll
            for (int s = 0; s < 4; ++s) bcb[s] = tl.bc[(8 * g + 4 + s) * 8 + pr];
        };
        float yv[8];
        f2 a[8];
        auto exps = [&](const Grp& G) {
#pragma unroll
            for (int s = 0; s < 8; ++s) {                          // the decays do not depend on h: all 16 exp first, so that none of the
                const f2 x = A2 * G.dt4[s >> 2][s & 3];            // transcendental results is wanted right behind its instruction
                a[s] = f2{fast_exp2(x.x), fast_exp2(x.y)};
                if (STATE_ONLY) sd += G.dt4[s >> 2][s & 3];
            }
        };
        auto half = [&](const Grp& G, const f4 (&bc4)[4], int hf) {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const f4 bc = bc4[s];
                h = a[4 * hf + s] * h + f2{bc.x, bc.y} * G.du4[hf][s];
                if (!STATE_ONLY) yv[4 * hf + s] = fmaf(h.y, bc.w, h.x * bc.z);
            }
        };
        auto finish = [&](const Grp& G, int g) {
            if (!STATE_ONLY) {
                const float ys = reduce_pairs8(yv);                // this lane: step 8g + pr of channel cl
                ytile[cur][(8 * g + pr) * EPS + cl] = ys + G.epu;
            }
        };
        Grp ga, gb;
        S2_STAMP(1)
        load_a(ga, 0);
#define SB __builtin_amdgcn_sched_barrier(0)
#pragma unroll
        for (int g = 0; g < TT / 8; g += 2) {
            load_b(g); SB;
            exps(ga); half(ga, ga.bc, 0); SB;
            load_a(gb, g + 1); SB;
            half(ga, bcb, 1); finish(ga, g); SB;
            load_b(g + 1); SB;
            exps(gb); half(gb, gb.bc, 0); SB;
            if (g + 2 < TT / 8) load_a(ga, g + 2);
            SB;
            half(gb, bcb, 1); finish(gb, g + 1); SB;
        }
#undef SB
        S2_STAMP(2)
        lds_barrier();
        S2_STAMP(4)
    }
    S2_STAMP_FLUSH(STATE_ONLY ? 36 : 0)
    if (STATE_ONLY) {
        *reinterpret_cast<f2*>(p.hstate + sbase) = h;
        if (pr == 0) p.sdelta[((size_t)b * p.nchunks + c) * p.ED + e] = sd;
    }
}

template <typename T>
int sscan2_fwd_launch(const S2Fwd& p, hipStream_t st) {
    const dim3 blk(512), grid((unsigned)(p.ED / CB), p.nchunks, p.B);
    if (p.nchunks > 1) hipLaunchKernelGGL((sscan2_fwd_kernel<T, true>), grid, blk, 0, st, p);      // local end states + sum dt of every chunk
    hipLaunchKernelGGL((sscan2_fwd_kernel<T, false>), grid, blk, 0, st, p);
    return gfe_launch_status();
}


// ------------------------------------------------------------------------------------------------
// backward
// ------------------------------------------------------------------------------------------------
struct S2Bwd {
    const void* u; const void* delta; const void* z; const void* Bm; const void* Cm; const void* dy;
    const void* yscan;                  // (B, L, ED): hs.C + D*u as left by the forward (needed with a gate: dz = dy * silu'(z) * yscan)
    const float* A; const float* D; const float* dbias;
    void* du; void* ddelta; void* dz;
    float* dAws;                        // (ED, 16) f32, zeroed, accumulated atomically (once per lane and chunk)
    float* dBws; float* dCws;           // (B, L, 16) f32, zeroed, accumulated atomically (one 64-lane atomic per two steps and block)
    float* dDws; float* dbiasws;        // (ED) f32, zeroed
    const float* ckpt;                  // (B, nseg, ED, 16) segment-start states left by the forward
    float* qstate;                      // (B, nchunks, ED, 16) adjoint carry (chunked plan only)
    const float* sdelta;                // (B, nchunks, ED)
    int B, L, ED, T, nchunks, softplus, nseg, bc_bf16;
    int ld_z, ld_bc, ld_dbc;            // row strides of z / dz, of the B / C rows, and of the dB / dC accumulation rows (see S2Fwd)
    int a_log;                          // A holds A_log: A = -exp(A_log), and dA_ws receives the gradient w.r.t. A_log (= dA * A)
    // Deterministic accumulation (round 4): with these two workspaces nothing is added atomically.  Every block leaves its contributions as
    // plain stores -- part_vec[b * nchunks + c][dA (ED x 16) | dD (ED) | dbias (ED)], part_bc[b][channel group][t][16 dB | 16 dC] -- and
    // sscan2_fold_kernel sums them in a fixed order (samples / chunks in order for the vectors, channel groups in order for the rows).
    // NULL: f32 atomics onto zeroed buffers (large L x B: the dB / dC partial rows would be 128 B per (group, b, t), 134 MB at config 2, B = 8).
    float* part_vec; float* part_bc;
};

constexpr int RSL = TT * 32 + 8;   // slab stride: + 8 floats so that the four (channel & 3) slabs of one ds_write_b64 fall on disjoint banks
struct BStage {                  // what the staging waves park for one 32-step segment (double-buffered)
    float dt[CB * TT];           // [channel][step] (dts_index)  softplus(delta + bias), 0 past the end
    float dtu[CB * TT];          //                              dt * u
    float g[CB * TT];            //                              dL/dy_scan = dy * silu(z) (dy without a gate), 0 past the end
    float eu[TT * EPS];          // [step][channel]: what the lane that owns (step, channel) after the pair butterflies needs
    float edt[TT * EPS];
    float esg[TT * EPS];         //   d softplus / d raw = sigmoid(raw) (1 without softplus), 0 past the end
    float eg[TT * EPS];
    f4 bc[TT * 8];               // [step][pair] {B[2p], B[2p+1], C[2p], C[2p+1]}
};

// One block = 32 channels x one chunk, segments of 32 steps walked from the chunk's end to its start:
//   recompute the segment forward from its checkpoint, keeping a_t = exp(dt_t A) and h_t of all 32 steps in registers (a lane owns two
//   states: 128 registers), then run the adjoint over the same steps -- no exp beyond the recompute's.
//   Sums over states (d(dt*u), d dt) go through the pair butterflies (dz needs the forward's pre-gate output, which the forward saved:
//   it is finished by the staging waves, element-wise, on the way in), sums over channels (dB, dC) through
//   v_permlane32_swap + quad DPP adds into an LDS slab that the block folds over its waves and adds to memory two steps per atomic.
// Wave-specialised like the forward (round 3): waves 0-3 run the recurrence and its adjoint, waves 4-7 (their SIMD partners) fetch and
// stage the next segment into the other LDS buffer, write the previous segment's du / ddelta / dz rows out and fold + publish its dB / dC
// rows.  Two barriers per segment: A (segment start: staging buffer ready, previous segment's outputs complete) and B (between recompute
// and adjoint: the previous outputs have been drained, the adjoint may overwrite `red` / `otile`).
// STATE_ONLY (K1' of the chunked plan): only the local adjoint carry q of the chunk from q = 0 (one barrier per segment).
template <typename T, bool STATE_ONLY>
__global__ __launch_bounds__(512) void sscan2_bwd_kernel(const S2Bwd p) {
    typedef typename Vec4<T>::type V4;
    __shared__ __attribute__((aligned(16))) BStage stg[2];
    __shared__ __attribute__((aligned(16))) float red[STATE_ONLY ? 4 : 16 * RSL];   // [wave][channel & 3] slabs of [step][16 dB | 16 dC]
    __shared__ __attribute__((aligned(16))) T otile[2][TT * EPS];       // du, ddelta rows on their way out
    const bool staging = __builtin_amdgcn_readfirstlane(threadIdx.x) >= 256;      // wave-uniform role
    const int tid = threadIdx.x & 255, lane = tid & 63, w = tid >> 6;
    const int e0 = xcd_paired_group(blockIdx.x, gridDim.x) * CB, c = blockIdx.y, b = blockIdx.z;
    const int t0 = c * p.T, t1 = min(p.L, t0 + p.T);
    const int nrows = t1 - t0;
    const int nsegc = (t1 - t0 + TT - 1) / TT;
    const bool has_z = p.z != nullptr;
    const int sr = tid >> 3, sc = tid & 7;
    const size_t rowbase = ((size_t)b * p.L + t0) * p.ED + e0 + 4 * sc;      // 64-bit bases once; per segment only a 32-bit row offset

    if (staging) {
        const T* __restrict__ u = (const T*)p.u;
        const T* __restrict__ dl = (const T*)p.delta;
        const T* __restrict__ z = (const T*)p.z;
        const T* __restrict__ dy = (const T*)p.dy;
        const T* __restrict__ ysc = (const T*)p.yscan;
        float sbias[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) sbias[k] = p.dbias ? p.dbias[e0 + 4 * sc + k] : 0.f;
        const int br = (tid & 127) >> 2, bq = tid & 3;
        const void* bcsrc = (tid < 128) ? p.Bm : p.Cm;
        V4 ru = Vec4<T>::zero(), rd = Vec4<T>::zero(), rz = Vec4<T>::zero(), rg = Vec4<T>::zero(), ry = Vec4<T>::zero();
        f4 rbc = f4{0.f, 0.f, 0.f, 0.f};
        const T* __restrict__ pu = u + rowbase;
        const T* __restrict__ pd = dl + rowbase;
        const T* __restrict__ pg = dy + rowbase;
        const size_t zbase = ((size_t)b * p.L + t0) * p.ld_z + e0 + 4 * sc;
        const T* __restrict__ pz = has_z ? z + zbase : nullptr;
        const T* __restrict__ py = (has_z && !STATE_ONLY) ? ysc + rowbase : nullptr;
        const size_t bcbase = ((size_t)b * p.L + t0) * p.ld_bc + 4 * bq;
        const float* __restrict__ pbc = (const float*)bcsrc + bcbase;
        const bf16_t* __restrict__ pbc16 = (const bf16_t*)bcsrc + bcbase;
        auto fetch = [&](int tb) {                                    // rows past the end: clamped here, masked in park
            const int row = min(tb - t0 + sr, nrows - 1);
            const int off = row * p.ED;
            if (!STATE_ONLY) ru = *reinterpret_cast<const V4*>(pu + off);
            rd = *reinterpret_cast<const V4*>(pd + off);
            rg = *reinterpret_cast<const V4*>(pg + off);
            if (has_z) rz = *reinterpret_cast<const V4*>(pz + row * p.ld_z);
            if (has_z && !STATE_ONLY) ry = *reinterpret_cast<const V4*>(py + off);
            if (!STATE_ONLY || tid >= 128) {
                const int boff = min(tb - t0 + br, nrows - 1) * p.ld_bc;
                if (p.bc_bf16) { const uint2 r = *reinterpret_cast<const uint2*>(pbc16 + boff); rbc = f4{__uint_as_float(r.x), __uint_as_float(r.y), 0.f, 0.f}; }
                else rbc = *reinterpret_cast<const f4*>(pbc + boff);
            }
        };
        auto bc_rows = [&](int tb) -> f4 {
            f4 v = rbc;
            if (p.bc_bf16) {
                const uint32_t lo = __float_as_uint(rbc.x), hi = __float_as_uint(rbc.y);
                v = f4{bf16lo_to_f32(lo), bf16hi_to_f32(lo), bf16lo_to_f32(hi), bf16hi_to_f32(hi)};
            }
            return (tb + br < t1) ? v : f4{0.f, 0.f, 0.f, 0.f};
        };
        auto park = [&](BStage& tl, int tb) {
            float fu[4] = {0.f, 0.f, 0.f, 0.f}, fd[4], fz[4], fg[4], fy[4] = {0.f, 0.f, 0.f, 0.f};
            if (!STATE_ONLY) Vec4<T>::unpack(ru, fu);
            Vec4<T>::unpack(rd, fd); Vec4<T>::unpack(rg, fg);
            if (has_z) Vec4<T>::unpack(rz, fz);
            if (has_z && !STATE_ONLY) Vec4<T>::unpack(ry, fy);
            const bool valid = tb + sr < t1;
            f4 vu, vdt, vsg, vg, vgz;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float raw = fd[k] + sbias[k];
                float sg = 1.f;
                float dt = softplus_nb(raw, &sg);
                if (!p.softplus) { dt = raw; sg = 1.f; }
                float gg = fg[k], gz = 0.f;
                if (has_z) {
                    const float sz = sigmoidf_(fz[k]);
                    gz = fg[k] * sz * (1.f + fz[k] * (1.f - sz)) * fy[k];  // dz = dy * d/dz [z sigmoid(z)] * (hs.C + D*u)
                    gg = fg[k] * fz[k] * sz;
                }
                if (!valid) { dt = 0.f; sg = 0.f; gg = 0.f; gz = 0.f; }
                const int ix = dts_index(4 * sc + k, sr);
                tl.dt[ix] = dt;
                if (!STATE_ONLY) tl.dtu[ix] = dt * fu[k];
                tl.g[ix] = gg;
                vu[k] = fu[k]; vdt[k] = dt; vsg[k] = sg; vg[k] = gg; vgz[k] = gz;
            }
            if (!STATE_ONLY) {
                *reinterpret_cast<f4*>(&tl.eu[sr * EPS + 4 * sc]) = vu;
                *reinterpret_cast<f4*>(&tl.edt[sr * EPS + 4 * sc]) = vdt;
                *reinterpret_cast<f4*>(&tl.esg[sr * EPS + 4 * sc]) = vsg;
                *reinterpret_cast<f4*>(&tl.eg[sr * EPS + 4 * sc]) = vg;
                if (has_z && valid) *reinterpret_cast<V4*>((T*)p.dz + zbase + (size_t)(tb - t0 + sr) * p.ld_z) = Vec4<T>::pack(vgz);   // finished here
            }
            float* bcp = reinterpret_cast<float*>(&tl.bc[br * 8 + 2 * bq]) + (tid < 128 ? 0 : 2);
            if (!STATE_ONLY || tid >= 128) {
                const f4 v = bc_rows(tb);
                *reinterpret_cast<f2*>(bcp) = f2{v.x, v.y};
                *reinterpret_cast<f2*>(bcp + 4) = f2{v.z, v.w};
            }
        };
        // ---- rows out, and the block's dB / dC rows, of the segment that starts at step tb.  Wave w folds the 16 partial slabs for steps
        // 8w .. 8w+7 with ds_read_b128, writes the sums back as rows of 32 (its own rows: no barrier, only its own lgkmcnt) and adds them to
        // memory as contiguous 64-lane atomics -- two steps x (16 dB | 16 dC) per instruction, the access shape float atomics run at full rate on.
        auto drain = [&](int tb) {
            const int r = tb - t0 + sr;
            if (r < nrows) {
                const size_t off = rowbase + (size_t)r * p.ED;
                *reinterpret_cast<V4*>((T*)p.du + off) = *reinterpret_cast<const V4*>(&otile[0][sr * EPS + 4 * sc]);
                *reinterpret_cast<V4*>((T*)p.ddelta + off) = *reinterpret_cast<const V4*>(&otile[1][sr * EPS + 4 * sc]);
            }
            const int ts = 8 * w + (lane >> 3), jq = (lane & 7) * 4;
            f4 acc = *reinterpret_cast<const f4*>(&red[ts * 32 + jq]);
#pragma unroll
            for (int k = 1; k < 16; ++k) acc += *reinterpret_cast<const f4*>(&red[k * RSL + ts * 32 + jq]);
            *reinterpret_cast<f4*>(&red[ts * 32 + jq]) = acc;          // slab 0, this wave's rows only
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int v = lane + 64 * i, t2 = 8 * w + (v >> 5), j = v & 31;
                const float sum = red[t2 * 32 + j];
                if (tb + t2 < t1) {
                    if (p.part_bc) p.part_bc[(((size_t)b * gridDim.x + blockIdx.x) * p.L + tb + t2) * 32 + j] = sum;
                    else atomicAdd((j < 16 ? p.dBws : p.dCws) + ((size_t)b * p.L + tb + t2) * p.ld_dbc + (j & 15), sum);
                }
            }
        };

        fetch(t0 + (nsegc - 1) * TT);
        park(stg[0], t0 + (nsegc - 1) * TT);
        if (nsegc > 1) fetch(t0 + (nsegc - 2) * TT);
        lds_barrier();                                                   // A_0
        for (int k = nsegc - 1, i = 0; k >= 0; --k, ++i) {
            const int tb = t0 + k * TT;
            if (!STATE_ONLY) {
                if (i > 0) drain(tb + TT);
                lds_barrier();                                           // B_i
            }
            if (k > 0) {
                park(stg[(i + 1) & 1], tb - TT);
                if (k > 1) fetch(tb - 2 * TT);
            }
            lds_barrier();                                               // A_{i+1}
        }
        if (!STATE_ONLY) drain(t0);
        return;
    }

    // ---- scan waves
    __builtin_amdgcn_s_setprio(2);
    const int pr = (lane >> 2) & 7, cw = (lane & 3) | ((lane >> 5) << 2);
    const int cl = 8 * w + cw;
    const int e = e0 + cl;
    f2 An = f2{p.A[(size_t)e * 16 + 2 * pr], p.A[(size_t)e * 16 + 2 * pr + 1]};
    if (p.a_log) An = f2{-fast_exp2(An.x * GFE_LOG2E), -fast_exp2(An.y * GFE_LOG2E)};       // A = -exp(A_log)
    const f2 A2 = An * GFE_LOG2E;
    const float Dv = p.D ? p.D[e] : 0.f;
    const size_t sbase = (((size_t)b * p.nchunks + c) * p.ED + e) * 16 + 2 * pr;
    f2 q = f2{0.f, 0.f};
    if (!STATE_ONLY && p.nchunks > 1) q = chunk_carry<true>(p.qstate, p.sdelta, A2, b, c, e, pr, p.nchunks, p.ED);
    f2 dAacc = f2{0.f, 0.f};
    float dDacc = 0.f, dbacc = 0.f;
    // the segment's start state: fetched one segment ahead (wanted by the very first instruction of phase 1: an HBM round trip there
    // cost 1 300 cycles per segment)
    const float* __restrict__ pck = STATE_ONLY ? nullptr : p.ckpt + (((size_t)b * p.nseg + t0 / SEG) * p.ED + e) * 16 + 2 * pr;
    const size_t ckstride = (size_t)p.ED * 16;
    f2 hck_next = f2{0.f, 0.f};
    if (!STATE_ONLY) hck_next = *reinterpret_cast<const f2*>(pck + (size_t)(nsegc - 1) * ckstride);
    lds_barrier();                                                       // A_0
    S2_STAMP_DECL
    for (int k = nsegc - 1, i = 0; k >= 0; --k, ++i) {
        const BStage& tl = stg[i & 1];
        S2_STAMP(2)
        const f2 hck = hck_next;
        if (k > 0 && !STATE_ONLY) hck_next = *reinterpret_cast<const f2*>(pck + (size_t)(k - 1) * ckstride);
        S2_STAMP(7)

        if (STATE_ONLY) {
#pragma unroll
            for (int j4 = TT / 4 - 1; j4 >= 0; --j4) {
                const f4 dt4 = *reinterpret_cast<const f4*>(&tl.dt[dts_index(cl, 4 * j4)]);
                const f4 g4 = *reinterpret_cast<const f4*>(&tl.g[dts_index(cl, 4 * j4)]);
#pragma unroll
                for (int s = 3; s >= 0; --s) {
                    const f4 bc = tl.bc[(4 * j4 + s) * 8 + pr];
                    const f2 x = A2 * dt4[s];
                    const f2 a = f2{fast_exp2(x.x), fast_exp2(x.y)};
                    q = a * (f2{bc.z, bc.w} * g4[s] + q);
                }
            }
            lds_barrier();                                               // A_{i+1}
            continue;
        }

        // ---- phase 1: the segment's states, forward from the checkpoint; phase 2: the adjoint, last step first.
        // Work unit = 4 steps.  As in the forward, the LDS reads of the NEXT unit are issued ahead of the current unit's arithmetic:
        // two register sets X / Y by unit parity, pinned with sched_barrier.
        f2 av[TT], hs[TT];
        struct H4 { f4 dt4, du4, g4, bc[4]; };
        auto load_u = [&](H4& H, int qd) {
            H.dt4 = *reinterpret_cast<const f4*>(&tl.dt[dts_index(cl, 4 * qd)]);
            H.du4 = *reinterpret_cast<const f4*>(&tl.dtu[dts_index(cl, 4 * qd)]);
#pragma unroll
            for (int s = 0; s < 4; ++s) H.bc[s] = tl.bc[(4 * qd + s) * 8 + pr];
        };
        auto load_g = [&](H4& H, int qd) { H.g4 = *reinterpret_cast<const f4*>(&tl.g[dts_index(cl, 4 * qd)]); };
        f2 h = hck;
        auto fwd_u = [&](const H4& H, int qd) {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int t = 4 * qd + s;
                const f4 bc = H.bc[s];
                const f2 x = A2 * H.dt4[s];
                av[t] = f2{fast_exp2(x.x), fast_exp2(x.y)};
                h = av[t] * h + f2{bc.x, bc.y} * H.du4[s];
                hs[t] = h;
            }
        };
        struct Own { float u, dt, sg, g; };
        auto load_own = [&](Own& O, int g8) {
            const int r = 8 * g8 + pr;
            O.u = tl.eu[r * EPS + cl]; O.dt = tl.edt[r * EPS + cl]; O.sg = tl.esg[r * EPS + cl]; O.g = tl.eg[r * EPS + cl];
        };
        float ddtu_p[8], ddt_p[8];
        auto bwd_u = [&](const H4& H, const Own& O, int qd) {
#pragma unroll
            for (int s = 3; s >= 0; --s) {
                const int t = 4 * qd + s;
                const f4 bc = H.bc[s];
                const f2 Bv = f2{bc.x, bc.y}, Cv = f2{bc.z, bc.w};
                const float gt = H.g4[s], dtv = H.dt4[s], dtu = H.du4[s];
                const f2 dh = Cv * gt + q;
                const f2 dC = hs[t] * gt;
                const f2 dB = dh * dtu;
                ddtu_p[4 * (qd & 1) + s] = fmaf(dh.y, Bv.y, dh.x * Bv.x);
                const f2 hp = (t == 0) ? hck : hs[t == 0 ? 0 : t - 1];
                q = av[t] * dh;
                const f2 da = q * hp;                                      // dL/d(dt*A) of this (t, pair) = dh * a * h_{t-1}
                dAacc += da * dtv;
                ddt_p[4 * (qd & 1) + s] = fmaf(da.y, An.y, da.x * An.x);
                // dB / dC sum over channels.  In the wave: v_permlane32_swap pairs dB with dC (lanes < 32 end up with the dB sum over lane
                // bit 5, lanes >= 32 with the dC sum).  The remaining 4 (channel & 3) x 4 (wave) partials are folded by the staging waves from
                // LDS: every lane stores, so there is no exec masking and no basic-block break inside the unrolled steps.
                // (ds_add_f32 into one shared row instead was 15x slower: ~900 cycles per instruction with 4 lanes per address.)
                const auto sx = __builtin_amdgcn_permlane32_swap(__float_as_uint(dB.x), __float_as_uint(dC.x), false, false);
                const auto sy = __builtin_amdgcn_permlane32_swap(__float_as_uint(dB.y), __float_as_uint(dC.y), false, false);
                *reinterpret_cast<f2*>(&red[(w * 4 + (lane & 3)) * RSL + t * 32 + (lane >> 5) * 16 + 2 * pr]) =
                    f2{__uint_as_float(sx[0]) + __uint_as_float(sx[1]), __uint_as_float(sy[0]) + __uint_as_float(sy[1])};
            }
            if ((qd & 1) == 0) {                                           // group complete: the lane that owns (step r, channel cl) finishes it
                const float ddtu = reduce_pairs8(ddtu_p);
                const float ddtA = reduce_pairs8(ddt_p);
                const int r = 4 * qd + pr;
                const float dd = fmaf(ddtu, O.u, ddtA) * O.sg;
                IO<T>::st(&otile[0][r * EPS + cl], fmaf(ddtu, O.dt, Dv * O.g));
                IO<T>::st(&otile[1][r * EPS + cl], dd);
                dDacc = fmaf(O.g, O.u, dDacc);
                dbacc += dd;
            }
        };
#define SB __builtin_amdgcn_sched_barrier(0)
        {
            H4 X, Y;
            Own oa, ob;
            load_u(X, 0); SB;
            load_u(Y, 1); SB; fwd_u(X, 0); SB;
            load_u(X, 2); SB; fwd_u(Y, 1); SB;
            load_u(Y, 3); SB; fwd_u(X, 2); SB;
            load_u(X, 4); SB; fwd_u(Y, 3); SB;
            S2_STAMP(8)
            load_u(Y, 5); SB; fwd_u(X, 4); SB;
            load_u(X, 6); SB; fwd_u(Y, 5); SB;
            load_u(Y, 7); load_g(Y, 7); SB; fwd_u(X, 6); SB;
            load_g(X, 6); load_own(oa, 3); SB; fwd_u(Y, 7); SB;
            S2_STAMP(3)
            lds_barrier();                                               // B_i: the previous segment's red / otile have been drained
            SB;
            bwd_u(Y, oa, 7); SB;
            load_u(Y, 5); load_g(Y, 5); SB; bwd_u(X, oa, 6); SB;
            load_u(X, 4); load_g(X, 4); load_own(ob, 2); SB; bwd_u(Y, ob, 5); SB;
            load_u(Y, 3); load_g(Y, 3); SB; bwd_u(X, ob, 4); SB;
            S2_STAMP(9)
            load_u(X, 2); load_g(X, 2); load_own(oa, 1); SB; bwd_u(Y, oa, 3); SB;
            load_u(Y, 1); load_g(Y, 1); SB; bwd_u(X, oa, 2); SB;
            load_u(X, 0); load_g(X, 0); load_own(ob, 0); SB; bwd_u(Y, ob, 1); SB;
            bwd_u(X, ob, 0);
        }
#undef SB
        S2_STAMP(4)
        lds_barrier();                                                   // A_{i+1}
        S2_STAMP(5)
    }
    S2_STAMP_FLUSH(16)
    if (STATE_ONLY) {
        *reinterpret_cast<f2*>(p.qstate + sbase) = q;
        return;
    }
    if (p.a_log) dAacc *= An;                                            // d/dA_log = dA * dA/dA_log = dA * A
    // dD / dbias: the 8 owner lanes of a channel (pair bits 2, 3, 4) hold partial sums over their steps
    dDacc += __shfl_xor(dDacc, 4, 64); dbacc += __shfl_xor(dbacc, 4, 64);
    dDacc += __shfl_xor(dDacc, 8, 64); dbacc += __shfl_xor(dbacc, 8, 64);
    dDacc += __shfl_xor(dDacc, 16, 64); dbacc += __shfl_xor(dbacc, 16, 64);
    if (p.part_vec) {                                                    // this (sample, chunk)'s row of partials: plain stores, summed in order later
        float* pv = p.part_vec + ((size_t)b * p.nchunks + c) * ((size_t)p.ED * 18);
        *reinterpret_cast<f2*>(pv + (size_t)e * 16 + 2 * pr) = dAacc;
        if (pr == 0) { pv[(size_t)p.ED * 16 + e] = dDacc; pv[(size_t)p.ED * 17 + e] = dbacc; }
        return;
    }
    atomicAdd(p.dAws + (size_t)e * 16 + 2 * pr, dAacc.x);
    atomicAdd(p.dAws + (size_t)e * 16 + 2 * pr + 1, dAacc.y);
    if (pr == 0) {
        if (p.dDws) atomicAdd(p.dDws + e, dDacc);
        if (p.dbiasws) atomicAdd(p.dbiasws + e, dbacc);
    }
}

// Fixed-order sums of the backward's partials (S2Bwd::part_vec / part_bc).  Blocks [0, nb_vec): dA / dD / dbias += sum over the (sample,
// chunk) rows in order (accumulating: the targets may be the optimizer's gradient slots); blocks [nb_vec, ...): dB / dC rows = sum over the
// channel groups in order (plain store: the rows need no zero fill).
__global__ __launch_bounds__(256) void sscan2_fold_kernel(const S2Bwd p, int nb_vec, int G) {
    if ((int)blockIdx.x < nb_vec) {
        const int i = blockIdx.x * 256 + threadIdx.x, n = p.ED * 18;
        if (i >= n) return;
        float s = 0.f;
        const int rows = p.B * p.nchunks;
        for (int k = 0; k < rows; ++k) s += p.part_vec[(size_t)k * n + i];
        if (i < p.ED * 16) p.dAws[i] += s;
        else if (i < p.ED * 17) { if (p.dDws) p.dDws[i - p.ED * 16] += s; }
        else if (p.dbiasws) p.dbiasws[i - p.ED * 17] += s;
        return;
    }
    const int64_t i = (int64_t)((int)blockIdx.x - nb_vec) * 256 + threadIdx.x;           // (b, t, j)
    if (i >= (int64_t)p.B * p.L * 32) return;
    const int j = (int)(i & 31);
    const int64_t bt = i >> 5;
    const int b = (int)(bt / p.L), t = (int)(bt - (int64_t)b * p.L);
    // four chains (groups g, g + 1, g + 2, g + 3 of every four), folded in a fixed order: one chain is G dependent L2 round trips (10 us at 32)
    const float* pb = p.part_bc + ((size_t)b * G * p.L + t) * 32 + j;
    const size_t gs = (size_t)p.L * 32;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int g = 0;
    for (; g + 3 < G; g += 4) { s0 += pb[g * gs]; s1 += pb[(g + 1) * gs]; s2 += pb[(g + 2) * gs]; s3 += pb[(g + 3) * gs]; }
    for (; g < G; ++g) s0 += pb[g * gs];
    (j < 16 ? p.dBws : p.dCws)[(size_t)bt * p.ld_dbc + (j & 15)] = (s0 + s1) + (s2 + s3);
}

template <typename T>
int sscan2_bwd_launch(const S2Bwd& p, hipStream_t st) {
    const dim3 blk(512), grid((unsigned)(p.ED / CB), p.nchunks, p.B);
    if (p.nchunks > 1) hipLaunchKernelGGL((sscan2_bwd_kernel<T, true>), grid, blk, 0, st, p);      // local adjoint carries of every chunk
    hipLaunchKernelGGL((sscan2_bwd_kernel<T, false>), grid, blk, 0, st, p);
    if (p.part_vec) {
        const int nb_vec = (int)ceil_div((int64_t)p.ED * 18, 256);
        const int64_t nb_bc = ceil_div((int64_t)p.B * p.L * 32, 256);
        hipLaunchKernelGGL(sscan2_fold_kernel, dim3((unsigned)(nb_vec + nb_bc)), dim3(256), 0, st, p, nb_vec, p.ED / CB);
    }
    return gfe_launch_status();
}

}  // namespace

extern "C" {

int gfe_sscan2_plan(int64_t B, int64_t L, int64_t ED, int chunk_req, int* T_out, int* nchunks_out) {
    GFE_REQUIRE(B > 0 && L > 0 && ED > 0 && ED % CB == 0 && T_out && nchunks_out, GFE_ERR_SHAPE);
    int64_t T = L;
    if (chunk_req > 0) {
        T = ceil_div(chunk_req, SEG) * SEG;                       // chunk starts must be checkpoint positions
    } else {
        const int64_t waves = B * (ED / 8);                       // one wave = 8 channels x 8 state pairs
        if (waves < 768) {                                        // cannot fill 1024 SIMDs: cut L (two passes + carry)
            const int64_t want = ceil_div(2048, waves);           // two waves per SIMD: measured -8 % (B = 1) / -7 % (B = 4) kernel time vs one
            T = ceil_div(ceil_div(L, want), SEG) * SEG;
            if (T < 64) T = 64;
        }
    }
    if (T > L) T = L;
    *T_out = (int)T;
    *nchunks_out = (int)ceil_div(L, T);
    return GFE_OK;
}

int gfe_sscan2_fwd(const void* u, const void* delta, const float* A, const void* Bm, const void* Cm,
                   const float* D, const void* z, const float* delta_bias, void* y, void* yscan,
                   float* hstate, float* sdelta, float* ckpt,
                   int64_t B, int64_t L, int64_t ED, int T, int delta_softplus, int dtype, int bc_dtype,
                   int64_t ld_z, int64_t ld_bc, int a_is_log, void* stream) {
    GFE_REQUIRE(bc_dtype == GFE_F32 || bc_dtype == GFE_BF16, GFE_ERR_DTYPE);
    GFE_REQUIRE(u && delta && A && Bm && Cm && y, GFE_ERR_NULL);
    if (ld_z <= 0) ld_z = ED;
    if (ld_bc <= 0) ld_bc = 16;
    GFE_REQUIRE(ld_z >= ED && ld_z % 4 == 0 && ld_bc >= 16 && ld_bc % 4 == 0 && ld_z <= 0x7fffffff && ld_bc <= 0x7fffffff, GFE_ERR_SHAPE);
    GFE_REQUIRE(B > 0 && L > 0 && ED > 0 && T > 0 && ED % CB == 0 && B <= 65535, GFE_ERR_SHAPE);
    S2Fwd p;
    p.u = u; p.delta = delta; p.z = z; p.Bm = Bm; p.Cm = Cm; p.A = A; p.D = D; p.dbias = delta_bias; p.y = y; p.yscan = yscan;
    p.hstate = hstate; p.sdelta = sdelta; p.ckpt = ckpt;
    p.B = (int)B; p.L = (int)L; p.ED = (int)ED; p.T = T; p.nchunks = (int)ceil_div(L, T); p.softplus = delta_softplus;
    p.nseg = (int)ceil_div(L, SEG); p.bc_bf16 = bc_dtype == GFE_BF16;
    p.ld_z = (int)ld_z; p.ld_bc = (int)ld_bc; p.a_log = a_is_log != 0;
    GFE_REQUIRE(p.nchunks <= 65535, GFE_ERR_SHAPE);
    GFE_REQUIRE(p.nchunks == 1 || (hstate && sdelta && T % SEG == 0), GFE_ERR_SHAPE);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == GFE_F32) return sscan2_fwd_launch<float>(p, st);
    if (dtype == GFE_BF16) return sscan2_fwd_launch<bf16_t>(p, st);
    return GFE_ERR_DTYPE;
}

int gfe_sscan2_bwd(const void* u, const void* delta, const float* A, const void* Bm, const void* Cm,
                   const float* D, const void* z, const float* delta_bias, const void* dy, const void* yscan,
                   void* du, void* ddelta, void* dz,
                   float* dA_ws, float* dB_ws, float* dC_ws, float* dD_ws, float* dbias_ws,
                   const float* ckpt, float* qstate, const float* sdelta, float* part_vec, float* part_bc,
                   int64_t B, int64_t L, int64_t ED, int T, int delta_softplus, int dtype, int bc_dtype,
                   int64_t ld_z, int64_t ld_bc, int64_t ld_dbc, int a_is_log, void* stream) {
    GFE_REQUIRE(bc_dtype == GFE_F32 || bc_dtype == GFE_BF16, GFE_ERR_DTYPE);
    if (ld_z <= 0) ld_z = ED;
    if (ld_bc <= 0) ld_bc = 16;
    if (ld_dbc <= 0) ld_dbc = 16;
    GFE_REQUIRE(ld_z >= ED && ld_z % 4 == 0 && ld_bc >= 16 && ld_bc % 4 == 0 && ld_dbc >= 16 && ld_z <= 0x7fffffff && ld_bc <= 0x7fffffff && ld_dbc <= 0x7fffffff, GFE_ERR_SHAPE);
    GFE_REQUIRE(u && delta && A && Bm && Cm && dy && du && ddelta && dA_ws && dB_ws && dC_ws && ckpt, GFE_ERR_NULL);
    GFE_REQUIRE(!z || (dz && yscan), GFE_ERR_NULL);
    GFE_REQUIRE((part_vec == nullptr) == (part_bc == nullptr), GFE_ERR_NULL);
    GFE_REQUIRE(B > 0 && L > 0 && ED > 0 && T > 0 && ED % CB == 0 && B <= 65535, GFE_ERR_SHAPE);
    GFE_REQUIRE(!part_vec || (B * L * 32 <= 0x7fffffff * (int64_t)256 && ED * 18 <= 0x7fffffff), GFE_ERR_SHAPE);
    S2Bwd p;
    p.part_vec = part_vec; p.part_bc = part_bc;
    p.u = u; p.delta = delta; p.z = z; p.Bm = Bm; p.Cm = Cm; p.dy = dy; p.yscan = yscan; p.A = A; p.D = D; p.dbias = delta_bias;
    p.du = du; p.ddelta = ddelta; p.dz = dz; p.dAws = dA_ws; p.dBws = dB_ws; p.dCws = dC_ws; p.dDws = dD_ws; p.dbiasws = dbias_ws;
    p.ckpt = ckpt; p.qstate = qstate; p.sdelta = sdelta;
    p.B = (int)B; p.L = (int)L; p.ED = (int)ED; p.T = T; p.nchunks = (int)ceil_div(L, T); p.softplus = delta_softplus;
    p.nseg = (int)ceil_div(L, SEG); p.bc_bf16 = bc_dtype == GFE_BF16;
    p.ld_z = (int)ld_z; p.ld_bc = (int)ld_bc; p.ld_dbc = (int)ld_dbc; p.a_log = a_is_log != 0;
    GFE_REQUIRE(p.nchunks <= 65535, GFE_ERR_SHAPE);
    GFE_REQUIRE(p.nchunks == 1 || (qstate && sdelta && T % SEG == 0), GFE_ERR_SHAPE);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == GFE_F32) return sscan2_bwd_launch<float>(p, st);
    if (dtype == GFE_BF16) return sscan2_bwd_launch<bf16_t>(p, st);
    return GFE_ERR_DTYPE;
}

#ifdef GFE_S2_STAMPS
int gfe_dbg_s2_stamps(unsigned long long* host_out) { return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_s2_stamps), sizeof(unsigned long long) * 48) == hipSuccess ? 0 : -4; }
#endif

}  // extern "C"
