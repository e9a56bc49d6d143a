// Fused selective scan (Mamba S6 recurrence), N = 16 states -- single-pass formulation for gfx950.
//
// Replaces, like sscan.hip (which stays for N = 4 / 8), the reference's
//   cross_atten/mamba.py:265-286  MambaBlock.selective_scan  (deltaA/deltaB/BX -> pscan -> hs@C + D*x)
//   cross_atten/mamba.py:243-259  softplus(delta + dt_proj.bias) and the y*silu(z) gate of the `selective_scan_fn` plug-in
//   cross_atten/pscan.py:151-224  PScan.forward / backward (the recurrence and its adjoint)
//
// Why a second formulation: the scan is vector-ISSUE bound -- one v_exp_f32 and four f32 operations per state-step -- so the
// only lever is the number of instructions issued per state-step.  sscan.hip keeps a channel's 16 states in one lane; filling the chip then
// needs chunks along L and TWO passes over every chunk (local end state, then the real pass), i.e. two exp per state-step.  Here a lane owns
// one channel x one PAIR of states: B*ED*8 lanes fill 1024 SIMDs from B = 8 up without cutting L, every state-step is computed once, the pair
// keeps the arithmetic in v_pk_* form, and a segment's states fit in registers for the backward (no third exp, no LDS checkpoints).
// L is cut into chunks (two passes + carry) only when B*ED/8 waves cannot fill the chip (B = 1).
//
// Work split (round 6): block = 8 waves = 32 channels of one (batch, chunk): 4 SCAN waves, each 8 channels x 8 state pairs
//     lane bits {2,3,4} = pair p,  lane bits {0,1,5} = channel c within the wave
// + 4 STAGING waves, their SIMD partners.  The scan waves keep only what is per (step, channel, state) and hand every sum over lanes to the
// LDS array as raw per-lane partials; the staging waves fetch the rows (branch-free raw buffer I/O: counted vmcnt waits), do the
// per-(step, channel) math once (softplus, SiLU, products), park {dt, dt*u} and {B[2p], B[2p+1], C[2p], C[2p+1]} in LDS, and finish the
// previous tile's outputs from the partial rows (in-lane f4 sums) as whole-row stores.  What a lone wave pays per instruction, and why this
// formulation sits on its floor at B = 8, is measured in DESIGN.md 4.3 (tools/probes/lone_wave_mix.hip, tools/scan_exp/).
#include "common.h"

namespace {

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

constexpr int TT = 32;       // steps per LDS tile == checkpoint interval: the backward recomputes one tile at a time with its states in registers
constexpr int CB = 32;       // channels per block (4 waves x 8)
constexpr int SEG = TT;

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
// (GFE_LB: the same barrier between two timing-fuzz sites -- common.h; nothing in the product build)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
#define GFE_LB() do { GFE_FUZZ(); lds_barrier(); GFE_FUZZ(); } while (0)

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
                // the first folded chunk meets H = 0: its own sum of dt is not needed -- and not written: the state passes skip the chunk
                // whose end state nobody folds (the last one forward, the first one backward), so that slot of sdelta is uninitialised memory
                const f2 x = A2 * sdv[j];
                H = (k0 + j == 0) ? loc[j] : f2{fast_exp2(x.x), fast_exp2(x.y)} * H + loc[j];
            }
        }
    }
    return H;
}

// ------------------------------------------------------------------------------------------------
// forward (round 6 formulation)
// ------------------------------------------------------------------------------------------------
// Row I/O of the staging waves goes through raw buffer instructions: a row past the chunk's end is an out-of-range offset, which the
// hardware answers with zeros (loads) or drops (stores).  That keeps the staging waves' instruction stream free of exec-masked branches, so
// hipcc's waitcnt pass sees a straight line and emits COUNTED s_waitcnt vmcnt(N) in front of the first use of a fetched row -- with the
// `if (row < nrows)` blocks of rounds 2-5 it had to assume the worst and emitted vmcnt(0), i.e. every tile waited for the previous
// tile's output stores (and, in the backward, for its float atomics: 600-3000 cycles each, MI355X_MICROARCH.md).
typedef unsigned u2v __attribute__((ext_vector_type(2)));
typedef unsigned u4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t row_rsrc(const void* base, unsigned bytes) {
    const uint64_t a = (uint64_t)base;
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)a), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32));
    return __builtin_amdgcn_make_buffer_rsrc((void*)(((uint64_t)hi << 32) | lo), 0, __builtin_amdgcn_readfirstlane((int)(base ? bytes : 0u)), 0x00020000);
}

// four consecutive channels of one row: 16 B (f32) or 8 B (bf16)
template <typename T> struct Row4;
template <> struct Row4<float> {
    typedef u4v raw;
    static __device__ __forceinline__ raw ld(__amdgpu_buffer_rsrc_t r, unsigned off) { return __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0); }
    static __device__ __forceinline__ void st(f4 v, __amdgpu_buffer_rsrc_t r, unsigned off) {
        __builtin_amdgcn_raw_buffer_store_b128(u4v{__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)}, r, off, 0, 0);
    }
    static __device__ __forceinline__ f4 unpack(raw v) { return f4{__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)}; }
    static __device__ __forceinline__ raw zero() { return raw{0u, 0u, 0u, 0u}; }
};
template <> struct Row4<bf16_t> {
    typedef u2v raw;
    static __device__ __forceinline__ raw ld(__amdgpu_buffer_rsrc_t r, unsigned off) { return __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, 0); }
    static __device__ __forceinline__ void st(f4 v, __amdgpu_buffer_rsrc_t r, unsigned off) {
        __builtin_amdgcn_raw_buffer_store_b64(u2v{pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w)}, r, off, 0, 0);
    }
    static __device__ __forceinline__ f4 unpack(raw v) { return f4{bf16lo_to_f32(v.x), bf16hi_to_f32(v.x), bf16lo_to_f32(v.y), bf16hi_to_f32(v.y)}; }
    static __device__ __forceinline__ raw zero() { return raw{0u, 0u}; }
};

// a * b.y for a register pair b: hipcc selects the op_sel form for element 1 of a pair it loaded as such, but moves element 3 of a 16-byte
// LDS fragment into a fresh register first (one v_mov per odd step)
__device__ __forceinline__ f2 pk_mul_hi(f2 a, f2 b) {
    f2 r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

constexpr int PS = 40;           // row stride (floats) of the per-pair partial rows: 32 channels + 8: a step's 8 rows are 1280 B = 5 x 256 B, so the partials
                                 // of two steps leave as ONE ds_write2st64_b32 (pair rows p and p+4 share banks: 2-way, free on a 4-byte store; the staging
                                 // waves' ds_read_b128 of two steps' rows collide 2-way too: 8 reads per lane and tile)
constexpr int YP_TILE = TT * 8 * PS;

// LDS image of one 32-step tile as the scan waves read it
struct FTile {
    f4 dd[TT / 2 * CB];          // [step / 2][slot] {dt, dt*u} of steps 2j and 2j+1: dt = softplus(delta + bias), 0 past the chunk end -- one
                                 //                  ds_read_b128 per two steps (hipcc fuses two ds_read_b64 of a [step][channel] image into ds_read2_b64: half rate)
    f4 bc[TT * 8];               // [step][pair]     {B[2p], B[2p+1], C[2p], C[2p+1]}                        -- one ds_read_b128 per step
};
// slot of a channel in FTile::dd: the staging lanes of a row write 8 bytes each for channels 4*sc + j, sc = 0..7 -- with the identity those are
// 16 dwords apart (4-way bank conflict on every ds_write_b64); the XOR spreads the 8 lanes over the 8 bank quads
__device__ __forceinline__ int dd_slot(int ch) { return ch ^ (ch >> 3); }

// One block = 32 channels x one chunk = 8 waves: 4 SCAN waves (0-3) + 4 STAGING waves (4-7, their SIMD partners).
//
// Round 6: the scan waves keep ONLY what is per (step, channel, state): a = exp2(A dt) (v_pk_mul + 2 v_exp), h = a h + B dt u (v_pk_mul +
// v_pk_fma) and the lane's share of the output, C[2p] h[2p] + C[2p+1] h[2p+1] (v_mul + v_fmac) -- which they DROP INTO LDS as it is, one
// ds_write_b32 per step, [step][pair][channel].  Everything that is per (step, channel) belongs to the staging waves: they fetch the rows,
// compute dt / dt*u once, and finish the outputs of the previous tile: the sum over the 8 pair rows is 7 in-lane f4 additions on
// ds_read_b128 fragments (the LDS array does the transposition that rounds 2-5 did with a permlane / DPP butterfly in the scan wave: 76
// of its 330 vector instructions per tile, plus 35 s_nop hazard pads), then + D u, the gate, the rounding and whole-row stores.  D u and
// silu(z) never touch LDS: the staging lane that computed them when it parked the tile is the lane that finishes the tile two barriers
// later (two register sets, loop unrolled by two).
// STATE_ONLY (first pass of the chunked plan): end state from h = 0 and the chunk's sum of dt; no outputs.
template <typename T, typename TBC, bool STATE_ONLY>
__global__ __launch_bounds__(512) void sscan2_fwd_kernel(const S2Fwd p) {
    typedef Row4<T> R;
    typedef Row4<TBC> RBC;
    GFE_FUZZ_INIT();
    __shared__ __attribute__((aligned(16))) FTile tiles[3];     // a ring of three: tile k+1 is complete one barrier BEFORE the scan waves finish tile k, so they read its first steps ahead of the barrier
    __shared__ __attribute__((aligned(16))) float ypart[STATE_ONLY ? TT * CB : 2 * YP_TILE];   // STATE_ONLY: the staging lanes' sums of dt on their way to sdelta
    const bool staging = __builtin_amdgcn_readfirstlane(threadIdx.x) >= 256;      // wave-uniform role
    const int tid = threadIdx.x & 255, lane = tid & 63, w = tid >> 6;
    const int e0 = xcd_paired_group(blockIdx.x, gridDim.x) * CB, c = blockIdx.y, b = blockIdx.z;
    const int t0 = c * p.T, t1 = min(p.L, t0 + p.T);
    const int nrows = t1 - t0;
    const int nt = (nrows + TT - 1) / TT;
    const size_t row0 = (size_t)b * p.L + t0;

    if (staging) {
        const int sr = tid >> 3, sc = tid & 7;                       // row sr of the tile, channels 4*sc .. 4*sc+3 of the block
        const bool has_z = !STATE_ONLY && p.z != nullptr;
        const unsigned esz = sizeof(T);
        const unsigned span = (unsigned)((nrows - 1) * p.ED + CB) * esz;            // this block's bytes of a (B, L, ED) tensor from its first row on
        const __amdgpu_buffer_rsrc_t rs_u = row_rsrc((const T*)p.u + row0 * p.ED + e0, span);
        const __amdgpu_buffer_rsrc_t rs_d = row_rsrc((const T*)p.delta + row0 * p.ED + e0, span);
        const __amdgpu_buffer_rsrc_t rs_z = row_rsrc(has_z ? (const T*)p.z + row0 * p.ld_z + e0 : nullptr, (unsigned)((nrows - 1) * p.ld_z + CB) * esz);
        const __amdgpu_buffer_rsrc_t rs_y = row_rsrc(STATE_ONLY ? nullptr : (T*)p.y + row0 * p.ED + e0, span);
        const __amdgpu_buffer_rsrc_t rs_ys = row_rsrc((STATE_ONLY || !p.yscan) ? nullptr : (T*)p.yscan + row0 * p.ED + e0, span);
        // B / C role: waves 4-5 fetch B, 6-7 fetch C: row br of the tile, floats 4*bq .. 4*bq+3
        const int br = (tid & 127) >> 2, bq = tid & 3;
        const bool isB = tid < 128;
        const unsigned bsz = sizeof(TBC);
        const __amdgpu_buffer_rsrc_t rs_bc = row_rsrc((const char*)(isB ? p.Bm : p.Cm) + row0 * p.ld_bc * bsz, (unsigned)((nrows - 1) * p.ld_bc + 16) * bsz);
        const unsigned rowb = (unsigned)p.ED * esz, rowbz = (unsigned)p.ld_z * esz, rowbbc = (unsigned)p.ld_bc * bsz;
        const unsigned off_e = (unsigned)(sr * p.ED + 4 * sc) * esz, off_z = (unsigned)(sr * p.ld_z + 4 * sc) * esz, off_bc = (unsigned)(br * p.ld_bc + 4 * bq) * bsz;
        float sbias[4], sD[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            sbias[k] = p.dbias ? p.dbias[e0 + 4 * sc + k] : 0.f;
            sD[k] = (!STATE_ONLY && p.D) ? p.D[e0 + 4 * sc + k] : 0.f;
        }
        typename R::raw ru = R::zero(), rd = R::zero(), rz = R::zero();
        typename RBC::raw rbc = RBC::zero();
        f4 sdt = f4{0.f, 0.f, 0.f, 0.f};
        struct Keep { f4 du, gate; };                                 // D*u and silu(z) of a parked tile, until that tile's outputs are finished
        auto fetch = [&](int k) {                                     // global -> registers, tile k (rows past the end: zeros)
            const unsigned kr = (unsigned)(k * TT);
            ru = R::ld(rs_u, off_e + kr * rowb);
            rd = R::ld(rs_d, off_e + kr * rowb);
            if (!STATE_ONLY) rz = R::ld(rs_z, off_z + kr * rowbz);
            if (!STATE_ONLY || isB) rbc = RBC::ld(rs_bc, off_bc + kr * rowbbc);
        };
        auto park = [&](int k, int buf, Keep& K) {                    // registers -> LDS (tiles[buf], buf = k mod 3); the per-(t, channel) math happens once, here
            const f4 fu = R::unpack(ru), fd = R::unpack(rd), fz = R::unpack(rz);
            const bool valid = k * TT + sr < nrows;
            FTile& tl = tiles[buf];
            float dt[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float raw = fd[j] + sbias[j];
                float v = softplus_nb(raw);
                asm volatile("" : "+v"(v));                           // (keeps the four softplus chains in one block: a uniform branch per element would serialise them)
                v = p.softplus ? v : raw;
                dt[j] = valid ? v : 0.f;                              // a = exp2(0) = 1, dt*u = 0: steps past the end leave the state alone
                if (!STATE_ONLY) { K.du[j] = sD[j] * fu[j]; K.gate[j] = has_z ? siluf_(fz[j]) : 1.f; }
                else sdt[j] += dt[j];
            }
            f2* ddp = reinterpret_cast<f2*>(&tl.dd[(sr >> 1) * CB]) + (sr & 1);
#pragma unroll
            for (int j = 0; j < 4; ++j) ddp[2 * dd_slot(4 * sc + j)] = f2{dt[j], dt[j] * fu[j]};
            if (!STATE_ONLY || isB) {
                const f4 v = RBC::unpack(rbc);
                float* bcp = reinterpret_cast<float*>(&tl.bc[br * 8 + 2 * bq]) + (isB ? 0 : 2);
                *reinterpret_cast<f2*>(bcp) = f2{v.x, v.y};
                *reinterpret_cast<f2*>(bcp + 4) = f2{v.z, v.w};
            }
        };
        auto finish = [&](int k, const Keep& K) {                     // tile k's outputs: 8 pair rows -> one value per (step, channel)
            const float* yp = &ypart[(k & 1) * YP_TILE + sr * 8 * PS + 4 * sc];
            f4 acc = *reinterpret_cast<const f4*>(yp);
#pragma unroll
            for (int q = 1; q < 8; ++q) acc += *reinterpret_cast<const f4*>(yp + q * PS);
            const f4 yb = acc + K.du;
            const unsigned off = off_e + (unsigned)(k * TT) * rowb;
            R::st(yb * K.gate, rs_y, off);
            R::st(yb, rs_ys, off);                                    // (no yscan: zero-sized resource, the store is dropped)
        };
        Keep k0, k1, k2;                                              // tile j <-> set j mod 3 <-> tiles[j mod 3]
        fetch(0);
        park(0, 0, k0);
        fetch(1);
        park(1, 1, k1);
        fetch(2);
        GFE_LB();                                                // tiles 0 and 1 are ready
        park(2, 2, k2);                                               // iteration 0: nothing to finish yet
        fetch(3);
        GFE_LB();
        for (int k = 1; k < nt; k += 3) {                             // iteration k: the scan waves are on tile k; finish k-1, park k+2, fetch k+3
            if (!STATE_ONLY) finish(k - 1, k0);
            park(k + 2, 0, k0);
            fetch(k + 3);
            GFE_LB();
            if (k + 1 < nt) {
                if (!STATE_ONLY) finish(k, k1);
                park(k + 3, 1, k1);
                fetch(k + 4);
                GFE_LB();
            }
            if (k + 2 < nt) {
                if (!STATE_ONLY) finish(k + 1, k2);
                park(k + 4, 2, k2);
                fetch(k + 5);
                GFE_LB();
            }
        }
        if (!STATE_ONLY) {
            const int r3 = (nt - 1) % 3;
            if (r3 == 0) finish(nt - 1, k0); else if (r3 == 1) finish(nt - 1, k1); else finish(nt - 1, k2);
        } else {
            *reinterpret_cast<f4*>(&ypart[sr * CB + 4 * sc]) = sdt;
            GFE_LB();
            if (tid < CB) {
                float s = 0.f;
#pragma unroll 8
                for (int r = 0; r < TT; ++r) s += ypart[r * CB + tid];
                p.sdelta[((size_t)b * p.nchunks + c) * p.ED + e0 + tid] = s;
            }
        }
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
    const bool do_ck = !STATE_ONLY && p.ckpt != nullptr;         // (uniform)
    float* ckp = p.ckpt + ((((size_t)b * p.nseg + t0 / SEG) * p.ED + e) * 16 + 2 * pr);
    const size_t ckstride = (size_t)p.ED * 16;
    GFE_LB();
    constexpr int LD = 6;
    f4 ddr[TT / 2], bcr[TT];
    int kb = 0;                                                      // k mod 3
    auto head_reads = [&](int buf) {                                 // LDS reads of a tile's first LD + 2 steps
        const f4* ddp = &tiles[buf].dd[dd_slot(cl)];
        const f4* bcp = &tiles[buf].bc[pr];
#pragma unroll
        for (int t = 0; t < LD + 2; ++t) {
            if ((t & 1) == 0) ddr[t >> 1] = ddp[(t >> 1) * CB];
            bcr[t] = bcp[t * 8];
        }
    };
    head_reads(0);
    S2_STAMP_DECL
    for (int k = 0; k < nt; ++k) {
        S2_STAMP(0)
        if (do_ck) { *reinterpret_cast<f2*>(ckp) = h; ckp += ckstride; }     // every tile starts a segment (t0 is a multiple of SEG)
        const f4* ddp = &tiles[kb].dd[dd_slot(cl)];
        const f4* bcp = &tiles[kb].bc[pr];
        float* yp = &ypart[STATE_ONLY ? 0 : (k & 1) * YP_TILE + pr * PS + cl];
        kb = kb == 2 ? 0 : kb + 1;
        // The wave is ALONE with this stream for most of its life (its partner finishes a tile's staging in a third of the time), and a lone
        // in-order wave pays the full result latency of every instruction whose operand is younger than ~4 instructions: compiled as a
        // unit of arithmetic the tile ran at 87 cycles per step for 7 vector instructions (profiles/r06/scan_ablation.txt).  So the tile is
        // written as a software pipeline over steps, one instruction per statement, the order pinned with sched_barrier:
        //   step t+LD: LDS reads   t+2: x = A dt, xb = B dt u   t+1: a = exp2(x)   t: h = a h + xb   t-1: p = C . h   t-2: p -> LDS
        // so that every operand was produced at least four instructions earlier (h alternates between two register pairs).
        f2 xr[2], ar[2], xbr[3], hr[2];
        float pp[4];
        hr[1] = h;                                                   // h[-1]
#define SB __builtin_amdgcn_sched_barrier(0)
#pragma unroll
        for (int t = -2; t < TT + 2; ++t) {
            if (!STATE_ONLY && t - 1 >= 0 && t - 1 < TT) { pp[(t + 3) & 3] = hr[(t + 1) & 1].x * bcr[t - 1].z; SB; }
            if (t + 2 < TT) { const f4 d = ddr[(t + 2) >> 1]; xr[t & 1] = A2 * (((t + 2) & 1) ? d.z : d.x); SB; }
            if (t + 1 >= 0 && t + 1 < TT) { ar[(t + 1) & 1].x = fast_exp2(xr[(t + 1) & 1].x); SB; }
            if (t >= 0 && t < TT) { hr[t & 1] = __builtin_elementwise_fma(ar[t & 1], hr[(t + 1) & 1], xbr[(t + 3) % 3]); SB; }
            if (!STATE_ONLY && t - 1 >= 0 && t - 1 < TT) { pp[(t + 3) & 3] = fmaf(hr[(t + 1) & 1].y, bcr[t - 1].w, pp[(t + 3) & 3]); SB; }
            if (t + 1 >= 0 && t + 1 < TT) { ar[(t + 1) & 1].y = fast_exp2(xr[(t + 1) & 1].y); SB; }
            if (t + 2 < TT) {
                const f4 d = ddr[(t + 2) >> 1];
                const f2 Bv = f2{bcr[t + 2].x, bcr[t + 2].y};
                if ((t + 2) & 1) xbr[(t + 5) % 3] = pk_mul_hi(Bv, __builtin_shufflevector(d, d, 2, 3)); else xbr[(t + 5) % 3] = Bv * d.y;
                SB;
            }
            if (!STATE_ONLY && t - 2 >= 0 && ((t - 2) & 1)) { yp[(t - 3) * 8 * PS] = pp[(t + 1) & 3]; yp[(t - 2) * 8 * PS] = pp[(t + 2) & 3]; }   // steps t-3, t-2: one ds_write2st64_b32
            if (t + LD + 2 < TT && t + LD + 2 >= LD + 2) {
                if (((t + LD + 2) & 1) == 0) ddr[(t + LD + 2) >> 1] = ddp[((t + LD + 2) >> 1) * CB];
                bcr[t + LD + 2] = bcp[(t + LD + 2) * 8];
            }
            SB;
        }
#undef SB
        h = hr[(TT - 1) & 1];
        head_reads(kb);                                              // the next tile's first steps: in flight across the barrier (that tile was complete one barrier ago)
        if (!STATE_ONLY) {
            // the barrier publishes this tile's partial rows: LDS operations complete in order, so "at most the head reads outstanding" means every write has landed
            GFE_FUZZ();
            asm volatile("s_waitcnt lgkmcnt(%0)\n\ts_barrier" :: "n"(LD + 2 + (LD + 2 + 1) / 2) : "memory");
            GFE_FUZZ();
            S2_STAMP(4)
            continue;
        }
        S2_STAMP(2)
        GFE_LB();
        S2_STAMP(4)
    }
    S2_STAMP_FLUSH(STATE_ONLY ? 36 : 0)
    if (STATE_ONLY) {
        *reinterpret_cast<f2*>(p.hstate + sbase) = h;
        GFE_LB();                                               // (the staging waves' sum of dt)
    }
}

template <typename T, typename TBC>
int sscan2_fwd_launch2(const S2Fwd& p, hipStream_t st) {
    const dim3 blk(512), grid((unsigned)(p.ED / CB), p.nchunks, p.B);
    // local end states + sum dt of every chunk but the LAST (nobody folds its end state: with two chunks that is half of this pass)
    if (p.nchunks > 1) hipLaunchKernelGGL((sscan2_fwd_kernel<T, TBC, true>), dim3(grid.x, grid.y - 1, grid.z), blk, 0, st, p);
    hipLaunchKernelGGL((sscan2_fwd_kernel<T, TBC, false>), grid, blk, 0, st, p);
    return gfe_launch_status();
}
template <typename T>
int sscan2_fwd_launch(const S2Fwd& p, hipStream_t st) {
    return p.bc_bf16 ? sscan2_fwd_launch2<T, bf16_t>(p, st) : sscan2_fwd_launch2<T, float>(p, st);
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
constexpr int PB = 36;             // row stride (f2) of the {d(dt*u), d dt} partial rows: 32 channels + 4 (conflict-free ds_write_b64 from the scan waves)
struct BTile {                   // what the staging waves park for one 32-step segment (double-buffered)
    f4 dd[TT / 2 * CB];          // [step / 2][slot]  {dt, dt*u} of two steps, as FTile::dd
    float g[TT * CB];            // [step][channel]   dL/dy_scan = dy * silu(z) (dy without a gate), 0 past the end
    f4 bc[TT * 8];               // [step][pair]      {B[2p], B[2p+1], C[2p], C[2p+1]}
};

// One block = 32 channels x one chunk, segments of 32 steps walked from the chunk's end to its start:
//   phase 1: recompute the segment forward from its checkpoint, keeping a_t = exp(dt_t A) and h_t of all 32 steps in registers (a lane owns
//            two states: 128 registers); phase 2: the adjoint over the same steps, last first -- no exp beyond the recompute's.
// Round 6, same principle as the forward: the scan waves (0-3) keep what is per (step, channel, state) and hand every sum over lanes to
// the LDS array:
//   * d(dt*u) and d dt (sums over the 16 states of a channel): the lane's two-state share of both, folded ONCE over lane bit 4 with the
//     step parity (v_permlane16_swap + add: the lanes of even rows keep step 2j, odd rows step 2j+1), goes out as one ds_write_b64 per two
//     steps; the staging lane that owns (step, 4 channels) adds the four partial rows with f4 additions and finishes du / ddelta from the
//     u, dt, sigmoid and g it has kept in registers since it parked that segment (rounds 2-5: two 19-instruction butterflies per 8 steps in the
//     scan wave, four more LDS arrays for the owner lanes, and the finishing arithmetic on the dependent wave);
//   * dB, dC (sums over the block's 32 channels): v_permlane32_swap pairs dB with dC, the 16 partial slabs are folded by the staging waves.
// Staging waves (4-7): fetch + park the next segment, finish + write the previous segment's du / ddelta rows, fold + publish its dB / dC rows.
// Their row I/O is branch-free (raw buffer instructions, see the forward), and the atomics of a row past the end add 0 to a clamped
// address: hipcc counts the queue exactly, so the wait in front of a parked row is vmcnt(N younger operations), not vmcnt(0) behind the
// previous segment's float atomics (which stay counted for 600-3000 cycles).
// Two barriers per segment: A (segment start: staging buffer ready, previous segment's outputs complete) and B (between recompute and adjoint:
// the previous outputs have been drained, the adjoint may overwrite `red` / `part`).
// STATE_ONLY (first pass of the chunked plan): only the local adjoint carry q of the chunk from q = 0 (one barrier per segment).
template <typename T, typename TBC, bool STATE_ONLY, bool DET>
__global__ __launch_bounds__(512) void sscan2_bwd_kernel(const S2Bwd p) {
    typedef Row4<T> R;
    typedef Row4<TBC> RBC;
    GFE_FUZZ_INIT();
    __shared__ __attribute__((aligned(16))) BTile stg[2];
    __shared__ __attribute__((aligned(16))) float red[STATE_ONLY ? 4 : 16 * RSL];   // [wave][channel & 3] slabs of [step][16 dB | 16 dC]
    __shared__ __attribute__((aligned(16))) f2 part[STATE_ONLY ? 2 : TT * 4 * PB];  // [step][pair & 3][channel] {d(dt*u), d dt} summed over lane bit 4
    const bool staging = __builtin_amdgcn_readfirstlane(threadIdx.x) >= 256;      // wave-uniform role
    const int tid = threadIdx.x & 255, lane = tid & 63, w = tid >> 6;
    const int e0 = xcd_paired_group(blockIdx.x, gridDim.x) * CB, c = blockIdx.y + (STATE_ONLY ? 1 : 0), b = blockIdx.z;     // (the state pass runs chunks 1 .. nchunks-1: nobody folds chunk 0's carry)
    const int t0 = c * p.T, t1 = min(p.L, t0 + p.T);
    const int nrows = t1 - t0;
    const int K = (nrows + TT - 1) / TT;                                // segments of this chunk
    const size_t row0 = (size_t)b * p.L + t0;

    if (staging) {
        const int sr = tid >> 3, sc = tid & 7;
        const bool has_z = p.z != nullptr;
        const unsigned esz = sizeof(T);
        const unsigned span = (unsigned)((nrows - 1) * p.ED + CB) * esz, spanz = (unsigned)((nrows - 1) * p.ld_z + CB) * esz;
        const __amdgpu_buffer_rsrc_t rs_u = row_rsrc(STATE_ONLY ? nullptr : (const T*)p.u + row0 * p.ED + e0, span);
        const __amdgpu_buffer_rsrc_t rs_d = row_rsrc((const T*)p.delta + row0 * p.ED + e0, span);
        const __amdgpu_buffer_rsrc_t rs_g = row_rsrc((const T*)p.dy + row0 * p.ED + e0, span);
        const __amdgpu_buffer_rsrc_t rs_z = row_rsrc(has_z ? (const T*)p.z + row0 * p.ld_z + e0 : nullptr, spanz);
        const __amdgpu_buffer_rsrc_t rs_ys = row_rsrc((has_z && !STATE_ONLY) ? (const T*)p.yscan + row0 * p.ED + e0 : nullptr, span);
        const __amdgpu_buffer_rsrc_t rs_du = row_rsrc(STATE_ONLY ? nullptr : (T*)p.du + row0 * p.ED + e0, span);
        const __amdgpu_buffer_rsrc_t rs_dd = row_rsrc(STATE_ONLY ? nullptr : (T*)p.ddelta + row0 * p.ED + e0, span);
        const __amdgpu_buffer_rsrc_t rs_dz = row_rsrc((has_z && !STATE_ONLY) ? (T*)p.dz + row0 * p.ld_z + e0 : nullptr, spanz);
        const int br = (tid & 127) >> 2, bq = tid & 3;
        const bool isB = tid < 128;
        const unsigned bsz = sizeof(TBC);
        const __amdgpu_buffer_rsrc_t rs_bc = row_rsrc((const char*)(isB ? p.Bm : p.Cm) + row0 * p.ld_bc * bsz, (unsigned)((nrows - 1) * p.ld_bc + 16) * bsz);
        const __amdgpu_buffer_rsrc_t rs_pbc = row_rsrc((DET && !STATE_ONLY) ? p.part_bc + (((size_t)b * gridDim.x + blockIdx.x) * p.L + t0) * 32 : nullptr, (unsigned)nrows * 128u);
        const unsigned rowb = (unsigned)p.ED * esz, rowbz = (unsigned)p.ld_z * esz, rowbbc = (unsigned)p.ld_bc * bsz;
        const unsigned off_e = (unsigned)(sr * p.ED + 4 * sc) * esz, off_z = (unsigned)(sr * p.ld_z + 4 * sc) * esz, off_bc = (unsigned)(br * p.ld_bc + 4 * bq) * bsz;
        float sbias[4], sD[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            sbias[k] = p.dbias ? p.dbias[e0 + 4 * sc + k] : 0.f;
            sD[k] = (!STATE_ONLY && p.D) ? p.D[e0 + 4 * sc + k] : 0.f;
        }
        typename R::raw ru = R::zero(), rd = R::zero(), rz = R::zero(), rg = R::zero(), ry = R::zero();
        typename RBC::raw rbc = RBC::zero();
        f4 dDacc = f4{0.f, 0.f, 0.f, 0.f}, dbacc = f4{0.f, 0.f, 0.f, 0.f};
        struct Keep { f4 u, dt, sg, g; };                             // a parked segment's per-(t, channel) values, until its du / ddelta rows are finished
        auto fetch = [&](int k) {                                     // segment k (k < 0 / rows past the end: zeros)
            const unsigned kr = (unsigned)(k * TT);
            if (!STATE_ONLY) ru = R::ld(rs_u, off_e + kr * rowb);
            rd = R::ld(rs_d, off_e + kr * rowb);
            rg = R::ld(rs_g, off_e + kr * rowb);
            rz = R::ld(rs_z, off_z + kr * rowbz);
            if (!STATE_ONLY) ry = R::ld(rs_ys, off_e + kr * rowb);
            if (!STATE_ONLY || !isB) rbc = RBC::ld(rs_bc, off_bc + kr * rowbbc);
        };
        auto park = [&](int k, BTile& tl, Keep& S) {
            const f4 fu = R::unpack(ru), fd = R::unpack(rd), fg = R::unpack(rg), fz = R::unpack(rz), fy = R::unpack(ry);
            const bool valid = (unsigned)(k * TT + sr) < (unsigned)nrows;
            f4 gz;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float raw = fd[j] + sbias[j];
                float sg;
                float dt = softplus_nb(raw, &sg);
                asm volatile("" : "+v"(dt), "+v"(sg));
                if (!p.softplus) { dt = raw; sg = 1.f; }
                float gg = fg[j], z_ = 0.f;
                if (has_z) {
                    const float sz = sigmoidf_(fz[j]);
                    z_ = fg[j] * sz * (1.f + fz[j] * (1.f - sz)) * fy[j];      // dz = dy * d/dz [z sigmoid(z)] * (hs.C + D*u)
                    gg = fg[j] * fz[j] * sz;
                }
                S.u[j] = fu[j]; S.dt[j] = valid ? dt : 0.f; S.sg[j] = valid ? sg : 0.f; S.g[j] = valid ? gg : 0.f; gz[j] = z_;
            }
            f2* ddp = reinterpret_cast<f2*>(&tl.dd[(sr >> 1) * CB]) + (sr & 1);
#pragma unroll
            for (int j = 0; j < 4; ++j) ddp[2 * dd_slot(4 * sc + j)] = f2{S.dt[j], S.dt[j] * fu[j]};
            *reinterpret_cast<f4*>(&tl.g[sr * CB + 4 * sc]) = S.g;
            if (!STATE_ONLY) R::st(gz, rs_dz, off_z + (unsigned)(k * TT) * rowbz);        // finished here (no gate / row past the end: dropped)
            if (!STATE_ONLY || !isB) {
                const f4 v = RBC::unpack(rbc);
                float* bcp = reinterpret_cast<float*>(&tl.bc[br * 8 + 2 * bq]) + (isB ? 0 : 2);
                *reinterpret_cast<f2*>(bcp) = f2{v.x, v.y};
                *reinterpret_cast<f2*>(bcp + 4) = f2{v.z, v.w};
            }
        };
        // ---- the outputs of segment k: du / ddelta rows from the four partial rows, and the block's dB / dC rows: wave w folds the 16 slabs
        // for steps 8w .. 8w+7 with ds_read_b128, writes the sums back as rows of 32 (its own rows: no barrier, only its own lgkmcnt) and adds
        // them to memory as contiguous 64-lane atomics -- two steps x (16 dB | 16 dC) per instruction, the access shape float atomics run at
        // full rate on (DET: plain stores of the block's partial rows instead).
        auto drain = [&](int k, const Keep& S) {
            const f4* pp = reinterpret_cast<const f4*>(&part[sr * 4 * PB + 4 * sc]);
            f4 s0 = pp[0], s1 = pp[1];
#pragma unroll
            for (int q = 1; q < 4; ++q) { s0 += pp[q * (PB / 2)]; s1 += pp[q * (PB / 2) + 1]; }
            const f4 ddtu = f4{s0.x, s0.z, s1.x, s1.z}, ddtA = f4{s0.y, s0.w, s1.y, s1.w};
            f4 du, dd;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                du[j] = fmaf(ddtu[j], S.dt[j], sD[j] * S.g[j]);
                dd[j] = fmaf(ddtu[j], S.u[j], ddtA[j]) * S.sg[j];
                dDacc[j] = fmaf(S.g[j], S.u[j], dDacc[j]);
                dbacc[j] += dd[j];
            }
            const unsigned off = off_e + (unsigned)(k * TT) * rowb;
            R::st(du, rs_du, off);
            R::st(dd, rs_dd, off);
            const int ts = 8 * w + (lane >> 3), jq = (lane & 7) * 4;
            f4 acc = *reinterpret_cast<const f4*>(&red[ts * 32 + jq]);
#pragma unroll
            for (int q = 1; q < 16; ++q) acc += *reinterpret_cast<const f4*>(&red[q * RSL + ts * 32 + jq]);
            *reinterpret_cast<f4*>(&red[ts * 32 + jq]) = acc;          // slab 0, this wave's rows only
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int v = lane + 64 * i, t2 = 8 * w + (v >> 5), j = v & 31;
                const float sum = red[t2 * 32 + j];
                const int row = k * TT + t2;
                if (DET) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(sum), rs_pbc, (unsigned)(row * 32 + j) * 4u, 0, 0);
                else atomicAdd((j < 16 ? p.dBws : p.dCws) + (row0 + min(row, nrows - 1)) * p.ld_dbc + (j & 15), row < nrows ? sum : 0.f);
            }
        };

        if (STATE_ONLY) {
            Keep S;
            fetch(K - 1);
            park(K - 1, stg[0], S);
            fetch(K - 2);
            GFE_LB();
            for (int k = K - 1, i = 0; k >= 0; --k, ++i) {
                park(k - 1, stg[(i + 1) & 1], S);
                fetch(k - 2);
                GFE_LB();
            }
            return;
        }
        Keep s0, s1;
        fetch(K - 1);
        park(K - 1, stg[0], s1);
        fetch(K - 2);
        GFE_LB();                                                   // A_0
        GFE_LB();                                                   // B_0 (iteration 0: nothing to drain)
        park(K - 2, stg[1], s0);
        fetch(K - 3);
        GFE_LB();                                                   // A_1
        for (int i = 1; i < K; i += 2) {                                 // iteration i: the scan waves are on segment K-1-i
            drain(K - i, s1);
            GFE_LB();                                               // B_i
            park(K - 2 - i, stg[(i + 1) & 1], s1);
            fetch(K - 3 - i);
            GFE_LB();                                               // A_{i+1}
            if (i + 1 < K) {
                drain(K - 1 - i, s0);
                GFE_LB();
                park(K - 3 - i, stg[i & 1], s0);
                fetch(K - 4 - i);
                GFE_LB();
            }
        }
        if (K & 1) drain(0, s1); else drain(0, s0);
        // dD / dbias: this lane's sums over its rows of every segment; the 32 row lanes of a channel quad meet in LDS
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        GFE_LB();                                                   // (every wave's last fold is done with `red`)
        *reinterpret_cast<f4*>(&red[sr * CB + 4 * sc]) = dDacc;
        *reinterpret_cast<f4*>(&red[TT * CB + sr * CB + 4 * sc]) = dbacc;
        GFE_LB();
        if (tid < 2 * CB) {
            const int ch = tid & (CB - 1), which = tid >> 5;
            float s = 0.f;
#pragma unroll 8
            for (int r = 0; r < TT; ++r) s += red[which * TT * CB + r * CB + ch];
            if (DET) p.part_vec[((size_t)b * p.nchunks + c) * ((size_t)p.ED * 18) + (size_t)p.ED * (16 + which) + e0 + ch] = s;
            else {
                float* dst = which ? p.dbiasws : p.dDws;
                if (dst) atomicAdd(dst + e0 + ch, s);
            }
        }
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
    const size_t sbase = (((size_t)b * p.nchunks + c) * p.ED + e) * 16 + 2 * pr;
    f2 q = f2{0.f, 0.f};
    if (!STATE_ONLY && p.nchunks > 1) q = chunk_carry<true>(p.qstate, p.sdelta, A2, b, c, e, pr, p.nchunks, p.ED);
    f2 dAacc = f2{0.f, 0.f};
    // the segment's start state: fetched one segment ahead (wanted by the very first instruction of phase 1: an HBM round trip there
    // cost 1 300 cycles per segment)
    const float* __restrict__ pck = STATE_ONLY ? nullptr : p.ckpt + (((size_t)b * p.nseg + t0 / SEG) * p.ED + e) * 16 + 2 * pr;
    const size_t ckstride = (size_t)p.ED * 16;
    f2 hck_next = f2{0.f, 0.f};
    if (!STATE_ONLY) hck_next = *reinterpret_cast<const f2*>(pck + (size_t)(K - 1) * ckstride);
    float* redp = &red[STATE_ONLY ? 0 : (w * 4 + (lane & 3)) * RSL + (lane >> 5) * 16 + 2 * pr];
    f2* partp = &part[STATE_ONLY ? 0 : (((lane >> 4) & 1) * 4 + ((lane >> 2) & 3)) * PB + cl];
    GFE_LB();                                                       // A_0
    S2_STAMP_DECL
    for (int k = K - 1, i = 0; k >= 0; --k, ++i) {
        const BTile& tl = stg[i & 1];
        const f4* ddp = &tl.dd[dd_slot(cl)];
        const float* gp = &tl.g[cl];
        const f4* bcp = &tl.bc[pr];
        S2_STAMP(2)
        const f2 hck = hck_next;
        if (k > 0 && !STATE_ONLY) hck_next = *reinterpret_cast<const f2*>(pck + (size_t)(k - 1) * ckstride);
        S2_STAMP(7)

        // Work unit = 4 steps; as in the forward, the LDS reads of the NEXT unit are issued in front of the current unit's arithmetic
        // (two register sets X / Y by unit parity, pinned with sched_barrier).
        struct H4 { f4 dd[2]; f4 bc[4]; float g[4]; };
        auto load_u = [&](H4& H, int u, bool with_g) {
            H.dd[0] = ddp[(2 * u) * CB]; H.dd[1] = ddp[(2 * u + 1) * CB];
#pragma unroll
            for (int s = 0; s < 4; ++s) H.bc[s] = bcp[(4 * u + s) * 8];
            if (with_g) {
#pragma unroll
                for (int s = 0; s < 4; ++s) H.g[s] = gp[(4 * u + s) * CB];
            }
        };
#define SB __builtin_amdgcn_sched_barrier(0)
        if (STATE_ONLY) {
            H4 X, Y;
            auto adj_u = [&](const H4& H) {
                f2 a[4];
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const f2 x = A2 * H.dd[s >> 1][2 * (s & 1)];
                    a[s] = f2{fast_exp2(x.x), fast_exp2(x.y)};
                }
#pragma unroll
                for (int s = 3; s >= 0; --s) q = a[s] * __builtin_elementwise_fma(f2{H.bc[s].z, H.bc[s].w}, f2{H.g[s], H.g[s]}, q);
            };
            load_u(X, TT / 4 - 1, true);
#pragma unroll
            for (int u = TT / 4 - 1; u >= 0; u -= 2) {
                load_u(Y, u - 1, true); SB;
                adj_u(X); SB;
                if (u - 2 >= 0) load_u(X, u - 2, true);
                SB;
                adj_u(Y); SB;
            }
            GFE_LB();                                               // A_{i+1}
            continue;
        }

        // Both phases are software pipelines over steps, one instruction per statement, pinned with sched_barrier (see the forward): a lone
        // in-order wave pays the result latency of every operand younger than ~4 instructions, and nothing else runs on its SIMD for two
        // thirds of its life.  LDS fragments live in rotating windows (8 steps of B / C / g, 4 step pairs of {dt, dt*u}), read LDB steps ahead.
        f2 av[TT], hs[TT];
        constexpr int LDB = 4;
        f4 bcw[8], ddw[4];
        float gw[8];
        {   // ---- phase 1: a_t, h_t of the segment.  step t+LDB+2: reads   t+2: x = A dt, xb = B dt u   t+1: a = exp2(x)   t: h = a h + xb
            f2 xr[2], xbr[3];
#pragma unroll
            for (int t = 0; t < LDB + 2; ++t) {
                if ((t & 1) == 0) ddw[(t >> 1) & 3] = ddp[(t >> 1) * CB];
                bcw[t & 7] = bcp[t * 8];
            }
            SB;
#pragma unroll
            for (int t = -2; t < TT; ++t) {
                if (t + 2 < TT) { const f4 d = ddw[((t + 2) >> 1) & 3]; xr[t & 1] = A2 * (((t + 2) & 1) ? d.z : d.x); SB; }
                if (t + 1 >= 0 && t + 1 < TT) { av[t + 1].x = fast_exp2(xr[(t + 1) & 1].x); SB; }
                if (t >= 0) { hs[t] = __builtin_elementwise_fma(av[t], t == 0 ? hck : hs[t == 0 ? 0 : t - 1], xbr[(t + 3) % 3]); SB; }
                if (t + 1 >= 0 && t + 1 < TT) { av[t + 1].y = fast_exp2(xr[(t + 1) & 1].y); SB; }
                if (t + 2 < TT) {
                    const f4 d = ddw[((t + 2) >> 1) & 3], bc = bcw[(t + 2) & 7];
                    const f2 Bv = f2{bc.x, bc.y};
                    if ((t + 2) & 1) xbr[(t + 5) % 3] = pk_mul_hi(Bv, __builtin_shufflevector(d, d, 2, 3)); else xbr[(t + 5) % 3] = Bv * d.y;
                    SB;
                }
                if (t + LDB + 2 < TT && t + LDB + 2 >= LDB + 2) {
                    const int s = t + LDB + 2;
                    if ((s & 1) == 0) ddw[(s >> 1) & 3] = ddp[(s >> 1) * CB];
                    bcw[s & 7] = bcp[s * 8];
                }
                SB;
            }
        }
        // the adjoint's first LDB steps: their reads are in flight across barrier B (they are inputs: nothing the staging waves still write)
#pragma unroll
        for (int s = TT - 1; s >= TT - LDB; --s) {
            if (s & 1) ddw[(s >> 1) & 3] = ddp[(s >> 1) * CB];
            bcw[s & 7] = bcp[s * 8];
            gw[s & 7] = gp[s * CB];
        }
        SB;
        S2_STAMP(3)
        GFE_LB();                                                   // B_i: the previous segment's red / part have been drained
        SB;
        {   // ---- phase 2: the adjoint, last step first.  Stage A (step t): dh, dC, q, dB, da;  stage B (step t+1): the lane's shares of d(dt*u)
            // and d dt, dB | dC over lane bit 5 -> red, dA;  stage C (steps t+2, t+3 when t is even): fold over lane bit 4 -> part.
            f2 dh[2], dC[2], dB[2], da[2], pu[4];
            f2 sxy0, sxy1, uxy0, uxy1;
#pragma unroll
            for (int t = TT - 1; t >= -2; --t) {
                const int s = t + 1;                                       // stage B's step
                const bool A_ = t >= 0, B_ = s >= 0 && s < TT, C_ = (t + 2) < TT && ((t + 2) & 1) == 0 && t + 2 >= 0;
                const f4 bcA = bcw[t & 7], bcB = bcw[s & 7];
                const f4 dA_ = ddw[(t >> 1) & 3], dB_ = ddw[(s >> 1) & 3];
                const float dtvB = (s & 1) ? dB_.z : dB_.x;
                if (A_) { const float gt = gw[t & 7]; dh[t & 1] = __builtin_elementwise_fma(f2{bcA.z, bcA.w}, f2{gt, gt}, q); SB; }
                if (B_) { pu[s & 3].x = dh[s & 1].x * bcB.x; SB; }
                if (C_) { const auto ux = __builtin_amdgcn_permlane16_swap(__float_as_uint(pu[(t + 2) & 3].x), __float_as_uint(pu[(t + 3) & 3].x), false, false); uxy0.x = __uint_as_float(ux[0]); uxy1.x = __uint_as_float(ux[1]); SB; }
                if (A_) { dC[t & 1] = hs[t] * gw[t & 7]; SB; }
                if (B_) { const auto sx = __builtin_amdgcn_permlane32_swap(__float_as_uint(dB[s & 1].x), __float_as_uint(dC[s & 1].x), false, false); sxy0.x = __uint_as_float(sx[0]); sxy1.x = __uint_as_float(sx[1]); SB; }
                if (C_) { const auto uy = __builtin_amdgcn_permlane16_swap(__float_as_uint(pu[(t + 2) & 3].y), __float_as_uint(pu[(t + 3) & 3].y), false, false); uxy0.y = __uint_as_float(uy[0]); uxy1.y = __uint_as_float(uy[1]); SB; }
                if (A_) { q = av[t] * dh[t & 1]; SB; }
                if (B_) { pu[s & 3].x = fmaf(dh[s & 1].y, bcB.y, pu[s & 3].x); SB; }
                if (A_) {
                    if (t & 1) dB[t & 1] = pk_mul_hi(dh[t & 1], __builtin_shufflevector(dA_, dA_, 2, 3)); else dB[t & 1] = dh[t & 1] * dA_.y;
                    SB;
                }
                if (B_) { const auto sy = __builtin_amdgcn_permlane32_swap(__float_as_uint(dB[s & 1].y), __float_as_uint(dC[s & 1].y), false, false); sxy0.y = __uint_as_float(sy[0]); sxy1.y = __uint_as_float(sy[1]); SB; }
                if (A_) { da[t & 1] = q * (t == 0 ? hck : hs[t == 0 ? 0 : t - 1]); SB; }     // dL/d(dt*A) of this (t, pair) = dh * a * h_{t-1}
                if (B_) { pu[s & 3].y = da[s & 1].x * An.x; SB; }
                if (B_) { dAacc = __builtin_elementwise_fma(da[s & 1], f2{dtvB, dtvB}, dAacc); SB; }
                if (C_) { partp[(t + 2) * 4 * PB] = uxy0 + uxy1; SB; }
                if (B_) { *reinterpret_cast<f2*>(redp + s * 32) = sxy0 + sxy1; SB; }
                if (B_) { pu[s & 3].y = fmaf(da[s & 1].y, An.y, pu[s & 3].y); SB; }
                if (t - LDB >= 0) {
                    const int r = t - LDB;
                    if (r & 1) ddw[(r >> 1) & 3] = ddp[(r >> 1) * CB];
                    bcw[r & 7] = bcp[r * 8];
                    gw[r & 7] = gp[r * CB];
                }
                SB;
            }
        }
#undef SB
        S2_STAMP(4)
        GFE_LB();                                                   // A_{i+1}
        S2_STAMP(5)
    }
    S2_STAMP_FLUSH(16)
    if (STATE_ONLY) {
        *reinterpret_cast<f2*>(p.qstate + sbase) = q;
        return;
    }
    GFE_LB(); GFE_LB();                                        // (the staging waves' dD / dbias sums)
    if (p.a_log) dAacc *= An;                                            // d/dA_log = dA * dA/dA_log = dA * A
    if (DET) {                                                           // this (sample, chunk)'s row of partials: plain stores, summed in order later
        float* pv = p.part_vec + ((size_t)b * p.nchunks + c) * ((size_t)p.ED * 18);
        *reinterpret_cast<f2*>(pv + (size_t)e * 16 + 2 * pr) = dAacc;
        return;
    }
    atomicAdd(p.dAws + (size_t)e * 16 + 2 * pr, dAacc.x);
    atomicAdd(p.dAws + (size_t)e * 16 + 2 * pr + 1, dAacc.y);
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

template <typename T, typename TBC, bool DET>
int sscan2_bwd_launch3(const S2Bwd& p, hipStream_t st) {
    const dim3 blk(512), grid((unsigned)(p.ED / CB), p.nchunks, p.B);
    // local adjoint carries of every chunk but the FIRST (the kernel numbers its chunks from 1 in this mode)
    if (p.nchunks > 1) hipLaunchKernelGGL((sscan2_bwd_kernel<T, TBC, true, false>), dim3(grid.x, grid.y - 1, grid.z), blk, 0, st, p);
    hipLaunchKernelGGL((sscan2_bwd_kernel<T, TBC, false, DET>), grid, blk, 0, st, p);
    if (DET) {
        const int nb_vec = (int)ceil_div((int64_t)p.ED * 18, 256);
        const int64_t nb_bc = ceil_div((int64_t)p.B * p.L * 32, 256);
        hipLaunchKernelGGL(sscan2_fold_kernel, dim3((unsigned)(nb_vec + nb_bc)), dim3(256), 0, st, p, nb_vec, p.ED / CB);
    }
    return gfe_launch_status();
}
template <typename T>
int sscan2_bwd_launch(const S2Bwd& p, hipStream_t st) {
    if (p.part_vec) return p.bc_bf16 ? sscan2_bwd_launch3<T, bf16_t, true>(p, st) : sscan2_bwd_launch3<T, float, true>(p, st);
    return p.bc_bf16 ? sscan2_bwd_launch3<T, bf16_t, false>(p, st) : sscan2_bwd_launch3<T, float, false>(p, st);
}

}  // namespace

extern "C" {

int gfe_sscan2_plan(int64_t B, int64_t L, int64_t ED, int chunk_req, int* T_out, int* nchunks_out) {
    GFE_REQUIRE(B > 0 && L > 0 && ED > 0 && ED % CB == 0 && T_out && nchunks_out, GFE_ERR_SHAPE);
    int64_t T = L;
    if (chunk_req > 0) {
        T = ceil_div(chunk_req, SEG) * SEG;                       // chunk starts must be checkpoint positions
    } else {
        const int64_t waves = B * (ED / 8);                       // one scan wave = 8 channels x 8 state pairs
        if (waves < 768) {                                        // cannot fill 1024 SIMDs: cut L (two passes + carry)
            const int64_t want = ceil_div(2048, waves);           // two rounds of one-block-per-CU grids (a block's LDS image is 116-133 KB): measured 3-4 % under one round
                                                                  // of twice as long chunks at B = 1 and B = 4, equal at B = 2 (profiles/r06/scan_batch_sweep.txt)
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
    GFE_REQUIRE((L + 4 * TT) * (ld_z > ED ? ld_z : ED) * 4 < ((int64_t)1 << 32) && (L + 4 * TT) * ld_bc * 4 < ((int64_t)1 << 32), GFE_ERR_SHAPE);   // 32-bit row offsets inside a chunk (buffer addressing)
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
    GFE_REQUIRE((L + 4 * TT) * (ld_z > ED ? ld_z : ED) * 4 < ((int64_t)1 << 32) && (L + 4 * TT) * ld_bc * 4 < ((int64_t)1 << 32), GFE_ERR_SHAPE);   // 32-bit row offsets inside a chunk (buffer addressing)
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
