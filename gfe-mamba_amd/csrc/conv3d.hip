// Implicit-GEMM 3-D convolution for gfx950 (bf16 MFMA 16x16x32, f32 accumulate), channels-last (NDHWC) activations.
//
// One kernel core serves every convolution of the frozen generator (pytorch3dunet/unet3d/buildingblocks.py):
//   * SingleConv 'gcr'/'gc': GroupNorm -> Conv3d k3 p1 no-bias [-> ReLU]        (buildingblocks.py:38-67, 108-115)
//   * ResNetBlock.conv1: Conv3d k1 with bias; residual add + ReLU epilogue        (buildingblocks.py:191-198, 218-229)
//   * TransposeConvUpsampling: ConvTranspose3d k3 s2 p1 no-bias, followed by the nearest resize 2n-1 -> 2n
//     (dst j <- src max(j-1,0)) and the summation join with the encoder features   (buildingblocks.py:355-358, 523-537)
// by describing a convolution as a *tap list*: out[v] = sum_taps W[tap] . x[v + off(tap)].  The transposed conv is
// 8 such lists, one per output-parity class (1/2/4/8 taps; out[2i] = w1 x[i], out[2i+1] = w2 x[i] + w0 x[i+1] per axis).
//
// GEMM view (per block): rows = 64 output channels (MFMA A operand = weights), cols = 512 output voxels (8x8x8 tile,
// MFMA B operand = activations), K = taps x Cin walked as 32-channel slabs (one 16x16x32 MFMA K-step per tap).
// 8 waves, each owns one d-plane of the tile: 4 voxel tiles x NT channel tiles of f32x4 accumulators.
//
// Staging is pure DMA: `buffer_load_dwordx4 ... lds` moves 1 KiB per wave instruction straight from global memory into
// LDS (per-lane source address, lane-linear destination) -- no VGPR staging, no ds_write, no per-element VALU work:
//   * activation halo tile [10x10x10 voxels][32 ch] bf16, double-buffered (the next (tile, group, slab) unit lands while
//     the current one is multiplied); out-of-volume voxels get an out-of-range buffer offset, which the hardware returns as
//     zeros = the conv's zero padding;
//   * weight stages [3 taps][64 ch][32] bf16, double-buffered.
// Both images use 64-B rows with XOR-swizzled 16-B chunks (applied on the SOURCE address; the LDS destination of a DMA is
// linear), so every ds_read_b128 fragment read is bank-conflict free (tools/lds_bank_model.py).
// GroupNorm never touches the loop: x_hat = s[b,c]*x + t[b,c] is folded into per-sample weights bf16(W*s_b) and a bias
// table indexed by the voxel's boundary class (which taps fall outside the volume: the reference pads x_hat with zeros
// AFTER the norm, so the shift term only counts taps that are inside) -- gfe_conv3d_fold_groupnorm below.
// Operands are swapped (weights = A) and weight rows are permuted at pack time so a lane ends up with 16 consecutive
// channels of one voxel -> contiguous 32-B stores.  Persistent blocks walk contiguous tile ranges (1 block per CU).
//
// The epilogue also produces the GroupNorm partials of what it stores (for the next layer's fold), and the 8 parity classes of a
// transposed conv can run as one launch with the class as the innermost dimension of the blocks' work list (template flag MC).
//
// (Round-1 history, measured on the 64->64 @96^3 conv: VGPR-staged v1 671 TFLOP/s; v2 with prefetch registers, swizzled LDS
// and division-free index math 730-780; ablation showed the non-MFMA instruction stream cost 2x the MFMA time -> this DMA form,
// 970-1030 TFLOP/s.  DESIGN.md 4.1 has the ablation table and the list of DMA-issue variants that were measured and dropped.)
#include <atomic>
#include <mutex>
#include "common.h"
#include "convt3d.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) short s16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((address_space(3))) void* lds_void_t;

#if defined(GFE_EXP_NOMFMA)   // timing experiment only: everything but the matrix instructions
#define GFE_MFMA(a, b, c) (c)
#else
#define GFE_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0)
#endif
#if defined(GFE_EXP_HALFLDS)
#define GFE_XSET(set) 0
#else
#define GFE_XSET(set) (set)
#endif

namespace {

// Buffer descriptor from provably wave-uniform inputs: without the readfirstlanes hipcc cannot prove uniformity and wraps
// EVERY buffer_load..lds in a waterfall loop (readfirstlane x4, v_cmp_eq_u64, s_and_saveexec, ..., s_cbranch_execnz).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t uniform_rsrc(const void* base, unsigned bytes) {
    const uint64_t a = (uint64_t)base;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a), hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
    return __builtin_amdgcn_make_buffer_rsrc((void*)(((uint64_t)hi << 32) | lo), 0, (int)__builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}
__device__ __forceinline__ int rfl(int v) { return __builtin_amdgcn_readfirstlane(v); }

// sum over the 16 lanes of a DPP row (lanes with equal lane >> 4), result in every lane: row_ror:8, row_ror:4, then quad xor 2 / xor 1
__device__ __forceinline__ float row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4e, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xb1, 0xf, 0xf, false));
    return v;
}

#if !defined(GFE_CONV_TD)
#define GFE_CONV_TD 8
#endif
constexpr int TD = GFE_CONV_TD, TH = 8, TW = 8;   // output tile (class-grid voxels)
constexpr int A_BUFS = TD == 8 ? 2 : 1;        // TD 4: two independent 4-wave blocks per CU, each with ONE tile buffer (the other block fills its stalls)
constexpr int NBLK = TD == 8 ? 256 : 512;      // persistent blocks
constexpr int NWAVES = TD;                     // one wave per output d-plane
constexpr int NTHREADS = NWAVES * 64;
constexpr int VSTRIDE = 64;                    // bytes per voxel / weight row in LDS (32 bf16)
constexpr int PD = TD + 2, PH = TH + 2, PW = TW + 2;      // fixed LDS pitches of the halo tile
constexpr int A_VOX = PD * PH * PW;            // 1000
constexpr int A_PIECES = (A_VOX + 15) / 16;    // 1-KiB DMA pieces (16 voxels each): 63
// DMA issue is split by wave so that each wave's vmcnt tracks ONE stream: waves 0-3 move weight stages (needed every stage),
// waves 4-7 move the next unit's activation tile (needed once per unit).  A single wave doing both would have to drain the
// tile (an HBM round trip) every time it waits for its weight pieces: vmcnt retires in order.
constexpr int DMA_WAVES = NWAVES / 2;
constexpr int A_PER_WAVE = (A_PIECES + DMA_WAVES - 1) / DMA_WAVES;   // 16
constexpr int A_BYTES = A_PER_WAVE * DMA_WAVES * 1024;               // 65,536 (the 64th piece is all out-of-range lanes -> zeros)
constexpr unsigned OOB = 0x80000000u;          // buffer offset beyond num_records -> the load returns zeros

struct ConvParams {
    const bf16_t* x; const bf16_t* w; const float* bias; const float* bias_tab; const bf16_t* res; bf16_t* y;
    const float* res1_x; const float* res1_w; const float* res1_b;   // residual computed on the fly from a one-channel volume: w[c] * x + b[c]
    bf16_t* pool_y;                                                    // RES1 only: also write MaxPool3d(2) of the (ReLU'd) result, (B, D/2, H/2, W/2, Cout)
    const float* out1_w; float out1_b; float* out1_y;                  // OUT1: a 1x1x1 conv Cout -> 1 of the result instead of storing the result
    long long w_batch_stride;                  // elements between per-sample weight sets (0: shared)
    int B, D, H, W, Cin, Cout, CoutPad;
    int OD, OH, OW;
    int ntaps, nslab, ngroups;
    int lo_d, lo_h, lo_w, LD, LH, LW;
    int ostride, op_d, op_h, op_w, oshift;
    int relu;
    float* stats; int stats_nblk, stats_slot0;  // GroupNorm partials of the OUTPUT: (B, stats_nblk, 2, Cout) f32, slot = slot0 + tile-in-sample
    int ntd, nth, ntw, tiles_per_block;
    int brick;                                 // single-class launches: tile order inside a sample -- 0: tw fastest; 1: bricks of 4 (w) x 4 (h) x 2 (d) tiles
    int toff[27];                              // LDS byte offset of each tap inside the halo tile
    int txor[27];                              // 32 when the tap shifts the row parity (swizzle term), else 0
    // multi-class launches (the 8 parity classes of a transposed conv in ONE launch, work item = (tile, class)): class c uses taps
    // c_tap0[c] .. c_tap0[c] + c_ntaps[c] - 1 of toff / txor, the weight set at element offset c_woff[c], output parity c_op[c]
    int ncls; int c_ntaps[8]; int c_tap0[8]; int c_op[8]; long long c_woff[8]; unsigned w_bytes;
    // dynamic tile scheduling (single-class launches): sched[0..7] = per-XCD ticket counters, sched[8] = finished blocks; zero between
    // launches (the last block to finish resets them).  NULL: every block walks its static share.
    int* sched;
    // work-item index -> (sample, tile, class) without integer divisions in the unit loop: n / d = umulhi(n, mg_d) with mg_d = 2^32 / d + 1, exact while
    // n * d < 2^32 (conv_launch checks, and passes 0 = "divide" otherwise, or for d == 1)
    unsigned mg_per, mg_nbw, mg_nbh, mg_ntw, mg_nth, mg_ntd, mg_ncls;
};

struct TilePos { int b, td, th, tw, cls; };

#if defined(GFE_EXP_STAMP)     // diagnostic build only: in-kernel cycle stamps of one block (tools/conv_stamps.py); never in the product library
__device__ unsigned long long* g_stamp_buf = nullptr;
#define GFE_STAMP(slot) do { if (stamp_on && lane == 0 && stamp_i < 4096) { g_stamp_buf[(wave * 4096 + stamp_i) * 2] = (slot); g_stamp_buf[(wave * 4096 + stamp_i) * 2 + 1] = __builtin_amdgcn_s_memtime(); ++stamp_i; } } while (0)
#else
#define GFE_STAMP(slot) do {} while (0)
#endif

template <int NT, int TPS, bool REG27, bool STATS, bool MC, bool RES1 = false, bool OUT1 = false>
__global__ __launch_bounds__(NTHREADS, 2) void conv_igemm_kernel(const ConvParams p) {
#if defined(__HIP_DEVICE_COMPILE__)       // the host pass only needs the launch stub (the body uses device-only buffer/LDS-DMA builtins)
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    constexpr int WROWS_TAP = NT * 16;                       // weight rows (output channels of a group) per tap
    constexpr int WSTAGE_ROWS = TPS * WROWS_TAP;
    constexpr int W_PIECES = (WSTAGE_ROWS + 15) / 16;        // 1-KiB pieces per stage
    constexpr int W_PER_WAVE = (W_PIECES + DMA_WAVES - 1) / DMA_WAVES;
    uint8_t* sA = smem;                                      // 2 x A_BYTES
    uint8_t* sW = smem + A_BUFS * A_BYTES;                   // 2 x W_PIECES KiB
    float* sRed = reinterpret_cast<float*>(smem + A_BUFS * A_BYTES + 2 * W_PIECES * 1024);   // [NWAVES][2][WROWS_TAP] GroupNorm partials of a tile
    float* sR1 = sRed + NWAVES * 2 * WROWS_TAP;          // RES1: [2][64] weight / bias of the 1x1x1 lift whose output is the residual

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lq = lane >> 4, lr = lane & 15;
    GFE_FUZZ_INIT();
#if defined(GFE_EXP_KPRIO)     // experiment: issue priority over the waves of OTHER kernels that share the SIMD (the head's small launches on the second stream)
    __builtin_amdgcn_s_setprio(GFE_EXP_KPRIO);
#endif
    if constexpr (RES1) { if (tid < 64) { sR1[tid] = p.res1_w[tid]; sR1[64 + tid] = p.res1_b[tid]; } }      // visible after the first stage barrier
    if constexpr (OUT1) { if (tid < 64) sR1[tid] = p.out1_w[tid]; }
    const int ntiles = p.B * p.ntd * p.nth * p.ntw * (MC ? p.ncls : 1);       // work items: (tile, class) with the class innermost
    // XCD-aware placement: blocks are dealt round-robin over the 8 XCDs (block b and b+8 share an L2), so the blocks of one XCD
    // get ADJACENT tile ranges -- halo planes shared by neighbouring ranges are then served by that XCD's own L2 instead of being
    // fetched once per XCD (measured: 2.0 GB FETCH_SIZE for a 0.9 GB input before the remap).  Speed only, never correctness.
    const int nb = gridDim.x, xq = nb >> 3, xr = nb & 7, xcd = blockIdx.x & 7, xi = blockIdx.x >> 3;
    const int vb = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + xi;       // bijective for any grid size
    // ... and INTERLEAVED inside the XCD's range (block xi of an XCD with nbx blocks takes items xi, xi + nbx, ...): at any time the
    // XCD works on ~nbx consecutive tiles, so the halo faces shared with the w- and h-neighbours are in its L2 while they are wanted
    // (a block walking its own contiguous range met its h-neighbour 12 tiles = 40 us later, long after the L2 had turned over).
    // Multi-class launches keep contiguous ranges: there the same block re-reads a tile for the next parity class.
    const int nbx = xq + (xcd < xr ? 1 : 0);                 // blocks on this XCD
    const int xcd_begin = min(ntiles, (vb - xi) * p.tiles_per_block), xcd_end = min(ntiles, (vb - xi + nbx) * p.tiles_per_block);
    const int tile_stride = MC ? 1 : nbx;
    const int tile_begin = MC ? vb * p.tiles_per_block : xcd_begin + xi;
    const int tile_end = MC ? min(ntiles, tile_begin + p.tiles_per_block) : xcd_end;
    const int upt = p.ngroups * p.nslab;                     // units per tile: (group, slab)
    const int my_tiles = tile_end > tile_begin ? (tile_end - tile_begin + tile_stride - 1) / tile_stride : 0;
    // Dynamic scheduling: the blocks of an XCD draw the tiles of the XCD's range in order from a ticket counter instead of walking fixed
    // shares.  At any time the XCD still works on ~nbx consecutive tiles (the L2 argument above), but a block that shares its CU with another
    // stream's kernels (the head of the previous batch in the two-stream step, RCCL's all-reduce in a multi-rank run) simply draws fewer
    // tiles, where a static share made the whole launch wait for it.  The ticket for tile k+1 is drawn while tile k is computed (its DMA is
    // issued one unit ahead), by thread 0, and travels through LDS behind the stage barriers.  Results do not depend on who computes a tile.
    __shared__ int s_ticket;
    const bool dyn = !MC && p.sched != nullptr && upt >= 2;
    auto finish = [&]() {                                    // the last block of the launch leaves the counters at zero for the next one
        GFE_FUZZ();
        if (dyn && tid == 0) {
            __threadfence();
            if (atomicAdd(p.sched + 8, 1) == (int)gridDim.x - 1) {
#pragma unroll
                for (int i = 0; i < 9; ++i) p.sched[i] = 0;
            }
        }
    };
    int dyn_first = 0;
    if (dyn) {
        GFE_FUZZ();
        if (tid == 0) s_ticket = atomicAdd(p.sched + xcd, 1);
        __syncthreads();
        GFE_FUZZ();
        dyn_first = xcd_begin + __builtin_amdgcn_readfirstlane(s_ticket);
        __syncthreads();                                     // (thread 0 overwrites the ticket at the top of the first unit)
        if (dyn_first >= xcd_end) { finish(); return; }
    }
    const int nunits = dyn ? 0x7fffffff : my_tiles * upt;
    if (nunits <= 0) return;
#if defined(GFE_EXP_STAMP)
    const bool stamp_on = g_stamp_buf != nullptr && blockIdx.x == 101;
    int stamp_i = 0;
#endif

    // ---- fragment read bases
    int abase[4];
#pragma unroll
    for (int xt = 0; xt < 4; ++xt) {
        const int lh = 2 * xt + (lr >> 3), lw = lr & 7;
        abase[xt] = ((wave * PH + lh) * PW + lw) * VSTRIDE + ((lq ^ ((lh & 1) << 1)) * 16);
    }
    const int wbase = lr * VSTRIDE + ((lq ^ ((lr >> 1) & 3)) * 16);

    // ---- thread-constant DMA coordinates.  Activation piece k = dw + 4*j (dw = wave % 4) moves voxels 16k..16k+15:
    // lane -> voxel 16k + lane/4, LDS chunk slot lane%4, which must hold data chunk slot ^ ((row & 1) << 1).
    const int dw = wave & (DMA_WAVES - 1);
    const bool a_wave = wave >= DMA_WAVES;
    // Per piece one thread constant: the byte offset of the lane's 16 B from the tile's first halo voxel (slab 0), or OOB for lanes
    // outside the halo box.  Tiles whose halo box lies inside the volume (58 % at 96^3) need no per-lane arithmetic at all: the tile
    // origin goes into the scalar offset of the buffer instruction.  Boundary tiles recompute the lane's coordinates.
    constexpr bool FASTA = REG27;       // the tap-list variants have no registers to spare for the second per-piece constant
    unsigned arel[FASTA ? A_PER_WAVE : 1];
    int acoord[A_PER_WAVE];          // ld | lh << 4 | lw << 8 | (chunk*16) << 12 | valid << 20   (boundary tiles)
#pragma unroll
    for (int j = 0; j < A_PER_WAVE; ++j) {
        const int k = dw + DMA_WAVES * j, v = 16 * k + (lane >> 2);
        const int ld = v / (PH * PW), rem = v - ld * (PH * PW), lh = rem / PW, lw = rem - lh * PW;
        const int c = (lane & 3) ^ ((lh & 1) << 1);
        const bool valid = k < A_PIECES && v < A_VOX && ld < p.LD && lh < p.LH && lw < p.LW;
        if constexpr (FASTA) arel[j] = valid ? (unsigned)((((ld * p.H + lh) * p.W + lw) * p.Cin) * 2 + c * 16) : OOB;
        acoord[j] = ld | (lh << 4) | (lw << 8) | ((c * 16) << 12) | ((valid ? 1 : 0) << 20);
    }
    const bool full_slabs = (p.Cin & 31) == 0;
    // weight piece k = dw + 4*j moves stage rows 16k..16k+15 (row = tap_local*WROWS_TAP + r): lane -> row 16k + lane/4
    unsigned wvoff[W_PER_WAVE];
#pragma unroll
    for (int j = 0; j < W_PER_WAVE; ++j) {
        const int k = dw + DMA_WAVES * j, R = 16 * k + (lane >> 2);
        const int tl = R / WROWS_TAP, r = R - tl * WROWS_TAP;
        const int c = (lane & 3) ^ ((r >> 1) & 3);
        wvoff[j] = (k < W_PIECES && R < WSTAGE_ROWS) ? (unsigned)((tl * p.CoutPad + r) * 64 + c * 16) : OOB;
    }

    // (block-uniform unsigned arithmetic: a 32-bit division by a kernel argument is ~25 scalar + 5 vector instructions, and the unit loop used to
    // do sixteen of them per unit on every wave, with the matrix pipe idle -- round 5)
    auto fdiv = [&](unsigned n, unsigned d, unsigned mg) -> unsigned { return mg ? __umulhi(n, mg) : n / d; };
    auto decode = [&](int t_) {
        TilePos q;
        q.cls = 0;
        unsigned t = (unsigned)t_;
        if constexpr (MC) { const unsigned tq = fdiv(t, (unsigned)p.ncls, p.mg_ncls); q.cls = (int)(t - tq * (unsigned)p.ncls); t = tq; }
        if (!MC && p.brick) {
            // 32 consecutive work items = one 4 x 4 x 2 brick of tiles: what the 32 blocks of an XCD work on at one time then shares halo faces in
            // all three directions inside the XCD's L2 (in tw-fastest order the d-neighbour of a tile is 144 items = 4.5 rounds away)
            const unsigned per = (unsigned)(p.ntd * p.nth * p.ntw);
            const unsigned bb = fdiv(t, per, p.mg_per);
            const unsigned r = t - bb * per, br = r >> 5, in = r & 31;
            const unsigned nbw = (unsigned)p.ntw >> 2, nbh = (unsigned)p.nth >> 2;
            const unsigned q1 = fdiv(br, nbw, p.mg_nbw), bw_ = br - q1 * nbw;
            const unsigned bd_ = fdiv(q1, nbh, p.mg_nbh), bh_ = q1 - bd_ * nbh;
            q.b = (int)bb;
            q.tw = (int)(bw_ * 4 + (in & 3)); q.th = (int)(bh_ * 4 + ((in >> 2) & 3)); q.td = (int)(bd_ * 2 + (in >> 4));
            return q;
        }
        const unsigned q1 = fdiv(t, (unsigned)p.ntw, p.mg_ntw), q2 = fdiv(q1, (unsigned)p.nth, p.mg_nth), q3 = fdiv(q2, (unsigned)p.ntd, p.mg_ntd);
        q.tw = (int)(t - q1 * (unsigned)p.ntw); q.th = (int)(q1 - q2 * (unsigned)p.nth); q.td = (int)(q2 - q3 * (unsigned)p.ntd); q.b = (int)q3;
        return q;
    };
    int cur_t = dyn ? dyn_first : tile_begin, nxt_t = cur_t;  // work-item indices (block-uniform)
    TilePos cur = decode(cur_t);
    TilePos nxt = cur;

    const size_t sample_elems = (size_t)p.D * p.H * p.W * p.Cin;
    const unsigned sample_bytes = (unsigned)(sample_elems * 2);
    const unsigned wset_bytes = (unsigned)((size_t)p.nslab * p.ntaps * p.CoutPad * 64);     // one sample's weights

    // ---- DMA issue
    auto a_dma = [&](const TilePos& q, int slab, int buf, int j0, int j1) {
        const __amdgpu_buffer_rsrc_t rs = uniform_rsrc(p.x + (size_t)rfl(q.b) * sample_elems, sample_bytes);
        const int d0 = rfl(q.td) * TD + p.lo_d, h0 = rfl(q.th) * TH + p.lo_h, w0 = rfl(q.tw) * TW + p.lo_w;
        const int lds_off = rfl(buf) * A_BYTES;
        slab = rfl(slab);
        const bool interior = FASTA && full_slabs && d0 >= 0 && h0 >= 0 && w0 >= 0 && d0 + p.LD <= p.D && h0 + p.LH <= p.H && w0 + p.LW <= p.W;
        if (interior) {                                                          // wave-uniform
            const unsigned soff = (unsigned)((((d0 * p.H + h0) * p.W + w0) * p.Cin + slab * 32) * 2);
#pragma unroll
            for (int j = 0; j < A_PER_WAVE; ++j)
                if (j >= j0 && j < j1)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t)(sA + lds_off + (dw + DMA_WAVES * j) * 1024), 16, arel[FASTA ? j : 0], soff, 0, 0);
        } else {
#pragma unroll
            for (int j = 0; j < A_PER_WAVE; ++j) {
                const int k = dw + DMA_WAVES * j;
                if (j >= j0 && j < j1) {                                         // wave-uniform
                    const int ac = acoord[j];
                    const int gd = d0 + (ac & 15), gh = h0 + ((ac >> 4) & 15), gw = w0 + ((ac >> 8) & 15);
                    const int cb = (ac >> 12) & 0xff;                            // chunk byte offset inside the slab
                    const bool ok = ((ac >> 20) & 1) && (unsigned)gd < (unsigned)p.D && (unsigned)gh < (unsigned)p.H &&
                                    (unsigned)gw < (unsigned)p.W && slab * 32 + (cb >> 1) < p.Cin;
                    const unsigned voff = ok ? (unsigned)((((gd * p.H + gh) * p.W + gw) * p.Cin + slab * 32) * 2 + cb) : OOB;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t)(sA + lds_off + k * 1024), 16, voff, 0, 0, 0);
                }
            }
        }
    };
    auto w_dma = [&](const TilePos& q, int group, int slab, int stage, int buf) {
        // taps beyond ntaps in the last stage read past the slab (still inside the set, or OOB -> zeros): never multiplied
        const int ntaps_q = MC ? p.c_ntaps[rfl(q.cls)] : p.ntaps;
        const __amdgpu_buffer_rsrc_t rs = MC ? uniform_rsrc(p.w + p.c_woff[rfl(q.cls)], p.w_bytes - (unsigned)(p.c_woff[rfl(q.cls)] * 2))
                                             : uniform_rsrc(p.w + (size_t)rfl(q.b) * p.w_batch_stride, wset_bytes);
        const unsigned soff = (unsigned)rfl((int)((((size_t)slab * ntaps_q + stage * TPS) * p.CoutPad + group * WROWS_TAP) * 64));
        const int lds_off = rfl(buf) * (W_PIECES * 1024);
#pragma unroll
        for (int j = 0; j < W_PER_WAVE; ++j) {
            const int k = dw + DMA_WAVES * j;
            if (k < W_PIECES)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t)(sW + lds_off + k * 1024), 16, wvoff[j], soff, 0, 0);
        }
    };

    f32x4 acc[4][NT];
    // GroupNorm partials of the stored output, see the epilogue.  64-channel tiles (NT == 4) keep ONE sum per 8 consecutive channels
    // (v_dot2c_f32_bf16 adds a packed pair straight into it): GroupNorm groups of >= 8 channels never need finer ones (the host
    // refuses the combination otherwise), it is 4 registers instead of 32 and a quarter of the epilogue's VALU work.
    constexpr bool OCT = STATS && NT == 4;
    constexpr int NSTAT = !STATS ? 1 : (OCT ? 2 : 4 * NT);
    float gs[NSTAT], gq[NSTAT];
    if constexpr (STATS) {
#pragma unroll
        for (int i = 0; i < NSTAT; ++i) { gs[i] = 0.f; gq[i] = 0.f; }
    }

#if !defined(GFE_EXP_NO_APRIO)
    // Static issue priority for the younger half of the block (waves 4-7, here also the activation-DMA waves): between the two waves of a
    // SIMD the older one wins every arbitration, so the younger half trails into each stage barrier (MI355X_MICROARCH.md, "Static priority
    // for the younger half").  One s_setprio for the whole kernel, no per-stage flips: 64->64 @96^3 1.485-1.493 -> 1.473 ms, 128->128 @48^3
    // 0.746 -> 0.731, the step 702-708 -> 713-715 volumes/s (same box, alternating runs).  Priority for the weight waves instead: slower (1.50).
    if (a_wave) __builtin_amdgcn_s_setprio(1);
#endif
#if defined(GFE_EXP_WPRIO)     // ... or for the older half (the weight-DMA waves, whose pieces every stage barrier waits for)
    if (!a_wave) __builtin_amdgcn_s_setprio(GFE_EXP_WPRIO);
#endif
    GFE_FUZZ();
    if (a_wave) a_dma(cur, 0, 0, 0, A_PER_WAVE); else w_dma(cur, 0, 0, 0, 0);
    GFE_FUZZ();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    int gstage = 0;                                   // global stage counter: weight buffer = gstage & 1

    int ut = 0, group = 0, slab = 0;                  // the unit inside its tile: ut = group * nslab + slab, carried along instead of divided out of u
    for (int u = 0; u < nunits; ++u) {
        int ut1 = ut + 1, group1 = group, slab1 = slab + 1;
        if (slab1 == p.nslab) { slab1 = 0; ++group1; }
        if (ut1 == upt) { ut1 = 0; group1 = 0; }
        bool next_unit = u + 1 < nunits;
        if (dyn) {
            // first unit of a tile: draw the ticket of the tile after it; last unit (>= one stage barrier later): read it
            GFE_FUZZ();
            if (ut == 0 && tid == 0) s_ticket = atomicAdd(p.sched + xcd, 1);
            GFE_FUZZ();
            if (ut1 == 0) {
                nxt_t = xcd_begin + __builtin_amdgcn_readfirstlane(*(volatile int*)&s_ticket);
                next_unit = nxt_t < xcd_end;
                if (next_unit) nxt = decode(nxt_t);
            }
        } else if (next_unit && ut1 == 0) { nxt_t += tile_stride; nxt = decode(nxt_t); }
        const uint8_t* aT = sA + (A_BUFS == 2 ? (u & 1) * A_BYTES : 0);
        const int ntaps_u = MC ? p.c_ntaps[cur.cls] : p.ntaps, tap0_u = MC ? p.c_tap0[cur.cls] : 0;
        const int nstage = (ntaps_u + TPS - 1) / TPS;

        if (slab == 0) {
#pragma unroll
            for (int xt = 0; xt < 4; ++xt)
#pragma unroll
                for (int ct = 0; ct < NT; ++ct) acc[xt][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
        }

        for (int s = 0; s < nstage; ++s, ++gstage) {
            // everything issued so far (this unit's tile, this stage's weights) has landed and is visible to all waves
            GFE_STAMP(1);
#if !defined(GFE_EXP_NOBAR)    // timing experiment only: no per-stage wait + barrier
            // weight waves drain their pieces every stage but the first: at s == 0 everything a wave had issued (weights of this
            // stage, the unit's tile) was drained BEFORE the previous unit's epilogue (below), so that epilogue's stores stay in
            // flight across this barrier instead of being waited for (vmcnt retires in order)
            GFE_FUZZ();
            if (!a_wave && s > 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if constexpr (A_BUFS == 1) { if (a_wave && s == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }   // the tile issued after the previous unit
            __builtin_amdgcn_s_waitcnt(0xc07f);              // lgkmcnt(0): no LDS read of the previous stage is still pending
            GFE_FUZZ();
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");                   // no LDS access is scheduled across the barrier
            GFE_FUZZ();
#endif
            GFE_STAMP(2);
            // refill the buffers nobody reads any more: next stage's weights, and (once per unit) the next unit's tile
            auto issue_dma = [&](bool do_w, bool do_a) {
            GFE_FUZZ();
#if !defined(GFE_EXP_NOW)      // timing experiment only: weights are never restaged
            if (!a_wave && do_w) {
                if (s + 1 < nstage) w_dma(cur, group, slab, s + 1, (gstage + 1) & 1);
                else if (next_unit) w_dma(nxt, group1, slab1, 0, (gstage + 1) & 1);
            }
#endif
#if !defined(GFE_EXP_NOA)      // timing experiment only: the activation tile is never restaged
            // issued as one burst in stage 0 by the four activation waves.  Measured alternatives, all slower on 64->64 @96^3 (1.58 ms):
            // 2 pieces per stage (1.93), every wave 1/8 of the tile in "its" stage (1.79), all DMA on the weight waves (1.67), block-
            // staggered burst stage (no change), and an address-arithmetic-free burst with the tile origin in the scalar offset
            // (2.11: the 64 misses then sit in front of the next stages' weight pieces in the CU's in-order vector-memory path)
            // Round 6: the burst goes out in TWO halves, behind the first tap of stages 0 and 1 (all of it in stage 0 when the unit has one stage):
            // the weight pieces of stage 2, issued at the top of stage 1, then queue behind 32 tile pieces instead of 64 in the CU's in-order
            // vector-memory path.  64->64 @96^3, alternating runs on two boxes: 1.430-1.436 -> 1.412-1.414 ms and 1.378-1.387 -> 1.360-1.365 ms
            // (profiles/r06/conv_burst_split_ab.txt; halves at stages 0 / 2 or 0 / 4, thirds, quarters: no gain or a loss).  Same pieces, same
            // images, same arithmetic: every wave still drains its vmcnt before the unit's epilogue.
            if constexpr (A_BUFS == 2) {
                if (a_wave && do_a && next_unit) {
                    constexpr int AH = A_PER_WAVE / 2;
                    if (s == 0) a_dma(nxt, slab1, (u + 1) & 1, 0, nstage >= 2 ? AH : A_PER_WAVE);
                    else if (s == 1) a_dma(nxt, slab1, (u + 1) & 1, AH, A_PER_WAVE);
                }
            }
#endif
            };
            // 27-tap path: the first tap's fragment reads go out BEFORE the DMA instructions, so their LDS round trip runs under the 3 (weight
            // waves) / 16 (activation waves, stage 0) DMA issues instead of in front of the first MFMA: 1.449-1.456 -> 1.427-1.439 ms
#if defined(GFE_EXP_DMA_FIRST)
            constexpr bool DMA_LATE = false;
#else
            constexpr bool DMA_LATE = REG27;
#endif
            if constexpr (!DMA_LATE) issue_dma(true, true);

            GFE_STAMP(3);
            const uint8_t* wb = sW + (gstage & 1) * (W_PIECES * 1024) + wbase;
            if constexpr (REG27) {
                // regular 3x3x3: stage s = (kd, kh), taps kw = 0..2 -> one base per voxel tile + immediate offsets
                const int kd = s / 3, kh = s - kd * 3;
                const int sbase = (kd * PH + kh) * PW * VSTRIDE, sx = (kh & 1) << 5;
                int ax[4];
#pragma unroll
                for (int xt = 0; xt < 4; ++xt) ax[xt] = (abase[xt] + sbase) ^ sx;
                // software pipeline over the stage's taps: the 8 fragment reads of tap t+1 are in flight while the 16 MFMAs of tap t
                // issue (two register sets), so only the first tap of a stage exposes the LDS latency
#if defined(GFE_EXP_NOPIPE_OCT)   // timing experiment only: the statistics variant without the second fragment set
                constexpr bool PIPE = !STATS;
#else
                constexpr bool PIPE = !STATS || OCT;  // per-channel GroupNorm partials (NT < 4) leave no room for the second register set
#endif
                bf16x8 xf[PIPE ? 2 : 1][4], wf[PIPE ? 2 : 1][NT];
                auto frag_load = [&](int tl, int set) {
#if defined(GFE_EXP_HALFLDS)   // timing experiment only: the activation fragments of tap 0 are reused for taps 1, 2 (a third of the reads)
                    if (tl == 0)
#endif
#pragma unroll
                    for (int xt = 0; xt < 4; ++xt) xf[set][xt] = *reinterpret_cast<const bf16x8*>(aT + ax[xt] + tl * VSTRIDE);
#pragma unroll
                    for (int ct = 0; ct < NT; ++ct) wf[set][ct] = *reinterpret_cast<const bf16x8*>(wb + (tl * WROWS_TAP + ct * 16) * VSTRIDE);
                };
#if defined(GFE_EXP_NOSTART)   // timing experiment only: no exposed first-tap fragment reads at the start of a stage
                if constexpr (PIPE) { if (gstage == 0) frag_load(0, 0); __builtin_amdgcn_sched_barrier(0); }
#else
                if constexpr (PIPE) { frag_load(0, 0); __builtin_amdgcn_sched_barrier(0); }
#endif
                // The activation waves' tile burst (stage 0) goes out after the first tap's MFMAs: the weight waves' three pieces for the next
                // stage are then in the CU's in-order vector-memory queue AHEAD of the burst's 64 HBM misses, and the burst's ~1 100 issue
                // cycles start under the partner wave's MFMAs: 64->64 @96^3 1.389-1.399 -> 1.382-1.384 ms, +stats 1.427-1.444 -> 1.418-1.427,
                // 128->128 @48^3 0.689-0.700 -> 0.683-0.699 (alternating runs, profiles/r04/conv_tile_burst_ab.txt).  After the SECOND tap it is
                // 12 % slower: the burst then runs past the stage's MFMAs and the activation waves trail into the barrier.
#if defined(GFE_EXP_A_AFTER)    // experiments: another tap, or -1 = together with the weight pieces (round 2's order)
                constexpr int A_AFTER = PIPE ? GFE_EXP_A_AFTER : -1;
#else
                constexpr int A_AFTER = PIPE ? 0 : -1;
#endif
                if constexpr (DMA_LATE) { issue_dma(true, A_AFTER < 0); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
                for (int tl = 0; tl < TPS; ++tl) {
                    const int set = PIPE ? (tl & 1) : 0;
                    if constexpr (PIPE) { if (tl + 1 < TPS) frag_load(tl + 1, (tl + 1) & 1); }
                    else frag_load(tl, 0);
                    if constexpr (PIPE) {
                        // the next tap's 8 fragment reads ride in the shadows of this tap's MFMAs: one ds_read_b128 per two MFMAs
#pragma unroll
                        for (int ct = 0; ct < NT; ++ct)
#pragma unroll
                            for (int xt = 0; xt < 4; ++xt) acc[xt][ct] = GFE_MFMA(wf[set][ct], xf[GFE_XSET(set)][xt], acc[xt][ct]);
                        if (tl + 1 < TPS) {
#pragma unroll
                            for (int i = 0; i < 2 * NT; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, 2, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        if constexpr (DMA_LATE && A_AFTER >= 0) { if (tl == A_AFTER) { issue_dma(false, true); __builtin_amdgcn_sched_barrier(0); } }
                    } else {
                        __builtin_amdgcn_sched_barrier(0);
                        __builtin_amdgcn_s_setprio(1);
#pragma unroll
                        for (int ct = 0; ct < NT; ++ct)
#pragma unroll
                            for (int xt = 0; xt < 4; ++xt) acc[xt][ct] = GFE_MFMA(wf[set][ct], xf[set][xt], acc[xt][ct]);
                        __builtin_amdgcn_s_setprio(0);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            } else {
#pragma unroll
                for (int tl = 0; tl < TPS; ++tl) {
                    const int tap = s * TPS + tl;
                    if (tap < ntaps_u) {                          // wave-uniform
                        const int toff = p.toff[tap0_u + tap], txor = p.txor[tap0_u + tap];
                        bf16x8 xf[4];
#pragma unroll
                        for (int xt = 0; xt < 4; ++xt) xf[xt] = *reinterpret_cast<const bf16x8*>(aT + ((abase[xt] + toff) ^ txor));
#pragma unroll
                        for (int ct = 0; ct < NT; ++ct) {
                            const bf16x8 wf = *reinterpret_cast<const bf16x8*>(wb + (tl * WROWS_TAP + ct * 16) * VSTRIDE);
#pragma unroll
                            for (int xt = 0; xt < 4; ++xt) acc[xt][ct] = GFE_MFMA(wf, xf[xt], acc[xt][ct]);
                        }
                    }
                }
            }
        }

        GFE_STAMP(4);
        if constexpr (A_BUFS == 1) {
            // single tile buffer: once every wave has read its last fragments the next unit's tile is fetched over it, under the epilogue
            // (and under the CU's other block); the activation waves wait for it at the next unit's first barrier
            if (next_unit) {
                __builtin_amdgcn_s_waitcnt(0xc07f);
                GFE_FUZZ();
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                GFE_FUZZ();
                if (a_wave) a_dma(nxt, slab1, 0, 0, A_PER_WAVE);
            }
            GFE_FUZZ();
            if (!a_wave) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            GFE_FUZZ();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // see the stage barrier: nothing but the epilogue's stores crosses the unit boundary
        }

#if defined(GFE_EXP_NOEPI)     // timing experiment only: results are never stored
        if (slab == p.nslab - 1 && !next_unit) {
#else
        if (slab == p.nslab - 1) {
#endif
            // ---- epilogue: (GroupNorm shift | bias), skip/residual add, ReLU, bf16 store.  Weight rows were permuted at pack
            // time (row ct*16 + 4*lq + r <-> channel lq*4*NT + 4*ct + r): this lane owns 4*NT consecutive channels.
            const int b = cur.b, d0 = cur.td * TD, h0 = cur.th * TH, w0 = cur.tw * TW;
            const int cd = d0 + wave;
            const int c0 = group * WROWS_TAP + lq * 4 * NT;
            // GroupNorm partials of what this tile stores (the next layer normalises it): per-channel sum / sum of squares
            // of the ROUNDED values, so the statistics describe exactly the tensor the consumer reads.  The per-lane sums persist
            // across the block's tiles and are reduced / stored only when the (sample, channel group) changes or the block ends.
            // 64-channel single-class launches are always stride-1 convs (the host sends anything else to the multi-class variant)
            constexpr bool FAST_OK = !MC && (!STATS || OCT) && NT == 4;
            // Fused MaxPool3d(2) of the result (the encoder's pooling, buildingblocks.py:284, of the first block: saves re-reading the
            // 906 MB tensor).  w and h neighbours of a voxel are lanes lane^1 / lane^8 of the same wave (DPP), the d neighbour is the wave
            // next door: the 2x2 maxima go through the tile buffer this unit has just finished with (the next unit's tile lands in the
            // other one), wave pairs meet there behind a barrier, the even wave writes the 4x4x(its plane pair) pooled voxels as whole
            // 128-byte rows.  ReLU'd bf16 values are non-negative, so the maximum is v_pk_max_i16 on the packed words.
            uint8_t* pool_scr = smem + (A_BUFS == 2 ? (u & 1) * A_BYTES : 0);
            bool pooling = false;
            if constexpr (RES1 && FAST_OK) {
                pooling = p.pool_y != nullptr;                                  // block-uniform
                if (pooling) {                                                   // every wave has read its last fragments of this tile
                    __builtin_amdgcn_s_waitcnt(0xc07f);
                    GFE_FUZZ();
                    __builtin_amdgcn_s_barrier();
                    GFE_FUZZ();
                    asm volatile("" ::: "memory");
                }
            }
            if constexpr (FAST_OK) {
              if (cd < p.D && c0 < p.Cout) {
                // ---- the common case (stride-1 conv, one destination per voxel): a 64-bit scalar tile base + 32-bit lane offsets, no
                // duplicate-plane loops, ReLU on the packed bf16 words (v_pk_max_i16: a negative float is a negative int16, and
                // relu(round(x)) == round(relu(x))).  ~40 % fewer VALU instructions than the general path below -- the epilogue is
                // VALU-issue-bound.
                const size_t tile_base = ((((size_t)b * p.D + d0) * p.H + h0) * p.W + w0) * p.Cout;      // scalar
                const bf16_t* res_t = p.res ? p.res + tile_base : nullptr;
                bf16_t* y_t = p.y + tile_base;
#pragma unroll
                for (int xt = 0; xt < 4; ++xt) {
                    const int ch_ = h0 + 2 * xt + (lr >> 3), cw_ = w0 + (lr & 7);
                    if (ch_ >= p.H || cw_ >= p.W) continue;
                    const unsigned o = (unsigned)(((wave * p.H + 2 * xt + (lr >> 3)) * p.W + (lr & 7)) * p.Cout + c0);   // inside the tile's box
                    uint4 rv[2];                                                 // skip / residual rows first: one L2 round trip together with the bias rows
                    if (res_t) { const uint4* rp = reinterpret_cast<const uint4*>(res_t + o); rv[0] = rp[0]; rv[1] = rp[1]; }
                    if (p.bias_tab) {
                        const int cls = (cd == 0) | ((cd == p.D - 1) << 1) | ((ch_ == 0) << 2) | ((ch_ == p.H - 1) << 3) |
                                        ((cw_ == 0) << 4) | ((cw_ == p.W - 1) << 5);
                        const float4* bt = reinterpret_cast<const float4*>(p.bias_tab + (unsigned)((b * 64 + cls) * p.CoutPad + c0));
#pragma unroll
                        for (int i = 0; i < NT; ++i) {
                            const float4 t = bt[i];
                            acc[xt][i][0] += t.x; acc[xt][i][1] += t.y; acc[xt][i][2] += t.z; acc[xt][i][3] += t.w;
                        }
                    }
                    if (p.bias) {
                        const float4* bv = reinterpret_cast<const float4*>(p.bias + c0);
#pragma unroll
                        for (int i = 0; i < NT; ++i) {
                            const float4 t = bv[i];
                            acc[xt][i][0] += t.x; acc[xt][i][1] += t.y; acc[xt][i][2] += t.z; acc[xt][i][3] += t.w;
                        }
                    }
                    if constexpr (RES1) {
                        // the residual of the generator's first block is its 1x1x1 lift of the one-channel input, r_c = w_c x + b_c: computed here
                        // from the voxel's x instead of being written (906 MB at 96^3, B=8) by one kernel and read back by this one
                        const float xval = p.res1_x[((size_t)b * p.D + cd) * p.H * p.W + (size_t)ch_ * p.W + cw_];
                        const float4* w4 = reinterpret_cast<const float4*>(sR1 + c0);
                        const float4* b4 = reinterpret_cast<const float4*>(sR1 + 64 + c0);
#pragma unroll
                        for (int i = 0; i < NT; ++i) {
                            const float4 wv = w4[i], bv = b4[i];
                            acc[xt][i][0] += fmaf(wv.x, xval, bv.x); acc[xt][i][1] += fmaf(wv.y, xval, bv.y);
                            acc[xt][i][2] += fmaf(wv.z, xval, bv.z); acc[xt][i][3] += fmaf(wv.w, xval, bv.w);
                        }
                    }
                    float o1 = 0.f;
                    uint32_t pq[8];                                               // RES1 pooling: the voxel's 16 packed channels
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const uint32_t rw[4] = {rv[h].x, rv[h].y, rv[h].z, rv[h].w};
                        uint32_t pk[4];
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            f32x4 a = acc[xt][2 * h + j];
                            if (res_t) {
                                a[0] += bf16lo_to_f32(rw[2 * j]); a[1] += bf16hi_to_f32(rw[2 * j]);
                                a[2] += bf16lo_to_f32(rw[2 * j + 1]); a[3] += bf16hi_to_f32(rw[2 * j + 1]);
                            }
                            pk[2 * j] = pack_bf16x2(a[0], a[1]); pk[2 * j + 1] = pack_bf16x2(a[2], a[3]);
                        }
                        if (p.relu) {
#pragma unroll
                            for (int j = 0; j < 4; ++j)
                                pk[j] = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2, pk[j]), s16x2{0, 0}));
                        }
                        if constexpr (OCT) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                const bf16x2 v = __builtin_bit_cast(bf16x2, pk[j]);
                                gs[h] = __builtin_amdgcn_fdot2_f32_bf16(v, __builtin_bit_cast(bf16x2, 0x3f803f80u), gs[h], false);
                                gq[h] = __builtin_amdgcn_fdot2_f32_bf16(v, v, gq[h], false);
                            }
                        }
                        if constexpr (OUT1) {
                            // the generator's final 1x1x1 conv (Cout -> 1) applied to the rounded result; the result itself is not needed
                            const float4* w4 = reinterpret_cast<const float4*>(sR1 + c0 + 8 * h);
                            const float4 wa = w4[0], wb = w4[1];
                            o1 = fmaf(bf16lo_to_f32(pk[0]), wa.x, o1); o1 = fmaf(bf16hi_to_f32(pk[0]), wa.y, o1);
                            o1 = fmaf(bf16lo_to_f32(pk[1]), wa.z, o1); o1 = fmaf(bf16hi_to_f32(pk[1]), wa.w, o1);
                            o1 = fmaf(bf16lo_to_f32(pk[2]), wb.x, o1); o1 = fmaf(bf16hi_to_f32(pk[2]), wb.y, o1);
                            o1 = fmaf(bf16lo_to_f32(pk[3]), wb.z, o1); o1 = fmaf(bf16hi_to_f32(pk[3]), wb.w, o1);
                        } else {
#if defined(GFE_EXP_NT_STORE)     // experiment: streaming (non-temporal) output stores, so that a round's 2 MB of output per XCD does not push the halo faces out of L2
                            typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
                            __builtin_nontemporal_store((u32x4_t){pk[0], pk[1], pk[2], pk[3]}, reinterpret_cast<u32x4_t*>(y_t + o) + h);
#else
                            reinterpret_cast<uint4*>(y_t + o)[h] = make_uint4(pk[0], pk[1], pk[2], pk[3]);
#endif
                        }
                        if constexpr (RES1) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) pq[4 * h + j] = pk[j];
                        }
                    }
                    if constexpr (RES1) {
                        if (pooling) {
#pragma unroll
                            for (int k = 0; k < 8; ++k) {
                                const uint32_t wn = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)pq[k], 0xb1, 0xf, 0xf, true);     // lane ^ 1: w neighbour
                                pq[k] = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2, pq[k]), __builtin_bit_cast(s16x2, wn)));
                                const uint32_t hn = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)pq[k], 0x128, 0xf, 0xf, true);    // lane ^ 8: h neighbour
                                pq[k] = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2, pq[k]), __builtin_bit_cast(s16x2, hn)));
                            }
                            if ((lr & 9) == 0) {                                  // even w, even h: [wave][xt][w / 2][lq] x 32 B
                                uint4* dst = reinterpret_cast<uint4*>(pool_scr + ((((wave * 4 + xt) * 4 + (lr >> 1)) * 4 + lq) << 5));
                                dst[0] = make_uint4(pq[0], pq[1], pq[2], pq[3]); dst[1] = make_uint4(pq[4], pq[5], pq[6], pq[7]);
                            }
                        }
                    }
                    if constexpr (OUT1) {
                        o1 += __shfl_xor(o1, 16, 64);                           // the four channel quads of the voxel (lanes 16 apart)
                        o1 += __shfl_xor(o1, 32, 64);
                        if (lq == 0) p.out1_y[((size_t)b * p.D + cd) * p.H * p.W + (size_t)ch_ * p.W + cw_] = o1 + p.out1_b;
                    }
                }
              }
            } else if (cd < p.D && c0 < p.Cout) {
#pragma unroll
                for (int xt = 0; xt < 4; ++xt) {
                    const int ch_ = h0 + 2 * xt + (lr >> 3), cw_ = w0 + (lr & 7);
                    if (ch_ >= p.H || cw_ >= p.W) continue;
                    const int opar = MC ? p.c_op[cur.cls] : (p.op_d | (p.op_h << 1) | (p.op_w << 2));
                    const int od = p.ostride * cd + (opar & 1), oh = p.ostride * ch_ + ((opar >> 1) & 1), ow = p.ostride * cw_ + ((opar >> 2) & 1);
                    // transposed conv: raw output has 2n-1 planes per axis; class-1 positions past it do not exist
                    if (p.ostride == 2 && (od > 2 * p.D - 2 || oh > 2 * p.H - 2 || ow > 2 * p.W - 2)) continue;
                    // finish the tile in place (the accumulators are dead afterwards): 4 channels at a time keeps the epilogue's
                    // temporaries at one f32x4 -- with the GroupNorm partials live a 16-wide copy spills
                    if (p.bias_tab) {
                        // boundary class: which neighbours of this voxel fall outside the volume
                        const int cls = (cd == 0) | ((cd == p.D - 1) << 1) | ((ch_ == 0) << 2) | ((ch_ == p.H - 1) << 3) |
                                        ((cw_ == 0) << 4) | ((cw_ == p.W - 1) << 5);
                        const float4* bt = reinterpret_cast<const float4*>(p.bias_tab + ((size_t)b * 64 + cls) * p.CoutPad + c0);
#pragma unroll
                        for (int i = 0; i < NT; ++i) {
                            const float4 t = bt[i];
                            acc[xt][i][0] += t.x; acc[xt][i][1] += t.y; acc[xt][i][2] += t.z; acc[xt][i][3] += t.w;
                        }
                    }
                    if (p.bias) {
#pragma unroll
                        for (int i = 0; i < NT; ++i)
#pragma unroll
                            for (int r = 0; r < 4; ++r) acc[xt][i][r] += p.bias[c0 + 4 * i + r];
                    }
                    const int nd = (p.oshift && od == 0) ? 2 : 1, nh = (p.oshift && oh == 0) ? 2 : 1, nw = (p.oshift && ow == 0) ? 2 : 1;
                    for (int zd = 0; zd < nd; ++zd)
                        for (int zh = 0; zh < nh; ++zh)
                            for (int zw = 0; zw < nw; ++zw) {
                                const int dd_ = zd ? 0 : od + p.oshift, dh_ = zh ? 0 : oh + p.oshift, dw_ = zw ? 0 : ow + p.oshift;
                                if (dd_ >= p.OD || dh_ >= p.OH || dw_ >= p.OW) continue;
                                const size_t o = ((((size_t)b * p.OD + dd_) * p.OH + dh_) * p.OW + dw_) * p.Cout + c0;
                                // 16-byte accesses: the lane's channels c0 + 8*h .. + 7 are accumulator tiles 2h and 2h + 1
                                constexpr int NV = NT >= 2 ? NT / 2 : 1;           // vectors per voxel (NT == 1: one 8-byte vector)
                                uint4 rv[NV];
                                if (p.res) {
                                    if constexpr (NT >= 2) {
                                        const uint4* rp = reinterpret_cast<const uint4*>(p.res + o);
#pragma unroll
                                        for (int h = 0; h < NV; ++h) rv[h] = rp[h];
                                    } else {
                                        const uint2 t = *reinterpret_cast<const uint2*>(p.res + o);
                                        rv[0] = make_uint4(t.x, t.y, 0, 0);
                                    }
                                }
#pragma unroll
                                for (int h = 0; h < NV; ++h) {
                                    uint32_t pk[4] = {0, 0, 0, 0};
                                    const uint32_t rw[4] = {rv[h].x, rv[h].y, rv[h].z, rv[h].w};
#pragma unroll
                                    for (int j = 0; j < (NT >= 2 ? 2 : 1); ++j) {
                                        const int i = 2 * h + j;
                                        float w0_ = acc[xt][i][0], w1_ = acc[xt][i][1], w2_ = acc[xt][i][2], w3_ = acc[xt][i][3];
                                        if (p.res) {
                                            w0_ += bf16lo_to_f32(rw[2 * j]); w1_ += bf16hi_to_f32(rw[2 * j]);
                                            w2_ += bf16lo_to_f32(rw[2 * j + 1]); w3_ += bf16hi_to_f32(rw[2 * j + 1]);
                                        }
                                        if (p.relu) { w0_ = fmaxf(w0_, 0.f); w1_ = fmaxf(w1_, 0.f); w2_ = fmaxf(w2_, 0.f); w3_ = fmaxf(w3_, 0.f); }
                                        pk[2 * j] = pack_bf16x2(w0_, w1_); pk[2 * j + 1] = pack_bf16x2(w2_, w3_);
                                        if constexpr (STATS && !OCT) {
                                            const float r0 = bf16lo_to_f32(pk[2 * j]), r1 = bf16hi_to_f32(pk[2 * j]);
                                            const float r2 = bf16lo_to_f32(pk[2 * j + 1]), r3 = bf16hi_to_f32(pk[2 * j + 1]);
                                            gs[4 * i] += r0; gs[4 * i + 1] += r1; gs[4 * i + 2] += r2; gs[4 * i + 3] += r3;
                                            gq[4 * i] = fmaf(r0, r0, gq[4 * i]); gq[4 * i + 1] = fmaf(r1, r1, gq[4 * i + 1]);
                                            gq[4 * i + 2] = fmaf(r2, r2, gq[4 * i + 2]); gq[4 * i + 3] = fmaf(r3, r3, gq[4 * i + 3]);
                                        }
                                    }
                                    if constexpr (OCT) {
#pragma unroll
                                        for (int j = 0; j < 4; ++j) {
                                            const bf16x2 v = __builtin_bit_cast(bf16x2, pk[j]);
                                            gs[h] = __builtin_amdgcn_fdot2_f32_bf16(v, __builtin_bit_cast(bf16x2, 0x3f803f80u), gs[h], false);
                                            gq[h] = __builtin_amdgcn_fdot2_f32_bf16(v, v, gq[h], false);
                                        }
                                    }
                                    if constexpr (NT >= 2) reinterpret_cast<uint4*>(p.y + o)[h] = make_uint4(pk[0], pk[1], pk[2], pk[3]);
                                    else *reinterpret_cast<uint2*>(p.y + o) = make_uint2(pk[0], pk[1]);
                                }
                            }
                }
            }
            if constexpr (RES1 && FAST_OK) {
                if (pooling) {
                    __builtin_amdgcn_s_waitcnt(0xc07f);
                    GFE_FUZZ();
                    __builtin_amdgcn_s_barrier();
                    asm volatile("" ::: "memory");
                    GFE_FUZZ();
                    if (!(wave & 1)) {                                           // lane -> (h pair xt, w pair, channel quad) of plane pair wave / 2
                        const int xt = lane >> 4, wq = (lane >> 2) & 3, q4 = lane & 3;
                        const uint4* a = reinterpret_cast<const uint4*>(pool_scr + ((((wave * 4 + xt) * 4 + wq) * 4 + q4) << 5));
                        const uint4* c = reinterpret_cast<const uint4*>(pool_scr + (((((wave + 1) * 4 + xt) * 4 + wq) * 4 + q4) << 5));
                        const uint4 a0 = a[0], a1 = a[1], c0_ = c[0], c1_ = c[1];
                        auto mx = [](uint32_t x, uint32_t y) { return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2, x), __builtin_bit_cast(s16x2, y))); };
                        const int pd_ = (d0 >> 1) + (wave >> 1), ph_ = (h0 >> 1) + xt, pw_ = (w0 >> 1) + wq;
                        if (pd_ < (p.D >> 1) && ph_ < (p.H >> 1) && pw_ < (p.W >> 1)) {
                            uint4* dst = reinterpret_cast<uint4*>(p.pool_y + ((((size_t)b * (p.D >> 1) + pd_) * (p.H >> 1) + ph_) * (p.W >> 1) + pw_) * p.Cout + q4 * 16);
                            dst[0] = make_uint4(mx(a0.x, c0_.x), mx(a0.y, c0_.y), mx(a0.z, c0_.z), mx(a0.w, c0_.w));
                            dst[1] = make_uint4(mx(a1.x, c1_.x), mx(a1.y, c1_.y), mx(a1.z, c1_.z), mx(a1.w, c1_.w));
                        }
                    }
                }
            }
            if constexpr (STATS) {
                // flush when the next epilogue belongs to another TILE (or channel group), or there is none: block-uniform.  One slot per
                // tile of the sample (the parity classes of a transposed conv are the innermost work-list dimension: one slot for all 8),
                // so that the grouping of the partial sums -- and with it every rounding of the statistics -- is a function of the
                // sample alone: a volume comes out bit-identical whatever batch it rides in (round 2 flushed per run of a block's tiles,
                // which depends on B: batch 8 differed from batch 1 by 1e-2 of the maximum after twelve layers of flipped bf16 roundings).
                // Multi-class launches (MC) flush per (tile, class): a block's item range may end inside a tile's classes, so a slot per
                // tile would be written by two blocks (with <= 256 items every class was its own block and seven eighths of the
                // sums were lost); rounding the ranges to whole tiles instead cost the parallelism of small launches (27 blocks at batch 1).
                const bool flush = !next_unit || p.ngroups > 1 || nxt.b != cur.b || nxt.td != cur.td || nxt.th != cur.th || nxt.tw != cur.tw ||
                                   (MC && nxt.cls != cur.cls);
                if (flush) {
                    // 16 voxel lanes of a DPP row -> one value; 8 d-plane waves -> LDS; one plain store per (slot, channel, stat)
#pragma unroll
                    for (int i = 0; i < NSTAT; ++i) { gs[i] = row16_sum(gs[i]); gq[i] = row16_sum(gq[i]); }
                    if (lr == 0) {
                        float4* r0 = reinterpret_cast<float4*>(sRed + (wave * 2) * WROWS_TAP + lq * 4 * NT);
                        float4* r1 = reinterpret_cast<float4*>(sRed + (wave * 2 + 1) * WROWS_TAP + lq * 4 * NT);
#pragma unroll
                        for (int i = 0; i < NT; ++i) {
                            if constexpr (OCT) {        // the 8-channel sum goes to the first channel of the octet, zeros to the other seven
                                r0[i] = make_float4((i & 1) ? 0.f : gs[i >> 1], 0.f, 0.f, 0.f);
                                r1[i] = make_float4((i & 1) ? 0.f : gq[i >> 1], 0.f, 0.f, 0.f);
                            } else {
                                r0[i] = make_float4(gs[4 * i], gs[4 * i + 1], gs[4 * i + 2], gs[4 * i + 3]);
                                r1[i] = make_float4(gq[4 * i], gq[4 * i + 1], gq[4 * i + 2], gq[4 * i + 3]);
                            }
                        }
                    }
#pragma unroll
                    for (int i = 0; i < NSTAT; ++i) { gs[i] = 0.f; gq[i] = 0.f; }
                    GFE_FUZZ();
                    __syncthreads();
                    GFE_FUZZ();
                    if (tid < 2 * WROWS_TAP) {
                        float t = 0.f;
#pragma unroll
                        for (int wv_ = 0; wv_ < NWAVES; ++wv_) t += sRed[wv_ * 2 * WROWS_TAP + tid];
                        const int st = tid / WROWS_TAP, c = group * WROWS_TAP + (tid - st * WROWS_TAP);
                        // slot = tile index inside the sample (x class for the multi-group transposed conv, which flushes per class)
                        const int tix = (cur.td * p.nth + cur.th) * p.ntw + cur.tw;
                        const int slot = p.stats_slot0 + (MC ? tix * p.ncls + cur.cls : tix);
                        if (c < p.Cout) p.stats[(((size_t)b * p.stats_nblk + slot) * 2 + st) * p.Cout + c] = t;
                    }
                }
            }
        }
        GFE_STAMP(5);
        GFE_FUZZ();
        if (!next_unit) break;
        if (ut1 == 0) {
            if (dyn) { cur_t = nxt_t; cur = nxt; }
            else { cur_t += tile_stride; cur = decode(min(cur_t, ntiles - 1)); }
        }
        ut = ut1; group = group1; slab = slab1;
    }
    finish();
#endif
}

// ---- GroupNorm folding -------------------------------------------------------------------------------------------
// w32: [nslab][ntaps][CoutPad][32] f32 (packed, row-permuted).  Per sample b:
//   wout[b][slab][tap][row][k] = bf16(w32[slab][tap][row][k] * scale[b][slab*32 + k])
//   T[b][tap][row]             = sum_{slab,k} w32[slab][tap][row][k] * shift[b][slab*32 + k]
__global__ __launch_bounds__(256) void fold_scale_kernel(const float* __restrict__ w32, const float* __restrict__ scale,
                                                         const float* __restrict__ shift, bf16_t* __restrict__ wout, float* __restrict__ T,
                                                         int nslab, int ntaps, int CoutPad, int Cin) {
    // one thread per 8 consecutive k of one (slab, tap, row): coalesced 32-B reads / 16-B writes; T holds one partial per (sample, slab,
    // tap, row), summed over the slabs in a fixed order by fold_bias_kernel: no atomics, no zero fill, bit-reproducible bias tables
    const int b = blockIdx.y;
    const int64_t nrows = (int64_t)nslab * ntaps * CoutPad;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;        // (slab*ntaps*CoutPad + tap*CoutPad + row) * 4 + chunk
    if (i >= nrows * 4) return;
    const int64_t rowi = i >> 2;
    const int chunk = (int)(i & 3);
    const int sl = (int)(rowi / ((int64_t)ntaps * CoutPad));
    const int64_t tr = rowi - (int64_t)sl * ntaps * CoutPad;           // tap*CoutPad + row
    const float4 w0 = *reinterpret_cast<const float4*>(w32 + rowi * 32 + chunk * 8);
    const float4 w1 = *reinterpret_cast<const float4*>(w32 + rowi * 32 + chunk * 8 + 4);
    const float w[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
    float o[8], acc = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int c = sl * 32 + chunk * 8 + k;
        const float sc = c < Cin ? scale[(size_t)b * Cin + c] : 0.f, t = c < Cin ? shift[(size_t)b * Cin + c] : 0.f;
        o[k] = w[k] * sc;
        acc = fmaf(w[k], t, acc);
    }
    if (wout)                                                          // (NULL: only the bias fold is wanted -- the caller builds its weights itself)
        *reinterpret_cast<uint4*>(wout + ((size_t)b * nrows + rowi) * 32 + chunk * 8) =
            make_uint4(pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]), pack_bf16x2(o[4], o[5]), pack_bf16x2(o[6], o[7]));
    acc += __shfl_xor(acc, 1, 64);                                     // the 4 chunks of a row sit in adjacent lanes
    acc += __shfl_xor(acc, 2, 64);
    if (chunk == 0) T[((size_t)b * nslab + sl) * ntaps * CoutPad + tr] = acc;
}

// bias_tab[b][cls][channel] = sum over the taps that stay inside the volume for boundary class cls of T[b][tap][row(channel)]
// Block = 64 packed rows x 4 class quarters: the <= 27 tap values of a row (summed over the slabs in a fixed order) are formed ONCE, by the
// four quarters together, and parked in LDS; every thread then builds 16 of the row's 64 classes from them.  (Round 2 ran one 64-thread
// block per sample with 27 x nslab dependent loads and 64 x 27 adds per thread: 53 us a launch, nine launches per step on the critical
// stream; same sums in the same order here, so the tables are bit-identical.)
__global__ __launch_bounds__(1024) void fold_bias_kernel(const float* __restrict__ T, float* __restrict__ tab, const int8_t* __restrict__ taps,
                                                         int ntaps, int CoutPad, int NT, int nslab) {
    // 64 packed rows x 16 thread groups (round 5: 4 groups of 256 threads took 12-13 us a launch, nine launches per generator pass on the stream that bounds
    // the step -- each thread waited for 7 x nslab loads and then walked 16 classes x 27 taps; now <= 2 x nslab loads and 4 classes per thread)
    __shared__ float stv[27][64];
    __shared__ int sneed[27];
    const int b = blockIdx.y, rl = threadIdx.x & 63, q = threadIdx.x >> 6;       // q: 0..15
    const int rho = blockIdx.x * 64 + rl;                    // packed row
    const bool live = rho < CoutPad;
    {
        constexpr int MAXS = 8;
        const bool fits = nslab <= MAXS;
        float v[2][MAXS];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int t = q + 16 * i;
#pragma unroll
            for (int sl = 0; sl < MAXS; ++sl)
                v[i][sl] = (fits && t < 27 && t < ntaps && live && sl < nslab) ? T[(((size_t)b * nslab + sl) * ntaps + t) * CoutPad + rho] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int t = q + 16 * i;
            if (t >= 27) continue;
            float a = 0.f;                                   // the slabs in slab order: the same sums as before
            if (fits) {
#pragma unroll
                for (int sl = 0; sl < MAXS; ++sl) { if (sl < nslab) a += v[i][sl]; }
            } else if (t < ntaps && live) {
                for (int sl = 0; sl < nslab; ++sl) a += T[(((size_t)b * nslab + sl) * ntaps + t) * CoutPad + rho];
            }
            stv[t][rl] = a;
        }
    }
    if (threadIdx.x < 27) {
        // class bits (1 d == 0, 2 d == D-1, 4 h == 0, 8 h == H-1, 16 w == 0, 32 w == W-1) of the faces the tap steps over: the tap leaves
        // the volume for boundary class cls iff cls & need
        const int t = threadIdx.x;
        int need = 63;
        if (t < ntaps) {
            const int dd = taps[3 * t], dh = taps[3 * t + 1], dw = taps[3 * t + 2];
            need = (dd < 0 ? 1 : dd > 0 ? 2 : 0) | (dh < 0 ? 4 : dh > 0 ? 8 : 0) | (dw < 0 ? 16 : dw > 0 ? 32 : 0);
        }
        sneed[t] = need;
    }
    __syncthreads();
    if (!live) return;
    float tv[27];
    int need[27];
#pragma unroll
    for (int t = 0; t < 27; ++t) { tv[t] = stv[t][rl]; need[t] = sneed[t]; }
    // packed row -> channel (same permutation as the weight packing)
    const int g = rho / (NT * 16), r = rho - g * (NT * 16);
    const int ch = g * NT * 16 + ((r >> 2) & 3) * 4 * NT + (r >> 4) * 4 + (r & 3);
#pragma unroll
    for (int ci = 0; ci < 4; ++ci) {
        const int cls = 4 * q + ci;
        float acc = 0.f;
#pragma unroll
        for (int t = 0; t < 27; ++t) {                       // the taps in tap order: the same sums as before
            if (t < ntaps && !(cls & need[t])) acc += tv[t];
        }
        tab[((size_t)b * 64 + cls) * CoutPad + ch] = acc;
    }
}

// Ticket counters of the dynamic tile scheduler: 16-int slots, zeroed once (a kernel leaves its slot at zero).  EAGER launches take the
// next slot of a 256-slot ring: two launches meet on a slot only 256 conv launches apart.  A CAPTURED launch keeps its slot for the life of
// the graph, so it draws from a pool of its own that is never recycled (a replayed graph next to an eager generator on another
// stream could have shared a ring slot and split one ticket stream); when that pool is used up, captured launches fall back to static tile
// shares (sched = NULL).  Allocated outside stream capture only; GFE_CONV_STATIC=1 turns the scheduler off.
}  // namespace
static int g_conv_reserved_cus = -1;
int conv_reserved_cus() {
    if (g_conv_reserved_cus < 0) { const char* e = getenv("GFE_CONV_RESERVE_CUS"); g_conv_reserved_cus = e ? atoi(e) : 0; }
    return g_conv_reserved_cus < 0 ? 0 : (g_conv_reserved_cus > 128 ? 128 : g_conv_reserved_cus);
}
extern "C" int gfe_conv_reserve_cus(int n) { const int old = conv_reserved_cus(); g_conv_reserved_cus = n < 0 ? 0 : n; return old; }
int* conv_sched_slot(hipStream_t st) {
    constexpr unsigned RING = 256, CAPTURED = 4096;
    static int* pool = nullptr;                          // [RING eager slots | CAPTURED dedicated slots] x 16 ints
    static std::mutex mu;
    static std::atomic<unsigned> seq{0}, cap_seq{0};
    static const bool off = getenv("GFE_CONV_STATIC") != nullptr && getenv("GFE_CONV_STATIC")[0] == '1';
    if (off) return nullptr;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) != hipSuccess) return nullptr;
    const bool capturing = cs != hipStreamCaptureStatusNone;
    if (!pool) {
        std::lock_guard<std::mutex> lock(mu);
        if (!pool) {
            if (capturing) return nullptr;               // (allocation and the one-time fill are not capturable)
            int* q = nullptr;
            const size_t bytes = (size_t)(RING + CAPTURED) * 16 * sizeof(int);
            if (hipMalloc((void**)&q, bytes) != hipSuccess) return nullptr;
            if (hipMemset(q, 0, bytes) != hipSuccess) { (void)hipFree(q); return nullptr; }
            pool = q;
        }
    }
    if (capturing) {
        const unsigned i = cap_seq.fetch_add(1);
        return i < CAPTURED ? pool + 16 * (size_t)(RING + i) : nullptr;
    }
    return pool + 16 * (size_t)(seq.fetch_add(1) % RING);
}
namespace {

template <int NT, int TPS, bool REG27, bool STATS, bool MC = false, bool RES1 = false, bool OUT1 = false>
int conv_launch(const ConvParams& p, hipStream_t st) {
    constexpr int W_PIECES = (TPS * NT * 16 + 15) / 16;
    const size_t lds = A_BUFS * (size_t)A_BYTES + 2 * (size_t)W_PIECES * 1024 + (size_t)NWAVES * 2 * NT * 16 * sizeof(float) + ((RES1 || OUT1) ? 512 : 0);
    const int64_t tiles = (int64_t)p.B * p.ntd * p.nth * p.ntw * (MC ? p.ncls : 1);
    if (tiles > 0x7fffffff) return GFE_ERR_SHAPE;
    // persistent blocks: one resident block per CU x 256 CUs, each walking a contiguous tile range
    ConvParams q = p;
    q.sched = MC ? nullptr : conv_sched_slot(st);
    static const bool brick_on = !(getenv("GFE_CONV_BRICK") != nullptr && getenv("GFE_CONV_BRICK")[0] == '0');     // GFE_CONV_BRICK=0: tw-fastest order
    q.brick = (!MC && brick_on && p.ntw % 4 == 0 && p.nth % 4 == 0 && p.ntd % 2 == 0) ? 1 : 0;
    {
        const uint64_t per = (uint64_t)p.ntd * p.nth * p.ntw;
        auto magic = [&](uint64_t d) -> unsigned { return (d >= 2 && (uint64_t)tiles * d < (1ull << 32)) ? (unsigned)((1ull << 32) / d) + 1u : 0u; };
        q.mg_per = magic(per); q.mg_nbw = magic((uint64_t)p.ntw >> 2); q.mg_nbh = magic((uint64_t)p.nth >> 2);
        q.mg_ntw = magic((uint64_t)p.ntw); q.mg_nth = magic((uint64_t)p.nth); q.mg_ntd = magic((uint64_t)p.ntd); q.mg_ncls = MC ? magic((uint64_t)p.ncls) : 0u;
    }
    // the grid need not cover every CU: gfe_conv_reserve_cus(n) leaves n CUs to the kernels of another stream (with or without the ticket
    // scheduler: on a CU-masked stream -- gfe_stream_create_cu_mask -- blocks beyond the mask's CUs would only queue up for a second round)
    const int nblk = NBLK == 256 ? NBLK - conv_reserved_cus() : NBLK;
    q.tiles_per_block = (int)ceil_div(tiles, nblk);
    const dim3 grid((unsigned)ceil_div(tiles, q.tiles_per_block));
    static bool attr_set = false;
    if (!attr_set) { (void)hipFuncSetAttribute((const void*)conv_igemm_kernel<NT, TPS, REG27, STATS, MC, RES1, OUT1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr_set = true; }
    hipLaunchKernelGGL((conv_igemm_kernel<NT, TPS, REG27, STATS, MC, RES1, OUT1>), grid, dim3(NTHREADS), lds, st, q);
    return gfe_launch_status();
}

}  // namespace

extern "C" {

int gfe_conv3d_tiles(int64_t D, int64_t H, int64_t W) { return (int)(ceil_div(D, TD) * ceil_div(H, TH) * ceil_div(W, TW)); }

int gfe_conv3d_stat_slots(int64_t B, int64_t D, int64_t H, int64_t W, int64_t Cout) {
    (void)B; (void)Cout;
    return gfe_conv3d_tiles(D, H, W);                                        // one slot per tile of a sample, whatever the batch
}

int gfe_convt3d_stat_slots(int64_t B, int64_t D, int64_t H, int64_t W, int64_t Cout) {
    (void)B;
    const int64_t tps = gfe_conv3d_tiles(D, H, W);
    if (gfe_conv3d_cout_pad(Cout) > 64) return (int)(tps * 8);                // (tile, class) slots
    const int slots_res = convt_resident_tiles(D, H, W);                      // the resident kernel: one slot per tile of ITS tile grid;
    return (int)(tps * 8) > slots_res ? (int)(tps * 8) : slots_res;           // the streamed multi-class kernel: (tile, class) slots (the dispatch depends on Cin)
}

#if defined(GFE_EXP_STAMP)
int gfe_debug_set_stamp_buffer(void* buf) { return hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_buf), &buf, sizeof(buf)) == hipSuccess ? 0 : -4; }
#endif

int gfe_conv3d_cout_pad(int64_t Cout) {
    if (Cout <= 16) return 16;
    if (Cout <= 32) return 32;
    return (int)(ceil_div(Cout, 64) * 64);
}

int gfe_conv3d_fold_groupnorm(const float* w_packed_f32, const float* gn_scale, const float* gn_shift, void* w_out, float* T_ws,
                              float* bias_tab, const int8_t* tap_offsets_dev, int64_t B, int64_t Cin, int64_t Cout, int ntaps, void* stream) {
    GFE_REQUIRE(w_packed_f32 && gn_scale && gn_shift && T_ws && bias_tab && tap_offsets_dev, GFE_ERR_NULL);      // (w_out may be NULL: bias table only)
    GFE_REQUIRE(B > 0 && B <= 65535 && Cin > 0 && Cout > 0 && ntaps >= 1 && ntaps <= 27, GFE_ERR_SHAPE);
    const int cp = gfe_conv3d_cout_pad(Cout), nslab = (int)ceil_div(Cin, 32);
    const int NT = (cp < 64 ? cp : 64) / 16;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(fold_scale_kernel, dim3((unsigned)ceil_div((int64_t)nslab * ntaps * cp * 4, 256), (unsigned)B), dim3(256), 0, st,
                       w_packed_f32, gn_scale, gn_shift, (bf16_t*)w_out, T_ws, nslab, ntaps, cp, (int)Cin);
    hipLaunchKernelGGL(fold_bias_kernel, dim3((unsigned)ceil_div((int64_t)cp, 64), (unsigned)B), dim3(1024), 0, st,
                       T_ws, bias_tab, tap_offsets_dev, ntaps, cp, NT, nslab);
    return gfe_launch_status();
}

static int conv_igemm_impl(const void* x, const void* w_packed, int64_t w_batch_stride, const float* bias, const float* bias_tab,
                           const void* res, void* y,
                           int64_t B, int64_t D, int64_t H, int64_t W, int64_t Cin, int64_t Cout,
                           int64_t OD, int64_t OH, int64_t OW,
                           int ntaps, const int8_t* tap_offsets /* host, ntaps x 3 (dd,dh,dw) */,
                           int ostride, int op_d, int op_h, int op_w, int oshift, int relu,
                           float* stats_ws, int64_t stats_nblk, int64_t stats_slot0,
                           const float* res1_x, const float* res1_w, const float* res1_b,
                           const float* out1_w, float out1_b, float* out1_y, void* pool_y, void* stream) {
    GFE_REQUIRE(x && w_packed && (y || out1_y) && tap_offsets, GFE_ERR_NULL);
    GFE_REQUIRE(B > 0 && D > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, GFE_ERR_SHAPE);
    GFE_REQUIRE(Cin % 8 == 0 && Cout % 8 == 0 && ntaps >= 1 && ntaps <= 27, GFE_ERR_SHAPE);
    GFE_REQUIRE(ostride == 1 || ostride == 2, GFE_ERR_SHAPE);
    GFE_REQUIRE(D * H * W * Cin * 2 < 0x7fffffffLL, GFE_ERR_SHAPE);             // one sample addressable by a 32-bit buffer offset
    ConvParams p;
    p.x = (const bf16_t*)x; p.w = (const bf16_t*)w_packed; p.bias = bias; p.bias_tab = bias_tab;
    p.res = (const bf16_t*)res; p.y = (bf16_t*)y; p.w_batch_stride = w_batch_stride;
    p.res1_x = res1_x; p.res1_w = res1_w; p.res1_b = res1_b;
    p.out1_w = out1_w; p.out1_b = out1_b; p.out1_y = out1_y; p.pool_y = (bf16_t*)pool_y;
    p.B = (int)B; p.D = (int)D; p.H = (int)H; p.W = (int)W; p.Cin = (int)Cin; p.Cout = (int)Cout;
    p.CoutPad = gfe_conv3d_cout_pad(Cout);
    p.OD = (int)OD; p.OH = (int)OH; p.OW = (int)OW;
    p.ntaps = ntaps; p.nslab = (int)ceil_div(Cin, 32);
    const int NT = (p.CoutPad < 64 ? p.CoutPad : 64) / 16;
    p.ngroups = p.CoutPad / (NT * 16);
    GFE_REQUIRE((int64_t)p.nslab * ntaps * p.CoutPad * 64 < 0x7fffffffLL, GFE_ERR_SHAPE);
    int lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
    for (int t = 0; t < ntaps; ++t) {
        const int8_t* o = tap_offsets + 3 * t;
        for (int a = 0; a < 3; ++a) {
            GFE_REQUIRE(o[a] >= -1 && o[a] <= 1, GFE_ERR_SHAPE);
            if (o[a] < lo[a]) lo[a] = o[a];
            if (o[a] > hi[a]) hi[a] = o[a];
        }
    }
    p.lo_d = lo[0]; p.lo_h = lo[1]; p.lo_w = lo[2];
    p.LD = TD + hi[0] - lo[0]; p.LH = TH + hi[1] - lo[1]; p.LW = TW + hi[2] - lo[2];
    for (int t = 0; t < ntaps; ++t) {
        const int8_t* o = tap_offsets + 3 * t;
        p.toff[t] = (((o[0] - lo[0]) * PH + (o[1] - lo[1])) * PW + (o[2] - lo[2])) * VSTRIDE;
        p.txor[t] = ((o[1] - lo[1]) & 1) ? 32 : 0;
    }
    p.ostride = ostride; p.op_d = op_d; p.op_h = op_h; p.op_w = op_w; p.oshift = oshift; p.relu = relu;
    if (ostride == 1) {
        GFE_REQUIRE(OD == D && OH == H && OW == W && oshift == 0 && op_d == 0 && op_h == 0 && op_w == 0, GFE_ERR_SHAPE);
    } else {
        GFE_REQUIRE(OD == 2 * D - 1 + oshift && OH == 2 * H - 1 + oshift && OW == 2 * W - 1 + oshift, GFE_ERR_SHAPE);
    }
    p.ntd = (int)ceil_div(D, TD); p.nth = (int)ceil_div(H, TH); p.ntw = (int)ceil_div(W, TW);
    p.stats = stats_ws; p.stats_nblk = (int)stats_nblk; p.stats_slot0 = (int)stats_slot0;
    p.ncls = 1; p.w_bytes = 0;
    for (int c = 0; c < 8; ++c) { p.c_ntaps[c] = 0; p.c_tap0[c] = 0; p.c_op[c] = 0; p.c_woff[c] = 0; }
    if (stats_ws) GFE_REQUIRE(stats_slot0 >= 0 && stats_slot0 + gfe_conv3d_stat_slots(B, D, H, W, Cout) <= stats_nblk && stats_nblk <= 0x7fffffff, GFE_ERR_SHAPE);
    hipStream_t st = (hipStream_t)stream;
    // regular 3x3x3 tap list in canonical order -> immediate-offset fast path
    bool reg27 = ntaps == 27 && ostride == 1;
    for (int t = 0; t < ntaps && reg27; ++t)
        reg27 = tap_offsets[3 * t] == t / 9 - 1 && tap_offsets[3 * t + 1] == (t / 3) % 3 - 1 && tap_offsets[3 * t + 2] == t % 3 - 1;
    if (NT == 4 && ostride != 1) {
        // one parity class of a transposed conv with >= 64 output channels, launched on its own: the single-class 64-channel kernels
        // carry only the stride-1 epilogue, so this goes through the multi-class variant with one class
        GFE_REQUIRE(!stats_ws || Cout % 64 == 0, GFE_ERR_SHAPE);
        p.ncls = 1; p.c_ntaps[0] = ntaps; p.c_tap0[0] = 0; p.c_woff[0] = 0; p.c_op[0] = op_d | (op_h << 1) | (op_w << 2);
        p.w_bytes = (unsigned)((size_t)p.nslab * ntaps * p.CoutPad * 64);
        return stats_ws ? conv_launch<4, 3, false, true, true>(p, st) : conv_launch<4, 3, false, false, true>(p, st);
    }
    if (stats_ws) {
        GFE_REQUIRE(NT < 4 || Cout % 64 == 0, GFE_ERR_SHAPE);         // 64-channel tiles write 8-channel sums (see OCT in the kernel)
        if (NT == 1) return conv_launch<1, 3, false, true>(p, st);
        if (NT == 2) return conv_launch<2, 3, false, true>(p, st);
        return reg27 ? conv_launch<4, 3, true, true>(p, st) : conv_launch<4, 3, false, true>(p, st);
    }
    if (NT == 1) return conv_launch<1, 3, false, false>(p, st);
    if (NT == 2) return conv_launch<2, 3, false, false>(p, st);
    if (out1_y) {
        // final 1x1x1 conv fused: only the plain 27-tap 64-channel variant carries that epilogue
        GFE_REQUIRE(out1_w && !res1_x && reg27 && NT == 4 && Cout == 64, GFE_ERR_SHAPE);
        return conv_launch<4, 3, true, false, false, false, true>(p, st);
    }
    if (res1_x) {
        // residual from a one-channel volume: only the plain 27-tap 64-channel variant carries that epilogue
        GFE_REQUIRE(res1_w && res1_b && !res && reg27 && NT == 4 && Cout == 64, GFE_ERR_SHAPE);
        GFE_REQUIRE(!pool_y || (relu && D % 2 == 0 && H % 2 == 0 && W % 2 == 0), GFE_ERR_SHAPE);      // pooled maxima are taken on ReLU'd (non-negative) bf16 bits
        return conv_launch<4, 3, true, false, false, true>(p, st);
    }
    return reg27 ? conv_launch<4, 3, true, false>(p, st) : conv_launch<4, 3, false, false>(p, st);
}

int gfe_conv3d_igemm(const void* x, const void* w_packed, int64_t w_batch_stride, const float* bias, const float* bias_tab,
                     const void* res, void* y,
                     int64_t B, int64_t D, int64_t H, int64_t W, int64_t Cin, int64_t Cout,
                     int64_t OD, int64_t OH, int64_t OW,
                     int ntaps, const int8_t* tap_offsets,
                     int ostride, int op_d, int op_h, int op_w, int oshift, int relu,
                     float* stats_ws, int64_t stats_nblk, int64_t stats_slot0, void* stream) {
    return conv_igemm_impl(x, w_packed, w_batch_stride, bias, bias_tab, res, y, B, D, H, W, Cin, Cout, OD, OH, OW, ntaps, tap_offsets,
                           ostride, op_d, op_h, op_w, oshift, relu, stats_ws, stats_nblk, stats_slot0, nullptr, nullptr, nullptr, nullptr, 0.f, nullptr, nullptr, stream);
}

int gfe_conv3d_k3_lift_residual(const void* x, const void* w_packed, int64_t w_batch_stride, const float* bias_tab, void* y,
                                int64_t B, int64_t D, int64_t H, int64_t W, int64_t Cin, int64_t Cout, const int8_t* tap_offsets, int relu,
                                const float* vol, const float* lift_w, const float* lift_b, void* pool_out, void* stream) {
    GFE_REQUIRE(vol && lift_w && lift_b, GFE_ERR_NULL);
    return conv_igemm_impl(x, w_packed, w_batch_stride, nullptr, bias_tab, nullptr, y, B, D, H, W, Cin, Cout, D, H, W, 27, tap_offsets,
                           1, 0, 0, 0, 0, relu, nullptr, 0, 0, vol, lift_w, lift_b, nullptr, 0.f, nullptr, pool_out, stream);
}

int gfe_conv3d_k3_out1(const void* x, const void* w_packed, int64_t w_batch_stride, const float* bias_tab, const void* res,
                       int64_t B, int64_t D, int64_t H, int64_t W, int64_t Cin, int64_t Cout, const int8_t* tap_offsets, int relu,
                       const float* out_w, float out_b, float* out_y, void* stream) {
    GFE_REQUIRE(out_w && out_y, GFE_ERR_NULL);
    return conv_igemm_impl(x, w_packed, w_batch_stride, nullptr, bias_tab, res, nullptr, B, D, H, W, Cin, Cout, D, H, W, 27, tap_offsets,
                           1, 0, 0, 0, 0, relu, nullptr, 0, 0, nullptr, nullptr, nullptr, out_w, out_b, out_y, nullptr, stream);
}

int gfe_convt3d_k3s2_fused(const void* x, const void* w_packed, const int64_t* cls_woff, const int* cls_ntaps, const int8_t* cls_parity,
                           const int8_t* tap_offsets, int64_t w_elems, const void* res, void* y,
                           int64_t B, int64_t D, int64_t H, int64_t W, int64_t Cin, int64_t Cout, int64_t OD, int64_t OH, int64_t OW,
                           int oshift, float* stats_ws, int64_t stats_nblk, void* stream) {
    GFE_REQUIRE(x && w_packed && cls_woff && cls_ntaps && cls_parity && tap_offsets && y, GFE_ERR_NULL);
    GFE_REQUIRE(B > 0 && D > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && Cin % 8 == 0 && Cout % 8 == 0, GFE_ERR_SHAPE);
    GFE_REQUIRE(gfe_conv3d_cout_pad(Cout) >= 64, GFE_ERR_SHAPE);              // the multi-class variant is built for 64-channel groups
    GFE_REQUIRE(!stats_ws || Cout % 64 == 0, GFE_ERR_SHAPE);                  // ... whose partials are 8-channel sums
    GFE_REQUIRE(oshift == 0 || oshift == 1, GFE_ERR_SHAPE);
    GFE_REQUIRE(OD == 2 * D - 1 + oshift && OH == 2 * H - 1 + oshift && OW == 2 * W - 1 + oshift, GFE_ERR_SHAPE);
    GFE_REQUIRE(D * H * W * Cin * 2 < 0x7fffffffLL && w_elems * 2 < 0xffffffffLL, GFE_ERR_SHAPE);
    if (convt_resident_fits(Cin, Cout) && !getenv("GFE_CONVT_STREAMED")) {
        // all input channels of a 4x8x8 tile fit in LDS: stage them once for the 8 classes (convt3d.hip)
        ConvTParams q;
        q.x = (const uint16_t*)x; q.w = (const uint16_t*)w_packed; q.res = (const uint16_t*)res; q.y = (uint16_t*)y; q.stats = stats_ws;
        q.B = (int)B; q.D = (int)D; q.H = (int)H; q.W = (int)W; q.Cin = (int)Cin; q.Cout = (int)Cout; q.CoutPad = 64;
        q.OD = (int)OD; q.OH = (int)OH; q.OW = (int)OW; q.nslab = (int)(Cin / 32); q.oshift = oshift; q.stats_nblk = (int)stats_nblk;
        q.w_bytes = (unsigned)(w_elems * 2);
        int total = 0;
        for (int c = 0; c < 8; ++c) {
            const int nt = cls_ntaps[c];
            GFE_REQUIRE((nt == 1 || nt == 2 || nt == 4 || nt == 8) && cls_woff[c] >= 0 && cls_woff[c] % 8 == 0, GFE_ERR_SHAPE);
            q.c_ntaps[c] = nt; q.c_lg[c] = nt == 1 ? 0 : nt == 2 ? 1 : nt == 4 ? 2 : 3; q.c_tap0[c] = total; q.c_woff[c] = cls_woff[c];
            q.c_op[c] = (cls_parity[3 * c] & 1) | ((cls_parity[3 * c + 1] & 1) << 1) | ((cls_parity[3 * c + 2] & 1) << 2);
            total += nt;
        }
        GFE_REQUIRE(total == 27, GFE_ERR_SHAPE);
        for (int t = 0; t < total; ++t) {
            const int8_t* o = tap_offsets + 3 * t;
            GFE_REQUIRE(o[0] >= 0 && o[0] <= 1 && o[1] >= 0 && o[1] <= 1 && o[2] >= 0 && o[2] <= 1, GFE_ERR_SHAPE);
            q.toff[t] = ((o[0] * PH + o[1]) * PW + o[2]) * VSTRIDE;
            q.txor[t] = (o[1] & 1) ? 32 : 0;
        }
        if (stats_ws) GFE_REQUIRE(gfe_convt3d_stat_slots(B, D, H, W, Cout) <= stats_nblk && stats_nblk <= 0x7fffffff, GFE_ERR_SHAPE);
        return convt_resident_launch(q, (hipStream_t)stream);
    }
    ConvParams p;
    p.x = (const bf16_t*)x; p.w = (const bf16_t*)w_packed; p.bias = nullptr; p.bias_tab = nullptr;
    p.res = (const bf16_t*)res; p.y = (bf16_t*)y; p.w_batch_stride = 0;
    p.res1_x = nullptr; p.res1_w = nullptr; p.res1_b = nullptr; p.out1_w = nullptr; p.out1_b = 0.f; p.out1_y = nullptr; p.pool_y = nullptr;
    p.B = (int)B; p.D = (int)D; p.H = (int)H; p.W = (int)W; p.Cin = (int)Cin; p.Cout = (int)Cout;
    p.CoutPad = gfe_conv3d_cout_pad(Cout);
    p.OD = (int)OD; p.OH = (int)OH; p.OW = (int)OW;
    p.nslab = (int)ceil_div(Cin, 32); p.ngroups = p.CoutPad / 64;
    p.ncls = 8; p.w_bytes = (unsigned)(w_elems * 2);
    int total = 0;
    for (int c = 0; c < 8; ++c) {
        GFE_REQUIRE(cls_ntaps[c] >= 1 && cls_ntaps[c] <= 8 && cls_woff[c] >= 0 && cls_woff[c] % 8 == 0, GFE_ERR_SHAPE);
        p.c_ntaps[c] = cls_ntaps[c]; p.c_tap0[c] = total; p.c_woff[c] = cls_woff[c];
        p.c_op[c] = (cls_parity[3 * c] & 1) | ((cls_parity[3 * c + 1] & 1) << 1) | ((cls_parity[3 * c + 2] & 1) << 2);
        total += cls_ntaps[c];
    }
    GFE_REQUIRE(total <= 27, GFE_ERR_SHAPE);
    p.ntaps = total;
    // one halo box for all classes: offsets are 0 / +1 per axis
    for (int t = 0; t < total; ++t) {
        const int8_t* o = tap_offsets + 3 * t;
        GFE_REQUIRE(o[0] >= 0 && o[0] <= 1 && o[1] >= 0 && o[1] <= 1 && o[2] >= 0 && o[2] <= 1, GFE_ERR_SHAPE);
        p.toff[t] = ((o[0] * PH + o[1]) * PW + o[2]) * VSTRIDE;
        p.txor[t] = (o[1] & 1) ? 32 : 0;
    }
    p.lo_d = p.lo_h = p.lo_w = 0; p.LD = TD + 1; p.LH = TH + 1; p.LW = TW + 1;
    p.ostride = 2; p.op_d = p.op_h = p.op_w = 0; p.oshift = oshift; p.relu = 0;
    p.ntd = (int)ceil_div(D, TD); p.nth = (int)ceil_div(H, TH); p.ntw = (int)ceil_div(W, TW);
    p.stats = stats_ws; p.stats_nblk = (int)stats_nblk; p.stats_slot0 = 0;
    if (stats_ws) GFE_REQUIRE(gfe_convt3d_stat_slots(B, D, H, W, Cout) <= stats_nblk && stats_nblk <= 0x7fffffff, GFE_ERR_SHAPE);
    hipStream_t st = (hipStream_t)stream;
    return stats_ws ? conv_launch<4, 3, false, true, true>(p, st) : conv_launch<4, 3, false, false, true>(p, st);
}

}  // extern "C"
