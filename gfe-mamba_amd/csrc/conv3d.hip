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
// GEMM view (per block): rows = output channels (MFMA A operand = weights), cols = 256 output voxels (4x8x8 tile,
// MFMA B operand = activations), K = taps x Cin, walked as 32-channel slabs (one 16x16x32 MFMA K-step per tap).
//   LDS: activation halo tile [6x10x10 voxels][32 ch] bf16, 64-B voxel stride with XOR-swizzled 16-B chunks (fragment
//        reads are bank-conflict free, see below), GroupNorm scale/shift applied while staging,
//        zero padding written as exact zeros (the reference pads AFTER the norm);
//        weights [taps-per-stage][Cout tile][32] bf16 double-buffered, next stage prefetched to registers under the MFMAs.
//   Each wave owns one d-plane of the tile: 4 voxel tiles x NT channel tiles of f32x4 accumulators.
//   Operands are swapped (weights = A) so a lane ends up with 4 consecutive channels of one voxel -> 8-B stores.
// 2 blocks/CU (<= 80 KB LDS each) so one block's staging overlaps the other's MFMAs.
#include "common.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#if defined(GFE_EXP_NOMFMA)   // timing experiment only: everything but the matrix instructions
#define GFE_MFMA(a, b, c) (c)
#else
#define GFE_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0)
#endif
#if defined(GFE_EXP_NOBAR)    // timing experiment only: no per-stage barrier
#define GFE_STAGE_BARRIER() do {} while (0)
#else
#define GFE_STAGE_BARRIER() __syncthreads()
#endif

namespace {

constexpr int TD = 4, TH = 8, TW = 8;          // output tile (class-grid voxels)
constexpr int VSTRIDE = 64;                    // bytes per voxel / weight row in LDS (32 bf16, no padding; XOR-swizzled chunks)
constexpr int PH = TH + 2, PW = TW + 2;        // fixed LDS pitches of the halo tile (rows x columns), whatever the tap extent
constexpr int A_MAX_VOX = (TD + 2) * PH * PW;
constexpr int A_BYTES = A_MAX_VOX * VSTRIDE;   // 38,400
// Bank-conflict-free ds_read_b128 fragments (MI355X: 64 banks x 4 B, 16-lane groups {0-3,12-15,20-27},...; modelled in
// tools/lds_bank_model.py): the 16-B chunk c of voxel (row lh, col lw) lives at chunk slot c ^ ((lh & 1) << 1); the chunk c
// of weight row r lives at slot c ^ ((r >> 1) & 3).  Both fragment reads then take the ideal 4 LDS cycles (was 12 / 8 with
// an 80-B padded stride).

struct ConvTap { int8_t dd, dh, dw; uint8_t pad; };

struct ConvParams {
    const bf16_t* x; const bf16_t* w; const float* gn_scale; const float* gn_shift; const float* bias;
    const bf16_t* res; bf16_t* y;
    int B, D, H, W, Cin, Cout, CoutPad;
    int OD, OH, OW;
    int ntaps, nslab;
    int lo_d, lo_h, lo_w, LD, LH, LW;
    int ostride, op_d, op_h, op_w, oshift;
    int relu;
    int ntd, nth, ntw, tiles_per_block;
    int LHW, NV;
    int toff[27];                             // LDS byte offset of each tap inside the halo tile
    int txor[27];                             // 32 when the tap shifts the row parity (swizzle term), else 0
};

struct TilePos { int b, td, th, tw; };

template <int NT, int TPS, bool PIPE, bool REG27>
__global__ __launch_bounds__(256, 2) void conv_igemm_kernel(const ConvParams p) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint8_t* sA = smem;
    uint8_t* sW = smem + A_BYTES;
    constexpr int WROWS_TAP = NT * 16;                       // weight rows (output channels) per tap
    constexpr int WSTAGE_BYTES = TPS * WROWS_TAP * VSTRIDE;
    constexpr int WJ = (WROWS_TAP * 4 + 255) / 256;          // 16-B chunks per thread per tap
    constexpr int PLANES = TD + 2;                           // d-planes of the halo tile
    constexpr int AITEMS = 2 * PLANES;                       // 16-B chunks per thread per activation tile
    constexpr int PLANE_BYTES = PH * PW * VSTRIDE;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lq = lane >> 4, lr = lane & 15;
    const int cg = blockIdx.y;
    const int ntiles = p.B * p.ntd * p.nth * p.ntw;

    // a block walks a contiguous range of tiles (adjacent in w, then h, d, b: halo lines of the next tile are L2-warm)
    const int tile_begin = blockIdx.x * p.tiles_per_block;
    const int tile_end = min(ntiles, tile_begin + p.tiles_per_block);
    const int nunits = (tile_end - tile_begin) * p.nslab;   // unit = (tile, 32-channel slab)
    if (nunits <= 0) return;

    f32x4 acc[4][NT];

    // per-lane LDS byte offsets of the 4 voxel tiles (tap offset added per tap) and of the weight fragment
    int abase[4];
#pragma unroll
    for (int xt = 0; xt < 4; ++xt) {
        const int lh = 2 * xt + (lr >> 3), lw = lr & 7;
        abase[xt] = ((wave * PH + lh) * PW + lw) * VSTRIDE + ((lq ^ ((lh & 1) << 1)) * 16);
    }
    const int wbase = lr * VSTRIDE + ((lq ^ ((lr >> 1) & 3)) * 16);
    const int nstage = (p.ntaps + TPS - 1) / TPS;

    // ---- thread-constant staging coordinates: each thread moves one 16-B chunk of two voxels of every d-plane.
    // No per-item index arithmetic is left in the loop (constant divisors only, evaluated once).
    const int chunk = tid & 3;                               // 8-channel chunk inside the 32-channel slab
    const int v0 = tid >> 2, v1 = 64 + (tid >> 2);           // voxel index inside a PH x PW plane
    const int lh0 = v0 / PW, lw0 = v0 - lh0 * PW, lh1 = v1 / PW, lw1 = v1 - lh1 * PW;
    const bool in0 = lh0 < p.LH && lw0 < p.LW;
    const bool in1 = v1 < PH * PW && lh1 < p.LH && lw1 < p.LW;
    const int lds0 = (lh0 * PW + lw0) * VSTRIDE + ((chunk ^ ((lh0 & 1) << 1)) * 16);
    const int lds1 = (lh1 * PW + lw1) * VSTRIDE + ((chunk ^ ((lh1 & 1) << 1)) * 16);

    TilePos cur;
    {
        int t = tile_begin;
        cur.tw = t % p.ntw; t /= p.ntw;
        cur.th = t % p.nth; t /= p.nth;
        cur.td = t % p.ntd; cur.b = t / p.ntd;
    }
    TilePos nxt = cur;                                       // position of the unit being prefetched
    auto advance = [&](TilePos& q) {
        if (++q.tw == p.ntw) { q.tw = 0; if (++q.th == p.nth) { q.th = 0; if (++q.td == p.ntd) { q.td = 0; ++q.b; } } }
    };

    // ---- register staging of one unit's activation halo tile (raw bf16)
    uint4 areg[AITEMS];
    unsigned amask = 0;
    auto aload = [&](const TilePos& q, int slab, int pl0, int pl1) {
        const int d0 = q.td * TD + p.lo_d, h0 = q.th * TH + p.lo_h, w0 = q.tw * TW + p.lo_w;
        const int cch = slab * 32 + chunk * 8;
        const bool ch_ok = cch < p.Cin;
        const int gh0 = h0 + lh0, gw0 = w0 + lw0, gh1 = h0 + lh1, gw1 = w0 + lw1;
        const bool ok0 = in0 && ch_ok && (unsigned)gh0 < (unsigned)p.H && (unsigned)gw0 < (unsigned)p.W;
        const bool ok1 = in1 && ch_ok && (unsigned)gh1 < (unsigned)p.H && (unsigned)gw1 < (unsigned)p.W;
        const int off0 = (gh0 * p.W + gw0) * p.Cin + cch, off1 = (gh1 * p.W + gw1) * p.Cin + cch;
        const size_t plane = (size_t)p.H * p.W * p.Cin;
        const bf16_t* base = p.x + (size_t)q.b * p.D * plane;
        if (pl0 == 0) amask = 0;
#pragma unroll
        for (int pl = 0; pl < PLANES; ++pl) {
            if (pl < pl0 || pl >= pl1) continue;
            const int gd = d0 + pl;
            const bool dok = pl < p.LD && (unsigned)gd < (unsigned)p.D;           // wave-uniform
            areg[2 * pl] = make_uint4(0, 0, 0, 0);
            areg[2 * pl + 1] = make_uint4(0, 0, 0, 0);
            if (dok) {
                const bf16_t* pb = base + (size_t)gd * plane;
                if (ok0) { areg[2 * pl] = *reinterpret_cast<const uint4*>(pb + off0); amask |= 1u << (2 * pl); }
                if (ok1) { areg[2 * pl + 1] = *reinterpret_cast<const uint4*>(pb + off1); amask |= 2u << (2 * pl); }
            }
        }
    };
    // GroupNorm applied while writing to LDS; out-of-volume voxels stay exact zeros (the reference pads AFTER the norm)
    auto astore = [&](const TilePos& q, int slab, int pl0, int pl1) {
        float sc[8], sh[8];
        const int cch = slab * 32 + chunk * 8;
        if (p.gn_scale && cch < p.Cin) {
            const float4* ps = reinterpret_cast<const float4*>(p.gn_scale + (size_t)q.b * p.Cin + cch);
            const float4* pt = reinterpret_cast<const float4*>(p.gn_shift + (size_t)q.b * p.Cin + cch);
            const float4 s0 = ps[0], s1 = ps[1], t0 = pt[0], t1 = pt[1];
            sc[0] = s0.x; sc[1] = s0.y; sc[2] = s0.z; sc[3] = s0.w; sc[4] = s1.x; sc[5] = s1.y; sc[6] = s1.z; sc[7] = s1.w;
            sh[0] = t0.x; sh[1] = t0.y; sh[2] = t0.z; sh[3] = t0.w; sh[4] = t1.x; sh[5] = t1.y; sh[6] = t1.z; sh[7] = t1.w;
        }
#pragma unroll
        for (int k = 0; k < AITEMS; ++k) {
            const int pl = k >> 1;
            if (pl < pl0 || pl >= pl1) continue;
            if (pl < p.LD && ((k & 1) ? in1 : in0)) {
                uint4 v = areg[k];
                if (p.gn_scale && ((amask >> k) & 1u)) {
                    v.x = pack_bf16x2(fmaf(bf16lo_to_f32(v.x), sc[0], sh[0]), fmaf(bf16hi_to_f32(v.x), sc[1], sh[1]));
                    v.y = pack_bf16x2(fmaf(bf16lo_to_f32(v.y), sc[2], sh[2]), fmaf(bf16hi_to_f32(v.y), sc[3], sh[3]));
                    v.z = pack_bf16x2(fmaf(bf16lo_to_f32(v.z), sc[4], sh[4]), fmaf(bf16hi_to_f32(v.z), sc[5], sh[5]));
                    v.w = pack_bf16x2(fmaf(bf16lo_to_f32(v.w), sc[6], sh[6]), fmaf(bf16hi_to_f32(v.w), sc[7], sh[7]));
                }
                *reinterpret_cast<uint4*>(sA + pl * PLANE_BYTES + ((k & 1) ? lds1 : lds0)) = v;
            }
        }
    };

    // ---- weight stages: global [slab][tap][CoutPad][32]; thread-constant (row, chunk) inside a tap
    constexpr bool DEEP = false;      // 3-stage register weight pipeline: correct, but 52 VGPR spills at NT=4 (615 vs 729 TFLOP/s)
    constexpr int WSETS = (REG27 && DEEP) ? 3 : 1;
    uint4 wq[WSETS][TPS * WJ];
    auto wload = [&](uint4 (&wr)[TPS * WJ], int slab, int stage) {
        const bf16_t* wslab = p.w + ((size_t)slab * p.ntaps * p.CoutPad + (size_t)cg * WROWS_TAP) * 32;
#pragma unroll
        for (int tl = 0; tl < TPS; ++tl) {
#if defined(GFE_EXP_WHOT)
            const int tap = 0 * (stage * TPS + tl);      // timing experiment only: every stage re-reads the same (hot) weights
#else
            const int tap = stage * TPS + tl;
#endif
#pragma unroll
            for (int j = 0; j < WJ; ++j) {
                const int itl = tid + j * 256;
                wr[tl * WJ + j] = make_uint4(0, 0, 0, 0);
                if (itl < WROWS_TAP * 4 && tap < p.ntaps)
                    wr[tl * WJ + j] = *reinterpret_cast<const uint4*>(wslab + ((size_t)tap * p.CoutPad + (itl >> 2)) * 32 + (itl & 3) * 8);
            }
        }
    };
    auto wstore = [&](const uint4 (&wr)[TPS * WJ], int buf) {
#if defined(GFE_EXP_NOW)
        if (buf >= 0) return;                             // timing experiment only: weights are never restaged
#endif
#pragma unroll
        for (int tl = 0; tl < TPS; ++tl)
#pragma unroll
            for (int j = 0; j < WJ; ++j) {
                const int itl = tid + j * 256, rr = itl >> 2;
                if (itl < WROWS_TAP * 4)
                    *reinterpret_cast<uint4*>(sW + buf * WSTAGE_BYTES + (tl * WROWS_TAP + rr) * VSTRIDE + (((itl & 3) ^ ((rr >> 1) & 3)) * 16)) = wr[tl * WJ + j];
            }
    };

    if (PIPE) aload(cur, 0, 0, PLANES);
    wload(wq[0], 0, 0);
    if constexpr (REG27 && DEEP) { wload(wq[1], 0, 1); wload(wq[2], 0, 2); }

    for (int u = 0; u < nunits; ++u) {
        const int slab = u % p.nslab;
        // registers hold unit u (activations + stage-0 weights); every wave passed the barrier that ended unit u-1
#if defined(GFE_EXP_NOA)
        if (u == 0)
#endif
        if (PIPE) {
            astore(cur, slab, 0, PLANES);
        } else {      // wide variant: no register budget to hold a tile across the MFMA loop -> two batches of 3 planes
            aload(cur, slab, 0, PLANES / 2); astore(cur, slab, 0, PLANES / 2);
            aload(cur, slab, PLANES / 2, PLANES); astore(cur, slab, PLANES / 2, PLANES);
        }
        wstore(wq[0], 0);
        __syncthreads();
        const bool next_unit = u + 1 < nunits;
        const int nslab_next = (slab + 1 == p.nslab) ? 0 : slab + 1;
        if constexpr (REG27 && DEEP) wload(wq[0], slab, 3);   // set 0 was just drained: stage 3 goes out now
        if (next_unit) {
            if (nslab_next == 0) advance(nxt);
#if !defined(GFE_EXP_NOA)
            if (PIPE) aload(nxt, nslab_next, 0, PLANES);     // in flight under this unit's MFMAs
#endif
        }
        if (slab == 0) {
#pragma unroll
            for (int xt = 0; xt < 4; ++xt)
#pragma unroll
                for (int ct = 0; ct < NT; ++ct) acc[xt][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
        }

        if constexpr (REG27 && DEEP) {
            // regular 3x3x3, 9 stages = (kd, kh), 3 taps kw = 0..2 each.  Weight pipeline: the loads of stage s+4 are issued
            // when stage s ends (3 stages ~ 1 us ahead of their ds_write: L2 latency under load is longer than one stage).
            for (int kd = 0; kd < 3; ++kd) {
#pragma unroll
                for (int kh = 0; kh < 3; ++kh) {
                    const int s = kd * 3 + kh;
                    if constexpr (!DEEP) {
                        if (s + 1 < 9) wload(wq[0], slab, s + 1);
                        else if (next_unit) wload(wq[0], nslab_next, 0);
                    }
                    const uint8_t* wb = sW + (s & 1) * WSTAGE_BYTES + wbase;
                    const int sbase = (kd * PH + kh) * PW * VSTRIDE, sx = (kh & 1) << 5;
                    int ax[4];
#pragma unroll
                    for (int xt = 0; xt < 4; ++xt) ax[xt] = (abase[xt] + sbase) ^ sx;
#pragma unroll
                    for (int tl = 0; tl < TPS; ++tl) {
                        bf16x8 xf[4];
#pragma unroll
                        for (int xt = 0; xt < 4; ++xt) xf[xt] = *reinterpret_cast<const bf16x8*>(sA + ax[xt] + tl * VSTRIDE);
#pragma unroll
                        for (int ct = 0; ct < NT; ++ct) {
                            const bf16x8 wf = *reinterpret_cast<const bf16x8*>(wb + (tl * WROWS_TAP + ct * 16) * VSTRIDE);
#pragma unroll
                            for (int xt = 0; xt < 4; ++xt)
                                acc[xt][ct] = GFE_MFMA(wf, xf[xt], acc[xt][ct]);
                        }
                    }
                    if constexpr (DEEP) {
                        // stage s+1 lives in set (kh+1)%3; store it (unless it is the next unit's stage 0, stored at that unit's
                        // top), then refill the set with stage s+4 (wrapping into the next unit's slab)
                        const int st1 = s + 1, st4 = s + 4;
                        if (st1 < 9) wstore(wq[(kh + 1) % WSETS], st1 & 1);
                        if (st4 < 9) wload(wq[(kh + 1) % WSETS], slab, st4);
                        else if (next_unit && st4 - 9 < 3) wload(wq[(kh + 1) % WSETS], nslab_next, st4 - 9);
                    } else {
                        if (s + 1 < 9) wstore(wq[0], (s + 1) & 1);
                    }
                    GFE_STAGE_BARRIER();
                }
            }
        } else {
            for (int s = 0; s < nstage; ++s) {
                const bool more = s + 1 < nstage;
                if (more) wload(wq[0], slab, s + 1);
                else if (next_unit) wload(wq[0], nslab_next, 0);
                const uint8_t* wb = sW + (s & 1) * WSTAGE_BYTES + wbase;
                if constexpr (REG27) {
                    // regular 3x3x3: stage s = (kd, kh), taps kw = 0..2 -> one base per voxel tile + immediate offsets
                    const int kd = s / 3, kh = s - kd * 3;
                    const int sbase = (kd * PH + kh) * PW * VSTRIDE, sx = (kh & 1) << 5;
                    int ax[4];
#pragma unroll
                    for (int xt = 0; xt < 4; ++xt) ax[xt] = (abase[xt] + sbase) ^ sx;
#pragma unroll
                    for (int tl = 0; tl < TPS; ++tl) {
                        bf16x8 xf[4];
#pragma unroll
                        for (int xt = 0; xt < 4; ++xt) xf[xt] = *reinterpret_cast<const bf16x8*>(sA + ax[xt] + tl * VSTRIDE);
#pragma unroll
                        for (int ct = 0; ct < NT; ++ct) {
                            const bf16x8 wf = *reinterpret_cast<const bf16x8*>(wb + (tl * WROWS_TAP + ct * 16) * VSTRIDE);
#pragma unroll
                            for (int xt = 0; xt < 4; ++xt)
                                acc[xt][ct] = GFE_MFMA(wf, xf[xt], acc[xt][ct]);
                        }
                    }
                } else {
#pragma unroll
                for (int tl = 0; tl < TPS; ++tl) {
                    const int tap = s * TPS + tl;
                    if (tap < p.ntaps) {                          // wave-uniform
                        const int toff = p.toff[tap], txor = p.txor[tap];     // txor: bit 5 toggles when the tap moves to an odd row
                        bf16x8 xf[4];
#pragma unroll
                        for (int xt = 0; xt < 4; ++xt) xf[xt] = *reinterpret_cast<const bf16x8*>(sA + ((abase[xt] + toff) ^ txor));
#pragma unroll
                        for (int ct = 0; ct < NT; ++ct) {
                            const bf16x8 wf = *reinterpret_cast<const bf16x8*>(wb + (tl * WROWS_TAP + ct * 16) * VSTRIDE);
#pragma unroll
                            for (int xt = 0; xt < 4; ++xt)
                                acc[xt][ct] = GFE_MFMA(wf, xf[xt], acc[xt][ct]);
                        }
                    }
                }
                }
                if (more) wstore(wq[0], (s + 1) & 1);
                GFE_STAGE_BARRIER();
            }
        }
#if defined(GFE_EXP_NOEPI)
        if (slab != p.nslab - 1 || u + 1 < nunits) { if (slab == p.nslab - 1) advance(cur); continue; }
#endif
        if (slab != p.nslab - 1) continue;

        // ---- epilogue: bias, skip/residual add, ReLU, bf16 store (lane: voxel = lr of tile xt, channels 16*ct + 4*lq + 0..3)
        const int b = cur.b, d0 = cur.td * TD, h0 = cur.th * TH, w0 = cur.tw * TW;
        const int cd = d0 + wave;
#pragma unroll
        for (int xt = 0; xt < 4; ++xt) {
            const int ch_ = h0 + 2 * xt + (lr >> 3), cw_ = w0 + (lr & 7);
            if (cd >= p.D || ch_ >= p.H || cw_ >= p.W) continue;
            const int od = p.ostride * cd + p.op_d, oh = p.ostride * ch_ + p.op_h, ow = p.ostride * cw_ + p.op_w;
            // transposed conv: raw output has 2n-1 planes per axis; class-1 positions past it do not exist
            if (p.ostride == 2 && (od > 2 * p.D - 2 || oh > 2 * p.H - 2 || ow > 2 * p.W - 2)) continue;
            const int nd = (p.oshift && od == 0) ? 2 : 1, nh = (p.oshift && oh == 0) ? 2 : 1, nw = (p.oshift && ow == 0) ? 2 : 1;
            for (int zd = 0; zd < nd; ++zd)
                for (int zh = 0; zh < nh; ++zh)
                    for (int zw = 0; zw < nw; ++zw) {
                        const int dd_ = zd ? 0 : od + p.oshift, dh_ = zh ? 0 : oh + p.oshift, dw_ = zw ? 0 : ow + p.oshift;
                        if (dd_ >= p.OD || dh_ >= p.OH || dw_ >= p.OW) continue;
                        const size_t vox = (((size_t)b * p.OD + dd_) * p.OH + dh_) * p.OW + dw_;
                        // weight rows were permuted at pack time (row ct*16 + 4*lq + r  <->  channel lq*4*NT + 4*ct + r), so this lane
                        // owns 4*NT CONSECUTIVE output channels of the voxel: one 8*NT-byte contiguous piece (full 128-B rows per voxel)
                        const int c0 = cg * NT * 16 + lq * 4 * NT;
                        if (c0 >= p.Cout) continue;
                        float v[4 * NT];
#pragma unroll
                        for (int ct = 0; ct < NT; ++ct)
#pragma unroll
                            for (int r = 0; r < 4; ++r) v[4 * ct + r] = acc[xt][ct][r];
                        if (p.bias) {
#pragma unroll
                            for (int i = 0; i < 4 * NT; ++i) v[i] += p.bias[c0 + i];
                        }
                        const size_t o = vox * p.Cout + c0;
                        if (p.res) {
                            const uint2* rp = reinterpret_cast<const uint2*>(p.res + o);
#pragma unroll
                            for (int i = 0; i < NT; ++i) {
                                const uint2 rv = rp[i];
                                v[4 * i] += bf16lo_to_f32(rv.x); v[4 * i + 1] += bf16hi_to_f32(rv.x);
                                v[4 * i + 2] += bf16lo_to_f32(rv.y); v[4 * i + 3] += bf16hi_to_f32(rv.y);
                            }
                        }
                        if (p.relu) {
#pragma unroll
                            for (int i = 0; i < 4 * NT; ++i) v[i] = fmaxf(v[i], 0.f);
                        }
                        uint2* yp = reinterpret_cast<uint2*>(p.y + o);
#pragma unroll
                        for (int i = 0; i < NT; ++i) yp[i] = make_uint2(pack_bf16x2(v[4 * i], v[4 * i + 1]), pack_bf16x2(v[4 * i + 2], v[4 * i + 3]));
                    }
        }
        advance(cur);
    }
}

template <int NT, int TPS, bool PIPE, bool REG27>
int conv_launch(const ConvParams& p, hipStream_t st) {
    const size_t lds = A_BYTES + 2 * (size_t)TPS * NT * 16 * VSTRIDE;
    const int64_t tiles = (int64_t)p.B * p.ntd * p.nth * p.ntw;
    if (tiles > 0x7fffffff) return GFE_ERR_SHAPE;
    const int groups = p.CoutPad / (NT * 16);
    // persistent blocks: ~2 resident blocks per CU x 256 CUs, each walking a contiguous tile range
    ConvParams q = p;
    q.tiles_per_block = (int)ceil_div(tiles * groups, 512);
    const dim3 grid((unsigned)ceil_div(tiles, q.tiles_per_block), (unsigned)groups);
    static bool attr_set = false;
    if (!attr_set) { (void)hipFuncSetAttribute((const void*)conv_igemm_kernel<NT, TPS, PIPE, REG27>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr_set = true; }
    hipLaunchKernelGGL((conv_igemm_kernel<NT, TPS, PIPE, REG27>), grid, dim3(256), lds, st, q);
    return gfe_launch_status();
}

}  // namespace

extern "C" {

int gfe_conv3d_cout_pad(int64_t Cout) {
    if (Cout <= 16) return 16;
    if (Cout <= 32) return 32;
    if (Cout <= 64) return 64;
    return (int)(ceil_div(Cout, 128) * 128);
}

int gfe_conv3d_igemm(const void* x, const void* w_packed, const float* gn_scale, const float* gn_shift, const float* bias,
                     const void* res, void* y,
                     int64_t B, int64_t D, int64_t H, int64_t W, int64_t Cin, int64_t Cout,
                     int64_t OD, int64_t OH, int64_t OW,
                     int ntaps, const int8_t* tap_offsets /* host, ntaps x 3 (dd,dh,dw) */,
                     int ostride, int op_d, int op_h, int op_w, int oshift, int relu, void* stream) {
    GFE_REQUIRE(x && w_packed && y && tap_offsets, GFE_ERR_NULL);
    GFE_REQUIRE((gn_scale == nullptr) == (gn_shift == nullptr), GFE_ERR_NULL);
    GFE_REQUIRE(B > 0 && D > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, GFE_ERR_SHAPE);
    GFE_REQUIRE(Cin % 8 == 0 && Cout % 8 == 0 && ntaps >= 1 && ntaps <= 27, GFE_ERR_SHAPE);
    GFE_REQUIRE(ostride == 1 || ostride == 2, GFE_ERR_SHAPE);
    ConvParams p;
    p.x = (const bf16_t*)x; p.w = (const bf16_t*)w_packed; p.gn_scale = gn_scale; p.gn_shift = gn_shift; p.bias = bias;
    p.res = (const bf16_t*)res; p.y = (bf16_t*)y;
    p.B = (int)B; p.D = (int)D; p.H = (int)H; p.W = (int)W; p.Cin = (int)Cin; p.Cout = (int)Cout;
    p.CoutPad = gfe_conv3d_cout_pad(Cout);
    p.OD = (int)OD; p.OH = (int)OH; p.OW = (int)OW;
    p.ntaps = ntaps; p.nslab = (int)ceil_div(Cin, 32);
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
    p.LHW = p.LH * p.LW; p.NV = p.LD * p.LHW;
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
    hipStream_t st = (hipStream_t)stream;
    // regular 3x3x3 tap list in canonical order -> immediate-offset fast path
    bool reg27 = ntaps == 27 && ostride == 1;
    for (int t = 0; t < ntaps && reg27; ++t)
        reg27 = tap_offsets[3 * t] == t / 9 - 1 && tap_offsets[3 * t + 1] == (t / 3) % 3 - 1 && tap_offsets[3 * t + 2] == t % 3 - 1;
    if (p.CoutPad == 16) return conv_launch<1, 3, true, false>(p, st);
    if (p.CoutPad == 32) return conv_launch<2, 3, true, false>(p, st);
    if (p.CoutPad == 64) return reg27 ? conv_launch<4, 3, true, true>(p, st) : conv_launch<4, 3, true, false>(p, st);
    return conv_launch<8, 1, false, false>(p, st);
}

}  // extern "C"
