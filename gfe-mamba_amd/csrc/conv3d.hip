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
//   LDS: activation halo tile [<=6x10x10 voxels][32 ch] bf16, 80-B voxel stride (64 B data + 16 B pad: the 16-lane
//        ds_read_b128 groups land on distinct 16-B bank slots), GroupNorm scale/shift applied while staging,
//        zero padding written as exact zeros (the reference pads AFTER the norm);
//        weights [taps-per-stage][Cout tile][32] bf16 double-buffered, next stage prefetched to registers under the MFMAs.
//   Each wave owns one d-plane of the tile: 4 voxel tiles x NT channel tiles of f32x4 accumulators.
//   Operands are swapped (weights = A) so a lane ends up with 4 consecutive channels of one voxel -> 8-B stores.
// 2 blocks/CU (<= 80 KB LDS each) so one block's staging overlaps the other's MFMAs.
#include "common.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

namespace {

constexpr int TD = 4, TH = 8, TW = 8;          // output tile (class-grid voxels)
constexpr int VSTRIDE = 80;                    // bytes per voxel / weight row in LDS (32 bf16 + 16 B pad)
constexpr int A_MAX_VOX = (TD + 2) * (TH + 2) * (TW + 2);
constexpr int A_BYTES = A_MAX_VOX * VSTRIDE;   // 48,000

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
    int ntd, nth, ntw;
    ConvTap taps[27];
};

template <int NT, int TPS>
__global__ __launch_bounds__(256, 2) void conv_igemm_kernel(const ConvParams p) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint8_t* sA = smem;
    uint8_t* sW = smem + A_BYTES;
    constexpr int WROWS = TPS * NT * 16;              // weight rows per stage
    constexpr int WSTAGE_BYTES = WROWS * VSTRIDE;
    constexpr int WITEMS = (WROWS * 4 + 255) / 256;   // 16-B chunks per thread per stage

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lq = lane >> 4, lr = lane & 15;

    // ---- tile decode (tiles adjacent in w/h/d are adjacent in block id: neighbours share halo lines in L2)
    int bid = blockIdx.x;
    const int tw_ = bid % p.ntw; bid /= p.ntw;
    const int th_ = bid % p.nth; bid /= p.nth;
    const int td_ = bid % p.ntd; bid /= p.ntd;
    const int b = bid;
    const int cg = blockIdx.y;
    const int d0 = td_ * TD, h0 = th_ * TH, w0 = tw_ * TW;
    const int LHW = p.LH * p.LW, NV = p.LD * LHW;

    f32x4 acc[4][NT];
#pragma unroll
    for (int xt = 0; xt < 4; ++xt)
#pragma unroll
        for (int ct = 0; ct < NT; ++ct) acc[xt][ct] = f32x4{0.f, 0.f, 0.f, 0.f};

    // per-lane LDS byte offsets of the 4 voxel tiles (tap offset added per tap) and of the weight fragment
    int abase[4];
#pragma unroll
    for (int xt = 0; xt < 4; ++xt) {
        const int lh = 2 * xt + (lr >> 3), lw = lr & 7;
        abase[xt] = ((wave * p.LH + lh) * p.LW + lw) * VSTRIDE + lq * 16;
    }
    const int wbase = lr * VSTRIDE + lq * 16;
    const int nstage = (p.ntaps + TPS - 1) / TPS;
    const int chunk = tid & 3;                         // this thread's 8-channel chunk inside a slab (fixed: 256 % 4 == 0)

    for (int slab = 0; slab < p.nslab; ++slab) {
        // weights of this slab / channel group: [slab][tap][CoutPad][32]
        const bf16_t* wslab = p.w + ((size_t)slab * p.ntaps * p.CoutPad + (size_t)cg * NT * 16) * 32;
        auto wload = [&](int stage, uint4 (&r)[WITEMS]) {
#pragma unroll
            for (int k = 0; k < WITEMS; ++k) {
                const int it = tid + k * 256;
                const int row = it >> 2, c = it & 3;
                const int tl = row / (NT * 16), rr = row - tl * (NT * 16);
                const int tap = stage * TPS + tl;
                r[k] = make_uint4(0, 0, 0, 0);
                if (it < WROWS * 4 && tap < p.ntaps)
                    r[k] = *reinterpret_cast<const uint4*>(wslab + ((size_t)tap * p.CoutPad + rr) * 32 + c * 8);
            }
        };
        auto wstore = [&](int buf, const uint4 (&r)[WITEMS]) {
#pragma unroll
            for (int k = 0; k < WITEMS; ++k) {
                const int it = tid + k * 256;
                if (it < WROWS * 4) *reinterpret_cast<uint4*>(sW + buf * WSTAGE_BYTES + (it >> 2) * VSTRIDE + (it & 3) * 16) = r[k];
            }
        };

        uint4 wr[WITEMS];
        wload(0, wr);
        __syncthreads();      // every wave is done reading the previous slab's tile and weight buffers

        // ---- stage the activation halo tile of this slab (GroupNorm applied; out-of-volume voxels are exact zeros)
        {
            const int cch = slab * 32 + chunk * 8;           // first channel of this thread's chunk
            const bool ch_ok = cch < p.Cin;
            float sc[8], sh[8];
            if (p.gn_scale && ch_ok) {
#pragma unroll
                for (int j = 0; j < 8; ++j) { sc[j] = p.gn_scale[(size_t)b * p.Cin + cch + j]; sh[j] = p.gn_shift[(size_t)b * p.Cin + cch + j]; }
            }
            for (int it = tid; it < NV * 4; it += 256) {
                const int lv = it >> 2;
                const int ld = lv / LHW, rem = lv - ld * LHW;
                const int lh = rem / p.LW, lw = rem - lh * p.LW;
                const int gd = d0 + p.lo_d + ld, gh = h0 + p.lo_h + lh, gw = w0 + p.lo_w + lw;
                uint4 v = make_uint4(0, 0, 0, 0);
                if (ch_ok && (unsigned)gd < (unsigned)p.D && (unsigned)gh < (unsigned)p.H && (unsigned)gw < (unsigned)p.W) {
                    v = *reinterpret_cast<const uint4*>(p.x + ((((size_t)b * p.D + gd) * p.H + gh) * p.W + gw) * p.Cin + cch);
                    if (p.gn_scale) {
                        v.x = pack_bf16x2(fmaf(bf16lo_to_f32(v.x), sc[0], sh[0]), fmaf(bf16hi_to_f32(v.x), sc[1], sh[1]));
                        v.y = pack_bf16x2(fmaf(bf16lo_to_f32(v.y), sc[2], sh[2]), fmaf(bf16hi_to_f32(v.y), sc[3], sh[3]));
                        v.z = pack_bf16x2(fmaf(bf16lo_to_f32(v.z), sc[4], sh[4]), fmaf(bf16hi_to_f32(v.z), sc[5], sh[5]));
                        v.w = pack_bf16x2(fmaf(bf16lo_to_f32(v.w), sc[6], sh[6]), fmaf(bf16hi_to_f32(v.w), sc[7], sh[7]));
                    }
                }
                *reinterpret_cast<uint4*>(sA + lv * VSTRIDE + chunk * 16) = v;
            }
        }
        wstore(0, wr);
        __syncthreads();

        for (int s = 0; s < nstage; ++s) {
            const bool more = s + 1 < nstage;
            if (more) wload(s + 1, wr);                       // in flight under this stage's MFMAs
            const uint8_t* wb = sW + (s & 1) * WSTAGE_BYTES + wbase;
#pragma unroll
            for (int tl = 0; tl < TPS; ++tl) {
                const int tap = s * TPS + tl;
                if (tap < p.ntaps) {                          // wave-uniform
                    const ConvTap tp = p.taps[tap];
                    const int toff = (((tp.dd - p.lo_d) * p.LH + (tp.dh - p.lo_h)) * p.LW + (tp.dw - p.lo_w)) * VSTRIDE;
                    bf16x8 wf[NT];
#pragma unroll
                    for (int ct = 0; ct < NT; ++ct)
                        wf[ct] = *reinterpret_cast<const bf16x8*>(wb + (tl * NT * 16 + ct * 16) * VSTRIDE);
#pragma unroll
                    for (int xt = 0; xt < 4; ++xt) {
                        const bf16x8 xf = *reinterpret_cast<const bf16x8*>(sA + abase[xt] + toff);
#pragma unroll
                        for (int ct = 0; ct < NT; ++ct)
                            acc[xt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ct], xf, acc[xt][ct], 0, 0, 0);
                    }
                }
            }
            if (more) wstore((s + 1) & 1, wr);
            __syncthreads();
        }
    }

    // ---- epilogue: bias, skip/residual add, ReLU, bf16 store (lane: voxel = lr of tile xt, channels 16*ct + 4*lq + 0..3)
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
#pragma unroll
                    for (int ct = 0; ct < NT; ++ct) {
                        const int c0 = (cg * NT + ct) * 16 + lq * 4;
                        if (c0 >= p.Cout) continue;
                        float v[4] = {acc[xt][ct][0], acc[xt][ct][1], acc[xt][ct][2], acc[xt][ct][3]};
                        if (p.bias) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) v[r] += p.bias[c0 + r];
                        }
                        const size_t o = vox * p.Cout + c0;
                        if (p.res) {
                            const uint2 rv = *reinterpret_cast<const uint2*>(p.res + o);
                            v[0] += bf16lo_to_f32(rv.x); v[1] += bf16hi_to_f32(rv.x);
                            v[2] += bf16lo_to_f32(rv.y); v[3] += bf16hi_to_f32(rv.y);
                        }
                        if (p.relu) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
                        }
                        *reinterpret_cast<uint2*>(p.y + o) = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
                    }
                }
    }
}

template <int NT, int TPS>
int conv_launch(const ConvParams& p, hipStream_t st) {
    const size_t lds = A_BYTES + 2 * (size_t)TPS * NT * 16 * VSTRIDE;
    const int64_t tiles = (int64_t)p.B * p.ntd * p.nth * p.ntw;
    if (tiles > 0x7fffffff) return GFE_ERR_SHAPE;
    const dim3 grid((unsigned)tiles, (unsigned)(p.CoutPad / (NT * 16)));
    static bool attr_set = false;
    if (!attr_set) { (void)hipFuncSetAttribute((const void*)conv_igemm_kernel<NT, TPS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr_set = true; }
    hipLaunchKernelGGL((conv_igemm_kernel<NT, TPS>), grid, dim3(256), lds, st, p);
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
        p.taps[t].dd = o[0]; p.taps[t].dh = o[1]; p.taps[t].dw = o[2]; p.taps[t].pad = 0;
    }
    p.lo_d = lo[0]; p.lo_h = lo[1]; p.lo_w = lo[2];
    p.LD = TD + hi[0] - lo[0]; p.LH = TH + hi[1] - lo[1]; p.LW = TW + hi[2] - lo[2];
    p.ostride = ostride; p.op_d = op_d; p.op_h = op_h; p.op_w = op_w; p.oshift = oshift; p.relu = relu;
    if (ostride == 1) {
        GFE_REQUIRE(OD == D && OH == H && OW == W && oshift == 0 && op_d == 0 && op_h == 0 && op_w == 0, GFE_ERR_SHAPE);
    } else {
        GFE_REQUIRE(OD == 2 * D - 1 + oshift && OH == 2 * H - 1 + oshift && OW == 2 * W - 1 + oshift, GFE_ERR_SHAPE);
    }
    p.ntd = (int)ceil_div(D, TD); p.nth = (int)ceil_div(H, TH); p.ntw = (int)ceil_div(W, TW);
    hipStream_t st = (hipStream_t)stream;
    if (p.CoutPad == 16) return conv_launch<1, 3>(p, st);
    if (p.CoutPad == 32) return conv_launch<2, 3>(p, st);
    if (p.CoutPad == 64) return conv_launch<4, 3>(p, st);
    return conv_launch<8, 1>(p, st);
}

}  // extern "C"
