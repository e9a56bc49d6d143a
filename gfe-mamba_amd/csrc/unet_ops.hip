// HBM-bound helpers of the frozen generator (channels-last bf16 activations): GroupNorm statistics, MaxPool3d(2),
// the 1->C input lift and C->1 output projection (1x1x1 convs), and the bottleneck fold/unfold permutations.
// Reference: pytorch3dunet/unet3d/buildingblocks.py:55-67 (GroupNorm before conv), :284,306-307 (MaxPool3d),
// :191-196 (conv1 1x1x1 + bias), pytorch3dunet/unet3d/model.py:123,162 (final_conv), :150-152 (md1 fold/unfold).
#include "common.h"

namespace {

// ---- GroupNorm statistics --------------------------------------------------------------------------------------
// pass 1: per-block per-channel (sum, sumsq) partials, 16-B loads; pass 2 (one block per sample): f64 reduction over
// blocks and over the channels of each group, then scale[b,c] = rstd*gamma, shift[b,c] = beta - mean*rstd*gamma, which
// is what the conv kernel applies while staging its input tile.
__global__ __launch_bounds__(256) void gn_partial_kernel(const bf16_t* __restrict__ x, float* __restrict__ ws,
                                                         int64_t S, int C, int vpb, int nblk) {
    extern __shared__ float lds[];               // [VI][2][C]: one slot per voxel lane, merged in lane order (round 5: was LDS float atomics,
                                                 // whose order -- and with it the last bits of the training forward -- changed from run to run)
    const int b = blockIdx.y, blk = blockIdx.x, tid = threadIdx.x;
    const int CP = C >> 3;                       // 16-B chunks per voxel
    const int c = tid % CP, vl = tid / CP, VI = 256 / CP;
    float s[8], q[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { s[j] = 0.f; q[j] = 0.f; }
    const int64_t v0 = (int64_t)blk * vpb;
    const int64_t v1 = v0 + vpb < S ? v0 + vpb : S;
    const bf16_t* xb = x + (size_t)b * S * C + c * 8;
    for (int64_t v = v0 + vl; v < v1; v += VI) {
        const uint4 w = *reinterpret_cast<const uint4*>(xb + (size_t)v * C);
        const float f[8] = {bf16lo_to_f32(w.x), bf16hi_to_f32(w.x), bf16lo_to_f32(w.y), bf16hi_to_f32(w.y),
                            bf16lo_to_f32(w.z), bf16hi_to_f32(w.z), bf16lo_to_f32(w.w), bf16hi_to_f32(w.w)};
#pragma unroll
        for (int j = 0; j < 8; ++j) { s[j] += f[j]; q[j] = fmaf(f[j], f[j], q[j]); }
    }
    if (vl < VI) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { lds[(vl * 2) * C + c * 8 + j] = s[j]; lds[(vl * 2 + 1) * C + c * 8 + j] = q[j]; }
    }
    __syncthreads();
    float* o = ws + ((size_t)b * nblk + blk) * 2 * C;
    for (int i = tid; i < 2 * C; i += 256) {
        float t = 0.f;
        for (int k = 0; k < VI; ++k) t += lds[k * 2 * C + i];
        o[i] = t;
    }
}

__global__ __launch_bounds__(1024) void gn_finalize_kernel(const float* __restrict__ ws, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float* __restrict__ scale,
                                                           float* __restrict__ shift, int64_t S, int C, int G, int nblk, float eps) {
    extern __shared__ double dl[];               // [2][C] channel sums, then [2][G] mean / rstd, then [tpc][2C] per-thread sums
    const int b = blockIdx.x, tid = threadIdx.x;
    // column c = tid % (2C) is summed by 1024/(2C) threads striding over the nblk partials (4 independent chains each), merged
    // through a plain LDS table (double-precision LDS atomics are CAS loops)
    const int ncol = 2 * C;
    const int tpc = 1024 / ncol;                 // threads per column (2C <= 1024)
    double* part_sums = dl + ncol + 2 * G;
    if (tid < tpc * ncol) {
        const int c = tid % ncol, part = tid / ncol;
        const float* col = ws + (size_t)b * nblk * ncol + c;
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
        int k = part;
        for (; k + 3 * tpc < nblk; k += 4 * tpc) {
            a0 += (double)col[(size_t)k * ncol]; a1 += (double)col[(size_t)(k + tpc) * ncol];
            a2 += (double)col[(size_t)(k + 2 * tpc) * ncol]; a3 += (double)col[(size_t)(k + 3 * tpc) * ncol];
        }
        for (; k < nblk; k += tpc) a0 += (double)col[(size_t)k * ncol];
        part_sums[part * ncol + c] = (a0 + a1) + (a2 + a3);
    }
    __syncthreads();
    for (int c = tid; c < ncol; c += 1024) {
        double a = 0.0;
        for (int q = 0; q < tpc; ++q) a += part_sums[q * ncol + c];
        dl[c] = a;
    }
    __syncthreads();
    double* gm = dl + 2 * C;
    const int cg = C / G;
    for (int g = tid; g < G; g += 1024) {
        double s = 0.0, q = 0.0;
        for (int j = 0; j < cg; ++j) { s += dl[g * cg + j]; q += dl[C + g * cg + j]; }
        const double n = (double)cg * (double)S;
        const double mean = s / n;
        double var = q / n - mean * mean;      // biased variance, as nn.GroupNorm
        if (var < 0.0) var = 0.0;
        gm[g] = mean; gm[G + g] = 1.0 / sqrt(var + (double)eps);
    }
    __syncthreads();
    for (int c = tid; c < C; c += 1024) {
        const int g = c / cg;
        const float sc = (float)gm[G + g] * gamma[c];
        scale[(size_t)b * C + c] = sc;
        shift[(size_t)b * C + c] = beta[c] - (float)gm[g] * sc;
    }
}

// ---- MaxPool3d(kernel 2, stride 2, floor) -------------------------------------------------------------------
__device__ __forceinline__ uint32_t max_bf16x2(uint32_t a, uint32_t b) {
    const float lo = fmaxf(bf16lo_to_f32(a), bf16lo_to_f32(b)), hi = fmaxf(bf16hi_to_f32(a), bf16hi_to_f32(b));
    return (__float_as_uint(lo) >> 16) | (__float_as_uint(hi) & 0xffff0000u);     // exact: inputs are bf16 values
}
__global__ __launch_bounds__(256) void maxpool_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y,
                                                      int B, int D, int H, int W, int C) {
    const int CP = C >> 3, OD = D >> 1, OH = H >> 1, OW = W >> 1;
    const int64_t total = (int64_t)B * OD * OH * OW * CP;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % CP); int64_t v = i / CP;
        const int ow = (int)(v % OW); v /= OW;
        const int oh = (int)(v % OH); v /= OH;
        const int od = (int)(v % OD); const int b = (int)(v / OD);
        uint4 m;
        bool first = true;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int d = 2 * od + (k >> 2), h = 2 * oh + ((k >> 1) & 1), w = 2 * ow + (k & 1);
            const uint4 t = *reinterpret_cast<const uint4*>(x + ((((size_t)b * D + d) * H + h) * W + w) * C + c * 8);
            if (first) { m = t; first = false; }
            else { m.x = max_bf16x2(m.x, t.x); m.y = max_bf16x2(m.y, t.y); m.z = max_bf16x2(m.z, t.z); m.w = max_bf16x2(m.w, t.w); }
        }
        *reinterpret_cast<uint4*>(y + ((((size_t)b * OD + od) * OH + oh) * OW + ow) * C + c * 8) = m;
    }
}

// ---- 1 -> C pointwise conv with bias (encoders.0.basic_module.conv1), input f32 or bf16 single channel ----------
template <typename TI>
__global__ __launch_bounds__(256) void conv_in1_kernel(const TI* __restrict__ x, const float* __restrict__ w,
                                                       const float* __restrict__ bias, bf16_t* __restrict__ y, int64_t nvox, int C) {
    const int CP = C >> 3;
    const int64_t total = nvox * CP;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % CP); const int64_t v = i / CP;
        const float xv = IO<TI>::ld(x + v);
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = fmaf(w[c * 8 + j], xv, bias[c * 8 + j]);
        *reinterpret_cast<uint4*>(y + (size_t)v * C + c * 8) =
            make_uint4(pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]), pack_bf16x2(o[4], o[5]), pack_bf16x2(o[6], o[7]));
    }
}

// the same with the GroupNorm partials of y (sum / sum of squares of the rounded outputs per channel): block (blk, b) owns voxels
// [blk*vpb, (blk+1)*vpb) of sample b and writes slot blk -- the layout gn_partial_kernel produces, without re-reading y
template <typename TI>
__global__ __launch_bounds__(256) void conv_in1_stats_kernel(const TI* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                             bf16_t* __restrict__ y, float* __restrict__ ws, int64_t S, int C, int vpb, int nblk) {
    extern __shared__ float lds[];               // [2][C]
    const int b = blockIdx.y, blk = blockIdx.x, tid = threadIdx.x;
    const int CP = C >> 3;
    const int c = tid % CP, vl = tid / CP, VI = 256 / CP;
    for (int i = tid; i < 2 * C; i += 256) lds[i] = 0.f;
    __syncthreads();
    float wc[8], bc[8], s[8], q[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { wc[j] = w[c * 8 + j]; bc[j] = bias[c * 8 + j]; s[j] = 0.f; q[j] = 0.f; }
    const int64_t v0 = (int64_t)blk * vpb;
    const int64_t v1 = v0 + vpb < S ? v0 + vpb : S;
    for (int64_t v = v0 + vl; v < v1; v += VI) {
        const float xv = IO<TI>::ld(x + (size_t)b * S + v);
        uint32_t pk[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            pk[j] = pack_bf16x2(fmaf(wc[2 * j], xv, bc[2 * j]), fmaf(wc[2 * j + 1], xv, bc[2 * j + 1]));
            const float r0 = bf16lo_to_f32(pk[j]), r1 = bf16hi_to_f32(pk[j]);
            s[2 * j] += r0; s[2 * j + 1] += r1; q[2 * j] = fmaf(r0, r0, q[2 * j]); q[2 * j + 1] = fmaf(r1, r1, q[2 * j + 1]);
        }
        *reinterpret_cast<uint4*>(y + ((size_t)b * S + v) * C + c * 8) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
    }
    // fixed-order merge of the VI voxel lanes of a channel chunk (deterministic, unlike LDS float atomics)
    float* red = lds + 2 * C;                    // [VI][2][C]
#pragma unroll
    for (int j = 0; j < 8; ++j) { red[(vl * 2) * C + c * 8 + j] = s[j]; red[(vl * 2 + 1) * C + c * 8 + j] = q[j]; }
    __syncthreads();
    float* o = ws + ((size_t)b * nblk + blk) * 2 * C;
    for (int i = tid; i < 2 * C; i += 256) {
        float t = 0.f;
        for (int k = 0; k < VI; ++k) t += red[k * 2 * C + i];
        o[i] = t;
    }
}

// folds nblk partial slots 32-ways: out[b][r] = sum of slots r, r+32, ... (fixed order)
__global__ __launch_bounds__(256) void gn_reduce_kernel(const float* __restrict__ ws, float* __restrict__ out, int nblk, int ncol) {
    const int b = blockIdx.y, r = blockIdx.x;
    for (int c = threadIdx.x; c < ncol; c += 256) {
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        int k = r;
        for (; k + 96 < nblk; k += 128) {
            a0 += ws[((size_t)b * nblk + k) * ncol + c];      a1 += ws[((size_t)b * nblk + k + 32) * ncol + c];
            a2 += ws[((size_t)b * nblk + k + 64) * ncol + c]; a3 += ws[((size_t)b * nblk + k + 96) * ncol + c];
        }
        for (; k < nblk; k += 32) a0 += ws[((size_t)b * nblk + k) * ncol + c];
        out[((size_t)b * 32 + r) * ncol + c] = (a0 + a1) + (a2 + a3);
    }
}

// ---- C -> 1 pointwise conv with bias (final_conv), output f32; 16-B lanes, C/8 lanes per voxel ------------------
__global__ __launch_bounds__(256) void conv_out1_kernel(const bf16_t* __restrict__ x, const float* __restrict__ w, float bias,
                                                        float* __restrict__ y, int64_t nvox, int C) {
    const int CP = C >> 3;                       // power of two <= 64
    const int64_t total = nvox * CP;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i0 = (int64_t)blockIdx.x * 256; i0 < total; i0 += stride) {
        const int64_t i = i0 + threadIdx.x;
        float acc = 0.f;
        if (i < total) {
            const int c = (int)(i % CP);
            const uint4 t = *reinterpret_cast<const uint4*>(x + (size_t)i * 8);
            const float* wc = w + c * 8;
            acc = bf16lo_to_f32(t.x) * wc[0] + bf16hi_to_f32(t.x) * wc[1] + bf16lo_to_f32(t.y) * wc[2] + bf16hi_to_f32(t.y) * wc[3] +
                  bf16lo_to_f32(t.z) * wc[4] + bf16hi_to_f32(t.z) * wc[5] + bf16lo_to_f32(t.w) * wc[6] + bf16hi_to_f32(t.w) * wc[7];
        }
        for (int o = CP >> 1; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
        if (i < total && (i % CP) == 0) y[i / CP] = acc + bias;
    }
}

// ---- bottleneck fold 'b c (md1 md2) h w -> b c (h md1) (md2 w)' on channels-last data, and its inverse ----------
// src voxel (d = i1*md2 + i2, h, w)  <->  dst pixel (row = h*md1 + i1, col = i2*W + w); pure index permutation.
__global__ __launch_bounds__(256) void fold_mid_kernel(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst,
                                                       int B, int D, int H, int W, int C, int md1, int inverse) {
    const int CP = C >> 3, md2 = D / md1;
    const int64_t total = (int64_t)B * D * H * W * CP;
    const int cols = md2 * W;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % CP); int64_t v = i / CP;
        const int w = (int)(v % W); v /= W;
        const int h = (int)(v % H); v /= H;
        const int d = (int)(v % D); const int b = (int)(v / D);
        const int i1 = d / md2, i2 = d - i1 * md2;
        const size_t vox = (((size_t)b * D + d) * H + h) * W + w;
        const size_t pix = ((size_t)b * (H * md1) + (h * md1 + i1)) * cols + (i2 * W + w);
        const size_t s = (inverse ? pix : vox) * C + c * 8, t = (inverse ? vox : pix) * C + c * 8;
        *reinterpret_cast<uint4*>(dst + t) = *reinterpret_cast<const uint4*>(src + s);
    }
}

static unsigned grid_for(int64_t total) {
    int64_t g = ceil_div(total, 256);
    return (unsigned)(g < 1 ? 1 : (g > 8192 ? 8192 : g));
}


// ---- 3x3x3 convolution of a SINGLE-channel volume with per-sample effective weights -------------------------------
// The first ResNetBlock of the generator computes conv2(GroupNorm(conv1(x))) where conv1 is a 1x1x1 lift of the one-channel MRI
// volume: r_c = w1_c x + b1_c.  GroupNorm is affine per (sample, channel), x_hat_c = s_c r_c + t_c, so the 64 -> 64 convolution
// over x_hat collapses algebraically into a ONE-channel 27-tap convolution of x itself:
//     y_o(v) = sum_tap Weff[o][tap] x(v + tap) + sum_{tap inside} Beff[o][tap],   Weff = sum_c W[o][c][tap] s_c w1_c,
//     Beff = sum_c W[o][c][tap] (s_c b1_c + t_c)   (the second term is the usual boundary-class bias table: the reference pads x_hat
//     with zeros AFTER the norm).  196 GFLOP per volume become 0.24, and the kernel is bound by writing y.
// It runs on the matrix cores: per 16 voxels the B operand is the voxels' 27 neighbours (padded to K = 32, gathered from an LDS halo
// tile and rounded to bf16: 8 taps per lane), the A operand the sample's effective weights (64 channels x 32, bf16, loaded once per
// block into registers with the same row permutation as the conv kernel, so a lane ends up with 16 consecutive channels of one voxel)
// -- four v_mfma_f32_16x16x32_bf16 per 16 voxels, and the kernel is bound by writing y.  (A VALU formulation, one or two voxels per
// thread x 64 f32 accumulators, took 0.73 / 0.51 ms at 96^3, B=8: LDS-broadcast- resp. scalar-load-bound.)
// One block = 8 waves = the 8 d-planes of an 8x8x8 tile, 4 voxel tiles of 2 x 8 voxels per wave, as in conv3d.hip; the next tile's halo is
// fetched into registers while the current one is computed.  Also writes the GroupNorm partials of y (8-channel sums, slot = block).
typedef __attribute__((ext_vector_type(2))) __bf16 c1_bf16x2;
typedef __attribute__((ext_vector_type(8))) __bf16 c1_bf16x8;
typedef __attribute__((ext_vector_type(4))) float c1_f32x4;
template <typename TI>
__global__ __launch_bounds__(512) void conv3d_c1_k3_kernel(const TI* __restrict__ x, const float* __restrict__ weff, const float* __restrict__ tab,
                                                           bf16_t* __restrict__ y, float* __restrict__ ws, int nblk, int D, int H, int W, int relu) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int C = 64;
    __shared__ float xt[10 * 10 * 10];
    __shared__ __attribute__((aligned(16))) float tab0[C];
    __shared__ float red[8 * 16];
    const int b = blockIdx.y, blk = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    GFE_FUZZ_INIT();
    const int lq = lane >> 4, lr = lane & 15;
    const int ntd = (D + 7) >> 3, nth = (H + 7) >> 3, ntw = (W + 7) >> 3, ntiles = ntd * nth * ntw;
    const int per = (ntiles + nblk - 1) / nblk, t_begin = blk * per, t_end = min(ntiles, t_begin + per);
    const size_t S = (size_t)D * H * W;
    if (tid < C) tab0[tid] = tab[(size_t)b * 64 * C + tid];
    // A operand: lane (lq, lr) holds k = 8 lq .. 8 lq + 7 (taps; 27..31 are zero) of MFMA row lr of channel tile ct, where row m stands
    // for channel (m >> 2) * 16 + 4 ct + (m & 3) -- the output rows 4 lq + r of tile ct are then channels lq * 16 + 4 ct + r
    c1_bf16x8 wfrag[4];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
        const int ch = (lr >> 2) * 16 + 4 * ct + (lr & 3);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int t = 8 * lq + j;
            wfrag[ct][j] = (__bf16)(t < 27 ? weff[((size_t)b * 27 + t) * C + ch] : 0.f);
        }
    }
    // B operand gather: LDS offsets of this lane's 8 taps relative to a voxel's halo position (tap t -> (t/9, (t/3)%3, t%3))
    int toff[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int t = 8 * lq + j, tt = t < 27 ? t : 0;
        toff[j] = ((tt / 9) * 10 + (tt / 3) % 3) * 10 + tt % 3;
    }
    const bool lastq = lq == 3;                                  // taps 27..31 do not exist: zeros (the weights there are zero as well)
    float gs[2] = {0.f, 0.f}, gq[2] = {0.f, 0.f};
    float pre[2];
    auto fetch = [&](int t) {
        const int tw = t % ntw, th = (t / ntw) % nth, td = t / (ntw * nth);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int i = tid + 512 * j;
            const int hz = i / 100, hy = (i / 10) % 10, hx = i % 10;
            const int gd = td * 8 + hz - 1, gh = th * 8 + hy - 1, gw = tw * 8 + hx - 1;
            pre[j] = (i < 1000 && (unsigned)gd < (unsigned)D && (unsigned)gh < (unsigned)H && (unsigned)gw < (unsigned)W)
                         ? IO<TI>::ld(x + (size_t)b * S + ((size_t)gd * H + gh) * W + gw) : 0.f;
        }
    };
    if (t_begin < t_end) fetch(t_begin);
    for (int t = t_begin; t < t_end; ++t) {
        const int tw = t % ntw, th = (t / ntw) % nth, td = t / (ntw * nth);
        const int d0 = td * 8, h0 = th * 8, w0 = tw * 8;
        GFE_FUZZ();
        __syncthreads();                                        // (previous tile's readers are done; first pass: tab0 visible)
        GFE_FUZZ();
#pragma unroll
        for (int j = 0; j < 2; ++j)
            if (tid + 512 * j < 1000) xt[tid + 512 * j] = pre[j];
        GFE_FUZZ();
        __syncthreads();
        GFE_FUZZ();
        if (t + 1 < t_end) fetch(t + 1);
        const int gd = d0 + wave;
        if (gd >= D) continue;                                  // (no barrier below in this iteration)
#pragma unroll
        for (int xt_ = 0; xt_ < 4; ++xt_) {
            const int ly = 2 * xt_ + (lr >> 3), lx = lr & 7;
            const int vbase = (wave * 10 + ly) * 10 + lx;        // halo index of the voxel's (-1, -1, -1) neighbour
            c1_bf16x8 xf;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float v = xt[vbase + toff[j]];
                xf[j] = (__bf16)((lastq && j >= 3) ? 0.f : v);
            }
            c1_f32x4 acc[4];
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) acc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wfrag[ct], xf, c1_f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
            const int gh = h0 + ly, gw = w0 + lx;
            if (gh >= H || gw >= W) continue;
            const int cls = (gd == 0) | ((gd == D - 1) << 1) | ((gh == 0) << 2) | ((gh == H - 1) << 3) | ((gw == 0) << 4) | ((gw == W - 1) << 5);
            const float4* br = cls == 0 ? reinterpret_cast<const float4*>(tab0 + lq * 16) : reinterpret_cast<const float4*>(tab + ((size_t)b * 64 + cls) * C + lq * 16);
            bf16_t* yo = y + ((size_t)b * S + ((size_t)gd * H + gh) * W + gw) * C + lq * 16;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                uint32_t pk[4];
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    const float4 bv = br[2 * h + jj];
                    c1_f32x4 a = acc[2 * h + jj];
                    a[0] += bv.x; a[1] += bv.y; a[2] += bv.z; a[3] += bv.w;
                    if (relu) { a[0] = a[0] > 0.f ? a[0] : 0.f; a[1] = a[1] > 0.f ? a[1] : 0.f; a[2] = a[2] > 0.f ? a[2] : 0.f; a[3] = a[3] > 0.f ? a[3] : 0.f; }
                    pk[2 * jj] = pack_bf16x2(a[0], a[1]); pk[2 * jj + 1] = pack_bf16x2(a[2], a[3]);
                }
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const c1_bf16x2 pv = __builtin_bit_cast(c1_bf16x2, pk[jj]);
                    gs[h] = __builtin_amdgcn_fdot2_f32_bf16(pv, __builtin_bit_cast(c1_bf16x2, 0x3f803f80u), gs[h], false);
                    gq[h] = __builtin_amdgcn_fdot2_f32_bf16(pv, pv, gq[h], false);
                }
                reinterpret_cast<uint4*>(yo)[h] = make_uint4(pk[0], pk[1], pk[2], pk[3]);
            }
        }
    }
    // GroupNorm partials of what this block stored: octet lq * 2 + h; the 16 voxel lanes of a row, then the 8 waves through LDS (fixed order)
    __syncthreads();
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        float s_ = gs[h], q_ = gq[h];
#pragma unroll
        for (int m = 8; m >= 1; m >>= 1) { s_ += __shfl_xor(s_, m, 64); q_ += __shfl_xor(q_, m, 64); }
        if (lr == 0) { red[wave * 16 + lq * 2 + h] = s_; red[wave * 16 + 8 + lq * 2 + h] = q_; }
    }
    __syncthreads();
    if (tid < 2 * C) {
        const int st = tid >> 6, c = tid & 63;
        float v = 0.f;                    // the 8-channel sum goes to the first channel of the octet, zeros to the other seven
        if ((c & 7) == 0) for (int wv_ = 0; wv_ < 8; ++wv_) v += red[wv_ * 16 + st * 8 + (c >> 3)];
        ws[(((size_t)b * nblk + blk) * 2 + st) * C + c] = v;
    }
#endif
}

// ---- GroupNorm affine of a 1x1x1 lift of a one-channel volume, without materialising the lift ---------------------------------
// r_c = w_c x + b_c per voxel, so the per-(sample, group) statistics GroupNorm needs follow from the first two moments of x:
//   E[r_c] = w_c E[x] + b_c,   E[r_c^2] = w_c^2 E[x^2] + 2 w_c b_c E[x] + b_c^2,   averaged over the group's channels.
__global__ __launch_bounds__(256) void vol_moments_kernel(const float* __restrict__ x, double* __restrict__ part, int64_t S, int nblk) {
    __shared__ double rs[4], rq[4];
    const int b = blockIdx.y, blk = blockIdx.x, tid = threadIdx.x;
    const float* xb = x + (size_t)b * S;
    double s = 0.0, q = 0.0;
    const int64_t n4 = ((uintptr_t)xb % 16 == 0) ? S / 4 : 0;
    for (int64_t i = (int64_t)blk * 256 + tid; i < n4; i += (int64_t)nblk * 256) {
        const float4 v = reinterpret_cast<const float4*>(xb)[i];
        s += (double)v.x + (double)v.y + (double)v.z + (double)v.w;
        q += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
    }
    for (int64_t i = n4 * 4 + (int64_t)blk * 256 + tid; i < S; i += (int64_t)nblk * 256) { const double v = xb[i]; s += v; q += v * v; }
    for (int m = 32; m >= 1; m >>= 1) { s += __shfl_xor(s, m, 64); q += __shfl_xor(q, m, 64); }
    if ((tid & 63) == 0) { rs[tid >> 6] = s; rq[tid >> 6] = q; }
    __syncthreads();
    if (tid == 0) { part[((size_t)b * nblk + blk) * 2] = rs[0] + rs[1] + rs[2] + rs[3]; part[((size_t)b * nblk + blk) * 2 + 1] = rq[0] + rq[1] + rq[2] + rq[3]; }
}
__global__ __launch_bounds__(256) void lift_gn_affine_kernel(const double* __restrict__ part, const float* __restrict__ w, const float* __restrict__ bias,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ scale,
                                                             float* __restrict__ shift, int64_t S, int C, int G, int nblk, float eps) {
    __shared__ double mom[2];
    __shared__ double gm[2 * 256];
    const int b = blockIdx.x, tid = threadIdx.x;
    if (tid == 0) {
        double s = 0.0, q = 0.0;
        for (int k = 0; k < nblk; ++k) { s += part[((size_t)b * nblk + k) * 2]; q += part[((size_t)b * nblk + k) * 2 + 1]; }
        mom[0] = s / (double)S; mom[1] = q / (double)S;
    }
    __syncthreads();
    const int cg = C / G;
    for (int g = tid; g < G; g += 256) {
        double m1 = 0.0, m2 = 0.0;
        for (int j = 0; j < cg; ++j) {
            const double wc = w[g * cg + j], bc = bias[g * cg + j];
            m1 += wc * mom[0] + bc;
            m2 += wc * wc * mom[1] + 2.0 * wc * bc * mom[0] + bc * bc;
        }
        m1 /= cg; m2 /= cg;
        double var = m2 - m1 * m1;                 // biased variance, as nn.GroupNorm
        if (var < 0.0) var = 0.0;
        gm[g] = m1; gm[G + g] = 1.0 / sqrt(var + (double)eps);
    }
    __syncthreads();
    for (int c = tid; c < C; c += 256) {
        const int g = c / cg;
        const float sc = (float)gm[G + g] * gamma[c];
        scale[(size_t)b * C + c] = sc;
        shift[(size_t)b * C + c] = beta[c] - (float)gm[g] * sc;
    }
}

// conv2(GroupNorm(conv1(x))) as ONE convolution of x (ResNetBlock._conv2_through_lift, buildingblocks.py:38-67 restated): the operands of the
// per-sample effective-weight product and its result's re-layout, one launch each where torch spent seven (they sit between two convs on the
// generator's stream, which bounds the step).  Same f32 operations in the same order as the torch expressions they replace (no contraction).
//   rhs[c][b * Cin + i] = scale[b][c] * w1[c][i]           shift2[b][c] = scale[b][c] * b1[c] + shift[b][c]
__global__ __launch_bounds__(256) void lift_fold_prep_kernel(const float* __restrict__ scale, const float* __restrict__ shift, const float* __restrict__ w1,
                                                             const float* __restrict__ b1, float* __restrict__ rhs, float* __restrict__ shift2, int B, int C, int Cin) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x, n = (int64_t)C * B * Cin;
    if (idx < n) {
        const int c = (int)(idx / ((int64_t)B * Cin)), r = (int)(idx - (int64_t)c * B * Cin), b = r / Cin, i = r - b * Cin;
        rhs[idx] = __fmul_rn(scale[(size_t)b * C + c], w1[(size_t)c * Cin + i]);
    }
    if (idx < (int64_t)B * C) {
        // `scale * b1 + shift` is two roundings in the expression this replaces; the library is built with -ffp-contract=fast and hipcc contracts
        // __fmul_rn / __fadd_rn all the same, so the product is made opaque
        float m;
        asm volatile("v_mul_f32_e32 %0, %1, %2" : "=v"(m) : "v"(scale[idx]), "v"(b1[idx % C]));
        shift2[idx] = m + shift[idx];
    }
}
//   out[b][slab][tap][o][k] = bf16(weff[tap * cp + o][b * Cin + slab * 32 + k])   (k past Cin: 0) -- the packed per-sample weight sets of gfe_conv3d_igemm
__global__ __launch_bounds__(256) void lift_fold_pack_kernel(const float* __restrict__ weff, bf16_t* __restrict__ out, int B, int Cin, int cp, int nslab, int ntaps) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x, n = (int64_t)B * nslab * ntaps * cp * 4;
    if (idx >= n) return;
    const int chunk = (int)(idx & 3);
    int64_t r = idx >> 2;
    const int o = (int)(r % cp); r /= cp;
    const int tap = (int)(r % ntaps); r /= ntaps;
    const int slab = (int)(r % nslab), b = (int)(r / nslab);
    const int i0 = slab * 32 + chunk * 8;
    const float* src = weff + ((size_t)tap * cp + o) * ((size_t)B * Cin) + (size_t)b * Cin + i0;
    float v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = i0 + k < Cin ? src[k] : 0.f;
    *reinterpret_cast<uint4*>(out + (idx >> 2) * 32 + chunk * 8) = make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7]));
}

}  // namespace

extern "C" {

int gfe_lift_fold_prep(const float* scale, const float* shift, const float* w1, const float* b1, float* rhs, float* shift2,
                       int64_t B, int64_t C, int64_t Cin, void* stream) {
    GFE_REQUIRE(scale && shift && w1 && b1 && rhs && shift2, GFE_ERR_NULL);
    GFE_REQUIRE(B > 0 && C > 0 && Cin > 0 && B * C * Cin < 0x7fffffffLL, GFE_ERR_SHAPE);
    hipLaunchKernelGGL(lift_fold_prep_kernel, dim3((unsigned)ceil_div(C * B * Cin, 256)), dim3(256), 0, (hipStream_t)stream, scale, shift, w1, b1, rhs, shift2, (int)B, (int)C, (int)Cin);
    return gfe_launch_status();
}

int gfe_lift_fold_pack(const float* weff, void* w_out, int64_t B, int64_t Cin, int64_t cout_pad, int ntaps, void* stream) {
    GFE_REQUIRE(weff && w_out, GFE_ERR_NULL);
    GFE_REQUIRE(B > 0 && Cin > 0 && cout_pad > 0 && ntaps >= 1 && ntaps <= 27, GFE_ERR_SHAPE);
    const int64_t nslab = ceil_div(Cin, 32), n = B * nslab * ntaps * cout_pad * 4;
    GFE_REQUIRE(n < 0x7fffffff00LL, GFE_ERR_SHAPE);
    hipLaunchKernelGGL(lift_fold_pack_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, weff, (bf16_t*)w_out, (int)B, (int)Cin, (int)cout_pad, (int)nslab, ntaps);
    return gfe_launch_status();
}

int gfe_groupnorm_plan(int64_t S, int* vox_per_block, int* nblk) {
    int64_t vpb = ceil_div(S, 96);
    if (vpb < 256) vpb = 256;
    *vox_per_block = (int)vpb;
    *nblk = (int)ceil_div(S, vpb);
    return GFE_OK;
}

int gfe_groupnorm_scale_shift(const void* x, const float* gamma, const float* beta, float* scale, float* shift, float* ws,
                              int64_t B, int64_t S, int64_t C, int64_t G, float eps, void* stream) {
    GFE_REQUIRE(x && gamma && beta && scale && shift && ws, GFE_ERR_NULL);
    GFE_REQUIRE(B > 0 && S > 0 && C >= 8 && C % 8 == 0 && (256 % (C / 8)) == 0 && G > 0 && C % G == 0 && G <= 256 && 2 * C <= 1024, GFE_ERR_SHAPE);
    int vpb, nblk;
    gfe_groupnorm_plan(S, &vpb, &nblk);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(gn_partial_kernel, dim3(nblk, (unsigned)B), dim3(256), (size_t)(256 / (C / 8)) * 2 * C * sizeof(float), st,
                       (const bf16_t*)x, ws, S, (int)C, vpb, nblk);
    hipLaunchKernelGGL(gn_finalize_kernel, dim3((unsigned)B), dim3(1024), (2 * C + 2 * G + 1024) * sizeof(double), st,
                       ws, gamma, beta, scale, shift, S, (int)C, (int)G, nblk, eps);
    return gfe_launch_status();
}

int gfe_groupnorm_from_partials(const float* ws, int64_t nblk, const float* gamma, const float* beta, float* scale, float* shift,
                                float* ws2, int64_t B, int64_t S, int64_t C, int64_t G, float eps, void* stream) {
    GFE_REQUIRE(ws && gamma && beta && scale && shift, GFE_ERR_NULL);
    GFE_REQUIRE(B > 0 && B <= 65535 && S > 0 && nblk > 0 && nblk <= 0x7fffffff && C > 0 && G > 0 && C % G == 0 && G <= 256 && 2 * C <= 1024, GFE_ERR_SHAPE);
    hipStream_t st = (hipStream_t)stream;
    if (nblk > 512) {
        GFE_REQUIRE(ws2, GFE_ERR_NULL);
        hipLaunchKernelGGL(gn_reduce_kernel, dim3(32, (unsigned)B), dim3(256), 0, st, ws, ws2, (int)nblk, (int)(2 * C));
        ws = ws2; nblk = 32;
    }
    hipLaunchKernelGGL(gn_finalize_kernel, dim3((unsigned)B), dim3(1024), (2 * C + 2 * G + 1024) * sizeof(double), st,
                       ws, gamma, beta, scale, shift, S, (int)C, (int)G, (int)nblk, eps);
    return gfe_launch_status();
}

int gfe_conv_in1_nblk(int64_t S) { return (int)ceil_div(S, 2048); }

int gfe_conv_in1_stats(const void* x, const float* w, const float* bias, void* y, float* stats_ws, int64_t B, int64_t S, int64_t C,
                       int in_dtype, void* stream) {
    GFE_REQUIRE(x && w && bias && y && stats_ws, GFE_ERR_NULL);
    GFE_REQUIRE(B > 0 && B <= 65535 && S > 0 && C >= 8 && C % 8 == 0 && (256 % (C / 8)) == 0 && C <= 512, GFE_ERR_SHAPE);
    const int nblk = gfe_conv_in1_nblk(S), vpb = 2048;
    const size_t lds = (2 * C + (256 / (C / 8)) * 2 * C) * sizeof(float);
    const dim3 grid((unsigned)nblk, (unsigned)B);
    if (in_dtype == GFE_F32)
        hipLaunchKernelGGL((conv_in1_stats_kernel<float>), grid, dim3(256), lds, (hipStream_t)stream, (const float*)x, w, bias, (bf16_t*)y, stats_ws, S, (int)C, vpb, nblk);
    else if (in_dtype == GFE_BF16)
        hipLaunchKernelGGL((conv_in1_stats_kernel<bf16_t>), grid, dim3(256), lds, (hipStream_t)stream, (const bf16_t*)x, w, bias, (bf16_t*)y, stats_ws, S, (int)C, vpb, nblk);
    else return GFE_ERR_DTYPE;
    return gfe_launch_status();
}

int gfe_conv3d_c1_k3_nblk(int64_t B, int64_t D, int64_t H, int64_t W) {
    (void)B;      // the block count PER SAMPLE must not depend on the batch: a block's voxels are one GroupNorm partial (batch-invariant statistics)
    const int64_t tiles = ceil_div(D, 8) * ceil_div(H, 8) * ceil_div(W, 8);
    const int64_t want = 64;                                      // 512 eight-wave blocks at the bench's 8 volumes per GPU
    return (int)(tiles < want ? tiles : want);
}

int gfe_conv3d_c1_k3(const void* x, const float* weff, const float* bias_tab, void* y, float* stats_ws, int64_t stats_nblk,
                     int64_t B, int64_t D, int64_t H, int64_t W, int64_t C, int in_dtype, int relu, void* stream) {
    GFE_REQUIRE(x && weff && bias_tab && y && stats_ws, GFE_ERR_NULL);
    GFE_REQUIRE(B > 0 && B <= 65535 && D > 0 && H > 0 && W > 0 && C == 64 && D * H * W < 0x7fffffffLL, GFE_ERR_SHAPE);
    const int nblk = gfe_conv3d_c1_k3_nblk(B, D, H, W);
    GFE_REQUIRE(stats_nblk == nblk, GFE_ERR_SHAPE);
    const dim3 grid((unsigned)nblk, (unsigned)B);
    if (in_dtype == GFE_F32)
        hipLaunchKernelGGL((conv3d_c1_k3_kernel<float>), grid, dim3(512), 0, (hipStream_t)stream, (const float*)x, weff, bias_tab, (bf16_t*)y, stats_ws, nblk, (int)D, (int)H, (int)W, relu);
    else if (in_dtype == GFE_BF16)
        hipLaunchKernelGGL((conv3d_c1_k3_kernel<bf16_t>), grid, dim3(512), 0, (hipStream_t)stream, (const bf16_t*)x, weff, bias_tab, (bf16_t*)y, stats_ws, nblk, (int)D, (int)H, (int)W, relu);
    else return GFE_ERR_DTYPE;
    return gfe_launch_status();
}

int gfe_lift_groupnorm_affine(const float* x, const float* w, const float* bias, const float* gamma, const float* beta,
                              float* scale, float* shift, double* ws, int64_t B, int64_t S, int64_t C, int64_t G, float eps, void* stream) {
    GFE_REQUIRE(x && w && bias && gamma && beta && scale && shift && ws, GFE_ERR_NULL);
    GFE_REQUIRE(B > 0 && B <= 65535 && S > 0 && C > 0 && G > 0 && G <= 256 && C % G == 0, GFE_ERR_SHAPE);
    const int nblk = 64;                                           // ws: B * 64 * 2 doubles
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(vol_moments_kernel, dim3(nblk, (unsigned)B), dim3(256), 0, st, x, ws, S, nblk);
    hipLaunchKernelGGL(lift_gn_affine_kernel, dim3((unsigned)B), dim3(256), 0, st, ws, w, bias, gamma, beta, scale, shift, S, (int)C, (int)G, nblk, eps);
    return gfe_launch_status();
}

int gfe_maxpool3d_2(const void* x, void* y, int64_t B, int64_t D, int64_t H, int64_t W, int64_t C, void* stream) {
    GFE_REQUIRE(x && y, GFE_ERR_NULL);
    GFE_REQUIRE(B > 0 && D >= 2 && H >= 2 && W >= 2 && C % 8 == 0, GFE_ERR_SHAPE);
    const int64_t total = B * (D / 2) * (H / 2) * (W / 2) * (C / 8);
    hipLaunchKernelGGL(maxpool_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)x, (bf16_t*)y, (int)B, (int)D, (int)H, (int)W, (int)C);
    return gfe_launch_status();
}

int gfe_conv_in1(const void* x, const float* w, const float* bias, void* y, int64_t nvox, int64_t C, int in_dtype, void* stream) {
    GFE_REQUIRE(x && w && bias && y, GFE_ERR_NULL);
    GFE_REQUIRE(nvox > 0 && C % 8 == 0, GFE_ERR_SHAPE);
    const unsigned g = grid_for(nvox * (C / 8));
    if (in_dtype == GFE_F32)
        hipLaunchKernelGGL((conv_in1_kernel<float>), dim3(g), dim3(256), 0, (hipStream_t)stream, (const float*)x, w, bias, (bf16_t*)y, nvox, (int)C);
    else if (in_dtype == GFE_BF16)
        hipLaunchKernelGGL((conv_in1_kernel<bf16_t>), dim3(g), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, w, bias, (bf16_t*)y, nvox, (int)C);
    else return GFE_ERR_DTYPE;
    return gfe_launch_status();
}

int gfe_conv_out1(const void* x, const float* w, float bias, float* y, int64_t nvox, int64_t C, void* stream) {
    GFE_REQUIRE(x && w && y, GFE_ERR_NULL);
    const int64_t CP = C / 8;
    GFE_REQUIRE(nvox > 0 && C % 8 == 0 && CP <= 64 && (CP & (CP - 1)) == 0, GFE_ERR_SHAPE);
    hipLaunchKernelGGL(conv_out1_kernel, dim3(grid_for(nvox * CP)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, w, bias, y, nvox, (int)C);
    return gfe_launch_status();
}

int gfe_fold_mid(const void* src, void* dst, int64_t B, int64_t D, int64_t H, int64_t W, int64_t C, int md1, int inverse, void* stream) {
    GFE_REQUIRE(src && dst, GFE_ERR_NULL);
    GFE_REQUIRE(B > 0 && md1 > 0 && D % md1 == 0 && C % 8 == 0, GFE_ERR_SHAPE);
    const int64_t total = B * D * H * W * (C / 8);
    hipLaunchKernelGGL(fold_mid_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)src, (bf16_t*)dst, (int)B, (int)D, (int)H, (int)W, (int)C, md1, inverse);
    return gfe_launch_status();
}

}  // extern "C"
