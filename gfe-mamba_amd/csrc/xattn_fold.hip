// One-query cross-attention over the image condition with the K / V projections FOLDED AWAY.
//
// Reference: cross_atten/sd_cross_atten.py:49-70, called with ONE query per sample at cross_atten/mamba_transformer.py:122-124 on the
// condition of :89-94 (rearrange 'b c h w d -> b (c d) (h w)' of [mri, pet]: key j of image m is the (h w)-vector vol_m[b, :, j]).
// The reference materialises K = W_k y + b_k and V = W_v y + b_v for all keys: 2 x (B*keys x d_cross) . (d_cross x E) = 3.6 GFLOP per
// sample forward plus two weight gradients -- 94 % of the trainable FLOPs, and the only place where the trainable half needed bf16
// operands.  With one query per sample and head h (rows h*dh .. h*dh+dh-1 of the weights):
//
//   scores   s_hj = q_h . (W_k,h y_j + b_k,h) / sqrt(dh) = (r_h . y_j) / sqrt(dh) + const_h,   r_h = W_k,h^T q_h       (const_h drops out of the softmax)
//   output   o_h  = sum_j p_hj (W_v,h y_j + b_v,h)       = W_v,h c_h + b_v,h,                  c_h = sum_j p_hj y_j    (sum_j p_hj = 1)
//
// so the pass reads the condition twice and each weight once, ~0.08 GFLOP per sample, all in exact f32 -- and the condition is read IN PLACE
// from the f32 volumes (vol_m[b] viewed as an (h w) x d matrix with d fastest): no bf16 copy, no transposed copy, no K / V tensors.
// Backward (do = gradient of the attention output, before out_proj):
//   dW_v,h += do_h (x) c_h      db_v += do       dc_h = W_v,h^T do_h       dp_hj = dc_h . y_j
//   ds_hj = p_hj (dp_hj - sum_j' p_hj' dp_hj') / sqrt(dh)                  dr_h = sum_j ds_hj y_j
//   dq_h = W_k,h dr_h           dW_k,h += q_h (x) dr_h                     db_k = 0 exactly (sum_j ds_hj = 0; the reference's value is round-off)
// The weight gradients are rank-B updates per head (one streaming read-modify-write of each gradient).  Every sum has one owner and one
// order (per-chunk partials + ordered folds): bit-reproducible.  Five launches each way on the head's dependent chain,
//   fwd: wt_vec (r) -> vol_hw (score partials) -> softmax -> vol_j (c) -> w_rows (o)
//   bwd: wt_vec (dc) -> vol_hw (dp partials) -> softmax-bwd (ds) -> vol_j (dr) -> w_rows (dq)
// and the weight gradients -- leaves of the backward: nothing reads them but the optimizer -- as two rank-B update launches that the
// caller may enqueue on a side stream (gfe_cross_attn_q1_folded_wgrad; gfe_hip.train_ops._leaf).
// All of it is bandwidth / latency bound (the condition is 7 MB per sample, the two weights 19 MB each at 96^3), so the kernels are shaped
// by bytes in flight: every lane keeps 8-16 independent 16-byte loads outstanding, and the two contractions whose operand would need a
// transposition (sum over the keys with the condition's rows running along the keys) run on the exact-f32 matrix cores
// (v_mfma_f32_16x16x4_f32: a k-ordered fmaf chain), whose A operand IS a 16-byte load along a row.
#include "common.h"

namespace {

constexpr int XB = 8;                  // samples per pass of the weight kernels (their partial results live in registers)
constexpr int XH = 8;                  // heads per pass of the scalar condition kernels

typedef __attribute__((ext_vector_type(4))) float xf_f32x4;

struct XfImgs { const float* p[4]; };

template <int VEC> struct XVec;
template <> struct XVec<4> { typedef float4 T; };
template <> struct XVec<1> { typedef float T; };
__device__ __forceinline__ float xv_get(const float4& v, int k) { return k == 0 ? v.x : k == 1 ? v.y : k == 2 ? v.z : v.w; }
__device__ __forceinline__ float xv_get(const float& v, int) { return v; }

// ---- (A) out[b][h][hw] = sum_i W[h*dh + i][hw] * v[b][h*dh + i]                                   (r = W_k^T q; dc = W_v^T do)
// Vector form: grid (ceil(HW / 256), H), 4 waves; a lane owns 4 consecutive columns, wave w the head's rows w, w + 4, ...: up to 16
// independent 16-byte loads per lane are in flight at once (the first version walked 64 rows 4 dwords at a time and ran at 1.3 TB/s),
// XB samples x 4 columns in registers, the four waves' partial sums meet in LDS in a fixed order.
__global__ __launch_bounds__(256) void xf_wt_vec4_kernel(const float* __restrict__ W, const float* __restrict__ v, float* __restrict__ out,
                                                         int B, int H, int dh, int HW, int b0) {
    extern __shared__ __attribute__((aligned(16))) float xs[];        // vs [dh][XB] | red [3][XB][256]
    const int h = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, E = H * dh;
    const int nb = min(XB, B - b0);
    float* vs = xs;
    float* red = xs + ((dh * XB + 3) & ~3);
    for (int e = tid; e < dh * XB; e += 256) {
        const int i = e / XB, b = e - i * XB;
        vs[e] = b < nb ? v[(size_t)(b0 + b) * E + h * dh + i] : 0.f;
    }
    __syncthreads();
    const int c4 = blockIdx.x * 64 + lane;                           // float4 column
    const bool ok = c4 * 4 < HW;                                     // HW % 4 == 0
    float acc[XB][4];
#pragma unroll
    for (int b = 0; b < XB; ++b)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[b][q] = 0.f;
    const float4* wp = reinterpret_cast<const float4*>(W + (size_t)h * dh * HW) + c4;
    const int hw4 = HW >> 2;
    constexpr int U = 8;
    if (ok) {
        for (int i0 = wave; i0 < dh; i0 += 4 * U) {
            float4 w[U];
#pragma unroll
            for (int k = 0; k < U; ++k) { const int i = i0 + 4 * k; if (i < dh) w[k] = wp[(size_t)i * hw4]; }
#pragma unroll
            for (int k = 0; k < U; ++k) {
                const int i = i0 + 4 * k;
                if (i < dh) {
                    const float4 v0 = *reinterpret_cast<const float4*>(vs + i * XB), v1 = *reinterpret_cast<const float4*>(vs + i * XB + 4);
                    const float vv[XB] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
                    for (int b = 0; b < XB; ++b) {
                        acc[b][0] = fmaf(w[k].x, vv[b], acc[b][0]); acc[b][1] = fmaf(w[k].y, vv[b], acc[b][1]);
                        acc[b][2] = fmaf(w[k].z, vv[b], acc[b][2]); acc[b][3] = fmaf(w[k].w, vv[b], acc[b][3]);
                    }
                }
            }
        }
    }
    if (wave > 0) {
#pragma unroll
        for (int b = 0; b < XB; ++b) *reinterpret_cast<float4*>(red + (((wave - 1) * XB + b) * 64 + lane) * 4) = make_float4(acc[b][0], acc[b][1], acc[b][2], acc[b][3]);
    }
    __syncthreads();
    if (wave == 0 && ok) {
#pragma unroll
        for (int b = 0; b < XB; ++b) {
            if (b < nb) {
                float4 s = make_float4(acc[b][0], acc[b][1], acc[b][2], acc[b][3]);
#pragma unroll
                for (int w_ = 0; w_ < 3; ++w_) {
                    const float4 t = *reinterpret_cast<const float4*>(red + ((w_ * XB + b) * 64 + lane) * 4);
                    s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
                }
                reinterpret_cast<float4*>(out + ((size_t)(b0 + b) * H + h) * HW)[c4] = s;
            }
        }
    }
}
// Scalar form (sizes / alignments the 16-byte path cannot take: test geometries): grid (ceil(HW / 256), H), a thread owns one column.
__global__ __launch_bounds__(256) void xf_wt_vec1_kernel(const float* __restrict__ W, const float* __restrict__ v, float* __restrict__ out,
                                                         int B, int H, int dh, int HW, int b0) {
    extern __shared__ __attribute__((aligned(16))) float xs[];        // [dh][XB]
    const int h = blockIdx.y, hw = blockIdx.x * 256 + threadIdx.x, E = H * dh;
    const int nb = min(XB, B - b0);
    for (int e = threadIdx.x; e < dh * XB; e += 256) {
        const int i = e / XB, b = e - i * XB;
        xs[e] = b < nb ? v[(size_t)(b0 + b) * E + h * dh + i] : 0.f;
    }
    __syncthreads();
    if (hw >= HW) return;
    float acc[XB];
#pragma unroll
    for (int b = 0; b < XB; ++b) acc[b] = 0.f;
    const float* wp = W + (size_t)h * dh * HW + hw;
    for (int i = 0; i < dh; ++i) {
        const float w = wp[(size_t)i * HW];
#pragma unroll
        for (int b = 0; b < XB; ++b) acc[b] = fmaf(w, xs[i * XB + b], acc[b]);
    }
#pragma unroll
    for (int b = 0; b < XB; ++b) if (b < nb) out[((size_t)(b0 + b) * H + h) * HW + hw] = acc[b];
}

// ---- (B) part[b][img][chunk][h][j] = sum_{hw in chunk} a[b][h][hw] * vol_img[b][hw][j]             (score partials; dp partials)
// grid (nchunk, n_img, B); thread (jg, sub): VEC consecutive j, rows sub, sub + TH, ... of the chunk; XH heads in registers.
template <int VEC>
__global__ __launch_bounds__(256) void xf_vol_hw_kernel(const XfImgs imgs, const float* __restrict__ a, float* __restrict__ part,
                                                        int H, int HW, int D3, int CH, int nchunk, int h0) {
    typedef typename XVec<VEC>::T V;
    extern __shared__ __attribute__((aligned(16))) float xs[];        // a_s [CH][XH] | red [TH][XH][D3]
    const int chunk = blockIdx.x, img = blockIdx.y, b = blockIdx.z, n_img = gridDim.y;
    const int hw0 = chunk * CH, nrow = min(CH, HW - hw0);
    const int NJ = (D3 + VEC - 1) / VEC, TH = 256 / NJ;
    float* a_s = xs;
    float* red = xs + CH * XH;
    for (int e = threadIdx.x; e < CH * XH; e += 256) {
        const int h = e / CH, r = e - h * CH;                         // coalesced reads of a's rows, transposed store
        a_s[r * XH + h] = (r < nrow && h0 + h < H) ? a[((size_t)b * H + h0 + h) * HW + hw0 + r] : 0.f;
    }
    __syncthreads();
    const int jg = threadIdx.x % NJ, sub = threadIdx.x / NJ;
    float acc[XH][VEC];
#pragma unroll
    for (int h = 0; h < XH; ++h)
#pragma unroll
        for (int k = 0; k < VEC; ++k) acc[h][k] = 0.f;
    if (sub < TH) {
        const float* vp = imgs.p[img] + ((size_t)b * HW + hw0) * D3 + jg * VEC;
        constexpr int U = 8;
        int r = sub;
        for (; r + (U - 1) * TH < nrow; r += U * TH) {
            V x[U];
#pragma unroll
            for (int k = 0; k < U; ++k) x[k] = *reinterpret_cast<const V*>(vp + (size_t)(r + k * TH) * D3);
#pragma unroll
            for (int k = 0; k < U; ++k) {
                const float4 a0 = *reinterpret_cast<const float4*>(a_s + (r + k * TH) * XH), a1 = *reinterpret_cast<const float4*>(a_s + (r + k * TH) * XH + 4);
                const float av[XH] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
#pragma unroll
                for (int h = 0; h < XH; ++h)
#pragma unroll
                    for (int q = 0; q < VEC; ++q) acc[h][q] = fmaf(av[h], xv_get(x[k], q), acc[h][q]);
            }
        }
        for (; r < nrow; r += TH) {
            const V x = *reinterpret_cast<const V*>(vp + (size_t)r * D3);
#pragma unroll
            for (int h = 0; h < XH; ++h)
#pragma unroll
                for (int q = 0; q < VEC; ++q) acc[h][q] = fmaf(a_s[r * XH + h], xv_get(x, q), acc[h][q]);
        }
#pragma unroll
        for (int h = 0; h < XH; ++h)
#pragma unroll
            for (int q = 0; q < VEC; ++q) red[((size_t)sub * XH + h) * (NJ * VEC) + jg * VEC + q] = acc[h][q];
    }
    __syncthreads();
    const int nh = min(XH, H - h0);
    for (int e = threadIdx.x; e < nh * D3; e += 256) {
        const int h = e / D3, j = e - h * D3;
        float s = 0.f;
        for (int t = 0; t < TH; ++t) s += red[((size_t)t * XH + h) * (NJ * VEC) + j];        // fixed order
        part[((((size_t)b * n_img + img) * nchunk + chunk) * H + h0 + h) * D3 + j] = s;
    }
}

// ---- (C) softmax over the keys of one (sample, head) (forward) / its backward: grid (H, B), a thread per key folds the key's chunk
// partials (four interleaved chains, fixed order), the block's waves meet in LDS
template <bool BWD>
__global__ __launch_bounds__(256) void xf_softmax_kernel(const float* __restrict__ part, float* __restrict__ p, float* __restrict__ ds,
                                                         int H, int D3, int n_img, int nchunk, float scale) {
    __shared__ float red[8];
    const int h = blockIdx.x, b = blockIdx.y, keys = n_img * D3, tid = threadIdx.x;
    float* pr = p + ((size_t)b * H + h) * keys;
    float m = -INFINITY, l = 0.f, dot = 0.f;
    float sv[4];                                                       // keys <= 1024: up to 4 per thread (static indices: registers)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int key = tid + 256 * k;
        sv[k] = 0.f;
        if (key < keys) {
            const int img = key / D3, j = key - img * D3;
            const float* pp = part + ((((size_t)b * n_img + img) * nchunk) * H + h) * D3 + j;
            const size_t cs = (size_t)H * D3;
            float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
            int c = 0;
            for (; c + 4 <= nchunk; c += 4) { s0 += pp[c * cs]; s1 += pp[(c + 1) * cs]; s2 += pp[(c + 2) * cs]; s3 += pp[(c + 3) * cs]; }
            for (; c < nchunk; ++c) s0 += pp[c * cs];
            const float s = (s0 + s1) + (s2 + s3);
            if constexpr (!BWD) { sv[k] = s * scale; m = fmaxf(m, sv[k]); }
            else { sv[k] = s; dot = fmaf(pr[key], s, dot); }
        }
    }
    auto block_red = [&](float v, bool is_max) {
        v = is_max ? wave_max(v) : wave_sum(v);
        __syncthreads();
        if ((tid & 63) == 0) red[tid >> 6] = v;
        __syncthreads();
        return is_max ? fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])) : (red[0] + red[1]) + (red[2] + red[3]);
    };
    if constexpr (!BWD) {
        m = block_red(m, true);
#pragma unroll
        for (int k = 0; k < 4; ++k) if (tid + 256 * k < keys) { sv[k] = expf(sv[k] - m); l += sv[k]; }
        l = block_red(l, false);
        const float inv = 1.0f / l;
#pragma unroll
        for (int k = 0; k < 4; ++k) if (tid + 256 * k < keys) pr[tid + 256 * k] = sv[k] * inv;
    } else {
        dot = block_red(dot, false);
        float* dr = ds + ((size_t)b * H + h) * keys;
#pragma unroll
        for (int k = 0; k < 4; ++k) if (tid + 256 * k < keys) dr[tid + 256 * k] = pr[tid + 256 * k] * (sv[k] - dot) * scale;
    }
}

// ---- (D) out[b][h][hw] = sum_{img, j} w[b][h][img * D3 + j] * vol_img[b][hw][j]                    (c = sum_j p_j y_j; dr = sum_j ds_j y_j)
// Matrix-core form (exact f32): D[m = hw][n = head] += A[m][k = key] B[k][n] with v_mfma_f32_16x16x4_f32.  Lane (lr, lq) = (lane & 15,
// lane >> 4) supplies A[m = lr][k = lq]: ONE 16-byte load of row hw0 + lr at columns j0 + 4 lq .. + 3 feeds four MFMAs (the i-th takes
// element i, i.e. k-slot lq stands for key j0 + 4 lq + i: a fixed permutation of the sum, the B operand uses the same one) -- the
// condition's rows run along the keys, so nothing is transposed and nothing passes through LDS.  B[k][n = lr]: w[b][head lr][...], all keys of
// the sample held in registers (KS float4 per lane) across the wave's row tiles.  D: lane holds rows 4 lq .. 4 lq + 3 of column (head) lr:
// one 16-byte store per head row.  grid (ceil(tiles / (4 * TPW)), B, ceil(H / 16)), 4 waves, TPW row tiles of 16 per wave.
template <int KS>      // float4 k-groups per lane: n_img * ceil(D3 / 16) <= KS
__global__ __launch_bounds__(256) void xf_vol_j_mfma_kernel(const XfImgs imgs, const float* __restrict__ w, float* __restrict__ out,
                                                            int H, int HW, int D3, int n_img, int tpw) {
    const int b = blockIdx.y, h0 = blockIdx.z * 16, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lr = lane & 15, lq = lane >> 4, keys = n_img * D3;
    const int nstep = (D3 + 15) >> 4;                                 // 16-key steps per image
    float4 bw[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const int img = s / nstep, j = (s - img * nstep) * 16 + 4 * lq;
        bw[s] = (img < n_img && j < D3 && h0 + lr < H) ? *reinterpret_cast<const float4*>(w + ((size_t)b * H + h0 + lr) * keys + img * D3 + j)
                                                        : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const int ntile = (HW + 15) >> 4;
    const int t0 = (blockIdx.x * 4 + wave) * tpw;
    for (int t = t0; t < min(ntile, t0 + tpw); ++t) {
        const int hw = t * 16 + lr;
        const bool rok = hw < HW;
        float4 a[KS];
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int img = s / nstep, j = (s - img * nstep) * 16 + 4 * lq;
            a[s] = (img < n_img && j < D3 && rok) ? *reinterpret_cast<const float4*>(imgs.p[img < n_img ? img : 0] + ((size_t)b * HW + hw) * D3 + j)
                                                  : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        xf_f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};      // two chains (a dependent f32 MFMA waits 40 cycles, an independent one 32)
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s].x, bw[s].x, acc, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s].y, bw[s].y, acc1, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s].z, bw[s].z, acc, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s].w, bw[s].w, acc1, 0, 0, 0);
        }
        acc += acc1;
        const int hwo = t * 16 + 4 * lq;                              // HW % 4 == 0: a lane's four rows are inside or outside together
        if (h0 + lr < H && hwo < HW) *reinterpret_cast<float4*>(out + ((size_t)b * H + h0 + lr) * HW + hwo) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    }
}
// Scalar form: grid (ceil(HW / 64), B), one wave per block: the 64 x D3 tile of an image (contiguous in memory) goes through LDS so that
// lane = row hw; the weights w[b][h][:] are wave-uniform (scalar operands).  Odd row pitch: conflict-free row-per-lane reads.
__global__ __launch_bounds__(64) void xf_vol_j1_kernel(const XfImgs imgs, const float* __restrict__ w, float* __restrict__ out,
                                                       int H, int HW, int D3, int n_img, int h0) {
    extern __shared__ __attribute__((aligned(16))) float xs[];        // [64][pitch]
    const int b = blockIdx.y, hw0 = blockIdx.x * 64, lane = threadIdx.x;
    const int nrow = min(64, HW - hw0);
    const int pitch = D3 | 1, keys = n_img * D3;
    const int nh = min(XH, H - h0);
    float acc[XH];
#pragma unroll
    for (int h = 0; h < XH; ++h) acc[h] = 0.f;
    for (int img = 0; img < n_img; ++img) {
        const float* vp = imgs.p[img] + ((size_t)b * HW + hw0) * D3;
        __syncthreads();                                              // the previous image's rows have been read
        for (int f = lane; f < nrow * D3; f += 64) {
            const int r = f / D3, j = f - r * D3;
            xs[r * pitch + j] = vp[f];
        }
        __syncthreads();
        const float* wp = w + ((size_t)b * H + h0) * keys + img * D3;   // wave-uniform
        const float* rowp = xs + (lane < nrow ? lane : 0) * pitch;
        for (int j = 0; j < D3; ++j) {
            const float x = rowp[j];
#pragma unroll
            for (int h = 0; h < XH; ++h) if (h < nh) acc[h] = fmaf(wp[(size_t)h * keys + j], x, acc[h]);
        }
    }
    if (lane < nrow) {
#pragma unroll
        for (int h = 0; h < XH; ++h) if (h < nh) out[((size_t)b * H + h0 + h) * HW + hw0 + lane] = acc[h];
    }
}

// ---- (E) out[b][n] = sum_hw W[n][hw] * a[b][n / dh][hw] (+ bias[n])                               (o = W_v c + b_v; dq = W_k dr)
// grid (E / R): a block owns R rows of ONE head (the a rows are loaded once for all of them: they come out of L2 E / R times in all),
// threads stride over hw in VEC units; XB samples in registers, summed wave -> LDS -> thread in a fixed order
template <int VEC, int R>
__global__ __launch_bounds__(256) void xf_w_rows_kernel(const float* __restrict__ W, const float* __restrict__ a, const float* __restrict__ bias,
                                                        float* __restrict__ out, int B, int H, int dh, int HW, int b0) {
    typedef typename XVec<VEC>::T V;
    __shared__ float red[4][R][XB];
    const int n0 = blockIdx.x * R, h = n0 / dh, E = H * dh, tid = threadIdx.x;
    const int nb = min(XB, B - b0);
    float acc[R][XB];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int b = 0; b < XB; ++b) acc[r][b] = 0.f;
    const int nv = HW / VEC;
    for (int f = tid; f < nv; f += 256) {
        V wv[R], av[XB];
#pragma unroll
        for (int r = 0; r < R; ++r) wv[r] = reinterpret_cast<const V*>(W + (size_t)(n0 + r) * HW)[f];
#pragma unroll
        for (int b = 0; b < XB; ++b) if (b < nb) av[b] = reinterpret_cast<const V*>(a + ((size_t)(b0 + b) * H + h) * HW)[f];
#pragma unroll
        for (int b = 0; b < XB; ++b) {
            if (b < nb) {
#pragma unroll
                for (int r = 0; r < R; ++r)
#pragma unroll
                    for (int q = 0; q < VEC; ++q) acc[r][b] = fmaf(xv_get(wv[r], q), xv_get(av[b], q), acc[r][b]);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int b = 0; b < XB; ++b) acc[r][b] = wave_sum(acc[r][b]);
    if ((tid & 63) == 0) {
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int b = 0; b < XB; ++b) red[tid >> 6][r][b] = acc[r][b];
    }
    __syncthreads();
    if (tid < R * XB) {
        const int r = tid / XB, b = tid - r * XB;
        if (b < nb) out[(size_t)(b0 + b) * E + n0 + r] = ((red[0][r][b] + red[1][r][b]) + (red[2][r][b] + red[3][r][b])) + (bias ? bias[n0 + r] : 0.f);
    }
}

// ---- (F) dW[n][hw] += sum_b u[b][n] * a[b][n / dh][hw]   (dW_v += do (x) c; dW_k += q (x) dr), and dbias[n] += sum_b u[b][n] by the
// blocks of the first column range.  A leaf of the backward.  grid (ceil(HW / (256 VEC)), ceil(dh / RR), H): a thread owns VEC columns,
// loads the XB samples' a values once and walks RR rows of the head -- 2 RR independent 16-byte accesses in flight
template <int VEC>
__global__ __launch_bounds__(256) void xf_rank_update_kernel(float* __restrict__ dW, const float* __restrict__ u, const float* __restrict__ a,
                                                             float* __restrict__ dbias, int B, int H, int dh, int HW, int b0) {
    typedef typename XVec<VEC>::T V;
    constexpr int RR = 8;
    __shared__ float us[RR][XB];
    const int h = blockIdx.z, i0 = blockIdx.y * RR, E = H * dh, tid = threadIdx.x;
    const int nb = min(XB, B - b0), nr = min(RR, dh - i0);
    if (tid < RR * XB) {
        const int r = tid / XB, b = tid - r * XB;
        us[r][b] = (r < nr && b < nb) ? u[(size_t)(b0 + b) * E + h * dh + i0 + r] : 0.f;
    }
    __syncthreads();
    if (dbias && blockIdx.x == 0 && tid < nr) {
        float s = 0.f;
#pragma unroll
        for (int b = 0; b < XB; ++b) s += us[tid][b];
        dbias[h * dh + i0 + tid] += s;
    }
    const int f = blockIdx.x * 256 + tid;
    if (f >= HW / VEC) return;
    V av[XB];
#pragma unroll
    for (int b = 0; b < XB; ++b) if (b < nb) av[b] = reinterpret_cast<const V*>(a + ((size_t)(b0 + b) * H + h) * HW)[f];
    V g[RR];
#pragma unroll
    for (int r = 0; r < RR; ++r) if (r < nr) g[r] = reinterpret_cast<const V*>(dW + (size_t)(h * dh + i0 + r) * HW)[f];
#pragma unroll
    for (int r = 0; r < RR; ++r) {
        if (r < nr) {
            float t[VEC];
#pragma unroll
            for (int q = 0; q < VEC; ++q) t[q] = xv_get(g[r], q);
#pragma unroll
            for (int b = 0; b < XB; ++b) {
                if (b < nb) {
#pragma unroll
                    for (int q = 0; q < VEC; ++q) t[q] = fmaf(us[r][b], xv_get(av[b], q), t[q]);
                }
            }
            V* dst = reinterpret_cast<V*>(dW + (size_t)(h * dh + i0 + r) * HW) + f;
            if constexpr (VEC == 4) *dst = make_float4(t[0], t[1], t[2], t[3]);
            else *dst = t[0];
        }
    }
}

inline int xf_chunk(int64_t HW) { return HW >= 4096 ? 256 : 64; }

struct XfShape { int B, H, dh, HW, D3, n_img; bool v4; };

void xf_wt_vec(const XfShape& s, const float* W, const float* v, float* out, hipStream_t st) {
    for (int b0 = 0; b0 < s.B; b0 += XB) {
        if (s.v4) hipLaunchKernelGGL(xf_wt_vec4_kernel, dim3((unsigned)ceil_div(s.HW, 256), (unsigned)s.H), dim3(256),
                                     (((size_t)s.dh * XB + 3) & ~(size_t)3) * sizeof(float) + 3 * XB * 256 * sizeof(float), st, W, v, out, s.B, s.H, s.dh, s.HW, b0);
        else hipLaunchKernelGGL(xf_wt_vec1_kernel, dim3((unsigned)ceil_div(s.HW, 256), (unsigned)s.H), dim3(256), (size_t)s.dh * XB * sizeof(float), st,
                                W, v, out, s.B, s.H, s.dh, s.HW, b0);
    }
}
int xf_vol_hw(const XfShape& s, const XfImgs& im, const float* a, float* part, hipStream_t st) {
    const int CH = xf_chunk(s.HW), nchunk = (int)ceil_div(s.HW, CH), VEC = s.v4 ? 4 : 1;
    const int NJ = (s.D3 + VEC - 1) / VEC, TH = 256 / NJ;
    const size_t lds = ((size_t)CH * XH + (size_t)TH * XH * NJ * VEC) * sizeof(float);
    const dim3 grid((unsigned)nchunk, (unsigned)s.n_img, (unsigned)s.B);
    for (int h0 = 0; h0 < s.H; h0 += XH) {
        if (s.v4) hipLaunchKernelGGL((xf_vol_hw_kernel<4>), grid, dim3(256), lds, st, im, a, part, s.H, s.HW, s.D3, CH, nchunk, h0);
        else hipLaunchKernelGGL((xf_vol_hw_kernel<1>), grid, dim3(256), lds, st, im, a, part, s.H, s.HW, s.D3, CH, nchunk, h0);
    }
    return nchunk;
}
void xf_vol_j(const XfShape& s, const XfImgs& im, const float* w, float* out, hipStream_t st) {
    const int ks = s.n_img * (int)ceil_div(s.D3, 16);
    if (s.v4 && ks <= 16) {
        const int ntile = (int)ceil_div(s.HW, 16);
        int tpw = 1;
        while (tpw < 8 && ceil_div(ntile, 4 * tpw) * s.B > 1024) tpw *= 2;        // enough blocks to fill the chip, the weights' fragments reused tpw times
        const dim3 grid((unsigned)ceil_div(ntile, 4 * tpw), (unsigned)s.B, (unsigned)ceil_div(s.H, 16));
        if (ks <= 4) hipLaunchKernelGGL((xf_vol_j_mfma_kernel<4>), grid, dim3(256), 0, st, im, w, out, s.H, s.HW, s.D3, s.n_img, tpw);
        else if (ks <= 12) hipLaunchKernelGGL((xf_vol_j_mfma_kernel<12>), grid, dim3(256), 0, st, im, w, out, s.H, s.HW, s.D3, s.n_img, tpw);
        else hipLaunchKernelGGL((xf_vol_j_mfma_kernel<16>), grid, dim3(256), 0, st, im, w, out, s.H, s.HW, s.D3, s.n_img, tpw);
        return;
    }
    for (int h0 = 0; h0 < s.H; h0 += XH)
        hipLaunchKernelGGL(xf_vol_j1_kernel, dim3((unsigned)ceil_div(s.HW, 64), (unsigned)s.B), dim3(64), (size_t)64 * (s.D3 | 1) * sizeof(float), st,
                           im, w, out, s.H, s.HW, s.D3, s.n_img, h0);
}
void xf_w_rows(const XfShape& s, const float* W, const float* a, const float* bias, float* out, hipStream_t st) {
    const int E = s.H * s.dh;
    for (int b0 = 0; b0 < s.B; b0 += XB) {
        if (s.v4 && s.dh % 2 == 0) hipLaunchKernelGGL((xf_w_rows_kernel<4, 2>), dim3((unsigned)(E / 2)), dim3(256), 0, st, W, a, bias, out, s.B, s.H, s.dh, s.HW, b0);
        else if (s.v4) hipLaunchKernelGGL((xf_w_rows_kernel<4, 1>), dim3((unsigned)E), dim3(256), 0, st, W, a, bias, out, s.B, s.H, s.dh, s.HW, b0);
        else hipLaunchKernelGGL((xf_w_rows_kernel<1, 1>), dim3((unsigned)E), dim3(256), 0, st, W, a, bias, out, s.B, s.H, s.dh, s.HW, b0);
    }
}
void xf_rank_update(const XfShape& s, float* dW, const float* u, const float* a, float* dbias, hipStream_t st) {
    const int VEC = s.v4 ? 4 : 1;
    const dim3 grid((unsigned)ceil_div(s.HW / VEC, 256), (unsigned)ceil_div(s.dh, 8), (unsigned)s.H);
    for (int b0 = 0; b0 < s.B; b0 += XB) {
        if (s.v4) hipLaunchKernelGGL((xf_rank_update_kernel<4>), grid, dim3(256), 0, st, dW, u, a, dbias, s.B, s.H, s.dh, s.HW, b0);
        else hipLaunchKernelGGL((xf_rank_update_kernel<1>), grid, dim3(256), 0, st, dW, u, a, dbias, s.B, s.H, s.dh, s.HW, b0);
    }
}

bool xf_shape_ok(int64_t B, int64_t H, int64_t dh, int64_t HW, int64_t D3, int n_img) {
    return B > 0 && B <= 65535 && H > 0 && H <= 64 && dh > 0 && dh <= 1024 && HW > 0 && HW < (1 << 26) && D3 > 0 && D3 <= 256 && n_img >= 1 && n_img <= 4 &&
           (int64_t)n_img * D3 <= 1024 && H * dh <= 65535;
}
bool xf_aligned(std::initializer_list<const void*> ps) {
    uintptr_t m = 0;
    for (const void* p : ps) m |= (uintptr_t)p;
    return (m & 15) == 0;
}

}  // namespace

extern "C" {

int gfe_cross_attn_q1_folded_chunks(int64_t HW) { return (int)ceil_div(HW, xf_chunk(HW)); }

int gfe_cross_attn_q1_folded_fwd(const float* q, const float* Wk, const float* Wv, const float* bv,
                                 const float* img0, const float* img1, const float* img2, const float* img3, int n_img,
                                 float* r_ws, float* part_ws, float* p, float* c, float* o,
                                 int64_t B, int64_t H, int64_t dh, int64_t HW, int64_t D3, void* stream) {
    GFE_REQUIRE(q && Wk && Wv && img0 && r_ws && part_ws && p && c && o, GFE_ERR_NULL);
    GFE_REQUIRE(xf_shape_ok(B, H, dh, HW, D3, n_img), GFE_ERR_SHAPE);
    XfImgs im = {{img0, img1, img2, img3}};
    for (int i = 0; i < n_img; ++i) GFE_REQUIRE(im.p[i], GFE_ERR_NULL);
    for (int i = n_img; i < 4; ++i) im.p[i] = img0;
    hipStream_t st = (hipStream_t)stream;
    XfShape s = {(int)B, (int)H, (int)dh, (int)HW, (int)D3, n_img,
                 HW % 4 == 0 && D3 % 4 == 0 && xf_aligned({Wk, Wv, r_ws, c, p, im.p[0], im.p[1], im.p[2], im.p[3]})};
    xf_wt_vec(s, Wk, q, r_ws, st);
    const int nchunk = xf_vol_hw(s, im, r_ws, part_ws, st);
    hipLaunchKernelGGL((xf_softmax_kernel<false>), dim3((unsigned)H, (unsigned)B), dim3(256), 0, st, part_ws, p, nullptr, (int)H, (int)D3, n_img, nchunk,
                       1.0f / sqrtf((float)dh));
    xf_vol_j(s, im, p, c, st);
    xf_w_rows(s, Wv, c, bv, o, st);
    return gfe_launch_status();
}

int gfe_cross_attn_q1_folded_bwd(const float* d_o, const float* Wk, const float* Wv,
                                 const float* img0, const float* img1, const float* img2, const float* img3, int n_img,
                                 const float* p, float* dc_ws, float* part_ws, float* ds_ws, float* dr, float* dq,
                                 int64_t B, int64_t H, int64_t dh, int64_t HW, int64_t D3, void* stream) {
    GFE_REQUIRE(d_o && Wk && Wv && img0 && p && dc_ws && part_ws && ds_ws && dr && dq, GFE_ERR_NULL);
    GFE_REQUIRE(xf_shape_ok(B, H, dh, HW, D3, n_img), GFE_ERR_SHAPE);
    XfImgs im = {{img0, img1, img2, img3}};
    for (int i = 0; i < n_img; ++i) GFE_REQUIRE(im.p[i], GFE_ERR_NULL);
    for (int i = n_img; i < 4; ++i) im.p[i] = img0;
    hipStream_t st = (hipStream_t)stream;
    XfShape s = {(int)B, (int)H, (int)dh, (int)HW, (int)D3, n_img,
                 HW % 4 == 0 && D3 % 4 == 0 && xf_aligned({Wk, Wv, dc_ws, dr, ds_ws, im.p[0], im.p[1], im.p[2], im.p[3]})};
    xf_wt_vec(s, Wv, d_o, dc_ws, st);
    const int nchunk = xf_vol_hw(s, im, dc_ws, part_ws, st);
    hipLaunchKernelGGL((xf_softmax_kernel<true>), dim3((unsigned)H, (unsigned)B), dim3(256), 0, st, part_ws, (float*)p, ds_ws, (int)H, (int)D3, n_img, nchunk,
                       1.0f / sqrtf((float)dh));
    xf_vol_j(s, im, ds_ws, dr, st);
    xf_w_rows(s, Wk, dr, nullptr, dq, st);
    return gfe_launch_status();
}

int gfe_cross_attn_q1_folded_wgrad(const float* d_o, const float* q, const float* c, const float* dr, float* dWk, float* dWv, float* dbv,
                                   int64_t B, int64_t H, int64_t dh, int64_t HW, void* stream) {
    GFE_REQUIRE(d_o && q && c && dr && dWk && dWv, GFE_ERR_NULL);
    GFE_REQUIRE(xf_shape_ok(B, H, dh, HW, 4, 1), GFE_ERR_SHAPE);
    hipStream_t st = (hipStream_t)stream;
    XfShape s = {(int)B, (int)H, (int)dh, (int)HW, 4, 1, HW % 4 == 0 && xf_aligned({c, dr, dWk, dWv})};
    xf_rank_update(s, dWv, d_o, c, dbv, st);
    xf_rank_update(s, dWk, q, dr, nullptr, st);
    return gfe_launch_status();
}

}  // extern "C"
