// The small operators of the trainable head (Cross_mamba_both), forward and backward, f32: everything between the Mamba stack's
// GEMMs that the reference runs as strings of elementwise / softmax / LayerNorm torch ops.  All launch-bound sizes (B rows of 512):
// one launch per operator and direction instead of 5-20 ATen launches.
//
//   embed_tokens   cross_atten/mamba_transformer.py:97-117   offset add + Embedding gather + NumericalEmbedder + cls + concat
//   mean_tokens    mamba_transformer.py:122                  mean over the token axis
//   cross_attn_q1  cross_atten/sd_cross_atten.py:58-68       one query per sample against Lkv keys, H heads: q k^T / sqrt(dh), softmax, @ v
//   layernorm      mamba_transformer.py:79-82, corss_ft_transformer.py:16   nn.LayerNorm(dim) over (rows, dim)
//   geglu          corss_ft_transformer.py:10-13, 19         x * gelu(gates) (exact erf GELU) followed by Dropout(p)
//   bce_sigmoid    classify_mamba.py:104                     BCELoss(sigmoid(logit), y), mean over the batch, log clamped at -100
#include "common.h"
#include "attn_drop.h"

namespace {

// ---- embed_tokens --------------------------------------------------------------------------------------------------------
// out[b] = [cls | emb[x_cat[b, j] + offset[j]] (j < ncat) | x_num[b, j] * w[j] + bias[j] (j < ncont) | feat[b] (nf rows)]
__global__ __launch_bounds__(256) void embed_tokens_fwd_kernel(const int64_t* __restrict__ x_cat, const int64_t* __restrict__ offsets,
                                                               const float* __restrict__ emb, const float* __restrict__ x_num,
                                                               const float* __restrict__ num_w, const float* __restrict__ num_b,
                                                               const float* __restrict__ cls, const float* __restrict__ feat, float* __restrict__ out,
                                                               int ncat, int ncont, int nf, int dim, int ntok) {
    const int L = 1 + ncat + ncont + nf;
    const int b = blockIdx.y, t = blockIdx.x;
    float* o = out + ((size_t)b * L + t) * dim;
    if (t == 0) {
        for (int d = threadIdx.x; d < dim; d += 256) o[d] = cls[d];
    } else if (t <= ncat) {
        const int j = t - 1;
        int64_t idx = x_cat[(size_t)b * ncat + j] + offsets[j];
        idx = idx < 0 ? 0 : (idx >= ntok ? ntok - 1 : idx);              // (torch raises on an out-of-range index; never on valid tables)
        for (int d = threadIdx.x; d < dim; d += 256) o[d] = emb[(size_t)idx * dim + d];
    } else if (t <= ncat + ncont) {
        const int j = t - 1 - ncat;
        const float xv = x_num[(size_t)b * ncont + j];
        for (int d = threadIdx.x; d < dim; d += 256) o[d] = fmaf(xv, num_w[(size_t)j * dim + d], num_b[(size_t)j * dim + d]);
    } else {
        const int j = t - 1 - ncat - ncont;
        for (int d = threadIdx.x; d < dim; d += 256) o[d] = feat[((size_t)b * nf + j) * dim + d];
    }
}

// gradients: one block per token slot t, threads over dim, loop over the batch in order: every sum has ONE owner thread and a fixed order.
// The embedding rows of slot j (rows offset[j] .. offset[j+1]-1 of the table: each categorical column has its own range,
// mamba_transformer.py:44-51) are touched by slot j's block only, and element d of each by thread d only, so the scatter is a plain
// read-modify-write in batch order (round 3: f32 atomics).  Out-of-range indices are clamped as in the forward (torch raises instead).
__global__ __launch_bounds__(256) void embed_tokens_bwd_kernel(const float* __restrict__ dout, const int64_t* __restrict__ x_cat,
                                                               const int64_t* __restrict__ offsets, const float* __restrict__ x_num,
                                                               float* __restrict__ d_emb, float* __restrict__ d_num_w, float* __restrict__ d_num_b,
                                                               float* __restrict__ d_cls, float* __restrict__ d_feat,
                                                               int B, int ncat, int ncont, int nf, int dim, int ntok) {
    const int L = 1 + ncat + ncont + nf;
    const int t = blockIdx.x;
    for (int d = threadIdx.x; d < dim; d += 256) {
        if (t == 0) {
            float s = 0.f;
            for (int b = 0; b < B; ++b) s += dout[((size_t)b * L) * dim + d];
            d_cls[d] += s;
        } else if (t <= ncat) {
            const int j = t - 1;
            for (int b = 0; b < B; ++b) {
                int64_t idx = x_cat[(size_t)b * ncat + j] + offsets[j];
                idx = idx < 0 ? 0 : (idx >= ntok ? ntok - 1 : idx);
                d_emb[(size_t)idx * dim + d] += dout[((size_t)b * L + t) * dim + d];
            }
        } else if (t <= ncat + ncont) {
            const int j = t - 1 - ncat;
            float sw = 0.f, sb = 0.f;
            for (int b = 0; b < B; ++b) {
                const float g = dout[((size_t)b * L + t) * dim + d];
                sw = fmaf(g, x_num[(size_t)b * ncont + j], sw);
                sb += g;
            }
            d_num_w[(size_t)j * dim + d] += sw;
            d_num_b[(size_t)j * dim + d] += sb;
        } else if (d_feat) {
            const int j = t - 1 - ncat - ncont;
            for (int b = 0; b < B; ++b) d_feat[((size_t)b * nf + j) * dim + d] = dout[((size_t)b * L + t) * dim + d];
        }
    }
}

// ---- mean over tokens ----------------------------------------------------------------------------------------------------
// block = (64 columns, sample): the four waves take every fourth token (four loads in flight each) and are folded in wave order -- one owner
// and one summation order per element.  (Round 3's one-thread-per-column loop walked the L tokens as L dependent round trips: 19 us at 37.)
__global__ __launch_bounds__(256) void mean_tokens_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int L, int dim) {
    __shared__ float part[4][64];
    const int b = blockIdx.y, wave = threadIdx.x >> 6, lane = threadIdx.x & 63, d = blockIdx.x * 64 + lane;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (d < dim) {
        const float* xb = x + (size_t)b * L * dim + d;
        int t = wave;
        for (; t + 12 < L; t += 16) {
            a0 += xb[(size_t)t * dim]; a1 += xb[(size_t)(t + 4) * dim]; a2 += xb[(size_t)(t + 8) * dim]; a3 += xb[(size_t)(t + 12) * dim];
        }
        for (; t < L; t += 4) a0 += xb[(size_t)t * dim];
    }
    part[wave][lane] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (wave == 0 && d < dim) y[(size_t)b * dim + d] = ((part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane])) / (float)L;
}
__global__ __launch_bounds__(256) void mean_tokens_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int L, int dim) {
    const int b = blockIdx.y, t = blockIdx.x;
    for (int d = threadIdx.x; d < dim; d += 256) dx[((size_t)b * L + t) * dim + d] = dy[(size_t)b * dim + d] / (float)L;
}

// ---- cross attention with ONE query per sample -------------------------------------------------------------------------------
// block = (head, sample), 256 threads.  scores over nk keys, softmax, out = sum_k p_k v_k.  probs (B, H, nk) kept for the backward.
__global__ __launch_bounds__(256) void cross_attn_q1_fwd_kernel(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
                                                                float* __restrict__ out, float* __restrict__ probs, int H, int nk, int dh, float scale) {
    extern __shared__ float sm[];                 // [nk] scores / probabilities, [16] reduction scratch
    float* sp = sm;
    float* scratch = sm + nk;
    const int h = blockIdx.x, b = blockIdx.y, dim = H * dh;
    const float* qp = q + (size_t)b * dim + h * dh;
    float mx = -INFINITY;
    const bool vec4 = ((dh | dim) & 3) == 0 && (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v) & 15) == 0;     // 16-byte rows: four dims per load
    for (int j = threadIdx.x; j < nk; j += 256) {
        const float* kp = k + ((size_t)b * nk + j) * dim + h * dh;
        float s = 0.f;
        if (vec4) {
            float s1 = 0.f;
            for (int d = 0; d < dh; d += 4) {
                const float4 qv = *reinterpret_cast<const float4*>(qp + d), kv = *reinterpret_cast<const float4*>(kp + d);
                s = fmaf(qv.x, kv.x, s); s1 = fmaf(qv.y, kv.y, s1); s = fmaf(qv.z, kv.z, s); s1 = fmaf(qv.w, kv.w, s1);
            }
            s += s1;
        } else {
            for (int d = 0; d < dh; ++d) s = fmaf(qp[d], kp[d], s);
        }
        s *= scale;
        sp[j] = s;
        mx = fmaxf(mx, s);
    }
    mx = wave_max(mx);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(scratch[0], scratch[1]), fmaxf(scratch[2], scratch[3]));
    float sum = 0.f;
    for (int j = threadIdx.x; j < nk; j += 256) {
        const float e = __expf(sp[j] - mx);
        sp[j] = e;
        sum += e;
    }
    sum = block_sum(sum, scratch + 4);
    const float inv = 1.0f / sum;
    for (int j = threadIdx.x; j < nk; j += 256) {
        const float pj = sp[j] * inv;
        sp[j] = pj;
        probs[((size_t)b * H + h) * nk + j] = pj;
    }
    __syncthreads();
    // out = sum_j p_j v_j: the four waves take every fourth key (four chains each) and are folded in wave order (round 3: dh threads walked
    // all nk keys one dependent load after the other -- 19 of the kernel's 25 us at 192 keys)
    float* po = scratch + 32;                     // [4][dh]
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int d = lane; d < dh; d += 64) {
        const float* vp = v + (size_t)b * nk * dim + h * dh + d;
        float o0 = 0.f, o1 = 0.f, o2 = 0.f, o3 = 0.f;
        int j = wave;
        for (; j + 12 < nk; j += 16) {
            o0 = fmaf(sp[j], vp[(size_t)j * dim], o0); o1 = fmaf(sp[j + 4], vp[(size_t)(j + 4) * dim], o1);
            o2 = fmaf(sp[j + 8], vp[(size_t)(j + 8) * dim], o2); o3 = fmaf(sp[j + 12], vp[(size_t)(j + 12) * dim], o3);
        }
        for (; j < nk; j += 4) o0 = fmaf(sp[j], vp[(size_t)j * dim], o0);
        po[wave * dh + d] = (o0 + o1) + (o2 + o3);
    }
    __syncthreads();
    for (int d = threadIdx.x; d < dh; d += 256) out[(size_t)b * dim + h * dh + d] = (po[d] + po[dh + d]) + (po[2 * dh + d] + po[3 * dh + d]);
}

// dv = p dout^T, dp = dout . v, ds = p (dp - sum p dp), dq = scale * sum ds k, dk = scale * ds q
__global__ __launch_bounds__(256) void cross_attn_q1_bwd_kernel(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
                                                                const float* __restrict__ probs, const float* __restrict__ dout,
                                                                float* __restrict__ dq, float* __restrict__ dk, float* __restrict__ dv,
                                                                int H, int nk, int dh, float scale) {
    extern __shared__ float sm[];                 // [nk] ds, [16] scratch
    float* sds = sm;
    float* scratch = sm + nk;
    const int h = blockIdx.x, b = blockIdx.y, dim = H * dh;
    const float* qp = q + (size_t)b * dim + h * dh;
    const float* dop = dout + (size_t)b * dim + h * dh;
    const float* pp = probs + ((size_t)b * H + h) * nk;
    float acc = 0.f;
    const bool vec4 = ((dh | dim) & 3) == 0 && (((uintptr_t)dout | (uintptr_t)v) & 15) == 0;
    for (int j = threadIdx.x; j < nk; j += 256) {
        const float* vp = v + ((size_t)b * nk + j) * dim + h * dh;
        float dp = 0.f;
        if (vec4) {
            float dp1 = 0.f;
            for (int d = 0; d < dh; d += 4) {
                const float4 gv = *reinterpret_cast<const float4*>(dop + d), vv = *reinterpret_cast<const float4*>(vp + d);
                dp = fmaf(gv.x, vv.x, dp); dp1 = fmaf(gv.y, vv.y, dp1); dp = fmaf(gv.z, vv.z, dp); dp1 = fmaf(gv.w, vv.w, dp1);
            }
            dp += dp1;
        } else {
            for (int d = 0; d < dh; ++d) dp = fmaf(dop[d], vp[d], dp);
        }
        sds[j] = dp;
        acc = fmaf(pp[j], dp, acc);
    }
    acc = block_sum(acc, scratch);
    for (int j = threadIdx.x; j < nk; j += 256) sds[j] = pp[j] * (sds[j] - acc) * scale;
    __syncthreads();
    // dk, dv rows (each thread: one (key, d) element at a time, coalesced over d)
    for (int i = threadIdx.x; i < nk * dh; i += 256) {
        const int j = i / dh, d = i - j * dh;
        const size_t o = ((size_t)b * nk + j) * dim + h * dh + d;
        dk[o] = sds[j] * qp[d];
        dv[o] = pp[j] * dop[d];
    }
    float* pq = scratch + 32;                     // [4][dh]: dq partials of the four waves (every fourth key each), folded in wave order
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int d = lane; d < dh; d += 64) {
        const float* kp = k + (size_t)b * nk * dim + h * dh + d;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int j = wave;
        for (; j + 12 < nk; j += 16) {
            s0 = fmaf(sds[j], kp[(size_t)j * dim], s0); s1 = fmaf(sds[j + 4], kp[(size_t)(j + 4) * dim], s1);
            s2 = fmaf(sds[j + 8], kp[(size_t)(j + 8) * dim], s2); s3 = fmaf(sds[j + 12], kp[(size_t)(j + 12) * dim], s3);
        }
        for (; j < nk; j += 4) s0 = fmaf(sds[j], kp[(size_t)j * dim], s0);
        pq[wave * dh + d] = (s0 + s1) + (s2 + s3);
    }
    __syncthreads();
    for (int d = threadIdx.x; d < dh; d += 256) dq[(size_t)b * dim + h * dh + d] = (pq[d] + pq[dh + d]) + (pq[2 * dh + d] + pq[3 * dh + d]);
}


// counter-based uniform in [0, 1): one 32-bit hash of (seed, element index) -- reproducible from (seed, index) alone, so the backward
// needs no stored mask and every rank / step draws its own stream by changing the seed
__device__ __forceinline__ float hash_uniform(uint64_t seed, uint32_t i) {
    uint64_t z = seed + 0x9E3779B97F4A7C15ull * (uint64_t)(i + 1);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    return (float)(uint32_t)(z >> 40) * (1.0f / 16777216.0f);
}
// ---- small self-attention (Jamba's AttentionSDPA, cross_atten/jamba.py:342-398: F.scaled_dot_product_attention, is_causal) ---------
// block = (head, sample); q, k, v rows of one head staged in LDS (L <= 64, dh <= 64); probs (B, H, L, L) kept for the backward.
// p_drop > 0: dropout on the attention probabilities (the generator's ViT in training: `attn = self.dropout(attn)`, vit_pytorch_diy/vit.py:59) --
// the kept probabilities are scaled by 1 / (1 - p); the mask is a counter-based hash of (seed, element), regenerated in the backward; `probs`
// keeps the UNdropped softmax (what the softmax backward needs).
__global__ __launch_bounds__(256) void sdpa_small_fwd_kernel(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
                                                             float* __restrict__ out, float* __restrict__ probs, int H, int L, int dh, float scale, int causal,
                                                             float p_drop, uint64_t seed) {
    extern __shared__ float sm[];
    float* sq = sm; float* sk = sq + L * dh; float* sv = sk + L * dh; float* sp = sv + L * dh;      // sp: [L][L]
    const int h = blockIdx.x, b = blockIdx.y, dim = H * dh;
    for (int i = threadIdx.x; i < L * dh; i += 256) {
        const int t = i / dh, d = i - t * dh;
        const size_t o = ((size_t)b * L + t) * dim + h * dh + d;
        sq[i] = q[o]; sk[i] = k[o]; sv[i] = v[o];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < L * L; i += 256) {
        const int r = i / L, c = i - r * L;
        float s = -INFINITY;
        if (!causal || c <= r) {
            s = 0.f;
            for (int d = 0; d < dh; ++d) s = fmaf(sq[r * dh + d], sk[c * dh + d], s);
            s *= scale;
        }
        sp[i] = s;
    }
    __syncthreads();
    for (int r = threadIdx.x; r < L; r += 256) {                    // one row per thread (L <= 64)
        float mx = -INFINITY;
        for (int c = 0; c < L; ++c) mx = fmaxf(mx, sp[r * L + c]);
        float sum = 0.f;
        for (int c = 0; c < L; ++c) { const float e = __expf(sp[r * L + c] - mx); sp[r * L + c] = e; sum += e; }
        const float inv = 1.0f / sum;
        const float keep = 1.0f / (1.0f - p_drop);
        for (int c = 0; c < L; ++c) {
            const size_t gi = (((size_t)b * H + h) * L + r) * L + c;
            float pv = sp[r * L + c] * inv;
            probs[gi] = pv;
            if (p_drop > 0.f) pv = hash_uniform(seed, (uint32_t)gi) < p_drop ? 0.f : pv * keep;
            sp[r * L + c] = pv;
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < L * dh; i += 256) {
        const int r = i / dh, d = i - r * dh;
        float o = 0.f;
        for (int c = 0; c < L; ++c) o = fmaf(sp[r * L + c], sv[c * dh + d], o);
        out[((size_t)b * L + r) * dim + h * dh + d] = o;
    }
}
__global__ __launch_bounds__(256) void sdpa_small_bwd_kernel(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
                                                             const float* __restrict__ probs, const float* __restrict__ dout,
                                                             float* __restrict__ dq, float* __restrict__ dk, float* __restrict__ dv,
                                                             int H, int L, int dh, float scale, float p_drop, uint64_t seed) {
    extern __shared__ float sm[];
    float* sq = sm; float* sk = sq + L * dh; float* sv = sk + L * dh; float* so = sv + L * dh; float* sp = so + L * dh; float* sd = sp + L * L;
    float* smk = sd + L * L;                                         // dropout factor per probability: 0 or 1 / (1 - p) (1 without dropout)
    const int h = blockIdx.x, b = blockIdx.y, dim = H * dh;
    for (int i = threadIdx.x; i < L * dh; i += 256) {
        const int t = i / dh, d = i - t * dh;
        const size_t o = ((size_t)b * L + t) * dim + h * dh + d;
        sq[i] = q[o]; sk[i] = k[o]; sv[i] = v[o]; so[i] = dout[o];
    }
    for (int i = threadIdx.x; i < L * L; i += 256) {
        const size_t gi = ((size_t)b * H + h) * L * L + i;
        sp[i] = probs[gi];
        smk[i] = p_drop > 0.f ? (hash_uniform(seed, (uint32_t)gi) < p_drop ? 0.f : 1.0f / (1.0f - p_drop)) : 1.f;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < L * L; i += 256) {                 // dP = (dO V^T) * dropout factor
        const int r = i / L, c = i - r * L;
        float s = 0.f;
        for (int d = 0; d < dh; ++d) s = fmaf(so[r * dh + d], sv[c * dh + d], s);
        sd[i] = s * smk[i];
    }
    __syncthreads();
    for (int r = threadIdx.x; r < L; r += 256) {                     // dS = P (dP - sum_c P dP) * scale
        float acc = 0.f;
        for (int c = 0; c < L; ++c) acc = fmaf(sp[r * L + c], sd[r * L + c], acc);
        for (int c = 0; c < L; ++c) sd[r * L + c] = sp[r * L + c] * (sd[r * L + c] - acc) * scale;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < L * dh; i += 256) {
        const int t = i / dh, d = i - t * dh;
        float gq = 0.f, gk = 0.f, gv = 0.f;
        for (int c = 0; c < L; ++c) {
            gq = fmaf(sd[t * L + c], sk[c * dh + d], gq);            // dQ_t = sum_c dS[t][c] K_c
            gk = fmaf(sd[c * L + t], sq[c * dh + d], gk);            // dK_t = sum_r dS[r][t] Q_r
            gv = fmaf(sp[c * L + t] * smk[c * L + t], so[c * dh + d], gv);   // dV_t = sum_r dropped P[r][t] dO_r
        }
        const size_t o = ((size_t)b * L + t) * dim + h * dh + d;
        dq[o] = gq; dk[o] = gk; dv[o] = gv;
    }
}

// ---- LayerNorm over (rows, dim): one block per row ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ln_rows_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          float* __restrict__ y, float* __restrict__ mean, float* __restrict__ rstd, int dim, float eps) {
    __shared__ float scratch[32];
    const int r = blockIdx.x;
    const float* xp = x + (size_t)r * dim;
    float s = 0.f;
    for (int d = threadIdx.x; d < dim; d += 256) s += xp[d];
    const float mu = block_sum(s, scratch) / (float)dim;
    float v = 0.f;
    for (int d = threadIdx.x; d < dim; d += 256) { const float c = xp[d] - mu; v = fmaf(c, c, v); }
    const float var = block_sum(v, scratch + 16) / (float)dim;            // biased variance (torch.nn.LayerNorm)
    const float rs = rsqrtf(var + eps);
    for (int d = threadIdx.x; d < dim; d += 256) y[(size_t)r * dim + d] = fmaf((xp[d] - mu) * rs, gamma[d], beta[d]);
    if (threadIdx.x == 0) { mean[r] = mu; rstd[r] = rs; }
}
// dx = rstd * (g - mean(g) - xhat * mean(g * xhat)), g = dy * gamma; dgamma += sum_rows dy * xhat, dbeta += sum_rows dy.
// Blocks 0 .. rows-1 own a row each (dx); blocks rows .. rows + ceil(dim/64) - 1 own 64 COLUMNS each: their four waves split the rows,
// fold through LDS in wave order and add to dgamma / dbeta with a plain read-modify-write -- one owner and one summation order per
// element, no atomics (rows < LN_TALL_MIN_ROWS here; round 3 added every row's term with an f32 atomic).
__global__ __launch_bounds__(256) void ln_rows_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ mean,
                                                          const float* __restrict__ rstd, const float* __restrict__ dy, float* __restrict__ dx,
                                                          float* __restrict__ dgamma, float* __restrict__ dbeta, int rows, int dim) {
    __shared__ float scratch[32];
    __shared__ float colred[2][4][64];
    if ((int)blockIdx.x >= rows) {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        const int d = ((int)blockIdx.x - rows) * 64 + lane;
        float ag = 0.f, ab = 0.f;
        if (d < dim)
            for (int r = wave; r < rows; r += 4) {
                const float g = dy[(size_t)r * dim + d];
                ag = fmaf(g, (x[(size_t)r * dim + d] - mean[r]) * rstd[r], ag);
                ab += g;
            }
        colred[0][wave][lane] = ag; colred[1][wave][lane] = ab;
        __syncthreads();
        if (wave == 0 && d < dim) {
            dgamma[d] += (colred[0][0][lane] + colred[0][1][lane]) + (colred[0][2][lane] + colred[0][3][lane]);
            dbeta[d] += (colred[1][0][lane] + colred[1][1][lane]) + (colred[1][2][lane] + colred[1][3][lane]);
        }
        return;
    }
    const int r = blockIdx.x;
    const float mu = mean[r], rs = rstd[r];
    const float* xp = x + (size_t)r * dim;
    const float* gp = dy + (size_t)r * dim;
    float s1 = 0.f, s2 = 0.f;
    for (int d = threadIdx.x; d < dim; d += 256) {
        const float xh = (xp[d] - mu) * rs, g = gp[d] * gamma[d];
        s1 += g; s2 = fmaf(g, xh, s2);
    }
    s1 = block_sum(s1, scratch) / (float)dim;
    s2 = block_sum(s2, scratch + 16) / (float)dim;
    for (int d = threadIdx.x; d < dim; d += 256) {
        const float xh = (xp[d] - mu) * rs, g = gp[d] * gamma[d];
        dx[(size_t)r * dim + d] = rs * (g - s1 - xh * s2);
    }
}

// ---- LayerNorm backward over MANY rows (the 3-D ViT's 13,832 token rows of 512: one block per row with two atomics per element onto 2 * dim
// addresses took 314 us per call).  A wave owns a row at a time (row sums by shuffles, no barrier), keeps its share of dgamma / dbeta in
// registers over the block's rows, the block's four waves combine through LDS into ONE partial row per block, and a second launch adds the
// partials to dgamma / dbeta in a fixed order: no atomics, run-to-run identical.  dim <= 2048 (8 float4 per lane).
constexpr int LN_TALL_MIN_ROWS = 256, LN_TALL_MAX_DIM = 2048, LN_TALL_MAX_BLOCKS = 1024;
template <int NI>
__global__ __launch_bounds__(256) void ln_tall_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ mean,
                                                          const float* __restrict__ rstd, const float* __restrict__ dy, float* __restrict__ dx,
                                                          float* __restrict__ part, int rows, int dim, int rows_per_block) {
    __shared__ float4 red[2][3][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r0 = blockIdx.x * rows_per_block, r1 = min(rows, r0 + rows_per_block);
    float4 gm[NI], dg[NI], db[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int c = 4 * lane + 256 * i;
        gm[i] = c < dim ? *reinterpret_cast<const float4*>(gamma + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        dg[i] = make_float4(0.f, 0.f, 0.f, 0.f); db[i] = dg[i];
    }
    const float inv_dim = 1.0f / (float)dim;
    for (int r = r0 + wave; r < r1; r += 4) {
        const float mu = mean[r], rs = rstd[r];
        float4 xh[NI], g[NI];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int c = 4 * lane + 256 * i;
            float4 xv = make_float4(mu, mu, mu, mu), dv = make_float4(0.f, 0.f, 0.f, 0.f);
            if (c < dim) { xv = *reinterpret_cast<const float4*>(x + (size_t)r * dim + c); dv = *reinterpret_cast<const float4*>(dy + (size_t)r * dim + c); }
            xh[i] = make_float4((xv.x - mu) * rs, (xv.y - mu) * rs, (xv.z - mu) * rs, (xv.w - mu) * rs);
            g[i] = make_float4(dv.x * gm[i].x, dv.y * gm[i].y, dv.z * gm[i].z, dv.w * gm[i].w);
            s1 += (g[i].x + g[i].y) + (g[i].z + g[i].w);
            s2 += (g[i].x * xh[i].x + g[i].y * xh[i].y) + (g[i].z * xh[i].z + g[i].w * xh[i].w);
            dg[i].x = fmaf(dv.x, xh[i].x, dg[i].x); dg[i].y = fmaf(dv.y, xh[i].y, dg[i].y); dg[i].z = fmaf(dv.z, xh[i].z, dg[i].z); dg[i].w = fmaf(dv.w, xh[i].w, dg[i].w);
            db[i].x += dv.x; db[i].y += dv.y; db[i].z += dv.z; db[i].w += dv.w;
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
        s1 *= inv_dim; s2 *= inv_dim;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int c = 4 * lane + 256 * i;
            if (c < dim)
                *reinterpret_cast<float4*>(dx + (size_t)r * dim + c) = make_float4(rs * (g[i].x - s1 - xh[i].x * s2), rs * (g[i].y - s1 - xh[i].y * s2),
                                                                                  rs * (g[i].z - s1 - xh[i].z * s2), rs * (g[i].w - s1 - xh[i].w * s2));
        }
    }
    // the block's partial row: waves 1..3 park their sums, wave 0 adds them in order and writes part[block][0 | 1][dim]
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        if (wave > 0) { red[0][wave - 1][lane] = dg[i]; red[1][wave - 1][lane] = db[i]; }
        __syncthreads();
        if (wave == 0) {
            const int c = 4 * lane + 256 * i;
            float4 a = dg[i], b = db[i];
#pragma unroll
            for (int w = 0; w < 3; ++w) {
                const float4 pa = red[0][w][lane], pb = red[1][w][lane];
                a.x += pa.x; a.y += pa.y; a.z += pa.z; a.w += pa.w; b.x += pb.x; b.y += pb.y; b.z += pb.z; b.w += pb.w;
            }
            if (c < dim) {
                *reinterpret_cast<float4*>(part + ((size_t)blockIdx.x * 2) * dim + c) = a;
                *reinterpret_cast<float4*>(part + ((size_t)blockIdx.x * 2 + 1) * dim + c) = b;
            }
        }
        __syncthreads();
    }
}
// part: [nblk][2 * dim] (dgamma | dbeta partial rows).  A block owns a strip of 64 columns: 16 float4 column quads x 16 row groups; every row
// group sums its rows in order, the 16 group sums are added in order through LDS -> one fixed summation tree.
__global__ __launch_bounds__(256) void ln_tall_bwd_reduce_kernel(const float* __restrict__ part, float* __restrict__ dgamma, float* __restrict__ dbeta, int dim, int nblk) {
    __shared__ float4 red[16][16];
    const int cq = threadIdx.x & 15, rg = threadIdx.x >> 4;
    const int c = blockIdx.x * 64 + 4 * cq;                  // column of [0, 2 * dim)
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c < 2 * dim) {
        const int per = (nblk + 15) / 16, b0 = rg * per, b1 = min(nblk, b0 + per);
#pragma unroll 8
        for (int b = b0; b < b1; ++b) {
            const float4 v = *reinterpret_cast<const float4*>(part + (size_t)b * 2 * dim + c);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
    }
    red[rg][cq] = s;
    __syncthreads();
    if (rg == 0 && c < 2 * dim) {
        float4 t = red[0][cq];
#pragma unroll
        for (int g = 1; g < 16; ++g) { const float4 v = red[g][cq]; t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w; }
        float* out = c < dim ? dgamma + c : dbeta + (c - dim);      // dim % 4 == 0: a quad never straddles the two halves
        out[0] += t.x; out[1] += t.y; out[2] += t.z; out[3] += t.w;
    }
}

// ---- LayerNorm over LONG rows (the generator ViT's LayerNorm(patch_dim), patch_dim = 262,144 at 128^3, 64 rows): one block per row
// leaves 192 CUs idle and walks 1 MB three times from one CU (1.8 ms backward).  A row is cut into `splits` segments, grid (splits, rows):
//   moments: per-segment mean and centred second moment (two local passes, the segment stays in L2), combined with Chan's formula
//   apply:   every block recombines the <= 64 partials of its row in double and normalises its segment
__global__ __launch_bounds__(256) void ln_long_moments_kernel(const float* __restrict__ x, float* __restrict__ part, int dim, int seg) {
    __shared__ float scratch[32];
    const int r = blockIdx.y, sp = blockIdx.x, d0 = sp * seg, d1 = min(dim, d0 + seg);
    const float* xp = x + (size_t)r * dim;
    float s = 0.f;
    for (int d = d0 + threadIdx.x; d < d1; d += 256) s += xp[d];
    const float mu = block_sum(s, scratch) / (float)(d1 - d0);
    float v = 0.f;
    for (int d = d0 + threadIdx.x; d < d1; d += 256) { const float c = xp[d] - mu; v = fmaf(c, c, v); }
    v = block_sum(v, scratch + 16);
    if (threadIdx.x == 0) { part[((size_t)r * gridDim.x + sp) * 2] = mu; part[((size_t)r * gridDim.x + sp) * 2 + 1] = v; }
}
__device__ __forceinline__ void ln_long_combine(const float* __restrict__ part, int r, int splits, int dim, int seg, float eps, float& mu_o, float& rs_o) {
    double n = 0.0, mean = 0.0, m2 = 0.0;
    for (int k = 0; k < splits; ++k) {
        const double nk = (double)(min(dim, (k + 1) * seg) - k * seg), mk = part[((size_t)r * splits + k) * 2], vk = part[((size_t)r * splits + k) * 2 + 1];
        const double delta = mk - mean, tot = n + nk;
        m2 += vk + delta * delta * n * nk / tot;
        mean += delta * nk / tot;
        n = tot;
    }
    mu_o = (float)mean;
    rs_o = (float)(1.0 / sqrt(m2 / n + (double)eps));                  // biased variance (torch.nn.LayerNorm)
}
__global__ __launch_bounds__(256) void ln_long_apply_kernel(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            const float* __restrict__ part, float* __restrict__ y, float* __restrict__ mean,
                                                            float* __restrict__ rstd, int dim, int seg, float eps) {
    const int r = blockIdx.y, sp = blockIdx.x, d0 = sp * seg, d1 = min(dim, d0 + seg);
    float mu, rs;
    ln_long_combine(part, r, gridDim.x, dim, seg, eps, mu, rs);
    const float* xp = x + (size_t)r * dim;
    for (int d = d0 + threadIdx.x; d < d1; d += 256) y[(size_t)r * dim + d] = fmaf((xp[d] - mu) * rs, gamma[d], beta[d]);
    if (sp == 0 && threadIdx.x == 0) { mean[r] = mu; rstd[r] = rs; }
}
// backward: segment sums of g = dy * gamma and g * xhat, then dx / dgamma / dbeta per segment
__global__ __launch_bounds__(256) void ln_long_bwd_sums_kernel(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ mean,
                                                               const float* __restrict__ rstd, const float* __restrict__ dy, float* __restrict__ part,
                                                               int dim, int seg) {
    __shared__ float scratch[32];
    const int r = blockIdx.y, sp = blockIdx.x, d0 = sp * seg, d1 = min(dim, d0 + seg);
    const float mu = mean[r], rs = rstd[r];
    const float* xp = x + (size_t)r * dim;
    const float* gp = dy + (size_t)r * dim;
    float s1 = 0.f, s2 = 0.f;
    for (int d = d0 + threadIdx.x; d < d1; d += 256) {
        const float xh = (xp[d] - mu) * rs, g = gp[d] * gamma[d];
        s1 += g; s2 = fmaf(g, xh, s2);
    }
    s1 = block_sum(s1, scratch);
    s2 = block_sum(s2, scratch + 16);
    if (threadIdx.x == 0) { part[((size_t)r * gridDim.x + sp) * 2] = s1; part[((size_t)r * gridDim.x + sp) * 2 + 1] = s2; }
}
// grid (splits, rows + 1): y < rows applies dx to one row segment; y == rows owns the segment's COLUMNS and sums dgamma / dbeta over all
// rows in row order (one owner per element, no atomics; the rows are few here: B * patches)
__global__ __launch_bounds__(256) void ln_long_bwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ mean,
                                                                const float* __restrict__ rstd, const float* __restrict__ dy, const float* __restrict__ part,
                                                                float* __restrict__ dx, float* __restrict__ dgamma, float* __restrict__ dbeta, int rows, int dim, int seg) {
    const int r = blockIdx.y, sp = blockIdx.x, d0 = sp * seg, d1 = min(dim, d0 + seg);
    if (r == rows) {
        for (int d = d0 + threadIdx.x; d < d1; d += 256) {
            float ag = 0.f, ab = 0.f;
            for (int q = 0; q < rows; ++q) {
                const float g = dy[(size_t)q * dim + d];
                ag = fmaf(g, (x[(size_t)q * dim + d] - mean[q]) * rstd[q], ag);
                ab += g;
            }
            dgamma[d] += ag; dbeta[d] += ab;
        }
        return;
    }
    double a1 = 0.0, a2 = 0.0;
    for (int k = 0; k < (int)gridDim.x; ++k) { a1 += part[((size_t)r * gridDim.x + k) * 2]; a2 += part[((size_t)r * gridDim.x + k) * 2 + 1]; }
    const float s1 = (float)(a1 / dim), s2 = (float)(a2 / dim);
    const float mu = mean[r], rs = rstd[r];
    const float* xp = x + (size_t)r * dim;
    const float* gp = dy + (size_t)r * dim;
    for (int d = d0 + threadIdx.x; d < d1; d += 256) {
        const float xh = (xp[d] - mu) * rs, g = gp[d] * gamma[d];
        dx[(size_t)r * dim + d] = rs * (g - s1 - xh * s2);
    }
}
constexpr int LN_LONG_DIM = 16384, LN_LONG_SEG = 4096, LN_LONG_MAX_SPLITS = 64;

// ---- GEGLU + dropout ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float gelu_erf_(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_erf_grad_(float x) {
    return 0.5f * (1.0f + erff(x * 0.70710678118654752f)) + x * 0.3989422804014327f * __expf(-0.5f * x * x);
}
// seed_step: NULL, or a device counter added (times an odd constant) to the seed -- a HIP graph bakes the by-value seed into the node,
// so a replayed step draws a fresh mask only through memory (the captured step increments the counter once, before its forward)
__global__ __launch_bounds__(256) void geglu_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t rows, int F, float p_drop, uint64_t seed,
                                                        const int64_t* __restrict__ seed_step) {
    if (seed_step) seed += (uint64_t)(*seed_step) * 0xD1342543DE82EF95ull;
    const float keep = 1.0f / (1.0f - p_drop);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < rows * F; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / F; const int f = (int)(i - r * F);
        const float a = x[r * 2 * F + f], g = x[r * 2 * F + F + f];
        float o = a * gelu_erf_(g);
        if (p_drop > 0.f) o = hash_uniform(seed, (uint32_t)i) < p_drop ? 0.f : o * keep;
        y[i] = o;
    }
}
__global__ __launch_bounds__(256) void geglu_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dx,
                                                        int64_t rows, int F, float p_drop, uint64_t seed, const int64_t* __restrict__ seed_step) {
    if (seed_step) seed += (uint64_t)(*seed_step) * 0xD1342543DE82EF95ull;
    const float keep = 1.0f / (1.0f - p_drop);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < rows * F; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / F; const int f = (int)(i - r * F);
        const float a = x[r * 2 * F + f], g = x[r * 2 * F + F + f];
        float d = dy[i];
        if (p_drop > 0.f) d = hash_uniform(seed, (uint32_t)i) < p_drop ? 0.f : d * keep;
        dx[r * 2 * F + f] = d * gelu_erf_(g);
        dx[r * 2 * F + F + f] = d * a * gelu_erf_grad_(g);
    }
}

// ---- BCELoss(sigmoid(z), y), mean over the batch ---------------------------------------------------------------------------
// torch.nn.BCELoss clamps both logs at -100 in the forward; its backward is (p - y) / max(p (1 - p), 1e-12) (ATen
// binary_cross_entropy_backward), which autograd then multiplies by the sigmoid's p (1 - p): (p - y) / n unless p (1 - p) < 1e-12,
// where the quotient shrinks the gradient (z = -30, y = 1: -0.094 / n, not -1 / n) -- reproduced as the reference computes it.
__global__ __launch_bounds__(256) void bce_sigmoid_fwd_kernel(const float* __restrict__ z, const float* __restrict__ y, float* __restrict__ loss,
                                                              float* __restrict__ dz, int n) {
    __shared__ float scratch[16];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) {
        const float p = 1.0f / (1.0f + expf(-z[i]));
        const float lp = fmaxf(logf(p), -100.0f), lq = fmaxf(logf(1.0f - p), -100.0f);
        s -= y[i] * lp + (1.0f - y[i]) * lq;
        if (dz) {
            const float pq = p * (1.0f - p);
            dz[i] = (p - y[i]) / fmaxf(pq, 1e-12f) * pq / (float)n;
        }
    }
    s = block_sum(s, scratch);
    if (threadIdx.x == 0) loss[0] = s / (float)n;
}

// exact-erf GELU as an operator of its own (vit_pytorch_diy/vit_3d.py:21 / vit.py:19 inside FeedForward when the module TRAINS: the inference
// path has it in the GEMM epilogue; under autograd the pre-activation has to survive for the backward): x, y f32 or bf16, 4 values per lane
// nn.Dropout(p) as an operator of its own (the residual-branch dropouts of vit.py:24-27, 62 / vit_3d.py:25-28 in training mode): y = x * keep / (1 - p),
// keep = a counter-based hash of (seed, element) -- the attention kernels' hash (attn_drop.h) -- so the backward applies the SAME launch to dy
// and no mask is stored; f32 or bf16, 4 / 8 values per lane
template <typename T> struct GeluVec;
template <typename T>
__global__ __launch_bounds__(256) void dropout_kernel(const T* __restrict__ x, T* __restrict__ y, int64_t n, int vec, uint32_t thr, uint32_t seed_lo, uint32_t seed_hi, float inv_keep);
// 16 bytes per lane and access (4 f32 / 8 bf16), a scalar tail; n4 = number of whole 16-byte groups
template <> struct GeluVec<float> { static constexpr int N = 4; };
template <> struct GeluVec<bf16_t> { static constexpr int N = 8; };
template <typename T>
__device__ __forceinline__ void gelu_unpack(const uint4& v, float (&f)[GeluVec<T>::N]) {
    if constexpr (GeluVec<T>::N == 4) { f[0] = __uint_as_float(v.x); f[1] = __uint_as_float(v.y); f[2] = __uint_as_float(v.z); f[3] = __uint_as_float(v.w); }
    else { f[0] = bf16lo_to_f32(v.x); f[1] = bf16hi_to_f32(v.x); f[2] = bf16lo_to_f32(v.y); f[3] = bf16hi_to_f32(v.y);
           f[4] = bf16lo_to_f32(v.z); f[5] = bf16hi_to_f32(v.z); f[6] = bf16lo_to_f32(v.w); f[7] = bf16hi_to_f32(v.w); }
}
template <typename T>
__device__ __forceinline__ uint4 gelu_pack(const float (&f)[GeluVec<T>::N]) {
    if constexpr (GeluVec<T>::N == 4) return make_uint4(__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2]), __float_as_uint(f[3]));
    else return make_uint4(pack_bf16x2(f[0], f[1]), pack_bf16x2(f[2], f[3]), pack_bf16x2(f[4], f[5]), pack_bf16x2(f[6], f[7]));
}
template <typename T>
__global__ __launch_bounds__(256) void gelu_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int64_t n, int vec) {
    constexpr int N = GeluVec<T>::N;
    const int64_t nv = vec ? n / N : 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nv; i += (int64_t)gridDim.x * 256) {
        float f[N];
        gelu_unpack<T>(reinterpret_cast<const uint4*>(x)[i], f);
#pragma unroll
        for (int k = 0; k < N; ++k) f[k] = gelu_erf_(f[k]);
        reinterpret_cast<uint4*>(y)[i] = gelu_pack<T>(f);
    }
    for (int64_t i = nv * N + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) IO<T>::st(y + i, gelu_erf_(IO<T>::ld(x + i)));
}
template <typename T>
__global__ __launch_bounds__(256) void dropout_kernel(const T* __restrict__ x, T* __restrict__ y, int64_t n, int vec, uint32_t thr, uint32_t seed_lo, uint32_t seed_hi, float inv_keep) {
    constexpr int N = GeluVec<T>::N;
    const int64_t nv = vec ? n / N : 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nv; i += (int64_t)gridDim.x * 256) {
        float f[N];
        gelu_unpack<T>(reinterpret_cast<const uint4*>(x)[i], f);
        const uint64_t e0 = (uint64_t)i * N;
        const uint32_t sd = attn_drop_seed(seed_lo, seed_hi, (uint32_t)(e0 >> 32));
#pragma unroll
        for (int k = 0; k < N; ++k) f[k] = attn_drop_hash(sd, (uint32_t)e0 + (uint32_t)k) < thr ? 0.f : f[k] * inv_keep;
        reinterpret_cast<uint4*>(y)[i] = gelu_pack<T>(f);
    }
    for (int64_t i = nv * N + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const uint32_t sd = attn_drop_seed(seed_lo, seed_hi, (uint32_t)((uint64_t)i >> 32));
        IO<T>::st(y + i, attn_drop_hash(sd, (uint32_t)i) < thr ? 0.f : IO<T>::ld(x + i) * inv_keep);
    }
}
template <typename T>
__global__ __launch_bounds__(256) void gelu_bwd_kernel(const T* __restrict__ x, const T* __restrict__ dy, T* __restrict__ dx, int64_t n, int vec) {
    constexpr int N = GeluVec<T>::N;
    const int64_t nv = vec ? n / N : 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nv; i += (int64_t)gridDim.x * 256) {
        float f[N], d[N];
        gelu_unpack<T>(reinterpret_cast<const uint4*>(x)[i], f);
        gelu_unpack<T>(reinterpret_cast<const uint4*>(dy)[i], d);
#pragma unroll
        for (int k = 0; k < N; ++k) f[k] = d[k] * gelu_erf_grad_(f[k]);
        reinterpret_cast<uint4*>(dx)[i] = gelu_pack<T>(f);
    }
    for (int64_t i = nv * N + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        IO<T>::st(dx + i, IO<T>::ld(dy + i) * gelu_erf_grad_(IO<T>::ld(x + i)));
}

}  // namespace

extern "C" {

int gfe_embed_tokens_fwd(const int64_t* x_cat, const int64_t* offsets, const float* emb, const float* x_num, const float* num_w,
                         const float* num_b, const float* cls, const float* feat, float* out,
                         int64_t B, int64_t ncat, int64_t ncont, int64_t nf, int64_t dim, int64_t ntok, void* stream) {
    GFE_REQUIRE(cls && out && (ncat == 0 || (x_cat && offsets && emb)) && (ncont == 0 || (x_num && num_w && num_b)) && (nf == 0 || feat), GFE_ERR_NULL);
    GFE_REQUIRE(B > 0 && B <= 65535 && ncat >= 0 && ncont >= 0 && nf >= 0 && dim > 0, GFE_ERR_SHAPE);
    const int64_t L = 1 + ncat + ncont + nf;
    hipLaunchKernelGGL(embed_tokens_fwd_kernel, dim3((unsigned)L, (unsigned)B), dim3(256), 0, (hipStream_t)stream, x_cat, offsets, emb, x_num, num_w, num_b,
                       cls, feat, out, (int)ncat, (int)ncont, (int)nf, (int)dim, (int)ntok);
    return gfe_launch_status();
}

int gfe_embed_tokens_bwd(const float* dout, const int64_t* x_cat, const int64_t* offsets, const float* x_num,
                         float* d_emb, float* d_num_w, float* d_num_b, float* d_cls, float* d_feat,
                         int64_t B, int64_t ncat, int64_t ncont, int64_t nf, int64_t dim, int64_t ntok, void* stream) {
    GFE_REQUIRE(dout && d_cls && (ncat == 0 || (x_cat && offsets && d_emb)) && (ncont == 0 || (x_num && d_num_w && d_num_b)), GFE_ERR_NULL);
    GFE_REQUIRE(B > 0 && dim > 0, GFE_ERR_SHAPE);
    const int64_t L = 1 + ncat + ncont + nf;
    hipLaunchKernelGGL(embed_tokens_bwd_kernel, dim3((unsigned)L), dim3(256), 0, (hipStream_t)stream, dout, x_cat, offsets, x_num, d_emb, d_num_w, d_num_b,
                       d_cls, d_feat, (int)B, (int)ncat, (int)ncont, (int)nf, (int)dim, (int)ntok);
    return gfe_launch_status();
}

int gfe_mean_tokens_fwd(const float* x, float* y, int64_t B, int64_t L, int64_t dim, void* stream) {
    GFE_REQUIRE(x && y, GFE_ERR_NULL);
    GFE_REQUIRE(B > 0 && L > 0 && dim > 0, GFE_ERR_SHAPE);
    GFE_REQUIRE(B <= 65535, GFE_ERR_SHAPE);
    hipLaunchKernelGGL(mean_tokens_fwd_kernel, dim3((unsigned)ceil_div(dim, 64), (unsigned)B), dim3(256), 0, (hipStream_t)stream, x, y, (int)L, (int)dim);
    return gfe_launch_status();
}
int gfe_mean_tokens_bwd(const float* dy, float* dx, int64_t B, int64_t L, int64_t dim, void* stream) {
    GFE_REQUIRE(dy && dx, GFE_ERR_NULL);
    GFE_REQUIRE(B > 0 && B <= 65535 && L > 0 && dim > 0, GFE_ERR_SHAPE);
    hipLaunchKernelGGL(mean_tokens_bwd_kernel, dim3((unsigned)L, (unsigned)B), dim3(256), 0, (hipStream_t)stream, dy, dx, (int)L, (int)dim);
    return gfe_launch_status();
}

int gfe_cross_attn_q1_fwd(const float* q, const float* k, const float* v, float* out, float* probs,
                          int64_t B, int64_t H, int64_t nk, int64_t dh, float scale, void* stream) {
    GFE_REQUIRE(q && k && v && out && probs, GFE_ERR_NULL);
    GFE_REQUIRE(B > 0 && B <= 65535 && H > 0 && nk > 0 && nk <= 8192 && dh > 0, GFE_ERR_SHAPE);
    hipLaunchKernelGGL(cross_attn_q1_fwd_kernel, dim3((unsigned)H, (unsigned)B), dim3(256), (size_t)(nk + 64 + 4 * dh) * sizeof(float), (hipStream_t)stream,
                       q, k, v, out, probs, (int)H, (int)nk, (int)dh, scale);
    return gfe_launch_status();
}
int gfe_cross_attn_q1_bwd(const float* q, const float* k, const float* v, const float* probs, const float* dout,
                          float* dq, float* dk, float* dv, int64_t B, int64_t H, int64_t nk, int64_t dh, float scale, void* stream) {
    GFE_REQUIRE(q && k && v && probs && dout && dq && dk && dv, GFE_ERR_NULL);
    GFE_REQUIRE(B > 0 && B <= 65535 && H > 0 && nk > 0 && nk <= 8192 && dh > 0, GFE_ERR_SHAPE);
    hipLaunchKernelGGL(cross_attn_q1_bwd_kernel, dim3((unsigned)H, (unsigned)B), dim3(256), (size_t)(nk + 64 + 4 * dh) * sizeof(float), (hipStream_t)stream,
                       q, k, v, probs, dout, dq, dk, dv, (int)H, (int)nk, (int)dh, scale);
    return gfe_launch_status();
}

int gfe_sdpa_small_fwd(const float* q, const float* k, const float* v, float* out, float* probs,
                       int64_t B, int64_t H, int64_t L, int64_t dh, float scale, int causal, float p_drop, int64_t seed, void* stream) {
    GFE_REQUIRE(q && k && v && out && probs, GFE_ERR_NULL);
    GFE_REQUIRE(B > 0 && B <= 65535 && H > 0 && L > 0 && L <= 64 && dh > 0 && dh <= 64 && p_drop >= 0.f && p_drop < 1.f && B * H * L * L <= 0x7fffffff, GFE_ERR_SHAPE);
    hipLaunchKernelGGL(sdpa_small_fwd_kernel, dim3((unsigned)H, (unsigned)B), dim3(256), (size_t)(3 * L * dh + L * L) * sizeof(float), (hipStream_t)stream,
                       q, k, v, out, probs, (int)H, (int)L, (int)dh, scale, causal, p_drop, (uint64_t)seed);
    return gfe_launch_status();
}
int gfe_sdpa_small_bwd(const float* q, const float* k, const float* v, const float* probs, const float* dout,
                       float* dq, float* dk, float* dv, int64_t B, int64_t H, int64_t L, int64_t dh, float scale, float p_drop, int64_t seed, void* stream) {
    GFE_REQUIRE(q && k && v && probs && dout && dq && dk && dv, GFE_ERR_NULL);
    GFE_REQUIRE(B > 0 && B <= 65535 && H > 0 && L > 0 && L <= 64 && dh > 0 && dh <= 64 && p_drop >= 0.f && p_drop < 1.f && B * H * L * L <= 0x7fffffff, GFE_ERR_SHAPE);
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute((const void*)sdpa_small_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 131072); attr = true; }
    hipLaunchKernelGGL(sdpa_small_bwd_kernel, dim3((unsigned)H, (unsigned)B), dim3(256), (size_t)(4 * L * dh + 3 * L * L) * sizeof(float), (hipStream_t)stream,
                       q, k, v, probs, dout, dq, dk, dv, (int)H, (int)L, (int)dh, scale, p_drop, (uint64_t)seed);
    return gfe_launch_status();
}

int gfe_layernorm_rows_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd, float* ws,
                           int64_t rows, int64_t dim, float eps, void* stream) {
    GFE_REQUIRE(x && gamma && beta && y && mean && rstd, GFE_ERR_NULL);
    GFE_REQUIRE(rows > 0 && dim > 0, GFE_ERR_SHAPE);
    if (dim >= LN_LONG_DIM && ws && rows <= 65535) {
        int splits = (int)ceil_div(dim, (int64_t)LN_LONG_SEG); if (splits > LN_LONG_MAX_SPLITS) splits = LN_LONG_MAX_SPLITS;
        const int seg = (int)ceil_div(dim, (int64_t)splits);
        hipLaunchKernelGGL(ln_long_moments_kernel, dim3((unsigned)splits, (unsigned)rows), dim3(256), 0, (hipStream_t)stream, x, ws, (int)dim, seg);
        hipLaunchKernelGGL(ln_long_apply_kernel, dim3((unsigned)splits, (unsigned)rows), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, ws, y, mean, rstd, (int)dim, seg, eps);
        return gfe_launch_status();
    }
    hipLaunchKernelGGL(ln_rows_fwd_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, y, mean, rstd, (int)dim, eps);
    return gfe_launch_status();
}
int gfe_layernorm_rows_bwd(const float* x, const float* gamma, const float* mean, const float* rstd, const float* dy, float* dx,
                           float* dgamma, float* dbeta, float* ws, int64_t rows, int64_t dim, void* stream) {
    GFE_REQUIRE(x && gamma && mean && rstd && dy && dx && dgamma && dbeta, GFE_ERR_NULL);
    GFE_REQUIRE(rows > 0 && dim > 0, GFE_ERR_SHAPE);
    if (dim >= LN_LONG_DIM && ws && rows <= 65535) {
        int splits = (int)ceil_div(dim, (int64_t)LN_LONG_SEG); if (splits > LN_LONG_MAX_SPLITS) splits = LN_LONG_MAX_SPLITS;
        const int seg = (int)ceil_div(dim, (int64_t)splits);
        hipLaunchKernelGGL(ln_long_bwd_sums_kernel, dim3((unsigned)splits, (unsigned)rows), dim3(256), 0, (hipStream_t)stream, x, gamma, mean, rstd, dy, ws, (int)dim, seg);
        hipLaunchKernelGGL(ln_long_bwd_apply_kernel, dim3((unsigned)splits, (unsigned)rows + 1), dim3(256), 0, (hipStream_t)stream, x, gamma, mean, rstd, dy, ws, dx, dgamma, dbeta, (int)rows, (int)dim, seg);
        return gfe_launch_status();
    }
    if (ws && rows >= LN_TALL_MIN_ROWS && dim <= LN_TALL_MAX_DIM && dim % 4 == 0 && rows <= 0x7fffffff) {
        const int rpb = (int)ceil_div(rows, (int64_t)LN_TALL_MAX_BLOCKS) < 8 ? 8 : (int)ceil_div(rows, (int64_t)LN_TALL_MAX_BLOCKS);
        const int nblk = (int)ceil_div(rows, (int64_t)rpb);
        const int ni = (int)ceil_div(dim, 256);
        hipStream_t st = (hipStream_t)stream;
#define GFE_LN_TALL(NI) hipLaunchKernelGGL(ln_tall_bwd_kernel<NI>, dim3((unsigned)nblk), dim3(256), 0, st, x, gamma, mean, rstd, dy, dx, ws, (int)rows, (int)dim, rpb)
        if (ni <= 1) GFE_LN_TALL(1); else if (ni <= 2) GFE_LN_TALL(2); else if (ni <= 4) GFE_LN_TALL(4); else GFE_LN_TALL(8);
#undef GFE_LN_TALL
        hipLaunchKernelGGL(ln_tall_bwd_reduce_kernel, dim3((unsigned)ceil_div(2 * dim, 64)), dim3(256), 0, st, ws, dgamma, dbeta, (int)dim, nblk);
        return gfe_launch_status();
    }
    GFE_REQUIRE(rows + ceil_div(dim, 64) <= 0x7fffffff, GFE_ERR_SHAPE);
    hipLaunchKernelGGL(ln_rows_bwd_kernel, dim3((unsigned)(rows + ceil_div(dim, 64))), dim3(256), 0, (hipStream_t)stream, x, gamma, mean, rstd, dy, dx, dgamma, dbeta, (int)rows, (int)dim);
    return gfe_launch_status();
}

int gfe_gelu_fwd(const void* x, void* y, int64_t n, int dtype, void* stream) {
    GFE_REQUIRE(x && y, GFE_ERR_NULL);
    GFE_REQUIRE(n > 0, GFE_ERR_SHAPE);
    const int vec = (((uintptr_t)x | (uintptr_t)y) & 15) == 0;
    int64_t g = ceil_div(n, 256 * 8); if (g > 8192) g = 8192;
    if (dtype == GFE_F32) hipLaunchKernelGGL((gelu_fwd_kernel<float>), dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, (const float*)x, (float*)y, n, vec);
    else if (dtype == GFE_BF16) hipLaunchKernelGGL((gelu_fwd_kernel<bf16_t>), dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (bf16_t*)y, n, vec);
    else return GFE_ERR_DTYPE;
    return gfe_launch_status();
}
int gfe_gelu_bwd(const void* x, const void* dy, void* dx, int64_t n, int dtype, void* stream) {
    GFE_REQUIRE(x && dy && dx, GFE_ERR_NULL);
    GFE_REQUIRE(n > 0, GFE_ERR_SHAPE);
    const int vec = (((uintptr_t)x | (uintptr_t)dy | (uintptr_t)dx) & 15) == 0;
    int64_t g = ceil_div(n, 256 * 8); if (g > 8192) g = 8192;
    if (dtype == GFE_F32) hipLaunchKernelGGL((gelu_bwd_kernel<float>), dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, (const float*)x, (const float*)dy, (float*)dx, n, vec);
    else if (dtype == GFE_BF16) hipLaunchKernelGGL((gelu_bwd_kernel<bf16_t>), dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (const bf16_t*)dy, (bf16_t*)dx, n, vec);
    else return GFE_ERR_DTYPE;
    return gfe_launch_status();
}

int gfe_dropout(const void* x, void* y, int64_t n, float p_drop, int64_t seed, int dtype, void* stream) {
    GFE_REQUIRE(x && y, GFE_ERR_NULL);
    GFE_REQUIRE(n > 0 && p_drop >= 0.f && p_drop < 1.f, GFE_ERR_SHAPE);
    const int vec = (((uintptr_t)x | (uintptr_t)y) & 15) == 0;
    const uint32_t thr = attn_drop_threshold(p_drop), lo = (uint32_t)(uint64_t)seed, hi = (uint32_t)((uint64_t)seed >> 32);
    const float inv_keep = 1.0f / (1.0f - p_drop);
    int64_t g = ceil_div(n, 256 * 8); if (g > 8192) g = 8192;
    if (dtype == GFE_F32) hipLaunchKernelGGL((dropout_kernel<float>), dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, (const float*)x, (float*)y, n, vec, thr, lo, hi, inv_keep);
    else if (dtype == GFE_BF16) hipLaunchKernelGGL((dropout_kernel<bf16_t>), dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (bf16_t*)y, n, vec, thr, lo, hi, inv_keep);
    else return GFE_ERR_DTYPE;
    return gfe_launch_status();
}

int gfe_geglu_fwd(const float* x, float* y, int64_t rows, int64_t F, float p_drop, int64_t seed, const int64_t* seed_step, void* stream) {
    GFE_REQUIRE(x && y, GFE_ERR_NULL);
    GFE_REQUIRE(rows > 0 && F > 0 && rows * F <= 0x7fffffff && p_drop >= 0.f && p_drop < 1.f, GFE_ERR_SHAPE);
    int64_t g = ceil_div(rows * F, 256); if (g > 1024) g = 1024;
    hipLaunchKernelGGL(geglu_fwd_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, x, y, rows, (int)F, p_drop, (uint64_t)seed, seed_step);
    return gfe_launch_status();
}
int gfe_geglu_bwd(const float* x, const float* dy, float* dx, int64_t rows, int64_t F, float p_drop, int64_t seed, const int64_t* seed_step, void* stream) {
    GFE_REQUIRE(x && dy && dx, GFE_ERR_NULL);
    GFE_REQUIRE(rows > 0 && F > 0 && rows * F <= 0x7fffffff && p_drop >= 0.f && p_drop < 1.f, GFE_ERR_SHAPE);
    int64_t g = ceil_div(rows * F, 256); if (g > 1024) g = 1024;
    hipLaunchKernelGGL(geglu_bwd_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, x, dy, dx, rows, (int)F, p_drop, (uint64_t)seed, seed_step);
    return gfe_launch_status();
}

int gfe_bce_sigmoid(const float* logits, const float* y, float* loss, float* dlogits, int64_t n, void* stream) {
    GFE_REQUIRE(logits && y && loss, GFE_ERR_NULL);
    GFE_REQUIRE(n > 0 && n <= 0x7fffffff, GFE_ERR_SHAPE);
    hipLaunchKernelGGL(bce_sigmoid_fwd_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, logits, y, loss, dlogits, (int)n);
    return gfe_launch_status();
}

}  // extern "C"
