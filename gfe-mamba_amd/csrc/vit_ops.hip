// Bottleneck-ViT helpers (vit_pytorch_diy/vit.py): LayerNorm over rows that may be gathered/scattered in patch order,
// the fused small attention (n = 25 tokens in the reference geometry), the token-axis Linear of from_patch_embedding,
// and the cls/pos-embedding add.  All HBM/latency bound; the GEMMs live in gemm.hip.
#include "common.h"

namespace {

// A logical row r of `len = nseg*seglen` elements stored as nseg contiguous segments:
//   base(r, s) = (r / rpb)*batch_stride + ((r % rpb) / n_inner)*outer_stride + ((r % rpb) % n_inner)*inner_stride + s*seg_stride
// Plain contiguous rows: nseg = 1, rpb = R, n_inner = 1, outer_stride = len.
// Patch rows of a channels-last image (B, Himg, Wimg, C) with p x p patches ('b c (h p1) (w p2) -> b (h w) (p1 p2 c)',
// vit.py:96): nseg = p (p1), seglen = p*C (p2, c), rpb = (Himg/p)*(Wimg/p), n_inner = Wimg/p,
//   batch_stride = Himg*Wimg*C, outer_stride = p*Wimg*C, inner_stride = p*C, seg_stride = Wimg*C.
struct RowMap { int64_t batch_stride, outer_stride, inner_stride, seg_stride; int rpb, n_inner, nseg, seglen; };

__device__ __forceinline__ int64_t row_base(const RowMap& m, int r) {
    const int rb = r / m.rpb, rr = r - rb * m.rpb;
    const int ro = rr / m.n_inner, ri = rr - ro * m.n_inner;
    return (int64_t)rb * m.batch_stride + (int64_t)ro * m.outer_stride + (int64_t)ri * m.inner_stride;
}

template <typename T> struct V8;
template <> struct V8<bf16_t> {
    static __device__ __forceinline__ void ld(const bf16_t* p, float (&o)[8]) {
        const uint4 v = *reinterpret_cast<const uint4*>(p);
        o[0] = bf16lo_to_f32(v.x); o[1] = bf16hi_to_f32(v.x); o[2] = bf16lo_to_f32(v.y); o[3] = bf16hi_to_f32(v.y);
        o[4] = bf16lo_to_f32(v.z); o[5] = bf16hi_to_f32(v.z); o[6] = bf16lo_to_f32(v.w); o[7] = bf16hi_to_f32(v.w);
    }
    static __device__ __forceinline__ void st(bf16_t* p, const float (&o)[8]) {
        *reinterpret_cast<uint4*>(p) = make_uint4(pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]), pack_bf16x2(o[4], o[5]), pack_bf16x2(o[6], o[7]));
    }
};
template <> struct V8<float> {
    static __device__ __forceinline__ void ld(const float* p, float (&o)[8]) {
        const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
        o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
    }
    static __device__ __forceinline__ void st(float* p, const float (&o)[8]) {
        *reinterpret_cast<float4*>(p) = make_float4(o[0], o[1], o[2], o[3]);
        *reinterpret_cast<float4*>(p + 4) = make_float4(o[4], o[5], o[6], o[7]);
    }
};

// LayerNorm(len), eps, affine (gamma/beta f32, indexed by logical position).  One block per row; the row is read twice
// (statistics, then normalise) -- the second read of a <= 300 KB row comes from L2.  f32 statistics.
template <typename TI, typename TO>
__global__ __launch_bounds__(1024) void layernorm_kernel(const TI* __restrict__ x, TO* __restrict__ y, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, RowMap mi, RowMap mo, float eps) {
    __shared__ float red[32];
    const int r = blockIdx.x;
    const int64_t ib = row_base(mi, r), ob = row_base(mo, r);
    const int len = mi.nseg * mi.seglen, nvec = len >> 3;
    float s = 0.f, q = 0.f;
    for (int v = threadIdx.x; v < nvec; v += blockDim.x) {
        const int pos = v << 3, sg = pos / mi.seglen, off = pos - sg * mi.seglen;
        float f[8];
        V8<TI>::ld(x + ib + (int64_t)sg * mi.seg_stride + off, f);
#pragma unroll
        for (int j = 0; j < 8; ++j) { s += f[j]; q = fmaf(f[j], f[j], q); }
    }
    s = block_sum(s, red);
    q = block_sum(q, red + 16);
    const float mean = s / (float)len;
    const float var = fmaxf(q / (float)len - mean * mean, 0.f);
    const float rstd = rsqrtf(var + eps);
    for (int v = threadIdx.x; v < nvec; v += blockDim.x) {
        const int pos = v << 3, sg = pos / mi.seglen, off = pos - sg * mi.seglen;
        const int sgo = pos / mo.seglen, offo = pos - sgo * mo.seglen;
        float f[8], g[8], bt[8];
        V8<TI>::ld(x + ib + (int64_t)sg * mi.seg_stride + off, f);
        V8<float>::ld(gamma + pos, g);
        V8<float>::ld(beta + pos, bt);
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = fmaf((f[j] - mean) * rstd, g[j], bt[j]);
        V8<TO>::st(y + ob + (int64_t)sgo * mo.seg_stride + offo, f);
    }
}

// softmax(q k^T * scale) v for short sequences: block = (batch, head) x 4 query rows (one per wave); q, k, v rows are read with
// arbitrary row strides so packed qkv buffers work.  nq, nk <= 256, dh <= 64.  f32 math, bf16 I/O.  K and V (a few KB) are staged
// per block: re-staging them for every 4 queries is cheaper than walking all queries of a head sequentially in one block.
__global__ __launch_bounds__(256) void attn_small_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v,
                                                         bf16_t* __restrict__ o, int H, int nq, int nk, int dh,
                                                         int64_t qb, int64_t qr, int64_t kb, int64_t kr, int64_t vb, int64_t vr,
                                                         int64_t ob, int64_t orow, float scale) {
    extern __shared__ float sm[];                 // K [nk][dh+1], V [nk][dh+1], P [4][nk], Q [4][dh]
    const int b = blockIdx.x / H, h = blockIdx.x - b * H;
    const int ld = dh + 1;
    float* sK = sm; float* sV = sm + nk * ld; float* sP = sV + nk * ld; float* sQ = sP + 4 * nk;
    for (int i = threadIdx.x; i < nk * dh; i += 256) {
        const int r = i / dh, c = i - r * dh;
        sK[r * ld + c] = bf16_to_f32(k[b * kb + (int64_t)r * kr + h * dh + c]);
        sV[r * ld + c] = bf16_to_f32(v[b * vb + (int64_t)r * vr + h * dh + c]);
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int iq = blockIdx.y * 4 + wave;
    if (iq < nq && lane < dh) sQ[wave * dh + lane] = bf16_to_f32(q[b * qb + (int64_t)iq * qr + h * dh + lane]) * scale;
    __syncthreads();
    if (iq >= nq) return;
    float* P = sP + wave * nk;
    const float* Q = sQ + wave * dh;
    float mx = -3.0e38f;
    for (int j = lane; j < nk; j += 64) {
        float sc = 0.f;
        for (int c = 0; c < dh; ++c) sc = fmaf(Q[c], sK[j * ld + c], sc);          // Q[c]: LDS broadcast
        P[j] = sc; mx = fmaxf(mx, sc);
    }
    mx = wave_max(mx);
    float sum = 0.f;
    for (int j = lane; j < nk; j += 64) { const float e = __expf(P[j] - mx); P[j] = e; sum += e; }
    sum = wave_sum(sum);
    const float inv = 1.0f / sum;
    __builtin_amdgcn_wave_barrier();
    if (lane < dh) {
        float acc = 0.f;
        for (int j = 0; j < nk; ++j) acc = fmaf(P[j], sV[j * ld + lane], acc);
        o[b * ob + (int64_t)iq * orow + h * dh + lane] = f32_to_bf16(acc * inv);
    }
}

// from_patch_embedding's Linear over the TOKEN axis (vit.py:104-106): y[b][j][d] = sum_i W[j][i] x[b][i][d] + bias[j]
template <typename TI>
__global__ __launch_bounds__(256) void token_mix_kernel(const TI* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                        bf16_t* __restrict__ y, int B, int nin, int nout, int dim) {
    const int64_t total = (int64_t)B * nout * dim;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int d = (int)(i % dim); const int64_t t = i / dim;
        const int j = (int)(t % nout), b = (int)(t / nout);
        float acc = bias[j];
        for (int k = 0; k < nin; ++k) acc = fmaf(w[j * nin + k], IO<TI>::ld(x + ((size_t)b * nin + k) * dim + d), acc);
        y[i] = f32_to_bf16(acc);
    }
}

// x[b][0] = cls + pos[0];  x[b][1+i] = tok[b][i] + pos[1+i]   (vit.py:127-130), f32
__global__ __launch_bounds__(256) void vit_embed_kernel(const float* __restrict__ tok, const float* __restrict__ cls, const float* __restrict__ pos,
                                                        float* __restrict__ x, int B, int n, int dim) {
    const int64_t total = (int64_t)B * (n + 1) * dim;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int d = (int)(i % dim); const int64_t t = i / dim;
        const int j = (int)(t % (n + 1)), b = (int)(t / (n + 1));
        const float base = j == 0 ? cls[d] : tok[((size_t)b * n + j - 1) * dim + d];
        x[i] = base + pos[j * dim + d];
    }
}

template <typename TI, typename TO>
int ln_launch(const void* x, void* y, const float* g, const float* b, const RowMap& mi, const RowMap& mo, int64_t rows, float eps, hipStream_t st) {
    const int len = mi.nseg * mi.seglen;
    int threads = len / 8;
    threads = threads >= 1024 ? 1024 : (threads <= 64 ? 64 : ((threads + 63) / 64) * 64);
    hipLaunchKernelGGL((layernorm_kernel<TI, TO>), dim3((unsigned)rows), dim3(threads), 0, st, (const TI*)x, (TO*)y, g, b, mi, mo, eps);
    return gfe_launch_status();
}

}  // namespace

extern "C" {

/* map arrays: {batch_stride, outer_stride, inner_stride, seg_stride, rpb, n_inner, nseg, seglen} (int64 each, HOST pointers) */
int gfe_layernorm(const void* x, void* y, const float* gamma, const float* beta, const int64_t* in_map, const int64_t* out_map,
                  int64_t rows, float eps, int in_dtype, int out_dtype, void* stream) {
    GFE_REQUIRE(x && y && gamma && beta && in_map && out_map, GFE_ERR_NULL);
    RowMap mi{in_map[0], in_map[1], in_map[2], in_map[3], (int)in_map[4], (int)in_map[5], (int)in_map[6], (int)in_map[7]};
    RowMap mo{out_map[0], out_map[1], out_map[2], out_map[3], (int)out_map[4], (int)out_map[5], (int)out_map[6], (int)out_map[7]};
    GFE_REQUIRE(rows > 0 && mi.nseg > 0 && mi.seglen % 8 == 0 && mo.seglen % 8 == 0 && mi.nseg * mi.seglen == mo.nseg * mo.seglen, GFE_ERR_SHAPE);
    GFE_REQUIRE(mi.rpb > 0 && mi.n_inner > 0 && mo.rpb > 0 && mo.n_inner > 0, GFE_ERR_SHAPE);
    hipStream_t st = (hipStream_t)stream;
    if (in_dtype == GFE_BF16 && out_dtype == GFE_BF16) return ln_launch<bf16_t, bf16_t>(x, y, gamma, beta, mi, mo, rows, eps, st);
    if (in_dtype == GFE_F32 && out_dtype == GFE_BF16) return ln_launch<float, bf16_t>(x, y, gamma, beta, mi, mo, rows, eps, st);
    if (in_dtype == GFE_F32 && out_dtype == GFE_F32) return ln_launch<float, float>(x, y, gamma, beta, mi, mo, rows, eps, st);
    if (in_dtype == GFE_BF16 && out_dtype == GFE_F32) return ln_launch<bf16_t, float>(x, y, gamma, beta, mi, mo, rows, eps, st);
    return GFE_ERR_DTYPE;
}

int gfe_attention_small(const void* q, const void* k, const void* v, void* o, int64_t B, int64_t H, int64_t nq, int64_t nk, int64_t dh,
                        int64_t q_batch, int64_t q_row, int64_t k_batch, int64_t k_row, int64_t v_batch, int64_t v_row,
                        int64_t o_batch, int64_t o_row, float scale, void* stream) {
    GFE_REQUIRE(q && k && v && o, GFE_ERR_NULL);
    GFE_REQUIRE(B > 0 && H > 0 && nq > 0 && nk > 0 && nk <= 256 && dh > 0 && dh <= 64, GFE_ERR_SHAPE);
    const size_t lds = (2 * (size_t)nk * (dh + 1) + 4 * (size_t)nk + 4 * (size_t)dh) * sizeof(float);
    GFE_REQUIRE(lds <= 64 * 1024 && B * H <= 0x7fffffff && nq <= 65535 * 4, GFE_ERR_SHAPE);
    hipLaunchKernelGGL(attn_small_kernel, dim3((unsigned)(B * H), (unsigned)ceil_div(nq, 4)), dim3(256), lds, (hipStream_t)stream,
                       (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (bf16_t*)o, (int)H, (int)nq, (int)nk, (int)dh,
                       q_batch, q_row, k_batch, k_row, v_batch, v_row, o_batch, o_row, scale);
    return gfe_launch_status();
}

int gfe_token_mix(const void* x, const float* w, const float* bias, void* y, int64_t B, int64_t nin, int64_t nout, int64_t dim,
                  int in_dtype, void* stream) {
    GFE_REQUIRE(x && w && bias && y, GFE_ERR_NULL);
    GFE_REQUIRE(B > 0 && nin > 0 && nout > 0 && dim > 0, GFE_ERR_SHAPE);
    int64_t g = ceil_div(B * nout * dim, 256);
    if (g > 4096) g = 4096;
    if (in_dtype == GFE_F32)
        hipLaunchKernelGGL((token_mix_kernel<float>), dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, (const float*)x, w, bias, (bf16_t*)y, (int)B, (int)nin, (int)nout, (int)dim);
    else if (in_dtype == GFE_BF16)
        hipLaunchKernelGGL((token_mix_kernel<bf16_t>), dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, w, bias, (bf16_t*)y, (int)B, (int)nin, (int)nout, (int)dim);
    else return GFE_ERR_DTYPE;
    return gfe_launch_status();
}

int gfe_vit_embed(const float* tok, const float* cls, const float* pos, float* x, int64_t B, int64_t n, int64_t dim, void* stream) {
    GFE_REQUIRE(tok && cls && pos && x, GFE_ERR_NULL);
    GFE_REQUIRE(B > 0 && n > 0 && dim > 0, GFE_ERR_SHAPE);
    int64_t g = ceil_div(B * (n + 1) * dim, 256);
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(vit_embed_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, tok, cls, pos, x, (int)B, (int)n, (int)dim);
    return gfe_launch_status();
}

}  // extern "C"
