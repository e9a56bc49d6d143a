// Sparse mixture-of-experts MLP for Jamba's SparseMoEBlock (cross_atten/jamba.py:441-535: router Linear -> softmax -> top-k ->
// per-expert MLP down(silu(gate(x)) * up(x)) weighted by the routing probability and summed), row f-2 of SURVEY 8.
//
// The reference (and round 2's transcription of it) walks the experts in a Python loop: one_hot + where + index_add per expert, a host
// read of the hit counts, 16 x 3 GEMM launches per layer.  Here the (token, expert) pairs are sorted by expert ON THE DEVICE (a stable
// counting sort: deterministic order, no host sync) and every projection of ALL experts is ONE grouped GEMM launch whose blocks read their
// expert's row range from the segment table; the combine step is a gather (each token reads its k rows), so nothing is accumulated with
// atomics and results are bit-reproducible.  Exact-f32 operands on the f32 matrix cores (v_mfma_f32_16x16x4_f32), like the rest of the
// trainable head (the reference trains it in fp32, classify_mamba.py:69-74).
#include "common.h"

typedef __attribute__((ext_vector_type(4))) float f32x4;

namespace {

// ---- routing ------------------------------------------------------------------------------------------------------------------------
// One block.  Phase 1: softmax + top-k per token (first maximum wins ties, as torch.topk does on equal values in practice).  Phase 2:
// stable counting sort of the pairs p = t * K + k by expert: thread i owns a contiguous chunk of pairs, counts them per expert into LDS,
// the counts are scanned over the threads per expert, and each thread hands out positions to its chunk in order.
constexpr int RT = 256;          // routing threads
constexpr int MAXE = 64;         // experts

__global__ __launch_bounds__(RT) void moe_route_kernel(const float* __restrict__ logits, int T, int E, int K, float* __restrict__ rw,
                                                       int* __restrict__ sel, int* __restrict__ tok_sorted, int* __restrict__ pos,
                                                       int* __restrict__ seg) {
    extern __shared__ int cnt[];                 // [RT][E] chunk counts, then bases; + seg[E + 1]
    int* sseg = cnt + RT * E;
    const int tid = threadIdx.x, P = T * K;
    for (int t = tid; t < T; t += RT) {
        float l[MAXE];
        float mx = -INFINITY;
        for (int e = 0; e < E; ++e) { l[e] = logits[(size_t)t * E + e]; mx = fmaxf(mx, l[e]); }
        float sum = 0.f;
        for (int e = 0; e < E; ++e) { l[e] = __expf(l[e] - mx); sum += l[e]; }
        const float inv = 1.0f / sum;
        for (int k = 0; k < K; ++k) {
            int best = 0; float bv = -1.f;
            for (int e = 0; e < E; ++e) if (l[e] > bv) { bv = l[e]; best = e; }
            rw[(size_t)t * K + k] = bv * inv;
            sel[(size_t)t * K + k] = best;
            l[best] = -2.f;                      // taken
        }
    }
    __syncthreads();
    const int chunk = (P + RT - 1) / RT, p0 = tid * chunk, p1 = min(P, p0 + chunk);
    for (int e = 0; e < E; ++e) cnt[tid * E + e] = 0;
    for (int p = p0; p < p1; ++p) cnt[tid * E + sel[p]]++;
    __syncthreads();
    if (tid < E) {                                // exclusive scan over the threads, per expert
        int run = 0;
        for (int i = 0; i < RT; ++i) { const int c = cnt[i * E + tid]; cnt[i * E + tid] = run; run += c; }
        sseg[tid + 1] = run;                      // total of expert tid (turned into offsets below)
    }
    __syncthreads();
    if (tid == 0) {
        int run = 0;
        sseg[0] = 0;
        for (int e = 0; e < E; ++e) { const int c = sseg[e + 1]; sseg[e] = run; run += c; }
        sseg[E] = run;
    }
    __syncthreads();
    if (tid <= E) seg[tid] = sseg[tid];
    for (int p = p0; p < p1; ++p) {
        const int e = sel[p];
        const int q = sseg[e] + cnt[tid * E + e]++;
        tok_sorted[q] = p / K;
        pos[p] = q;
    }
}

// d logits from d routing weights: rw_k = softmax(logits)[sel_k] (no renormalisation over the selected ones, jamba.py:487-489):
//   d logits[j] = p_j * (sum_k drw_k [j == sel_k] - sum_k drw_k p_sel_k)
__global__ __launch_bounds__(256) void moe_route_bwd_kernel(const float* __restrict__ logits, const int* __restrict__ sel, const float* __restrict__ drw,
                                                            float* __restrict__ dlogits, int T, int E, int K) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= T) return;
    float l[MAXE];
    float mx = -INFINITY;
    for (int e = 0; e < E; ++e) { l[e] = logits[(size_t)t * E + e]; mx = fmaxf(mx, l[e]); }
    float sum = 0.f;
    for (int e = 0; e < E; ++e) { l[e] = __expf(l[e] - mx); sum += l[e]; }
    const float inv = 1.0f / sum;
    float dot = 0.f;
    for (int k = 0; k < K; ++k) dot = fmaf(drw[(size_t)t * K + k], l[sel[(size_t)t * K + k]] * inv, dot);
    for (int e = 0; e < E; ++e) {
        float g = -dot;
        for (int k = 0; k < K; ++k) if (sel[(size_t)t * K + k] == e) g += drw[(size_t)t * K + k];
        dlogits[(size_t)t * E + e] = l[e] * inv * g;
    }
}

// ---- grouped GEMM -------------------------------------------------------------------------------------------------------------------
// C_e[M][N] (+)= op(A_e)[M][K] op(B_e)[N][K]^T for every expert e = blockIdx.z, f32, 32 x 32 output tiles, BK-deep k steps.
// An operand is K-major (element (row, k) at base[row * ld + k]) or reduction-major (tr: at base[k * ld + row]); its OUTER memory index
// (row when K-major, k when reduction-major) is either plain (a weight / gradient matrix of the expert, reached through a pointer table)
// or a TOKEN index inside the expert's segment of the sorted pair list: seg[e] + i, optionally sent through a gather table (the rows of
// x that belong to the pairs).  The extent of a token dimension is the expert's pair count.
struct Operand {
    const float* base;             // token operands
    const float* const* table;     // per-expert pointers (weights)
    int64_t ld;
    const int* gather;             // token operands: row = gather[seg[e] + i] instead of seg[e] + i
    int tr, token;
};
struct MoeGemmParams {
    Operand A, B;
    float* C; float* const* Ctable; int64_t ldc; int c_token;
    const int* seg;
    int M, N, K;                   // fixed extents (ignored for whichever dimension is the token dimension)
    int accumulate;
};

template <int BK>
__device__ __forceinline__ void moe_tile_load(const Operand& o, const float* __restrict__ base, int seg0, int row0, int nrows, int k0, int kend, int t,
                                              float (&v)[32 * BK / 256]) {
    constexpr int E = 32 * BK / 256;
    int outer, inner, outer_n, inner_n;
    if (!o.tr) { outer = row0 + t / (BK / E); inner = k0 + E * (t % (BK / E)); outer_n = nrows; inner_n = kend; }
    else { outer = k0 + t / (32 / E); inner = row0 + E * (t % (32 / E)); outer_n = kend; inner_n = nrows; }
    if (outer >= outer_n) {
#pragma unroll
        for (int i = 0; i < E; ++i) v[i] = 0.f;
        return;
    }
    const int64_t mrow = !o.token ? outer : (o.gather ? o.gather[seg0 + outer] : seg0 + outer);
    const float* q = base + mrow * o.ld + inner;
#pragma unroll
    for (int i = 0; i < E; ++i) v[i] = (inner + i < inner_n) ? q[i] : 0.f;
}
template <int BK>
__device__ __forceinline__ void moe_tile_store(float* __restrict__ s, int tr, int t, const float (&v)[32 * BK / 256]) {
    constexpr int E = 32 * BK / 256, LD = 32 + 16;
    if (!tr) {
        const int r = t / (BK / E), k = E * (t % (BK / E));
#pragma unroll
        for (int i = 0; i < E; ++i) s[(k + i) * LD + r] = v[i];
    } else {
        const int k = t / (32 / E), r = E * (t % (32 / E));
#pragma unroll
        for (int i = 0; i < E; ++i) s[k * LD + r + i] = v[i];
    }
}

template <int BK>
__global__ __launch_bounds__(256) void moe_gemm_kernel(const MoeGemmParams p) {
    constexpr int LD = 32 + 16, E = 32 * BK / 256;
    __shared__ __attribute__((aligned(16))) float As[BK * LD];
    __shared__ __attribute__((aligned(16))) float Bs[BK * LD];
    const int e = blockIdx.z, seg0 = p.seg[e], cnt = p.seg[e + 1] - seg0;
    // the token dimension of an operand: its rows when K-major, the reduction when reduction-major
    const int M = (p.A.token && !p.A.tr) ? cnt : p.M;
    const int N = (p.B.token && !p.B.tr) ? cnt : p.N;
    const int K = ((p.A.token && p.A.tr) || (p.B.token && p.B.tr)) ? cnt : p.K;
    const int m0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
    if (m0 >= M || n0 >= N) return;
    const float* Ab = p.A.token ? p.A.base : p.A.table[e];
    const float* Bb = p.B.token ? p.B.base : p.B.table[e];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, wm = w >> 1, wn = w & 1, lr = lane & 15, lq = lane >> 4;
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    float ra[E], rb[E];
    if (K > 0) {
        moe_tile_load<BK>(p.A, Ab, seg0, m0, M, 0, K, t, ra);
        moe_tile_load<BK>(p.B, Bb, seg0, n0, N, 0, K, t, rb);
    }
    for (int k0 = 0; k0 < K; k0 += BK) {
        __syncthreads();
        moe_tile_store<BK>(As, p.A.tr, t, ra);
        moe_tile_store<BK>(Bs, p.B.tr, t, rb);
        __syncthreads();
        if (k0 + BK < K) {
            moe_tile_load<BK>(p.A, Ab, seg0, m0, M, k0 + BK, K, t, ra);
            moe_tile_load<BK>(p.B, Bb, seg0, n0, N, k0 + BK, K, t, rb);
        }
#pragma unroll
        for (int kk = 0; kk < BK; kk += 4) {
            const float a = As[(kk + lq) * LD + wm * 16 + lr];
            const float b = Bs[(kk + lq) * LD + wn * 16 + lr];
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
        }
    }
    float* Cb = p.c_token ? p.C + (size_t)seg0 * p.ldc : p.Ctable[e];
    const int n = n0 + wn * 16 + lr;
    if (n >= N) return;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int m = m0 + wm * 16 + 4 * lq + r;
        if (m < M) {
            float* c = Cb + (size_t)m * p.ldc + n;
            *c = p.accumulate ? *c + acc[r] : acc[r];
        }
    }
}

// ---- element-wise pieces -------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void moe_act_fwd_kernel(const float* __restrict__ g, const float* __restrict__ u, float* __restrict__ h, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) h[i] = siluf_(g[i]) * u[i];
}
__global__ __launch_bounds__(256) void moe_act_bwd_kernel(const float* __restrict__ g, const float* __restrict__ u, const float* __restrict__ dh,
                                                          float* __restrict__ dg, float* __restrict__ du, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float gv = g[i], sg = sigmoidf_(gv), d = dh[i];
        du[i] = d * gv * sg;
        dg[i] = d * u[i] * sg * (1.f + gv * (1.f - sg));
    }
}
// out[t] = sum_k w[t][k] * rows[pos[t][k]]  (w NULL: plain sum) -- a gather per token: no atomics, fixed order
__global__ __launch_bounds__(256) void moe_combine_kernel(const float* __restrict__ rows, const float* __restrict__ w, const int* __restrict__ pos,
                                                          float* __restrict__ out, int K, int D, int accumulate) {
    const int t = blockIdx.x;
    for (int d = threadIdx.x; d < D; d += 256) {
        float acc = 0.f;
        for (int k = 0; k < K; ++k) acc = fmaf(w ? w[(size_t)t * K + k] : 1.f, rows[(size_t)pos[(size_t)t * K + k] * D + d], acc);
        float* o = out + (size_t)t * D + d;
        *o = accumulate ? *o + acc : acc;
    }
}
// backward of the weighted combine: do[pos[t][k]] = w[t][k] * dout[t];  dw[t][k] = <dout[t], o[pos[t][k]]>
__global__ __launch_bounds__(256) void moe_combine_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ o, const float* __restrict__ w,
                                                              const int* __restrict__ pos, float* __restrict__ dor, float* __restrict__ dw, int K, int D) {
    __shared__ float red[4];
    const int t = blockIdx.x;
    for (int k = 0; k < K; ++k) {
        const size_t q = (size_t)pos[(size_t)t * K + k];
        const float wk = w[(size_t)t * K + k];
        float dot = 0.f;
        for (int d = threadIdx.x; d < D; d += 256) {
            const float g = dout[(size_t)t * D + d];
            dor[q * D + d] = wk * g;
            dot = fmaf(g, o[q * D + d], dot);
        }
        dot = wave_sum(dot);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = dot;
        __syncthreads();
        if (threadIdx.x == 0) dw[(size_t)t * K + k] = red[0] + red[1] + red[2] + red[3];
        __syncthreads();
    }
}

int moe_gemm_launch(const MoeGemmParams& p, int E, int max_m, int max_n, int bk32, hipStream_t st) {
    const dim3 grid((unsigned)ceil_div(max_n, 32), (unsigned)ceil_div(max_m, 32), (unsigned)E);
    if (grid.y > 65535 || grid.x > 0x7fffffff) return GFE_ERR_SHAPE;
    if (bk32) hipLaunchKernelGGL((moe_gemm_kernel<32>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((moe_gemm_kernel<128>), grid, dim3(256), 0, st, p);
    return gfe_launch_status();
}

}  // namespace

extern "C" {

int gfe_moe_route(const float* logits, int64_t T, int64_t E, int64_t K, float* rw, int32_t* sel, int32_t* tok_sorted, int32_t* pos, int32_t* seg,
                  void* stream) {
    GFE_REQUIRE(logits && rw && sel && tok_sorted && pos && seg, GFE_ERR_NULL);
    GFE_REQUIRE(T > 0 && E > 1 && E <= MAXE && K >= 1 && K <= E && T * K <= 0x7fffffff, GFE_ERR_SHAPE);
    const size_t lds = (size_t)(RT * E + E + 1) * sizeof(int);           // 65 796 B at E = 64: above the 64 KiB default
    static size_t attr_lds = 0;
    if (lds > 64 * 1024 && lds > attr_lds) { (void)hipFuncSetAttribute((const void*)moe_route_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr_lds = lds; }
    hipLaunchKernelGGL(moe_route_kernel, dim3(1), dim3(RT), lds, (hipStream_t)stream,
                       logits, (int)T, (int)E, (int)K, rw, sel, tok_sorted, pos, seg);
    return gfe_launch_status();
}

int gfe_moe_route_bwd(const float* logits, const int32_t* sel, const float* drw, float* dlogits, int64_t T, int64_t E, int64_t K, void* stream) {
    GFE_REQUIRE(logits && sel && drw && dlogits, GFE_ERR_NULL);
    GFE_REQUIRE(T > 0 && E > 1 && E <= MAXE && K >= 1 && K <= E, GFE_ERR_SHAPE);
    hipLaunchKernelGGL(moe_route_bwd_kernel, dim3((unsigned)ceil_div(T, 256)), dim3(256), 0, (hipStream_t)stream, logits, sel, drw, dlogits, (int)T, (int)E, (int)K);
    return gfe_launch_status();
}

/* rows-of-the-segment GEMM: C[seg[e] + r][n] (+)= sum_k A[row(e, r)][k] * W_e(n, k), row(e, r) = gather ? gather[seg[e] + r] : seg[e] + r;
 * W_e = w_table[e], K-major (element (n, k) at [n * ldw + k]: a Linear's forward) or, with w_tr, reduction-major ([k * ldw + n]: its dgrad). */
int gfe_moe_gemm_rows(const float* A, int64_t lda, const int32_t* gather, const float* const* w_table, int64_t ldw, int w_tr, float* C, int64_t ldc,
                      const int32_t* seg, int64_t E, int64_t P, int64_t N, int64_t K, int accumulate, void* stream) {
    GFE_REQUIRE(A && w_table && C && seg, GFE_ERR_NULL);
    GFE_REQUIRE(E > 0 && E <= MAXE && P > 0 && N > 0 && K > 0 && P <= 0x7fffffff && N <= 0x7fffffff && K <= 0x7fffffff, GFE_ERR_SHAPE);
    MoeGemmParams p;
    p.A = Operand{A, nullptr, lda, gather, 0, 1};
    p.B = Operand{nullptr, w_table, ldw, nullptr, w_tr != 0, 0};
    p.C = C; p.Ctable = nullptr; p.ldc = ldc; p.c_token = 1; p.seg = seg; p.M = 0; p.N = (int)N; p.K = (int)K; p.accumulate = accumulate != 0;
    return moe_gemm_launch(p, (int)E, (int)P, (int)N, 0, (hipStream_t)stream);
}

/* weight gradient of every expert: dW_e[n][k] (+)= sum_{r in segment e} dY[seg[e] + r][n] * X[row(e, r)][k]; dW_e = dw_table[e] (N x K, ld = lddw). */
int gfe_moe_gemm_wgrad(const float* dY, int64_t lddy, const float* X, int64_t ldx, const int32_t* gather, float* const* dw_table, int64_t lddw,
                       const int32_t* seg, int64_t E, int64_t N, int64_t K, int accumulate, void* stream) {
    GFE_REQUIRE(dY && X && dw_table && seg, GFE_ERR_NULL);
    GFE_REQUIRE(E > 0 && E <= MAXE && N > 0 && K > 0 && N <= 0x7fffffff && K <= 0x7fffffff, GFE_ERR_SHAPE);
    MoeGemmParams p;
    p.A = Operand{dY, nullptr, lddy, nullptr, 1, 1};
    p.B = Operand{X, nullptr, ldx, gather, 1, 1};
    p.C = nullptr; p.Ctable = dw_table; p.ldc = lddw; p.c_token = 0; p.seg = seg; p.M = (int)N; p.N = (int)K; p.K = 0; p.accumulate = accumulate != 0;
    return moe_gemm_launch(p, (int)E, (int)N, (int)K, 1, (hipStream_t)stream);
}

int gfe_moe_act_fwd(const float* g, const float* u, float* h, int64_t n, void* stream) {
    GFE_REQUIRE(g && u && h, GFE_ERR_NULL);
    GFE_REQUIRE(n > 0, GFE_ERR_SHAPE);
    int64_t blocks = ceil_div(n, 256); if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(moe_act_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, g, u, h, n);
    return gfe_launch_status();
}
int gfe_moe_act_bwd(const float* g, const float* u, const float* dh, float* dg, float* du, int64_t n, void* stream) {
    GFE_REQUIRE(g && u && dh && dg && du, GFE_ERR_NULL);
    GFE_REQUIRE(n > 0, GFE_ERR_SHAPE);
    int64_t blocks = ceil_div(n, 256); if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(moe_act_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, g, u, dh, dg, du, n);
    return gfe_launch_status();
}
int gfe_moe_combine(const float* rows, const float* w, const int32_t* pos, float* out, int64_t T, int64_t K, int64_t D, int accumulate, void* stream) {
    GFE_REQUIRE(rows && pos && out, GFE_ERR_NULL);
    GFE_REQUIRE(T > 0 && T <= 0x7fffffff && K >= 1 && D > 0 && D <= 0x7fffffff, GFE_ERR_SHAPE);
    hipLaunchKernelGGL(moe_combine_kernel, dim3((unsigned)T), dim3(256), 0, (hipStream_t)stream, rows, w, pos, out, (int)K, (int)D, accumulate != 0);
    return gfe_launch_status();
}
int gfe_moe_combine_bwd(const float* dout, const float* o, const float* w, const int32_t* pos, float* d_rows, float* dw, int64_t T, int64_t K, int64_t D,
                        void* stream) {
    GFE_REQUIRE(dout && o && w && pos && d_rows && dw, GFE_ERR_NULL);
    GFE_REQUIRE(T > 0 && T <= 0x7fffffff && K >= 1 && D > 0 && D <= 0x7fffffff, GFE_ERR_SHAPE);
    hipLaunchKernelGGL(moe_combine_bwd_kernel, dim3((unsigned)T), dim3(256), 0, (hipStream_t)stream, dout, o, w, pos, d_rows, dw, (int)K, (int)D);
    return gfe_launch_status();
}

}  // extern "C"
