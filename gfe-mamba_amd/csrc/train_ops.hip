// Training-step kernels that are pure HBM streams:
//   * Combine_classfier_vit_mid (classify/classifier.py:324-333): Linear(H*W -> S) applied per channel to cat(mid_input,
//     mid_output); here directly on the generator's channels-last bf16 mid features (no concat / transpose copy), forward
//     and weight gradient (the mid features carry no gradient: classify_mamba.py:100).
//   * per-PARAMETER gradient clip + Adam (classify_mamba.py:64, 106-108) on flat f32 buffers, one pass for all tensors,
//     also refreshing the bf16 copies of the weights the GEMMs read.
//   * transposing f32 -> bf16 copies that build the image condition (cross_atten/mamba_transformer.py:89-94).
#include "common.h"

namespace {

// ---- head Linear forward: out[b][src*C + c][s] += sum_{hw in chunk} mid_src[b][hw][c] * W[s][hw] --------------------
// A skinny reduction over hw (13 824 rows of 512 B per (sample, tensor)): HBM-bound.  grid (chunks, 2*B), blockIdx.y = b*2 + src.
// Thread = (row slot, 16-byte channel chunk): every lane streams; 4 rows in flight per thread; Wt is W transposed to (HW, S) so a
// row's S weights are one float4.  The block's row slots are folded through LDS into ONE partial slab per block (part[chunk][b][src][c][s]);
// mid_linear_fold_kernel sums the chunks in a fixed order and adds the bias: no atomics, the head's first activation is run-to-run
// identical (round 3: one f32 atomic per (channel, s) and block onto a zeroed output).
template <int S>
__global__ __launch_bounds__(256) void mid_linear_fwd_kernel(const bf16_t* __restrict__ mid_in, const bf16_t* __restrict__ mid_out,
                                                             const float* __restrict__ Wt, float* __restrict__ part,
                                                             int HW, int C, int rows_per_block) {
    static_assert(S == 4, "a row's weights are read as one float4");
    extern __shared__ float red[];                       // [row slots][S][C]
    const int b = blockIdx.y >> 1, src = blockIdx.y & 1;
    const bf16_t* mid = (src ? mid_out : mid_in) + (size_t)b * HW * C;
    const int CP = C >> 3, RPP = 256 / CP;               // chunks per row, rows per pass
    const int chunk = threadIdx.x % CP, rs = threadIdx.x / CP;
    const int r0 = blockIdx.x * rows_per_block, r1 = min(HW, r0 + rows_per_block);
    float acc[S][8];
#pragma unroll
    for (int s = 0; s < S; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[s][j] = 0.f;
    if (rs < RPP) {
        for (int rb = r0 + rs; rb < r1; rb += 4 * RPP) {
            uint4 v[4]; float4 w[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int r = rb + k * RPP;
                const int rc = r < r1 ? r : r0;                                  // clamped: the tail contributes with weight 0
                v[k] = *reinterpret_cast<const uint4*>(mid + (size_t)rc * C + chunk * 8);
                w[k] = r < r1 ? *reinterpret_cast<const float4*>(Wt + (size_t)rc * S) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float f[8] = {bf16lo_to_f32(v[k].x), bf16hi_to_f32(v[k].x), bf16lo_to_f32(v[k].y), bf16hi_to_f32(v[k].y),
                                    bf16lo_to_f32(v[k].z), bf16hi_to_f32(v[k].z), bf16lo_to_f32(v[k].w), bf16hi_to_f32(v[k].w)};
                const float ws[4] = {w[k].x, w[k].y, w[k].z, w[k].w};
#pragma unroll
                for (int s = 0; s < S; ++s)
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[s][j] = fmaf(f[j], ws[s], acc[s][j]);
            }
        }
#pragma unroll
        for (int s = 0; s < S; ++s)
#pragma unroll
            for (int j = 0; j < 8; ++j) red[(rs * S + s) * C + chunk * 8 + j] = acc[s][j];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < S * C; i += 256) {
        const int s = i / C, c = i - s * C;
        float t = 0.f;
        for (int k = 0; k < RPP; ++k) t += red[(k * S + s) * C + c];
        part[((size_t)blockIdx.x * gridDim.y + blockIdx.y) * C * S + (size_t)c * S + s] = t;
    }
}
// out[b][src*C + c][s] = bias[s] + sum_chunks part[chunk][b*2 + src][c][s]   (fixed order)
__global__ __launch_bounds__(256) void mid_linear_fold_kernel(const float* __restrict__ part, const float* __restrict__ bias, float* __restrict__ out,
                                                              int nchunks, int64_t n, int S) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;            // four chains, fixed fold (one chain = nchunks dependent round trips)
    int k = 0;
    for (; k + 3 < nchunks; k += 4) {
        t0 += part[(size_t)k * n + i]; t1 += part[(size_t)(k + 1) * n + i]; t2 += part[(size_t)(k + 2) * n + i]; t3 += part[(size_t)(k + 3) * n + i];
    }
    for (; k < nchunks; ++k) t0 += part[(size_t)k * n + i];
    out[i] = ((t0 + t1) + (t2 + t3)) + (bias ? bias[i % S] : 0.f);
}

// ---- head Linear weight gradient: dW[s][hw] = sum_{b,src,c} dout[b][src*C+c][s] * mid_src[b][hw][c] -------------------
// All of dout (B x 2C x S f32, 64 KB at the model's size) sits in LDS; a row of hw is handled by C/8 lanes (16 B each), which stream
// the row's 2*B pieces with all loads in flight, multiply against the LDS copy and fold their S partial sums with DPP adds.
template <int S>
__global__ __launch_bounds__(256) void mid_linear_wgrad_kernel(const bf16_t* __restrict__ mid_in, const bf16_t* __restrict__ mid_out,
                                                               const float* __restrict__ dout, float* __restrict__ dW,
                                                               int B, int HW, int C, int rows_per_block) {
    static_assert(S == 4, "dout rows are read as float4");
    extern __shared__ __attribute__((aligned(16))) float sd[];          // [2B][C][S]
    const int nbs = 2 * B;
    const int CP = C >> 3;                               // lanes per row (a power of two <= 64: checked by the host)
    // LDS image [bs][j][chunk] (channel c = 8 chunk + j): for a fixed j the row's CP lanes then read CONSECUTIVE float4 -- in channel order the
    // lanes were 8 float4 = 128 B apart and every ds_read_b128 of the inner loop was 8-way conflicted: the kernel ran at the LDS's pace (72 us
    // for 113 MB at B = 8; round 5, found in the head's kernel trace)
    for (int i = threadIdx.x; i < nbs * C; i += 256) {
        const int bs = i / C, c = i - bs * C, b = bs >> 1, src = bs & 1;
        reinterpret_cast<float4*>(sd)[bs * C + (c & 7) * CP + (c >> 3)] = *reinterpret_cast<const float4*>(dout + ((size_t)b * 2 * C + src * C + c) * S);
    }
    __syncthreads();
    const int RPW = 64 / CP;                             // rows per wave pass
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int chunk = lane % CP, rsub = lane / CP;
    const int r0 = blockIdx.x * rows_per_block, r1 = min(HW, r0 + rows_per_block);
    for (int rb = r0 + wave * RPW; rb < r1; rb += 4 * RPW) {
        const int r = rb + rsub;
        const int rc = r < r1 ? r : r0;
        float acc[S] = {0.f, 0.f, 0.f, 0.f};
        for (int bs0 = 0; bs0 < nbs; bs0 += 8) {
            uint4 v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int bs = bs0 + k < nbs ? bs0 + k : nbs - 1;
                const bf16_t* mid = ((bs & 1) ? mid_out : mid_in) + ((size_t)(bs >> 1) * HW + rc) * C;
                v[k] = *reinterpret_cast<const uint4*>(mid + chunk * 8);
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (bs0 + k < nbs) {
                    const float f[8] = {bf16lo_to_f32(v[k].x), bf16hi_to_f32(v[k].x), bf16lo_to_f32(v[k].y), bf16hi_to_f32(v[k].y),
                                        bf16lo_to_f32(v[k].z), bf16hi_to_f32(v[k].z), bf16lo_to_f32(v[k].w), bf16hi_to_f32(v[k].w)};
                    const float4* dv = reinterpret_cast<const float4*>(sd) + (size_t)(bs0 + k) * C + chunk;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float4 d = dv[j * CP];
                        acc[0] = fmaf(f[j], d.x, acc[0]); acc[1] = fmaf(f[j], d.y, acc[1]);
                        acc[2] = fmaf(f[j], d.z, acc[2]); acc[3] = fmaf(f[j], d.w, acc[3]);
                    }
                }
            }
        }
        // fold the CP lanes of the row (xor butterflies stay inside the row's aligned lane group)
#pragma unroll
        for (int s = 0; s < S; ++s)
            for (int o = CP >> 1; o > 0; o >>= 1) acc[s] += __shfl_xor(acc[s], o, 64);
        if (chunk == 0 && r < r1) {
#pragma unroll
            for (int s = 0; s < S; ++s) dW[(size_t)s * HW + r] = acc[s];
        }
    }
}

// ---- per-tensor clip + Adam on flat buffers -----------------------------------------------------------------------------
// chunk table (host-built, device-resident): {offset, length, tensor id} with chunks never crossing a tensor boundary.
struct Chunk { int64_t off; int len; int tid; };

// Squared norms WITHOUT atomics (round 4): the clip factor of a tensor must come out bit-identical on every rank and in every run --
// replicas that clip by factors one ulp apart drift apart (tests/test_multirank_gpu.py compares two ranks' parameters with torch.equal).
// Pass 1 leaves one partial per chunk; pass 2 (one block per tensor) finds the tensor's chunk range in the table (chunks are sorted by
// tensor id) and sums its partials in a fixed order.
__global__ __launch_bounds__(256) void sqnorm_kernel(const float* __restrict__ g, const Chunk* __restrict__ chunks, float* __restrict__ part, float gscale) {
    __shared__ float red[16];
    const Chunk ck = chunks[blockIdx.x];
    float s = 0.f;
    const int n4 = ck.len >> 2;                                   // chunk offsets are multiples of 8 elements (FlatAdam): 16-byte lanes
    const float4* g4 = reinterpret_cast<const float4*>(g + ck.off);
    for (int i = threadIdx.x; i < n4; i += 256) {
        const float4 t = g4[i];
        const float a = t.x * gscale, b = t.y * gscale, c = t.z * gscale, d = t.w * gscale;
        s = fmaf(a, a, s); s = fmaf(b, b, s); s = fmaf(c, c, s); s = fmaf(d, d, s);
    }
    for (int i = 4 * n4 + threadIdx.x; i < ck.len; i += 256) { const float v = g[ck.off + i] * gscale; s = fmaf(v, v, s); }
    s = block_sum(s, red);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
}
__device__ __forceinline__ int chunk_lower_bound(const Chunk* __restrict__ chunks, int n, int tid) {     // first chunk with .tid >= tid
    int lo = 0, hi = n;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (chunks[mid].tid < tid) lo = mid + 1; else hi = mid; }
    return lo;
}
__global__ __launch_bounds__(256) void sqnorm_fold_kernel(const Chunk* __restrict__ chunks, int nchunks, const float* __restrict__ part, float* __restrict__ norm2) {
    __shared__ float red[16];
    const int tid = blockIdx.x;
    const int c0 = chunk_lower_bound(chunks, nchunks, tid), c1 = chunk_lower_bound(chunks, nchunks, tid + 1);
    float s = 0.f;
    for (int i = c0 + threadIdx.x; i < c1; i += 256) s += part[i];
    s = block_sum(s, red);
    if (threadIdx.x == 0) norm2[tid] = s;
}

__global__ __launch_bounds__(256) void clip_adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                        float* __restrict__ v, bf16_t* __restrict__ p16, const Chunk* __restrict__ chunks,
                                                        const float* __restrict__ norm2, float gscale, float max_norm, float step_size, float b1, float b2,
                                                        float eps, float rsqrt_bc2) {
    const Chunk ck = chunks[blockIdx.x];
    // torch.nn.utils.clip_grad_norm_ on ONE tensor: coef = max_norm / (norm + 1e-6), clamped to 1 (classify_mamba.py:106-107)
    const float coef = fminf(max_norm / (sqrtf(norm2[ck.tid]) + 1e-6f), 1.0f) * gscale;
    // torch.optim.Adam: p -= (lr / bc1) * m / (sqrt(v) / sqrt(bc2) + eps); lr / bc1 and 1 / sqrt(bc2) come from the host in double
    auto upd = [&](float gi, float& mi, float& vi, float& pi) {
        gi *= coef;
        mi = fmaf(b1, mi, (1.f - b1) * gi);
        vi = fmaf(b2, vi, (1.f - b2) * gi * gi);
        pi = pi - step_size * mi / (sqrtf(vi) * rsqrt_bc2 + eps);
    };
    const int n4 = ck.len >> 2;                                   // 16-byte lanes: seven f32 streams and one bf16 stream at full width
    const float4* g4 = reinterpret_cast<const float4*>(g + ck.off);
    float4* m4 = reinterpret_cast<float4*>(m + ck.off);
    float4* v4 = reinterpret_cast<float4*>(v + ck.off);
    float4* p4 = reinterpret_cast<float4*>(p + ck.off);
    for (int i = threadIdx.x; i < n4; i += 256) {
        const float4 gv = g4[i];
        float4 mv = m4[i], vv = v4[i], pv = p4[i];
        upd(gv.x, mv.x, vv.x, pv.x); upd(gv.y, mv.y, vv.y, pv.y); upd(gv.z, mv.z, vv.z, pv.z); upd(gv.w, mv.w, vv.w, pv.w);
        m4[i] = mv; v4[i] = vv; p4[i] = pv;
        if (p16) reinterpret_cast<uint2*>(p16 + ck.off)[i] = make_uint2(pack_bf16x2(pv.x, pv.y), pack_bf16x2(pv.z, pv.w));
    }
    for (int i = 4 * n4 + threadIdx.x; i < ck.len; i += 256) {
        const int64_t k = ck.off + i;
        float mi = m[k], vi = v[k], pn = p[k];
        upd(g[k], mi, vi, pn);
        m[k] = mi; v[k] = vi; p[k] = pn;
        if (p16) p16[k] = f32_to_bf16(pn);
    }
}

// ---- out[b][c][r] (bf16) = in[b][r][c] (f32): image (HW x D3) -> condition rows (D3 x HW) ------------------------------
__global__ __launch_bounds__(256) void transpose_f32_to_bf16_kernel(const float* __restrict__ in, bf16_t* __restrict__ out,
                                                                    int R, int Cc, int64_t in_batch, int64_t out_batch, int64_t ldo) {
    __shared__ float t[64][65];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const float* ib = in + (size_t)blockIdx.z * in_batch;
    bf16_t* ob = out + (size_t)blockIdx.z * out_batch;
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        const int r = i >> 6, c = i & 63;
        t[r][c] = (r0 + r < R && c0 + c < Cc) ? ib[(size_t)(r0 + r) * Cc + c0 + c] : 0.f;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        const int c = i >> 6, r = i & 63;
        if (r0 + r < R && c0 + c < Cc) ob[(size_t)(c0 + c) * ldo + r0 + r] = f32_to_bf16(t[r][c]);
    }
}

// out[r][b*ldb_cols + col_off + c] (bf16) = in[b][r][c] (f32): rows re-interleaved across the batch (cond^T for the wgrad)
__global__ __launch_bounds__(256) void interleave_rows_kernel(const float* __restrict__ in, bf16_t* __restrict__ out, int B, int R, int Cc,
                                                              int64_t ldo, int per_batch_cols, int col_off) {
    const int64_t total = (int64_t)B * R * Cc;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % Cc); const int64_t t = i / Cc;
        const int r = (int)(t % R), b = (int)(t / R);
        out[(size_t)r * ldo + (size_t)b * per_batch_cols + col_off + c] = f32_to_bf16(in[i]);
    }
}

// column sums of a row-major (M, N) f32 matrix (bias gradients): block = 256 threads covers 64 columns x 4 row-strips
// out[c] (+)= sum over rows [blockIdx.y*rows_per_block, ...) of x[r][c]: 4 row strips x 64 columns per block, strips folded in LDS.
// One row block: plain store / read-modify-write; several: f32 atomics onto a zeroed (or accumulating) out.
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ x, float* __restrict__ out, int M, int N, int64_t ld, int accumulate,
                                                     int rows_per_block, float* __restrict__ part) {
    __shared__ float red[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), strip = threadIdx.x >> 6;
    const int r0 = blockIdx.y * rows_per_block, r1 = min(M, r0 + rows_per_block);
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};         // eight loads in flight per strip (two made 296 rows a 21 us walk), fixed fold
    if (c < N) {
        int r = r0 + strip;
        for (; r + 28 < r1; r += 32) {
#pragma unroll
            for (int i = 0; i < 8; ++i) a[i] += x[(size_t)(r + 4 * i) * ld + c];
        }
        for (; r < r1; r += 4) a[0] += x[(size_t)r * ld + c];
    }
    red[strip][threadIdx.x & 63] = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    __syncthreads();
    if (strip == 0 && c < N) {
        const float t = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
        if (gridDim.y > 1 && part) part[(size_t)blockIdx.y * N + c] = t;              // one slot per row block, folded in order by colsum_fold_kernel
        else if (gridDim.y > 1) atomicAdd(out + c, t);
        else out[c] = accumulate ? out[c] + t : t;
    }
}

__global__ __launch_bounds__(256) void colsum_fold_kernel(const float* __restrict__ part, float* __restrict__ out, int N, int nblk, int accumulate) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= N) return;
    float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;
    int k = 0;
    for (; k + 3 < nblk; k += 4) { t0 += part[(size_t)k * N + c]; t1 += part[(size_t)(k + 1) * N + c]; t2 += part[(size_t)(k + 2) * N + c]; t3 += part[(size_t)(k + 3) * N + c]; }
    for (; k < nblk; ++k) t0 += part[(size_t)k * N + c];
    const float t = (t0 + t1) + (t2 + t3);
    out[c] = accumulate ? out[c] + t : t;
}

int64_t colsum_rblocks(int64_t M, int64_t N) {
    int64_t rblocks = M <= 4096 ? 1 : ceil_div((int64_t)256, ceil_div(N, 64));
    if (rblocks > ceil_div(M, 32)) rblocks = ceil_div(M, 32);
    if (rblocks < 1) rblocks = 1;
    const int64_t rpb = ceil_div(M, rblocks);
    return ceil_div(M, rpb);
}

}  // namespace

extern "C" {

int gfe_colsum_rblocks(int64_t M, int64_t N) { return (M > 0 && N > 0) ? (int)colsum_rblocks(M, N) : 0; }

int gfe_colsum_f32_ws(const float* x, float* out, float* ws, int64_t M, int64_t N, int64_t ld, int accumulate, void* stream) {
    GFE_REQUIRE(x && out && ws, GFE_ERR_NULL);
    GFE_REQUIRE(M > 0 && N > 0 && M <= 0x7fffffff && N <= 0x7fffffff, GFE_ERR_SHAPE);
    const int64_t rblocks = colsum_rblocks(M, N);
    const int rpb = (int)ceil_div(M, rblocks);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(colsum_kernel, dim3((unsigned)ceil_div(N, 64), (unsigned)rblocks), dim3(256), 0, st, x, out, (int)M, (int)N, ld, accumulate, rpb, rblocks > 1 ? ws : nullptr);
    if (rblocks > 1) hipLaunchKernelGGL(colsum_fold_kernel, dim3((unsigned)ceil_div(N, 256)), dim3(256), 0, st, ws, out, (int)N, (int)rblocks, accumulate);
    return gfe_launch_status();
}

int gfe_colsum_f32(const float* x, float* out, int64_t M, int64_t N, int64_t ld, int accumulate, void* stream) {
    GFE_REQUIRE(x && out, GFE_ERR_NULL);
    GFE_REQUIRE(M > 0 && N > 0 && M <= 0x7fffffff && N <= 0x7fffffff, GFE_ERR_SHAPE);
    // the column blocks alone are ceil(N/64): tall inputs (the 3-D ViT's 13 832 token rows) split the rows too until the launch has a few
    // hundred blocks, and then add with f32 atomics; up to 4 096 rows (every bias gradient of the classifier head: 8 ... 296 rows) ONE
    // block owns a column and sums its rows in a fixed order -- the trainable half of the step is run-to-run identical (round 4)
    int64_t rblocks = M <= 4096 ? 1 : ceil_div((int64_t)256, ceil_div(N, 64));
    if (rblocks > ceil_div(M, 32)) rblocks = ceil_div(M, 32);
    if (rblocks < 1) rblocks = 1;
    const int rpb = (int)ceil_div(M, rblocks);
    rblocks = ceil_div(M, rpb);
    hipStream_t st = (hipStream_t)stream;
    if (rblocks > 1 && !accumulate) gfe_zero_async(out, (size_t)N * sizeof(float), st);
    hipLaunchKernelGGL(colsum_kernel, dim3((unsigned)ceil_div(N, 64), (unsigned)rblocks), dim3(256), 0, st, x, out, (int)M, (int)N, ld, accumulate, rpb, (float*)nullptr);
    return gfe_launch_status();
}

static void mid_linear_geometry(int64_t B, int64_t HW, int* rpb, int64_t* chunks) {
    // ~512 blocks: enough to stream at HBM rate
    int64_t want = ceil_div((int64_t)512, 2 * B);
    int r = (int)ceil_div(HW, want);
    if (r < 32) r = 32;
    *rpb = r;
    *chunks = ceil_div(HW, r);
}
int gfe_mid_linear_plan(int64_t B, int64_t HW, int* nchunks) {
    GFE_REQUIRE(nchunks, GFE_ERR_NULL);
    GFE_REQUIRE(B > 0 && HW > 0, GFE_ERR_SHAPE);
    int rpb; int64_t chunks;
    mid_linear_geometry(B, HW, &rpb, &chunks);
    *nchunks = (int)chunks;
    return GFE_OK;
}
int gfe_mid_linear_fwd(const void* mid_in, const void* mid_out, const float* Wt, const float* bias, float* out, float* ws,
                       int64_t B, int64_t HW, int64_t C, int64_t S, void* stream) {
    GFE_REQUIRE(mid_in && mid_out && Wt && out && ws, GFE_ERR_NULL);
    GFE_REQUIRE(B > 0 && B <= 32767 && HW > 0 && C % 8 == 0 && C >= 8 && C <= 2048 && 256 % (C / 8) == 0 && S == 4, GFE_ERR_SHAPE);
    int rpb; int64_t chunks;
    mid_linear_geometry(B, HW, &rpb, &chunks);
    const dim3 grid((unsigned)chunks, (unsigned)(2 * B));
    const size_t lds = (size_t)(256 / (C / 8)) * S * C * sizeof(float);
    GFE_REQUIRE(lds <= 64 * 1024, GFE_ERR_SHAPE);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL((mid_linear_fwd_kernel<4>), grid, dim3(256), lds, st,
                       (const bf16_t*)mid_in, (const bf16_t*)mid_out, Wt, ws, (int)HW, (int)C, rpb);
    const int64_t n = 2 * B * C * S;
    hipLaunchKernelGGL(mid_linear_fold_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, st, ws, bias, out, (int)chunks, n, (int)S);
    return gfe_launch_status();
}

int gfe_mid_linear_wgrad(const void* mid_in, const void* mid_out, const float* dout, float* dW,
                         int64_t B, int64_t HW, int64_t C, int64_t S, void* stream) {
    GFE_REQUIRE(mid_in && mid_out && dout && dW, GFE_ERR_NULL);
    const int64_t CP = C / 8;
    GFE_REQUIRE(B > 0 && HW > 0 && C % 8 == 0 && S == 4 && CP >= 1 && CP <= 64 && (CP & (CP - 1)) == 0, GFE_ERR_SHAPE);
    const size_t lds = (size_t)2 * B * C * S * sizeof(float);
    GFE_REQUIRE(lds <= 128 * 1024, GFE_ERR_SHAPE);
    static size_t attr_lds = 0;
    if (lds > 64 * 1024 && lds > attr_lds) { (void)hipFuncSetAttribute((const void*)mid_linear_wgrad_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr_lds = lds; }
    const int rpb = 32;
    hipLaunchKernelGGL((mid_linear_wgrad_kernel<4>), dim3((unsigned)ceil_div(HW, rpb)), dim3(256), lds,
                       (hipStream_t)stream, (const bf16_t*)mid_in, (const bf16_t*)mid_out, dout, dW, (int)B, (int)HW, (int)C, rpb);
    return gfe_launch_status();
}

int gfe_clip_adam(float* p, const float* g, float* m, float* v, void* p_bf16, const void* chunks, int64_t nchunks,
                  float* norm2, int64_t ntensors, float* partials, float grad_scale, float max_norm, double lr, double beta1, double beta2, double eps,
                  int64_t step, void* stream) {
    GFE_REQUIRE(p && g && m && v && chunks && norm2 && partials, GFE_ERR_NULL);
    GFE_REQUIRE(nchunks > 0 && nchunks <= 0x7fffffff && ntensors > 0 && ntensors <= nchunks && step >= 1, GFE_ERR_SHAPE);
    // bias corrections as torch.optim.Adam computes them (Python doubles): 1 - 0.999^t in f32 loses ~1e-5 relative to cancellation
    const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
    hipStream_t st = (hipStream_t)stream;
    // max_norm = +inf (plain Adam, the generator's optimizer): the clip coefficient is 1 for every tensor, whatever the norms (zeros here)
    if (max_norm < 3.0e38f) {
        hipLaunchKernelGGL(sqnorm_kernel, dim3((unsigned)nchunks), dim3(256), 0, st, g, (const Chunk*)chunks, partials, grad_scale);
        hipLaunchKernelGGL(sqnorm_fold_kernel, dim3((unsigned)ntensors), dim3(256), 0, st, (const Chunk*)chunks, (int)nchunks, partials, norm2);
    }
    hipLaunchKernelGGL(clip_adam_kernel, dim3((unsigned)nchunks), dim3(256), 0, st, p, g, m, v, (bf16_t*)p_bf16, (const Chunk*)chunks,
                       norm2, grad_scale, max_norm, (float)(lr / bc1), (float)beta1, (float)beta2, (float)eps, (float)(1.0 / sqrt(bc2)));
    return gfe_launch_status();
}

int gfe_transpose_f32_to_bf16(const float* in, void* out, int64_t batch, int64_t R, int64_t Cc, int64_t out_batch_stride, int64_t ldo, void* stream) {
    GFE_REQUIRE(in && out, GFE_ERR_NULL);
    GFE_REQUIRE(batch > 0 && batch <= 65535 && R > 0 && Cc > 0, GFE_ERR_SHAPE);
    hipLaunchKernelGGL(transpose_f32_to_bf16_kernel, dim3((unsigned)ceil_div(Cc, 64), (unsigned)ceil_div(R, 64), (unsigned)batch), dim3(256), 0,
                       (hipStream_t)stream, in, (bf16_t*)out, (int)R, (int)Cc, R * Cc, out_batch_stride, ldo);
    return gfe_launch_status();
}

int gfe_interleave_rows_bf16(const float* in, void* out, int64_t B, int64_t R, int64_t Cc, int64_t ldo, int64_t per_batch_cols,
                             int64_t col_off, void* stream) {
    GFE_REQUIRE(in && out, GFE_ERR_NULL);
    GFE_REQUIRE(B > 0 && R > 0 && Cc > 0, GFE_ERR_SHAPE);
    int64_t g = ceil_div(B * R * Cc, 256);
    if (g > 8192) g = 8192;
    hipLaunchKernelGGL(interleave_rows_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, in, (bf16_t*)out, (int)B, (int)R, (int)Cc,
                       ldo, (int)per_batch_cols, (int)col_off);
    return gfe_launch_status();
}

}  // extern "C"
