// Exact-f32 GEMM on the f32 matrix cores (v_mfma_f32_16x16x4_f32) for the trainable head's small Linears.
//
// The reference trains the classifier head in fp32 (classify_mamba.py:69-74: the accelerate/fp16 path is commented out), and its
// Mamba projections run at M = B*37 rows (cross_atten/mamba.py:204, 235-238, 223), the cross-attention q / out projections and the
// GEGLU feed-forward at M = B rows (sd_cross_atten.py:42-45, corss_ft_transformer.py:15-22): launch-bound sizes where bf16 operands
// buy nothing and cost parity.  This kernel keeps f32 operands end to end: the f32-input MFMA is bit-for-bit a
// k-ordered fmaf chain (cdna_hip_programming.md 3, 'FP32-input MFMA'), 157 TFLOP/s dense -- ample for <= 300-row problems.
//
//   C[M][N] (+)= op(A)[M][K] . op(B)[N][K]^T (+ bias[n])       all f32, no alignment requirements (guarded scalar loads: the
//   1-wide logit layer and the 25/37-wide test models take the same path)
//   a_tr / b_tr: the operand is stored reduction-major, element (row, k) at base[k * ld + row] -- the dgrad / wgrad forms of a
//   Linear without a transpose pass:  forward y = x W^T: (x, W);  dgrad dx = dy W: (dy, W reduction-major);
//   wgrad dW = dy^T x: (dy reduction-major, x reduction-major, accumulate into the gradient buffer).
//
// Tile T x T x 32 (T = 64 or 32), 4 waves as 2 x 2, each wave (T/32)^2 MFMA tiles of 16 x 16; both operands live in LDS as [k][row]
// (row stride T + 16 floats: the two k values a 32-lane half reads sit on disjoint bank halves), next tile register-prefetched under
// the MFMAs (16-byte loads when base, ld and the extent allow).  The f32 MFMA is 32 cycles per instruction and a k-step costs one
// exposed memory round trip (one tile of prefetch), so these launch-bound shapes (<= 300 rows) want MANY SMALL blocks with DEEP
// k-steps: T = 32 with 128-deep tiles unless 64 x 64 x 32 tiles already fill the chip.  split_k > 1: blockIdx.z takes a K range and adds with f32 atomics (atomics cost more than they save
// beyond a few splits: 2.4 MB of output x 8 splits is 15 us of atomic traffic).
#include <cstdlib>
#include "common.h"

typedef __attribute__((ext_vector_type(4))) float f32x4;

namespace {


struct GemmF32Params {
    const float* A; const float* B; float* C; const float* bias;
    int64_t lda, ldb, ldc;
    int M, N, K, a_tr, b_tr, accumulate, a_vec, b_vec, ksplit;
    float* part;             // split-K partials [range][M][N], or NULL (f32 atomics)
};

// E = T * BK / 256 elements of a T-row x BK-deep operand tile per thread (256 threads).  K-major source: row = t % T, k = E (t / T) + i
// (consecutive lanes take consecutive ROWS: their LDS stores, E k-rows of T + 16 floats apart, then fall on consecutive banks -- with
// consecutive lanes on consecutive k chunks, 16 (T + 16) floats apart = a multiple of 64 banks, every store was an 8-way conflict; the two
// k chunks of a wave still share their banks: 2-way).  Reduction-major source: k = t / (T / E), row = E (t % (T / E)) + i.  vec: 16-byte
// loads are legal for this operand (host-checked alignment).  The (row, k) -> LDS mapping and the k order of the sums are unchanged.
template <int T, int BK>
__device__ __forceinline__ void f32_tile_load(const float* __restrict__ base, int64_t ld, int tr, int vec, int row0, int nrows, int k0, int kend, int t, float (&v)[T * BK / 256]) {
    constexpr int E = T * BK / 256;
    int outer, inner, outer_n, inner_n;                      // element i of this thread: base[outer * ld + inner + i]
    if (!tr) { outer = row0 + t % T; inner = k0 + E * (t / T); outer_n = nrows; inner_n = kend; }
    else { outer = k0 + t / (T / E); inner = row0 + E * (t % (T / E)); outer_n = kend; inner_n = nrows; }
    const float* q = base + (size_t)outer * ld + inner;
    if (vec && outer < outer_n && inner + E <= inner_n) {
#pragma unroll
        for (int j = 0; j < E / 4; ++j) {
            const f32x4 x = *reinterpret_cast<const f32x4*>(q + 4 * j);
            v[4 * j] = x[0]; v[4 * j + 1] = x[1]; v[4 * j + 2] = x[2]; v[4 * j + 3] = x[3];
        }
    } else {
#pragma unroll
        for (int i = 0; i < E; ++i) v[i] = (outer < outer_n && inner + i < inner_n) ? q[i] : 0.f;
    }
}
// LDS column swizzle of k-row k: columns are XORed with 4 ((k & 3) ^ 2 ((k >> 2) & 1)) -- a permutation of the four 4-float pieces inside
// every aligned 16-column group.  The MFMA fragment reads take 16 consecutive columns of rows k .. k + 3 (k % 4 == 0), a whole group per
// row: still one bank per lane.  The reduction-major tile stores (16-byte vectors, lanes 2k and 2k + 1 on row k, 48-float rows) hit every
// bank-start four times without it; with it the 16 lanes of a store phase cover the 64 banks once.
__device__ __forceinline__ int f32_swz(int k) { return ((k & 3) ^ ((k >> 1) & 2)) << 2; }

template <int T, int BK>
__device__ __forceinline__ void f32_tile_store(float* __restrict__ s, int tr, int t, const float (&v)[T * BK / 256]) {
    constexpr int E = T * BK / 256, LD = T + 16;
    if (!tr) {
        const int r = t % T, k = E * (t / T);
#pragma unroll
        for (int i = 0; i < E; ++i) s[(k + i) * LD + (r ^ f32_swz(k + i))] = v[i];
    } else {
        const int k = t / (T / E), r = E * (t % (T / E));
#pragma unroll
        for (int j = 0; j < E / 4; ++j) *reinterpret_cast<f32x4*>(&s[k * LD + ((r + 4 * j) ^ f32_swz(k))]) = f32x4{v[4 * j], v[4 * j + 1], v[4 * j + 2], v[4 * j + 3]};
    }
}

template <int T, int FBK>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const GemmF32Params p) {
    constexpr int LD = T + 16, NI = T / 32, E = T * FBK / 256;       // NI x NI MFMA tiles of 16 x 16 per wave
    __shared__ __attribute__((aligned(16))) float As[FBK * LD];
    __shared__ __attribute__((aligned(16))) float Bs[FBK * LD];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int wm = w >> 1, wn = w & 1;
    const int m0 = blockIdx.y * T, n0 = blockIdx.x * T;
    const int lr = lane & 15, lq = lane >> 4;
    f32x4 acc[NI][NI];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    float ra[E], rb[E];
    const int kbeg = blockIdx.z * p.ksplit, kend = min(p.K, kbeg + p.ksplit);
    f32_tile_load<T, FBK>(p.A, p.lda, p.a_tr, p.a_vec, m0, p.M, kbeg, kend, t, ra);
    f32_tile_load<T, FBK>(p.B, p.ldb, p.b_tr, p.b_vec, n0, p.N, kbeg, kend, t, rb);
    for (int k0 = kbeg; k0 < kend; k0 += FBK) {
        __syncthreads();                                   // everybody is done with the previous tile
        f32_tile_store<T, FBK>(As, p.a_tr, t, ra);
        f32_tile_store<T, FBK>(Bs, p.b_tr, t, rb);
        __syncthreads();
        if (k0 + FBK < kend) {
            f32_tile_load<T, FBK>(p.A, p.lda, p.a_tr, p.a_vec, m0, p.M, k0 + FBK, kend, t, ra);
            f32_tile_load<T, FBK>(p.B, p.ldb, p.b_tr, p.b_vec, n0, p.N, k0 + FBK, kend, t, rb);
        }
#pragma unroll
        for (int kk = 0; kk < FBK; kk += 4) {
            float a[NI], b[NI];
#pragma unroll
            for (int i = 0; i < NI; ++i) a[i] = As[(kk + lq) * LD + ((wm * (T / 2) + i * 16 + lr) ^ f32_swz(kk + lq))];       // A[row = lane & 15][k = lane >> 4]
#pragma unroll
            for (int j = 0; j < NI; ++j) b[j] = Bs[(kk + lq) * LD + ((wn * (T / 2) + j * 16 + lr) ^ f32_swz(kk + lq))];       // B[k = lane >> 4][col = lane & 15]
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    }
    // D: row = 4 (lane >> 4) + reg, col = lane & 15.  An accumulation target is read up front, every element in flight at once (in program
    // order each read-modify-write would be a dependent round trip: no load moves above the previous element's store).
    float oldv[NI][NI][4];
    const bool rmw = p.accumulate && gridDim.z == 1;
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = n0 + wn * (T / 2) + j * 16 + lr, m = m0 + wm * (T / 2) + i * 16 + 4 * lq + r;
                oldv[i][j][r] = (rmw && n < p.N && m < p.M) ? p.C[(size_t)m * p.ldc + n] : 0.f;
            }
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int n = n0 + wn * (T / 2) + j * 16 + lr;
            if (n >= p.N) continue;
            const float bv = (p.bias && blockIdx.z == 0) ? p.bias[n] : 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + wm * (T / 2) + i * 16 + 4 * lq + r;
                if (m < p.M) {
                    float* c = p.C + (size_t)m * p.ldc + n;
                    const float v = acc[i][j][r] + bv;
                    if (gridDim.z > 1 && p.part) p.part[((size_t)blockIdx.z * p.M + m) * p.N + n] = acc[i][j][r];   // summed in a fixed order afterwards
                    else if (gridDim.z > 1) atomicAdd(c, v);     // C zeroed by the caller (or a gradient buffer being added to)
                    else *c = v + oldv[i][j][r];
                }
            }
        }
}

// C = (accumulate ? C : 0) + bias + part[0] + part[1] + ...: the split-K ranges in a fixed order (bit-reproducible)
__global__ __launch_bounds__(256) void gemm_f32_reduce_kernel(const float* __restrict__ part, float* __restrict__ C, const float* __restrict__ bias,
                                                              int M, int N, int64_t ldc, int nsplit, int accumulate) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < (int64_t)M * N; i += (int64_t)gridDim.x * 256) {
        const int m = (int)(i / N), n = (int)(i - (int64_t)m * N);
        float s = bias ? bias[n] : 0.f;
        for (int z = 0; z < nsplit; ++z) s += part[(size_t)z * M * N + i];
        float* c = C + (size_t)m * ldc + n;
        *c = accumulate ? *c + s : s;
    }
}


// ---- the in-block K split ("ks") path: the head's launch-bound shapes (M <= ~300 rows, both operands 16-byte aligned, K % 16 == 0) ----
// One block owns one output tile and its waves cut K between them; nothing is staged in LDS: a lane loads 16 bytes per operand row
// straight into registers and the four k values it holds feed four MFMAs.  The f32 MFMA reduces over k = lane >> 4 of both operands,
// so giving MFMA i of a 16-deep group the k values {4 (lane >> 4) + i} of A *and* of B is the same sum in another (fixed) order.
//   K-major operand (row r, 16-byte load at k = kbase + 4 (lane >> 4)):  element i -> MFMA i of the group;
//   reduction-major B (BTR; element (n, k) at B[k * ldb + n]): one 16-byte load per MFMA i at k = kbase + 4 (lane >> 4) + i covers
//   columns n0 + 4 (lane & 15) .. + 3 = the wave's four column tiles (tile j holds column 4 (lane & 15) + j: a permutation the
//   epilogue undoes by storing the four tiles of a row as one 16-byte vector).
// Rows / columns beyond M / N are clamped on the load side (their products are never stored).  The waves' partial tiles are summed in a
// fixed binary tree through LDS (bit-reproducible, no second launch, no atomics); wave 0 adds bias / the accumulation target and stores.
// Against the staged kernel above at the head's sizes (the round-4 trace of the step, profiles/r04): its K-major tile stores hit LDS
// 8-way bank conflicts, every 128-deep k-step exposed a barrier pair and a memory round trip, K was cut over the GRID with a reduction
// launch behind 33 of the 88 products of a step, and the one-column-block dgrad of dt_proj (N = 32, K = 1024) ran on 10 blocks for 18 us.
template <int NA, int NB, bool BTR, int GB>
__global__ __launch_bounds__(1024) void gemm_f32_ks_kernel(const GemmF32Params p, const int kw) {
    static_assert(!BTR || NB == 4, "reduction-major B: a 16-byte load spans the wave's four column tiles");
    constexpr int TM = 16 * NA, TN = 16 * NB, NBV = BTR ? 4 : NB;
    extern __shared__ __attribute__((aligned(16))) uint8_t ks_smem[];
    f32x4* red = reinterpret_cast<f32x4*>(ks_smem);           // [slot][tile][lane], nw / 2 slots: half of the waves' tiles per round
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int lr = lane & 15, lq = lane >> 4;
    const int m0 = blockIdx.y * TM, n0 = blockIdx.x * TN;
    f32x4 acc[NA][NB];
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int kbeg = w * kw, kend = min(p.K, kbeg + kw);
    const float* ap[NA];
    const float* bp[BTR ? 1 : NB];
#pragma unroll
    for (int a = 0; a < NA; ++a) ap[a] = p.A + (size_t)min(m0 + 16 * a + lr, p.M - 1) * p.lda + 4 * lq;
    if constexpr (BTR) bp[0] = p.B + (size_t)(4 * lq) * p.ldb + min(n0 + 4 * lr, p.N - 4);
    else {
#pragma unroll
        for (int b = 0; b < NB; ++b) bp[b] = p.B + (size_t)min(n0 + 16 * b + lr, p.N - 1) * p.ldb + 4 * lq;
    }
    auto load = [&](int k, f32x4 (&av)[GB][NA], f32x4 (&bv)[GB][NBV]) {
#pragma unroll
        for (int g = 0; g < GB; ++g) {
            const int kg = k + 16 * g;
#if defined(GFE_KS_EXP_NOLOAD)   // timing experiment only: operands never loaded
            if (false) {
#else
            if (kg < kend) {                                  // wave-uniform
#endif
#pragma unroll
                for (int a = 0; a < NA; ++a) av[g][a] = *reinterpret_cast<const f32x4*>(ap[a] + kg);
#pragma unroll
                for (int i = 0; i < NBV; ++i) {
                    if constexpr (BTR) bv[g][i] = *reinterpret_cast<const f32x4*>(bp[0] + (size_t)(kg + i) * p.ldb);
                    else bv[g][i] = *reinterpret_cast<const f32x4*>(bp[i] + kg);
                }
            } else {
#pragma unroll
                for (int a = 0; a < NA; ++a) av[g][a] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int i = 0; i < NBV; ++i) bv[g][i] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
    };
    f32x4 ca[GB][NA], cb[GB][NBV], na[GB][NA], nb_[GB][NBV];
    if (kbeg < kend) load(kbeg, ca, cb);
    for (int k = kbeg; k < kend; k += 16 * GB) {
        const bool more = k + 16 * GB < kend;
        if (more) load(k + 16 * GB, na, nb_);
#pragma unroll
        for (int g = 0; g < GB; ++g)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int a = 0; a < NA; ++a)
#pragma unroll
                    for (int b = 0; b < NB; ++b) {
                        float bval;
                        if constexpr (BTR) bval = cb[g][i][b]; else bval = cb[g][b][i];
#if defined(GFE_KS_EXP_NOMFMA)    // timing experiment only: one fma per MFMA
                        acc[a][b][0] = fmaf(ca[g][a][i], bval, acc[a][b][0]);
#else
                        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(ca[g][a][i], bval, acc[a][b], 0, 0, 0);
#endif
                    }
        if (more) {
#pragma unroll
            for (int g = 0; g < GB; ++g) {
#pragma unroll
                for (int a = 0; a < NA; ++a) ca[g][a] = na[g][a];
#pragma unroll
                for (int i = 0; i < NBV; ++i) cb[g][i] = nb_[g][i];
            }
        }
    }
    // Wave 0 finishes the tile.  The accumulation target's values are fetched here, all in flight at once and under the tree below (a
    // read-modify-write per element in program order is a dependent L2 round trip each: no load can move above the previous element's store).
    float old[NA][NB][4];
    f32x4 old4[NA][4];
    const bool vec_c = (p.ldc & 3) == 0 && ((uintptr_t)p.C & 15) == 0;
    if (w == 0) {
        if constexpr (BTR) {
#pragma unroll
            for (int a = 0; a < NA; ++a)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int n = n0 + 4 * lr, m = m0 + 16 * a + 4 * lq + r;
                    old4[a][r] = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (p.accumulate && n < p.N && m < p.M) {
                        const float* c = p.C + (size_t)m * p.ldc + n;
                        if (vec_c) old4[a][r] = *reinterpret_cast<const f32x4*>(c);
                        else old4[a][r] = f32x4{c[0], c[1], c[2], c[3]};
                    }
                }
        } else {
#pragma unroll
            for (int b = 0; b < NB; ++b)
#pragma unroll
                for (int a = 0; a < NA; ++a)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int n = n0 + 16 * b + lr, m = m0 + 16 * a + 4 * lq + r;
                        old[a][b][r] = (p.accumulate && n < p.N && m < p.M) ? p.C[(size_t)m * p.ldc + n] : 0.f;
                    }
        }
    }
    // fixed binary tree over the waves: round `half`: waves [half, 2 half) park their tiles, waves [0, half) add them
#if defined(GFE_KS_EXP_NOTREE)   // timing experiment only: the waves' tiles are never summed
    for (int half = 0; half >= 1; half >>= 1) {
#else
    for (int half = nw >> 1; half >= 1; half >>= 1) {
#endif
        if (w >= half && w < 2 * half) {
#pragma unroll
            for (int a = 0; a < NA; ++a)
#pragma unroll
                for (int b = 0; b < NB; ++b) red[((w - half) * NA * NB + a * NB + b) * 64 + lane] = acc[a][b];
        }
        __syncthreads();
        if (w < half) {
#pragma unroll
            for (int a = 0; a < NA; ++a)
#pragma unroll
                for (int b = 0; b < NB; ++b) acc[a][b] += red[(w * NA * NB + a * NB + b) * 64 + lane];
        }
        if (half > 1) __syncthreads();
    }
    if (w != 0) return;
    // D: row = 4 (lane >> 4) + reg, col = lane & 15 of the tile
    if constexpr (BTR) {
        const int n = n0 + 4 * lr;                            // the four tiles of a row: columns n .. n + 3
        if (n >= p.N) return;
        const bool vec = vec_c;
        f32x4 bv4 = f32x4{0.f, 0.f, 0.f, 0.f};
        if (p.bias) bv4 = f32x4{p.bias[n], p.bias[n + 1], p.bias[n + 2], p.bias[n + 3]};
#pragma unroll
        for (int a = 0; a < NA; ++a)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + 16 * a + 4 * lq + r;
                if (m >= p.M) continue;
                float* c = p.C + (size_t)m * p.ldc + n;
                const f32x4 v = (f32x4{acc[a][0][r], acc[a][1][r], acc[a][2][r], acc[a][3][r]} + bv4) + old4[a][r];
                if (vec) {
                    *reinterpret_cast<f32x4*>(c) = v;
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) c[j] = v[j];
                }
            }
    } else {
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const int n = n0 + 16 * b + lr;
            if (n >= p.N) continue;
            const float bv = p.bias ? p.bias[n] : 0.f;
#pragma unroll
            for (int a = 0; a < NA; ++a)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = m0 + 16 * a + 4 * lq + r;
                    if (m >= p.M) continue;
                    p.C[(size_t)m * p.ldc + n] = (acc[a][b][r] + bv) + old[a][b][r];
                }
        }
    }
}

struct KsPlan { int na, nw, kw; };
// true: the shape takes the in-block K split (operands aligned for 16-byte loads, K-major A, K a multiple of 16, launch-bound size)
bool ks_plan(const float* A, int64_t lda, int a_tr, const float* B, int64_t ldb, int b_tr, int64_t M, int64_t N, int64_t K, KsPlan* out) {
    static const bool off = getenv("GFE_F32_NO_KS") != nullptr;       // experiments: the staged kernel everywhere
    if (off || a_tr || (K & 15) || ((uintptr_t)A & 15) || (lda & 3) || ((uintptr_t)B & 15) || (ldb & 3)) return false;
    if (b_tr && ((N & 3) || N < 4)) return false;
    if (ceil_div(M, 64) * ceil_div(N, 64) >= 512) return false;       // enough 64 x 64 tiles to fill the chip: the staged kernel's ground
    // the head's row counts only (B x 37 tokens, B rows).  The frozen generator's own f32 products -- the per-sample effective weights of
    // the collapsed conv1 -> GroupNorm -> conv2 chains, 27 * Cout rows by B * Cin columns -- stay on the staged kernel, whose K cut does not
    // depend on the batch: a volume must come out bit for bit the same whatever batch it rides in (tests/test_configs_gpu.py), and the number
    // of waves per tile below is a function of the tile count
    if (M > 512) return false;
    const int64_t groups = K / 16;
    KsPlan pl;
    pl.na = 2;
    if (b_tr && ceil_div(M, 32) * ceil_div(N, 64) * (groups < 16 ? groups : 16) < 2048) pl.na = 1;
    const int64_t tiles = b_tr ? ceil_div(M, 16 * pl.na) * ceil_div(N, 64) : ceil_div(M, 32) * ceil_div(N, 32);
    // few tiles and a long K with a K-major B (x_proj forward, the 8-row feed-forward's second product): 16 waves per tile are not enough
    // parallelism and each would walk > 2 batches -- the staged kernel with K cut over the grid measured faster there (9.0 vs 12.5 us,
    // 10.3 vs 16.5: profiles/r04/gemm_f32_shapes.txt).  The reduction-major form keeps the in-block split: dt_proj's dgrad writes into a
    // column range of a wider matrix with no split-K workspace, i.e. 10 staged blocks walking K = 1024 (16 us in the head's own trace)
    static const bool force_few = getenv("GFE_F32_KS_FEW") != nullptr;     // experiments: the in-block split also for few-tile / long-K shapes
    if (!b_tr && tiles < 32 && groups > 32 && !force_few) return false;
    // K-major B with enough 32 x 32 tiles to fill the chip without cutting K (in_proj forward: 640): the staged kernel in one launch, 15.9 vs
    // 17.9 us since its tile stores stopped colliding on their LDS banks
    if (!b_tr && tiles >= 512) return false;
    // (Measured and dropped: 64 x 32 tiles for the K-major form and 32 x 64 for every reduction-major shape -- a quarter less operand traffic,
    // half the waves: in_proj forward 17.9 -> 19.2 us, out_proj forward 13.0 -> 19.7, in_proj dgrad 16.4 -> 23.9.  These launches are bound by
    // how many independent load streams are in flight, not by the bytes.)
    int nw = 1;
    while (nw < 16 && tiles * nw < 2048 && 2 * nw <= groups) nw *= 2;
    static const char* nw_env = getenv("GFE_F32_KS_NW");               // experiments: a fixed number of waves per block
    if (nw_env) { nw = atoi(nw_env); while (nw > 1 && nw > groups) nw >>= 1; }
    pl.nw = nw;
    pl.kw = (int)(ceil_div(groups, nw) * 16);
    if (out) *out = pl;
    return true;
}

}  // namespace

extern "C" {

int gfe_gemm_f32_inblock(const float* A, int64_t lda, int a_tr, const float* B, int64_t ldb, int b_tr, int64_t M, int64_t N, int64_t K) {
    return ks_plan(A, lda, a_tr, B, ldb, b_tr, M, N, K, nullptr) ? 1 : 0;
}

int gfe_gemm_f32(const float* A, int64_t lda, int a_tr, const float* B, int64_t ldb, int b_tr, float* C, int64_t ldc,
                 int64_t M, int64_t N, int64_t K, const float* bias, int accumulate, int split_k, float* splitk_ws, void* stream) {
    GFE_REQUIRE(A && B && C, GFE_ERR_NULL);
    GFE_REQUIRE(M > 0 && N > 0 && K > 0 && M <= 0x7fffffff && N <= 0x7fffffff && K <= 0x7fffffff, GFE_ERR_SHAPE);
    GFE_REQUIRE(ceil_div(M, 32) <= 65535 && split_k >= 1 && split_k <= 64, GFE_ERR_SHAPE);
    GemmF32Params p;
    p.A = A; p.B = B; p.C = C; p.bias = bias; p.lda = lda; p.ldb = ldb; p.ldc = ldc;
    p.M = (int)M; p.N = (int)N; p.K = (int)K; p.a_tr = a_tr != 0; p.b_tr = b_tr != 0; p.accumulate = accumulate != 0;
    // 16-byte loads: 16-byte aligned base and leading dimension; a thread's 8 elements run along k (K-major) or along rows
    p.a_vec = ((uintptr_t)A % 16 == 0) && lda % 4 == 0;
    p.b_vec = ((uintptr_t)B % 16 == 0) && ldb % 4 == 0;
    KsPlan ks;
    if (ks_plan(A, lda, a_tr, B, ldb, b_tr, M, N, K, &ks)) {
        p.a_vec = p.b_vec = 1; p.ksplit = (int)K; p.part = nullptr;
        const dim3 block(64 * ks.nw);
        const size_t slot = 1024;                                      // one 16 x 16 f32 tile
        if (!b_tr) hipLaunchKernelGGL((gemm_f32_ks_kernel<2, 2, false, 2>), dim3((unsigned)ceil_div(N, 32), (unsigned)ceil_div(M, 32)), block, (ks.nw / 2) * 4 * slot, (hipStream_t)stream, p, ks.kw);
        else if (ks.na == 2) hipLaunchKernelGGL((gemm_f32_ks_kernel<2, 4, true, 1>), dim3((unsigned)ceil_div(N, 64), (unsigned)ceil_div(M, 32)), block, (ks.nw / 2) * 8 * slot, (hipStream_t)stream, p, ks.kw);
        else hipLaunchKernelGGL((gemm_f32_ks_kernel<1, 4, true, 2>), dim3((unsigned)ceil_div(N, 64), (unsigned)ceil_div(M, 16)), block, (ks.nw / 2) * 4 * slot, (hipStream_t)stream, p, ks.kw);
        return gfe_launch_status();
    }
    p.ksplit = (int)(ceil_div(ceil_div(K, split_k), 128) * 128);
    const unsigned nz = (unsigned)ceil_div(K, p.ksplit);
    p.part = nz > 1 ? splitk_ws : nullptr;
    // 64 x 64 tiles when they alone put a block on every CU; otherwise four times as many 32 x 32 blocks, a quarter of the MFMAs each
    // (from one 64 x 64 tile per CU: the head's weight gradients -- K = 296 token rows, three 128-deep steps of a 32 x 32 tile, each with its
    // exposed round trip -- took two rounds of 32 x 32 blocks; head graph 1 789 -> 1 768 us per replay; 128 tiles: 1 838.  GFE_F32_T64_MIN: experiments)
    static const int64_t t64_min = getenv("GFE_F32_T64_MIN") ? atoll(getenv("GFE_F32_T64_MIN")) : 256;
    if (ceil_div(M, 64) * ceil_div(N, 64) * nz >= t64_min)
        hipLaunchKernelGGL((gemm_f32_kernel<64, 32>), dim3((unsigned)ceil_div(N, 64), (unsigned)ceil_div(M, 64), nz), dim3(256), 0, (hipStream_t)stream, p);
    else
        hipLaunchKernelGGL((gemm_f32_kernel<32, 128>), dim3((unsigned)ceil_div(N, 32), (unsigned)ceil_div(M, 32), nz), dim3(256), 0, (hipStream_t)stream, p);
    if (p.part) {
        int64_t blocks = ceil_div(M * N, 256); if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(gemm_f32_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p.part, C, bias, (int)M, (int)N, ldc, (int)nz, accumulate != 0);
    }
    return gfe_launch_status();
}

}  // extern "C"
