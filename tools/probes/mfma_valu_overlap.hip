// Can a gfx950 SIMD run vector instructions in the shadow of a matrix instruction?  (round 6: attn2.hip interleaves one 32x32x16 MFMA with four
// v_exp_f32 per chunk and the matrix and vector times still ADD.)  Per loop iteration: NM independent-enough MFMAs (two accumulator chains) and
// NV vector instructions of one kind, either as blocks (all MFMAs, then all vector work) or interleaved one MFMA : NV / NM vector instructions.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_valu_overlap.hip -o exp_build/mfma_valu_overlap && exp_build/mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

// MODE 0: MFMAs only; 1: vector only; 2: blocks (MFMAs then vector); 3: interleaved.   VK 0: v_exp_f32, 1: v_fma_f32, 2: v_cvt_pk_bf16_f32-like (v_pk_mul_f32)
// MK 0: 32x32x16 bf16 (8 passes), 1: 16x16x32 bf16 (4 passes... x2 to keep the cycles), PER: vector instructions per MFMA
template <int MODE, int VK, int MK, int PER>
__global__ __launch_bounds__(512) void loop(float* out, int iters, unsigned long long* clk) {
    bf16x8 a = __builtin_bit_cast(bf16x8, make_uint4(0x3c003c00u + threadIdx.x, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u)), b = a;
    f32x16 c0 = {0}, c1 = {0};
    f32x4 d0 = {0, 0, 0, 0}, d1 = {0, 0, 0, 0};
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = 0.5f + i * 0.01f + threadIdx.x * 1e-7f;
    const float kb = 1.0001f, kc = 1e-6f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        auto mfma = [&](int i) {
            if constexpr (MK == 0) {
                if (i & 1) c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0); else c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
            } else {
                if (i & 1) d1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, d1, 0, 0, 0); else d0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, d0, 0, 0, 0);
            }
        };
        auto vec = [&](int i) {
            if constexpr (VK == 0) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i & 15]));
            if constexpr (VK == 1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i & 15]) : "v"(kb), "v"(kc));
            if constexpr (VK == 2) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[i & 15]) : "v"(kb));
        };
        if constexpr (MODE == 0 || MODE == 2) {
#pragma unroll
            for (int i = 0; i < 8; ++i) mfma(i);
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (MODE == 1 || MODE == 2) {
#pragma unroll
            for (int i = 0; i < 8 * PER; ++i) vec(i);
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (MODE == 3) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                mfma(i);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int k = 0; k < PER; ++k) vec(i * PER + k);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += v[i] + c0[i] + c1[i];
    s += d0[0] + d1[0];
    if (s == 12345.678f) out[threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

template <int MODE, int VK, int MK, int PER>
void run(const char* name, int waves_per_simd, float* d, unsigned long long* clk) {
    const int threads = 256 * waves_per_simd, blocks = 256, iters = 2000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    loop<MODE, VK, MK, PER><<<blocks, threads>>>(d, iters, clk);
    (void)hipDeviceSynchronize();
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        loop<MODE, VK, MK, PER><<<blocks, threads>>>(d, iters, clk);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    unsigned long long h[2]; (void)hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    const double ghz = (double)h[0] / (double)h[1] * 0.1;
    const double cyc = best * 1e-3 * ghz * 1e9 / (double)iters;        // SIMD cycles per iteration (8 MFMAs + 8 PER vector instructions per wave)
    printf("%-44s waves/SIMD=%d: %.3f ms  clock %.2f GHz  %7.1f cycles per iteration per SIMD = %6.1f per wave\n", name, waves_per_simd, best, ghz, cyc, cyc / waves_per_simd);
}

#define ROW(VK, MK, PER, label) \
    run<0, VK, MK, PER>(label ": 8 MFMA only", w, d, clk); \
    run<1, VK, MK, PER>(label ": vector only", w, d, clk); \
    run<2, VK, MK, PER>(label ": blocks", w, d, clk); \
    run<3, VK, MK, PER>(label ": interleaved", w, d, clk);

// The attention kernel's data flow: per phase a 4-step S chain into one score accumulator (from zero) + 4 PV steps into two output
// accumulators, while the exponentials work IN PLACE on the other score accumulator (written by the previous phase's chain); the PV steps
// read the probabilities of the phase before as their B operand.  Two phases per iteration.  Ingredients of attn2 added one by one:
// LDS = the fragment reads (4 ds_read_b128 + 8 ds_read_b64_tr_b16 per phase, each refilling an A operand behind its MFMA);
// PACK = 0: the B operand is raw score bits; 1: v_cvt_pk_bf16_f32 x 8; 2: + v_permlane32_swap x 4; 3: + the four 4x4x4 row-sum MFMAs.
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;
template <int NEXP, int LDS, int PACK>
__global__ __launch_bounds__(512) void flow(float* out, int iters, unsigned long long* clk) {
    __shared__ __attribute__((aligned(16))) uint8_t sm[32768];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) reinterpret_cast<unsigned*>(sm)[i] = 0x3c003c00u;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    bf16x8 kf[4], vf[4];
    for (int i = 0; i < 4; ++i) { kf[i] = __builtin_bit_cast(bf16x8, make_uint4(0x3c003c00u + threadIdx.x, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u)); vf[i] = kf[i]; }
    const bf16x8 q = kf[0];
    const f32x16 z = {0};
    f32x16 s0 = {0}, s1 = {0}, o0 = {0}, o1 = {0};
    bf16x8 pb0[2] = {q, q}, pb1[2] = {q, q};
    f32x4 la = {0, 0, 0, 0}, lb = {0, 0, 0, 0};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int ph = 0; ph < 2; ++ph) {
            f32x16& sc = ph ? s0 : s1;            // chain target
            f32x16& se = ph ? s1 : s0;            // exponentials in place
            bf16x8 (&pin)[2] = ph ? pb1 : pb0;    // probabilities of the phase before (B operand of the PV steps)
            bf16x8 (&pout)[2] = ph ? pb0 : pb1;   // this phase's softmax output
            unsigned w[4];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                constexpr int ORD[8] = {0, 1, 4, 2, 5, 3, 6, 7};
                const int m = ORD[i];
                if (m < 4) {
                    sc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[m], q, m == 0 ? z : sc, 0, 0, 0);
                    if constexpr (LDS) kf[m] = *reinterpret_cast<const bf16x8*>(sm + ((it + m) & 7) * 1024 + lane * 16);
                } else {
                    const int u = m - 4;
                    if (u & 1) o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[u], pin[u >> 1], o1, 0, 0, 0); else o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[u], pin[u >> 1], o0, 0, 0, 0);
                    if constexpr (LDS) {
                        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(sm + 8192 + ((it + u) & 7) * 1024 + lane * 8));
                        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(sm + 8192 + ((it + u) & 7) * 1024 + 512 + lane * 8));
                        union { struct { s16x4 a, b; } h; bf16x8 v; } uu; uu.h.a = lo; uu.h.b = hi; vf[u] = uu.v;
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                if (i < 4) {
#pragma unroll
                    for (int k = 0; k < NEXP; ++k) { const int r = (i * NEXP + k) & 15; se[r] = __builtin_amdgcn_exp2f(se[r]); }
                }
                const int tt = i == 2 || i == 3 || i == 4 ? 0 : 1;
                if constexpr (PACK == 0) {
                    if (i == 4 || i == 6) pout[tt] = __builtin_bit_cast(bf16x8, make_uint4(__float_as_uint(se[8 * tt]), __float_as_uint(se[8 * tt + 1]), __float_as_uint(se[8 * tt + 2]), __float_as_uint(se[8 * tt + 3])));
                } else {
                    if (i == 2 || i == 5) {
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            typedef __attribute__((ext_vector_type(2))) float f32x2_t; typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
                            const f32x2_t v2 = {se[8 * tt + 2 * k], se[8 * tt + 2 * k + 1]};
                            w[k] = __builtin_bit_cast(unsigned, __builtin_convertvector(v2, bf16x2_t));
                        }
                    }
                    if (i == 3 || i == 6) {
                        if constexpr (PACK >= 2) {
                            const auto x0 = __builtin_amdgcn_permlane32_swap(w[0], w[2], false, false);
                            const auto x1 = __builtin_amdgcn_permlane32_swap(w[1], w[3], false, false);
                            w[0] = x0[0]; w[1] = x1[0]; w[2] = x0[1]; w[3] = x1[1];
                        }
                        pout[tt] = __builtin_bit_cast(bf16x8, make_uint4(w[0], w[1], w[2], w[3]));
                    }
                    if constexpr (PACK >= 3) {
                        if (i == 4 || i == 7) {
                            const s16x4 ones = {0x3f80, 0x3f80, 0x3f80, 0x3f80};
                            la = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(ones, __builtin_bit_cast(s16x4, u32x2{w[0], w[1]}), la, 0, 0, 0);
                            lb = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(ones, __builtin_bit_cast(s16x4, u32x2{w[2], w[3]}), lb, 0, 0, 0);
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = la[0] + lb[0];
    for (int i = 0; i < 16; ++i) s += s0[i] + s1[i] + o0[i] + o1[i];
    if (s == 12345.678f) out[threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}
template <int NEXP, int LDS, int PACK>
void run_flow(int waves_per_simd, float* d, unsigned long long* clk) {
    const int threads = 256 * waves_per_simd, blocks = 256, iters = 2000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    flow<NEXP, LDS, PACK><<<blocks, threads>>>(d, iters, clk);
    (void)hipDeviceSynchronize();
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        flow<NEXP, LDS, PACK><<<blocks, threads>>>(d, iters, clk);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    unsigned long long h[2]; (void)hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    const double ghz = (double)h[0] / (double)h[1] * 0.1;
    const double cyc = best * 1e-3 * ghz * 1e9 / (double)iters;
    printf("attention data flow: 16 MFMA + %2d exp2, lds reads %d, pack level %d   waves/SIMD=%d: %.3f ms  clock %.2f GHz  %7.1f cycles per iteration per SIMD = %6.1f per wave\n", 8 * NEXP, LDS, PACK, waves_per_simd, best, ghz, cyc, cyc / waves_per_simd);
}

int main() {
    float* d; (void)hipMalloc(&d, 8192);
    unsigned long long* clk; (void)hipMalloc(&clk, 16);
    for (int w : {1, 2}) {
        ROW(0, 0, 4, "32x32x16 + 4 v_exp per MFMA")
        ROW(1, 0, 6, "32x32x16 + 6 v_fma per MFMA")
        ROW(2, 0, 6, "32x32x16 + 6 v_mul per MFMA")
        ROW(0, 1, 2, "16x16x32 + 2 v_exp per MFMA")
        ROW(1, 1, 3, "16x16x32 + 3 v_fma per MFMA")
    }
    for (int w : {1, 2}) { run_flow<0, 0, 0>(w, d, clk); run_flow<4, 0, 0>(w, d, clk); run_flow<4, 1, 0>(w, d, clk); run_flow<4, 0, 1>(w, d, clk); run_flow<4, 0, 2>(w, d, clk); run_flow<4, 0, 3>(w, d, clk); run_flow<4, 1, 3>(w, d, clk); }
    return 0;
}
