// Sustained VALU issue rates on gfx950 by instruction kind and waves per SIMD -- the numbers the selective scan's floor is priced with
// (DESIGN.md 4.3).  Registers only, no memory.  cycles = SIMD cycles per wave-instruction at the 2.4 GHz nominal clock (the clock the
// chip really holds is printed from s_memtime / s_memrealtime).
//   hipcc --offload-arch=gfx950 -O3 tools/probes/valu_rates.hip -o exp_build/valu_rates && exp_build/valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));

enum { K_FMA, K_EXP, K_PKFMA, K_MIX_1E4F, K_DPPADD, K_MIX_1E4F_1D, K_PKMIX, K_CNDMASK, K_MIX_2E8F_SPLIT };

template <int KIND>
__global__ __launch_bounds__(1024) void loop(float* out, int iters, unsigned long long* clk) {
    float a[8], b = 1.0001f + threadIdx.x * 1e-9f, c = 1e-6f;
    f2 p[8];
    for (int i = 0; i < 8; ++i) { a[i] = 0.5f + i * 0.01f + threadIdx.x * 1e-7f; p[i] = f2{a[i], a[i] * 0.5f}; }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if constexpr (KIND == K_FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                if constexpr (KIND == K_EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
                if constexpr (KIND == K_PKFMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(f2{b, b}), "v"(f2{c, c}));
                if constexpr (KIND == K_CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b) : );
                if constexpr (KIND == K_DPPADD) asm volatile("v_add_f32_dpp %0, %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(b));
                if constexpr (KIND == K_MIX_1E4F) {          // the scan's inner body: 1 exp + 4 plain per state-step
                    asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
                    asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2" : "+v"(p[i].x) : "v"(b), "v"(c));
                }
                if constexpr (KIND == K_MIX_1E4F_1D) {
                    asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
                    asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2" : "+v"(p[i].x) : "v"(b), "v"(c));
                    asm volatile("v_add_f32_dpp %0, %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(p[i].y) : "v"(b));
                }
                if constexpr (KIND == K_PKMIX) {             // two states: 2 exp + 4 packed
                    asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1" : "+v"(a[i]), "+v"(a[(i + 1) & 7]));
                    asm volatile("v_pk_fma_f32 %0, %0, %1, %2\n v_pk_fma_f32 %0, %0, %1, %2\n v_pk_fma_f32 %0, %0, %1, %2\n v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(f2{b, b}), "v"(f2{c, c}));
                }
                if constexpr (KIND == K_MIX_2E8F_SPLIT) {    // even waves only exp, odd waves only fma: do the pipes overlap across waves?
                    if ((threadIdx.x >> 6) & 1) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
                    else asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2" : "+v"(p[i].x) : "v"(b), "v"(c));
                }
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += a[i] + p[i].x + p[i].y;
    if (s == 12345.678f) out[threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

template <int KIND>
void run(const char* name, int per_iter, int waves_per_simd, float* d, unsigned long long* clk) {
    const int threads = 256 * (waves_per_simd > 4 ? 4 : waves_per_simd);       // up to 16 waves per block
    const int blocks = 256 * (waves_per_simd > 4 ? waves_per_simd / 4 : 1);
    const int iters = 4000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    loop<KIND><<<blocks, threads>>>(d, iters, clk);
    (void)hipDeviceSynchronize();
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        loop<KIND><<<blocks, threads>>>(d, iters, clk);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    unsigned long long h[2]; (void)hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    const double ghz = (double)h[0] / (double)h[1] * 0.1;
    const double groups = (double)iters * 64.0 * waves_per_simd;                // instruction groups per SIMD
    const double cyc = best * 1e-3 * ghz * 1e9 / groups;                        // SIMD cycles per group (all waves of the SIMD together)
    printf("%-28s waves/SIMD=%d  %.3f ms  clock %.2f GHz  %.2f cycles per group of %d instr per wave-slot (%.2f / instr)\n", name, waves_per_simd, best, ghz,
           cyc, per_iter, cyc / per_iter);
}

int main() {
    float* d; (void)hipMalloc(&d, 8192);
    unsigned long long* clk; (void)hipMalloc(&clk, 16);
    for (int w : {1, 2, 4, 8}) {
        run<K_FMA>("v_fma_f32", 1, w, d, clk);
        run<K_EXP>("v_exp_f32", 1, w, d, clk);
        run<K_PKFMA>("v_pk_fma_f32", 1, w, d, clk);
        run<K_CNDMASK>("v_cndmask_b32", 1, w, d, clk);
        run<K_DPPADD>("v_add_f32 dpp", 1, w, d, clk);
        run<K_MIX_1E4F>("1 exp + 4 fma", 5, w, d, clk);
        run<K_MIX_1E4F_1D>("1 exp + 4 fma + 1 dpp add", 6, w, d, clk);
        run<K_PKMIX>("2 exp + 4 pk_fma", 6, w, d, clk);
        if (w >= 2) run<K_MIX_2E8F_SPLIT>("odd waves exp / even 4 fma", 1, w, d, clk);
    }
    return 0;
}
