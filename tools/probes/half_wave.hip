// Does gfx950 skip the second 32-lane pass of a wave64 vector instruction when EXEC[63:32] == 0?  (round 6: the selective scan at B = 8 has
// exactly one wave of work per SIMD; if half-empty waves cost half the vector time, two half waves per SIMD would interleave their issue.)
//   hipcc --offload-arch=gfx950 -O3 tools/probes/half_wave.hip -o exp_build/half_wave && exp_build/half_wave
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ __launch_bounds__(1024) void loop(float* out, int iters, int active_lanes, unsigned long long* clk) {
    float a[8], b = 1.0001f + threadIdx.x * 1e-9f, c = 1e-6f;
    f2 p[8];
    for (int i = 0; i < 8; ++i) { a[i] = 0.5f + i * 0.01f + threadIdx.x * 1e-7f; p[i] = f2{a[i], a[i] * 0.5f}; }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    if ((int)(threadIdx.x & 63) < active_lanes) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 8; ++r) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    if constexpr (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                    if constexpr (KIND == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
                    if constexpr (KIND == 2) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(f2{b, b}), "v"(f2{c, c}));
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
void run(const char* name, int waves_per_simd, int active, float* d, unsigned long long* clk) {
    const int threads = 256 * waves_per_simd, blocks = 256, iters = 4000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    loop<KIND><<<blocks, threads>>>(d, iters, active, clk);
    (void)hipDeviceSynchronize();
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        loop<KIND><<<blocks, threads>>>(d, iters, active, clk);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    unsigned long long h[2]; (void)hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    const double ghz = (double)h[0] / (double)h[1] * 0.1;
    const double cyc = best * 1e-3 * ghz * 1e9 / ((double)iters * 64.0);        // SIMD cycles per instruction slot (all waves of the SIMD issue one each)
    printf("%-14s waves/SIMD=%d active lanes %2d: %.3f ms  clock %.2f GHz  %.2f cycles per round (%.2f per wave-instruction)\n", name, waves_per_simd, active, best, ghz, cyc, cyc / waves_per_simd);
}

int main() {
    float* d; (void)hipMalloc(&d, 8192);
    unsigned long long* clk; (void)hipMalloc(&clk, 16);
    for (int w : {1, 2, 4})
        for (int act : {64, 32, 16}) {
            run<0>("v_fma_f32", w, act, d, clk);
            run<1>("v_exp_f32", w, act, d, clk);
            run<2>("v_pk_fma_f32", w, act, d, clk);
        }
    return 0;
}
