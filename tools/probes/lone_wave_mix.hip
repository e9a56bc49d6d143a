// Marginal cost of one more instruction in a lone wave's vector stream (one wave per SIMD), gfx950.  Base stream = the selective scan's
// step: 3 v_pk_*_f32 + 2 v_exp_f32 + 2 plain f32, software-pipelined (no operand younger than 4 instructions).  Variants add one
// s_waitcnt / s_nop / ds_read_b128 / ds_write_b32 / v_mov per step.   hipcc --offload-arch=gfx950 -O3 tools/probes/lone_wave_mix.hip -o exp_build/lone_wave_mix
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

template <int KIND>
__global__ __launch_bounds__(512) void loop(float* out, int iters, unsigned long long* clk) {
    __shared__ f4 lds[2048];
    f2 x[4], h = {0.1f, 0.2f}, A2 = {-1.1f, -2.2f}, B = {0.3f, 0.4f};
    float p[4], c0 = 0.5f, c1 = 0.25f, dt = 0.01f + threadIdx.x * 1e-6f, dtu = 0.02f;
    f4 r0 = {0, 0, 0, 0}, r1 = r0;
    for (int i = 0; i < 4; ++i) { x[i] = f2{-0.01f * i, -0.02f * i}; p[i] = 0.f; }
    for (int i = threadIdx.x; i < 2048; i += blockDim.x) lds[i] = f4{0.1f, 0.2f, 0.3f, 0.4f};
    __syncthreads();
    const unsigned la = (threadIdx.x & 63) * 16, wa = (threadIdx.x & 63) * 4 + 16384;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), rt0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const int a = s & 3, b = (s + 1) & 3, c = (s + 2) & 3, d = (s + 3) & 3;
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(p[a]) : "v"(h.x), "v"(c0));
            asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(x[c]) : "v"(A2), "v"(f2{dt, dtu}));
            asm volatile("v_exp_f32 %0, %1" : "=v"(x[b].x) : "v"(x[b].x));
            asm volatile("v_pk_fma_f32 %0, %1, %0, %2" : "+v"(h) : "v"(x[a]), "v"(x[d]));
            asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(p[a]) : "v"(h.y), "v"(c1));
            asm volatile("v_exp_f32 %0, %1" : "=v"(x[b].y) : "v"(x[b].y));
            asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(x[d]) : "v"(B), "v"(f2{dt, dtu}));
            if constexpr (KIND == 1) asm volatile("s_waitcnt lgkmcnt(15)");
            if constexpr (KIND == 2) asm volatile("s_nop 0");
            if constexpr (KIND == 3) asm volatile("ds_read_b128 %0, %1" : "=v"(r0) : "v"(la));
            if constexpr (KIND == 4) asm volatile("ds_write_b32 %0, %1" :: "v"(wa), "v"(p[c]));
            if constexpr (KIND == 5) asm volatile("v_mov_b32 %0, %1" : "=v"(p[d]) : "v"(c0));
            if constexpr (KIND == 6) { asm volatile("ds_read_b128 %0, %1" : "=v"(r0) : "v"(la)); asm volatile("ds_write_b32 %0, %1" :: "v"(wa), "v"(p[c])); if (s & 1) asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(r1) : "v"(la)); }
            if constexpr (KIND == 7) { asm volatile("ds_read_b128 %0, %1" : "=v"(r0) : "v"(la)); asm volatile("ds_write_b32 %0, %1" :: "v"(wa), "v"(p[c])); if (s & 1) asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(r1) : "v"(la)); asm volatile("s_waitcnt lgkmcnt(12)"); }
            if constexpr (KIND == 8) { if ((s & 3) == 3) { for (int q = 0; q < 6; ++q) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r0) : "v"(la), "n"(0)); for (int q = 0; q < 4; ++q) asm volatile("ds_write_b32 %0, %1" :: "v"(wa), "v"(p[q])); } }
            if constexpr (KIND == 9) { if ((s & 7) == 7) { for (int q = 0; q < 12; ++q) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r0) : "v"(la), "n"(0)); for (int q = 0; q < 8; ++q) asm volatile("ds_write_b32 %0, %1" :: "v"(wa), "v"(p[q & 3])); } }
            if constexpr (KIND == 10) { if ((s & 3) == 3) { for (int q = 0; q < 6; ++q) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r0) : "v"(la), "n"(0)); } if ((s & 3) == 1) { for (int q = 0; q < 4; ++q) asm volatile("ds_write_b32 %0, %1" :: "v"(wa), "v"(p[q])); } }
            if constexpr (KIND == 11) { if ((s & 3) == 3) { for (int q = 0; q < 6; ++q) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r0) : "v"(la), "n"(0)); for (int q = 0; q < 2; ++q) asm volatile("ds_write2st64_b32 %0, %1, %2 offset0:0 offset1:5" :: "v"(wa), "v"(p[q]), "v"(p[q + 2])); } }
        }
        if constexpr (KIND == 3 || KIND == 4 || KIND >= 6) asm volatile("s_waitcnt lgkmcnt(0)");
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), rt1 = __builtin_amdgcn_s_memrealtime();
    float s = h.x + h.y + r0.x + r1.y;
    for (int i = 0; i < 4; ++i) s += x[i].x + x[i].y + p[i];
    if (s == 12345.678f) out[threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = rt1 - rt0; }
}

template <int KIND>
void run(const char* name, int waves_per_simd, float* d, unsigned long long* clk) {
    const int threads = 256 * waves_per_simd, blocks = 256, iters = 2000;
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
    const double cyc = best * 1e-3 * ghz * 1e9 / ((double)iters * 16.0);
    printf("%-44s waves/SIMD=%d: %.3f ms  clock %.2f GHz  %.2f cycles per step (all waves of the SIMD: one step each)\n", name, waves_per_simd, best, ghz, cyc);
}

int main() {
    float* d; (void)hipMalloc(&d, 8192);
    unsigned long long* clk; (void)hipMalloc(&clk, 16);
    for (int w : {1, 2}) {
        run<0>("base: 3 pk + 2 exp + 2 plain", w, d, clk);
        run<1>("base + s_waitcnt (satisfied)", w, d, clk);
        run<2>("base + s_nop 0", w, d, clk);
        run<8>("base + per 4 steps: 6 reads + 4 writes in a burst", w, d, clk);
        run<9>("base + per 8 steps: 12 reads + 8 writes in a burst", w, d, clk);
        run<10>("base + per 4 steps: 6 reads | 4 writes (two bursts)", w, d, clk);
        run<11>("base + per 4 steps: 6 reads + 2 write2st64 burst", w, d, clk);
        run<5>("base + v_mov_b32", w, d, clk);
        run<3>("base + ds_read_b128", w, d, clk);
        run<4>("base + ds_write_b32", w, d, clk);
        run<6>("base + 1.5 ds_read_b128 + ds_write_b32", w, d, clk);
        run<7>("base + 1.5 ds_read_b128 + ds_write_b32 + waitcnt", w, d, clk);
    }
    return 0;
}
