// Sustained bf16 MFMA rate of the chip (registers only, no memory): what "MFMA-bound" can mean on this box at the clock it holds.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int SHAPE, bool RANDOM>
__global__ __launch_bounds__(512) void mfma_loop(float* out, int iters) {
#if defined(__HIP_DEVICE_COMPILE__)
    // RANDOM: operands with random bit patterns (4 register sets, rotated) -- the clock the chip holds under MFMA load depends on how
    // many bits toggle, constants give the marketing number
    bf16x8 av[4], bv[4];
    unsigned h = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    for (int j = 0; j < 4; ++j)
        for (int i = 0; i < 8; ++i) {
            h = h * 1664525u + 1013904223u;
            const float fa = RANDOM ? ((int)(h >> 8) - (1 << 23)) * (1.0f / (1 << 23)) : 1.0f;
            h = h * 1664525u + 1013904223u;
            const float fb = RANDOM ? ((int)(h >> 8) - (1 << 23)) * (1.0f / (1 << 23)) : 0.5f;
            av[j][i] = (__bf16)fa; bv[j][i] = (__bf16)fb;
        }
    float s = 0.f;
    if constexpr (SHAPE == 16) {
        f32x4 acc[16];
        for (int i = 0; i < 16; ++i) acc[i] = f32x4{0, 0, 0, 0};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[i & 3], bv[(i >> 2) & 3], acc[i], 0, 0, 0);
        }
        for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][3];
    } else {
        f32x16 acc[4];
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[i], bv[(i + r) & 3], acc[i], 0, 0, 0);
        }
        for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][15];
    }
    if (s == 12345.678f) out[threadIdx.x] = s;
#endif
}

template <int SHAPE, bool RANDOM>
void run(int threads, int blocks, int iters, float* d) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    mfma_loop<SHAPE, RANDOM><<<blocks, threads>>>(d, iters);
    (void)hipDeviceSynchronize();
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        mfma_loop<SHAPE, RANDOM><<<blocks, threads>>>(d, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        const double fl = (double)blocks * (threads / 64) * iters * (SHAPE == 16 ? 16 * 16384.0 : 8 * 32768.0);
        printf("mfma %dx%d %s waves/CU=%d blocks=%d iters=%d: %.3f ms  %.0f TFLOP/s\n", SHAPE, SHAPE, RANDOM ? "random-bits" : "constants  ", threads / 64, blocks, iters, ms, fl / ms / 1e9);
    }
}

int main() {
    float* d; (void)hipMalloc(&d, 4096);
    for (int iters : {20000, 200000}) {
        run<16, false>(512, 256, iters, d);      // 2 waves per SIMD, like the conv kernel
        run<16, true>(512, 256, iters, d);
        run<16, true>(256, 256, iters, d);       // 1 wave per SIMD
        run<32, false>(512, 256, iters, d);
        run<32, true>(512, 256, iters, d);
    }
    return 0;
}
