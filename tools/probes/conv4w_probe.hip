// Timing probe (NOT a convolution: operands are whatever the buffers hold) of the conv block the verdicts of rounds 2-4 asked for and DESIGN 4.1 / 8 only priced:
// 4 waves per CU at up to 512 registers, each wave two d-planes of an 8x8x8 tile (8 voxel tiles x 4 channel tiles = 32 accumulators = 128 registers), TWO accumulator
// sets so that the finished tile's epilogue (bias add, ReLU, bf16 pack, 32-byte stores) is spread over the next tile's stages, every wave staging its own share of the
// LDS-DMA pieces (per (tile, 32-channel slab) unit: 16 tile pieces + 27 weight pieces per wave).  Same work per CU as conv_igemm_kernel<4,3,true> on 64 -> 64 @96^3, B = 8:
// 54 tiles x 2 units x 9 stages x 3 taps x 32 MFMA 16x16x32 per wave, 12 fragment reads per tap, the same LDS image sizes (2 x 64 KB tile buffers, 2 x 12 KB weight
// stages), one barrier per stage.  The DMA sources are CONTIGUOUS 1 KB pieces (the real halo gather is 64-byte halves of 128-byte voxels: slower), the fragment reads are
// lane-linear (conflict-free like the real swizzled image): the probe is an upper bound for the structure.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/conv4w_probe.hip -o /tmp/conv4w && /tmp/conv4w
// Variants: overlap = epilogue pieces inside the next tile's stages (two sets); serial = epilogue after the tile like today's kernel; nodma = no staging at all; noepi.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef int v4i_t __attribute__((ext_vector_type(4)));
typedef unsigned v4u_t __attribute__((ext_vector_type(4)));

constexpr int A_BYTES = 65536, W_BYTES = 12288, NTILE = 54;

struct P { const uint8_t* x; const uint8_t* w; uint8_t* y; size_t x_bytes; };

__device__ __forceinline__ v4i_t make_rsrc(const void* base, unsigned bytes) {
    const uint64_t a = (uint64_t)base;
    v4i_t r;
    r.x = __builtin_amdgcn_readfirstlane((int)(uint32_t)a);
    r.y = __builtin_amdgcn_readfirstlane((int)((uint32_t)(a >> 32) & 0xffffu));
    r.z = __builtin_amdgcn_readfirstlane((int)bytes);
    r.w = 0x00020000;
    return r;
}
__device__ __forceinline__ void dma16(const v4i_t& rs, unsigned lds_base, unsigned voff, unsigned soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" :: "s"(lds_base), "v"(voff), "s"(rs), "s"(soff) : "memory");
}
__device__ __forceinline__ uint32_t pack2(float a, float b) {
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
    bf16x2 v = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(uint32_t, v);
}

template <bool OVERLAP, bool DMA, bool EPI>
__global__ __launch_bounds__(256) void conv4w_kernel(const P p) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(1024))) uint8_t smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lq = lane >> 4, lr = lane & 15;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)smem;
    const v4i_t rsx = make_rsrc(p.x + (size_t)blockIdx.x * NTILE * 2 * A_BYTES, NTILE * 2 * A_BYTES), rsw = make_rsrc(p.w, 27 * 2 * 4096);
    const __amdgpu_buffer_rsrc_t rsy = __builtin_amdgcn_make_buffer_rsrc((void*)(p.y + (size_t)blockIdx.x * NTILE * 65536), 0, NTILE * 65536, 0x00020000);
    f32x4 acc[8][4], old[8][4];
#pragma unroll
    for (int v = 0; v < 8; ++v)
#pragma unroll
        for (int c = 0; c < 4; ++c) { acc[v][c] = f32x4{0.f, 0.f, 0.f, 0.f}; old[v][c] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    const f32x4 bias = {0.25f, -0.5f, 0.125f, 1.0f};
    const unsigned lane16 = (unsigned)lane * 16;
    int old_tile = 0;
    // one epilogue piece = the 16 channels of one voxel per lane (4 accumulators): bias, ReLU on the packed words, two 16-byte stores
    auto epi_piece = [&](f32x4 (&a)[8][4], int v, int tile) {
        uint32_t pk[8];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const f32x4 t = a[v][c] + bias;
            pk[2 * c] = pack2(t[0], t[1]); pk[2 * c + 1] = pack2(t[2], t[3]);
        }
        typedef short s16x2 __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int j = 0; j < 8; ++j) pk[j] = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2, pk[j]), s16x2{0, 0}));
        const unsigned o = (unsigned)tile * 65536u + (unsigned)((wave * 8 + v) * 16 + lr) * 128u + (unsigned)lq * 32u;
        __builtin_amdgcn_raw_buffer_store_b128(v4u_t{pk[0], pk[1], pk[2], pk[3]}, rsy, o, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(v4u_t{pk[4], pk[5], pk[6], pk[7]}, rsy, o + 16, 0, 0);
    };
    // prologue: unit 0's tile and stage 0's weights
    if (DMA) {
#pragma unroll
        for (int j = 0; j < 3; ++j) dma16(rsw, lds0 + 2 * A_BYTES + (wave * 3 + j) * 1024, lane16, (unsigned)(wave * 3 + j) * 1024u);
#pragma unroll
        for (int j = 0; j < 16; ++j) dma16(rsx, lds0 + (wave * 16 + j) * 1024, lane16, (unsigned)(wave * 16 + j) * 1024u);
    }
    int gstage = 0;
    for (int tile = 0; tile < NTILE; ++tile) {
#pragma unroll
        for (int unit = 0; unit < 2; ++unit) {
            const int u = tile * 2 + unit;
            const uint8_t* sA = smem + (u & 1) * A_BYTES;
#pragma unroll
            for (int s = 0; s < 9; ++s, ++gstage) {
                // everything this stage reads has landed: at s == 0 the tile burst of the previous unit's stage 0 and this stage's weights; later only the weights
                // (issued AFTER the burst at s == 0: vmcnt retires in order, so from s == 1 on the wait covers the burst as well -- two stages after it went out)
                if (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                if (DMA) {
                    const unsigned wdst = lds0 + 2 * A_BYTES + ((gstage + 1) & 1) * W_BYTES;
#pragma unroll
                    for (int j = 0; j < 3; ++j) dma16(rsw, wdst + (wave * 3 + j) * 1024, lane16, (unsigned)(((s + 1) % 9) * 12 + wave * 3 + j) * 1024u);
                }
                const uint8_t* sW = smem + 2 * A_BYTES + (gstage & 1) * W_BYTES;
#pragma unroll
                for (int tl = 0; tl < 3; ++tl) {
                    bf16x8 wf[4], xf[8];
#pragma unroll
                    for (int c = 0; c < 4; ++c) wf[c] = *reinterpret_cast<const bf16x8*>(sW + tl * 4096 + c * 1024 + lane16);
#pragma unroll
                    for (int v = 0; v < 8; ++v) xf[v] = *reinterpret_cast<const bf16x8*>(sA + (((((wave * 8 + v) * 2 + (s & 1)) * 1024) + ((tl + s) & 3) * 16384) & (A_BYTES - 1)) + lane16);
#pragma unroll
                    for (int c = 0; c < 4; ++c)
#pragma unroll
                        for (int v = 0; v < 8; ++v) acc[v][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[c], xf[v], acc[v][c], 0, 0, 0);
                    if (DMA && s == 0 && tl == 0 && u + 1 < 2 * NTILE) {
                        // the next unit's tile, one burst per wave behind the first tap (the real kernel's placement)
                        const unsigned adst = lds0 + ((u + 1) & 1) * A_BYTES;
#pragma unroll
                        for (int j = 0; j < 16; ++j) dma16(rsx, adst + (wave * 16 + j) * 1024, lane16, (unsigned)(u + 1) * (unsigned)A_BYTES + (unsigned)(wave * 16 + j) * 1024u);
                    }
                }
                if (OVERLAP && EPI) {
                    // the previous tile's epilogue, one voxel tile per stage over the first eight stages of this tile's first unit
                    if (unit == 0 && s < 8 && tile > 0) epi_piece(old, s, old_tile);
                }
            }
        }
        if (OVERLAP) {
#pragma unroll
            for (int v = 0; v < 8; ++v)
#pragma unroll
                for (int c = 0; c < 4; ++c) { old[v][c] = acc[v][c]; acc[v][c] = f32x4{0.f, 0.f, 0.f, 0.f}; }
            old_tile = tile;
        } else {
            if (EPI) {
#pragma unroll
                for (int v = 0; v < 8; ++v) epi_piece(acc, v, tile);
            }
#pragma unroll
            for (int v = 0; v < 8; ++v)
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[v][c] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    if (OVERLAP && EPI) {
#pragma unroll
        for (int v = 0; v < 8; ++v) epi_piece(old, v, old_tile);
    }
    if (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    {   // keep the accumulators alive in the variants that never store them
        float sink = 0.f;
#pragma unroll
        for (int v = 0; v < 8; ++v)
#pragma unroll
            for (int c = 0; c < 4; ++c) sink += acc[v][c][0] + acc[v][c][3] + old[v][c][1];
        if (sink == 12345.678f) p.y[threadIdx.x] = 1;
    }
#endif
}

template <bool OVERLAP, bool DMA, bool EPI>
static void run(const char* name, const P& p) {
    const size_t lds = 2 * A_BYTES + 2 * W_BYTES;
    (void)hipFuncSetAttribute((const void*)conv4w_kernel<OVERLAP, DMA, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) conv4w_kernel<OVERLAP, DMA, EPI><<<256, 256, lds>>>(p);
    (void)hipDeviceSynchronize();
    float best = 1e9f, sum = 0.f;
    const int reps = 20;
    for (int i = 0; i < reps; ++i) {
        (void)hipEventRecord(e0);
        conv4w_kernel<OVERLAP, DMA, EPI><<<256, 256, lds>>>(p);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best; sum += ms;
    }
    const hipError_t err = hipGetLastError();
    const double fl = 256.0 * 4 * NTILE * 2 * 9 * 3 * 32 * 16384.0;
    printf("%-44s mean %.3f ms (min %.3f)  %.0f TFLOP/s-equivalent = %.3f of 2500   [%s]\n", name, sum / reps, best, fl / (sum / reps) / 1e9, fl / (sum / reps) / 1e9 / 2500.0, hipGetErrorString(err));
}

__global__ void fill_random_bf16(uint16_t* q, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u + 12345u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        q[i] = (uint16_t)(((h & 1u) << 15) | ((120u + ((h >> 1) & 7u)) << 7) | ((h >> 4) & 127u));      // +-2^-7 .. 2^0, random mantissa
    }
}

int main(int argc, char** argv) {
    const bool random = argc > 1 && argv[1][0] == 'r';
    P p;
    p.x_bytes = (size_t)256 * NTILE * 2 * A_BYTES;                 // 1.8 GB: every (tile, slab) unit of every block reads its own 64 KB (no L2 reuse: the halo overlap is not modelled)
    void *x, *w, *y;
    if (hipMalloc(&x, p.x_bytes) != hipSuccess || hipMalloc(&w, 27 * 2 * 4096) != hipSuccess || hipMalloc(&y, (size_t)256 * NTILE * 65536) != hipSuccess) { printf("alloc failed\n"); return 1; }
    (void)hipMemset(x, 0x3c, p.x_bytes); (void)hipMemset(w, 0x3b, 27 * 2 * 4096); (void)hipMemset(y, 0, (size_t)256 * NTILE * 65536);
    if (random) {
        fill_random_bf16<<<4096, 256>>>((uint16_t*)x, p.x_bytes / 2);
        fill_random_bf16<<<64, 256>>>((uint16_t*)w, 27 * 2 * 4096 / 2);
        (void)hipDeviceSynchronize();
    }
    printf("operands: %s\n", random ? "random bf16 (sign, mantissa, 3 exponent bits)" : "one constant");
    p.x = (const uint8_t*)x; p.w = (const uint8_t*)w; p.y = (uint8_t*)y;
    printf("4-wave / 512-register conv block, timing probe: 256 blocks x 54 tiles (= 64 -> 64 @96^3, B = 8); conv_igemm_kernel<4,3,true> takes 1.29-1.34 ms for the same tiles\n");
    run<true, true, true>("two accumulator sets, epilogue overlapped", p);
    run<false, true, true>("one set, epilogue after the tile", p);
    run<true, false, true>("overlapped, no staging (LDS never refilled)", p);
    run<true, true, false>("staging, no epilogue", p);
    run<true, false, false>("MFMA + fragment reads + barriers only", p);
    return 0;
}
