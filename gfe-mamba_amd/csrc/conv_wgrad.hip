// Weight gradient of a 3-D convolution given as a tap list (main_gan_vit.py:76-79 backward of buildingblocks.py:46-52):
//     dW[tap][co][ci] = sum over voxels v of  dout[v][co] * x[v + off(tap)][ci]            (x zero outside the volume)
// on channels-last bf16 tensors, f32 accumulation.  One fused kernel for ALL taps: per tap the product is a (Co x V) @ (V x Ci) GEMM
// whose reduction runs over voxels, and 27 separate GEMMs re-read both tensors 27 times (29 GB for a 64->64 conv at 128^3, batch 2:
// HBM/L2-bound at 95 TFLOP/s).  Here a block owns a (64 output channels x 32 input channels x all taps) slice of dW -- 55,296 f32
// accumulators that never leave its registers -- and streams voxel tiles past them, so each tensor is read about once per slice.
//
// Block = 8 waves (2 per SIMD).  Wave w = (co half  w & 1,  tap group  w >> 1): a 32 x 32 output tile for each of its NI taps
// (taps  g, g + 4, g + 8, ...), v_mfma_f32_32x32x16_bf16, A = dout^T (M = co), B = x (N = ci), K = 16 voxels per instruction.
// Both operands want, per lane, 8 CONSECUTIVE VOXELS of one channel while memory (and the LDS image the DMA writes) is channels-last:
// ds_read_b64_tr_b16 reads a 4-voxel x 16-channel block and hands every lane one channel's four voxels -- the transpose is free.
//
// Staging is LDS-DMA (buffer_load_dwordx4 ... lds), double-buffered per unit = (8x8xTD voxel tile):
//     x halo tile   [(TD+2) x 10 x 10 voxels][32 ci]  64-B rows                 out-of-volume voxels: out-of-range offset -> zeros
//     dout tile     [2 co halves][TD x 8 x 8 voxels][32 co]  64-B rows           (the conv's zero padding, and ragged tiles)
// With 64-B rows a transposed read's 32-lane half covers 4 consecutive rows x 64 B = 256 contiguous bytes: conflict-free, no swizzle.
// A tap is a constant byte offset into the halo image; a K-chunk is two h-rows of 8 voxels (k = 8 * (row parity) + w).
// Blocks split the voxel tiles of one (ci slab, co group) job; each writes its slice of a partials tensor once, and a second small
// kernel sums the splits (deterministic, no atomics).
#include "common.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((address_space(3))) void* lds_void_t;
typedef __attribute__((address_space(3))) s16x4* lds_s16x4_t;

namespace {

constexpr int TD = 4, TH = 8, TW = 8;
constexpr int PD = TD + 2, PH = TH + 2, PW = TW + 2;
constexpr int X_ROWS = PD * PH * PW;                     // 600
constexpr int X_PIECES = (X_ROWS + 15) / 16;             // 38 one-KiB pieces (16 voxel rows each)
constexpr int X_BYTES = X_PIECES * 1024;
constexpr int D_ROWS = TD * TH * TW;                     // 256
constexpr int D_PLANE = D_ROWS * 64;                     // one co half: 16 KiB
constexpr int D_PIECES = 2 * D_ROWS / 16;                // 32
constexpr int BUF_BYTES = X_BYTES + 2 * D_PLANE;         // 71,680
constexpr int PIECES = X_PIECES + D_PIECES;              // 70
constexpr int NWAVES = 8;
constexpr int PER_WAVE = (PIECES + NWAVES - 1) / NWAVES; // 9
constexpr unsigned OOB = 0x80000000u;

struct WgradParams {
    const bf16_t* x; const bf16_t* d; float* part;
    int B, D, H, W, Ci, Co;
    int ntaps, ntd, nth, ntw, ntiles;
    int nslab, njobs, nsplit, xcd_map;
    int tapoff[28];                                      // byte offset of the tap inside the halo image
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t uniform_rsrc(const void* base, unsigned bytes) {
    const uint64_t a = (uint64_t)base;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a), hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
    return __builtin_amdgcn_make_buffer_rsrc((void*)(((uint64_t)hi << 32) | lo), 0, (int)__builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}
__device__ __forceinline__ int rfl(int v) { return __builtin_amdgcn_readfirstlane(v); }

__device__ __forceinline__ bf16x8 tr_frag(const uint8_t* base) {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t)(base));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t)(base + 4 * 64));      // voxels w + 4 .. w + 7: four rows on
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
}

template <int NI>
__global__ __launch_bounds__(NWAVES * 64, 1) void conv_wgrad_kernel(const WgradParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int chalf = wave & 1, tg = wave >> 1;
    // block -> (job, split).  With xcd_map the 8 blocks that are dealt to the 8 XCDs together stay on one job, and the jobs of one tile
    // range sit on the same XCD: they read the same x / dout tiles out of that XCD's L2.
    int job, sp;
    if (p.xcd_map) { job = (blockIdx.x >> 3) % p.njobs; sp = (blockIdx.x & 7) + 8 * (blockIdx.x / (8 * p.njobs)); }
    else { job = blockIdx.x % p.njobs; sp = blockIdx.x / p.njobs; }
    const int slab = job % p.nslab, group = job / p.nslab;
    const int per = (p.ntiles + p.nsplit - 1) / p.nsplit;
    const int t0 = min(p.ntiles, sp * per), t1 = min(p.ntiles, t0 + per);

    f32x16 acc[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    // ---- transposed-read lane bases: 16-lane group j -> k group (voxel row parity) j >> 1, channel block j & 1; lane 4q + pp of the
    // group addresses voxel w = q, channels 4pp .. 4pp + 3 of the block
    const int j = lane >> 4, q4 = (lane >> 2) & 3, pp = lane & 3;
    const int kg = j >> 1, cb = j & 1;
    const int xlane = (kg * PW + q4) * 64 + cb * 32 + pp * 8;
    const int dlane = X_BYTES + chalf * D_PLANE + (kg * TW + q4) * 64 + cb * 32 + pp * 8;
    int xtap[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) xtap[i] = xlane + p.tapoff[min(tg + 4 * i, p.ntaps - 1)];

    // ---- thread-constant DMA coordinates: piece k = wave + 8 jj.  x pieces: voxel row 16k + lane/4 of the halo image;
    // dout pieces: plane (k - 38) / 16, row 16 ((k - 38) % 16) + lane/4.  Packed: ld | lh << 4 | lw << 8 | valid << 12
    int coord[PER_WAVE];
#pragma unroll
    for (int jj = 0; jj < PER_WAVE; ++jj) {
        const int k = wave + NWAVES * jj;
        if (k < X_PIECES) {
            const int v = 16 * k + (lane >> 2);
            const int ld = v / (PH * PW), rem = v - ld * (PH * PW), lh = rem / PW, lw = rem - lh * PW;
            coord[jj] = ld | (lh << 4) | (lw << 8) | ((v < X_ROWS ? 1 : 0) << 12);
        } else {
            const int kd = k - X_PIECES, row = 16 * (kd & 15) + (lane >> 2);
            coord[jj] = (row >> 6) | (((row >> 3) & 7) << 4) | ((row & 7) << 8) | ((k < PIECES ? 1 : 0) << 12);
        }
    }
    const size_t xs_elems = (size_t)p.D * p.H * p.W * p.Ci, ds_elems = (size_t)p.D * p.H * p.W * p.Co;
    const int cchunk = (lane & 3) * 8;                     // the lane's 16 B = 8 channels inside the 32-channel row

    auto dma = [&](int t, int buf) {
        int r = rfl(t);
        const int tw = r % p.ntw; r /= p.ntw;
        const int th = r % p.nth; r /= p.nth;
        const int td = r % p.ntd, b = r / p.ntd;
        const int d0 = td * TD, h0 = th * TH, w0 = tw * TW;
        const __amdgpu_buffer_rsrc_t rx = uniform_rsrc(p.x + (size_t)b * xs_elems, (unsigned)(xs_elems * 2));
        const __amdgpu_buffer_rsrc_t rd = uniform_rsrc(p.d + (size_t)b * ds_elems, (unsigned)(ds_elems * 2));
        uint8_t* base = smem + rfl(buf) * BUF_BYTES;
#pragma unroll
        for (int jj = 0; jj < PER_WAVE; ++jj) {
            const int k = wave + NWAVES * jj;                                  // wave-uniform
            if (k >= PIECES) continue;
            const int c = coord[jj];
            const bool isx = k < X_PIECES;
            const int gd = d0 + (c & 15) - (isx ? 1 : 0), gh = h0 + ((c >> 4) & 15) - (isx ? 1 : 0), gw = w0 + ((c >> 8) & 15) - (isx ? 1 : 0);
            const bool ok = ((c >> 12) & 1) && (unsigned)gd < (unsigned)p.D && (unsigned)gh < (unsigned)p.H && (unsigned)gw < (unsigned)p.W;
            const int vox = (gd * p.H + gh) * p.W + gw;
            if (isx) {
                const unsigned off = ok ? (unsigned)((vox * p.Ci + slab * 32 + cchunk) * 2) : OOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_void_t)(base + k * 1024), 16, off, 0, 0, 0);
            } else {
                const int plane = (k - X_PIECES) >> 4;
                const unsigned off = ok ? (unsigned)((vox * p.Co + group * 64 + plane * 32 + cchunk) * 2) : OOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rd, (lds_void_t)(base + k * 1024), 16, off, 0, 0, 0);
            }
        }
    };

    GFE_FUZZ_INIT();
    if (t0 < t1) dma(t0, 0);
    for (int t = t0; t < t1; ++t) {
        const int buf = (t - t0) & 1;
        GFE_FUZZ();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_waitcnt(0xc07f);
        GFE_FUZZ();
        __builtin_amdgcn_s_barrier();                       // unit t has landed for every wave; nobody still reads the other buffer
        asm volatile("" ::: "memory");
        GFE_FUZZ();
        const uint8_t* sb = smem + buf * BUF_BYTES;
#pragma unroll
        for (int q = 0; q < TD * 4; ++q) {                  // K-chunk: plane q / 4, h rows 2 (q % 4) and + 1
            // the next unit's DMA instructions go out once the first chunk's fragment reads are in flight (their LDS round trip then runs
            // under the ~9 DMA issues instead of in front of the first MFMA): 1.025 -> 0.997 ms
            if (q == 1 && t + 1 < t1) { GFE_FUZZ(); dma(t + 1, buf ^ 1); }
            const int dz = q >> 2, hy0 = (q & 3) * 2;
            const bf16x8 a = tr_frag(sb + dlane + ((dz * TH + hy0) * TW) * 64);
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const bf16x8 bx = tr_frag(sb + xtap[i] + ((dz * PH + hy0) * PW) * 64);
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bx, acc[i], 0, 0, 0);
            }
        }
    }

    // ---- the block's slice of the partials: part[sp][tap][co][ci]; lane = ci column, register r = co row 8 (r / 4) + 4 (lane / 32) + r % 4
    const int ci = slab * 32 + (lane & 31);
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int tap = tg + 4 * i;
        if (tap >= p.ntaps) continue;
        float* o = p.part + (((size_t)sp * p.ntaps + tap) * p.Co + group * 64 + chalf * 32) * p.Ci + ci;
#pragma unroll
        for (int r = 0; r < 16; ++r) o[(size_t)(8 * (r >> 2) + 4 * (lane >> 5) + (r & 3)) * p.Ci] = acc[i][r];
    }
#endif
}

__global__ __launch_bounds__(256) void sum_splits_kernel(const float* __restrict__ part, float* __restrict__ out, int nsplit, int64_t n4) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int k = 0; k < nsplit; ++k) {
            const float4 v = reinterpret_cast<const float4*>(part)[(int64_t)k * n4 + i];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        reinterpret_cast<float4*>(out)[i] = s;
    }
}

int plan_splits(int64_t tiles, int njobs) {
    int64_t ns = 256 / njobs;
    if (ns < 1) ns = 1;
    if (ns > tiles) ns = tiles;
    if (ns >= 8) ns -= ns % 8;
    return (int)ns;
}

}  // namespace

extern "C" {

int gfe_conv3d_wgrad_splits(int64_t B, int64_t D, int64_t H, int64_t W, int64_t Ci, int64_t Co) {
    if (B <= 0 || D <= 0 || H <= 0 || W <= 0 || Ci <= 0 || Co <= 0 || Ci % 32 || Co % 64) return GFE_ERR_SHAPE;
    const int64_t tiles = B * ((D + TD - 1) / TD) * ((H + TH - 1) / TH) * ((W + TW - 1) / TW);
    return plan_splits(tiles, (int)((Ci / 32) * (Co / 64)));
}

int gfe_conv3d_wgrad(const void* x, const void* dout, float* part_ws, float* dw, const int8_t* taps_host, int ntaps,
                     int64_t B, int64_t D, int64_t H, int64_t W, int64_t Ci, int64_t Co, void* stream) {
    GFE_REQUIRE(x && dout && part_ws && dw && taps_host, GFE_ERR_NULL);
    GFE_REQUIRE(B > 0 && D > 0 && H > 0 && W > 0 && Ci > 0 && Co > 0 && Ci % 32 == 0 && Co % 64 == 0 && ntaps > 0 && ntaps <= 27, GFE_ERR_SHAPE);
    GFE_REQUIRE(D * H * W * (Ci > Co ? Ci : Co) * 2 < 0x7fffffffLL, GFE_ERR_SHAPE);          // 32-bit byte offsets inside a sample
    WgradParams p;
    p.x = (const bf16_t*)x; p.d = (const bf16_t*)dout; p.part = part_ws;
    p.B = (int)B; p.D = (int)D; p.H = (int)H; p.W = (int)W; p.Ci = (int)Ci; p.Co = (int)Co;
    p.ntaps = ntaps;
    p.ntd = (int)((D + TD - 1) / TD); p.nth = (int)((H + TH - 1) / TH); p.ntw = (int)((W + TW - 1) / TW);
    const int64_t tiles = B * p.ntd * p.nth * p.ntw;
    GFE_REQUIRE(tiles < 0x7fffffffLL, GFE_ERR_SHAPE);
    p.ntiles = (int)tiles;
    p.nslab = (int)(Ci / 32); p.njobs = p.nslab * (int)(Co / 64);
    p.nsplit = plan_splits(tiles, p.njobs);
    p.xcd_map = (p.nsplit % 8 == 0) ? 1 : 0;
    for (int t = 0; t < 28; ++t) p.tapoff[t] = 0;
    for (int t = 0; t < ntaps; ++t) {
        const int od = taps_host[3 * t], oh = taps_host[3 * t + 1], ow = taps_host[3 * t + 2];
        GFE_REQUIRE(od >= -1 && od <= 1 && oh >= -1 && oh <= 1 && ow >= -1 && ow <= 1, GFE_ERR_SHAPE);
        p.tapoff[t] = (((od + 1) * PH + (oh + 1)) * PW + (ow + 1)) * 64;
    }
    const size_t lds = 2 * (size_t)BUF_BYTES;
    const dim3 grid((unsigned)(p.njobs * p.nsplit));
    hipStream_t st = (hipStream_t)stream;
    if (ntaps > 8) {
        static bool a7 = false;
        if (!a7) { (void)hipFuncSetAttribute((const void*)conv_wgrad_kernel<7>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); a7 = true; }
        hipLaunchKernelGGL(conv_wgrad_kernel<7>, grid, dim3(NWAVES * 64), lds, st, p);
    } else {
        static bool a2 = false;
        if (!a2) { (void)hipFuncSetAttribute((const void*)conv_wgrad_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); a2 = true; }
        hipLaunchKernelGGL(conv_wgrad_kernel<2>, grid, dim3(NWAVES * 64), lds, st, p);
    }
    const int64_t n = (int64_t)ntaps * Co * Ci;
    int64_t blocks = (n / 4 + 255) / 256; if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(sum_splits_kernel, dim3((unsigned)blocks), dim3(256), 0, st, part_ws, dw, p.nsplit, n / 4);
    return gfe_launch_status();
}

}  // extern "C"
