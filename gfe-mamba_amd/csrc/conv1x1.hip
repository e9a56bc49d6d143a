// 1x1x1 convolution Cin -> Cout (+ bias) of a channels-last bf16 tensor as a plain streaming product, with the GroupNorm partials of
// what it stores.  Reference: the encoders' ResNetBlock.conv1 (pytorch3dunet/unet3d/buildingblocks.py:204-208: nn.Conv3d(in, out, 1) when
// in_channels != out_channels; :218-229 its result is conv2's input and conv3's residual).
//
// Why its own kernel (round 6).  Rounds 1-5 ran these through conv3d.hip's tap-list kernel: a 10^3 halo tile of every
// 32-channel slab staged into LDS for ONE tap, 142 us at 64 -> 128 @48^3 and 66 us at 128 -> 256 @24^3 (B = 8) for 340 / 85 MB of traffic.
// A 1x1x1 conv has no halo and no reuse of an activation beyond its own voxel: the MFMA operands ARE the memory layout.
//   * B operand (activations): lane (lr = lane & 15, lq = lane >> 4) of v_mfma_f32_16x16x32_bf16 holds k = 8 lq .. 8 lq + 7 of column lr:
//     k-step s is input channels 32 s + 8 lq .. + 7 of voxel lr, one 16-byte load per lane straight from global memory into the operand
//     register (the four lanes of a voxel read 64 contiguous bytes per instruction; KS = Cin / 32 instructions cover 16 whole voxel rows):
//     no LDS, no conversion.
//   * A operand (weights): the plain (Cout, Cin) bf16 matrix; row m = 4 q + r of the wave's MFMA tile ct is output channel
//     c0 + 32 (ct / 2) + 8 q + 4 (ct % 2) + r: the D layout (lane: column lr, rows 4 lq .. 4 lq + 3) then leaves lane-quarter lq with the
//     OCTETS 32 h + 8 lq .. + 7 (h = 0 .. NCT / 2 - 1) of one voxel, so that store h of the four lanes of a voxel is 64 contiguous bytes.
//     A wave keeps its NCT x KS weight fragments (64 registers) for its whole run of voxels.
//   * Epilogue: + bias, round to bf16, 16-byte stores, sums / sums of squares of the ROUNDED values per octet (v_dot2c_f32_bf16, as
//     conv3d.hip's 64-channel tiles) kept per lane over the wave's run, folded over the 16 voxel lanes and the block's waves once, one
//     slot per block: (B, nslot, 2, Cout) f32 with the octet's sum on its first channel and zeros on the other seven -- what
//     gn_finalize_kernel already reads.
// Block = 4 waves = VB voxels of ONE sample x all Cout: G = Cout / (16 NCT) channel groups x 4 / G interleaved voxel sub-runs.  Voxel
// tiles are software-pipelined three deep in registers; rows past the sample's end are out-of-range buffer offsets (zeros in, stores
// dropped, masked out of the statistics): no exec-masked branch, so the waits stay counted.  The grouping of the partial sums is a
// function of the sample alone (blocks never straddle samples): a volume comes out bit-identical whatever batch it rides in.
// HBM-bound by construction: 16 MFMAs (512 matrix cycles) per 6 KB of traffic.
#include "common.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

namespace {

constexpr unsigned OOB = 0x80000000u;          // buffer offset beyond num_records: loads return zeros, stores are dropped
constexpr int PF = 3;                          // voxel tiles in flight per wave

struct C1Params {
    const bf16_t* x; const bf16_t* w; const float* bias; bf16_t* y; float* stats;
    int V, Cin, Cout, vb, stats_nblk, stats_slot0;
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t c1_rsrc(const void* base, unsigned bytes) {
    const uint64_t a = (uint64_t)base;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a), hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
    return __builtin_amdgcn_make_buffer_rsrc((void*)(((uint64_t)hi << 32) | lo), 0, (int)__builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}
__device__ __forceinline__ float c1_row16_sum(float v) {        // sum over the 16 lanes of a DPP row (equal lane >> 4), result in every lane
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4e, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xb1, 0xf, 0xf, false));
    return v;
}

template <int KS, int NCT, bool STATS>
__global__ __launch_bounds__(256, 2) void conv1x1_kernel(const C1Params p) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int CW = 16 * NCT;                       // channels of a wave
    constexpr int NOCT = NCT / 2;                      // 8-channel sums per lane
    __shared__ float sRed[4][2][4 * NOCT * 4];         // [wave][sum | squares][lq][octet] (padded to 4 octets x 4 quarters)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 15, lq = lane >> 4;
    const int G = p.Cout / CW, VS = 4 / G;             // channel groups, voxel sub-runs (host: G in {1, 2, 4})
    const int g = wave % G, vs = wave / G;
    const int b = blockIdx.y;
    const int v0 = blockIdx.x * p.vb;                  // first voxel of the block inside its sample
    const int nt_blk = (min(p.V - v0, p.vb) + 15) >> 4;
    const int ntile = nt_blk > vs ? (nt_blk - vs + VS - 1) / VS : 0;      // tiles vs, vs + VS, ... of the block

    const unsigned xbytes = (unsigned)((size_t)p.V * p.Cin * 2), ybytes = (unsigned)((size_t)p.V * p.Cout * 2);
    const __amdgpu_buffer_rsrc_t rx = c1_rsrc(p.x + (size_t)b * p.V * p.Cin, xbytes);
    const __amdgpu_buffer_rsrc_t ry = c1_rsrc(p.y + (size_t)b * p.V * p.Cout, ybytes);
    const __amdgpu_buffer_rsrc_t rw = c1_rsrc(p.w, (unsigned)((size_t)p.Cout * p.Cin * 2));

    // ---- the wave's weights: NCT x KS fragments, rows permuted so that a lane ends with 4 NCT consecutive channels
    bf16x8 wf[NCT][KS];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
        const int ch = g * CW + 32 * (ct >> 1) + 8 * (lr >> 2) + 4 * (ct & 1) + (lr & 3);
#pragma unroll
        for (int s = 0; s < KS; ++s)
            wf[ct][s] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rw, (unsigned)((ch * p.Cin + 32 * s + 8 * lq) * 2), 0, 0));
    }
    const int c0 = g * CW + 8 * lq;                    // the lane's first output channel: it owns octets c0 + 32 h
    f32x4 bv[NCT];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
        bv[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (p.bias) bv[ct] = *reinterpret_cast<const f32x4*>(p.bias + c0 + 32 * (ct >> 1) + 4 * (ct & 1));
    }

    // ---- activation fragments, PF tiles ahead
    bf16x8 xf[PF][KS];
    auto x_off = [&](int i) -> unsigned {              // tile i of this wave -> byte offset of the lane's first chunk, or OOB
        const int v = v0 + 16 * (vs + VS * i) + lr;
        return (i < ntile && v < p.V) ? (unsigned)((v * p.Cin + 8 * lq) * 2) : OOB;
    };
    auto x_load = [&](int i, int set) {
        const unsigned o = x_off(i);
#pragma unroll
        for (int s = 0; s < KS; ++s) xf[set][s] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rx, o + 64 * s, 0, 0));
    };
#pragma unroll
    for (int i = 0; i < PF; ++i) x_load(i, i);

    float gs[NOCT], gq[NOCT];
#pragma unroll
    for (int i = 0; i < NOCT; ++i) { gs[i] = 0.f; gq[i] = 0.f; }

    auto tile = [&](int i, int set) {
        f32x4 acc[NCT];
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) acc[ct] = bv[ct];
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) acc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ct][s], xf[set][s], acc[ct], 0, 0, 0);
        x_load(i + PF, set);                           // the set is free again: the tile PF ahead goes out under this one's epilogue
        const int v = v0 + 16 * (vs + VS * i) + lr;
        const bool live = i < ntile && v < p.V;          // (tiles past the wave's run would land in the next block's voxels)
        const unsigned yo = live ? (unsigned)((v * p.Cout + c0) * 2) : OOB;
#pragma unroll
        for (int h = 0; h < NOCT; ++h) {
            u32x4 pk;
            pk[0] = pack_bf16x2(acc[2 * h][0], acc[2 * h][1]); pk[1] = pack_bf16x2(acc[2 * h][2], acc[2 * h][3]);
            pk[2] = pack_bf16x2(acc[2 * h + 1][0], acc[2 * h + 1][1]); pk[3] = pack_bf16x2(acc[2 * h + 1][2], acc[2 * h + 1][3]);
            __builtin_amdgcn_raw_buffer_store_b128(pk, ry, yo + 64 * h, 0, 0);
            if constexpr (STATS) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const bf16x2 q = __builtin_bit_cast(bf16x2, live ? pk[j] : 0u);        // rows past the sample's end hold the bias: not part of the tensor
                    gs[h] = __builtin_amdgcn_fdot2_f32_bf16(q, __builtin_bit_cast(bf16x2, 0x3f803f80u), gs[h], false);
                    gq[h] = __builtin_amdgcn_fdot2_f32_bf16(q, q, gq[h], false);
                }
            }
        }
    };
    // (unrolled by PF so that the register sets are static; every wave runs the same trip count: ntile differs by at most one between the
    // block's sub-runs and tiles past the end are all-OOB)
    const int trips = (((nt_blk + VS - 1) / VS) + PF - 1) / PF;
    for (int it = 0; it < trips; ++it) {
#pragma unroll
        for (int k = 0; k < PF; ++k) tile(it * PF + k, k);
    }

    if constexpr (STATS) {
        // 16 voxel lanes -> one value per (lq, octet); 4 waves -> LDS; one plain store per (slot, channel, stat)
#pragma unroll
        for (int i = 0; i < NOCT; ++i) { gs[i] = c1_row16_sum(gs[i]); gq[i] = c1_row16_sum(gq[i]); }
        if (lr == 0) {
#pragma unroll
            for (int i = 0; i < NOCT; ++i) { sRed[wave][0][lq * NOCT + i] = gs[i]; sRed[wave][1][lq * NOCT + i] = gq[i]; }
        }
        __syncthreads();
        // thread -> (stat, channel): channel c belongs to group c / CW, store h = (c % CW) / 32, lane-quarter (c % 32) / 8
        for (int e = tid; e < 2 * p.Cout; e += 256) {
            const int st = e / p.Cout, c = e - st * p.Cout;
            float t = 0.f;
            if ((c & 7) == 0) {
                const int cg = c / CW, r = c - cg * CW, o = r >> 5, q = (r & 31) >> 3;
                for (int s2 = 0; s2 < VS; ++s2) t += sRed[s2 * G + cg][st][q * NOCT + o];            // fixed order: deterministic
            }
            p.stats[(((size_t)b * p.stats_nblk + p.stats_slot0 + blockIdx.x) * 2 + st) * p.Cout + c] = t;
        }
    }
#endif
}

// voxels per block, a function of the SAMPLE alone (the partial sums' grouping must not depend on the batch): about 64 blocks per sample, at
// least 16 tiles -- at B = 8 one full round of two blocks per CU at 48^3 (1728 voxels = 108 tiles per block), 432 blocks of 16 tiles at 24^3.
// Measured against 128 blocks per sample (two rounds): 85.7 vs 89.7 us and 37.9 vs 40.3 us (profiles/r06/lift_conv_ab.txt).
int c1_vb(int64_t V) { const int64_t v = (ceil_div(V, 64) + 15) / 16 * 16; return (int)(v < 256 ? 256 : v); }

template <int KS, int NCT>
int c1_launch(const C1Params& p, int64_t B, hipStream_t st) {
    const dim3 grid((unsigned)ceil_div(p.V, p.vb), (unsigned)B);
    if (p.stats) hipLaunchKernelGGL((conv1x1_kernel<KS, NCT, true>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((conv1x1_kernel<KS, NCT, false>), grid, dim3(256), 0, st, p);
    return gfe_launch_status();
}

}  // namespace

extern "C" {

int gfe_conv1x1_stat_slots(int64_t V) { return (int)ceil_div(V, c1_vb(V)); }

int gfe_conv1x1(const void* x, const void* w, const float* bias, void* y, int64_t B, int64_t V, int64_t Cin, int64_t Cout,
                float* stats_ws, int64_t stats_nblk, int64_t stats_slot0, void* stream) {
    GFE_REQUIRE(x && w && y, GFE_ERR_NULL);
    GFE_REQUIRE(B >= 1 && B <= 65535 && V >= 1 && (Cin == 64 || Cin == 128) && Cout >= 64 && Cout % 64 == 0, GFE_ERR_SHAPE);
    GFE_REQUIRE(V * Cout * 2 < (int64_t)1 << 31 && V * Cin * 2 < (int64_t)1 << 31 && Cout * Cin * 2 < (int64_t)1 << 31, GFE_ERR_SHAPE);   // 32-bit buffer offsets per sample
    GFE_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)w & 15) == 0 && ((uintptr_t)y & 15) == 0 && ((uintptr_t)bias & 15) == 0, GFE_ERR_SHAPE);   // 16-byte accesses
    C1Params p;
    p.x = (const bf16_t*)x; p.w = (const bf16_t*)w; p.bias = bias; p.y = (bf16_t*)y; p.stats = stats_ws;
    p.V = (int)V; p.Cin = (int)Cin; p.Cout = (int)Cout; p.vb = c1_vb(V);
    p.stats_nblk = (int)stats_nblk; p.stats_slot0 = (int)stats_slot0;
    if (stats_ws) GFE_REQUIRE(stats_slot0 >= 0 && stats_slot0 + gfe_conv1x1_stat_slots(V) <= stats_nblk && stats_nblk <= 0x7fffffff, GFE_ERR_SHAPE);
    hipStream_t st = (hipStream_t)stream;
    // a wave holds NCT x KS weight fragments = 64 registers: 128 channels of a 64-deep product, 64 of a 128-deep one
    const bool wide = Cin == 64 && Cout % 128 == 0;
    const int64_t G = Cout / (wide ? 128 : 64);
    GFE_REQUIRE(G == 1 || G == 2 || G == 4, GFE_ERR_SHAPE);
    if (Cin == 64) return wide ? c1_launch<2, 8>(p, B, st) : c1_launch<2, 4>(p, B, st);
    return c1_launch<4, 4>(p, B, st);
}

}  // extern "C"
