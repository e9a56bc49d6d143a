// Internal interface between gfe_convt3d_k3s2_fused (conv3d.hip) and the LDS-resident transposed-conv kernel (convt3d.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define CONVT_TD 4            // tile depth: 4 x 8 x 8 input voxels, halo box 5 x 9 x 9
#define CONVT_MAX_SLABS 4     // 32-channel slabs resident at once (Cin <= 128): 4 x 32 KiB of the 160 KiB LDS
#define CONVT_NCLS 8

struct ConvTParams {
    const uint16_t* x; const uint16_t* w; const uint16_t* res; uint16_t* y; float* stats;
    int B, D, H, W, Cin, Cout, CoutPad, OD, OH, OW, nslab, oshift, stats_nblk;
    int ntd, nth, ntw, tiles_per_block;
    int c_ntaps[CONVT_NCLS], c_lg[CONVT_NCLS], c_tap0[CONVT_NCLS], c_op[CONVT_NCLS];
    long long c_woff[CONVT_NCLS]; unsigned w_bytes;
    int toff[27], txor[27];   // LDS byte offset / swizzle term of each tap inside a slab image (same as conv3d.hip's halo tile)
    int* sched;               // dynamic tile scheduling (conv3d.hip): [0..7] per-XCD ticket counters, [8] finished blocks; NULL = static shares
};

bool convt_resident_fits(int64_t Cin, int64_t Cout);
int convt_resident_grid(int64_t B, int64_t D, int64_t H, int64_t W, int* tiles_per_block);
int convt_resident_tiles(int64_t D, int64_t H, int64_t W);      // tiles (= GroupNorm partial slots) per sample
int convt_resident_launch(ConvTParams& p, hipStream_t st);
int conv_reserved_cus();                                         // conv3d.hip: CUs the persistent conv kernels leave free (gfe_conv_reserve_cus)
int* conv_sched_slot(hipStream_t st);                            // conv3d.hip: the next slot of the ticket-counter ring (NULL: off / not allocatable now)
