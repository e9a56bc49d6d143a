// Transposed 3-D convolution (k3, s2, p1, no bias) + nearest resize 2n-1 -> 2n + skip-sum + GroupNorm partials for gfx950, with the
// activation tile RESIDENT in LDS for all 8 output-parity classes.   Reference: pytorch3dunet/unet3d/buildingblocks.py:355-358, 523-537.
//
// Why a second kernel.  conv3d.hip runs the 8 parity classes as work items (tile, class) x 32-channel slabs; each such unit restages
// its 64 KB activation tile by LDS-DMA and then multiplies only 1-8 taps, so the launch is bound by the DMA rate of a CU (a few tens
// of GB/s on 64-B pieces; timing builds in DESIGN.md 4.1: 1.22 ms for 128 -> 64 @48^3 of which the MFMAs are 0.16).  Here a block
// stages ALL input channels of a smaller tile once -- 4 x 8 x 8 voxels, halo box 5 x 9 x 9, up to four 32-channel slabs = 128 KB --
// and walks the 8 classes over it: an eighth of the staging.  The next tile's slabs are fetched while the last (8-tap) class is
// still computing, slab by slab as that class finishes with them.
//
// Same conventions as conv3d.hip: weights are the MFMA A operand (v_mfma_f32_16x16x32_bf16, rows permuted at pack time so a lane owns 16
// consecutive channels of one voxel), activations the B operand, 64-B voxel rows with XOR-swizzled 16-B chunks applied on the DMA's source
// side, weight stages of 3 k-steps (a k-step = one (slab, tap) pair of the class, slab-major), double-buffered.
// Nine waves.  Waves 0-7 compute: wave = (d-plane of the tile, half of the plane's 4 voxel tiles) -> 2 x 4 accumulator tiles, k-steps
// software-pipelined over two fragment sets; waves 4-7 also issue the activation DMA.  Wave 8 only moves weight stages: the CU returns
// vector-memory data in order and vmcnt retires in order, so a wave that waits for a weight stage every stage cannot also have the
// class's skip-tensor rows in flight -- with the weights on their own wave, the compute waves request those rows at the START of a class
// and find them in registers in its epilogue (0.95 -> 0.92 ms; timing builds: the skip read costs 0.22 ms and the stores 0.07 of that,
// as memory throughput rather than latency -- each 16-B-per-lane access touches half a 128-B voxel row).
// Inputs are exactly those of gfe_convt3d_k3s2_fused (gfe_hip.h), which dispatches here when the tile fits.
#include "common.h"
#include "convt3d.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((address_space(3))) void* lds_void_t;

namespace {

__device__ __forceinline__ __amdgpu_buffer_rsrc_t uniform_rsrc(const void* base, unsigned bytes) {
    const uint64_t a = (uint64_t)base;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a), hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
    return __builtin_amdgcn_make_buffer_rsrc((void*)(((uint64_t)hi << 32) | lo), 0, (int)__builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}
__device__ __forceinline__ int rfl(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ float row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4e, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xb1, 0xf, 0xf, false));
    return v;
}

constexpr int TD = CONVT_TD, TH = 8, TW = 8;
constexpr int NWAVES = 8, W_PROD = 8, NTHREADS = 576, DMA_WAVES = 4;   // 8 compute waves + one weight-producer wave
constexpr int PH = 10, PW = 10, VSTRIDE = 64;           // LDS pitches (voxels) / bytes per voxel row of one slab: as in conv3d.hip
constexpr int SLAB_VOX = (TD + 1) * PH * PW;            // 500
constexpr int SLAB_PIECES = (SLAB_VOX + 15) / 16;       // 32 one-KiB DMA pieces
constexpr int SLAB_BYTES = SLAB_PIECES * 1024;
constexpr int A_PER_WAVE = SLAB_PIECES / DMA_WAVES;     // 8 pieces per activation wave and slab
constexpr int TPS = 3, W_PIECES = TPS * 64 / 16;        // weight stage: 3 k-steps x 64 rows = 12 pieces
constexpr unsigned OOB = 0x80000000u;
static_assert(SLAB_PIECES % DMA_WAVES == 0, "piece split");

struct TilePos { int b, td, th, tw; };

__global__ __launch_bounds__(NTHREADS) void convt_resident_kernel(const ConvTParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint8_t* sA = smem;                                           // nslab x SLAB_BYTES
    uint8_t* sW = smem + CONVT_MAX_SLABS * SLAB_BYTES;            // 2 x 12 KiB
    float* sRed = reinterpret_cast<float*>(sW + 2 * W_PIECES * 1024);   // [NWAVES][2][64]

    const int tid = threadIdx.x, lane = tid & 63, wave = rfl(tid >> 6);
    const int lq = lane >> 4, lr = lane & 15;
    GFE_FUZZ_INIT();
    const int ntiles = p.B * p.ntd * p.nth * p.ntw;
    // XCD-aware, interleaved tile walk (see conv3d.hip)
    const int nb = gridDim.x, xq = nb >> 3, xr = nb & 7, xcd = blockIdx.x & 7, xi = blockIdx.x >> 3;
    const int vb = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + xi;
    const int nbx = xq + (xcd < xr ? 1 : 0);
    const int xcd_begin = min(ntiles, (vb - xi) * p.tiles_per_block);
    int tile_begin = xcd_begin + xi;
    const int tile_end = min(ntiles, (vb - xi + nbx) * p.tiles_per_block);
    // dynamic tile scheduling (conv3d.hip): the XCD's blocks draw its tiles in order from a ticket counter; the ticket of tile k+1 is drawn at
    // the top of tile k by thread 0 and read, behind the classes' barriers, when the last class starts (the first use of the next tile)
    __shared__ int s_ticket;
    const bool dyn = p.sched != nullptr;
    auto finish = [&]() {
        if (dyn && tid == 0) {
            __threadfence();
            if (atomicAdd(p.sched + 8, 1) == (int)gridDim.x - 1) {
#pragma unroll
                for (int i = 0; i < 9; ++i) p.sched[i] = 0;
            }
        }
    };
    if (dyn) {
        GFE_FUZZ();
        if (tid == 0) s_ticket = atomicAdd(p.sched + xcd, 1);
        __syncthreads();
        GFE_FUZZ();
        tile_begin = xcd_begin + rfl(s_ticket);
        __syncthreads();
    }
    if (tile_begin >= tile_end) { finish(); return; }
    const int my_tiles = dyn ? 0x7fffffff : (tile_end - tile_begin + nbx - 1) / nbx;

    auto decode = [&](int t) {
        TilePos q;
        q.tw = t % p.ntw; t /= p.ntw;
        q.th = t % p.nth; t /= p.nth;
        q.td = t % p.ntd; q.b = t / p.ntd;
        return q;
    };

    // ---- compute mapping: plane pl of the tile, voxel tiles xt = 2*hx + j (rows 2xt, 2xt+1 of the plane)
    const bool w_prod = wave == W_PROD;                            // wave 8: weight DMA only (its vmcnt queue holds nothing else)
    const int pl = (wave >> 1) & 3, hx = wave & 1;
    int abase[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int lh = 2 * (2 * hx + j) + (lr >> 3), lw = lr & 7;
        abase[j] = ((pl * PH + lh) * PW + lw) * VSTRIDE + ((lq ^ ((lh & 1) << 1)) * 16);
    }
    const int wbase = lr * VSTRIDE + ((lq ^ ((lr >> 1) & 3)) * 16);

    // ---- DMA constants
    const int dw = wave & (DMA_WAVES - 1);
    const bool a_wave = wave >= DMA_WAVES && !w_prod;              // waves 4-7: activation DMA besides their compute share
    int acoord[A_PER_WAVE];          // ld | lh << 4 | lw << 8 | (chunk*16) << 12 | valid << 20
#pragma unroll
    for (int j = 0; j < A_PER_WAVE; ++j) {
        const int k = dw + DMA_WAVES * j, v = 16 * k + (lane >> 2);
        const int ld = v / (PH * PW), rem = v - ld * (PH * PW), lh = rem / PW, lw = rem - lh * PW;
        const int c = (lane & 3) ^ ((lh & 1) << 1);
        const bool valid = ld <= TD && lh <= TH && lw <= TW;
        acoord[j] = ld | (lh << 4) | (lw << 8) | ((c * 16) << 12) | ((valid ? 1 : 0) << 20);
    }
    // weight stage = 3 k-steps x 64 rows = 12 KiB contiguous in the packed buffer (CoutPad = 64): piece k = rows 16k + lane/4
    const unsigned wvoff = (unsigned)((lane >> 2) * 64 + (((lane & 3) ^ ((lane >> 3) & 3)) * 16));     // (row >> 1) & 3 == (lane >> 3) & 3
    const size_t sample_elems = (size_t)p.D * p.H * p.W * p.Cin;
    const unsigned sample_bytes = (unsigned)(sample_elems * 2);

    auto a_dma = [&](const TilePos& q, int slab) {                // one slab of a tile's halo box, by the four activation waves
        const __amdgpu_buffer_rsrc_t rs = uniform_rsrc(p.x + (size_t)rfl(q.b) * sample_elems, sample_bytes);
        const int d0 = rfl(q.td) * TD, h0 = rfl(q.th) * TH, w0 = rfl(q.tw) * TW;
        slab = rfl(slab);
#pragma unroll
        for (int j = 0; j < A_PER_WAVE; ++j) {
            const int ac = acoord[j];
            const int gd = d0 + (ac & 15), gh = h0 + ((ac >> 4) & 15), gw = w0 + ((ac >> 8) & 15);
            const bool ok = ((ac >> 20) & 1) && gd < p.D && gh < p.H && gw < p.W;
            const unsigned voff = ok ? (unsigned)((((gd * p.H + gh) * p.W + gw) * p.Cin + slab * 32) * 2 + ((ac >> 12) & 0xff)) : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t)(sA + slab * SLAB_BYTES + (dw + DMA_WAVES * j) * 1024), 16, voff, 0, 0, 0);
        }
    };
    auto w_dma = [&](int cls, int stage, int buf) {               // k-steps 3*stage .. 3*stage+2 of a class, by the producer wave
        cls = rfl(cls);
        const __amdgpu_buffer_rsrc_t rs = uniform_rsrc(p.w + p.c_woff[cls], p.w_bytes - (unsigned)(p.c_woff[cls] * 2));
        const int soff = rfl(stage) * (W_PIECES * 1024);          // past the class: next class's rows or zeros, never multiplied
        const int lds_off = rfl(buf) * (W_PIECES * 1024);
#pragma unroll
        for (int k = 0; k < W_PIECES; ++k)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t)(sW + lds_off + k * 1024), 16, wvoff, soff + k * 1024, 0, 0);
    };

    f32x4 acc[2][4];
    uint4 rpf[2][2];                                   // skip tensor of the class being computed (primary destination of the lane's 2 voxels)
    float gs[2] = {0.f, 0.f}, gq[2] = {0.f, 0.f};     // GroupNorm partials (8-channel sums, see conv3d.hip OCT) of what this block stores

    TilePos cur = decode(tile_begin);
    if (a_wave) { for (int sl = 0; sl < p.nslab; ++sl) a_dma(cur, sl); }
    if (w_prod) w_dma(CONVT_NCLS - 1, 0, 0);
    GFE_FUZZ();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    int gstage = 0;

    for (int ti = 0; ti < my_tiles; ++ti) {
        bool next_tile = !dyn && ti + 1 < my_tiles;
        TilePos nxt = next_tile ? decode(tile_begin + (ti + 1) * nbx) : cur;
        GFE_FUZZ();
        if (dyn && tid == 0) s_ticket = atomicAdd(p.sched + xcd, 1);
        int pf_slab = 0;                                          // slabs of the next tile already requested
        // classes in reverse buffer order: the host puts the heavy classes first, the 8-tap class must run LAST here
        for (int ci = CONVT_NCLS - 1; ci >= 0; --ci) {
            const int ntaps = p.c_ntaps[ci], lg = p.c_lg[ci], tap0 = p.c_tap0[ci];
            const int nk = ntaps * p.nslab, nst = (nk + TPS - 1) / TPS;
            GFE_FUZZ();
            if (dyn && ci == 0) {                                  // seven classes' worth of barriers after the draw
                const int nt_ = xcd_begin + rfl(*(volatile int*)&s_ticket);
                next_tile = nt_ < tile_end;
                if (next_tile) nxt = decode(nt_);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) acc[j][ct] = f32x4{0.f, 0.f, 0.f, 0.f};

            for (int st = 0; st < nst; ++st, ++gstage) {
                // st == 0: everything this wave had issued was drained at the end of the previous class (or before the loop); the
                // first class of a tile additionally needs the tail of the tile prefetch the activation waves issued at the tile end
                GFE_FUZZ();
                if (w_prod && st > 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the producer's queue holds weight stages only
                if (a_wave && st == 0 && ci == CONVT_NCLS - 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_waitcnt(0xc07f);
                GFE_FUZZ();
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                GFE_FUZZ();
                if (!w_prod && st == 0 && p.res) {
                    // skip-tensor prefetch: the compute waves wait on vmcnt only at the end of a class, so the rows this class's epilogue
                    // adds are requested now (primary destination of the lane's two voxels, exactly as the epilogue computes it) and
                    // arrive under the class's stages instead of as an exposed HBM round trip in the epilogue
                    const int opar_ = p.c_op[ci], cd_ = cur.td * TD + pl;
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const int ch_ = cur.th * TH + 2 * (2 * hx + j) + (lr >> 3), cw_ = cur.tw * TW + (lr & 7);
                        const int dd_ = 2 * cd_ + (opar_ & 1) + p.oshift, dh_ = 2 * ch_ + ((opar_ >> 1) & 1) + p.oshift, dw_ = 2 * cw_ + ((opar_ >> 2) & 1) + p.oshift;
                        if (cd_ < p.D && ch_ < p.H && cw_ < p.W && dd_ < p.OD && dh_ < p.OH && dw_ < p.OW) {
                            const uint4* rp = reinterpret_cast<const uint4*>(p.res + ((((size_t)cur.b * p.OD + dd_) * p.OH + dh_) * p.OW + dw_) * p.Cout + lq * 16);
                            rpf[j][0] = rp[0]; rpf[j][1] = rp[1];
                        }
                    }
                }
#if defined(CONVT_EXP_NOW)     // timing experiment only: weights are never restaged
                if (false) {
#else
                if (w_prod) {
#endif
                    if (st + 1 < nst) w_dma(ci, st + 1, (gstage + 1) & 1);
                    else if (ci > 0) w_dma(ci - 1, 0, (gstage + 1) & 1);
                    else if (next_tile) w_dma(CONVT_NCLS - 1, 0, (gstage + 1) & 1);
                } else if (a_wave && ci == 0 && next_tile) {
                    // last class: k-steps below 3*st are done by every wave (barrier above) -> the slabs they covered are free
#if !defined(CONVT_EXP_NOA)    // timing experiment only: the next tile is never fetched
                    while (pf_slab < p.nslab && ntaps * (pf_slab + 1) <= TPS * st) { a_dma(nxt, pf_slab); ++pf_slab; }
#endif
                }
                if (w_prod) continue;
                const uint8_t* wb = sW + (gstage & 1) * (W_PIECES * 1024) + wbase;
                // software pipeline over the stage's k-steps: the 6 fragment reads of k-step jk+1 are issued before the 8 MFMAs of
                // k-step jk (two register sets), so only the first k-step of a stage exposes the LDS latency
                bf16x8 xf[2][2], wf[2][4];
                auto frag_load = [&](int jk, int set) {
                    const int ks = st * TPS + jk;
                    const int sl = ks >> lg, tap = tap0 + (ks & (ntaps - 1));
                    const int toff = p.toff[tap], txor = p.txor[tap];
                    const uint8_t* aS = sA + sl * SLAB_BYTES;
#pragma unroll
                    for (int j = 0; j < 2; ++j) xf[set][j] = *reinterpret_cast<const bf16x8*>(aS + ((abase[j] + toff) ^ txor));
#pragma unroll
                    for (int ct = 0; ct < 4; ++ct) wf[set][ct] = *reinterpret_cast<const bf16x8*>(wb + (jk * 64 + ct * 16) * VSTRIDE);
                };
                frag_load(0, 0);                                             // (stage st exists, so its first k-step does)
#pragma unroll
                for (int jk = 0; jk < TPS; ++jk) {
                    const int set = jk & 1;
                    if (st * TPS + jk < nk) {                                    // block-uniform
                        if (jk + 1 < TPS && st * TPS + jk + 1 < nk) frag_load(jk + 1, set ^ 1);
#pragma unroll
                        for (int ct = 0; ct < 4; ++ct)
#pragma unroll
                            for (int j = 0; j < 2; ++j) acc[j][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[set][ct], xf[set][j], acc[j][ct], 0, 0, 0);
                    }
                }
            }
            GFE_FUZZ();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // only the epilogue's stores cross the class boundary
            if (ci == 0 && next_tile) {
                // the slabs the last stages were still reading: request them now, they land under this epilogue
                __builtin_amdgcn_s_waitcnt(0xc07f);
                GFE_FUZZ();
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                GFE_FUZZ();
#if !defined(CONVT_EXP_NOA)
                if (a_wave) { while (pf_slab < p.nslab) { a_dma(nxt, pf_slab); ++pf_slab; } }
#endif
            }

            // ---- epilogue of class ci: resize placement (dst = raw + oshift, dst 0 duplicates raw 0), skip-sum, bf16 store, partials
            const int opar = p.c_op[ci];
            const int cd = cur.td * TD + pl, c0 = lq * 16;
#if defined(CONVT_EXP_NOEPI)   // timing experiment only: only the block's very last class is stored
            if (!w_prod && cd < p.D && c0 < p.Cout && !next_tile && ci == 0) {
#else
            if (!w_prod && cd < p.D && c0 < p.Cout) {
#endif
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int xt = 2 * hx + j;
                    const int ch_ = cur.th * TH + 2 * xt + (lr >> 3), cw_ = cur.tw * TW + (lr & 7);
                    if (ch_ >= p.H || cw_ >= p.W) continue;
                    const int od = 2 * cd + (opar & 1), oh = 2 * ch_ + ((opar >> 1) & 1), ow = 2 * cw_ + ((opar >> 2) & 1);
                    if (od > 2 * p.D - 2 || oh > 2 * p.H - 2 || ow > 2 * p.W - 2) continue;     // raw output has 2n-1 positions per axis
                    const int nd = (p.oshift && od == 0) ? 2 : 1, nh = (p.oshift && oh == 0) ? 2 : 1, nw = (p.oshift && ow == 0) ? 2 : 1;
                    for (int zd = 0; zd < nd; ++zd)
                        for (int zh = 0; zh < nh; ++zh)
                            for (int zw = 0; zw < nw; ++zw) {
                                const int dd_ = zd ? 0 : od + p.oshift, dh_ = zh ? 0 : oh + p.oshift, dw_ = zw ? 0 : ow + p.oshift;
                                if (dd_ >= p.OD || dh_ >= p.OH || dw_ >= p.OW) continue;
                                const size_t o = ((((size_t)cur.b * p.OD + dd_) * p.OH + dh_) * p.OW + dw_) * p.Cout + c0;
                                uint4 rv[2] = {make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0)};
#if !defined(CONVT_EXP_NORES)  // timing experiment only: the skip tensor is not read
                                if (p.res) {
                                    if (zd == 0 && zh == 0 && zw == 0) { rv[0] = rpf[j][0]; rv[1] = rpf[j][1]; }          // prefetched at the class start
                                    else { const uint4* rp = reinterpret_cast<const uint4*>(p.res + o); rv[0] = rp[0]; rv[1] = rp[1]; }
                                }
#endif
#pragma unroll
                                for (int h = 0; h < 2; ++h) {
                                    const uint32_t rw[4] = {rv[h].x, rv[h].y, rv[h].z, rv[h].w};
                                    uint32_t pk[4];
#pragma unroll
                                    for (int jj = 0; jj < 2; ++jj) {
                                        const f32x4 a = acc[j][2 * h + jj];
                                        pk[2 * jj] = pack_bf16x2(a[0] + bf16lo_to_f32(rw[2 * jj]), a[1] + bf16hi_to_f32(rw[2 * jj]));
                                        pk[2 * jj + 1] = pack_bf16x2(a[2] + bf16lo_to_f32(rw[2 * jj + 1]), a[3] + bf16hi_to_f32(rw[2 * jj + 1]));
                                    }
                                    if (p.stats) {
#pragma unroll
                                        for (int jj = 0; jj < 4; ++jj) {
                                            const bf16x2 v = __builtin_bit_cast(bf16x2, pk[jj]);
                                            gs[h] = __builtin_amdgcn_fdot2_f32_bf16(v, __builtin_bit_cast(bf16x2, 0x3f803f80u), gs[h], false);
                                            gq[h] = __builtin_amdgcn_fdot2_f32_bf16(v, v, gq[h], false);
                                        }
                                    }
#if defined(CONVT_EXP_NOSTORE)  // timing experiment only: results are computed but only the block's last class is stored
                                    if (!next_tile && ci == 0)
#endif
                                    reinterpret_cast<uint4*>(p.y + o)[h] = make_uint4(pk[0], pk[1], pk[2], pk[3]);
                                }
                            }
                }
            }
        }
        // ---- GroupNorm partials: one flush and one slot per tile of the sample (block-uniform), so that the grouping of the partial sums
        // does not depend on the batch the sample rides in (see conv3d.hip)
        if (p.stats) {
#pragma unroll
            for (int i = 0; i < 2; ++i) { gs[i] = row16_sum(gs[i]); gq[i] = row16_sum(gq[i]); }
            if (lr == 0 && !w_prod) {
                float4* r0 = reinterpret_cast<float4*>(sRed + (wave * 2) * 64 + lq * 16);
                float4* r1 = reinterpret_cast<float4*>(sRed + (wave * 2 + 1) * 64 + lq * 16);
#pragma unroll
                for (int i = 0; i < 4; ++i) {        // the 8-channel sum goes to the first channel of the octet, zeros to the other seven
                    r0[i] = make_float4((i & 1) ? 0.f : gs[i >> 1], 0.f, 0.f, 0.f);
                    r1[i] = make_float4((i & 1) ? 0.f : gq[i >> 1], 0.f, 0.f, 0.f);
                }
            }
            gs[0] = gs[1] = gq[0] = gq[1] = 0.f;
            GFE_FUZZ();
            __syncthreads();
            GFE_FUZZ();
            if (tid < 128) {
                float t = 0.f;
#pragma unroll
                for (int wv_ = 0; wv_ < NWAVES; ++wv_) t += sRed[wv_ * 128 + tid];
                const int st = tid >> 6, c = tid & 63;
                const int tix = (cur.td * p.nth + cur.th) * p.ntw + cur.tw;
                if (c < p.Cout) p.stats[(((size_t)cur.b * p.stats_nblk + tix) * 2 + st) * p.Cout + c] = t;
            }
            __syncthreads();
        }
        if (!next_tile) break;
        cur = nxt;
    }
    finish();
#endif
}

}  // namespace

bool convt_resident_fits(int64_t Cin, int64_t Cout) { return Cin % 32 == 0 && Cin / 32 <= CONVT_MAX_SLABS && Cout == 64; }

int convt_resident_tiles(int64_t D, int64_t H, int64_t W) { return (int)(ceil_div(D, TD) * ceil_div(H, TH) * ceil_div(W, TW)); }

int convt_resident_grid(int64_t B, int64_t D, int64_t H, int64_t W, int* tiles_per_block) {
    const int64_t tiles = B * ceil_div(D, TD) * ceil_div(H, TH) * ceil_div(W, TW);
    const int64_t tpb = ceil_div(tiles, 256);
    if (tiles_per_block) *tiles_per_block = (int)tpb;
    return (int)ceil_div(tiles, tpb);
}

int convt_resident_launch(ConvTParams& p, hipStream_t st) {
    p.ntd = (int)ceil_div(p.D, TD); p.nth = (int)ceil_div(p.H, TH); p.ntw = (int)ceil_div(p.W, TW);
    int grid = convt_resident_grid(p.B, p.D, p.H, p.W, &p.tiles_per_block);
    p.sched = conv_sched_slot(st);
    if (p.sched && conv_reserved_cus() > 0) {                     // leave CUs to another stream's kernels (conv3d.hip)
        const int64_t tiles = (int64_t)p.B * p.ntd * p.nth * p.ntw;
        p.tiles_per_block = (int)ceil_div(tiles, (int64_t)(256 - conv_reserved_cus()));
        grid = (int)ceil_div(tiles, (int64_t)p.tiles_per_block);
    }
    const size_t lds = (size_t)CONVT_MAX_SLABS * SLAB_BYTES + 2 * (size_t)W_PIECES * 1024 + (size_t)NWAVES * 2 * 64 * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) { (void)hipFuncSetAttribute((const void*)convt_resident_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr_set = true; }
    hipLaunchKernelGGL(convt_resident_kernel, dim3(grid), dim3(NTHREADS), lds, st, p);
    return hipGetLastError() == hipSuccess ? GFE_OK : GFE_ERR_HIP;
}
