/* gfe_hip.h -- C-ABI of libgfe_hip.so: the MI355X (gfx950) kernels behind GFE-Mamba's classification hot path.
 *
 * The reference (Tinysqua/GFE-Mamba) is pure PyTorch; its only native plug-in slot on this path is
 * `MambaBlock.selective_scan_cuda` (cross_atten/mamba.py:180-186, call at :251).  Everything else here
 * replaces a stock torch op sequence of the reference; each entry point cites the reference lines it
 * stands in for.  The Python host code (gfe-mamba_amd/) binds these with ctypes; INTEGRATION.md shows
 * the stub a maintainer of the reference would add.
 *
 * Conventions (all entry points):
 *   - plain pointers + int64 sizes, no torch / C++ types; device pointers unless stated otherwise
 *   - return GFE_OK (0) or a negative GFE_ERR_*; never throw, never allocate, never synchronise
 *   - work is enqueued on `stream` (a hipStream_t passed as void*); buffers are owned by the caller
 *   - `dtype` is the storage type of activations (GFE_F32 / GFE_BF16); accumulation is always f32
 *   - workspaces are caller-allocated; "zeroed" workspaces must be zero on entry
 */
#ifndef GFE_HIP_H
#define GFE_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GFE_OK          0
#define GFE_ERR_NULL   -1   /* required pointer is NULL */
#define GFE_ERR_SHAPE  -2   /* unsupported / inconsistent shape */
#define GFE_ERR_DTYPE  -3   /* unsupported dtype */
#define GFE_ERR_HIP    -4   /* hipGetLastError() != hipSuccess after launch */

#define GFE_ABI_VERSION 49

#define GFE_F32  0
#define GFE_BF16 1

/* library / device */
int gfe_abi_version(void);                 /* bumps when a signature changes */
const char* gfe_build_arch(void);          /* "gfx950" */

/* ---------------------------------------------------------------------------------------------
 * Group A -- selective scan
 * ------------------------------------------------------------------------------------------- */

/* Chunk plan shared by forward and backward (chunk_req <= 0: automatic).  Host-only, no GPU work. */
int gfe_sscan_plan(int64_t B, int64_t L, int64_t ED, int64_t N, int chunk_req, int backward,
                   int* T_out, int* nchunks_out);

/* Fused selective scan, token-major layout.
 * Replaces cross_atten/mamba.py:265-286 (selective_scan) + :255-256 (softplus(delta+bias)) + :220-222 (y*silu(z)),
 * i.e. the contract of the reference's plug-in `selective_scan_fn` (mamba.py:251) without its transposes.
 *   u, delta, z, y : (B, L, ED) dtype        Bm, Cm : (B, L, N) f32 (always: rows are staged per wave in LDS)
 *   A : (ED, N) f32    D, delta_bias : (ED) f32 or NULL    z : NULL => no gate
 *   hstate (B, nchunks, N, ED) f32, sdelta (B, nchunks, ED) f32: workspaces, required when nchunks > 1;
 *   after the call hstate holds the state at the start of every chunk (kept for the backward).
 *   N in {4, 8, 16}, ED % 64 == 0.  T = chunk length from gfe_sscan_plan. */
int gfe_selective_scan_fwd(const void* u, const void* delta, const float* A, const float* Bm, const float* Cm,
                           const float* D, const void* z, const float* delta_bias, void* y,
                           float* hstate, float* sdelta,
                           int64_t B, int64_t L, int64_t ED, int64_t N, int T, int delta_softplus,
                           int dtype, void* stream);

/* Adjoint of the above (what autograd derives for mamba.py:265-286 through PScan.backward, pscan.py:188-224).
 *   du, ddelta, dz : (B, L, ED) dtype (ddelta is w.r.t. the raw delta when delta_softplus)
 *   dA_ws (N, ED) f32 [transposed], dB_ws, dC_ws (B, L, N) f32, dD_ws, dbias_ws (ED) f32: zeroed, accumulated atomically
 *   hstate, sdelta: as left by the forward run with the same T; qstate: workspace like hstate.  ED % 64 == 0. */
int gfe_selective_scan_bwd(const void* u, const void* delta, const float* A, const float* Bm, const float* Cm,
                           const float* D, const void* z, const float* delta_bias, const void* dy,
                           void* du, void* ddelta, void* dz,
                           float* dA_ws, float* dB_ws, float* dC_ws, float* dD_ws, float* dbias_ws,
                           const float* hstate, float* qstate, const float* sdelta,
                           int64_t B, int64_t L, int64_t ED, int64_t N, int T, int delta_softplus,
                           int dtype, void* stream);

/* The same operator for N = 16 in the single-pass formulation (csrc/sscan2.hip: a lane owns one channel x one PAIR of states, so
 * B*ED/8 waves fill the chip without cutting L; L is chunked -- two passes + carry -- only for small B).  Same reference lines as
 * gfe_selective_scan_fwd/_bwd (mamba.py:243-286, pscan.py:151-224); N = 16, ED % 32 == 0, u/delta/z/y 16-byte aligned.
 *   Bm, Cm: (B, L, 16) rows in `bc_dtype` (GFE_F32 or GFE_BF16), 16-byte aligned.
 *   gfe_sscan2_plan: T = chunk length (a multiple of 32 when nchunks > 1; chunk_req <= 0: automatic), host only.
 *   hstate (B, nchunks, ED, 16), sdelta (B, nchunks, ED) f32: workspaces, required when nchunks > 1.
 *   ckpt (B, ceil(L/32), ED, 16) f32 or NULL: the forward leaves the state at the start of every 32-step segment there; the backward
 *   recomputes one segment at a time from it with the segment's states in registers.
 *   ld_z, ld_bc (forward and backward), ld_dbc (backward): row strides in elements of z / dz, of the Bm / Cm rows and of the dB_ws / dC_ws rows
 *   (0 = contiguous: ED, 16, 16) -- the fused Mamba block (gfe_hip/mamba_block.py) hands the kernel z, B and C IN PLACE inside the
 *   in_proj / x_proj outputs (mamba.py:204-207, 235-236) and lets it write dz / dB / dC straight into the gradients of those outputs.
 *   a_is_log != 0: `A` holds the parameter A_log; the kernels use A = -exp(A_log) (mamba.py:232) and dA_ws receives d/dA_log.
 *   yscan (B, L, ED) dtype or NULL: the forward also leaves its output BEFORE the gate (hs.C + D*u) there; the backward needs it when z
 *   is given (dz = dy * silu'(z) * yscan) -- one more output row per step instead of a second sum over states per step in the backward. */
int gfe_sscan2_plan(int64_t B, int64_t L, int64_t ED, int chunk_req, int* T_out, int* nchunks_out);
int gfe_sscan2_fwd(const void* u, const void* delta, const float* A, const void* Bm, const void* Cm,
                   const float* D, const void* z, const float* delta_bias, void* y, void* yscan,
                   float* hstate, float* sdelta, float* ckpt,
                   int64_t B, int64_t L, int64_t ED, int T, int delta_softplus, int dtype, int bc_dtype,
                   int64_t ld_z, int64_t ld_bc, int a_is_log, void* stream);
/*   dA_ws (ED, 16), dB_ws / dC_ws (B, L, 16), dD_ws / dbias_ws (ED) f32; qstate: workspace like hstate (nchunks > 1); ckpt, sdelta: as
 *   left by gfe_sscan2_fwd with the same T.
 *   part_vec (B * nchunks * ED * 18 floats) and part_bc (B * (ED/32) * L * 32 floats), both or neither: with them NOTHING is added
 *   atomically -- every block stores its contributions there and a second launch sums them in a fixed order (run-to-run identical
 *   gradients): dA_ws / dD_ws / dbias_ws are ADDED to (they may be gradient buffers), the dB_ws / dC_ws rows are overwritten (no zero fill
 *   needed).  Both NULL: f32 atomics, and all five buffers must be zero on entry (large B x L, where part_bc would be 128 B per
 *   (channel group, b, t)). */
int gfe_sscan2_bwd(const void* u, const void* delta, const float* A, const void* Bm, const void* Cm,
                   const float* D, const void* z, const float* delta_bias, const void* dy, const void* yscan,
                   void* du, void* ddelta, void* dz,
                   float* dA_ws, float* dB_ws, float* dC_ws, float* dD_ws, float* dbias_ws,
                   const float* ckpt, float* qstate, const float* sdelta, float* part_vec, float* part_bc,
                   int64_t B, int64_t L, int64_t ED, int T, int delta_softplus, int dtype, int bc_dtype,
                   int64_t ld_z, int64_t ld_bc, int64_t ld_dbc, int a_is_log, void* stream);

/* Materialised scan H[t] = A[t]*H[t-1] + X[t] (H[-1] = 0) over dim 1 of (B, L, DN) tensors, DN = D*N flattened.
 * Drop-in for cross_atten/pscan.py:226 `pscan(A, X)` (PScan.forward, pscan.py:151-186); inputs are not modified.
 *   ws: (B, nchunks, 2, DN) f32 workspace when nchunks > 1 (nchunks = ceil(L/T)). */
int gfe_pscan_fwd(const void* A, const void* X, void* H, float* ws,
                  int64_t B, int64_t L, int64_t DN, int T, int dtype, void* stream);

/* PScan.backward (pscan.py:188-224): gX = reverse scan of gH with A shifted left by one; gA[t] = H[t-1]*gX[t], gA[0] = 0. */
int gfe_pscan_bwd(const void* A, const void* H, const void* gH, void* gA, void* gX, float* ws,
                  int64_t B, int64_t L, int64_t DN, int T, int dtype, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Group C -- frozen generator (pytorch3dunet/unet3d), channels-last (NDHWC) bf16 activations
 * ------------------------------------------------------------------------------------------- */

/* Padded output-channel count of the packed weight layout used by gfe_conv3d_igemm (16, 32, or a multiple of 64). Host-only. */
int gfe_conv3d_cout_pad(int64_t Cout);

/* Implicit-GEMM convolution described by a tap list: y[v] = sum_t W[t] . x[v + off(t)]  (+ bias_tab | bias, + res, ReLU).
 * Replaces, depending on the tap list / flags:
 *   SingleConv 'gcr'/'gc' = GroupNorm -> Conv3d k3 p1 no-bias [-> ReLU]  (buildingblocks.py:38-67): 27 taps, per-sample
 *     weights + bias_tab produced by gfe_conv3d_fold_groupnorm
 *   ResNetBlock.conv1 = Conv3d k1 + bias (buildingblocks.py:191-196): 1 tap, bias
 *   ResNetBlock tail `out += residual; ReLU` (buildingblocks.py:226-227): res + relu
 *   TransposeConvUpsampling = ConvTranspose3d k3 s2 p1 no-bias -> F.interpolate(nearest, 2n-1 -> 2n) -> `encoder_features + x`
 *   (buildingblocks.py:396-400, 523-537): 8 calls, one per output parity class (ostride 2, op_* parity, oshift 1, res = skip)
 *   x: (B, D, H, W, Cin) bf16.  y, res: (B, OD, OH, OW, Cout) bf16.  Cin % 8 == 0, Cout % 8 == 0.
 *   w_packed: [ceil(Cin/32)][ntaps][CoutPad][32] bf16, zero padded (tap order = tap_offsets order); inside each group of
 *     NT = min(CoutPad,64)/16 MFMA tiles, packed row ct*16 + 4*q + r holds output channel q*4*NT + 4*ct + r
 *     (gfe_hip/nn_ops.py:_row_perm).  w_batch_stride: elements between per-sample weight sets, 0 = one set for all samples.
 *   bias: (Cout) f32 or NULL.  bias_tab: (B, 64, CoutPad) f32 or NULL, indexed by the voxel's boundary class
 *     (d==0) | (d==D-1)<<1 | (h==0)<<2 | (h==H-1)<<3 | (w==0)<<4 | (w==W-1)<<5.
 *   tap_offsets: HOST pointer, ntaps x 3 int8 (dd, dh, dw) each in [-1, 1].
 *   ostride 1: OD,OH,OW == D,H,W.  ostride 2: output index = 2*i + op_*, shifted by oshift (0/1) with index 0 duplicated
 *   (nearest resize 2n-1 -> 2n: dst j <- src max(j-1, 0)); OD == 2*D - 1 + oshift.
 *   stats_ws (or NULL): GroupNorm partials of the tensor this call STORES, for the GroupNorm of the next SingleConv -- saves
 *   re-reading y.  Layout (B, stats_nblk, 2, Cout) f32, ZEROED by the caller: [.,slot,0,c] = sum, [.,slot,1,c] = sum of squares of
 *   the rounded bf16 outputs.  A persistent block keeps per-lane sums over its run of 8x8x8 tiles and stores them (plain stores,
 *   fixed order: deterministic) when its run leaves the sample / channel group or ends.  The call uses the slots
 *   stats_slot0 .. stats_slot0 + gfe_conv3d_stat_slots(B, D, H, W, Cout) - 1 (one per persistent block when Cout <= 64, one per
 *   tile otherwise); slots it does not write stay zero.  Calls that build one tensor together (the 8 parity classes of a
 *   transposed conv) use disjoint slot ranges;
 *   consume with gfe_groupnorm_from_partials.  Granularity: for Cout >= 64 (64-channel tiles) channel 8k holds the sums of channels
 *   8k .. 8k+7 and the other seven read zero -- exact for GroupNorm groups that are multiples of 8 channels, which is why Cout must
 *   then be a multiple of 64 (GFE_ERR_SHAPE otherwise; narrower outputs keep per-channel sums). */
int gfe_conv3d_igemm(const void* x, const void* w_packed, int64_t w_batch_stride, const float* bias, const float* bias_tab,
                     const void* res, void* y,
                     int64_t B, int64_t D, int64_t H, int64_t W, int64_t Cin, int64_t Cout,
                     int64_t OD, int64_t OH, int64_t OW,
                     int ntaps, const int8_t* tap_offsets,
                     int ostride, int op_d, int op_h, int op_w, int oshift, int relu,
                     float* stats_ws, int64_t stats_nblk, int64_t stats_slot0, void* stream);

/* Number of 8x8x8 output tiles per sample of gfe_conv3d_igemm on a (D, H, W) class grid = partial slots one call writes. Host-only. */
int gfe_conv3d_tiles(int64_t D, int64_t H, int64_t W);

/* The implicit-GEMM conv kernels run one persistent block per CU and draw their tiles from per-XCD ticket counters; with n > 0 they launch
 * 256 - n blocks, i.e. leave n CUs to the kernels of another stream (the head of the previous batch in the two-stream step; a conv block
 * fills a CU's registers and LDS, nothing else can share it).  Returns the previous value.  Process-wide; default 0 or GFE_CONV_RESERVE_CUS. */
int gfe_conv_reserve_cus(int n);

/* A HIP stream whose kernels run only on CU-mask bits lo .. hi-1 of the 256 (hipExtStreamCreateWithCUMask; bit i = CU slot i / 8 of XCD
 * i % 8, so ranges that are multiples of 8 take the same CUs from every XCD).  The two-stream step runs the head of batch k on bits
 * 0 .. n-1 and the generator of batch k+1 on bits n .. 255 with gfe_conv_reserve_cus(n): the head's latency-bound chain then advances all
 * the time instead of only in the gaps between conv launches that fill every CU.  The handle is a hipStream_t (wrap it with
 * torch.cuda.ExternalStream); destroy it with gfe_stream_destroy once nothing is queued on it. */
int gfe_stream_create_cu_range(int lo, int hi, void** stream);
int gfe_stream_destroy(void* stream);
/* GroupNorm-partial slots per sample one gfe_conv3d_igemm call with stats_ws writes (see above). Host-only. */
int gfe_conv3d_stat_slots(int64_t B, int64_t D, int64_t H, int64_t W, int64_t Cout);

/* 1x1x1 convolution Cin -> Cout + bias of a channels-last tensor, as a plain streaming product (ABI 49; csrc/conv1x1.hip).
 * Replaces: the encoders' ResNetBlock.conv1 = nn.Conv3d(in, out, 1) (pytorch3dunet/unet3d/buildingblocks.py:204-208, applied at :218-229),
 * which rounds 1-5 ran through gfe_conv3d_igemm with a one-tap list.
 *   x: (B, V, Cin) bf16, V = D*H*W voxels per sample.  w: (Cout, Cin) bf16, the Conv3d weight as it is (no packing).  bias: (Cout) f32 or NULL.
 *   y: (B, V, Cout) bf16.  All four pointers 16-byte aligned, V * max(Cin, Cout) * 2 < 2^31.  Cin in {64, 128}; Cout a multiple of 64 with Cout / 128 (Cin == 64 and Cout % 128 == 0) or Cout / 64 in {1, 2, 4}.
 *   stats_ws (or NULL): GroupNorm partials of y, layout (B, stats_nblk, 2, Cout) f32 as gfe_conv3d_igemm's: the call writes EVERY element of
 *   slots stats_slot0 .. stats_slot0 + gfe_conv1x1_stat_slots(V) - 1 (one per block of a sample, <= 64 of them, a function of V alone; sums of the rounded
 *   outputs per 8 consecutive channels on the octet's first channel, zeros on the other seven), so the workspace needs no zero fill. */
int gfe_conv1x1(const void* x, const void* w, const float* bias, void* y, int64_t B, int64_t V, int64_t Cin, int64_t Cout,
                float* stats_ws, int64_t stats_nblk, int64_t stats_slot0, void* stream);
/* GroupNorm-partial slots per sample one gfe_conv1x1 call with stats_ws writes. Host-only. */
int gfe_conv1x1_stat_slots(int64_t V);

/* The whole TransposeConvUpsampling + summation join (buildingblocks.py:396-400, 523-537) in ONE launch: the 8 output-parity
 * classes that gfe_conv3d_igemm takes as 8 calls become the innermost dimension of the persistent blocks' work list, so a block
 * re-reads an activation tile for the next class out of L2 and the classes pipeline into each other.
 *   w_packed: the classes' packed weight sets back to back (w_elems bf16 in total), class c at element offset cls_woff[c]
 *   (HOST int64[8], multiples of 8) with cls_ntaps[c] taps (HOST int[8], 1/2/2/2/4/4/4/8 in any order) and output parity
 *   cls_parity[3c..3c+2] (HOST int8, 0/1 per axis); tap_offsets: HOST int8, the classes' tap lists concatenated, offsets 0/+1.
 *   res (skip) may be NULL.  stats_ws: (B, stats_nblk, 2, Cout) f32 ZEROED, stats_nblk >= gfe_convt3d_stat_slots(...). */
int gfe_convt3d_stat_slots(int64_t B, int64_t D, int64_t H, int64_t W, int64_t Cout);
int gfe_convt3d_k3s2_fused(const void* x, const void* w_packed, const int64_t* cls_woff, const int* cls_ntaps, const int8_t* cls_parity,
                           const int8_t* tap_offsets, int64_t w_elems, const void* res, void* y,
                           int64_t B, int64_t D, int64_t H, int64_t W, int64_t Cin, int64_t Cout, int64_t OD, int64_t OH, int64_t OW,
                           int oshift, float* stats_ws, int64_t stats_nblk, void* stream);

/* Folds GroupNorm(x) = scale[b,c]*x + shift[b,c] (from gfe_groupnorm_scale_shift) into the convolution that consumes it
 * (create_conv order 'g' before 'c', buildingblocks.py:55-67; zero padding is applied AFTER the norm):
 *   w_out[b] = bf16(w_packed_f32 * scale[b, cin])                              (B sets in the gfe_conv3d_igemm layout; NULL: the bias table alone)
 *   bias_tab[b][class][co] = sum over taps inside the volume for that boundary class of sum_ci W[t][co][ci]*shift[b,ci]
 * w_packed_f32: the packed layout in f32.  T_ws: (B, ceil(Cin / 32), ntaps, CoutPad) f32 workspace (per-slab partials, summed in a fixed
 * order: the tables are bit-reproducible).  tap_offsets_dev: DEVICE int8 ntaps x 3. */
int gfe_conv3d_fold_groupnorm(const float* w_packed_f32, const float* gn_scale, const float* gn_shift, void* w_out, float* T_ws,
                              float* bias_tab, const int8_t* tap_offsets_dev, int64_t B, int64_t Cin, int64_t Cout, int ntaps, void* stream);

/* GroupNorm statistics -> per-(sample, channel) affine: scale = rstd*gamma, shift = beta - mean*rstd*gamma
 * (nn.GroupNorm(G, C), eps, biased variance; buildingblocks.py:55-67).  x: (B, S, C) bf16 channels-last, S = D*H*W.
 * ws: (B, nblk, 2, C) f32 workspace, nblk from gfe_groupnorm_plan.  scale, shift: (B, C) f32. */
int gfe_groupnorm_plan(int64_t S, int* vox_per_block, int* nblk);
int gfe_groupnorm_scale_shift(const void* x, const float* gamma, const float* beta, float* scale, float* shift, float* ws,
                              int64_t B, int64_t S, int64_t C, int64_t G, float eps, void* stream);

/* The same affine from partials a producer already wrote (gfe_conv3d_igemm / gfe_conv_in1 `stats_ws`): ws (B, nblk, 2, C) f32
 * per-slot channel sums / sums of squares over disjoint voxel sets covering all S voxels.  ws2: (B, 32, 2, C) f32 scratch
 * (used when nblk > 512: a first kernel folds the slots 32-ways in parallel). */
int gfe_groupnorm_from_partials(const float* ws, int64_t nblk, const float* gamma, const float* beta, float* scale, float* shift,
                                float* ws2, int64_t B, int64_t S, int64_t C, int64_t G, float eps, void* stream);

/* MaxPool3d(kernel 2, stride 2) (buildingblocks.py:284): (B, D, H, W, C) -> (B, D/2, H/2, W/2, C) bf16. */
int gfe_maxpool3d_2(const void* x, void* y, int64_t B, int64_t D, int64_t H, int64_t W, int64_t C, void* stream);

/* encoders.0.basic_module.conv1: Conv3d(1, C, 1) + bias (buildingblocks.py:191-196). x: (nvox) f32|bf16 -> y: (nvox, C) bf16. */
int gfe_conv_in1(const void* x, const float* w, const float* bias, void* y, int64_t nvox, int64_t C, int in_dtype, void* stream);
/* Same, per sample, additionally writing the GroupNorm partials of y: stats_ws (B, nblk, 2, C) f32 with nblk = gfe_conv_in1_nblk(S). */
int gfe_conv_in1_nblk(int64_t S);
int gfe_conv_in1_stats(const void* x, const float* w, const float* bias, void* y, float* stats_ws, int64_t B, int64_t S, int64_t C,
                       int in_dtype, void* stream);

/* final_conv: Conv3d(C, 1, 1) + bias (pytorch3dunet/unet3d/model.py:123,162). x: (nvox, C) bf16 -> y: (nvox) f32. */
int gfe_conv_out1(const void* x, const float* w, float bias, float* y, int64_t nvox, int64_t C, void* stream);

/* Bottleneck fold 'b c (md1 md2) h w -> b c (h md1) (md2 w)' (model.py:150) on channels-last data:
 * src (B, D, H, W, C) -> dst (B, H*md1, (D/md1)*W, C); inverse != 0 runs model.py:152 (dst layout -> src layout). */
int gfe_fold_mid(const void* src, void* dst, int64_t B, int64_t D, int64_t H, int64_t W, int64_t C, int md1, int inverse, void* stream);

/* ---------------------------------------------------------------------------------------------
 * GEMM (every nn.Linear of the path)
 * ------------------------------------------------------------------------------------------- */

/* C[M][N] = act(A[M][K] . B[N][K]^T + bias) + res   -- y = x W^T + b with W stored (N, K) like nn.Linear.weight.
 *   A, B bf16 (lda, ldb in elements, multiples of 8; K % 8 == 0, N % 4 == 0).  bias (N) f32 or NULL.
 *   res: (M, ldres) bf16|f32 or NULL.  act: 0 none, 1 exact-erf GELU.  C: bf16 or f32 (out_f32).
 *   split_k > 1: K is cut into split_k ranges that are ADDED into an f32 C the caller initialised (zeros, or an accumulation target;
 *   no act/res): with splitk_ws (split_k * M * N floats, ldc % 4 == 0) every range stores its tile there and a second launch sums the
 *   ranges in a fixed order (bit-reproducible); with splitk_ws == NULL the ranges add with f32 atomics. */
int gfe_gemm_bf16_nt(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc,
                     int64_t M, int64_t N, int64_t K, const float* bias, const void* res, int64_t ldres, int res_f32,
                     int act, int out_f32, int split_k, float* splitk_ws, void* stream);

/* The same GEMM with per-operand source modes, so that the backward of a Linear needs no cast / transpose pass:
 *   mode bit 0: the operand is f32 in memory (rounded to bf16 while it is staged; ld % 4 == 0 instead of % 8);
 *   mode bit 1: the operand is reduction-major, element (row, k) at base[k * ld + row]  (A: row = m, B: row = n).
 * Forward  y = x W^T      : A = x (f32, mode 1),          B = W16 (mode 0)
 * dgrad    dx = dy W      : A = dy (f32, mode 1),         B = W16 as (N,K) read reduction-major (mode 2)
 * wgrad    dW = dy^T x    : A = dy (f32, mode 3),         B = x (f32, mode 3); the reduction length M is then unconstrained.
 * K % 8 == 0 unless both operands are reduction-major.  Everything else as gfe_gemm_bf16_nt. */
int gfe_gemm_ex(const void* A, int64_t lda, int a_mode, const void* B, int64_t ldb, int b_mode, void* C, int64_t ldc,
                int64_t M, int64_t N, int64_t K, const float* bias, const void* res, int64_t ldres, int res_f32,
                int act, int out_f32, int split_k, float* splitk_ws, void* stream);
/* Diagnostic: how many GEMM calls of this process ran on the persistent LDS-DMA main loop (csrc/gemm_dma.hip: plain bf16 x bf16 K-major
 * operands, M >= 512, N % 128 == 0, K % 64 == 0, one K range; everything else stays on gemm_nt_kernel).  GFE_GEMM_NO_DMA=1 turns it off. */
int gfe_gemm_dma_launches(void);


/* Exact-f32 GEMM on the f32 matrix cores for the trainable head's small Linears (the reference trains the head in fp32:
 * classify_mamba.py:69-74; Mamba projections mamba.py:204, 235-238, 223; q / out projections sd_cross_atten.py:42-45; GEGLU FF
 * corss_ft_transformer.py:15-22; logits mamba_transformer.py:79-82).
 *   C[M][N] (+)= op(A)[M][K] op(B)[N][K]^T (+ bias[n]), all f32, any sizes / alignments.
 *   a_tr / b_tr != 0: the operand is stored reduction-major, element (row, k) at base[k * ld + row].
 *   accumulate != 0: add to C (a gradient buffer).  split_k > 1 with splitk_ws (split_k * M * N floats): every K range stores its
 *   tile there and a second launch writes C = (accumulate ? C : 0) + bias + the ranges in a fixed order (bit-reproducible; C needs no
 *   zero fill).  split_k > 1 with splitk_ws == NULL: the ranges are added with f32 atomics -- C must be zeroed, or be the buffer that
 *   is accumulated into. */
int gfe_gemm_f32(const float* A, int64_t lda, int a_tr, const float* B, int64_t ldb, int b_tr, float* C, int64_t ldc,
                 int64_t M, int64_t N, int64_t K, const float* bias, int accumulate, int split_k, float* splitk_ws, void* stream);

/* 1 when gfe_gemm_f32 takes the in-block K split for these operands (K-major A, both operands 16-byte aligned with leading dimensions
 * that are multiples of 4, K % 16 == 0, launch-bound size; reduction-major B also N % 4 == 0): one block per output tile, its waves cut
 * K between them and sum their tiles in a fixed tree through LDS -- split_k / splitk_ws are then ignored (no reduction launch, nothing to
 * allocate).  0: the staged kernel with the caller's split_k.  Same sums either way up to f32 summation order; both are bit-reproducible. */
int gfe_gemm_f32_inblock(const float* A, int64_t lda, int a_tr, const float* B, int64_t ldb, int b_tr, int64_t M, int64_t N, int64_t K);

/* conv2(GroupNorm(conv1(x))) of a ResNetBlock (buildingblocks.py:38-67) as one convolution of x: operands and result layout of the per-sample
 * effective-weight product W_eff = W2 . diag(scale_b) . W1 (one gfe_gemm_f32 for all samples).
 *   prep: rhs[c][b * Cin + i] = scale[b][c] * w1[c][i]   (C x B*Cin),   shift2[b][c] = scale[b][c] * b1[c] + shift[b][c]
 *   pack: w_out[b][slab][tap][o][k] = bf16(weff[tap * cout_pad + o][b * Cin + slab * 32 + k])  (k past Cin: 0): B weight sets in gfe_conv3d_igemm's layout */
int gfe_lift_fold_prep(const float* scale, const float* shift, const float* w1, const float* b1, float* rhs, float* shift2,
                       int64_t B, int64_t C, int64_t Cin, void* stream);
int gfe_lift_fold_pack(const float* weff, void* w_out, int64_t B, int64_t Cin, int64_t cout_pad, int ntaps, void* stream);

/* out[b][c][r] = in[b][r][c], bf16 (operand re-layout for dgrad / wgrad). */
int gfe_transpose_bf16(const void* in, void* out, int64_t batch, int64_t R, int64_t Cc, int64_t ldi, int64_t ldo, void* stream);

/* f32 <-> bf16 element cast (to_bf16 != 0: f32 -> bf16). */
int gfe_cast(const void* in, void* out, int64_t n, int to_bf16, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Bottleneck ViT helpers (vit_pytorch_diy/vit.py)
 * ------------------------------------------------------------------------------------------- */

/* LayerNorm(len) with affine f32 gamma/beta over `rows` logical rows (nn.LayerNorm, eps; vit.py:97,99,103,108 and every
 * pre-LN of the blocks).  Rows may be stored as `nseg` contiguous segments so that the patchify / un-patchify
 * rearranges (vit.py:96, 109) are folded into the load / store addressing.
 * in_map / out_map: HOST int64[8] = {batch_stride, outer_stride, inner_stride, seg_stride, rows_per_batch, n_inner, nseg, seglen}:
 *   base(r, s) = (r / rpb)*batch_stride + ((r % rpb) / n_inner)*outer_stride + ((r % rpb) % n_inner)*inner_stride + s*seg_stride. */
int gfe_layernorm(const void* x, void* y, const float* gamma, const float* beta, const int64_t* in_map, const int64_t* out_map,
                  int64_t rows, float eps, int in_dtype, int out_dtype, void* stream);

/* softmax(q k^T * scale) v per (batch, head) for short sequences (vit.py:55-62; also CrossAttention sd_cross_atten.py:61-65).
 * bf16 q/k/v/o addressed as base + b*batch_stride + row*row_stride + h*dh + c.  nk <= 256, dh <= 64. */
int gfe_attention_small(const void* q, const void* k, const void* v, void* o, int64_t B, int64_t H, int64_t nq, int64_t nk, int64_t dh,
                        int64_t q_batch, int64_t q_row, int64_t k_batch, int64_t k_row, int64_t v_batch, int64_t v_row,
                        int64_t o_batch, int64_t o_row, float scale, void* stream);

/* The same product for long sequences (flash-style: K/V tiles of 64 keys, online softmax, bf16 MFMA 32x32x16, f32 statistics):
 * the attention of vit_pytorch_diy/vit_3d.py:47-57 (SURVEY 8-d synthetic 3-D ViT, n = 1729).  dh == 64, any n >= 1; q/k/v/o
 * addressed as in gfe_attention_small (row strides multiples of 8 elements for q/k/v, 4 for o). */
int gfe_attention_fwd(const void* q, const void* k, const void* v, void* o, int64_t B, int64_t H, int64_t n, int64_t dh,
                      int64_t q_batch, int64_t q_row, int64_t k_batch, int64_t k_row, int64_t v_batch, int64_t v_row,
                      int64_t o_batch, int64_t o_row, float scale, void* stream);

/* Training forward of the same product: additionally writes the row statistic the backward restarts from,
 * nlse[b][h][row] = -(max + log2(sum)) in log2 units (f32, [B][H][npad], npad = n rounded up to 64; rows n .. npad-1 = -inf).
 * p_drop > 0: `attn = dropout(attn)` of vit_3d.py:56 on the probabilities (kept ones scaled by 1 / (1 - p)); the mask is a counter-based
 * hash of (seed, batch*head, row, key) (csrc/attn_drop.h) that the backward regenerates; n <= 65535 then. */
int gfe_attention_fwd_lse(const void* q, const void* k, const void* v, void* o, void* nlse, int64_t B, int64_t H, int64_t n, int64_t dh,
                          int64_t q_batch, int64_t q_row, int64_t k_batch, int64_t k_row, int64_t v_batch, int64_t v_row,
                          int64_t o_batch, int64_t o_row, float scale, float p_drop, int64_t seed, void* stream);

/* Backward of gfe_attention_fwd_lse (what autograd does for vit_pytorch_diy/vit_3d.py:47-57), same p_drop / seed: bf16 dq / dk / dv from
 * bf16 q / k / v / o / dout and the forward's nlse.  q/k/v share (in_batch, in_row), o/dout (o_batch, o_row), dq/dk/dv (g_batch, g_row)
 * (element strides; in/o multiples of 8, g of 4).  Deterministic: no atomics, one owner per output element.
 * Workspaces: qs_ws bf16 [B*H*npad*64], ndelta_ws f32 [B*H*npad].  dh == 64. */
int gfe_attention_bwd(const void* q, const void* k, const void* v, const void* o, const void* dout, const void* nlse,
                      void* dq, void* dk, void* dv, void* qs_ws, void* ndelta_ws, int64_t B, int64_t H, int64_t n, int64_t dh,
                      int64_t in_batch, int64_t in_row, int64_t o_batch, int64_t o_row, int64_t g_batch, int64_t g_row,
                      float scale, float p_drop, int64_t seed, void* stream);

/* from_patch_embedding's Linear over the token axis (vit.py:104-106): y[b][j][:] = sum_i W[j][i] x[b][i][:] + bias[j].
 * x: (B, nin, dim) f32|bf16, W: (nout, nin) f32, y: (B, nout, dim) bf16. */
int gfe_token_mix(const void* x, const float* w, const float* bias, void* y, int64_t B, int64_t nin, int64_t nout, int64_t dim,
                  int in_dtype, void* stream);

/* cls token + positional embedding (vit.py:127-130): x[b][0] = cls + pos[0], x[b][1+i] = tok[b][i] + pos[1+i]; all f32. */
int gfe_vit_embed(const float* tok, const float* cls, const float* pos, float* x, int64_t B, int64_t n, int64_t dim, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Group B / G -- trainable head and the optimiser step
 * ------------------------------------------------------------------------------------------- */

/* Combine_classfier_vit_mid (classify/classifier.py:329-333): Linear(H*W -> S) over cat([mid_input, mid_output], dim=1),
 * evaluated on the generator's channels-last mid features without materialising the concat:
 *   out[b][src*C + c][s] = bias[s] + sum_hw mid_src[b][hw][c] * W[s][hw]      (the final transpose is a host-side view)
 * mid_in, mid_out: (B, HW, C) bf16; Wt: the weight TRANSPOSED to (HW, S) f32 (a row's S weights are one 16-byte load); bias (S) or NULL;
 * out: (B, 2C, S) f32, overwritten.  ws: nchunks * 2B * C * S floats with nchunks from gfe_mid_linear_plan: one partial slab per block of
 * rows, summed in a fixed order by a second launch (no atomics: run-to-run identical).  S == 4, 256 % (C/8) == 0. */
int gfe_mid_linear_plan(int64_t B, int64_t HW, int* nchunks);
int gfe_mid_linear_fwd(const void* mid_in, const void* mid_out, const float* Wt, const float* bias, float* out, float* ws,
                       int64_t B, int64_t HW, int64_t C, int64_t S, void* stream);

/* Weight gradient of the above: dW[s][hw] = sum_{b,c} dout[b][c][s] * mid[b][hw][c]; dout: (B, 2C, S) f32; dW: (S, HW) f32. */
int gfe_mid_linear_wgrad(const void* mid_in, const void* mid_out, const float* dout, float* dW,
                         int64_t B, int64_t HW, int64_t C, int64_t S, void* stream);

/* ---- backward-pass helpers of the generator (main_gan_vit.py:68-82; csrc/gen_train.hip) ------------------------------
 * Channels-last bf16 tensors (B, V, C), C % 8 == 0.  The products of the backward reuse gfe_conv3d_* (dgrad = the forward kernel with
 * flipped taps and transposed weights) and gfe_gemm_ex (wgrad = voxel-reduction GEMM); these are the passes around them. */
/* x^ = scale[b,c] * x + shift[b,c]: the GroupNorm output (buildingblocks.py:55-67) materialised as the weight gradient's conv input. */
int gfe_gn_apply(const void* x, const float* scale, const float* shift, void* y, int64_t B, int64_t V, int64_t C, void* stream);
/* ReLU backward from the stored output: out = dy * (y > 0); n elements, n % 8 == 0. */
int gfe_mask_relu_bf16(const void* dy, const void* y, void* out, int64_t n, void* stream);
/* GroupNorm backward, pass 1: S1[b,c] += sum_v dx^, S2[b,c] += sum_v dx^ * (x - mu[b,c]) * rstd[b,c]  (mu / rstd: the group's values per channel).
 * ws (ABI 45): B * gfe_gn_bwd_sums_blocks(V) * 2C floats -- every block stores its partial row there and a second launch adds them in order
 * (bit-reproducible); NULL: f32 atomics.  Shape contract: C % 8 == 0, C <= 2048 and 256 % (C / 8) == 0 -- a 256-thread block is cut into
 * 256 / (C / 8) voxel lanes of C / 8 eight-channel threads, so C / 8 must divide 256 (C in {8, 16, 32, 64, 128, 256, 512, 1024, 2048}: every
 * generator width); anything else returns GFE_ERR_SHAPE.  The same holds for gfe_conv_out1_bwd (ws: gfe_gen_rows_blocks(rows) * (C + 1) floats) and
 * gfe_conv_in1_wgrad (gfe_gen_rows_blocks(rows) * 2C). */
int gfe_gn_bwd_sums_blocks(int64_t V);
int gfe_gen_rows_blocks(int64_t rows);
int gfe_gn_bwd_sums(const void* dxhat, const void* x, const float* mu, const float* rstd, float* S1_zeroed, float* S2_zeroed, float* ws,
                    int64_t B, int64_t V, int64_t C, void* stream);
/* pass 2: dx = rstd * (gamma * dx^ - coef_a[b,c] - xn * coef_b[b,c]) (+ add_in: a second gradient into the same tensor, or NULL);
 * coef_a / coef_b = the group's mean of gamma * S1 / gamma * S2 over (channels of the group x voxels), per channel. */
int gfe_gn_bwd_apply(const void* dxhat, const void* x, const float* mu, const float* rstd, const float* gamma, const float* coef_a, const float* coef_b,
                     const void* add_in, void* dx, int64_t B, int64_t V, int64_t C, void* stream);
/* nn.MaxPool3d(2) backward (buildingblocks.py:284): dy (B, D/2, H/2, W/2, C) -> dx (B, D, H, W, C), zero on entry; the first maximum of
 * a window in (d, h, w) order receives the gradient. */
int gfe_maxpool2_bwd(const void* x, const void* dy, void* dx_zeroed, int64_t B, int64_t D, int64_t H, int64_t W, int64_t C, void* stream);
/* final 1x1x1 conv C -> 1 (model.py:123, 162) backward: x (rows, C) bf16, dy (rows) f32: dx = dy * w; dw += sum dy * x; db += sum dy. */
int gfe_conv_out1_bwd(const void* x, const float* dy, const float* w, void* dx, float* dw_accum, float* db_accum, float* ws, int64_t rows, int64_t C, void* stream);
/* 1x1x1 lift gradients from dr (rows, C) bf16: db[c] += sum dr; with x (rows) f32 of a one-channel input also dw[c] += sum dr * x
 * (x == NULL and dw == NULL: bias gradient only). */
int gfe_conv_in1_wgrad(const float* x, const void* dr, float* dw_accum, float* db_accum, float* ws, int64_t rows, int64_t C, void* stream);

/* nn.L1Loss()(pred, target) -- the generator's reconstruction loss (main_gan_vit.py:72) -- value and gradient in one pass: loss[0] = mean |pred - target|,
 * dpred = sign(pred - target) / n; part_ws holds gfe_l1_loss_blocks(n) floats (per-block partial sums, added in order: no atomics). */
int gfe_l1_loss_blocks(int64_t n);
int gfe_l1_loss(const float* pred, const float* target, float* loss, float* dpred, float* part_ws, int64_t n, void* stream);

/* Weight gradient of a tap-list convolution (the backward of nn.Conv3d k3 p1 / the parity classes of ConvTranspose3d k3 s2 p1,
 * pytorch3dunet/unet3d/buildingblocks.py:46-52, 523-537), ALL taps in one launch (csrc/conv_wgrad.hip):
 *   dw[t][co][ci] = sum_v dout[v][co] * x[v + taps[t]][ci]     x (B, D, H, W, Ci), dout (B, D, H, W, Co) bf16 channels-last,
 *   taps_host: ntaps x 3 int8 offsets in {-1, 0, 1} (HOST memory, read before the call returns), ntaps <= 27,
 *   Ci % 32 == 0, Co % 64 == 0.  dw (ntaps, Co, Ci) f32 is WRITTEN; part_ws holds gfe_conv3d_wgrad_splits() * ntaps * Co * Ci floats. */
int gfe_conv3d_wgrad_splits(int64_t B, int64_t D, int64_t H, int64_t W, int64_t Ci, int64_t Co);
int gfe_conv3d_wgrad(const void* x, const void* dout, float* part_ws, float* dw, const int8_t* taps_host, int ntaps,
                     int64_t B, int64_t D, int64_t H, int64_t W, int64_t Ci, int64_t Co, void* stream);

/* ---- small operators of the trainable head, f32, one launch per operator and direction (csrc/head_ops.hip) ----------------- */

/* Token sequence of Cross_mamba_both (cross_atten/mamba_transformer.py:97-117): out (B, L, dim), L = 1 + ncat + ncont + nf,
 *   out[b] = [cls | emb[x_cat[b,j] + offsets[j]] | x_num[b,j] * num_w[j] + num_b[j] (NumericalEmbedder, corss_ft_transformer.py:159-163) | feat[b]].
 *   x_cat (B, ncat) int64, offsets (ncat) int64 (the categories_offset buffer), emb (ntok, dim), feat (B, nf, dim) or NULL when nf = 0.
 * bwd: d_emb / d_num_w / d_num_b / d_cls are ACCUMULATED into (gradient buffers), d_feat (B, nf, dim) is written (NULL: not wanted). */
int gfe_embed_tokens_fwd(const int64_t* x_cat, const int64_t* offsets, const float* emb, const float* x_num, const float* num_w,
                         const float* num_b, const float* cls, const float* feat, float* out,
                         int64_t B, int64_t ncat, int64_t ncont, int64_t nf, int64_t dim, int64_t ntok, void* stream);
int gfe_embed_tokens_bwd(const float* dout, const int64_t* x_cat, const int64_t* offsets, const float* x_num,
                         float* d_emb, float* d_num_w, float* d_num_b, float* d_cls, float* d_feat,
                         int64_t B, int64_t ncat, int64_t ncont, int64_t nf, int64_t dim, int64_t ntok, void* stream);

/* torch.mean(x, dim=1) over (B, L, dim) (mamba_transformer.py:122) and its adjoint. */
int gfe_mean_tokens_fwd(const float* x, float* y, int64_t B, int64_t L, int64_t dim, void* stream);
int gfe_mean_tokens_bwd(const float* dy, float* dx, int64_t B, int64_t L, int64_t dim, void* stream);

/* Core of CrossAttention with one query per sample (cross_atten/sd_cross_atten.py:58-68): q (B, H*dh), k / v (B, nk, H*dh),
 * out (B, H*dh) = softmax(q k^T * scale) v per head; probs (B, H, nk) is kept for the backward.  bwd writes dq, dk, dv. */
int gfe_cross_attn_q1_fwd(const float* q, const float* k, const float* v, float* out, float* probs,
                          int64_t B, int64_t H, int64_t nk, int64_t dh, float scale, void* stream);
int gfe_cross_attn_q1_bwd(const float* q, const float* k, const float* v, const float* probs, const float* dout,
                          float* dq, float* dk, float* dv, int64_t B, int64_t H, int64_t nk, int64_t dh, float scale, void* stream);

/* CrossAttention with one query per sample with the K / V projections folded away (cross_atten/sd_cross_atten.py:49-70 as called at
 * cross_atten/mamba_transformer.py:122-124 on the condition of :89-94; csrc/xattn_fold.hip): nothing of size keys x E is ever formed.
 *   q (B, E = H*dh) f32 = q_proj's output; Wk, Wv (E, HW) f32 = k_proj.weight / v_proj.weight; bv (E) = v_proj.bias or NULL
 *   (k_proj.bias shifts every score of a head by the same amount: it drops out of the softmax, its gradient is exactly zero);
 *   img0..img{n_img-1}: the condition images, each (B, HW, D3) f32 contiguous = the volume (B, 1, h, w, d) itself; key img*D3 + j is the
 *   (h w)-vector img[b, :, j] (rearrange 'b c h w d -> b (c d) (h w)').
 *   fwd writes p (B, H, n_img*D3) softmax probabilities, c (B, H, HW) = sum_j p_j y_j (both kept for the backward) and
 *   o (B, E) = W_v c + b_v = the attention output BEFORE out_proj.  Workspaces: r_ws (B, H, HW), part_ws (B, n_img, chunks(HW), H, D3).
 *   bwd: d_o (B, E) -> dq (B, E) and dr (B, H, HW) written (dr feeds _wgrad); workspaces dc_ws (B, H, HW), ds_ws (B, H, n_img*D3),
 *   part_ws as above.  _wgrad ACCUMULATES dWk, dWv (E, HW) and dbv (E) into gradient slots.  Every sum in a fixed order: bit-reproducible.
 * Limits: H <= 64, D3 <= 256, n_img <= 4, n_img*D3 <= 1024.  16-byte vector accesses when HW % 4 == 0, D3 % 4 == 0 and all bases are
 * 16-byte aligned, scalar accesses otherwise. */
int gfe_cross_attn_q1_folded_chunks(int64_t HW);
int gfe_cross_attn_q1_folded_fwd(const float* q, const float* Wk, const float* Wv, const float* bv,
                                 const float* img0, const float* img1, const float* img2, const float* img3, int n_img,
                                 float* r_ws, float* part_ws, float* p, float* c, float* o,
                                 int64_t B, int64_t H, int64_t dh, int64_t HW, int64_t D3, void* stream);
int gfe_cross_attn_q1_folded_bwd(const float* d_o, const float* Wk, const float* Wv,
                                 const float* img0, const float* img1, const float* img2, const float* img3, int n_img,
                                 const float* p, float* dc_ws, float* part_ws, float* ds_ws, float* dr, float* dq,
                                 int64_t B, int64_t H, int64_t dh, int64_t HW, int64_t D3, void* stream);
/* the weight gradients, leaves of the backward (the caller may enqueue them on another stream, behind _bwd):
 * dWv += d_o (x) c, dbv += sum_b d_o (dbv may be NULL), dWk += q (x) dr, per head; c from _fwd, dr from _bwd. */
int gfe_cross_attn_q1_folded_wgrad(const float* d_o, const float* q, const float* c, const float* dr, float* dWk, float* dWv, float* dbv,
                                   int64_t B, int64_t H, int64_t dh, int64_t HW, void* stream);

/* Small multi-head self-attention for Jamba's AttentionSDPA (cross_atten/jamba.py:342-398: F.scaled_dot_product_attention, is_causal
 * when no cache is passed): q, k, v, out (B, L, H*dh) f32, L <= 64, dh <= 64; probs (B, H, L, L) kept for the backward.
 * p_drop > 0: dropout on the attention probabilities (the generator's ViT in training, vit_pytorch_diy/vit.py:59), mask = hash of
 * (seed, element), the same (p_drop, seed) must be given to the backward; probs holds the softmax before the dropout. */
int gfe_sdpa_small_fwd(const float* q, const float* k, const float* v, float* out, float* probs,
                       int64_t B, int64_t H, int64_t L, int64_t dh, float scale, int causal, float p_drop, int64_t seed, void* stream);
int gfe_sdpa_small_bwd(const float* q, const float* k, const float* v, const float* probs, const float* dout,
                       float* dq, float* dk, float* dv, int64_t B, int64_t H, int64_t L, int64_t dh, float scale, float p_drop, int64_t seed, void* stream);

/* nn.LayerNorm(dim) over (rows, dim) f32 (mamba_transformer.py:79-82, corss_ft_transformer.py:16); mean / rstd (rows) kept for the
 * backward, which ACCUMULATES dgamma / dbeta (f32 atomics) and writes dx.  ws: NULL, or rows * 128 floats of scratch -- with it, rows of
 * dim >= 16384 (the generator ViT's LayerNorm(patch_dim), vit.py:101-105) are cut into up to 64 segments that run on different CUs.
 * Backward with >= 256 rows of dim <= 2048 (dim % 4 == 0; the 3-D ViT's token rows) and ws of 2 * dim * 1024 floats: per-block partial
 * rows + a fixed-order reduction instead of the atomics (deterministic). */
int gfe_layernorm_rows_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd, float* ws,
                           int64_t rows, int64_t dim, float eps, void* stream);
int gfe_layernorm_rows_bwd(const float* x, const float* gamma, const float* mean, const float* rstd, const float* dy, float* dx,
                           float* dgamma, float* dbeta, float* ws, int64_t rows, int64_t dim, void* stream);

/* Exact-erf GELU as an operator of its own and its adjoint dx = dy * gelu'(x) (vit_pytorch_diy/vit.py:19 / vit_3d.py:21 under autograd: the
 * inference path carries the activation in the GEMM epilogue; training has to keep the pre-activation).  n elements, dtype GFE_F32 | GFE_BF16. */
int gfe_gelu_fwd(const void* x, void* y, int64_t n, int dtype, void* stream);
int gfe_gelu_bwd(const void* x, const void* dy, void* dx, int64_t n, int dtype, void* stream);
/* nn.Dropout(p) in training mode (vit.py:24-27, 62; vit_3d.py:25-28): y = x * keep / (1 - p), keep = hash(seed, element) (csrc/attn_drop.h);
 * its backward is the same call on dy with the same (p, seed) -- no mask is stored.  n elements, dtype GFE_F32 | GFE_BF16. */
int gfe_dropout(const void* x, void* y, int64_t n, float p_drop, int64_t seed, int dtype, void* stream);

/* GEGLU (corss_ft_transformer.py:10-13: x, gates = chunk(2); x * gelu(gates), exact erf) followed by Dropout(p_drop) (:19):
 * x (rows, 2F) -> y (rows, F).  The mask is a counter-based hash of (seed, element index): the backward regenerates it from the
 * same seed; p_drop = 0 in eval mode.  seed_step: NULL, or a DEVICE int64 counter that is mixed into the seed when the kernel runs -- a HIP
 * graph bakes the by-value seed into its node, so a replayed step gets a fresh mask only through memory (the captured step increments
 * the counter once before its forward; forward and backward of one step read the same value). */
int gfe_geglu_fwd(const float* x, float* y, int64_t rows, int64_t F, float p_drop, int64_t seed, const int64_t* seed_step, void* stream);
int gfe_geglu_bwd(const float* x, const float* dy, float* dx, int64_t rows, int64_t F, float p_drop, int64_t seed, const int64_t* seed_step, void* stream);

/* BCELoss(sigmoid(logits), y), mean over n samples (classify_mamba.py:67, 104), logs clamped at -100 as torch does: loss[0], and
 * (dlogits != NULL) the gradient of that mean loss w.r.t. the logits in the same launch. */
int gfe_bce_sigmoid(const float* logits, const float* y, float* loss, float* dlogits, int64_t n, void* stream);

/* Per-PARAMETER clip_grad_norm_(p, max_norm) followed by one Adam step (classify_mamba.py:64, 106-108) over flat f32
 * buffers holding every trainable tensor back to back.  chunks: device array of {int64 offset, int32 length, int32 tensor_id}
 * (16 B each), sorted by tensor id, no chunk crossing a tensor boundary.  norm2: (ntensors) f32 and partials: (nchunks) f32, both
 * overwritten: every chunk leaves its sum of squares, one block per tensor adds its chunks' partials in a fixed order (no atomics: the
 * clip factor is bit-identical on every rank and in every run, so data-parallel replicas cannot drift apart).  grad_scale multiplies g
 * first (1/world_size after an all-reduce SUM).  p_bf16: optional flat bf16 copy of p refreshed in the same pass.
 * lr / betas / eps are doubles: the bias corrections 1 - beta^step are formed in double on the host, as torch.optim.Adam does. */
int gfe_clip_adam(float* p, const float* g, float* m, float* v, void* p_bf16, const void* chunks, int64_t nchunks,
                  float* norm2, int64_t ntensors, float* partials, float grad_scale, float max_norm, double lr, double beta1, double beta2, double eps,
                  int64_t step, void* stream);

/* RMSNorm (cross_atten/mamba.py:408-418): y = x * rsqrt(mean(x^2, -1) + eps) * w over (rows, dim) f32; rstd (rows) is kept for
 * the backward: dx = rstd*(dy*w - x*rstd^2*mean(dy*w*x)), dw_accum += sum_rows dy*x*rstd (column-owner blocks: one owner and one
 * summation order per element, no atomics).
 * ABI 43, the residual around a pre-norm block (mamba.py:103, `mixer(norm(x)) + x`) without launches of its own: xcopy (nullable) receives a
 * copy of x -- the buffer an accumulating out_proj then adds the block's output to; dadd (nullable, (rows, dim)) is added to dx -- the
 * gradient that reaches x through the residual branch. */
int gfe_rmsnorm_fwd(const float* x, const float* w, float* y, float* rstd, float* xcopy, int64_t rows, int64_t dim, float eps, void* stream);
int gfe_rmsnorm_bwd(const float* x, const float* w, const float* rstd, const float* dy, float* dx, float* dw_accum, const float* dadd,
                    int64_t rows, int64_t dim, void* stream);

/* Depthwise causal Conv1d(k = 4, padding = 3, [:L]) + bias + SiLU on (B, L, ED) f32 (cross_atten/mamba.py:128-131, 208-212);
 * w: (ED, 1, 4) as nn.Conv1d.weight, bias (ED) or NULL.  Backward: dx, and dw_accum / db_accum += the sum over the batch, formed from
 * per-sample partial rows in ws (B * 5 * ED floats) in sample order by a second launch (no atomics). */
int gfe_dwconv1d_silu_fwd(const float* x, int64_t ldx, const float* w, const float* bias, float* y, int64_t B, int64_t L, int64_t ED, int64_t KS, void* stream);
int gfe_dwconv1d_silu_bwd(const float* x, int64_t ldx, const float* w, const float* bias, const float* dy, float* dx, int64_t lddx, float* dw_accum, float* db_accum,
                          float* ws, int64_t B, int64_t L, int64_t ED, int64_t KS, void* stream);

/* Sparse mixture-of-experts MLP, cross_atten/jamba.py:441-535 (SparseMoEBlock: router -> softmax -> top-k -> per-expert
 * down(silu(gate(x)) * up(x)) -> weighted sum), row f-2.  csrc/moe.hip: the (token, expert) pairs are sorted by expert on the device
 * (stable counting sort, no host sync), every projection of ALL experts is one grouped exact-f32 GEMM, the combine is a gather.  f32.
 *   gfe_moe_route: logits (T, E) -> rw (T, K) routing probabilities, sel (T, K) expert ids (descending probability), tok_sorted (T*K)
 *     token of every pair in expert order, pos (T, K) position of pair (t, k) in that order, seg (E + 1) expert offsets.  E <= 64.
 *   gfe_moe_route_bwd: d logits from d rw (softmax over all E, gathered at sel; jamba.py:487-489).
 *   gfe_moe_gemm_rows: C[seg[e] + r][n] (+)= sum_k A[row][k] W_e(n, k) with row = gather ? gather[seg[e] + r] : seg[e] + r and
 *     W_e = w_table[e] K-major (a Linear's forward) or, w_tr != 0, reduction-major (its dgrad).  P = T*K rows in all.
 *   gfe_moe_gemm_wgrad: dW_e[n][k] (+)= sum_{r in segment e} dY[seg[e] + r][n] X[row][k], dW_e = dw_table[e].
 *   gfe_moe_act_fwd / _bwd: h = silu(g) * u and its adjoint.  gfe_moe_combine: out[t] (+)= sum_k w[t][k] rows[pos[t][k]] (w NULL: 1).
 *   gfe_moe_combine_bwd: d_rows[pos[t][k]] = w[t][k] dout[t], dw[t][k] = <dout[t], o[pos[t][k]]>. */
int gfe_moe_route(const float* logits, int64_t T, int64_t E, int64_t K, float* rw, int32_t* sel, int32_t* tok_sorted, int32_t* pos, int32_t* seg,
                  void* stream);
int gfe_moe_route_bwd(const float* logits, const int32_t* sel, const float* drw, float* dlogits, int64_t T, int64_t E, int64_t K, void* stream);
int gfe_moe_gemm_rows(const float* A, int64_t lda, const int32_t* gather, const float* const* w_table, int64_t ldw, int w_tr, float* C, int64_t ldc,
                      const int32_t* seg, int64_t E, int64_t P, int64_t N, int64_t K, int accumulate, void* stream);
int gfe_moe_gemm_wgrad(const float* dY, int64_t lddy, const float* X, int64_t ldx, const int32_t* gather, float* const* dw_table, int64_t lddw,
                       const int32_t* seg, int64_t E, int64_t N, int64_t K, int accumulate, void* stream);
int gfe_moe_act_fwd(const float* g, const float* u, float* h, int64_t n, void* stream);
int gfe_moe_act_bwd(const float* g, const float* u, const float* dh, float* dg, float* du, int64_t n, void* stream);
int gfe_moe_combine(const float* rows, const float* w, const int32_t* pos, float* out, int64_t T, int64_t K, int64_t D, int accumulate, void* stream);
int gfe_moe_combine_bwd(const float* dout, const float* o, const float* w, const int32_t* pos, float* d_rows, float* dw, int64_t T, int64_t K, int64_t D,
                        void* stream);

/* Single-token inference, cross_atten/mamba.py:342-405 (MambaBlock.step / ssm_step), f32:
 *   gfe_mamba_step_conv: xc = silu(conv1d over [cache (B, ED, 3) | x] + bias), cache_out = the window shifted by one (mamba.py:354-361, 371);
 *     x has row stride ldx (it is the first half of the in_proj output).
 *   gfe_mamba_step_ssm:  h_out = exp(dt A) h_in + dt B x, y = (h_out . C + D x) * silu(z) with dt = softplus(delta + delta_bias),
 *     A = -exp(A_log) (mamba.py:374-405); h_in NULL = zero state; Bm / Cm rows with stride ld_bc, z (NULL = no gate) with stride ld_z. */
int gfe_mamba_step_conv(const float* x, int64_t ldx, const float* cache_in, float* cache_out, const float* w, const float* bias, float* xc,
                        int64_t B, int64_t ED, int64_t KS, void* stream);
int gfe_mamba_step_ssm(const float* xc, const float* delta, const float* A_log, const float* Bm, const float* Cm, int64_t ld_bc,
                       const float* D, const float* delta_bias, const float* z, int64_t ld_z, const float* h_in, float* h_out, float* y,
                       int64_t B, int64_t ED, int64_t N, void* stream);

/* out[n] (+)= sum_m x[m][n] for a row-major (M, N) f32 matrix (accumulate != 0: added to what out holds) with row stride ld: the bias gradient of every nn.Linear. */
int gfe_colsum_f32(const float* x, float* out, int64_t M, int64_t N, int64_t ld, int accumulate, void* stream);
/* the same, bit-reproducible for any M: tall inputs write one partial row per row block into ws (gfe_colsum_rblocks(M, N) x N floats) and a
 * second launch folds them in order -- gfe_colsum_f32 adds them with f32 atomics beyond 4 096 rows (the 3-D ViT's 13 832 token rows). */
int gfe_colsum_rblocks(int64_t M, int64_t N);
int gfe_colsum_f32_ws(const float* x, float* out, float* ws, int64_t M, int64_t N, int64_t ld, int accumulate, void* stream);

/* Image condition (cross_atten/mamba_transformer.py:89-94): 'b c h w d -> (b c) (h w) d' then transpose(1, 2):
 * out[b][c][r] (bf16, row stride ldo, batch stride out_batch_stride) = in[b][r][c] (f32, contiguous (batch, R, Cc)). */
int gfe_transpose_f32_to_bf16(const float* in, void* out, int64_t batch, int64_t R, int64_t Cc, int64_t out_batch_stride, int64_t ldo, void* stream);

/* The same data laid out for the K/V weight gradient: out[r][b*per_batch_cols + col_off + c] (bf16, row stride ldo) = in[b][r][c]. */
int gfe_interleave_rows_bf16(const float* in, void* out, int64_t B, int64_t R, int64_t Cc, int64_t ldo, int64_t per_batch_cols,
                             int64_t col_off, void* stream);

/* 3x3x3 convolution (zero padding 1) of a ONE-channel volume with per-sample weights, + boundary-class bias table + optional ReLU, and
 * the GroupNorm partials of the result: what ResNetBlock.conv2(GroupNorm(conv1(x))) collapses to when conv1 is a 1x1x1 lift of a
 * single-channel input (pytorch3dunet/unet3d/buildingblocks.py:191-229, first encoder; the algebra is in csrc/unet_ops.hip).
 *   x: (B, D, H, W) f32|bf16; weff: (B, 27, C) f32, tap order (kd, kh, kw) with offsets k - 1; bias_tab: (B, 64, C) f32 indexed by the
 *   voxel's boundary class as in gfe_conv3d_igemm; y: (B, D, H, W, C) bf16; stats_ws: (B, nblk, 2, C) f32 with
 *   nblk = gfe_conv3d_c1_k3_nblk(B, D, H, W), every slot written (8-channel sums like gfe_conv3d_igemm).  C must be 64. */
int gfe_conv3d_c1_k3_nblk(int64_t B, int64_t D, int64_t H, int64_t W);
int gfe_conv3d_c1_k3(const void* x, const float* weff, const float* bias_tab, void* y, float* stats_ws, int64_t stats_nblk,
                     int64_t B, int64_t D, int64_t H, int64_t W, int64_t C, int in_dtype, int relu, void* stream);

/* The first ResNetBlock without its lifted tensor r = conv1(x) (one-channel x):
 *  - gfe_lift_groupnorm_affine: the GroupNorm(G groups) scale / shift (B, C) of r_c = w_c x + b_c from the first two moments of x
 *    (x: (B, S) f32; ws: B * 128 doubles of scratch);
 *  - gfe_conv3d_k3_lift_residual: gfe_conv3d_igemm for a stride-1 27-tap 64-channel conv (per-sample folded weights + bias table as
 *    usual, no statistics) whose residual is r, recomputed in the epilogue from vol (B, D, H, W) f32 and the lift's lift_w / lift_b (64).
 *    pool_out: NULL, or (B, D/2, H/2, W/2, 64) bf16 that receives nn.MaxPool3d(2) of the result (the next encoder's pooling,
 *    buildingblocks.py:284, 306-307) from the epilogue -- needs relu != 0 and even D, H, W; bit-identical to pooling y afterwards. */
int gfe_lift_groupnorm_affine(const float* x, const float* w, const float* bias, const float* gamma, const float* beta,
                              float* scale, float* shift, double* ws, int64_t B, int64_t S, int64_t C, int64_t G, float eps, void* stream);
int gfe_conv3d_k3_lift_residual(const void* x, const void* w_packed, int64_t w_batch_stride, const float* bias_tab, void* y,
                                int64_t B, int64_t D, int64_t H, int64_t W, int64_t Cin, int64_t Cout, const int8_t* tap_offsets, int relu,
                                const float* vol, const float* lift_w, const float* lift_b, void* pool_out, void* stream);

/* gfe_conv3d_igemm for a stride-1 27-tap 64-channel conv (per-sample folded weights + bias table, optional bf16 residual, ReLU) that is
 * followed by the generator's final 1x1x1 conv Cout -> 1 (model.py:123, 165): the 64-channel result is rounded to bf16 as usual but
 * not stored; out_y (B, D, H, W) f32 = sum_c out_w[c] * result_c + out_b. */
int gfe_conv3d_k3_out1(const void* x, const void* w_packed, int64_t w_batch_stride, const float* bias_tab, const void* res,
                       int64_t B, int64_t D, int64_t H, int64_t W, int64_t Cin, int64_t Cout, const int8_t* tap_offsets, int relu,
                       const float* out_w, float out_b, float* out_y, void* stream);

/* ---- input pipeline ------------------------------------------------------------------------------------
 * adaptive_normal (utils/data_normalization.py:20-48, applied per volume at dataloader/pic_table_loader.py:107): lo / hi = the order
 * statistics of the voxels >= 0 at ranks int((m-1)*0.001+0.5) / int((m-1)*0.999+0.5), y = clamp((x - (hi+lo)/2) / ((hi-lo)/2), -1, 1).
 * x, y: (B, n) f32 contiguous (y may alias x); ws: B * gfe_adaptive_normal_ws_words() uint32 of scratch, on return per volume
 * ws[0] = m, ws[5] / ws[6] = bit patterns of lo / hi.  Radix select instead of the reference's full sort; bit-exact.  A volume
 * without a voxel >= 0 (the reference raises IndexError) leaves m = 0 and an unspecified y: callers check ws[0]. */
int gfe_adaptive_normal_ws_words(void);
int gfe_adaptive_normal(const float* x, float* y, uint32_t* ws, int64_t B, int64_t n, void* stream);

/* The loader's Resized(keys=['image'], spatial_size=desired_shape) (dataloader/pic_table_loader.py:58; monai's default mode "area" =
 * torch F.interpolate(mode="area") = adaptive average pooling): B single-channel f32 volumes (B, D, H, W) -> (B, d, h, w);
 * out[i] = mean of in[floor(i D / d) .. ceil((i + 1) D / d)) along every axis. */
int gfe_resize_area(const float* x, float* y, int64_t B, int64_t D, int64_t H, int64_t W, int64_t d, int64_t h, int64_t w, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GFE_HIP_H */
