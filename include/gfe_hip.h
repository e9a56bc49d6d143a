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

#define GFE_ABI_VERSION 1

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
 *   u, delta, z, y : (B, L, ED) dtype        Bm, Cm : (B, L, N) dtype
 *   A : (ED, N) f32    D, delta_bias : (ED) f32 or NULL    z : NULL => no gate
 *   hstate (B, nchunks, N, ED) f32, sdelta (B, nchunks, ED) f32: workspaces, required when nchunks > 1;
 *   after the call hstate holds the state at the start of every chunk (kept for the backward).
 *   N in {4, 8, 16}.  T = chunk length from gfe_sscan_plan. */
int gfe_selective_scan_fwd(const void* u, const void* delta, const float* A, const void* Bm, const void* Cm,
                           const float* D, const void* z, const float* delta_bias, void* y,
                           float* hstate, float* sdelta,
                           int64_t B, int64_t L, int64_t ED, int64_t N, int T, int delta_softplus,
                           int dtype, void* stream);

/* Adjoint of the above (what autograd derives for mamba.py:265-286 through PScan.backward, pscan.py:188-224).
 *   du, ddelta, dz : (B, L, ED) dtype (ddelta is w.r.t. the raw delta when delta_softplus)
 *   dA_ws (N, ED) f32 [transposed], dB_ws, dC_ws (B, L, N) f32, dD_ws, dbias_ws (ED) f32: zeroed, accumulated atomically
 *   hstate, sdelta: as left by the forward run with the same T; qstate: workspace like hstate.  ED % 64 == 0. */
int gfe_selective_scan_bwd(const void* u, const void* delta, const float* A, const void* Bm, const void* Cm,
                           const float* D, const void* z, const float* delta_bias, const void* dy,
                           void* du, void* ddelta, void* dz,
                           float* dA_ws, float* dB_ws, float* dC_ws, float* dD_ws, float* dbias_ws,
                           const float* hstate, float* qstate, const float* sdelta,
                           int64_t B, int64_t L, int64_t ED, int64_t N, int T, int delta_softplus,
                           int dtype, void* stream);

/* Materialised scan H[t] = A[t]*H[t-1] + X[t] (H[-1] = 0) over dim 1 of (B, L, DN) tensors, DN = D*N flattened.
 * Drop-in for cross_atten/pscan.py:226 `pscan(A, X)` (PScan.forward, pscan.py:151-186); inputs are not modified.
 *   ws: (B, nchunks, 2, DN) f32 workspace when nchunks > 1 (nchunks = ceil(L/T)). */
int gfe_pscan_fwd(const void* A, const void* X, void* H, float* ws,
                  int64_t B, int64_t L, int64_t DN, int T, int dtype, void* stream);

/* PScan.backward (pscan.py:188-224): gX = reverse scan of gH with A shifted left by one; gA[t] = H[t-1]*gX[t], gA[0] = 0. */
int gfe_pscan_bwd(const void* A, const void* H, const void* gH, void* gA, void* gX, float* ws,
                  int64_t B, int64_t L, int64_t DN, int T, int dtype, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GFE_HIP_H */
