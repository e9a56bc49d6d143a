"""Autograd wrappers of the scan entry points (include/gfe_hip.h, Group A).

`selective_scan_tm`  : fused selective scan in the token-major layout MambaBlock already has
`selective_scan_fn`  : the reference's plug-in contract (cross_atten/mamba.py:243-252, channel-major layout)
`pscan`              : drop-in for cross_atten/pscan.py:226
"""
import ctypes
import os

import torch

from . import call, dtype_code, ptr, stream, lib


def sscan_plan(B, L, ED, N, chunk=0, backward=False):
    T, nc = ctypes.c_int(0), ctypes.c_int(0)
    rc = lib().gfe_sscan_plan(B, L, ED, N, int(chunk), int(backward), ctypes.byref(T), ctypes.byref(nc))
    if rc != 0:
        raise ValueError(f"gfe_sscan_plan: unsupported shape B={B} L={L} ED={ED} N={N}")
    return T.value, nc.value


def _common_dtype(*ts):
    ts = [t for t in ts if t is not None]
    return torch.bfloat16 if all(t.dtype == torch.bfloat16 for t in ts) else torch.float32


def _f32c(t):
    return None if t is None else t.detach().to(torch.float32).contiguous()


def sscan2_plan(B, L, ED, chunk=0):
    T, nc = ctypes.c_int(0), ctypes.c_int(0)
    rc = lib().gfe_sscan2_plan(B, L, ED, int(chunk), ctypes.byref(T), ctypes.byref(nc))
    if rc != 0:
        raise ValueError(f"gfe_sscan2_plan: unsupported shape B={B} L={L} ED={ED}")
    return T.value, nc.value


DET_PART_BC_MAX_BYTES = 4 << 20


def sscan2_det_ws(Bsz, L, ED, nc, dev):
    """Workspaces of gfe_sscan2_bwd's atomics-free accumulation (include/gfe_hip.h), or (None, None): the dB / dC partial rows cost 128 B per
    (32-channel group, b, t), so the fixed-order path is the default up to 4 MB of them (the classifier head: 37 tokens, 1.2 MB at 8
    samples -> its gradients are run-to-run identical) and f32 atomics stay the default at benchmark sizes (L = 4096: 16.8 MB per sample).
    GFE_SCAN_DETERMINISTIC=1 / =0 forces either."""
    nbc = Bsz * (ED // 32) * L * 32
    env = os.environ.get("GFE_SCAN_DETERMINISTIC")
    if env == "0" or (env != "1" and 4 * nbc > DET_PART_BC_MAX_BYTES):
        return None, None
    ws = torch.empty(Bsz * nc * ED * 18 + nbc, device=dev, dtype=torch.float32)
    return ws[:Bsz * nc * ED * 18], ws[Bsz * nc * ED * 18:]


def _al16(t):
    """The single-pass kernels move rows as 8/16-byte vectors: a contiguous view at an odd storage offset is copied once."""
    return t if t is None or t.data_ptr() % 16 == 0 else t.clone()


class _SelectiveScan2(torch.autograd.Function):
    """N = 16: csrc/sscan2.hip (a lane owns one channel x one pair of states; chunked along L only for small batches)."""

    @staticmethod
    def forward(ctx, u, delta, A, Bm, Cm, D, z, delta_bias, delta_softplus, chunk, need_grad):
        Bsz, L, ED = u.shape
        dt = _common_dtype(u, delta, Bm, Cm, z)
        cast = lambda t: None if t is None else _al16(t.detach().to(dt).contiguous())
        u_, d_, z_ = cast(u), cast(delta), cast(z)
        bcdt = torch.bfloat16 if Bm.dtype == torch.bfloat16 and Cm.dtype == torch.bfloat16 else torch.float32
        B_, C_ = _al16(Bm.detach().to(bcdt).contiguous()), _al16(Cm.detach().to(bcdt).contiguous())     # rows are read in their own dtype: no cast kernels
        A_, D_, b_ = _f32c(A), _f32c(D), _f32c(delta_bias)
        T, nc = sscan2_plan(Bsz, L, ED, chunk)
        dev = u.device
        hstate = sdelta = ckpt = None
        # one allocation for every f32 workspace of this call (chunk states, chunk sums of dt, segment checkpoints)
        sizes = [Bsz * nc * ED * 16 if nc > 1 else 0, Bsz * nc * ED if nc > 1 else 0, Bsz * (-(-L // 32)) * ED * 16 if need_grad else 0]
        if sum(sizes):
            ws = torch.empty(sum(sizes), device=dev, dtype=torch.float32)
            hs_, sd_, ck_ = torch.split(ws, sizes)
            if nc > 1:
                hstate, sdelta = hs_, sd_
            if need_grad:
                ckpt = ck_
        y = torch.empty((Bsz, L, ED), device=dev, dtype=dt)
        yscan = torch.empty_like(y) if (need_grad and z_ is not None) else None      # the pre-gate output, for dz in the backward
        call("gfe_sscan2_fwd", ptr(u_), ptr(d_), ptr(A_), ptr(B_), ptr(C_), ptr(D_), ptr(z_), ptr(b_), ptr(y), ptr(yscan),
             ptr(hstate), ptr(sdelta), ptr(ckpt), Bsz, L, ED, T, int(bool(delta_softplus)), dtype_code(dt), dtype_code(bcdt), 0, 0, 0, stream())
        ctx.save_for_backward(u_, d_, A_, B_, C_, D_, z_, b_, ckpt, sdelta, yscan)
        ctx.meta = (T, nc, bool(delta_softplus), dt,
                    tuple(None if t is None else t.dtype for t in (u, delta, A, Bm, Cm, D, z, delta_bias)))
        return y

    @staticmethod
    def backward(ctx, dy):
        u_, d_, A_, B_, C_, D_, z_, b_, ckpt, sdelta, yscan = ctx.saved_tensors
        T, nc, softplus, dt, in_dtypes = ctx.meta
        Bsz, L, ED = u_.shape
        dev = u_.device
        dy_ = _al16(dy.to(dt).contiguous())
        du = torch.empty_like(u_)
        dd = torch.empty_like(u_)
        dz = torch.empty_like(u_) if z_ is not None else None
        sizes = [ED * 16, Bsz * L * 16, Bsz * L * 16, ED, ED]                   # one zeroed f32 slab for every accumulated gradient
        slab = torch.zeros(sum(sizes), device=dev, dtype=torch.float32)
        dA_ws, dB_ws, dC_ws, dD_ws, db_ws = torch.split(slab, sizes)
        qstate = torch.empty((Bsz, nc, ED, 16), device=dev, dtype=torch.float32) if nc > 1 else None
        pvec, pbc = sscan2_det_ws(Bsz, L, ED, nc, dev)                          # fixed-order sums instead of atomics where that is cheap
        call("gfe_sscan2_bwd", ptr(u_), ptr(d_), ptr(A_), ptr(B_), ptr(C_), ptr(D_), ptr(z_), ptr(b_), ptr(dy_), ptr(yscan),
             ptr(du), ptr(dd), ptr(dz), ptr(dA_ws), ptr(dB_ws), ptr(dC_ws),
             ptr(dD_ws) if D_ is not None else None, ptr(db_ws) if b_ is not None else None,
             ptr(ckpt), ptr(qstate), ptr(sdelta), ptr(pvec), ptr(pbc), Bsz, L, ED, T, int(softplus), dtype_code(dt), dtype_code(B_.dtype), 0, 0, 0, 0, stream())
        to = lambda g, i: None if in_dtypes[i] is None else g.to(in_dtypes[i])
        if in_dtypes[3] is not None and in_dtypes[3] == in_dtypes[4] and in_dtypes[3] != torch.float32:
            # dB and dC sit next to each other in the slab: one cast launch for both (the B = 1 step of config 2 is 135 us: every launch shows)
            bc = slab[ED * 16:ED * 16 + 2 * Bsz * L * 16].to(in_dtypes[3])
            dB_ws, dC_ws = bc[:Bsz * L * 16], bc[Bsz * L * 16:]
        return (to(du, 0), to(dd, 1), to(dA_ws.view(ED, 16), 2), to(dB_ws.view(Bsz, L, 16), 3), to(dC_ws.view(Bsz, L, 16), 4),
                to(dD_ws, 5) if D_ is not None else None, to(dz, 6) if z_ is not None else None,
                to(db_ws, 7) if b_ is not None else None, None, None, None)


class _SelectiveScanTM(torch.autograd.Function):
    @staticmethod
    def forward(ctx, u, delta, A, Bm, Cm, D, z, delta_bias, delta_softplus, chunk, need_grad):
        Bsz, L, ED = u.shape
        N = A.shape[1]
        dt = _common_dtype(u, delta, Bm, Cm, z)
        cast = lambda t: None if t is None else t.detach().to(dt).contiguous()
        u_, d_, z_ = cast(u), cast(delta), cast(z)
        B_, C_ = _f32c(Bm), _f32c(Cm)            # the kernels take B / C rows in f32 (tiny tensors)
        A_, D_, b_ = _f32c(A), _f32c(D), _f32c(delta_bias)
        T, nc = sscan_plan(Bsz, L, ED, N, chunk, backward=need_grad)
        hstate = sdelta = None
        if nc > 1:
            hstate = torch.empty((Bsz, nc, N, ED), device=u.device, dtype=torch.float32)
            sdelta = torch.empty((Bsz, nc, ED), device=u.device, dtype=torch.float32)
        y = torch.empty((Bsz, L, ED), device=u.device, dtype=dt)
        call("gfe_selective_scan_fwd", ptr(u_), ptr(d_), ptr(A_), ptr(B_), ptr(C_), ptr(D_), ptr(z_), ptr(b_), ptr(y),
             ptr(hstate), ptr(sdelta), Bsz, L, ED, N, T, int(bool(delta_softplus)), dtype_code(dt), stream())
        ctx.save_for_backward(u_, d_, A_, B_, C_, D_, z_, b_, hstate, sdelta)
        ctx.meta = (T, nc, bool(delta_softplus), dt,
                    tuple(None if t is None else t.dtype for t in (u, delta, A, Bm, Cm, D, z, delta_bias)))
        return y

    @staticmethod
    def backward(ctx, dy):
        u_, d_, A_, B_, C_, D_, z_, b_, hstate, sdelta = ctx.saved_tensors
        T, nc, softplus, dt, in_dtypes = ctx.meta
        Bsz, L, ED = u_.shape
        N = A_.shape[1]
        dev = u_.device
        dy_ = dy.to(dt).contiguous()
        du = torch.empty_like(u_)
        dd = torch.empty_like(u_)
        dz = torch.empty_like(u_) if z_ is not None else None
        # one zeroed f32 slab for every atomically accumulated gradient
        sizes = [N * ED, Bsz * L * N, Bsz * L * N, ED, ED]
        slab = torch.zeros(sum(sizes), device=dev, dtype=torch.float32)
        dA_ws, dB_ws, dC_ws, dD_ws, db_ws = torch.split(slab, sizes)
        qstate = torch.empty_like(hstate) if nc > 1 else None
        call("gfe_selective_scan_bwd", ptr(u_), ptr(d_), ptr(A_), ptr(B_), ptr(C_), ptr(D_), ptr(z_), ptr(b_), ptr(dy_),
             ptr(du), ptr(dd), ptr(dz), ptr(dA_ws), ptr(dB_ws), ptr(dC_ws),
             ptr(dD_ws) if D_ is not None else None, ptr(db_ws) if b_ is not None else None,
             ptr(hstate), ptr(qstate), ptr(sdelta), Bsz, L, ED, N, T, int(softplus), dtype_code(dt), stream())
        to = lambda g, i: None if in_dtypes[i] is None else g.to(in_dtypes[i])
        dA = dA_ws.view(N, ED).t()
        return (to(du, 0), to(dd, 1), to(dA, 2), to(dB_ws.view(Bsz, L, N), 3), to(dC_ws.view(Bsz, L, N), 4),
                to(dD_ws, 5) if D_ is not None else None, to(dz, 6) if z_ is not None else None,
                to(db_ws, 7) if b_ is not None else None, None, None, None)


def selective_scan_tm(u, delta, A, Bm, Cm, D=None, z=None, delta_bias=None, delta_softplus=False, chunk=0):
    """Token-major fused selective scan.  u, delta, z: (B, L, ED); Bm, Cm: (B, L, N); A: (ED, N); returns (B, L, ED).

    y = selective_scan(u, softplus(delta + delta_bias), A, B, C, D) * silu(z)   (cross_atten/mamba.py:254-259, 220-222)
    """
    if not u.is_cuda:
        raise RuntimeError("gfe_hip selective scan needs CUDA/HIP tensors (no CPU fallback)")
    # the backward needs chunk-start states every <= 32 steps (LDS checkpoints); a forward that autograd does not record
    # (inference, torch.no_grad) uses coarse chunks instead and saves the state traffic
    need_grad = torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in (u, delta, A, Bm, Cm, D, z, delta_bias))
    if A.shape[1] == 16 and u.shape[2] % 32 == 0 and os.environ.get("GFE_SSCAN_LEGACY") != "1":
        return _SelectiveScan2.apply(u, delta, A, Bm, Cm, D, z, delta_bias, delta_softplus, chunk, need_grad)
    return _SelectiveScanTM.apply(u, delta, A, Bm, Cm, D, z, delta_bias, delta_softplus, chunk, need_grad)


def selective_scan_fn(u, delta, A, B, C, D=None, z=None, delta_bias=None, delta_softplus=False):
    """The reference's plug-in slot `MambaBlock.selective_scan_cuda` (cross_atten/mamba.py:180-186, 243-252).

    u, delta, z: (B, ED, L); B, C: (B, N, L); A: (ED, N); D, delta_bias: (ED).  Returns (B, ED, L).
    The reference hands over transposed views of token-major tensors (mamba.py:245-248), so the
    transposes below are free for u/B/C/z; only `delta` (computed channel-major at mamba.py:238) is copied.
    """
    tm = lambda t: None if t is None else t.transpose(1, 2)
    y = selective_scan_tm(tm(u), tm(delta), A, tm(B), tm(C), D, tm(z), delta_bias, delta_softplus)
    return y.transpose(1, 2)


class _PScan(torch.autograd.Function):
    @staticmethod
    def forward(ctx, A_in, X_in, chunk):
        Bsz, L, D, N = X_in.shape
        dt = _common_dtype(A_in, X_in)
        A = A_in.detach().to(dt).contiguous()
        X = X_in.detach().to(dt).contiguous()
        DN = D * N
        T, nc = _pscan_plan(Bsz, L, DN, dt, chunk)
        ws = torch.empty((Bsz, nc, 2, DN), device=A.device, dtype=torch.float32) if nc > 1 else None
        H = torch.empty_like(X)
        call("gfe_pscan_fwd", ptr(A), ptr(X), ptr(H), ptr(ws), Bsz, L, DN, T, dtype_code(dt), stream())
        ctx.save_for_backward(A, H)
        ctx.meta = (T, nc, dt, A_in.dtype, X_in.dtype)
        return H

    @staticmethod
    def backward(ctx, gH):
        A, H = ctx.saved_tensors
        T, nc, dt, adt, xdt = ctx.meta
        Bsz, L, D, N = H.shape
        DN = D * N
        g = gH.to(dt).contiguous()
        gA = torch.empty_like(H)
        gX = torch.empty_like(H)
        ws = torch.empty((Bsz, nc, 2, DN), device=A.device, dtype=torch.float32) if nc > 1 else None
        call("gfe_pscan_bwd", ptr(A), ptr(H), ptr(g), ptr(gA), ptr(gX), ptr(ws), Bsz, L, DN, T, dtype_code(dt), stream())
        return gA.to(adt), gX.to(xdt), None


def _pscan_plan(B, L, DN, dt, chunk):
    if chunk and chunk > 0:
        T = min(int(chunk), L)
    else:
        vec = 8 if dt == torch.bfloat16 else 4
        waves = B * max(1, DN // (vec * 64))
        want = -(-2048 // waves)
        T = L if want <= 1 else max(32, -(-L // want))
        T = min(T, L)
    return T, -(-L // T)


def pscan(A_in, X_in, chunk=0):
    """H[t] = A[t] * H[t-1] + X[t] over dim 1 of (B, L, D, N) tensors (cross_atten/pscan.py:151-186); differentiable in both."""
    if not X_in.is_cuda:
        raise RuntimeError("gfe_hip pscan needs CUDA/HIP tensors (no CPU fallback)")
    return _PScan.apply(A_in, X_in, chunk)
