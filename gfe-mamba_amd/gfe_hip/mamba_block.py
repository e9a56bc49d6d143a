"""One autograd node for a whole Mamba block (cross_atten/mamba.py:197-263: in_proj, chunk, depthwise causal conv + SiLU, x_proj, split,
dt_proj, A = -exp(A_log), selective scan with softplus / D / gate fused, out_proj).

The block's tensors are read and written IN PLACE as column ranges of the two projection outputs -- z = xz[:, ED:], B / C =
dBC[:, R:R+N] / dBC[:, R+N:] going in, dz / dB / dC coming back -- so the `chunk`, `split`, `.contiguous()` and `cat` copies of the op-by-op
graph (36 copies and 12 cats per step in round 2's profile), the -exp(A_log) elementwise chain and its backward, and most zero fills
disappear: 6 launches forward, 11 backward per block, no ATen kernel in between.  Parameter gradients go straight into the flat gradient
buffer's slots (train_ops._grad_slot) when the optimizer owns the parameters."""
import torch

from . import call, dtype_code, ptr, stream
from . import nn_ops as K
from .scan_ops import sscan2_det_ws, sscan2_plan
from .train_ops import _grad_slot, _leaf

F32 = torch.float32


def usable(block, x):
    cfg = block.config
    return (x.is_cuda and x.dtype == F32 and cfg.d_conv == 4 and cfg.d_state == 16 and cfg.d_inner % 32 == 0 and cfg.dt_rank % 4 == 0
            and not cfg.inner_layernorms and block.in_proj.bias is None and block.out_proj.bias is None and block.x_proj.bias is None
            and block.conv1d.bias is not None and block.dt_proj.bias is not None
            and all(p.dtype == F32 for p in (block.in_proj.weight, block.x_proj.weight, block.dt_proj.weight, block.out_proj.weight, block.A_log, block.D)))


def _gemm_into(a, a_t, b, b_t, out):
    """out (strided view allowed) = op(a) op(b)^T, exact f32, no accumulate"""
    M, Kd = (a.shape[1], a.shape[0]) if a_t else a.shape
    N = b.shape[1] if b_t else b.shape[0]
    ld = lambda t: t.stride(0) if t.shape[0] > 1 else t.shape[1]
    call("gfe_gemm_f32", ptr(a), ld(a), int(a_t), ptr(b), ld(b), int(b_t), ptr(out), ld(out), M, N, Kd, None, 0, 1, None, stream())
    return out


class _MambaBlockFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, in_w, conv_w, conv_b, x_w, dt_w, dt_b, A_log, D, out_w, norm_w=None, eps=0.0):
        """norm_w given: x is the residual stream and the node is the whole pre-norm block, mixer(RMSNorm(x)) + x (mamba.py:103): the norm
        kernel also leaves a copy of x that out_proj accumulates into, and the backward's norm kernel adds the residual branch's gradient --
        no add launches either way."""
        Bsz, L, Dm = x.shape
        ED, R, N = conv_w.shape[0], dt_w.shape[1], A_log.shape[1]
        W = R + 2 * N
        dev = x.device
        x2 = x.detach().reshape(Bsz * L, Dm)
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        det = lambda p: p.detach()
        xres = rstd = resid = None
        if norm_w is not None:
            xres = x2
            x2 = torch.empty_like(xres)
            rstd = torch.empty(Bsz * L, dtype=F32, device=dev)
            resid = torch.empty_like(xres)
            call("gfe_rmsnorm_fwd", ptr(xres), ptr(det(norm_w)), ptr(x2), ptr(rstd), ptr(resid), Bsz * L, Dm, float(eps), stream())
        xz = K.gemm_f32(x2, False, det(in_w), False)                                   # (BL, 2 ED) = [xs | z]            mamba.py:204-207
        xc = torch.empty((Bsz * L, ED), dtype=F32, device=dev)
        call("gfe_dwconv1d_silu_fwd", ptr(xz), 2 * ED, ptr(det(conv_w)), ptr(det(conv_b)), ptr(xc), Bsz, L, ED, 4, stream())   # :208-212
        dbc = K.gemm_f32(xc, False, det(x_w), False)                                   # (BL, R + 2N) = [delta_r | B | C]   :235-236
        delta = K.gemm_f32(dbc[:, :R], False, det(dt_w), False)                        # (BL, ED), bias added in the scan    :238
        need_grad = any(ctx.needs_input_grad)
        T, nc = sscan2_plan(Bsz, L, ED, 0)
        sizes = [Bsz * nc * ED * 16 if nc > 1 else 0, Bsz * nc * ED if nc > 1 else 0, Bsz * (-(-L // 32)) * ED * 16 if need_grad else 0]
        hstate = sdelta = ckpt = None
        if sum(sizes):
            hs_, sd_, ck_ = torch.split(torch.empty(sum(sizes), device=dev, dtype=F32), sizes)
            if nc > 1:
                hstate, sdelta = hs_, sd_
            if need_grad:
                ckpt = ck_
        y = torch.empty((Bsz * L, ED), dtype=F32, device=dev)
        yscan = torch.empty_like(y) if need_grad else None
        fcode = dtype_code(F32)
        call("gfe_sscan2_fwd", ptr(xc), ptr(delta), ptr(det(A_log)), ptr(dbc) + 4 * R, ptr(dbc) + 4 * (R + N), ptr(det(D)), ptr(xz) + 4 * ED,
             ptr(det(dt_b)), ptr(y), ptr(yscan), ptr(hstate), ptr(sdelta), ptr(ckpt), Bsz, L, ED, T, 1, fcode, fcode, 2 * ED, W, 1, stream())
        out = K.gemm_f32(y, False, det(out_w), False, accum_into=resid)                # :223 (+ x: resid holds the stream)
        ctx.save_for_backward(x2, xz, xc, dbc, delta, y, yscan, ckpt, sdelta, in_w, conv_w, conv_b, x_w, dt_w, dt_b, A_log, D, out_w, xres, rstd, norm_w)
        ctx.meta = (Bsz, L, Dm, ED, R, N, T, nc, x.dtype)
        return out.view(Bsz, L, Dm)

    @staticmethod
    def backward(ctx, dout):
        x2, xz, xc, dbc, delta, y, yscan, ckpt, sdelta, in_w, conv_w, conv_b, x_w, dt_w, dt_b, A_log, D, out_w, xres, rstd, norm_w = ctx.saved_tensors
        Bsz, L, Dm, ED, R, N, T, nc, xdt = ctx.meta
        W = R + 2 * N
        dev = x2.device
        d2 = dout.reshape(Bsz * L, Dm)
        if d2.dtype != F32 or not d2.is_contiguous():
            d2 = d2.float().contiguous()
        det = lambda p: p.detach()

        def wgrad(param, a, b):            # param.grad (+)= a^T b, both operands read reduction-major
            slot = _grad_slot(param)
            if slot is not None:             # a leaf of the backward chain: on the side stream inside train_ops.side_wgrads()
                _leaf(lambda: K.gemm_f32(a, True, b, True, accum_into=slot.view(slot.shape[0], -1)), slot, a, b)
                return None
            return K.gemm_f32(a, True, b, True).view(param.shape)

        def vec_slot(param, n):            # a zeroed accumulation target for an atomically summed vector gradient, or the optimizer's own slot
            slot = _grad_slot(param)
            return (slot, None) if slot is not None else ((lambda t: (t, t))(torch.zeros(n, dtype=F32, device=dev)))

        # out_proj
        dy = K.gemm_f32(d2, False, det(out_w), True)                                   # (BL, ED)
        g_out = wgrad(out_w, d2, y)
        # scan: dz / dB / dC land in the gradients of the projection outputs they were read from
        dxz = torch.empty((Bsz * L, 2 * ED), dtype=F32, device=dev)                    # [d xs | d z]
        pvec, pbc = sscan2_det_ws(Bsz, L, ED, nc, dev)                                 # fixed-order partial sums (no atomics) at the head's sizes
        # [d delta_r | dB | dC]: with the partial workspaces the scan's fold launch overwrites the dB / dC columns (no zero fill); else f32 atomics
        ddbc = (torch.empty if pbc is not None else torch.zeros)((Bsz * L, W), dtype=F32, device=dev)
        du = torch.empty((Bsz * L, ED), dtype=F32, device=dev)
        dd = torch.empty((Bsz * L, ED), dtype=F32, device=dev)
        (dA_t, g_A), (dD_t, g_D), (db_t, g_dtb) = vec_slot(A_log, ED * N), vec_slot(D, ED), vec_slot(dt_b, ED)
        qstate = torch.empty((Bsz, nc, ED, 16), dtype=F32, device=dev) if nc > 1 else None
        fcode = dtype_code(F32)
        call("gfe_sscan2_bwd", ptr(xc), ptr(delta), ptr(det(A_log)), ptr(dbc) + 4 * R, ptr(dbc) + 4 * (R + N), ptr(det(D)), ptr(xz) + 4 * ED,
             ptr(det(dt_b)), ptr(dy), ptr(yscan), ptr(du), ptr(dd), ptr(dxz) + 4 * ED, ptr(dA_t), ptr(ddbc) + 4 * R, ptr(ddbc) + 4 * (R + N),
             ptr(dD_t), ptr(db_t), ptr(ckpt), ptr(qstate), ptr(sdelta), ptr(pvec), ptr(pbc), Bsz, L, ED, T, 1, fcode, fcode, 2 * ED, W, W, 1, stream())
        # dt_proj (no bias here: it lives inside the scan)
        _gemm_into(dd, False, det(dt_w), True, ddbc[:, :R])                            # d delta_r = d delta . W_dt
        g_dtw = wgrad(dt_w, dd, dbc[:, :R])
        # x_proj: d xc = du + ddbc . W_x
        K.gemm_f32(ddbc, False, det(x_w), True, accum_into=du)
        g_xw = wgrad(x_w, ddbc, xc)
        # depthwise conv + SiLU: d xs goes into the first half of dxz
        sw, sb = _grad_slot(conv_w), _grad_slot(conv_b)
        dcw = sw if sw is not None else torch.zeros_like(conv_w, dtype=F32)
        dcb = sb if sb is not None else torch.zeros_like(conv_b, dtype=F32)
        cws = torch.empty(Bsz * 5 * ED, dtype=F32, device=dev)                          # per-sample partial rows of dw / db
        call("gfe_dwconv1d_silu_bwd", ptr(xz), 2 * ED, ptr(det(conv_w)), ptr(det(conv_b)), ptr(du), ptr(dxz), 2 * ED, ptr(dcw), ptr(dcb), ptr(cws),
             Bsz, L, ED, 4, stream())
        # in_proj
        need_dx = ctx.needs_input_grad[0] or norm_w is not None
        dx = K.gemm_f32(dxz, False, det(in_w), True) if need_dx else None
        g_in = wgrad(in_w, dxz, x2)
        g_norm = None
        if norm_w is not None:
            # through the norm, plus what reaches the stream directly (the residual): dx_stream = rmsnorm_bwd(d xn) + dout, one launch
            nslot = _grad_slot(norm_w)
            dnw = nslot if nslot is not None else torch.zeros_like(norm_w, dtype=F32)
            dstream = torch.empty_like(xres)
            call("gfe_rmsnorm_bwd", ptr(xres), ptr(det(norm_w)), ptr(rstd), ptr(dx), ptr(dstream), ptr(dnw), ptr(d2), Bsz * L, Dm, stream())
            dx = dstream
            g_norm = None if nslot is not None else dnw
        dx = dx.view(Bsz, L, Dm) if (dx is not None and ctx.needs_input_grad[0]) else None
        return (dx, g_in, None if sw is not None else dcw, None if sb is not None else dcb, g_xw, g_dtw,
                None if g_dtb is None else g_dtb, None if g_A is None else g_A.view(ED, N), None if g_D is None else g_D, g_out, g_norm, None)


def mamba_block(block, x):
    """MambaBlock.forward (mamba.py:197-225) as ONE autograd node; `usable(block, x)` must hold."""
    return _MambaBlockFn.apply(x, block.in_proj.weight, block.conv1d.weight, block.conv1d.bias, block.x_proj.weight, block.dt_proj.weight,
                               block.dt_proj.bias, block.A_log, block.D, block.out_proj.weight, None, 0.0)


def residual_mamba_block(block, norm, x):
    """ResidualBlock.forward (mamba.py:97-103), mixer(norm(x)) + x, as ONE autograd node: the RMSNorm, the block above and the residual add
    (whose two launches per layer -- forward add, backward gradient sum -- ride in the norm kernels); `usable(block, x)` must hold and
    norm.weight must be f32."""
    return _MambaBlockFn.apply(x, block.in_proj.weight, block.conv1d.weight, block.conv1d.bias, block.x_proj.weight, block.dt_proj.weight,
                               block.dt_proj.bias, block.A_log, block.D, block.out_proj.weight, norm.weight, norm.eps)
