"""Autograd plumbing for the trainable head: Linear over the MFMA GEMMs, the head Linear over channels-last mid features, the image
condition buffers, and the flat-buffer per-parameter-clip + Adam step (with the DP all-reduce).

Numerics: master weights, residual stream and reductions are f32.  The head's small Linears (Mamba projections, q / out projections,
GEGLU feed-forward, logits: launch-bound sizes) run on the exact-f32 MFMA GEMM, as the reference trains the head in fp32
(classify_mamba.py:69-74); only Linears wider than F32_LINEAR_MAX_K inputs (the cross-attention K / V projections over d_cross =
9 216 .. 25 600 image columns) round their operands to bf16 (f32 accumulate) -- and since round 5 the classifier's one-query call never
forms K or V at all (csrc/xattn_fold.hip), so on the classify path every trainable product is f32.
"""
import weakref

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

from . import call, lib, nn_ops as K, ptr, stream
from .nn_ops import BF16

_SHADOW = {}        # id(parameter) -> bf16 view kept fresh by FlatAdam (avoids a cast per use)
F32_LINEAR_MAX_K = 4096     # Linears with at most this many input features keep f32 operands (gfe_gemm_f32)


def _w16(weight):
    s = _SHADOW.get(id(weight))
    if s is not None and s[2]() is weight and s[1] == weight._version:     # (ids are reused once a parameter is freed)
        return s[0]
    return K.cast(weight.detach().float(), BF16)


def _grad_slot(param):
    """The parameter's gradient buffer if a backward may add into it directly: ONLY for parameters FlatAdam owns (it sets
    `_gfe_flat_grad` on them and hands each a view of its flat gradient buffer, zeroed at the start of the step), so the wgrad kernels
    accumulate in place and autograd's per-parameter AccumulateGrad add/copy launches (~90 per step) disappear.  Any other parameter
    -- whatever its .grad happens to hold -- gets its gradient returned to autograd (AccumulateGrad, hooks and torch.autograd.grad
    behave as usual)."""
    if not getattr(param, "_gfe_flat_grad", False):
        return None
    g = param.grad
    if g is None or g.dtype != torch.float32 or not g.is_contiguous() or g.shape != param.shape or not g.is_cuda:
        return None
    return g


# ---- weight gradients off the backward's critical chain -------------------------------------------------------------------------------
# The head's backward is a chain of ~150 small dependent launches; in the two-stream step each of them waits for a hole between the frozen
# generator's persistent conv launches, so the chain's LENGTH is what the step pays.  A weight gradient dW = dy^T x is a
# leaf of that chain -- nothing in the backward reads it, only the optimizer -- so inside `side_wgrads()` every weight-gradient GEMM whose
# target is one of the optimizer's own gradient slots is enqueued on a second stream (forked behind its operands' producers, joined
# before the optimizer): ~35 launches per step leave the chain and run beside it.  Same kernels, same arguments, same results (each slot
# has one writer; the fixed-order split-K sums are unchanged).  Operands are kept alive until the join, so their memory cannot be handed
# to a later allocation of the main stream while the side stream still reads it; capturable (fork / join inside the captured region).
class _Side:
    on = False
    stream = None
    masked = False           # the stream is confined to a step's head CUs (ClassifyStep._split_streams)
    keep = []


class side_wgrads:
    def __enter__(self):
        if _Side.stream is None:
            _Side.stream = torch.cuda.Stream()
        _Side.on = True
        return self

    def join(self):
        if _Side.keep:
            torch.cuda.current_stream().wait_stream(_Side.stream)
            _Side.keep.clear()

    def __exit__(self, *exc):
        self.join()
        _Side.on = False
        return False


class aux_region:
    """`with aux_region() as r:` runs its body on the side stream, forked behind everything the current stream has been given so far;
    `r.join()` (later, on the forking stream) makes that stream wait for it.  For work at the START of a forward that the main chain needs
    only near its end -- the image condition and its K / V projections (mamba_transformer.py:89-94, sd_cross_atten.py:52-53), which the
    Mamba stack does not read: 8 launches that the chain then does not wait for.  Autograd runs their backward on the same side stream
    (the engine replays a node on its forward's stream and orders the streams itself).  GFE_NO_SIDE_WGRAD=1 keeps everything on one stream."""

    def __init__(self):
        import os
        # not while a HIP graph is being captured: forking a second stream in the captured FORWARD made this ROCm's capture_end segfault
        # (whole-step capture, tests/test_head_gpu.py::test_graphed_step_matches_eager_step; the backward's fork -- side_wgrads -- captures
        # fine).  GFE_AUX_IN_CAPTURE=1 lifts the guard for experiments.
        self.on = (torch.cuda.is_available() and os.environ.get("GFE_NO_SIDE_WGRAD") != "1"
                   and (not torch.cuda.is_current_stream_capturing() or os.environ.get("GFE_AUX_IN_CAPTURE") == "1"))

    def __enter__(self):
        if self.on:
            if _Side.stream is None:
                _Side.stream = torch.cuda.Stream()
            self.main = torch.cuda.current_stream()
            _Side.stream.wait_stream(self.main)
            self.ctx = torch.cuda.stream(_Side.stream)
            self.ctx.__enter__()
        return self

    def __exit__(self, *exc):
        if self.on:
            self.ctx.__exit__(*exc)
        return False

    def join(self):
        if self.on:
            torch.cuda.current_stream().wait_stream(_Side.stream)


def _leaf(fn, slot, *operands):
    """Run fn() -- a weight-gradient launch accumulating into `slot` -- on the side stream when that is safe (see side_wgrads), else here."""
    import os
    # Not inside a HIP-graph capture: a captured fork / join replays SLOWER than the straight chain on this ROCm (B = 8, head graph: 688 vs
    # 708 volumes/s; B = 1: 283 vs 300) -- the eager step is where the side stream pays (706 -> 722).
    if (not _Side.on or slot is None or os.environ.get("GFE_NO_SIDE_WGRAD") == "1"
            or (torch.cuda.is_current_stream_capturing() and os.environ.get("GFE_AUX_IN_CAPTURE") != "1")):
        return fn()
    _Side.stream.wait_stream(torch.cuda.current_stream())        # behind everything that produced the operands (and earlier adds to the slot)
    with torch.cuda.stream(_Side.stream):
        fn()
    _Side.keep.extend(operands)
    return None


def _gemm(a16, b16, bias=None):
    """(M, K) x (N, K)^T -> (M, N) f32; zero-pads K to a multiple of 8 and N to a multiple of 4 (tiny test widths only)."""
    Kd, N = a16.shape[1], b16.shape[0]
    pk, pn = (-Kd) % 8, (-N) % 4
    if pk or pn or a16.stride(0) % 8 or b16.stride(0) % 8 or a16.stride(1) != 1 or b16.stride(1) != 1:
        a16 = F.pad(a16, (0, pk)).contiguous()
        b16 = F.pad(b16, (0, pk, 0, pn)).contiguous()
        if bias is not None and pn:
            bias = F.pad(bias, (0, pn))
    y = K.gemm_nt(a16, b16, bias=bias, out_dtype=torch.float32)
    return y[:, :N] if pn else y


class _LinearFn(torch.autograd.Function):
    """y = x W^T + b.  Optional precomputed bf16 operands: x16 (M, K) and xT16 (K, M) for inputs that are reused.

    Fast path (every Linear of the real model): the GEMM reads f32 activations / gradients in place and reads W or dy / x
    reduction-major for dgrad / wgrad (gfe_gemm_ex), so a Linear costs 1 launch forward and 3 backward.  Shapes the vector
    loads cannot take (K % 8, N % 8: tiny test widths, the 1-wide logit layer) go through cast + transpose + zero padding."""

    @staticmethod
    def forward(ctx, x, weight, bias, x16, xT16, exact=True):
        N, Kd = weight.shape
        xs = x.shape
        if exact and Kd <= F32_LINEAR_MAX_K and x16 is None and weight.dtype == torch.float32:
            # exact-f32 path (the reference's own precision for the head): 1 launch forward, 2-3 backward, no casts
            x32 = x.detach().reshape(-1, Kd)
            if x32.dtype != torch.float32 or x32.stride(-1) != 1:
                x32 = x32.float().contiguous()
            b32 = None if bias is None else bias.detach().float().contiguous()
            y = K.gemm_f32(x32, False, weight.detach(), False, bias=b32)
            ctx.save_for_backward(x32, None, weight)
            ctx.meta = (xs, bias is not None, x.dtype, "f32")
            ctx.bias_ref, ctx.w_ref = bias, weight
            return y.reshape(xs[:-1] + (N,))
        w16 = _w16(weight)
        b32 = None if bias is None else bias.detach().float().contiguous()
        xin = x16 if x16 is not None else x.detach().reshape(-1, Kd)
        if xin.dtype not in (BF16, torch.float32):
            xin = xin.float()
        fast = Kd % 8 == 0 and N % 8 == 0 and K._ex_ok(xin) and K._ex_ok(w16)
        if fast:
            y = K.gemm_ex(xin, False, w16, False, bias=b32)
            ctx.save_for_backward(xin, xT16, weight)
        else:
            if x16 is None:
                x16 = K.cast(x.detach().reshape(-1, Kd).float(), BF16)
            y = _gemm(x16, w16, b32)
            ctx.save_for_backward(x16, xT16, weight)
        ctx.meta = (xs, bias is not None, x.dtype, fast)
        ctx.bias_ref, ctx.w_ref = bias, weight
        return y.reshape(xs[:-1] + (N,))

    @staticmethod
    def backward(ctx, dy):
        xin, xT16, weight = ctx.saved_tensors
        xs, has_bias, xdt, fast = ctx.meta
        N, Kd = weight.shape
        dy2 = dy.reshape(-1, N)
        dx = dw = db = None
        d32 = dy2 if (dy2.dtype == torch.float32 and dy2.stride(1) == 1) else dy2.float().contiguous()
        if fast == "f32":
            if ctx.needs_input_grad[0]:
                dx = K.gemm_f32(d32, False, weight.detach(), True).reshape(xs).to(xdt)    # dy (M, N) . W (N, K) read reduction-major
            if ctx.needs_input_grad[1]:
                slot = _grad_slot(ctx.w_ref)
                dw = _leaf(lambda: K.gemm_f32(d32, True, xin, True, accum_into=slot), slot, d32, xin)   # dy^T . x, both read reduction-major
                dw = None if slot is not None else dw.to(weight.dtype)
        elif fast and K._ex_ok(d32):
            if ctx.needs_input_grad[0]:
                dx = K.gemm_ex(d32, False, _w16(weight), True).reshape(xs).to(xdt)        # dy (M, N) . W (N, K) read reduction-major
            if ctx.needs_input_grad[1]:
                slot = _grad_slot(ctx.w_ref)                                              # accumulate straight into weight.grad
                if xT16 is not None and K._ex_ok(xT16):
                    dw = _leaf(lambda: K.gemm_ex(d32, True, xT16, False, accum_into=slot), slot, d32, xT16)   # dy^T . (K, M)^T
                else:
                    dw = _leaf(lambda: K.gemm_ex(d32, True, xin, True, accum_into=slot), slot, d32, xin)      # dy^T . x, both read reduction-major
                dw = None if slot is not None else dw.to(weight.dtype)
        else:
            x16 = xin if xin.dtype == BF16 else K.cast(xin, BF16)
            dy16 = K.cast(d32, BF16)
            if ctx.needs_input_grad[0]:
                dx = _gemm(dy16, K.transpose_bf16(_w16(weight))).reshape(xs).to(xdt)      # (M, N) x (K, N)^T
            if ctx.needs_input_grad[1]:
                if xT16 is None:
                    xT16 = K.transpose_bf16(x16)                                          # (K, M)
                dw = _gemm(K.transpose_bf16(dy16), xT16).to(weight.dtype)                 # (N, M) x (K, M)^T
        if has_bias and ctx.needs_input_grad[2]:
            slot = _grad_slot(ctx.bias_ref)
            db = slot if slot is not None else torch.empty(N, dtype=torch.float32, device=d32.device)
            nrb = int(lib().gfe_colsum_rblocks(d32.shape[0], N))            # > 1 beyond 4 096 rows: partial rows + an ordered fold (bit-reproducible)
            cws = torch.empty(nrb * N, dtype=torch.float32, device=d32.device) if nrb > 1 else None
            if cws is None:
                _leaf(lambda: call("gfe_colsum_f32", ptr(d32), ptr(db), d32.shape[0], N, d32.stride(0), int(slot is not None), stream()), slot, d32)
            else:
                _leaf(lambda: call("gfe_colsum_f32_ws", ptr(d32), ptr(db), ptr(cws), d32.shape[0], N, d32.stride(0), int(slot is not None), stream()), slot, d32, cws)
            if slot is not None:
                db = None
        return dx, dw, db, None, None, None


def linear(x, weight, bias=None, x16=None, xT16=None, exact=True):
    """exact=False: bf16 matrix-core operands (f32 accumulation) also below F32_LINEAR_MAX_K inputs, where the default keeps the
    reference's own f32 precision (the classifier head's Linears)."""
    if not x.is_cuda:
        raise RuntimeError("gfe_hip linear needs CUDA/HIP tensors (no CPU fallback)")
    return _LinearFn.apply(x, weight, bias, x16, xT16, exact)


class Linear(nn.Linear):
    """nn.Linear whose forward/backward run on the bf16 MFMA GEMM (same parameters / state-dict keys)."""

    def forward(self, x):
        return linear(x, self.weight, self.bias)


class _QkvFlashAttnFn(torch.autograd.Function):
    """to_qkv (Linear, no bias) followed by softmax(q k^T * scale) v per head -- vit_pytorch_diy/vit_3d.py:49-59 -- as ONE autograd node for
    head dim 64 and any token count: 2 launches forward (bf16 MFMA GEMM writing the (B*T, 3*inner) bf16 projection, flash attention with the
    row statistic), 5 backward (gfe_attention_bwd's three + dgrad + wgrad reading the bf16 dq|dk|dv buffer in place).  The T x T score
    matrix never exists in memory either way (the reference materialises it twice: `dots`, `attn`)."""

    @staticmethod
    def forward(ctx, h, weight, heads, scale, p_drop):
        inner3, dim = weight.shape
        inner = inner3 // 3
        dh = inner // heads
        B, T = h.shape[0], h.shape[1]
        h2 = h.detach().reshape(B * T, dim)
        if h2.dtype not in (BF16, torch.float32):
            h2 = h2.float()
        w16 = _w16(weight)
        assert dh == 64 and dim % 8 == 0 and K._ex_ok(h2) and K._ex_ok(w16), "qkv_flash_attention: head dim 64, 16-byte aligned rows"
        qkv = K.gemm_ex(h2, False, w16, False, out_dtype=BF16)
        seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if p_drop > 0 else 0          # host-side draw from torch's CPU generator: no device sync
        o, nlse = K.attention_fwd(qkv[:, :inner], qkv[:, inner:2 * inner], qkv[:, 2 * inner:], B, heads, T, dh, scale, with_lse=True,
                                  dropout_p=p_drop, seed=seed)
        ctx.save_for_backward(h2, qkv, o, nlse, weight)
        ctx.meta = (B, T, heads, dh, float(scale), h.dtype, h.shape, float(p_drop), seed)
        return o.view(B, T, inner)

    @staticmethod
    def backward(ctx, do):
        h2, qkv, o, nlse, weight = ctx.saved_tensors
        B, T, heads, dh, scale, hdt, hs, p_drop, seed = ctx.meta
        inner = heads * dh
        d16 = do.reshape(B * T, inner)
        d16 = d16.contiguous() if d16.dtype == BF16 else K.cast(d16.float(), BF16)
        dqkv = torch.empty_like(qkv)
        K.attention_bwd(qkv[:, :inner], qkv[:, inner:2 * inner], qkv[:, 2 * inner:], o, d16, nlse, B, heads, T, dh, scale, dqkv=dqkv,
                        dropout_p=p_drop, seed=seed)
        dh_ = dw = None
        if ctx.needs_input_grad[0]:
            dh_ = K.gemm_ex(dqkv, False, _w16(weight), True).reshape(hs).to(hdt)          # dqkv (M, 3 inner) . W (3 inner, dim) read reduction-major
        if ctx.needs_input_grad[1]:
            slot = _grad_slot(weight)
            dw = K.gemm_ex(dqkv, True, h2, True, accum_into=slot)                          # dqkv^T . h, both read reduction-major
            dw = None if slot is not None else dw.to(weight.dtype)
        return dh_, dw, None, None, None


def qkv_flash_attention(h, to_qkv_weight, heads, scale, dropout_p=0.0):
    """h: (B, T, dim) f32|bf16 (the normed tokens) -> (B, T, heads*64) bf16 = rearrange(dropout(softmax(q k^T * scale)) v) with q, k, v =
    chunks of h @ to_qkv_weight^T (vit_3d.py:49-59), differentiable in h and the weight; dropout_p: the probabilities' dropout of :56."""
    if not h.is_cuda:
        raise RuntimeError("gfe_hip qkv_flash_attention needs CUDA/HIP tensors (no CPU fallback)")
    return _QkvFlashAttnFn.apply(h, to_qkv_weight, heads, scale, float(dropout_p))


class _MidLinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mid_in, mid_out, weight, bias):
        B, H, W, C = mid_in.shape
        S = weight.shape[0]
        out = torch.empty((B, 2 * C, S), dtype=torch.float32, device=mid_in.device)
        wt = weight.detach().float().t().contiguous()          # (HW, S): one 16-byte load per row in the kernel
        import ctypes
        nch = ctypes.c_int(0)
        call("gfe_mid_linear_plan", B, H * W, ctypes.byref(nch))
        ws = torch.empty(nch.value * out.numel(), dtype=torch.float32, device=mid_in.device)     # one partial slab per block of rows
        call("gfe_mid_linear_fwd", ptr(mid_in), ptr(mid_out), ptr(wt), ptr(bias.detach().float().contiguous()), ptr(out), ptr(ws),
             B, H * W, C, S, stream())
        ctx.save_for_backward(mid_in, mid_out)
        ctx.S = S
        return out

    @staticmethod
    def backward(ctx, dout):
        mid_in, mid_out = ctx.saved_tensors
        B, H, W, C = mid_in.shape
        d = dout.float().contiguous()
        dW = torch.empty((ctx.S, H * W), dtype=torch.float32, device=d.device)
        call("gfe_mid_linear_wgrad", ptr(mid_in), ptr(mid_out), ptr(d), ptr(dW), B, H * W, C, ctx.S, stream())
        return None, None, dW, d.sum((0, 1))


def mid_linear(mid_in_cl, mid_out_cl, weight, bias):
    """mid_*_cl: (B, H, W, C) bf16 channels-last; returns (B, 2C, S) f32 = Linear over (h w) of cat([in, out], dim=1)."""
    return _MidLinearFn.apply(mid_in_cl, mid_out_cl, weight, bias)


class _RMSNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, eps):
        xs = x.shape
        x2 = x.detach().float().reshape(-1, xs[-1]).contiguous()
        w_ = w.detach().float().contiguous()
        y = torch.empty_like(x2)
        rstd = torch.empty(x2.shape[0], dtype=torch.float32, device=x2.device)
        call("gfe_rmsnorm_fwd", ptr(x2), ptr(w_), ptr(y), ptr(rstd), None, x2.shape[0], x2.shape[1], float(eps), stream())
        ctx.save_for_backward(x2, w_, rstd)
        ctx.xs, ctx.w_ref = xs, w
        return y.view(xs)

    @staticmethod
    def backward(ctx, dy):
        x2, w_, rstd = ctx.saved_tensors
        d = dy.float().reshape(x2.shape).contiguous()
        dx = torch.empty_like(x2)
        slot = _grad_slot(ctx.w_ref)
        dw = slot if slot is not None else torch.zeros_like(w_)
        call("gfe_rmsnorm_bwd", ptr(x2), ptr(w_), ptr(rstd), ptr(d), ptr(dx), ptr(dw), None, x2.shape[0], x2.shape[1], stream())
        return dx.view(ctx.xs), (None if slot is not None else dw), None


def rmsnorm(x, weight, eps):
    """x * rsqrt(mean(x^2, -1) + eps) * weight (cross_atten/mamba.py:415-416), one kernel each way."""
    return _RMSNormFn.apply(x, weight, eps)


class _DwConvSiluFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, bias):
        B, L, ED = x.shape
        x_ = x.detach().float().contiguous()
        w_ = w.detach().float().contiguous()
        b_ = None if bias is None else bias.detach().float().contiguous()
        y = torch.empty_like(x_)
        call("gfe_dwconv1d_silu_fwd", ptr(x_), ED, ptr(w_), ptr(b_), ptr(y), B, L, ED, w_.shape[-1], stream())
        ctx.save_for_backward(x_, w_, b_)
        ctx.w_ref, ctx.b_ref = w, bias
        return y

    @staticmethod
    def backward(ctx, dy):
        x_, w_, b_ = ctx.saved_tensors
        B, L, ED = x_.shape
        d = dy.float().contiguous()
        dx = torch.empty_like(x_)
        sw, sb = _grad_slot(ctx.w_ref), (None if b_ is None else _grad_slot(ctx.b_ref))
        dw = sw if sw is not None else torch.zeros_like(w_)
        db = None if b_ is None else (sb if sb is not None else torch.zeros_like(b_))
        ws = torch.empty(B * 5 * ED, dtype=torch.float32, device=d.device)          # per-sample partial rows (summed in sample order)
        call("gfe_dwconv1d_silu_bwd", ptr(x_), ED, ptr(w_), ptr(b_), ptr(d), ptr(dx), ED, ptr(dw), ptr(db), ptr(ws), B, L, ED, w_.shape[-1], stream())
        return dx, (None if sw is not None else dw), (None if sb is not None or b_ is None else db)


def dwconv1d_silu(x, conv_weight, conv_bias):
    """silu(depthwise causal conv1d(x) + bias) on (B, L, ED) (cross_atten/mamba.py:208-212)."""
    return _DwConvSiluFn.apply(x, conv_weight, conv_bias)


class Condition:
    """Image condition of Cross_mamba_both (mamba_transformer.py:89-94: rearrange 'b c h w d -> b (c d) (h w)' of [mri, pet], keys = the
    d-slices of every image, d_cross = h*w features each).

    `images` holds what the one-query cross-attention reads (csrc/xattn_fold.hip): each volume as an f32 (B, h*w, d) matrix whose COLUMNS
    are the keys -- the volume's own memory when it is f32 and contiguous, so building a Condition then launches nothing.
    `cond` (B, keys, d_cross) / `condT` (d_cross, B*keys) are the bf16 copies the MATERIALISED K / V projections read (forward / weight
    gradient layout); they are built on first use only (multi-query calls, a condition that wants its own gradient)."""

    def __init__(self, images):
        B = images[0].shape[0]
        HW = images[0].shape[2] * images[0].shape[3]
        D3 = images[0].shape[4]
        n = len(images)
        self.B, self.keys, self.d_cross = B, n * D3, HW
        self.images = []
        for img in images:
            assert img.shape[1] == 1 and tuple(img.shape) == (B, 1) + tuple(images[0].shape[2:]), "condition images are single-channel volumes of one shape"
            src = img.detach().float().contiguous()
            if src.is_cuda and not torch.cuda.is_current_stream_capturing():
                src.record_stream(torch.cuda.current_stream())          # the condition may be built on a side stream (aux_region); under
                                                                        # capture the inputs are the graph's own static buffers
            self.images.append(src.view(B, HW, D3))
        self._cond = self._condT = None

    def _materialise(self):
        B, n, HW = self.B, len(self.images), self.d_cross
        D3 = self.keys // n
        dev = self.images[0].device
        self._cond = torch.empty((B, n * D3, HW), dtype=BF16, device=dev)
        self._condT = torch.empty((HW, B * n * D3), dtype=BF16, device=dev)
        for i, src in enumerate(self.images):
            call("gfe_transpose_f32_to_bf16", ptr(src), self._cond.data_ptr() + i * D3 * HW * 2, B, HW, D3, n * D3 * HW, HW, stream())
            call("gfe_interleave_rows_bf16", ptr(src), ptr(self._condT), B, HW, D3, B * n * D3, n * D3, i * D3, stream())

    @property
    def cond(self):
        if self._cond is None:
            self._materialise()
        return self._cond

    @property
    def condT(self):
        if self._condT is None:
            self._materialise()
        return self._condT


class FlatAdam:
    """Adam(lr, betas, eps) preceded by clip_grad_norm_(p, max_norm) on EACH parameter (classify_mamba.py:64, 106-108),
    over flat f32 buffers: one all-reduce (DP), one norm pass, one update pass for every tensor."""

    def __init__(self, params, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, max_norm=1.0, chunk=8192):
        self.params = [p for p in params]
        self.epoch = [0]
        dev = self.params[0].device
        sizes = [p.numel() for p in self.params]
        offs = np.concatenate([[0], np.cumsum([(s + 7) // 8 * 8 for s in sizes])])     # 32-B aligned tensors
        total = int(offs[-1])
        self.flat_p = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_m = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_v = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_p16 = torch.zeros(total, dtype=BF16, device=dev)
        rec = []
        for tid, (p, o, s) in enumerate(zip(self.params, offs[:-1], sizes)):
            o = int(o)
            self.flat_p[o:o + s].copy_(p.detach().reshape(-1))
            p.data = self.flat_p[o:o + s].view(p.shape)
            p.grad = self.flat_g[o:o + s].view(p.shape)
            p._gfe_flat_grad = True          # opt-in for the in-place wgrad accumulation (_grad_slot)
            p._gfe_epoch = self.epoch        # shared update counter: caches derived from p (packed conv weights) key on it
            for c0 in range(0, s, chunk):
                rec.append((o + c0, min(chunk, s - c0), tid))
        tab = np.zeros(len(rec), dtype=np.dtype([("off", "<i8"), ("len", "<i4"), ("tid", "<i4")]))
        tab["off"], tab["len"], tab["tid"] = zip(*rec)
        self.chunks = torch.from_numpy(tab.view(np.uint8).copy()).to(dev)
        self.nchunks = len(rec)
        self._chunk_tids = np.asarray(tab["tid"], dtype=np.int64)
        self.buckets = []                                      # set_buckets(n): bucketed all-reduce + update
        self.norm2 = torch.zeros(len(self.params) + self.nchunks, dtype=torch.float32, device=dev)     # per-tensor norms | per-chunk partials
        self.offs, self.sizes = offs, sizes
        self.lr, self.betas, self.eps, self.max_norm = lr, betas, eps, max_norm
        self.t = 0
        self.flat_p16.copy_(self.flat_p)
        self._register_shadows()

    def _register_shadows(self):
        for p, o, s in zip(self.params, self.offs[:-1], self.sizes):
            o = int(o)
            _SHADOW[id(p)] = (self.flat_p16[o:o + s].view(p.shape), p._version, weakref.ref(p, lambda _, k=id(p): _SHADOW.pop(k, None)))

    def zero_grad(self):
        self.wait_updated()
        self.flat_g.zero_()
        for p, o, s in zip(self.params, self.offs[:-1], self.sizes):      # autograd keeps accumulating in place
            if p.grad is None or p.grad.data_ptr() != self.flat_g.data_ptr() + int(o) * 4:
                p.grad = self.flat_g[int(o):int(o) + s].view(p.shape)

    def step(self, world_size=1, group=None, overlap=False):
        """overlap=True runs the all-reduce and the update on a side stream and returns at once: whatever does not read the
        parameters or the gradient buffer (the NEXT step's frozen-generator forward, 85 % of a step) then overlaps the collective.
        Every reader of parameters / gradients must call wait_updated() first (ClassifyStep does)."""
        if overlap:
            if getattr(self, "_side", None) is None:
                self._side = torch.cuda.Stream()
            self._side.wait_stream(torch.cuda.current_stream())          # the backward that filled flat_g
            with torch.cuda.stream(self._side):
                self._step(world_size, group)
                self._done = self._side.record_event()
        else:
            self.wait_updated()
            self._step(world_size, group)

    def wait_updated(self):
        """Make the current stream wait for an update that step(overlap=True) left running on the side stream."""
        ev = getattr(self, "_done", None)
        if ev is not None:
            torch.cuda.current_stream().wait_event(ev)
            self._done = None

    def set_buckets(self, n):
        """Split the update into n buckets of whole tensors with about equal numbers of elements, LAST parameters first (the order in which
        the backward finishes them): bucket k's slice of the flat gradient is all-reduced and then clipped + updated on its own, so that
        the collective of bucket k+1 runs under the update of bucket k (and, with RCCL, as several medium transfers instead of one of
        87 MB).  n = 1: one collective, one update (the default).  The arithmetic per element is the same either way: every tensor's
        norm, clip factor and Adam update see the same summed gradient (tests/test_dp_gloo.py compares the two bit for bit)."""
        n = max(1, min(int(n), len(self.params)))
        tids = self._chunk_tids
        total = float(sum(self.sizes))
        bounds, acc, target = [len(self.params)], 0.0, total / n
        for tid in range(len(self.params) - 1, 0, -1):
            acc += self.sizes[tid]
            if acc >= target * len(bounds) and len(bounds) < n:
                bounds.append(tid)
        bounds.append(0)
        self.buckets = []
        for hi, lo in zip(bounds[:-1], bounds[1:]):                # tensors [lo, hi)
            if hi <= lo:
                continue
            c0, c1 = int(np.searchsorted(tids, lo, "left")), int(np.searchsorted(tids, hi, "left"))
            self.buckets.append((int(self.offs[lo]), int(self.offs[hi]), c0, c1))
        return len(self.buckets)

    def _step(self, world_size, group):
        from .step import all_reduce_, allreduce_grads_, dp_mean_scale
        self.t += 1
        self.epoch[0] += 1
        nt = len(self.params)
        force = getattr(self, "force_collective", False)
        if len(self.buckets) > 1:
            scale = dp_mean_scale(world_size)
            for o0, o1, c0, c1 in self.buckets:
                if world_size > 1 or force:
                    all_reduce_(self.flat_g[o0:o1], group=group)
                # the chunk table is sorted by tensor: a bucket is a contiguous range of it (tensor norms outside the range come out 0 and are not read)
                call("gfe_clip_adam", ptr(self.flat_p), ptr(self.flat_g), ptr(self.flat_m), ptr(self.flat_v), ptr(self.flat_p16),
                     self.chunks.data_ptr() + 16 * c0, c1 - c0, ptr(self.norm2), nt, self.norm2.data_ptr() + 4 * nt, scale, self.max_norm, self.lr,
                     self.betas[0], self.betas[1], self.eps, self.t, stream())
            return
        scale = allreduce_grads_(self.flat_g, world_size, group, force=force)   # SUM over ranks; the mean is folded into grad_scale
        call("gfe_clip_adam", ptr(self.flat_p), ptr(self.flat_g), ptr(self.flat_m), ptr(self.flat_v), ptr(self.flat_p16),
             ptr(self.chunks), self.nchunks, ptr(self.norm2), nt, self.norm2.data_ptr() + 4 * nt, scale, self.max_norm, self.lr,
             self.betas[0], self.betas[1], self.eps, self.t, stream())
        # parameters are rewritten by the kernel (no version bump): the registered bf16 shadows stay current by construction
