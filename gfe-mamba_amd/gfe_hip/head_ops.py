"""Autograd wrappers of the head's small operators (include/gfe_hip.h, csrc/head_ops.hip): token embedding + concat, mean over tokens,
one-query cross attention, LayerNorm over rows, GEGLU + dropout, BCE(sigmoid).  f32, one launch per operator and direction."""
import torch

from . import call, dtype_code, lib, ptr, stream
from .train_ops import _grad_slot, _leaf


def call_int(name, *args):
    """entry points that return a count (>= 0) instead of a status"""
    return int(getattr(lib(), name)(*args))


def _f(t):
    return None if t is None else t.detach().float().contiguous()


def _need_cuda(t, what):
    if not t.is_cuda:
        raise RuntimeError(f"gfe_hip {what} needs CUDA/HIP tensors (no CPU fallback)")


def _acc_target(param):
    """(buffer to accumulate into, True if it is the parameter's own gradient slot)."""
    slot = _grad_slot(param)
    if slot is not None:
        return slot, True
    return torch.zeros(param.shape, dtype=torch.float32, device=param.device), False


class _EmbedTokens(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x_cat, offsets, emb, x_num, num_w, num_b, cls, feat):
        B = next(t.shape[0] for t in (x_cat, x_num, feat) if t is not None)
        ncat = 0 if x_cat is None else x_cat.shape[1]
        ncont = 0 if x_num is None else x_num.shape[1]
        nf = 0 if feat is None else feat.shape[1]
        dim = cls.shape[-1]
        xc = None if x_cat is None else x_cat.detach().to(torch.int64).contiguous()
        off = None if offsets is None else offsets.detach().to(torch.int64).contiguous()
        xn, ft = _f(x_num), _f(feat)
        out = torch.empty((B, 1 + ncat + ncont + nf, dim), dtype=torch.float32, device=cls.device)
        call("gfe_embed_tokens_fwd", ptr(xc), ptr(off), ptr(_f(emb)), ptr(xn), ptr(_f(num_w)), ptr(_f(num_b)), ptr(_f(cls)), ptr(ft), ptr(out),
             B, ncat, ncont, nf, dim, 0 if emb is None else emb.shape[0], stream())
        ctx.save_for_backward(xc, off, xn)
        ctx.refs = (emb, num_w, num_b, cls)
        ctx.meta = (B, ncat, ncont, nf, dim, feat is not None and feat.requires_grad)
        return out

    @staticmethod
    def backward(ctx, dout):
        xc, off, xn = ctx.saved_tensors
        emb, num_w, num_b, cls = ctx.refs
        B, ncat, ncont, nf, dim, want_feat = ctx.meta
        d = dout.float().contiguous()
        tg = [(_acc_target(p) if p is not None else (None, True)) for p in (emb, num_w, num_b, cls)]
        dfeat = torch.empty((B, nf, dim), dtype=torch.float32, device=d.device) if (nf and want_feat) else None
        call("gfe_embed_tokens_bwd", ptr(d), ptr(xc), ptr(off), ptr(xn), ptr(tg[0][0]), ptr(tg[1][0]), ptr(tg[2][0]), ptr(tg[3][0]), ptr(dfeat),
             B, ncat, ncont, nf, dim, 0 if emb is None else emb.shape[0], stream())
        g = [None if own else buf.view(p.shape) for (buf, own), p in zip(tg, (emb, num_w, num_b, cls))]
        return None, None, g[0], None, g[1], g[2], g[3], dfeat


def embed_tokens(x_cat, offsets, emb, x_num, num_w, num_b, cls, feat):
    """[cls | Embedding(x_cat + offsets) | x_num * w + b | feat] as one (B, L, dim) f32 tensor (mamba_transformer.py:97-117)."""
    _need_cuda(cls, "embed_tokens")
    return _EmbedTokens.apply(x_cat, offsets, emb, x_num, num_w, num_b, cls, feat)


class _MeanTokens(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        B, L, dim = x.shape
        x_ = _f(x)
        y = torch.empty((B, 1, dim), dtype=torch.float32, device=x.device)
        call("gfe_mean_tokens_fwd", ptr(x_), ptr(y), B, L, dim, stream())
        ctx.shape = (B, L, dim)
        return y

    @staticmethod
    def backward(ctx, dy):
        B, L, dim = ctx.shape
        dx = torch.empty((B, L, dim), dtype=torch.float32, device=dy.device)
        call("gfe_mean_tokens_bwd", ptr(dy.float().contiguous()), ptr(dx), B, L, dim, stream())
        return dx


def mean_tokens(x):
    """torch.mean(x, dim=1, keepdims=True) (mamba_transformer.py:122)."""
    _need_cuda(x, "mean_tokens")
    return _MeanTokens.apply(x)


class _CrossAttnQ1(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v, n_heads):
        B, nk, dim = k.shape
        dh = dim // n_heads
        q_, k_, v_ = _f(q).view(B, dim), _f(k), _f(v)
        out = torch.empty((B, dim), dtype=torch.float32, device=q.device)
        probs = torch.empty((B, n_heads, nk), dtype=torch.float32, device=q.device)
        scale = dh ** -0.5
        call("gfe_cross_attn_q1_fwd", ptr(q_), ptr(k_), ptr(v_), ptr(out), ptr(probs), B, n_heads, nk, dh, scale, stream())
        ctx.save_for_backward(q_, k_, v_, probs)
        ctx.meta = (B, n_heads, nk, dh, scale, q.shape)
        return out.view(q.shape)

    @staticmethod
    def backward(ctx, dout):
        q_, k_, v_, probs = ctx.saved_tensors
        B, H, nk, dh, scale, qshape = ctx.meta
        d = dout.float().contiguous()
        dq, dk, dv = torch.empty_like(q_), torch.empty_like(k_), torch.empty_like(v_)
        call("gfe_cross_attn_q1_bwd", ptr(q_), ptr(k_), ptr(v_), ptr(probs), ptr(d), ptr(dq), ptr(dk), ptr(dv), B, H, nk, dh, scale, stream())
        return dq.view(qshape), dk, dv, None


def cross_attn_q1(q, k, v, n_heads):
    """softmax(q k^T / sqrt(d_head)) v per head with ONE query per sample: q (B, 1, dim), k / v (B, nk, dim) -> (B, 1, dim)
    (sd_cross_atten.py:58-68 between the projections)."""
    _need_cuda(q, "cross_attn_q1")
    assert q.shape[1] == 1
    return _CrossAttnQ1.apply(q, k, v, n_heads)


class _CrossAttnQ1Folded(torch.autograd.Function):
    """q_proj's output -> attention output (before out_proj) of CrossAttention with one query per sample, K / V projections folded away
    (csrc/xattn_fold.hip): differentiable in q, k_proj.weight, v_proj.weight, v_proj.bias; the condition images carry no gradient (the
    generator is frozen, classify_mamba.py:100); k_proj.bias gets an exactly-zero gradient (it cannot move a softmax)."""

    @staticmethod
    def forward(ctx, q, wk, bk, wv, bv, n_heads, *imgs):
        qs = q.shape
        E, HW = wk.shape
        q_ = _f(q).reshape(-1, E)
        B, H, dh = q_.shape[0], n_heads, E // n_heads
        ims = [im.detach() for im in imgs]
        D3 = ims[0].shape[-1]
        assert all(im.dtype == torch.float32 and im.is_contiguous() and im.numel() == B * HW * D3 for im in ims) and 1 <= len(ims) <= 4
        keys = len(ims) * D3
        dev = q_.device
        wk_, wv_, bv_ = wk.detach(), wv.detach(), (None if bv is None else _f(bv))
        assert wk_.dtype == torch.float32 and wv_.dtype == torch.float32 and wk_.is_contiguous() and wv_.is_contiguous()
        nch = call_int("gfe_cross_attn_q1_folded_chunks", HW)
        r_ws = torch.empty((B, H, HW), dtype=torch.float32, device=dev)
        part = torch.empty(B * len(ims) * nch * H * D3, dtype=torch.float32, device=dev)
        p = torch.empty((B, H, keys), dtype=torch.float32, device=dev)
        c = torch.empty((B, H, HW), dtype=torch.float32, device=dev)
        o = torch.empty((B, E), dtype=torch.float32, device=dev)
        ip = [ptr(im) for im in ims] + [None] * (4 - len(ims))
        call("gfe_cross_attn_q1_folded_fwd", ptr(q_), ptr(wk_), ptr(wv_), ptr(bv_), *ip, len(ims), ptr(r_ws), ptr(part), ptr(p), ptr(c), ptr(o),
             B, H, dh, HW, D3, stream())
        ctx.save_for_backward(q_, p, c, *ims)
        ctx.refs = (wk, bk, wv, bv)
        ctx.meta = (B, H, dh, HW, D3, qs, r_ws, part)         # the two workspaces are reused by the backward (dc, partials)
        return o.view(qs)

    @staticmethod
    def backward(ctx, dout):
        q_, p, c, *ims = ctx.saved_tensors
        wk, bk, wv, bv = ctx.refs
        B, H, dh, HW, D3, qs, dc_ws, part = ctx.meta
        E = H * dh
        d = dout.float().reshape(B, E).contiguous()
        dev = d.device
        ds_ws = torch.empty_like(p)
        dr = torch.empty_like(c)
        dq = torch.empty((B, E), dtype=torch.float32, device=dev)
        ip = [ptr(im) for im in ims] + [None] * (4 - len(ims))
        call("gfe_cross_attn_q1_folded_bwd", ptr(d), ptr(wk.detach()), ptr(wv.detach()), *ip, len(ims), ptr(p),
             ptr(dc_ws), ptr(part), ptr(ds_ws), ptr(dr), ptr(dq), B, H, dh, HW, D3, stream())
        if not (ctx.needs_input_grad[1] or ctx.needs_input_grad[3] or (bv is not None and ctx.needs_input_grad[4])):
            return (dq.view(qs), None, None, None, None, None) + (None,) * len(ims)      # frozen k_proj / v_proj: no rank-update launches
        # the weight gradients are leaves of the backward: on the side stream when they go straight into the optimizer's slots (train_ops._leaf)
        (dwk, own_k), (dwv, own_v) = _acc_target(wk), _acc_target(wv)
        dbv, own_b = (None, True) if bv is None else _acc_target(bv)
        fn = lambda: call("gfe_cross_attn_q1_folded_wgrad", ptr(d), ptr(q_), ptr(c), ptr(dr), ptr(dwk), ptr(dwv), ptr(dbv), B, H, dh, HW, stream())
        if own_k and own_v and own_b:
            _leaf(fn, dwk, d, q_, c, dr)
        else:
            fn()
        gbk = None
        if bk is not None and _grad_slot(bk) is None:
            gbk = torch.zeros_like(bk)                         # exactly zero; a FlatAdam slot simply stays as zero_grad left it
        return (dq.view(qs), None if own_k else dwk, gbk, None if own_v else dwv, None if (own_b or bv is None) else dbv, None) + (None,) * len(ims)


def cross_attn_q1_folded(q, k_weight, k_bias, v_weight, v_bias, n_heads, images):
    """softmax(q (W_k y + b_k)^T / sqrt(d_head)) (W_v y + b_v) per head for ONE query per sample, q (B, 1, E) -> (B, 1, E), without ever
    forming K or V (sd_cross_atten.py:49-70 between q_proj and out_proj).  images: list of f32 contiguous (B, HW, D3) tensors, key
    i*D3 + j = images[i][b, :, j] -- the condition volumes (B, 1, h, w, d) of mamba_transformer.py:89-94 read in place."""
    _need_cuda(q, "cross_attn_q1_folded")
    return _CrossAttnQ1Folded.apply(q, k_weight, k_bias, v_weight, v_bias, n_heads, *images)


class _SdpaSmall(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v, n_heads, causal, p_drop):
        B, L, dim = q.shape
        dh = dim // n_heads
        q_, k_, v_ = _f(q), _f(k), _f(v)
        out = torch.empty_like(q_)
        probs = torch.empty((B, n_heads, L, L), dtype=torch.float32, device=q.device)
        scale = dh ** -0.5
        seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if p_drop > 0 else 0       # host-side draw from torch's CPU generator: no device sync
        call("gfe_sdpa_small_fwd", ptr(q_), ptr(k_), ptr(v_), ptr(out), ptr(probs), B, n_heads, L, dh, scale, int(bool(causal)), float(p_drop), seed, stream())
        ctx.save_for_backward(q_, k_, v_, probs)
        ctx.meta = (B, n_heads, L, dh, scale, float(p_drop), seed)
        return out

    @staticmethod
    def backward(ctx, dout):
        q_, k_, v_, probs = ctx.saved_tensors
        B, H, L, dh, scale, p_drop, seed = ctx.meta
        d = dout.float().contiguous()
        dq, dk, dv = torch.empty_like(q_), torch.empty_like(k_), torch.empty_like(v_)
        call("gfe_sdpa_small_bwd", ptr(q_), ptr(k_), ptr(v_), ptr(probs), ptr(d), ptr(dq), ptr(dk), ptr(dv), B, H, L, dh, scale, p_drop, seed, stream())
        return dq, dk, dv, None, None, None


def sdpa_small(q, k, v, n_heads, causal=True, dropout_p=0.0):
    """F.scaled_dot_product_attention over (B, L, H*dh) projections with equal query / key-value head counts, L <= 64, head dim <= 64
    (Jamba's attention layer at the classifier's 37 tokens: cross_atten/jamba.py:385-392; the generator's ViT in training, where
    dropout_p is the attention-probability dropout of vit_pytorch_diy/vit.py:59)."""
    _need_cuda(q, "sdpa_small")
    return _SdpaSmall.apply(q, k, v, n_heads, causal, float(dropout_p))


class _LayerNormRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        xs = x.shape
        x2 = _f(x).reshape(-1, xs[-1])
        g_, b_ = _f(gamma), _f(beta)
        rows, dim = x2.shape
        y = torch.empty_like(x2)
        st = torch.empty((2, rows), dtype=torch.float32, device=x.device)
        ws = torch.empty(rows * 128, dtype=torch.float32, device=x.device) if dim >= 16384 else None     # long rows: segment partials
        call("gfe_layernorm_rows_fwd", ptr(x2), ptr(g_), ptr(b_), ptr(y), ptr(st[0]), ptr(st[1]), ptr(ws), rows, dim, float(eps), stream())
        ctx.save_for_backward(x2, g_, st)
        ctx.refs = (gamma, beta)
        ctx.xs = xs
        return y.view(xs)

    @staticmethod
    def backward(ctx, dy):
        x2, g_, st = ctx.saved_tensors
        gamma, beta = ctx.refs
        rows, dim = x2.shape
        d = dy.float().reshape(rows, dim).contiguous()
        dx = torch.empty_like(x2)
        (dg, own_g), (db, own_b) = _acc_target(gamma), _acc_target(beta)
        ws = torch.empty(rows * 128, dtype=torch.float32, device=x2.device) if dim >= 16384 else None
        if rows >= 256 and dim <= 2048 and dim % 4 == 0:
            ws = torch.empty(2 * dim * 1024, dtype=torch.float32, device=x2.device)             # many rows: per-block partial rows, no atomics
        call("gfe_layernorm_rows_bwd", ptr(x2), ptr(g_), ptr(st[0]), ptr(st[1]), ptr(d), ptr(dx), ptr(dg), ptr(db), ptr(ws), rows, dim, stream())
        return dx.view(ctx.xs), (None if own_g else dg), (None if own_b else db), None


def layernorm_rows(x, gamma, beta, eps=1e-5):
    _need_cuda(x, "layernorm_rows")
    return _LayerNormRows.apply(x, gamma, beta, eps)


class LayerNorm(torch.nn.LayerNorm):
    """nn.LayerNorm over the last dimension (same parameters / state-dict keys), one launch each way."""

    def forward(self, x):
        return layernorm_rows(x, self.weight, self.bias, self.eps)


class _Geglu(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, p_drop, seed, step_ctr):
        xs = x.shape
        F2 = xs[-1]
        x2 = _f(x).reshape(-1, F2)
        y = torch.empty((x2.shape[0], F2 // 2), dtype=torch.float32, device=x.device)
        call("gfe_geglu_fwd", ptr(x2), ptr(y), x2.shape[0], F2 // 2, float(p_drop), int(seed), ptr(step_ctr), stream())
        ctx.save_for_backward(x2)
        ctx.meta = (xs, float(p_drop), int(seed), step_ctr)
        return y.view(xs[:-1] + (F2 // 2,))

    @staticmethod
    def backward(ctx, dy):
        (x2,) = ctx.saved_tensors
        xs, p_drop, seed, step_ctr = ctx.meta
        d = dy.float().reshape(x2.shape[0], -1).contiguous()
        dx = torch.empty_like(x2)
        call("gfe_geglu_bwd", ptr(x2), ptr(d), ptr(dx), x2.shape[0], x2.shape[1] // 2, p_drop, seed, ptr(step_ctr), stream())
        return dx.view(xs), None, None, None


_DROP_CALLS = [0]
_STEP_CTR = [None]          # device int64 counter mixed into every dropout seed: set while a step is captured into / replayed from a HIP graph


def set_dropout_step_counter(t):
    """t: None, or a 1-element int64 CUDA tensor.  A HIP graph bakes the host-drawn seeds into its nodes; with a counter registered the
    kernels add it (from memory) to their seed, and a captured step that increments it draws fresh masks on every replay."""
    assert t is None or (t.is_cuda and t.dtype == torch.int64 and t.numel() == 1)
    _STEP_CTR[0] = t



def geglu_dropout(x, p_drop=0.0, training=False):
    """x, gates = chunk(2, -1); dropout(x * gelu(gates), p) (corss_ft_transformer.py:10-13, 19).  The dropout mask comes from a
    counter-based hash seeded from torch's generator state (torch.initial_seed(): per-rank seeds give per-rank masks) and a call counter."""
    _need_cuda(x, "geglu")
    p = float(p_drop) if training else 0.0
    seed = 0
    if p > 0.0:
        _DROP_CALLS[0] += 1
        seed = (torch.initial_seed() * 1000003 + _DROP_CALLS[0]) & 0x7FFFFFFFFFFFFFFF
    return _Geglu.apply(x, p, seed, _STEP_CTR[0] if p > 0.0 else None)


class _Gelu(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x_ = x.detach().contiguous()
        if x_.dtype not in (torch.float32, torch.bfloat16):
            x_ = x_.float()
        y = torch.empty_like(x_)
        call("gfe_gelu_fwd", ptr(x_), ptr(y), x_.numel(), dtype_code(x_.dtype), stream())
        ctx.save_for_backward(x_)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x_,) = ctx.saved_tensors
        d = dy.to(x_.dtype).contiguous()
        dx = torch.empty_like(x_)
        call("gfe_gelu_bwd", ptr(x_), ptr(d), ptr(dx), x_.numel(), dtype_code(x_.dtype), stream())
        return dx


def gelu(x):
    """exact-erf GELU (torch.nn.functional.gelu's default; vit.py:19, vit_3d.py:21), one kernel each way, f32 or bf16."""
    _need_cuda(x, "gelu")
    return _Gelu.apply(x)


class _Dropout(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, p, seed):
        x_ = x.detach().contiguous()
        if x_.dtype not in (torch.float32, torch.bfloat16):
            x_ = x_.float()
        y = torch.empty_like(x_)
        call("gfe_dropout", ptr(x_), ptr(y), x_.numel(), float(p), seed, dtype_code(x_.dtype), stream())
        ctx.meta = (float(p), seed)
        return y

    @staticmethod
    def backward(ctx, dy):
        p, seed = ctx.meta
        d = dy.contiguous()
        if d.dtype not in (torch.float32, torch.bfloat16):
            d = d.float()
        dx = torch.empty_like(d)
        call("gfe_dropout", ptr(d), ptr(dx), d.numel(), p, seed, dtype_code(d.dtype), stream())
        return dx, None, None


def dropout(x, p, training=True):
    """nn.Dropout(p)(x) in training mode on the HIP path: the mask is a counter-based hash of (seed, element) -- seed from torch's generator seed
    (per-rank seeds give per-rank masks) and a call counter -- regenerated by the backward, never stored; identity when not training or p == 0."""
    if not training or p <= 0.0:
        return x
    _need_cuda(x, "dropout")
    if torch.cuda.is_current_stream_capturing():
        # the seed is a host-side counter passed BY VALUE: a captured graph would replay one mask for ever (geglu_dropout mixes a device
        # step counter in for that reason; this operator has none -- ADVICE r05)
        raise RuntimeError("gfe_hip dropout() inside a HIP-graph capture would bake its seed into the graph: capture in eval mode, or use geglu_dropout's device counter")
    _DROP_CALLS[0] += 1
    # operator tag 0x5D in the top byte of the counter term: the (seed, call) streams of dropout() and geglu_dropout() (7919 * m) cannot coincide
    seed = (torch.initial_seed() * 1000003 + ((0x5D << 40) | (_DROP_CALLS[0] & 0xFFFFFFFFFF))) & 0x7FFFFFFFFFFFFFFF
    return _Dropout.apply(x, float(p), seed)


class _SiluMul(torch.autograd.Function):
    @staticmethod
    def forward(ctx, g, u):
        g_, u_ = _f(g), _f(u)
        h = torch.empty_like(g_)
        call("gfe_moe_act_fwd", ptr(g_), ptr(u_), ptr(h), g_.numel(), stream())
        ctx.save_for_backward(g_, u_)
        return h

    @staticmethod
    def backward(ctx, dh):
        g_, u_ = ctx.saved_tensors
        d = dh.float().contiguous()
        dg, du = torch.empty_like(g_), torch.empty_like(u_)
        call("gfe_moe_act_bwd", ptr(g_), ptr(u_), ptr(d), ptr(dg), ptr(du), g_.numel(), stream())
        return dg, du


def silu_mul(gate, up):
    """silu(gate) * up (the un-routed Jamba MLP, cross_atten/jamba.py:535) on the experts' activation kernels, one launch each way."""
    _need_cuda(gate, "silu_mul")
    return _SiluMul.apply(gate, up)


class _BceSigmoid(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, y):
        z, t = _f(logits).reshape(-1), _f(y).reshape(-1)
        loss = torch.empty(1, dtype=torch.float32, device=z.device)
        dz = torch.empty_like(z)
        call("gfe_bce_sigmoid", ptr(z), ptr(t), ptr(loss), ptr(dz), z.numel(), stream())
        ctx.save_for_backward(dz)
        ctx.shape = logits.shape
        return loss.view(())

    @staticmethod
    def backward(ctx, dloss):
        (dz,) = ctx.saved_tensors
        return (dz * dloss).view(ctx.shape), None


def bce_sigmoid(logits, y):
    """nn.BCELoss()(sigmoid(logits), y.float()) (classify_mamba.py:67, 104): value and gradient from one launch."""
    _need_cuda(logits, "bce_sigmoid")
    return _BceSigmoid.apply(logits, y)
