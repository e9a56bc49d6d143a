"""CrossAttention / SelfAttention -- MI355X build of the reference's cross_atten/sd_cross_atten.py (:7-37, :39-70).
Same constructor arguments and state-dict keys (q_proj, k_proj, v_proj, out_proj / in_proj, out_proj).  The q / out projections run on
the exact-f32 MFMA GEMM.  CrossAttention is called with ONE query per sample by the classifier (mamba_transformer.py:122-124); for one
query the K / V projections of the image condition -- 94 % of the trainable FLOPs in the reference -- fold away algebraically
(gfe_cross_attn_q1_folded, csrc/xattn_fold.hip): scores = (W_k^T q) . y_j, output = W_v (sum_j p_j y_j) + b_v, exact f32, the condition
read in place from the f32 volumes.  The materialised path (project_kv + gfe_cross_attn_q1, bf16 K / V GEMMs) is kept for a condition
that needs its own gradient and for callers that pass precomputed kv.

No torch-math attention is left in these modules (round 4): SelfAttention runs on gfe_sdpa_small (sequences up to 64 tokens, head dim
<= 64: the sizes of the classifier's token stream; beyond that it raises); CrossAttention with several queries runs gfe_cross_attn_q1 over
(sample, query) pairs."""
import torch
from torch import nn

from gfe_hip.head_ops import cross_attn_q1, cross_attn_q1_folded, sdpa_small
from gfe_hip.train_ops import Condition, Linear, linear


class SelfAttention(nn.Module):
    def __init__(self, n_heads, d_embed, in_proj_bias=True, out_proj_bias=True):
        super().__init__()
        self.in_proj = Linear(d_embed, 3 * d_embed, bias=in_proj_bias)
        self.out_proj = Linear(d_embed, d_embed, bias=out_proj_bias)
        self.n_heads = n_heads
        self.d_head = d_embed // n_heads

    def forward(self, x, causal_mask=False):
        """sd_cross_atten.py:18-37: softmax(q k^T / sqrt(d_head) [+ causal mask]) v per head, then out_proj."""
        b, s, d = x.shape
        if s > 64 or self.d_head > 64:
            raise NotImplementedError(f"SelfAttention on the HIP path covers <= 64 tokens and head dim <= 64 (gfe_sdpa_small); got {s} tokens, "
                                      f"head dim {self.d_head}.  Long sequences with head dim 64: gfe_hip.train_ops.qkv_flash_attention.")
        q, k, v = self.in_proj(x).chunk(3, dim=-1)
        return self.out_proj(sdpa_small(q.contiguous(), k.contiguous(), v.contiguous(), self.n_heads, causal=bool(causal_mask)))


class CrossAttention(nn.Module):
    def __init__(self, n_heads, d_embed, d_cross, in_proj_bias=True, out_proj_bias=True):
        super().__init__()
        self.q_proj = Linear(d_embed, d_embed, bias=in_proj_bias)
        self.k_proj = Linear(d_cross, d_embed, bias=in_proj_bias)
        self.v_proj = Linear(d_cross, d_embed, bias=in_proj_bias)
        self.out_proj = Linear(d_embed, d_embed, bias=out_proj_bias)
        self.n_heads = n_heads
        self.d_head = d_embed // n_heads

    def project_kv(self, y):
        """k, v = k_proj(y), v_proj(y) (sd_cross_atten.py:52-53) MATERIALISED, for y: (B, Lkv, d_cross) tensor or a gfe_hip.train_ops.Condition
        (bf16 matrix-core operands beyond F32_LINEAR_MAX_K inputs).  Only the multi-query path needs them: with one query per sample
        forward() never forms K or V (csrc/xattn_fold.hip)."""
        if isinstance(y, Condition):
            a16, aT16 = y.cond.view(-1, y.d_cross), y.condT
            d = self.k_proj.weight.shape[0]
            k = linear(a16, self.k_proj.weight, self.k_proj.bias, x16=a16, xT16=aT16).view(y.B, y.keys, d)
            v = linear(a16, self.v_proj.weight, self.v_proj.bias, x16=a16, xT16=aT16).view(y.B, y.keys, d)
            return k, v
        return self.k_proj(y), self.v_proj(y)

    def _fold_fits(self, imgs):
        """the shape limits of gfe_cross_attn_q1_folded_* (csrc/xattn_fold.hip xf_shape_ok), checked here so that a larger condition falls
        back to the materialised path instead of raising GFE_ERR_SHAPE"""
        d3 = imgs[0].shape[2]
        return (len(imgs) <= 4 and all(im.shape[2] == d3 and im.shape[1] == imgs[0].shape[1] for im in imgs) and d3 <= 256
                and len(imgs) * d3 <= 1024 and self.n_heads <= 64 and self.d_head <= 1024)

    @staticmethod
    def _images(y):
        """the condition as a list of f32 (B, d_cross, keys_i) matrices whose COLUMNS are the keys: a Condition's volumes in place, or
        the transpose of a plain (B, Lkv, d_cross) tensor"""
        if isinstance(y, Condition):
            return y.images
        return [y.detach().float().transpose(1, 2).contiguous()]

    def forward(self, x, y, kv=None):
        """x: (B, Lq, d_embed); y: (B, Lkv, d_cross) tensor or a gfe_hip.train_ops.Condition (kv: project_kv(y) computed earlier, multi-query
        calls only).  sd_cross_atten.py:49-70.
        Lq == 1 -- every call the classifier makes (mamba_transformer.py:122-124) -- takes the folded kernels: scores = (W_k^T q) . y,
        output = W_v (sum_j p_j y_j) + b_v, exact f32, the condition read in place, K and V never formed.  Lq > 1 (the reference's
        Transformer_Cross, corss_ft_transformer.py:123-133, never instantiated by the classify scripts) materialises K / V and runs the
        one-query kernel over (sample, query) pairs (the class stays general, as the reference's is)."""
        b, lq, d = x.shape
        q = self.q_proj(x)
        if lq == 1 and kv is None and not (torch.is_tensor(y) and y.requires_grad):
            imgs = self._images(y)
            if self._fold_fits(imgs):
                return self.out_proj(cross_attn_q1_folded(q, self.k_proj.weight, self.k_proj.bias, self.v_proj.weight, self.v_proj.bias,
                                                          self.n_heads, imgs))
            # beyond the folded kernels' limits (include/gfe_hip.h: at most 4 volumes of at most 256 keys each, 1024 keys in all, 64 heads):
            # K and V materialised, the one-query kernel over them -- the path every call took before round 5
        k, v = kv if kv is not None else self.project_kv(y)
        if lq == 1:
            return self.out_proj(cross_attn_q1(q, k, v, self.n_heads))           # (a condition that itself wants a gradient)
        # several queries per sample (the reference's Transformer_Cross, corss_ft_transformer.py:123-133; never on the classify path): every
        # query is an independent one-query problem over its sample's K / V -- the same kernel on (B * Lq) "samples"; the broadcast of K / V over
        # the queries (and the sum of their gradients over them) is index glue, the attention arithmetic stays in gfe_cross_attn_q1
        nk = k.shape[1]
        kx = k.unsqueeze(1).expand(b, lq, nk, d).reshape(b * lq, nk, d)
        vx = v.unsqueeze(1).expand(b, lq, nk, d).reshape(b * lq, nk, d)
        o = cross_attn_q1(q.reshape(b * lq, 1, d), kx, vx, self.n_heads)
        return self.out_proj(o.view(b, lq, d))
