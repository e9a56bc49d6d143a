"""CrossAttention / SelfAttention -- MI355X build of the reference's cross_atten/sd_cross_atten.py (:7-37, :39-70).
Same constructor arguments and state-dict keys (q_proj, k_proj, v_proj, out_proj / in_proj, out_proj).  The four
q / out projections run on the exact-f32 MFMA GEMM; the K/V projections of the image condition (94 % of the trainable FLOPs) on the
bf16 one, reading the bf16 condition buffers of gfe_hip.train_ops.Condition in both GEMM layouts so neither forward nor weight
gradient re-casts or transposes the 28 MB condition; the one-query softmax attention between them is gfe_cross_attn_q1."""
import math

import torch
from torch import nn
from torch.nn import functional as F

from gfe_hip.head_ops import cross_attn_q1
from gfe_hip.train_ops import Condition, Linear, linear


class SelfAttention(nn.Module):
    def __init__(self, n_heads, d_embed, in_proj_bias=True, out_proj_bias=True):
        super().__init__()
        self.in_proj = Linear(d_embed, 3 * d_embed, bias=in_proj_bias)
        self.out_proj = Linear(d_embed, d_embed, bias=out_proj_bias)
        self.n_heads = n_heads
        self.d_head = d_embed // n_heads

    def forward(self, x, causal_mask=False):
        b, s, d = x.shape
        q, k, v = self.in_proj(x).chunk(3, dim=-1)
        q, k, v = [t.view(b, s, self.n_heads, self.d_head).transpose(1, 2) for t in (q, k, v)]
        weight = q @ k.transpose(-1, -2)
        if causal_mask:
            weight = weight.masked_fill(torch.ones_like(weight, dtype=torch.bool).triu(1), -torch.inf)
        weight = F.softmax(weight / math.sqrt(self.d_head), dim=-1)
        return self.out_proj((weight @ v).transpose(1, 2).reshape(b, s, d))


class CrossAttention(nn.Module):
    def __init__(self, n_heads, d_embed, d_cross, in_proj_bias=True, out_proj_bias=True):
        super().__init__()
        self.q_proj = Linear(d_embed, d_embed, bias=in_proj_bias)
        self.k_proj = Linear(d_cross, d_embed, bias=in_proj_bias)
        self.v_proj = Linear(d_cross, d_embed, bias=in_proj_bias)
        self.out_proj = Linear(d_embed, d_embed, bias=out_proj_bias)
        self.n_heads = n_heads
        self.d_head = d_embed // n_heads

    def forward(self, x, y):
        """x: (B, Lq, d_embed); y: (B, Lkv, d_cross) tensor or a gfe_hip.train_ops.Condition.  sd_cross_atten.py:49-70."""
        b, lq, d = x.shape
        q = self.q_proj(x)
        if isinstance(y, Condition):
            a16, aT16 = y.cond.view(-1, y.d_cross), y.condT
            k = linear(a16, self.k_proj.weight, self.k_proj.bias, x16=a16, xT16=aT16).view(b, y.keys, d)
            v = linear(a16, self.v_proj.weight, self.v_proj.bias, x16=a16, xT16=aT16).view(b, y.keys, d)
        else:
            k, v = self.k_proj(y), self.v_proj(y)
        if lq == 1 and q.is_cuda:
            return self.out_proj(cross_attn_q1(q, k, v, self.n_heads))          # the classify path: one query per sample, one kernel each way
        q = q.view(b, -1, self.n_heads, self.d_head).transpose(1, 2)
        k = k.view(b, -1, self.n_heads, self.d_head).transpose(1, 2)
        v = v.view(b, -1, self.n_heads, self.d_head).transpose(1, 2)
        weight = F.softmax((q @ k.transpose(-1, -2)) / math.sqrt(self.d_head), dim=-1)
        out = (weight @ v).transpose(1, 2).contiguous().view(b, lq, d)
        return self.out_proj(out)
