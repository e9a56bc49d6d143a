"""Jamba backbone (Mamba + attention + sparse mixture-of-experts MLPs) -- MI355X build of the reference's cross_atten/jamba.py
(JambaLMConfig :37-95, Jamba :258-306, AttentionLayer :308-340, AttentionSDPA :342-398, MambaLayer :400-439, SparseMoEBlock :441-517,
MLP :519-535, load_balancing_loss :537-556), the backbone of Cross_jamba_both (cross_atten/mamba_transformer.py:135-251).
Same class names, constructor arguments and state-dict keys.

Every projection runs on the exact-f32 MFMA GEMM (gfe_hip.train_ops.Linear), the Mamba mixers on the fused selective-scan kernels
(with the inner RMSNorms of Jamba, mamba.py:171-178), the causal attention on gfe_sdpa_small (37 tokens x 8 heads x 64), RMSNorm on
gfe_rmsnorm, the expert routing, the grouped expert projections and the weighted combine on csrc/moe.hip (gfe_hip/moe_ops.py).
Cached decoding (Jamba.step, jamba.py:298-306: a KV cache per attention layer in the reference's (B, kv heads, T, head dim) layout, the
conv window + SSM state per Mamba layer) runs one token per call on the step kernels and the one-query attention kernel.
Not built: the language-model wrapper JambaLM / from_pretrained (never used by the GFE-Mamba scripts)."""
import math
from dataclasses import dataclass, fields
from typing import Union

import torch
import torch.nn as nn
import torch.nn.functional as F

from cross_atten.mamba import MambaBlock, MambaConfig, RMSNorm
from gfe_hip.moe_ops import moe_mlp
from gfe_hip.head_ops import cross_attn_q1, sdpa_small, silu_mul
from gfe_hip.train_ops import Linear


@dataclass
class JambaLMConfig:
    d_model: int
    n_layers: int
    mlp_size: int
    initializer_range: float = 0.02
    rms_norm_eps: float = 1e-5
    # mamba related
    d_state: int = 16
    expand_factor: int = 2
    d_conv: int = 4
    dt_rank: Union[int, str] = 'auto'
    dt_min: float = 0.001
    dt_max: float = 0.1
    dt_init: str = "random"
    dt_scale: float = 1.0
    dt_init_floor = 1e-4
    bias: bool = False
    conv_bias: bool = True
    inner_layernorms: bool = True
    use_cuda: bool = False
    pscan: bool = True
    # attention related
    num_attention_heads: int = 32
    num_key_value_heads: int = 8
    attention_dropout: float = 0.
    # MoE related
    num_experts: int = 16
    num_experts_per_tok: int = 2
    # structure
    attn_layer_offset: int = 4
    attn_layer_period: int = 8
    expert_layer_offset: int = 1
    expert_layer_period: int = 2
    # language modeling
    vocab_size: int = 65536
    pad_token_id: int = 0
    tie_lm_weights: bool = True

    def __post_init__(self):
        self.d_inner = self.expand_factor * self.d_model
        if self.dt_rank == 'auto':
            self.dt_rank = math.ceil(self.d_model / 16)
        # the mixer's own record takes every field the two dataclasses share (jamba.py:89-95 passes them one by one); n_layers is unused there
        shared = {f.name for f in fields(MambaConfig)} & {f.name for f in fields(self)} - {"n_layers"}
        self.mamba_config = MambaConfig(n_layers=0, **{name: getattr(self, name) for name in sorted(shared)})


class Jamba(nn.Module):
    def __init__(self, config: JambaLMConfig):
        super().__init__()
        self.config = config
        layers = []
        for i in range(config.n_layers):                                        # jamba.py:266-276
            is_attn = (i - config.attn_layer_offset) % config.attn_layer_period == 0
            is_expert = (i - config.expert_layer_offset) % config.expert_layer_period == 0
            num_experts = config.num_experts if is_expert else 1
            layers.append((AttentionLayer if is_attn else MambaLayer)(config, num_experts=num_experts))
        self.layers = nn.ModuleList(layers)

    def forward(self, x):
        """x: (B, L, D) -> (x, [router logits of every layer])   (jamba.py:282-296)."""
        router_logits = []
        for layer in self.layers:
            (x, rl), _ = layer(x)
            router_logits.append(rl)
        return x, router_logits


    def step(self, x, caches):
        """Cached decoding, jamba.py:298-306: x (B, L, D) -- one token per call once the caches are warm (MambaLayer squeezes dim 1,
        jamba.py:421-423) -- and caches[i] from layers[i].get_empty_cache(...); returns (x, caches).  Under no_grad (or with nothing that
        requires grad) no graph is recorded; otherwise every layer runs on its differentiable operators, as the reference's plain torch code
        would (the attention kernels and MambaBlock.step both carry backwards)."""
        for i, layer in enumerate(self.layers):
            (x, _), caches[i] = layer(x, caches[i])
        return x, caches


class _MixerThenMoE(nn.Module):
    """What the two layer kinds share (jamba.py:308-340, 400-439): pre-norm token mixer + residual, pre-norm (routed) MLP + residual.
    Sub-module names are the reference's, so state dicts load either way."""

    def __init__(self, config: JambaLMConfig, num_experts: int, mixer_name: str, mixer: nn.Module):
        super().__init__()
        self.config = config
        self.add_module(mixer_name, mixer)
        top_k = config.num_experts_per_tok if num_experts > 1 else 1                # a single expert is not routed
        self.moe = SparseMoEBlock(config, num_experts=num_experts, num_experts_per_tok=top_k)
        for name in ("input_layernorm", "pre_moe_layernorm"):
            self.add_module(name, RMSNorm(config.d_model, eps=config.rms_norm_eps))

    def mix(self, n, cache):
        raise NotImplementedError

    def forward(self, x, cache=None):
        m, cache = self.mix(self.input_layernorm(x), cache)
        x = x + m
        h, router_logits = self.moe(self.pre_moe_layernorm(x))                   # jamba.py:332-335, 428-431
        return (x + h, router_logits), cache


class AttentionLayer(_MixerThenMoE):
    def __init__(self, config: JambaLMConfig, num_experts: int):
        super().__init__(config, num_experts, "self_attn", AttentionSDPA(config))

    def mix(self, n, cache):
        return self.self_attn(n, cache)                                          # jamba.py:325-329

    def get_empty_cache(self, batch_size, device):
        return (None, None)                                                      # jamba.py:340-341


class AttentionSDPA(nn.Module):
    def __init__(self, config: JambaLMConfig):
        super().__init__()
        self.config = config
        self.hidden_size = config.d_model
        self.num_heads = config.num_attention_heads
        self.head_dim = self.hidden_size // self.num_heads
        self.num_key_value_heads = config.num_key_value_heads
        self.num_key_value_groups = self.num_heads // self.num_key_value_heads
        self.attention_dropout = config.attention_dropout
        self.q_proj = Linear(self.hidden_size, self.num_heads * self.head_dim, bias=False)
        self.k_proj = Linear(self.hidden_size, self.num_key_value_heads * self.head_dim, bias=False)
        self.v_proj = Linear(self.hidden_size, self.num_key_value_heads * self.head_dim, bias=False)
        self.o_proj = Linear(self.num_heads * self.head_dim, self.hidden_size, bias=False)

    def forward(self, x, cache=None):
        p_drop = self.attention_dropout if self.training else 0.0                 # dropout_p of F.scaled_dot_product_attention (jamba.py:390-392)
        B, L, _ = x.shape
        q, k, v = self.q_proj(x), self.k_proj(x), self.v_proj(x)
        Hkv, dh = self.num_key_value_heads, self.head_dim
        if cache is not None:
            # KV cache (jamba.py:373-383): the cache keeps the reference's layout, (B, kv heads, T, head dim) per tensor, so caches are
            # interchangeable with the reference's; with a cache the reference calls SDPA with is_causal=False -- also for a multi-token
            # first call -- and so does this
            past_k, past_v = cache
            k4 = k.view(B, L, Hkv, dh).transpose(1, 2)
            v4 = v.view(B, L, Hkv, dh).transpose(1, 2)
            if past_k is not None:
                k4, v4 = torch.cat([past_k, k4], dim=2), torch.cat([past_v, v4], dim=2)
            cache = (k4, v4)
            T = k4.shape[2]
            k, v = k4.transpose(1, 2).reshape(B, T, Hkv * dh), v4.transpose(1, 2).reshape(B, T, Hkv * dh)
        else:
            T = L
        if self.num_key_value_groups > 1:                                        # GQA: repeat_kv (jamba.py:558-567) on the projection layout
            rep = lambda t: t.view(B, T, Hkv, 1, dh).expand(-1, -1, -1, self.num_key_value_groups, -1).reshape(B, T, -1)
            k, v = rep(k), rep(v)
        if cache is None:
            o = sdpa_small(q, k, v, self.num_heads, causal=True, dropout_p=p_drop)   # F.scaled_dot_product_attention(..., is_causal=True) (:390-392)
        elif L == 1:
            if p_drop > 0:
                raise NotImplementedError("attention dropout while decoding from a KV cache (the reference's training-mode corner) is not built")
            o = cross_attn_q1(q, k.contiguous(), v.contiguous(), self.num_heads)     # one query against the T cached keys: gfe_cross_attn_q1
        elif T == L:
            o = sdpa_small(q, k, v, self.num_heads, causal=False, dropout_p=p_drop)  # first call with an empty cache: non-causal, as the reference
        else:
            # several new tokens onto a warm cache (the module-level generality of jamba.py:373-392; Jamba.step itself feeds one token per call,
            # :421-423): every new token is a one-query problem over the T cached + new keys, non-causal as the reference's is_causal=False
            if p_drop > 0:
                raise NotImplementedError("attention dropout while decoding from a KV cache (the reference's training-mode corner) is not built")
            E = self.num_heads * dh
            kx = k.unsqueeze(1).expand(B, L, T, E).reshape(B * L, T, E)
            vx = v.unsqueeze(1).expand(B, L, T, E).reshape(B * L, T, E)
            o = cross_attn_q1(q.reshape(B * L, 1, E), kx, vx, self.num_heads).view(B, L, E)
        return self.o_proj(o), cache


class MambaLayer(_MixerThenMoE):
    def __init__(self, config: JambaLMConfig, num_experts: int):
        super().__init__(config, num_experts, "mamba", MambaBlock(config=config.mamba_config))

    def mix(self, n, cache):
        if cache is None:
            return self.mamba(n), None                                           # jamba.py:418-420
        m, cache = self.mamba.step(n.squeeze(1), cache)                          # :421-423, single token on the step kernels (mamba.py:342-405)
        return m.unsqueeze(1), cache

    def get_empty_cache(self, batch_size, device):
        return (None, torch.zeros(batch_size, self.config.d_inner, self.config.d_conv - 1, device=device))   # jamba.py:438-439


class SparseMoEBlock(nn.Module):
    def __init__(self, config: JambaLMConfig, num_experts: int, num_experts_per_tok: int):
        super().__init__()
        self.hidden_dim = config.d_model
        self.ffn_dim = config.mlp_size
        self.num_experts = num_experts
        self.top_k = num_experts_per_tok
        self.router = Linear(self.hidden_dim, self.num_experts, bias=False) if num_experts > 1 else None
        self.experts = nn.ModuleList([MLP(config) for _ in range(self.num_experts)])

    def forward(self, x):
        B, L, D = x.shape
        if self.num_experts == 1:                                                # no routing (jamba.py:470-478)
            return self.experts[0](x), torch.ones((B * L, 1), device=x.device, dtype=x.dtype, requires_grad=x.requires_grad)
        # routing, expert sort, the three grouped projections and the weighted combine: gfe_hip/moe_ops.py over csrc/moe.hip (one launch per
        # projection for ALL experts, no host synchronisation); the router logits are returned as the reference does (:517)
        out, router_logits = moe_mlp(x.reshape(-1, D), self.top_k, self.router.weight, [e.gate_proj.weight for e in self.experts],
                                     [e.up_proj.weight for e in self.experts], [e.down_proj.weight for e in self.experts])
        return out.reshape(B, L, D), router_logits


class MLP(nn.Module):
    def __init__(self, config: JambaLMConfig):
        super().__init__()
        self.hidden_dim = config.d_model
        self.ffn_dim = config.mlp_size
        self.gate_proj = Linear(self.hidden_dim, self.ffn_dim, bias=False)
        self.down_proj = Linear(self.ffn_dim, self.hidden_dim, bias=False)
        self.up_proj = Linear(self.hidden_dim, self.ffn_dim, bias=False)

    def forward(self, x):
        return self.down_proj(silu_mul(self.gate_proj(x), self.up_proj(x)))      # jamba.py:535: silu(gate) * up, one kernel each way


def load_balancing_loss(router_logits, num_experts, num_experts_per_tok):
    """The auxiliary balance term of jamba.py:537-556 (not used by the classification scripts; kept because callers of the reference can
    import it): num_experts * sum over (top-k slot, expert) of [fraction of tokens whose slot picked the expert] * [mean router probability of
    the expert], over the layers that route (more than one expert)."""
    logits = torch.cat([r for r in router_logits if r.shape[1] > 1], dim=0)
    probs = logits.softmax(dim=-1)                                                     # (tokens, E)
    chosen = probs.topk(num_experts_per_tok, dim=-1).indices                           # (tokens, k)
    picked = torch.zeros((num_experts_per_tok, num_experts), dtype=probs.dtype, device=probs.device)
    picked.scatter_add_(1, chosen.t(), torch.ones_like(chosen.t(), dtype=probs.dtype))     # tokens per (slot, expert)
    return num_experts * (picked / logits.shape[0] * probs.mean(dim=0)).sum()
