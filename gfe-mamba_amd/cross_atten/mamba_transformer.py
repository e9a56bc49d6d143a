"""Cross_mamba_both -- MI355X build of the classifier used by classify_mamba.py (reference:
cross_atten/mamba_transformer.py:11-133) -- and its ablation twin Cross_mamba_ablation (:254-385).  Same keyword-only
constructors, forward contracts and state-dict keys.  Additive: `d_cross` (default 160*160, the reference's hard-coded value at
:84 / :325) so that 96^3 / 128^3 volumes are constructible.  Cross_jamba_both (:135-251) swaps the Mamba stack for the Jamba backbone."""
import torch
import torch.nn.functional as F
from torch import nn

from cross_atten.corss_ft_transformer import FeedForward, GEGLU, NumericalEmbedder  # noqa: F401
from cross_atten.jamba import Jamba, JambaLMConfig
from cross_atten.mamba import Mamba, MambaConfig
from cross_atten.sd_cross_atten import CrossAttention
from gfe_hip.head_ops import LayerNorm, embed_tokens, mean_tokens
from gfe_hip.train_ops import Condition, Linear, aux_region

import os
_MATERIALISED_KV = os.environ.get("GFE_XATTN_MATERIALISED") == "1"


class Cross_mamba_both(nn.Module):
    def __init__(self, *, categories, num_continuous, dim, depth, heads, dim_head=16, dim_out=1, num_special_tokens=2,
                 attn_dropout=0., ff_dropout=0., cross_ff_multi=2, cross_ff_dropout=0.1, d_cross=160 * 160):
        super().__init__()
        assert all(map(lambda n: n > 0, categories)), 'number of each category must be positive'
        assert len(categories) + num_continuous > 0, 'input shape must not be null'
        self.num_categories = len(categories)
        self.num_unique_categories = sum(categories)
        self.num_special_tokens = num_special_tokens
        total_tokens = self.num_unique_categories + num_special_tokens
        if self.num_unique_categories > 0:
            categories_offset = F.pad(torch.tensor(list(categories)), (1, 0), value=num_special_tokens)
            self.register_buffer('categories_offset', categories_offset.cumsum(dim=-1)[:-1])     # :44-46
            self.categorical_embeds = nn.Embedding(total_tokens, dim)
        self.num_continuous = num_continuous
        if self.num_continuous > 0:
            self.numerical_embedder = NumericalEmbedder(dim, self.num_continuous)
        self.cls_token = nn.Parameter(torch.randn(1, 1, dim))
        self.transformer = Mamba(MambaConfig(d_model=dim, n_layers=depth, use_cuda=True))        # :64-65
        self.to_logits = nn.Sequential(LayerNorm(dim), Linear(dim, dim_out))
        self.final_cross = CrossAttention(n_heads=heads, d_embed=dim, d_cross=d_cross)           # :84
        self.final_feed = FeedForward(dim, mult=cross_ff_multi, dropout=cross_ff_dropout)        # :85

    def forward(self, x_categ, x_numer, feature_img, image_condition=None):
        assert x_categ.shape[-1] == self.num_categories, f'you must pass in {self.num_categories} values for your categories input'
        if image_condition is None:
            raise ValueError("Cross_mamba_both needs image_condition=[mri, pet] (the reference fails with NameError at :124)")
        # the condition is read in place from the f32 volumes by the one-query cross-attention (no K / V projections, no bf16 copies: round 5)
        if _MATERIALISED_KV:      # A/B switch (GFE_XATTN_MATERIALISED=1): round 4's path -- bf16 K / V projections of the condition on a side stream
            with aux_region() as aux:
                whole_condition = image_condition if isinstance(image_condition, Condition) else Condition(list(image_condition))
                kv = self.final_cross.project_kv(whole_condition)
        else:
            whole_condition = image_condition if isinstance(image_condition, Condition) else Condition(list(image_condition))   # :89-94
            kv = None
        x = self._tokens(x_categ, x_numer, feature_img)                                           # :97-117, one kernel each way
        x = self.transformer(x)
        x = mean_tokens(x)                                                                        # :122
        if kv is not None:
            aux.join()
            if kv[0].is_cuda and not torch.cuda.is_current_stream_capturing():       # allocated from the side stream's pool, read on this one
                for t_ in kv:
                    t_.record_stream(torch.cuda.current_stream())
        x = self.final_cross(x, whole_condition, kv=kv) + x                                       # :124
        x = self.final_feed(x) + x                                                                # :125
        return self.to_logits(x.squeeze(1))                                                       # :127-131


def _tokens(self, x_categ, x_numer, feature_img, table=True):
    """[cls | categorical embeddings (offset add + gather) | numerical embeddings | image tokens] (mamba_transformer.py:97-117)."""
    cat = table and self.num_unique_categories > 0
    num = table and self.num_continuous > 0
    ne = self.numerical_embedder if num else None
    assert ne is None or not hasattr(ne, 'linear'), "shrink_dim embedder is not on the classify path"
    return embed_tokens(x_categ if cat else None, self.categories_offset if cat else None, self.categorical_embeds.weight if cat else None,
                        x_numer if num else None, ne.weights if num else None, ne.biases if num else None, self.cls_token, feature_img)


Cross_mamba_both._tokens = _tokens


class Cross_mamba_ablation(Cross_mamba_both):
    """Reference: cross_atten/mamba_transformer.py:254-385 -- the same modules and state-dict keys as Cross_mamba_both; forward
    takes three switches: feature_img=None (table tokens only, :364), no_table=True (cls + image features only, :359) and
    image_condition=None (skips final_cross / final_feed, :373-375)."""

    def forward(self, x_categ, x_numer, feature_img=None, image_condition=None, no_table=False):
        assert x_categ.shape[-1] == self.num_categories, f'you must pass in {self.num_categories} values for your categories input'
        x = self._tokens(x_categ, x_numer, feature_img, table=not no_table)                       # :339-364
        x = self.transformer(x)
        x = mean_tokens(x)                                                                        # :371
        if image_condition is not None:
            cond = image_condition if isinstance(image_condition, Condition) else Condition(list(image_condition))
            x = self.final_cross(x, cond) + x                                                     # :374
            x = self.final_feed(x) + x                                                            # :375
        return self.to_logits(x.squeeze(1))                                                       # :377-381


class Cross_jamba_both(Cross_mamba_both):
    """Reference: cross_atten/mamba_transformer.py:135-251 -- Cross_mamba_both with the Jamba backbone (2*depth layers: Mamba mixers with
    inner RMSNorms, one causal-attention layer at index 4, 16-expert top-2 MoE MLPs on the odd layers) instead of the Mamba stack; same
    embedding, cross-attention, feed-forward and logit modules and state-dict keys."""

    def __init__(self, *, categories, num_continuous, dim, depth, heads, dim_head=16, dim_out=1, num_special_tokens=2,
                 attn_dropout=0., ff_dropout=0., cross_ff_multi=2, cross_ff_dropout=0.1, d_cross=160 * 160):
        super().__init__(categories=categories, num_continuous=num_continuous, dim=dim, depth=1, heads=heads, dim_head=dim_head,
                         dim_out=dim_out, num_special_tokens=num_special_tokens, attn_dropout=attn_dropout, ff_dropout=ff_dropout,
                         cross_ff_multi=cross_ff_multi, cross_ff_dropout=cross_ff_dropout, d_cross=d_cross)
        config = JambaLMConfig(d_model=dim, n_layers=depth * 2, use_cuda=True, mlp_size=dim * 2, attention_dropout=attn_dropout,
                               num_attention_heads=heads)                                     # :189-191
        self.transformer = Jamba(config)

    def forward(self, x_categ, x_numer, feature_img, image_condition=None):
        assert x_categ.shape[-1] == self.num_categories, f'you must pass in {self.num_categories} values for your categories input'
        if image_condition is None:
            raise ValueError("Cross_jamba_both needs image_condition=[mri, pet] (the reference fails with NameError at :242)")
        whole_condition = image_condition if isinstance(image_condition, Condition) else Condition(list(image_condition))   # :205-210
        x = self._tokens(x_categ, x_numer, feature_img)                                           # :212-235
        x = self.transformer(x)                                                                   # :239
        x = mean_tokens(x[0])                                                                     # :240
        x = self.final_cross(x, whole_condition) + x                                              # :242
        x = self.final_feed(x) + x                                                                # :243
        return self.to_logits(x.squeeze(1))                                                       # :245-251
