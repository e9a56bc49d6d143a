"""FT-Transformer pieces used by Cross_mamba_both -- MI355X build of the reference's cross_atten/corss_ft_transformer.py:
GEGLU :10-13, FeedForward :15-22, NumericalEmbedder :150-163 (same names / keys).  The other classes of the reference file
(Attention, Transformer*, Cross_transformer*, FTTransformer_cross*) are earlier baselines that classify_mamba.py never
instantiates and are out of scope."""
import torch
from torch import nn

from gfe_hip.head_ops import LayerNorm, geglu_dropout
from gfe_hip.train_ops import Linear


class GEGLU(nn.Module):
    def forward(self, x):
        return geglu_dropout(x)                       # one kernel each way (gfe_geglu_fwd / _bwd); CPU tensors raise (no CPU fallback)


class _FeedForward(nn.Sequential):
    """LayerNorm -> Linear -> GEGLU -> Dropout -> Linear with the reference's Sequential indices (parameters at .0, .1, .4).  GEGLU and
    the dropout behind it run as ONE kernel each way (the mask is regenerated in the backward from a seed, never stored)."""

    def forward(self, x):
        ln, lin1, _, drop, lin2 = self
        h = lin1(ln(x))
        h = geglu_dropout(h, drop.p, self.training and drop.training)
        return lin2(h)


def FeedForward(dim, mult=4, dropout=0.):
    return _FeedForward(LayerNorm(dim), Linear(dim, dim * mult * 2), GEGLU(), nn.Dropout(dropout), Linear(dim * mult, dim))


class NumericalEmbedder(nn.Module):
    def __init__(self, dim, num_numerical_types, shrink_dim=False):
        super().__init__()
        if shrink_dim:
            self.linear = Linear(num_numerical_types, num_numerical_types // 2)
            num_numerical_types = num_numerical_types // 2
        self.weights = nn.Parameter(torch.randn(num_numerical_types, dim))
        self.biases = nn.Parameter(torch.randn(num_numerical_types, dim))

    def forward(self, x):
        if hasattr(self, 'linear'):
            x = self.linear(x)
        return x.unsqueeze(-1) * self.weights + self.biases
