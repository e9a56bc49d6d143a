"""Combine_classfier_vit_mid -- MI355X build of the one classifier head classify_mamba.py uses (reference:
classify/classifier.py:324-333; the other seven heads of that file are unused by the hot path).  Same constructor and
state-dict keys (vit_mid_linear.{weight,bias}); additive `in_features` (default 320*120 = the reference's hard-coded
value) for non-native volume sizes."""
import torch
from torch import nn

from gfe_hip import nn_ops as K
from gfe_hip.train_ops import mid_linear


class Combine_classfier_vit_mid(nn.Module):
    def __init__(self, seq_length=1, in_features=320 * 120):
        super().__init__()
        self.vit_mid_linear = nn.Linear(in_features, seq_length)

    @staticmethod
    def _channels_last(t):
        """(B, C, H, W) -> channels-last bf16 (B, H, W, C); free for the generator's own outputs."""
        p = t.permute(0, 2, 3, 1)
        if p.dtype == torch.bfloat16 and p.is_contiguous():
            return p
        return K.cast(p.contiguous().float(), torch.bfloat16)

    def forward(self, mid_input, mid_output):
        if not mid_input.is_cuda:
            raise RuntimeError("no CPU fallback")
        a, b = self._channels_last(mid_input), self._channels_last(mid_output)
        lin = self.vit_mid_linear
        if lin.out_features == 4:
            f = mid_linear(a, b, lin.weight, lin.bias)                    # (B, 2C, S)
        else:   # other seq_length values (the reference default is 1): same arithmetic through the GEMM
            from gfe_hip.train_ops import linear
            x = torch.cat([a, b], dim=-1).flatten(1, 2).transpose(1, 2).contiguous()
            f = linear(x, lin.weight, lin.bias)
        return f.transpose(1, 2).contiguous()                             # classifier.py:332
