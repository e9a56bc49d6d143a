"""Residual 3D-UNet + bottleneck ViT generator -- MI355X build of the one model the classify_mamba hot path uses.

Reference: pytorch3dunet/unet3d/model.py:83-175 (Mid_UNet_vit) and :308-331 (Residual_mid_UNet3D_vit).  Same constructor
arguments, `forward(x, output_mid=False, output_vit_mid=False)` contract and state-dict keys.  Additive: `vol_size`
derives the constants the reference hard-codes for 160x160x96 volumes (image_size=(320,120), patch 40: model.py:107-117)
so that 96^3 / 128^3 volumes are constructible (geometry rule: SURVEY.md 8-d).

Forward-only, eval semantics (the generator is frozen: classify_mamba.py:53,100).  Activations are channels-last bf16 on
the GPU; the returned mid_input / mid_output are logical (B, C, H, W) views of channels-last bf16 memory and pet is
(B, 1, D, H, W) f32.
"""
import torch
from torch import nn

from gfe_hip import nn_ops as K
from pytorch3dunet.unet3d.buildingblocks import ResNetBlock, _PackCache, create_decoders, create_encoders
from vit_pytorch_diy import ViT


def vit_geometry(vol_size, md1=8):
    D1, D2, D3 = vol_size
    assert D1 % (4 * md1) == 0 and D2 % 4 == 0 and D3 % 4 == 0, "volume must satisfy D1 % 32 == 0, D2 % 4 == 0, D3 % 4 == 0"
    H, W, p = (D2 // 4) * md1, (D1 // (4 * md1)) * (D3 // 4), D2 // 4
    assert W % p == 0, "((D1/32)*(D3/4)) must be a multiple of D2/4"
    return (H, W), p


class Mid_UNet_vit(nn.Module):
    def __init__(self, in_channels, out_channels, final_sigmoid, basic_module, f_maps=(64, 128, 256, 512), layer_order='gcr',
                 num_groups=8, is_segmentation=True, conv_kernel_size=3, pool_kernel_size=2, conv_padding=1, conv_upscale=2,
                 upsample='default', dropout_prob=0.1, is3d=True, vol_size=(160, 160, 96), vit_kwargs=None):
        super().__init__()
        assert isinstance(f_maps, (list, tuple)) and len(f_maps) > 1
        self.encoders = create_encoders(in_channels, f_maps, basic_module, conv_kernel_size, conv_padding, conv_upscale,
                                        dropout_prob, layer_order, num_groups, pool_kernel_size, is3d)
        self.decoders = create_decoders(f_maps, basic_module, conv_kernel_size, conv_padding, layer_order, num_groups,
                                        upsample, dropout_prob, is3d)
        scale = 2 ** (len(f_maps) - 1)
        assert scale == 4, "bottleneck fold assumes two poolings (f_maps of length 3)"
        image_size, patch = vit_geometry(vol_size)
        kw = dict(image_size=image_size, patch_size=patch, dim=512, depth=4, heads=6, mlp_dim=2048, dropout=0.1,
                  emb_dropout=0.1, channels=f_maps[-1])                       # model.py:107-117
        kw.update(vit_kwargs or {})
        self.mid = ViT(**kw)
        self.mid_linear = nn.Linear(960, 1024)                                # model.py:119 (dead parameter, kept for the keys)
        self.final_conv = nn.Conv3d(f_maps[0], out_channels, 1)
        assert out_channels == 1, "hot path: single-channel PET output"
        if is_segmentation:
            self.final_activation = nn.Sigmoid() if final_sigmoid else nn.Softmax(dim=1)
        else:
            self.final_activation = None
        self.vol_size = tuple(vol_size)
        self._fc_pack = _PackCache()

    @staticmethod
    def _ncdhw(t):
        return t.permute(0, 4, 1, 2, 3)

    @torch.no_grad()
    def forward(self, x, output_mid=False, output_vit_mid=False):
        if not x.is_cuda:
            raise RuntimeError("the MI355X generator runs on the GPU only (no CPU fallback)")
        assert x.dim() == 5 and x.shape[1] == 1, "expects (B, 1, D, H, W)"
        x = x.contiguous()
        if x.dtype not in (torch.float32, torch.bfloat16):
            x = x.float()
        feats = []
        for enc in self.encoders:
            x = enc(x)
            feats.insert(0, x)
        feats = feats[1:]
        d, h, w = x.shape[1:4]
        mid_input = K.fold_mid(x, md1=8)                                      # model.py:150
        mid_output = self.mid(mid_input)
        x = K.fold_mid(mid_output, md1=8, inverse=True, shape=(d, h, w))      # model.py:152
        fc = self.final_conv
        fw, fb = self._fc_pack.get([fc.weight, fc.bias], lambda: (fc.weight.detach().float().view(-1).contiguous(),
                                                                  float(fc.bias.detach().float().cpu().item())))
        dec_feats = []
        pet = None
        for i, (dec, ef) in enumerate(zip(self.decoders, feats)):
            if i == len(self.decoders) - 1 and not output_mid and dec.fuses_out1():
                pet = dec(ef, x, out1=(fw, fb))              # final_conv in the last conv's epilogue: its 64-channel input is never stored
                break
            x = dec(ef, x)
            if output_mid:
                dec_feats.append(x)
        if pet is None:
            pet = K.conv_out1(x, fw, fb)
        if not self.training and self.final_activation is not None:
            pet = self.final_activation(pet)
        if output_mid:
            return [self._ncdhw(f) for f in reversed(feats)], [self._ncdhw(f) for f in reversed(dec_feats)], pet
        if output_vit_mid:
            return mid_input.permute(0, 3, 1, 2), mid_output.permute(0, 3, 1, 2), pet
        return pet


class Residual_mid_UNet3D_vit(Mid_UNet_vit):
    def __init__(self, in_channels, out_channels, final_sigmoid=True, f_maps=(64, 128, 256, 512), layer_order='gcr', num_groups=8,
                 is_segmentation=True, conv_padding=1, conv_upscale=2, upsample='default', dropout_prob=0.1, **kwargs):
        super().__init__(in_channels=in_channels, out_channels=out_channels, final_sigmoid=final_sigmoid, basic_module=ResNetBlock,
                         f_maps=f_maps, layer_order=layer_order, num_groups=num_groups, is_segmentation=is_segmentation,
                         conv_padding=conv_padding, conv_upscale=conv_upscale, upsample=upsample, dropout_prob=dropout_prob,
                         is3d=True, **kwargs)
