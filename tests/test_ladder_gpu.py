"""Per-stage error ladder of the frozen generator (pytorch3dunet/unet3d/model.py:137-175): for every stage -- the three encoder blocks,
the bottleneck ViT, the two decoder blocks (+ final 1x1x1 conv) -- two numbers against the oracle (fp32 CPU restatement, pinned to the
reference by tests/test_oracle_golden.py):

  local       the HIP stage's output vs the ORACLE's stage applied to the HIP path's own previous output (what this stage adds)
  cumulative  the HIP stage's output vs the oracle's chain from the input volume (what the head finally sees)

so that the bf16 budget of the chain (BASELINE.json: 1e-2 rel for conv / attention in bf16) is attributed layer by layer instead of
being read off the end of a 12-conv + ViT chain."""
import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err
from oracle import ref_ops as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _ncdhw(t):
    """channels-last (B, D, H, W, C) bf16 device tensor -> (B, C, D, H, W) f32 CPU"""
    return t.float().permute(0, 4, 1, 2, 3).contiguous().cpu()


def _ladder(gen, x, vit_heads, vit_depth):
    from gfe_hip import nn_ops as K
    sd = {k: v.detach().float().cpu() for k, v in gen.state_dict().items()}
    rows = []
    with torch.no_grad():
        # ---- HIP path, stage by stage (the same calls as Mid_UNet_vit.forward)
        hip, cur = [], x.to(DEV)
        for enc in gen.encoders:
            cur = enc(cur)
            hip.append(cur)
        e0, e1, e2 = hip
        d, h, w = e2.shape[1:4]
        mid_in = K.fold_mid(e2, md1=8)
        mid_out = gen.mid(mid_in)
        u = K.fold_mid(mid_out, md1=8, inverse=True, shape=(d, h, w))
        d0 = gen.decoders[0](e1, u)
        fc = gen.final_conv
        d1 = gen.decoders[1](e0, d0)
        pet = K.conv_out1(d1, fc.weight.detach().float().view(-1).contiguous(), float(fc.bias.item()))
        torch.cuda.synchronize()
        # ---- oracle chain
        xc = x.float().cpu()
        o_e0 = O.resnet_block(xc, sd, "encoders.0.basic_module.")
        o_e1 = O.resnet_block(F.max_pool3d(o_e0, 2), sd, "encoders.1.basic_module.")
        o_e2 = O.resnet_block(F.max_pool3d(o_e1, 2), sd, "encoders.2.basic_module.")
        patch = o_e2.shape[3]
        o_mo = O.vit_mid(O.fold_mid(o_e2), sd, "mid.", patch, vit_heads, vit_depth)
        o_d0 = O.decoder(o_e1, O.unfold_mid(o_mo, o_e2.shape[-1]), sd, "decoders.0.")
        o_d1 = O.decoder(o_e0, o_d0, sd, "decoders.1.")
        o_pet = F.conv3d(o_d1, sd["final_conv.weight"], sd["final_conv.bias"])
        # ---- oracle stages on the HIP path's own inputs
        h_e0, h_e1, h_e2, h_d0, h_d1 = (_ncdhw(t) for t in (e0, e1, e2, d0, d1))
        h_mo = mid_out.float().permute(0, 3, 1, 2).contiguous().cpu()                     # (B, H, W, C) channels-last -> (B, C, H, W)
        l_e1 = O.resnet_block(F.max_pool3d(h_e0, 2), sd, "encoders.1.basic_module.")
        l_e2 = O.resnet_block(F.max_pool3d(h_e1, 2), sd, "encoders.2.basic_module.")
        l_mo = O.vit_mid(O.fold_mid(h_e2), sd, "mid.", patch, vit_heads, vit_depth)
        l_d0 = O.decoder(h_e1, O.unfold_mid(h_mo, h_e2.shape[-1]), sd, "decoders.0.")
        l_d1 = O.decoder(h_e0, h_d0, sd, "decoders.1.")
        l_pet = F.conv3d(h_d1, sd["final_conv.weight"], sd["final_conv.bias"])
    for name, ours, loc, cum in (("encoders.0", h_e0, o_e0, o_e0), ("encoders.1", h_e1, l_e1, o_e1), ("encoders.2 (= mid_input)", h_e2, l_e2, o_e2),
                                 ("mid ViT (= mid_output)", h_mo, l_mo, o_mo), ("decoders.0", h_d0, l_d0, o_d0), ("decoders.1", h_d1, l_d1, o_d1),
                                 ("final_conv (= pet)", pet.float().cpu(), l_pet, o_pet)):
        rows.append((name, rel_err(ours, loc), rel_err(ours, cum)))
    return rows


@pytest.mark.parametrize("tag,vol,f_maps,vit,batch", [
    ("reduced 32^3 (T1 widths)", (32, 32, 32), (8, 16, 32), dict(dim=64, depth=2, heads=2, dim_head=16, mlp_dim=128), 2),
    ("full width 64^3", (64, 64, 64), (64, 128, 256), None, 2),
    ("full width 96^3 (the bench volume)", (96, 96, 96), (64, 128, 256), None, 1),
])
def test_generator_error_ladder(tag, vol, f_maps, vit, batch):
    import gfe_hip.det_init as det
    from pytorch3dunet.unet3d.model import Residual_mid_UNet3D_vit
    gen = Residual_mid_UNet3D_vit(1, 1, is_segmentation=False, f_maps=f_maps, vol_size=vol, vit_kwargs=vit)
    gen.load_state_dict(det.det_state_dict(gen.state_dict(), seed=41, prefix="ladder."))
    gen = gen.to(DEV).eval()
    x = det.det_inputs(batch, vol, seed=41)[0]
    rows = _ladder(gen, x, vit_heads=(vit or {}).get("heads", 6), vit_depth=(vit or {}).get("depth", 4))
    print("\nerror ladder, %s (rel = max |a-b| / max |b|):" % tag)
    for name, loc, cum in rows:
        print("  %-26s local %.2e   cumulative %.2e" % (name, loc, cum))
    for name, loc, cum in rows:
        assert loc < 1e-2, (name, loc)           # no single stage may spend the whole bf16 budget
    assert rows[-1][2] < 2e-2 and rows[3][2] < 2e-2
