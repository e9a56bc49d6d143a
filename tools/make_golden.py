#!/usr/bin/env python3
"""Generates tests/golden/*.npz by running the REFERENCE (imported from /root/reference) on CPU.

Build-container only: /root/reference does not exist on the GPU box and nothing under tests/, bench.py or
smoke() reads it.  Fixtures hold data only (inputs, weights for tiny cases, expected outputs); larger
cases regenerate their weights from gfe_hip/det_init.py on both sides.

    python tools/make_golden.py [--only t0|t1|t2|t3|t4|t5|t6|t7|t8|t9|t10|t11] [--out tests/golden]
"""
import argparse
import importlib.util
import os
import re
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"


def _load_det_init():
    spec = importlib.util.spec_from_file_location("det_init", os.path.join(ROOT, "gfe-mamba_amd", "gfe_hip", "det_init.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


det = _load_det_init()


def import_reference():
    """vit_pytorch_diy/dino.py:9 imports torchvision (absent here) -> stub it."""
    for name in ("torchvision", "torchvision.transforms"):
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.__path__ = []
            sys.modules[name] = m
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
    # utils/data_normalization.py:2-12 imports monai / nibabel (absent here) for its __main__ demo only; adaptive_normal uses torch alone
    for name in ("monai", "monai.transforms", "monai.utils", "nibabel"):
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.__path__ = []
            sys.modules[name] = m
    for n in ("Compose", "LoadImaged", "ToTensord", "EnsureChannelFirstd", "Spacingd", "ScaleIntensityRanged", "CropForegroundd", "Resized"):
        setattr(sys.modules["monai.transforms"], n, object)
    sys.path.insert(0, REF)
    import cross_atten.pscan as r_pscan
    import cross_atten.mamba as r_mamba
    import cross_atten.sd_cross_atten as r_xattn
    import cross_atten.corss_ft_transformer as r_ft
    import cross_atten.mamba_transformer as r_mt
    import classify.classifier as r_cls
    import pytorch3dunet.unet3d.model as r_model
    import pytorch3dunet.unet3d.buildingblocks as r_bb
    import vit_pytorch_diy.vit as r_vit
    import vit_pytorch_diy.vit_3d as r_vit3d
    import utils.data_normalization as r_norm
    return types.SimpleNamespace(norm=r_norm, pscan=r_pscan, mamba=r_mamba, xattn=r_xattn, ft=r_ft, mt=r_mt, cls=r_cls,
                                 model=r_model, bb=r_bb, vit=r_vit, vit3d=r_vit3d)


def npy(t):
    return t.detach().cpu().numpy()


def load_det(module, seed, prefix=""):
    sd = det.det_state_dict(module.state_dict(), seed=seed, prefix=prefix)
    module.load_state_dict(sd)
    return sd


def rnd(key, shape, seed=0, scale=1.0):
    g = np.random.Generator(np.random.Philox(key=[abs(hash(key)) % (2 ** 32), seed]))
    return torch.from_numpy((g.standard_normal(shape) * scale).astype(np.float32))


def rnd_det(key, shape, scale=1.0):
    import zlib
    g = np.random.Generator(np.random.Philox(key=[zlib.crc32(key.encode()), 12345]))
    return torch.from_numpy((g.standard_normal(shape) * scale).astype(np.float32))


# ------------------------------------------------------------------------------------------------
def t0(R, out):
    fx = {}
    # --- pscan forward + gradients (fp64 reference arithmetic on fp32 inputs)
    for L in (1, 2, 3, 4, 5, 37, 64, 100):
        A = torch.rand(2, L, 8, 4, generator=torch.Generator().manual_seed(L)).float() * 0.9 + 0.05
        X = rnd_det(f"pscan.X.{L}", (2, L, 8, 4))
        gH = rnd_det(f"pscan.gH.{L}", (2, L, 8, 4))
        A64 = A.double().requires_grad_(True)
        X64 = X.double().requires_grad_(True)
        H = R.pscan.pscan(A64, X64)
        H.backward(gH.double())
        fx[f"pscan_L{L}_A"], fx[f"pscan_L{L}_X"], fx[f"pscan_L{L}_gH"] = npy(A), npy(X), npy(gH)
        fx[f"pscan_L{L}_H"], fx[f"pscan_L{L}_gA"], fx[f"pscan_L{L}_gX"] = npy(H), npy(A64.grad), npy(X64.grad)
    np.savez_compressed(os.path.join(out, "t0_pscan.npz"), **fx)

    # --- MambaBlock.selective_scan / selective_scan_seq  (B,L,ED,N) = (2,37,64,16)
    fx = {}
    cfg = R.mamba.MambaConfig(d_model=32, n_layers=1)
    blk = R.mamba.MambaBlock(cfg)
    Bz, L, ED, N = 2, 37, 64, 16
    x = rnd_det("ss.x", (Bz, L, ED)).requires_grad_(True)
    delta = (torch.nn.functional.softplus(rnd_det("ss.delta", (Bz, L, ED)) - 3.0)).detach().requires_grad_(True)
    A = (-torch.exp(det.det_tensor("ss.A_log", (ED, N)))).requires_grad_(True)
    Bm = rnd_det("ss.B", (Bz, L, N)).requires_grad_(True)
    Cm = rnd_det("ss.C", (Bz, L, N)).requires_grad_(True)
    D = det.det_tensor("ss.D", (ED,)).requires_grad_(True)
    w = rnd_det("ss.w", (Bz, L, ED))
    y = blk.selective_scan(x, delta, A, Bm, Cm, D)
    y_seq = blk.selective_scan_seq(x, delta, A, Bm, Cm, D)
    (y * w).sum().backward()
    for k, v in dict(x=x, delta=delta, A=A, B=Bm, C=Cm, D=D, w=w, y=y, y_seq=y_seq).items():
        fx["ss_" + k] = npy(v)
    for k, v in dict(x=x, delta=delta, A=A, B=Bm, C=Cm, D=D).items():
        fx["ss_g" + k] = npy(v.grad)
    # plug-in contract (mamba.py:243-252): a torch callable with the selective_scan_fn layouts, installed at the slot
    z = rnd_det("ss.z", (Bz, L, ED))
    dbias = det.det_tensor("ss.dt_proj.bias", (ED,))
    draw = rnd_det("ss.draw", (Bz, L, ED))
    yfn = blk.selective_scan(x, torch.nn.functional.softplus(draw + dbias), A, Bm, Cm, D) * torch.nn.functional.silu(z)
    fx["ss_z"], fx["ss_dbias"], fx["ss_draw"], fx["ss_yfn"] = npy(z), npy(dbias), npy(draw), npy(yfn)
    np.savez_compressed(os.path.join(out, "t0_selective_scan.npz"), **fx)

    # --- MambaBlock.forward + RMSNorm + ResidualBlock (d_model 32)
    fx = {}
    torch.manual_seed(0)
    cfg = R.mamba.MambaConfig(d_model=32, n_layers=2)
    m = R.mamba.Mamba(cfg)
    sd = load_det(m, seed=1, prefix="t0mamba.")
    xin = rnd_det("mamba.x", (2, 37, 32)).requires_grad_(True)
    yb = m.layers[0].mixer(xin)
    yn = m.layers[0].norm(xin)
    ym = m(xin)
    w = rnd_det("mamba.w", (2, 37, 32))
    (ym * w).sum().backward()
    fx.update({"sd." + k: npy(v) for k, v in sd.items()})
    fx.update(x=npy(xin), w=npy(w), y_block0=npy(yb), y_norm0=npy(yn), y=npy(ym), gx=npy(xin.grad))
    fx.update({"g." + k: npy(p.grad) for k, p in m.named_parameters()})
    np.savez_compressed(os.path.join(out, "t0_mamba.npz"), **fx)

    # --- CrossAttention, GEGLU FeedForward, NumericalEmbedder, categories_offset
    fx = {}
    ca = R.xattn.CrossAttention(n_heads=2, d_embed=16, d_cross=24)
    sd = load_det(ca, seed=2, prefix="t0ca.")
    xq = rnd_det("ca.x", (3, 1, 16)).requires_grad_(True)
    yk = rnd_det("ca.y", (3, 6, 24)).requires_grad_(True)
    o = ca(xq, yk)
    w = rnd_det("ca.w", (3, 1, 16))
    (o * w).sum().backward()
    fx.update({"ca.sd." + k: npy(v) for k, v in sd.items()})
    fx.update({"ca.g." + k: npy(p.grad) for k, p in ca.named_parameters()})
    fx.update({"ca.x": npy(xq), "ca.y": npy(yk), "ca.w": npy(w), "ca.out": npy(o), "ca.gx": npy(xq.grad), "ca.gy": npy(yk.grad)})
    ff = R.ft.FeedForward(16, mult=2, dropout=0.1).eval()
    sd = load_det(ff, seed=3, prefix="t0ff.")
    xf = rnd_det("ff.x", (3, 1, 16)).requires_grad_(True)
    of = ff(xf)
    (of * w).sum().backward()
    fx.update({"ff.sd." + k: npy(v) for k, v in sd.items()})
    fx.update({"ff.g." + k: npy(p.grad) for k, p in ff.named_parameters()})
    fx.update({"ff.x": npy(xf), "ff.out": npy(of), "ff.gx": npy(xf.grad)})
    ne = R.ft.NumericalEmbedder(16, 5)
    sd = load_det(ne, seed=4, prefix="t0ne.")
    xn = rnd_det("ne.x", (3, 5))
    fx.update({"ne.sd." + k: npy(v) for k, v in sd.items()})
    fx.update({"ne.x": npy(xn), "ne.out": npy(ne(xn))})
    cm = R.mt.Cross_mamba_both(categories=(11, 2, 2, 4, 4, 3, 3), num_continuous=25, dim=16, depth=1, heads=2)
    fx["categories_offset"] = npy(cm.categories_offset)
    np.savez_compressed(os.path.join(out, "t0_head_ops.npz"), **fx)

    # --- ResNetBlock 'gcr', Decoder (deconv + nearest + sum), nearest index list
    fx = {}
    rb = R.bb.ResNetBlock(8, 16, kernel_size=3, order="gcr", num_groups=8).eval()
    sd = load_det(rb, seed=5, prefix="t0rb.")
    xr = rnd_det("rb.x", (2, 8, 8, 8, 8))
    fx.update({"rb.sd." + k: npy(v) for k, v in sd.items()})
    fx.update({"rb.x": npy(xr), "rb.out": npy(rb(xr))})
    rb2 = R.bb.ResNetBlock(16, 16, kernel_size=3, order="gcr", num_groups=8).eval()
    sd = load_det(rb2, seed=6, prefix="t0rb2.")
    xr2 = rnd_det("rb2.x", (2, 16, 6, 8, 4))
    fx.update({"rb2.sd." + k: npy(v) for k, v in sd.items()})
    fx.update({"rb2.x": npy(xr2), "rb2.out": npy(rb2(xr2))})
    dec = R.bb.Decoder(16, 8, basic_module=R.bb.ResNetBlock, conv_layer_order="gcr", num_groups=8, upsample="default").eval()
    sd = load_det(dec, seed=7, prefix="t0dec.")
    xd = rnd_det("dec.x", (2, 16, 4, 4, 4))
    ef = rnd_det("dec.ef", (2, 8, 8, 8, 8))
    up = dec.upsampling(encoder_features=ef, x=xd)
    fx.update({"dec.sd." + k: npy(v) for k, v in sd.items()})
    fx.update({"dec.x": npy(xd), "dec.ef": npy(ef), "dec.up": npy(up), "dec.out": npy(dec(ef, xd))})
    for n in (3, 4, 12, 24, 47):
        src = torch.arange(2 * n - 1, dtype=torch.float32).view(1, 1, -1, 1, 1).expand(1, 1, -1, 2, 2)
        r = torch.nn.functional.interpolate(src, size=(2 * n, 2, 2))
        fx[f"nearest_idx_{n}"] = npy(r[0, 0, :, 0, 0]).astype(np.int64)
    mp = torch.nn.MaxPool3d(2)
    xm = rnd_det("mp.x", (1, 4, 6, 8, 10))
    fx.update({"mp.x": npy(xm), "mp.out": npy(mp(xm))})
    np.savez_compressed(os.path.join(out, "t0_unet_ops.npz"), **fx)

    # --- ViT (image (64,8), patch 8, channels 32) incl. from_patch_embedding
    fx = {}
    vit = R.vit.ViT(image_size=(64, 8), patch_size=8, dim=64, depth=2, heads=2, dim_head=16, mlp_dim=128,
                    channels=32, dropout=0.1, emb_dropout=0.1).eval()
    sd = load_det(vit, seed=8, prefix="t0vit.")
    xi = rnd_det("vit.x", (2, 32, 64, 8))
    fx.update({"sd." + k: npy(v) for k, v in sd.items()})
    fx.update(x=npy(xi), out=npy(vit(xi)), tokens=npy(vit.to_patch_embedding(xi)))
    np.savez_compressed(os.path.join(out, "t0_vit.npz"), **fx)

    # --- vit_3d.ViT (the synthetic MFMA-attention row of SURVEY 8-d, reduced): 2 channels, 16^3 volume, 8^3 patches, 9 tokens,
    # plus a 16x16x48 volume with 4x4x8 patches (97 tokens: more than one key tile, ragged)
    for tag, kw, shape in (("a", dict(image_size=16, image_patch_size=8, frames=16, frame_patch_size=8, channels=2), (2, 2, 16, 16, 16)),
                           ("b", dict(image_size=16, image_patch_size=4, frames=48, frame_patch_size=8, channels=1), (1, 1, 48, 16, 16))):
        fx = {}
        v3 = R.vit3d.ViT(num_classes=3, dim=128, depth=2, heads=2, dim_head=64, mlp_dim=256, pool="cls", **kw).eval()
        sd = load_det(v3, seed=9, prefix="t0vit3d" + tag + ".")
        xi = rnd_det("vit3d.x" + tag, shape)
        fx.update({"sd." + k: npy(v) for k, v in sd.items()})
        tok = v3.to_patch_embedding(xi)
        fx.update(x=npy(xi), out=npy(v3(xi)), tokens=npy(tok))
        v3.pool = "mean"
        fx.update(out_mean=npy(v3(xi)))
        np.savez_compressed(os.path.join(out, "t0_vit3d_" + tag + ".npz"), **fx)

    # --- pure index maps, as exact integers
    from einops import rearrange
    fx = {}
    for name, shp in (("native", (40, 40, 24)), ("g96", (24, 24, 24)), ("g128", (32, 32, 32)), ("g32", (8, 8, 8))):
        n = int(np.prod(shp))
        src = torch.arange(n, dtype=torch.int64).view(1, 1, *shp)
        folded = rearrange(src, "b c (md1 md2) h w -> b c (h md1) (md2 w)", md1=8)
        back = rearrange(folded, "b c (h md1) (md2 w) -> b c (md1 md2) h w", md1=8, w=shp[2])
        assert torch.equal(back, src)
        fx[f"fold_{name}"] = npy(folded[0, 0])
    img = torch.arange(2 * 16 * 8, dtype=torch.int64).view(1, 2, 16, 8)
    pat = rearrange(img, "b c (h p1) (w p2) -> b (h w) (p1 p2 c)", p1=4, p2=4)
    fx["patchify_c2_16x8_p4"] = npy(pat[0])
    unp = rearrange(pat, "b (h w) (p1 p2 c) -> b c (h p1) (w p2)", p1=4, p2=4, h=4)
    assert torch.equal(unp, img)
    vol = torch.arange(2 * 1 * 4 * 6 * 8, dtype=torch.int64).view(2, 1, 4, 6, 8)
    cond = rearrange(vol, "b c h w d -> (b c) (h w) d").transpose(1, 2).contiguous()
    fx["condition_2x1x4x6x8"] = npy(cond)
    np.savez_compressed(os.path.join(out, "t0_index_maps.npz"), **fx)


# ------------------------------------------------------------------------------------------------
def build_reference_models(R, vol, f_maps, dim, depth, heads, vit_dim, vit_depth, vit_heads, vit_dim_head, vit_mlp,
                           cards, n_cont, seed):
    import torch.nn as nn
    D1, D2, D3 = vol
    H, W, p = (D2 // 4) * 8, (D1 // 32) * (D3 // 4), D2 // 4
    gen = R.model.Residual_mid_UNet3D_vit(1, 1, is_segmentation=False, f_maps=f_maps)
    gen.mid = R.vit.ViT(image_size=(H, W), patch_size=p, dim=vit_dim, depth=vit_depth, heads=vit_heads, dim_head=vit_dim_head,
                        mlp_dim=vit_mlp, dropout=0.1, emb_dropout=0.1, channels=f_maps[-1])
    head = R.cls.Combine_classfier_vit_mid(seq_length=4)
    head.vit_mid_linear = nn.Linear(H * W, 4)
    ft = R.mt.Cross_mamba_both(categories=cards, num_continuous=n_cont, dim=dim, depth=depth, heads=heads, dim_head=dim // heads)
    ft.final_cross = R.xattn.CrossAttention(n_heads=heads, d_embed=dim, d_cross=D1 * D2)
    load_det(gen, seed, "gen.")
    load_det(head, seed, "head.")
    load_det(ft, seed, "ft.")
    return gen.eval(), head, ft


def slices(t, n=256):
    f = t.detach().reshape(-1)
    step = max(1, f.numel() // n)
    return npy(f[::step][:n].double())


def sample_idx(numel, n=1024):
    """The element sample of the end-to-end gradient statistics (VERDICT r05 #6a): n indices drawn without replacement by a generator
    seeded with the tensor's size -- tests/test_head_gpu.py rebuilds the same indices; every element of a tensor up to n elements."""
    if numel <= n:
        return np.arange(numel)
    return np.sort(np.random.default_rng(numel).choice(numel, size=n, replace=False))


def run_step(R, gen, head, ft, vol, B, cards, n_cont, seed, full_grads):
    x, x_cat, x_num, y = det.det_inputs(B, vol, cards, n_cont, seed=seed)
    head.eval(); ft.eval()     # dropout off: parity fixtures are deterministic
    with torch.no_grad():
        mid_in, mid_out, pet = gen(x, output_vit_mid=True)
    feat = head(mid_in, mid_out)
    pred = ft(x_cat, x_num, feat, [x, pet])
    loss = torch.nn.BCELoss()(torch.sigmoid(pred.squeeze(1)), y.float())
    loss.backward()
    params = [("head." + k, p) for k, p in head.named_parameters()] + [("ft." + k, p) for k, p in ft.named_parameters()]
    fx = dict(pred=npy(pred.double()), loss=npy(loss.double()), feat=npy(feat.double()) if full_grads else slices(feat))
    for name, t in (("mid_input", mid_in), ("mid_output", mid_out), ("pet", pet)):
        fx[name + "_slice"] = slices(t)
        fx[name + "_sum"] = npy(t.double().sum())
        fx[name + "_abssum"] = npy(t.double().abs().sum())
        if full_grads:
            fx[name] = npy(t)
    for k, p in params:
        fx["gnorm." + k] = npy(p.grad.double().norm())
        fx["gslice." + k] = slices(p.grad, 64)
        fx["gsample." + k] = p.grad.detach().reshape(-1)[torch.from_numpy(sample_idx(p.grad.numel()))].numpy().astype(np.float32)
        fx["gamax." + k] = npy(p.grad.double().abs().max())
    # per-parameter clip (classify_mamba.py:106-107) then one Adam step (:64, :108)
    allp = [p for _, p in params]
    opt = torch.optim.Adam(allp, lr=1e-4)
    for p in allp:
        torch.nn.utils.clip_grad_norm_(p, max_norm=1.0)
    before = [p.detach().clone() for p in allp]
    opt.step()
    for (k, p), b0 in zip(params, before):
        fx["dnorm." + k] = npy((p.detach() - b0).double().norm())
        fx["dslice." + k] = slices(p.detach() - b0, 64)
    return fx


def t1(R, out):
    cards, n_cont, vol = (11, 2, 2, 4, 4, 3, 3), 25, (32, 32, 32)
    gen, head, ft = build_reference_models(R, vol, (8, 16, 32), 64, 2, 8, 64, 2, 2, 16, 128, cards, n_cont, seed=11)
    fx = run_step(R, gen, head, ft, vol, 2, cards, n_cont, seed=11, full_grads=True)
    fx["meta"] = np.array([32, 32, 32, 8, 16, 32, 64, 2, 8, 64, 2, 2, 16, 128, 11])
    np.savez_compressed(os.path.join(out, "t1_reduced_step.npz"), **fx)


def t2(R, out):
    cards, n_cont, vol = (11, 2, 2, 4, 4, 3, 3), 25, (96, 96, 96)
    gen, head, ft = build_reference_models(R, vol, (64, 128, 256), 512, 6, 8, 512, 4, 6, 64, 2048, cards, n_cont, seed=21)
    fx = run_step(R, gen, head, ft, vol, 2, cards, n_cont, seed=21, full_grads=False)
    fx["meta"] = np.array([96, 96, 96, 64, 128, 256, 512, 6, 8, 512, 4, 6, 64, 2048, 21])
    np.savez_compressed(os.path.join(out, "t2_full96_step.npz"), **fx)


def t7(R, out):
    """The reference's OWN geometry (config/classify_mamba_config.yaml:5-7: 160x160x96 volumes), the three modules built with exactly
    the constructor calls of classify_mamba.py:36-56 -- nothing re-instantiated: image_size (320,120) / patch 40 (model.py:107-117),
    Linear(320*120, 4) (classifier.py:327), d_cross = 160*160 (mamba_transformer.py:84) -- one sample, eval mode.  Also writes the
    state-dict key -> shape listing of the three default-geometry modules (the layout the authors' checkpoints have)."""
    import json
    cards, n_cont, vol = (11, 2, 2, 4, 4, 3, 3), 25, (160, 160, 96)
    gen = R.model.Residual_mid_UNet3D_vit(1, 1, is_segmentation=False, f_maps=(64, 128, 256))
    head = R.cls.Combine_classfier_vit_mid(seq_length=4)
    ft = R.mt.Cross_mamba_both(categories=cards, num_continuous=n_cont, dim=512, dim_out=1, depth=6, heads=8, attn_dropout=0.1,
                               ff_dropout=0.1, dim_head=512 // 8)
    listing = {name: {k: [list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in m.state_dict().items()}
               for name, m in (("gen", gen), ("head", head), ("ft", ft))}
    with open(os.path.join(out, "t7_native_state_dict.json"), "w") as f:
        json.dump(listing, f, indent=0, sort_keys=True)
    load_det(gen, 71, "gen.")
    load_det(head, 71, "head.")
    load_det(ft, 71, "ft.")
    fx = run_step(R, gen.eval(), head, ft, vol, 1, cards, n_cont, seed=71, full_grads=False)
    fx["meta"] = np.array([160, 160, 96, 64, 128, 256, 512, 6, 8, 512, 4, 6, 64, 2048, 71])
    np.savez_compressed(os.path.join(out, "t7_native_step.npz"), **fx)


def t9(R, out):
    """The reference's OWN autograd through its generator (main_gan_vit.py:68-82 minus the third-party losses): L1(model(mri), pet) backward
    through Residual_mid_UNet3D_vit at reduced width (f_maps 8/16/32, ViT 64 x 2 x 2) on 32^3, eval mode (dropout off).  Pins the BACKWARD of
    the oracle's generator restatement (whose forward t1 / t2 pin): loss, pet slices, and per-parameter gradient norms + slices."""
    import torch.nn.functional as F
    vol = (32, 32, 32)
    gen = R.model.Residual_mid_UNet3D_vit(1, 1, is_segmentation=False, f_maps=(8, 16, 32))
    H, W, p = (vol[1] // 4) * 8, (vol[0] // 32) * (vol[2] // 4), vol[1] // 4
    gen.mid = R.vit.ViT(image_size=(H, W), patch_size=p, dim=64, depth=2, heads=2, dim_head=16, mlp_dim=128, dropout=0.1, emb_dropout=0.1, channels=32)
    load_det(gen, 51, "gtrain.")
    gen.eval()
    x = det.det_inputs(2, vol, seed=51)[0]
    target = torch.tanh(torch.randn(2, 1, *vol, generator=torch.Generator().manual_seed(52)))
    pet = gen(x)
    loss = F.l1_loss(pet, target)
    loss.backward()
    fx = dict(loss=npy(loss.double()), pet_slice=slices(pet), pet_abssum=npy(pet.double().abs().sum()))
    for k, prm in gen.named_parameters():
        if prm.grad is None:
            fx["nograd." + k] = np.zeros(1)
            continue
        fx["gnorm." + k] = npy(prm.grad.double().norm())
        fx["gslice." + k] = slices(prm.grad, 64)
    np.savez_compressed(os.path.join(out, "t9_generator_grads.npz"), **fx)


def t10(R, out):
    """The reference's vit_3d.ViT under ITS autograd (vit_3d.py:47-57 trains through `dots` / `attn` / matmul): cross-entropy of the logits
    against fixed labels, backward; logits, loss, the input gradient and every parameter gradient (norm + strided slice; the to_qkv / to_out
    gradients of layer 0 in full).  Two geometries with dim_head 64: 97 tokens (two key tiles, ragged) and 321 tokens (two 256-row query
    blocks, six key tiles, ragged) -- the flash-attention backward's tile edges."""
    import torch.nn.functional as F
    for tag, kw, shape in (("b", dict(image_size=16, image_patch_size=4, frames=48, frame_patch_size=8, channels=1), (2, 1, 48, 16, 16)),
                           ("c", dict(image_size=32, image_patch_size=4, frames=40, frame_patch_size=8, channels=1), (2, 1, 40, 32, 32))):
        v3 = R.vit3d.ViT(num_classes=3, dim=128, depth=2, heads=2, dim_head=64, mlp_dim=256, pool="cls", **kw)
        sd = load_det(v3, seed=10, prefix="t10vit3d" + tag + ".")
        v3.train()                                       # dropout p = 0: train and eval coincide
        xi = rnd_det("t10.x" + tag, shape).requires_grad_()
        labels = torch.tensor([2, 0])
        out_ = v3(xi)
        loss = F.cross_entropy(out_, labels)
        loss.backward()
        fx = {"sd." + k: npy(v) for k, v in sd.items()}
        fx.update(x=npy(xi), labels=labels.numpy(), out=npy(out_), loss=npy(loss.double()), dx_slice=slices(xi.grad), dx_norm=npy(xi.grad.double().norm()))
        for k, prm in v3.named_parameters():
            fx["gnorm." + k] = npy(prm.grad.double().norm())
            fx["gslice." + k] = slices(prm.grad, 128)
        for k in ("transformer.layers.0.0.to_qkv.weight", "transformer.layers.0.0.to_out.0.weight"):
            fx["gfull." + k] = npy(dict(v3.named_parameters())[k].grad)
        np.savez_compressed(os.path.join(out, "t10_vit3d_grads_" + tag + ".npz"), **fx)


def t3(R, out):
    """Cross_mamba_ablation (cross_atten/mamba_transformer.py:254-385): the four forward variants + parameter gradients of each."""
    cards, n_cont, dim, depth, heads, vol, Bn = (5, 3, 2), 6, 64, 2, 8, (8, 12, 6), 3
    ft = R.mt.Cross_mamba_ablation(categories=cards, num_continuous=n_cont, dim=dim, depth=depth, heads=heads, dim_head=dim // heads)
    ft.final_cross = R.xattn.CrossAttention(n_heads=heads, d_embed=dim, d_cross=vol[0] * vol[1])      # :325 hard-codes 160*160
    load_det(ft, 31, "abl.")
    ft.eval()
    x, x_cat, x_num, y = det.det_inputs(Bn, vol, cards, n_cont, seed=31)
    pet = rnd_det("abl.pet", (Bn, 1) + vol)
    feat = rnd_det("abl.feat", (Bn, 4, dim))
    fx = dict(meta=np.array(list(cards) + [n_cont, dim, depth, heads] + list(vol) + [Bn]), pet=npy(pet), feat=npy(feat))
    cases = dict(full=dict(feature_img=feat, image_condition=[x, pet]), table_only=dict(feature_img=None, image_condition=[x, pet]),
                 no_table=dict(feature_img=feat, image_condition=[x, pet], no_table=True), no_cross=dict(feature_img=feat, image_condition=None))
    for name, kw in cases.items():
        ft.zero_grad()
        pred = ft(x_cat, x_num, **kw)
        loss = torch.nn.BCELoss()(torch.sigmoid(pred.squeeze(1)), y.float())
        loss.backward()
        fx[name + ".pred"] = npy(pred.double())
        fx[name + ".loss"] = npy(loss.double())
        for k, p in ft.named_parameters():
            fx[name + ".gnorm." + k] = npy(p.grad.double().norm()) if p.grad is not None else np.array(-1.0)
    np.savez_compressed(os.path.join(out, "t3_ablation.npz"), **fx)


def t4(R, out):
    """adaptive_normal (utils/data_normalization.py:20-48) on volumes with negatives, ties, signed zeros, a constant volume and infinities."""
    g = np.random.Generator(np.random.Philox(key=[404, 1]))
    cases = {}
    cases["normal"] = g.standard_normal((20, 24, 16)).astype(np.float32) * 300 + 100
    q = np.round(g.standard_normal((16, 16, 12)) * 4).astype(np.float32)          # heavy ties, exact zeros
    q[0, 0, :4] = -0.0
    cases["ties"] = q
    cases["positive"] = g.exponential(500.0, (32, 32, 32)).astype(np.float32)
    cases["constant"] = np.full((4, 5, 6), 7.5, dtype=np.float32)
    e = g.standard_normal((8, 8, 8)).astype(np.float32)
    e[1, 2, 3] = np.inf; e[2, 2, 2] = -np.inf; e[3, 3, 3] = np.nan
    cases["nonfinite"] = e
    cases["single"] = np.array([[[3.0]]], dtype=np.float32)
    cases["tiny_mixed"] = np.array([-5.0, 0.0, 2.0, -1.0, 9.0, 4.0, 4.0], dtype=np.float32).reshape(7, 1, 1)
    fx = {}
    for name, x in cases.items():
        y = R.norm.adaptive_normal(torch.from_numpy(x.copy()))
        fx[name + ".x"] = x
        fx[name + ".y"] = npy(y)
    np.savez_compressed(os.path.join(out, "t4_adaptive_normal.npz"), **fx)


def t6(R, out):
    """Cross_jamba_both (cross_atten/mamba_transformer.py:135-251) on the Jamba backbone (jamba.py:258-535): depth 3 -> 6 layers (attention
    at layer 4, 16-expert top-2 MoE on layers 1, 3, 5, Mamba mixers with inner layernorms elsewhere): logits, loss, per-parameter
    gradient norms and slices.  Weights are committed (257 small tensors): the router's top-2 choice must be reproduced exactly."""
    cards, n_cont, dim, depth, heads, vol, Bn = (5, 3, 2), 6, 64, 3, 8, (8, 12, 6), 3
    ft = R.mt.Cross_jamba_both(categories=cards, num_continuous=n_cont, dim=dim, depth=depth, heads=heads, dim_head=dim // heads)
    ft.final_cross = R.xattn.CrossAttention(n_heads=heads, d_embed=dim, d_cross=vol[0] * vol[1])      # :200 hard-codes 160*160
    load_det(ft, 41, "jam.")
    ft.eval()
    x, x_cat, x_num, y = det.det_inputs(Bn, vol, cards, n_cont, seed=41)
    pet = rnd_det("jam.pet", (Bn, 1) + vol)
    feat = rnd_det("jam.feat", (Bn, 4, dim))
    pred = ft(x_cat, x_num, feat, [x, pet])
    loss = torch.nn.BCELoss()(torch.sigmoid(pred.squeeze(1)), y.float())
    loss.backward()
    fx = dict(meta=np.array(list(cards) + [n_cont, dim, depth, heads] + list(vol) + [Bn]), pet=npy(pet), feat=npy(feat),
              pred=npy(pred.double()), loss=npy(loss.double()))
    for k, p in ft.named_parameters():
        fx["gnorm." + k] = npy(p.grad.double().norm()) if p.grad is not None else np.array(-1.0)
        if p.grad is not None:
            fx["gslice." + k] = slices(p.grad, 64)
    np.savez_compressed(os.path.join(out, "t6_jamba.npz"), **fx)


def t8(R, out):
    """Cross_jamba_both at the CLASSIFY configuration (classify_mamba.py:14, 36-51 with Cross_jamba_both in place of Cross_mamba_both:
    dim 512, depth 6 -> 12 Jamba layers, heads 8, 16-expert top-2 MoE on the odd layers: 208.5 M parameters), default constructor
    (d_cross = 160*160), two samples with native-size 160x160x96 image conditions.  Nothing but results is stored: weights and inputs
    regenerate from the deterministic initialiser on both sides."""
    cards, n_cont, dim, depth, heads, vol, Bn = (11, 2, 2, 4, 4, 3, 3), 25, 512, 6, 8, (160, 160, 96), 2
    ft = R.mt.Cross_jamba_both(categories=cards, num_continuous=n_cont, dim=dim, depth=depth, heads=heads, dim_head=dim // heads)
    load_det(ft, 81, "jam8.")
    ft.eval()
    x, x_cat, x_num, y = det.det_inputs(Bn, vol, cards, n_cont, seed=81)
    pet = rnd_det("jam8.pet", (Bn, 1) + vol)
    feat = rnd_det("jam8.feat", (Bn, 4, dim))
    pred = ft(x_cat, x_num, feat, [x, pet])
    loss = torch.nn.BCELoss()(torch.sigmoid(pred.squeeze(1)), y.float())
    loss.backward()
    fx = dict(meta=np.array(list(cards) + [n_cont, dim, depth, heads] + list(vol) + [Bn]), pred=npy(pred.double()), loss=npy(loss.double()),
              nparams=np.array(sum(p.numel() for p in ft.parameters())))
    for k, p in ft.named_parameters():
        fx["gnorm." + k] = npy(p.grad.double().norm()) if p.grad is not None else np.array(-1.0)
        if p.grad is not None:
            fx["gslice." + k] = slices(p.grad, 16).astype(np.float32)
    np.savez_compressed(os.path.join(out, "t8_jamba_classify.npz"), **fx)


def t5(R, out):
    """table/deal_table.py:28-61 `prepare_table` on a synthetic TADPOLE-like frame: bookkeeping and baseline columns to drop, string
    categoricals (with missing values and a numeric-looking string column that contains letters), numeric columns with missing and
    unparsable entries, a constant column.  The input travels as a CSV data file, the reference's outputs as arrays."""
    import importlib
    import pandas as pd
    dt = importlib.import_module("table.deal_table")
    g = np.random.default_rng(5)
    n = 40
    df = pd.DataFrame({
        "RID": np.arange(n), "PTID": [f"0{i % 7:02d}_S_{1000 + i % 7}" for i in range(n)],
        "EXAMDATE": [f"20{10 + i % 5}-0{1 + i % 9}-1{i % 9}" for i in range(n)], "LABEL": g.integers(0, 2, n).astype(float),
        "D2": 0, "SITE": g.integers(1, 60, n), "DX": g.choice(["CN", "MCI", "Dementia"], n), "COLPROT": "ADNI2", "ORIGPROT": "ADNI1",
        "Month": g.integers(0, 60, n), "M": g.integers(0, 60, n), "FDG": g.normal(1.2, 0.1, n), "PIB": np.nan, "AV45": g.normal(1.1, 0.2, n),
        "AGE": g.normal(72, 6, n).round(1), "PTGENDER": g.choice(["Male", "Female"], n), "PTEDUCAT": g.integers(8, 21, n),
        "PTETHCAT": g.choice(["Not Hisp/Latino", "Hisp/Latino", "Unknown"], n), "PTMARRY": g.choice(["Married", "Widowed", "Divorced", None], n),
        "APOE4": g.choice([0.0, 1.0, 2.0, np.nan], n), "ABETA": g.choice([">1700", "912.3", "1200", None], n),
        "TAU": g.choice(["210.5", "<80", "300", "155.1"], n), "MMSE": g.integers(18, 31, n).astype(float), "CDRSB": g.choice([0.0, 0.5, 1.0, 2.5], n),
        "Hippocampus": g.normal(7000, 900, n).round(0), "ICV": 1.5e6, "AGE_bl": 70.0, "MMSE_bl": 28.0, "Years_bl": g.random(n),
    })
    df.loc[3, "MMSE"] = np.nan
    df.loc[5, "Hippocampus"] = np.nan
    csv = os.path.join(out, "t5_table_input.csv")
    df.to_csv(csv, index=False)
    ref = dt.prepare_table(pd.read_csv(csv))                   # (the reference reads the table from a CSV too: pic_table_loader.py:64)
    fx = dict(cate_x=ref["cate_x"].to_numpy(dtype=np.int64), conti_x=ref["conti_x"].to_numpy(dtype=np.float64),
              num_cat=np.asarray(ref["num_cat"], dtype=np.int64), num_cont=np.asarray(ref["num_cont"]),
              cate_cols=np.asarray(list(ref["cate_x"].columns)), conti_cols=np.asarray(list(ref["conti_x"].columns)),
              info_cols=np.asarray(list(ref["info"].columns)))
    np.savez_compressed(os.path.join(out, "t5_table.npz"), **fx)


def t11(R, out):
    """dataloader/pic_table_loader.py:46-127 `MRI_classify`: which files survive the constructor's filter (the reference pops from the list
    it enumerates: kept bug for bug on our side), which table row every file is matched to (patient id, label, nearest EXAMDATE within 30
    days) and the label / cate_x / conti_x of every sample.  The image transforms are monai / nibabel (absent): stubbed, not pinned here.
    Inputs travel as data files: tests/golden/t11_table_input.csv and the file-name list inside t11_dataset.json."""
    import importlib
    import json
    import tempfile
    import pandas as pd
    # utils/common.py:6-8 imports torchvision.utils / matplotlib (absent) for its plotting helpers; date_difference needs neither
    for name in ("torchvision.utils", "matplotlib", "matplotlib.pyplot"):
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.__path__ = []
            sys.modules[name] = m
    sys.modules["torchvision.utils"].make_grid = None
    sys.modules["matplotlib"].pyplot = sys.modules["matplotlib.pyplot"]

    class _T:                                   # a transform that passes its argument through (LoadImaged, Compose, ...)
        def __init__(self, *a, **k):
            pass

        def __call__(self, x):
            return x
    for n in ("Compose", "LoadImaged", "ToTensord", "EnsureChannelFirstd", "Spacingd", "ScaleIntensityRanged", "CropForegroundd", "Resized"):
        setattr(sys.modules["monai.transforms"], n, _T)
    sys.modules["monai.utils"].first = None
    ptl = importlib.import_module("dataloader.pic_table_loader")
    g = np.random.default_rng(11)
    base = pd.read_csv(os.path.join(out, "t5_table_input.csv"))
    base["date_diff"] = g.integers(-5, 400, len(base))
    base.loc[7, "LABEL"] = np.nan
    csv = os.path.join(out, "t11_table_input.csv")
    base.to_csv(csv, index=False)
    names = []
    for i in range(len(base)):
        r = base.iloc[i]
        y, m, d = r["EXAMDATE"].split("-")
        lab = 0 if pd.isna(r["LABEL"]) else int(r["LABEL"])
        shift = int(g.integers(0, 45))                                   # some scans fall outside the 30-day window
        day = min(28, int(d) + shift % 12)
        if i % 3 != 2:
            names.append(f"{r['PTID']}-{y}_{m}_{day:02d}-{lab}.nii.gz")
        if i % 5 == 0:
            names.append(f"{r['PTID']}-{int(y) + 3}_{m}_{d}-{lab}.nii.gz")       # years away: no match
        if i % 7 == 0:
            names.append(f"{r['PTID']}-{y}_{m}_{d}-{1 - lab}.nii.gz")            # the other label
    names = sorted(set(names))
    fx = {"names": names, "cases": {}}
    for thr in (-1, 30, 200):
        with tempfile.TemporaryDirectory() as d:
            for n in names:
                open(os.path.join(d, n), "wb").close()
            real_glob = ptl.glob
            ptl.glob = lambda pat: sorted(real_glob(pat))                # (file-system order is not part of the contract)
            try:
                ds = ptl.MRI_classify(d, csv, (8, 8, 8), days_threshold=thr)
            finally:
                ptl.glob = real_glob
            kept = [os.path.basename(p) for p in ds.mri_nii]
            rows, labels, cate, conti = [], [], [], []
            for n in kept:
                found, idx = ds.find_index(n, ds.table_df["info"])
                rows.append([bool(found), int(idx)])
                labels.append(int(re.findall('-(\\d).nii.gz', n)[0]))
                cate.append([int(v) for v in ds.table_df["cate_x"].iloc[idx].values])
                conti.append([float(v) for v in ds.table_df["conti_x"].iloc[idx].values])
            fx["cases"][str(thr)] = dict(kept=kept, rows=rows, labels=labels, cate_x=cate, conti_x=conti)
    json.dump(fx, open(os.path.join(out, "t11_dataset.json"), "w"))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden"))
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    torch.set_grad_enabled(True)
    R = import_reference()
    for name, fn in (("t0", t0), ("t1", t1), ("t2", t2), ("t3", t3), ("t4", t4), ("t5", t5), ("t6", t6), ("t7", t7), ("t8", t8), ("t9", t9), ("t10", t10), ("t11", t11)):
        if not a.only or a.only == name:
            fn(R, a.out)
            print("wrote", name)
