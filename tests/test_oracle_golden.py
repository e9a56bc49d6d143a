"""Pins the oracle (oracle/ref_ops.py) against outputs of the reference itself (tests/golden/, made by
tools/make_golden.py in the build container).  CPU only."""
import numpy as np
import pytest
import torch

from conftest import golden, rel_err, sub_sd, tt
from oracle import ref_ops as O

TOL = 2e-5   # fp32 arithmetic both sides; different summation order only


@pytest.mark.parametrize("L", [1, 2, 3, 4, 5, 37, 64, 100])
def test_pscan_matches_reference(L):
    fx = golden("t0_pscan.npz")
    A, X, gH = (tt(fx[f"pscan_L{L}_{k}"], torch.float64) for k in ("A", "X", "gH"))
    H = O.pscan(A, X)
    assert rel_err(H, tt(fx[f"pscan_L{L}_H"])) < 1e-12
    gA, gX = O.pscan_grads(A, H, gH)
    assert rel_err(gX, tt(fx[f"pscan_L{L}_gX"])) < 1e-12
    assert (gA - tt(fx[f"pscan_L{L}_gA"])).abs().max() < 1e-12
    assert torch.all(gA[:, 0] == 0)


def test_selective_scan_matches_reference():
    fx = golden("t0_selective_scan.npz")
    g = lambda k: tt(fx["ss_" + k]).requires_grad_(True)
    x, delta, A, B, C, D = g("x"), g("delta"), g("A"), g("B"), g("C"), g("D")
    y = O.selective_scan(x, delta, A, B, C, D)
    assert rel_err(y, tt(fx["ss_y"])) < TOL
    assert rel_err(y, tt(fx["ss_y_seq"])) < TOL
    (y * tt(fx["ss_w"])).sum().backward()
    for k, v in dict(x=x, delta=delta, A=A, B=B, C=C, D=D).items():
        assert rel_err(v.grad, tt(fx["ss_g" + k])) < 1e-4, k
    # plug-in layout contract (mamba.py:243-252)
    tr = lambda t: t.detach().transpose(1, 2)
    yfn = O.selective_scan_fn(tr(x), tr(tt(fx["ss_draw"])), A.detach(), tr(B), tr(C), D.detach(), z=tr(tt(fx["ss_z"])),
                              delta_bias=tt(fx["ss_dbias"]), delta_softplus=True)
    assert rel_err(yfn.transpose(1, 2), tt(fx["ss_yfn"])) < TOL


def test_mamba_block_and_stack():
    fx = golden("t0_mamba.npz")
    sd = sub_sd(fx, "sd.")
    x = tt(fx["x"])
    assert rel_err(O.rmsnorm(x, sd["layers.0.norm.weight"]), tt(fx["y_norm0"])) < TOL
    assert rel_err(O.mamba_block(x, sd, "layers.0.mixer."), tt(fx["y_block0"])) < TOL
    sdg = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    xg = x.clone().requires_grad_(True)
    y = O.mamba(xg, sdg, "", 2)
    assert rel_err(y, tt(fx["y"])) < TOL
    (y * tt(fx["w"])).sum().backward()
    assert rel_err(xg.grad, tt(fx["gx"])) < 1e-4
    for k, v in sdg.items():
        assert rel_err(v.grad, tt(fx["g." + k])) < 2e-4, k


def test_head_ops():
    fx = golden("t0_head_ops.npz")
    sd = sub_sd(fx, "ca.sd.")
    x, y = tt(fx["ca.x"]).requires_grad_(True), tt(fx["ca.y"]).requires_grad_(True)
    sdg = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    o = O.cross_attention(x, y, sdg, "", 2)
    assert rel_err(o, tt(fx["ca.out"])) < TOL
    (o * tt(fx["ca.w"])).sum().backward()
    assert rel_err(x.grad, tt(fx["ca.gx"])) < 1e-4 and rel_err(y.grad, tt(fx["ca.gy"])) < 1e-4
    for k, v in sdg.items():
        if k == "k_proj.bias":     # exactly zero in exact arithmetic (softmax is shift-invariant): both sides are round-off
            assert v.grad.abs().max() < 1e-6 and np.abs(fx["ca.g." + k]).max() < 1e-6
            continue
        assert rel_err(v.grad, tt(fx["ca.g." + k])) < 1e-4, k
    sd = sub_sd(fx, "ff.sd.")
    assert rel_err(O.geglu_ff(tt(fx["ff.x"]), sd, ""), tt(fx["ff.out"])) < TOL
    sd = sub_sd(fx, "ne.sd.")
    ne = tt(fx["ne.x"]).unsqueeze(-1) * sd["weights"] + sd["biases"]
    assert rel_err(ne, tt(fx["ne.out"])) < TOL
    off = O.categories_offset((11, 2, 2, 4, 4, 3, 3))
    assert off.tolist() == fx["categories_offset"].tolist() == [2, 13, 15, 17, 21, 25, 28]


def test_unet_ops():
    fx = golden("t0_unet_ops.npz")
    assert rel_err(O.resnet_block(tt(fx["rb.x"]), sub_sd(fx, "rb.sd."), ""), tt(fx["rb.out"])) < TOL
    assert rel_err(O.resnet_block(tt(fx["rb2.x"]), sub_sd(fx, "rb2.sd."), ""), tt(fx["rb2.out"])) < TOL
    sd = sub_sd(fx, "dec.sd.")
    up = torch.nn.functional.conv_transpose3d(tt(fx["dec.x"]), sd["upsampling.upsample.conv_transposed.weight"], None, stride=2, padding=1)
    up = O.nearest_resize(up, (8, 8, 8))
    assert rel_err(up, tt(fx["dec.up"])) < TOL
    assert rel_err(O.decoder(tt(fx["dec.ef"]), tt(fx["dec.x"]), sd, ""), tt(fx["dec.out"])) < TOL
    for n in (3, 4, 12, 24, 47):   # nearest 2n-1 -> 2n duplicates the first plane: [0,0,1,2,...]
        src = torch.arange(2 * n - 1, dtype=torch.float32).view(1, 1, -1, 1, 1)
        got = O.nearest_resize(src, (2 * n, 1, 1)).flatten().long().tolist()
        assert got == fx[f"nearest_idx_{n}"].tolist() == [max(i - 1, 0) for i in range(2 * n)]
    assert torch.equal(torch.nn.functional.max_pool3d(tt(fx["mp.x"]), 2), tt(fx["mp.out"]))


def test_vit():
    fx = golden("t0_vit.npz")
    sd = sub_sd(fx, "sd.")
    out = O.vit_mid(tt(fx["x"]), sd, "", patch=8, heads=2, depth=2)
    assert rel_err(out, tt(fx["out"])) < TOL


def test_vit3d():
    for tag, pf, p in (("a", 8, 8), ("b", 8, 4)):
        fx = golden(f"t0_vit3d_{tag}.npz")
        sd = sub_sd(fx, "sd.")
        out, tok = O.vit3d(tt(fx["x"]), sd, "", frame_patch=pf, patch=p, heads=2, depth=2)
        assert rel_err(tok, tt(fx["tokens"])) < TOL and rel_err(out, tt(fx["out"])) < TOL
        out_mean, _ = O.vit3d(tt(fx["x"]), sd, "", frame_patch=pf, patch=p, heads=2, depth=2, pool="mean")
        assert rel_err(out_mean, tt(fx["out_mean"])) < TOL


def test_oracle_vit3d_backward_vs_reference_autograd():
    """Fixture t10 = the REFERENCE's own autograd through vit_3d.ViT (cross-entropy of the logits): autograd through the oracle's vit3d
    restatement gives the same loss, input gradient and parameter gradients -- the backward the flash-attention backward is compared with
    (tests/test_unet_gpu.py)."""
    import torch.nn.functional as F
    sl = lambda t, n: t.detach().reshape(-1)[::max(1, t.numel() // n)][:n].double()
    for tag, pf, p in (("b", 8, 4), ("c", 8, 4)):
        fx = golden(f"t10_vit3d_grads_{tag}.npz")
        tr = {k: v.clone().requires_grad_(True) for k, v in sub_sd(fx, "sd.").items()}
        x = tt(fx["x"]).requires_grad_()
        out, _ = O.vit3d(x, tr, "", frame_patch=pf, patch=p, heads=2, depth=2)
        loss = F.cross_entropy(out, torch.from_numpy(fx["labels"]))
        loss.backward()
        assert rel_err(out, tt(fx["out"])) < TOL and abs(loss.item() - float(fx["loss"])) < 1e-5
        assert rel_err(sl(x.grad, 256), tt(fx["dx_slice"])) < 1e-3
        for k in [k[len("gnorm."):] for k in fx if k.startswith("gnorm.")]:
            g = tr[k].grad
            e_n = abs(g.double().norm().item() - float(fx["gnorm." + k])) / max(float(fx["gnorm." + k]), 1e-12)
            assert e_n < 1e-4 and rel_err(sl(g, 128), tt(fx["gslice." + k])) < 1e-3, (tag, k, e_n)
        for k in [k[len("gfull."):] for k in fx if k.startswith("gfull.")]:
            assert rel_err(tr[k].grad, tt(fx["gfull." + k])) < 1e-4, (tag, k)


def test_index_maps_bit_exact():
    fx = golden("t0_index_maps.npz")
    for name, shp in (("native", (40, 40, 24)), ("g96", (24, 24, 24)), ("g128", (32, 32, 32)), ("g32", (8, 8, 8))):
        src = torch.arange(int(np.prod(shp)), dtype=torch.int64).view(1, 1, *shp)
        f = O.fold_mid(src)
        assert torch.equal(f[0, 0], tt(fx[f"fold_{name}"]))
        assert torch.equal(O.unfold_mid(f, shp[2]), src)
    img = torch.arange(2 * 16 * 8, dtype=torch.int64).view(1, 2, 16, 8)
    p = O.patchify(img, 4)
    assert torch.equal(p[0], tt(fx["patchify_c2_16x8_p4"]))
    assert torch.equal(O.unpatchify(p, 4, 4, 2), img)
    vol = torch.arange(2 * 4 * 6 * 8, dtype=torch.int64).view(2, 1, 4, 6, 8)
    assert torch.equal(O.build_condition([vol]), tt(fx["condition_2x1x4x6x8"]))


def test_c_scan_oracle():
    """oracle/scan_ref.c against the reference fixture (both the plain and the fused-gate contract)."""
    from oracle import c_oracle
    fx = golden("t0_selective_scan.npz")
    y = c_oracle.selective_scan(fx["ss_x"], fx["ss_delta"], fx["ss_A"], fx["ss_B"], fx["ss_C"], fx["ss_D"])
    assert rel_err(tt(y), tt(fx["ss_y_seq"])) < TOL
    y = c_oracle.selective_scan(fx["ss_x"], fx["ss_draw"], fx["ss_A"], fx["ss_B"], fx["ss_C"], fx["ss_D"], z=fx["ss_z"],
                                bias=fx["ss_dbias"], softplus=True)
    assert rel_err(tt(y), tt(fx["ss_yfn"])) < TOL


ABL_CASES = dict(full=dict(feat=True, cond=True, no_table=False), table_only=dict(feat=False, cond=True, no_table=False),
                 no_table=dict(feat=True, cond=True, no_table=True), no_cross=dict(feat=True, cond=False, no_table=False))


def ablation_setup(fx, device="cpu"):
    """Inputs + deterministic weights of tests/golden/t3_ablation.npz (tools/make_golden.py t3)."""
    import gfe_hip.det_init as det
    from cross_atten.mamba_transformer import Cross_mamba_ablation
    m = [int(v) for v in fx["meta"]]
    cards, (n_cont, dim, depth, heads), vol, Bn = tuple(m[:3]), m[3:7], tuple(m[7:10]), m[10]
    ft = Cross_mamba_ablation(categories=cards, num_continuous=n_cont, dim=dim, depth=depth, heads=heads, dim_head=dim // heads,
                              d_cross=vol[0] * vol[1])
    ft.load_state_dict(det.det_state_dict(ft.state_dict(), seed=31, prefix="abl."))
    x, x_cat, x_num, y = [t.to(device) for t in det.det_inputs(Bn, vol, cards, n_cont, seed=31)]
    return ft, depth, heads, x, x_cat, x_num, y, tt(fx["pet"], device=device), tt(fx["feat"], device=device)


def test_cross_mamba_ablation_variants():
    fx = golden("t3_ablation.npz")
    ft, depth, heads, x, x_cat, x_num, y, pet, feat = ablation_setup(fx)
    sd = {k: v.detach() for k, v in ft.state_dict().items()}
    for name, c in ABL_CASES.items():
        pred = O.cross_mamba_ablation(x_cat, x_num, feat if c["feat"] else None, [x, pet] if c["cond"] else None, sd, depth, heads,
                                      no_table=c["no_table"])
        assert rel_err(pred, tt(fx[name + ".pred"])) < TOL, name
    # the variants really differ (a fixture that ignored the switches would pass the loop above with one set of numbers)
    preds = [fx[n + ".pred"] for n in ABL_CASES]
    assert all(np.abs(preds[i] - preds[j]).max() > 1e-4 for i in range(4) for j in range(i))


AN_CASES = ("normal", "ties", "positive", "constant", "nonfinite", "single", "tiny_mixed")


def test_adaptive_normal_bit_exact():
    """oracle/ref_ops.adaptive_normal against the reference's own output (tests/golden/t4_adaptive_normal.npz): same f32 operations
    in the same order -> identical bits, NaNs in the same places."""
    fx = golden("t4_adaptive_normal.npz")
    for name in AN_CASES:
        got = O.adaptive_normal(torch.from_numpy(fx[name + ".x"].copy())).numpy()
        assert np.array_equal(got, fx[name + ".y"], equal_nan=True), name


def test_activation_pattern_replay_reproduces_the_plain_generator():
    """oracle.ref_ops.activation_pattern (used by the f-1 gradient parity tests): replaying the generator's OWN ReLU masks and max-pool
    selections must give the same output and the same parameter gradients as the plain evaluation."""
    import torch.nn.functional as F
    torch.manual_seed(0)
    f = (8, 16, 32)
    sd = {}
    def conv(pre, cin, cout):
        sd[pre + "groupnorm.weight"] = 1 + 0.1 * torch.randn(cin); sd[pre + "groupnorm.bias"] = 0.1 * torch.randn(cin)
        sd[pre + "conv.weight"] = torch.randn(cout, cin, 3, 3, 3) / (27 * cin) ** 0.5
    cin = 1
    for i, c in enumerate(f):
        pre = f"encoders.{i}.basic_module."
        sd[pre + "conv1.weight"] = torch.randn(c, cin, 1, 1, 1) / cin ** 0.5; sd[pre + "conv1.bias"] = 0.1 * torch.randn(c)
        conv(pre + "conv2.", c, c); conv(pre + "conv3.", c, c)
        cin = c
    x = torch.randn(1, 1, 8, 8, 8)
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items()}

    def encoders(p, record=None):
        t = x
        for i in range(3):
            if i > 0:
                if record is not None:
                    y, idx = F.max_pool3d(t, 2, return_indices=True)
                    m = torch.zeros_like(t).flatten(2).scatter_(2, idx.flatten(2), 1.0).view_as(t)
                    record.append(m)
                t = O._max_pool2(t)
            r = F.conv3d(t, p[f"encoders.{i}.basic_module.conv1.weight"], p[f"encoders.{i}.basic_module.conv1.bias"])
            if record is not None:
                o2 = O.single_conv(r, p, f"encoders.{i}.basic_module.conv2.", relu=False)
                record.append((o2 > 0).float())
                o3 = O.single_conv(F.relu(o2), p, f"encoders.{i}.basic_module.conv3.", relu=False)
                record.append(((o3 + r) > 0).float())
            t = O.resnet_block(t, p, f"encoders.{i}.basic_module.")
        return t

    rec = []
    with torch.no_grad():
        encoders(sd, rec)
    y0 = encoders(params)
    y0.square().sum().backward()
    g0 = {k: v.grad.clone() for k, v in params.items()}
    for v in params.values():
        v.grad = None
    with O.activation_pattern(rec):
        y1 = encoders(params)
    y1.square().sum().backward()
    assert torch.allclose(y0, y1, atol=1e-6)
    for k, v in params.items():
        assert torch.allclose(v.grad, g0[k], atol=1e-5, rtol=1e-5), k
    assert O._PATTERN is None


def test_oracle_generator_backward_vs_reference_autograd():
    """Fixture t9 = the REFERENCE's own autograd through Residual_mid_UNet3D_vit with an L1 loss (main_gan_vit.py:68-82 minus the third-party
    losses; reduced width, 32^3, eval mode): autograd through the oracle's generator restatement must give the same loss and the same
    parameter gradients -- what pins the backward that tests/test_gen_train_gpu.py compares the HIP training path with (row f-1)."""
    import importlib.util
    import os
    import torch.nn.functional as F
    from conftest import ROOT
    spec = importlib.util.spec_from_file_location("det_init", os.path.join(ROOT, "gfe-mamba_amd", "gfe_hip", "det_init.py"))
    det = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(det)
    fx = golden("t9_generator_grads.npz")
    keys = [k[len("gnorm."):] for k in fx if k.startswith("gnorm.")]
    nograd = [k[len("nograd."):] for k in fx if k.startswith("nograd.")]
    vol = (32, 32, 32)
    # the parameter shapes follow from the fixture's key list + the reduced geometry; regenerate the deterministic weights
    from pytorch3dunet.unet3d.model import Residual_mid_UNet3D_vit
    with torch.device("meta"):
        tmpl = Residual_mid_UNet3D_vit(1, 1, is_segmentation=False, f_maps=(8, 16, 32), vol_size=vol,
                                       vit_kwargs=dict(dim=64, depth=2, heads=2, dim_head=16, mlp_dim=128))
    shapes = {k: tuple(v.shape) for k, v in tmpl.state_dict().items()}
    sd = det.det_state_dict(shapes, seed=51, prefix="gtrain.")
    tr = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    x = det.det_inputs(2, vol, seed=51)[0]
    target = torch.tanh(torch.randn(2, 1, *vol, generator=torch.Generator().manual_seed(52)))
    _, _, pet = O.generator(x, tr, vit_heads=2, vit_depth=2)
    loss = F.l1_loss(pet, target)
    loss.backward()
    assert abs(loss.item() - float(fx["loss"])) < 1e-6 * max(1.0, float(fx["loss"]))
    sl = lambda t, n: t.detach().reshape(-1)[::max(1, t.numel() // n)][:n].double()
    assert rel_err(sl(pet, 256), tt(fx["pet_slice"])) < 2e-5
    assert set(keys) | set(nograd) == {k for k, v in sd.items() if v.dtype.is_floating_point}
    worst = 0.0
    for k in keys:
        g = tr[k].grad
        assert g is not None, k
        e_n = abs(g.double().norm().item() - float(fx["gnorm." + k])) / max(float(fx["gnorm." + k]), 1e-12)
        e_s = rel_err(sl(g, 64), tt(fx["gslice." + k]))
        worst = max(worst, e_n, e_s)
        assert e_n < 1e-4 and e_s < 1e-3, (k, e_n, e_s)
    for k in nograd:                                       # the reference's dead parameter (model.py:119 mid_linear): no gradient on either side
        assert tr[k].grad is None, k
    print("oracle generator backward vs the reference's autograd: worst relative error %.2e over %d tensors" % (worst, len(keys)))
