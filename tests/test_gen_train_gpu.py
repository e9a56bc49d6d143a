"""conv3d backward and the generator training step (main_gan_vit.py:68-82).  The differentiable generator forward
(gfe_hip/gen_train.py: autograd Functions over the forward's own HIP kernels + the gen_train.hip passes) against torch fp32 autograd of
the same layers, and end to end against autograd through the ORACLE's generator (pinned to the reference by tests/test_oracle_golden.py).
bf16 activations -> 1e-2-class tolerances per layer; the measured errors are printed."""
import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err
from oracle import ref_ops as O

pytestmark = pytest.mark.gpu
DEV = "cuda"
BF = torch.bfloat16


def _cl(t):
    return t.permute(0, 2, 3, 4, 1).contiguous()


def _nc(t):
    return t.permute(0, 4, 1, 2, 3)


def test_maxpool_backward_routes_to_the_first_maximum():
    from gfe_hip.gen_train import _MaxPoolFn
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 16, 6, 8, 10, generator=g).to(BF).float()
    x[0, :, 0:2, 0:2, 0:2] = 1.0                                   # a window of ties: the first voxel wins (ATen)
    xr = x.clone().requires_grad_(True)
    w = torch.randn(2, 16, 3, 4, 5, generator=g).to(BF).float()
    (F.max_pool3d(xr, 2) * w).sum().backward()
    xg = _cl(x).to(BF).to(DEV).requires_grad_(True)
    y = _MaxPoolFn.apply(xg)
    assert torch.equal(_nc(y).float().cpu(), F.max_pool3d(x, 2))
    y.backward(_cl(w).to(BF).to(DEV))
    assert torch.equal(_nc(xg.grad).float().cpu(), xr.grad)


@pytest.mark.parametrize("ci,co,shape,ntaps", [(64, 64, (2, 5, 9, 11), 27), (32, 128, (1, 4, 8, 8), 27), (96, 64, (1, 9, 3, 17), 27), (64, 128, (1, 6, 7, 5), 8)])
def test_fused_weight_gradient_kernel_vs_torch(ci, co, shape, ntaps):
    """gfe_conv3d_wgrad (all taps in one launch, transposed LDS reads) on ragged volumes: against torch's conv3d weight gradient in fp32
    on the same bf16-rounded tensors (27 taps), against the per-tap GEMM path (the 8-tap list of the transposed conv), and bit-for-bit
    run-to-run (fixed split order, no atomics)."""
    import gfe_hip.gen_train as GT
    B, D, H, W = shape
    g = torch.Generator().manual_seed(ci * co + ntaps)
    x = torch.randn(B, ci, D, H, W, generator=g).to(BF).float()
    d = torch.randn(B, co, D, H, W, generator=g).to(BF).float()
    xg, dg = _cl(x).to(BF).to(DEV), _cl(d).to(BF).to(DEV)
    taps = GT.K.CONV3_TAPS if ntaps == 27 else GT._CT_TAPS8
    out = GT.conv_wgrad(xg, dg, taps)
    assert out.shape == (ntaps, co, ci)
    if ntaps == 27:
        w = torch.zeros(co, ci, 3, 3, 3, requires_grad=True)
        (F.conv3d(x, w, padding=1) * d).sum().backward()
        ref = w.grad.permute(2, 3, 4, 0, 1).reshape(27, co, ci)
    else:
        import os
        os.environ["GFE_WGRAD_GEMM"] = "1"
        try:
            ref = GT.conv_wgrad(xg, dg, taps).cpu()
        finally:
            del os.environ["GFE_WGRAD_GEMM"]
    e = rel_err(out, ref)
    print("fused wgrad %d->%d %s taps=%d: %.2e" % (ci, co, shape, ntaps, e))
    assert e < 2e-5
    assert torch.equal(out, GT.conv_wgrad(xg, dg, taps))


def _l2(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


@pytest.mark.parametrize("cin,c,shape", [(1, 16, (2, 8, 8, 8)), (16, 32, (1, 6, 8, 10)), (64, 64, (1, 8, 8, 8))])
def test_resnet_block_forward_backward_vs_torch(cin, c, shape):
    """r = conv1(x); relu(conv3(GN(relu(conv2(GN(r))))) + r): output, input gradient and every parameter gradient against fp32 autograd
    of the oracle's block.  Gradients are compared at the SAME ReLU pattern (oracle.activation_pattern): the bf16 forward flips ~0.1 % of
    the near-zero ReLUs, and with the random cotangent used here each flip is an O(1) change of a few gradient terms (5-15 % of the
    gradient norm, printed); the flipped fraction itself is bounded separately."""
    import gfe_hip.gen_train as GT
    from pytorch3dunet.unet3d.buildingblocks import ResNetBlock
    B, D, H, W = shape
    g = torch.Generator().manual_seed(cin + c)
    blk = ResNetBlock(cin, c)
    with torch.no_grad():
        for k, p in blk.named_parameters():
            if "groupnorm.weight" in k:
                p.copy_(1 + 0.2 * torch.randn(p.shape, generator=g))
            elif p.dim() == 1:
                p.copy_(0.1 * torch.randn(p.shape, generator=g))
            else:
                p.copy_(0.2 * torch.randn(p.shape, generator=g))
    x = torch.randn(B, cin, D, H, W, generator=g)
    if cin > 1:
        x = x.to(BF).float()
    w = torch.randn(B, c, D, H, W, generator=g)
    blk = blk.to(DEV)
    xg = (x.to(DEV) if cin == 1 else _cl(x).to(BF).to(DEV)).requires_grad_(cin > 1)
    GT.PATTERN_LOG = []
    try:
        y = GT.resnet_block(blk, xg)
    finally:
        pattern, GT.PATTERN_LOG = [_nc(m).cpu() for m in GT.PATTERN_LOG], None
    y.backward(_cl(w).to(BF).to(DEV))

    def reference(masks):
        sd = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in blk.named_parameters()}
        xr = x.clone().requires_grad_(True)
        if masks is None:
            o = O.resnet_block(xr, sd, "")
        else:
            with O.activation_pattern(masks):
                o = O.resnet_block(xr, sd, "")
        (o * w).sum().backward()
        return o.detach(), sd, xr

    o_plain, sd_plain, _ = reference(None)
    o, sd, xr = reference(pattern)
    e_fwd = rel_err(_nc(y), o_plain)
    flipped = ((o_plain > 0) != pattern[-1]).float().mean().item()
    errs = {k: rel_err(p.grad, sd[k].grad) for k, p in blk.named_parameters()}
    if cin > 1:
        errs["dx"] = rel_err(_nc(xg.grad), xr.grad)
    plain = max(_l2(p.grad, sd_plain[k].grad) for k, p in blk.named_parameters())
    print("ResNetBlock %d->%d %s: forward %.2e, output ReLUs flipped %.4f, gradients at the same pattern %s; vs the fp32 pattern worst L2 %.2e"
          % (cin, c, shape, e_fwd, flipped, {k: "%.1e" % v for k, v in errs.items()}, plain))
    assert e_fwd < 1e-2 and flipped < 5e-3
    # measured 3e-3..6e-3; conv1.bias (a heavily cancelling sum of bf16-rounded dr over all voxels) 2.8e-2 on the 1->16 case
    assert max(v for k, v in errs.items() if k != "conv1.bias") < 1.5e-2 and errs.get("conv1.bias", 0.0) < 5e-2, errs
    assert plain < 0.3


@pytest.mark.parametrize("cin,cout,shape,full", [(32, 16, (1, 4, 6, 5), True), (128, 64, (1, 3, 4, 4), True), (16, 8, (2, 4, 4, 4), False)])
def test_transposed_conv_join_backward_vs_torch(cin, cout, shape, full):
    from gfe_hip.gen_train import _UpJoinFn
    from pytorch3dunet.unet3d.buildingblocks import TransposeConvUpsampling
    B, D, H, W = shape
    g = torch.Generator().manual_seed(cin)
    up = TransposeConvUpsampling(cin, cout)
    with torch.no_grad():
        up.upsample.conv_transposed.weight.copy_(torch.randn(cin, cout, 3, 3, 3, generator=g) / (27 * cin / 8) ** 0.5)
    x = torch.randn(B, cin, D, H, W, generator=g).to(BF).float()
    osz = [2 * n if full else 2 * n - 1 for n in (D, H, W)]
    enc = torch.randn(B, cout, *osz, generator=g).to(BF).float()
    w = torch.randn(B, cout, *osz, generator=g).to(BF).float()
    xr, er = x.clone().requires_grad_(True), enc.clone().requires_grad_(True)
    wr = up.upsample.conv_transposed.weight.detach().clone().requires_grad_(True)
    u = F.conv_transpose3d(xr, wr, stride=2, padding=1)
    if list(u.shape[2:]) != osz:
        u = F.interpolate(u, size=osz)                                         # nearest (buildingblocks.py:533)
    ((er + u) * w).sum().backward()
    up = up.to(DEV)
    xg, eg = _cl(x).to(BF).to(DEV).requires_grad_(True), _cl(enc).to(BF).to(DEV).requires_grad_(True)
    y = _UpJoinFn.apply(eg, xg, up.upsample.conv_transposed.weight, up)
    y.backward(_cl(w).to(BF).to(DEV))
    e = dict(dx=rel_err(_nc(xg.grad), xr.grad), denc=rel_err(_nc(eg.grad), er.grad), dw=rel_err(up.upsample.conv_transposed.weight.grad, wr.grad))
    print("ConvTranspose join %d->%d %s full=%s: %s" % (cin, cout, shape, full, {k: "%.1e" % v for k, v in e.items()}))
    assert e["denc"] == 0.0 and e["dx"] < 1e-2 and e["dw"] < 1e-2, e


@pytest.mark.parametrize("tag,vol,f_maps,vit,batch", [
    ("reduced width 32^3", (32, 32, 32), (8, 16, 32), dict(dim=64, depth=2, heads=2, dim_head=16, mlp_dim=128), 2),
    ("FULL width 64^3", (64, 64, 64), (64, 128, 256), None, 1),
])
def test_generator_training_gradients_vs_oracle_autograd(tag, vol, f_maps, vit, batch):
    """L1 loss and every parameter gradient of the generator training path against torch autograd through the oracle's generator (CPU fp32;
    its backward is pinned to the reference's own autograd by fixture t9, tests/test_oracle_golden.py) on the same deterministic weights,
    at the same activation pattern: the reduced-width model on 32^3 (the shape of t9) and the FULL-width model (64 / 128 / 256 channels,
    ViT 512 x 4 x 6: every channel count of the benchmarked training step) on 64^3."""
    import gfe_hip.det_init as det
    import gfe_hip.gen_train as GT
    from gfe_hip.gen_train import generator_forward_train
    from pytorch3dunet.unet3d.model import Residual_mid_UNet3D_vit
    gen = Residual_mid_UNet3D_vit(1, 1, is_segmentation=False, f_maps=f_maps, vol_size=vol, vit_kwargs=vit)
    vh, vd = (vit or {}).get("heads", 6), (vit or {}).get("depth", 4)
    sd = det.det_state_dict(gen.state_dict(), seed=51, prefix="gtrain.")
    gen.load_state_dict(sd)
    gen = gen.to(DEV).eval()                                      # eval: dropout off on both sides; gradients still flow
    x = det.det_inputs(batch, vol, seed=51)[0]
    target = torch.tanh(torch.randn(batch, 1, *vol, generator=torch.Generator().manual_seed(52)))
    tr = {k: v.clone().float().requires_grad_(True) for k, v in sd.items() if v.dtype.is_floating_point}
    GT.PATTERN_LOG = []
    try:
        pet = generator_forward_train(gen, x.to(DEV))
    finally:
        pattern, GT.PATTERN_LOG = [_nc(m).cpu() for m in GT.PATTERN_LOG], None
    loss = F.l1_loss(pet, target.to(DEV))
    loss.backward()
    pet_plain = O.generator(x, {k: v.float() for k, v in sd.items()}, vit_heads=vh, vit_depth=vd)[2]
    with O.activation_pattern(pattern):                            # same ReLU / max-pool pattern (see the block test)
        _, _, pet_ref = O.generator(x, tr, vit_heads=vh, vit_depth=vd)
    loss_ref = F.l1_loss(pet_ref, target)
    # L1's gradient sign(pet - target) / N is one more piecewise-constant factor: the cotangent is taken at our output's signs
    cot = torch.sign(pet.detach().float().cpu() - target) / target.numel()
    sign_flips = (torch.sign(pet_ref.detach() - target) != torch.sign(pet.detach().float().cpu() - target)).float().mean().item()
    (pet_ref * cot).sum().backward()
    assert sign_flips < 1e-2
    assert rel_err(pet, pet_plain) < 3e-2
    e_pet, e_loss = rel_err(pet, pet_ref), abs(loss.item() - loss_ref.item()) / loss_ref.item()
    errs = {}
    for k, p in gen.named_parameters():
        if k.startswith("mid_linear"):                            # dead parameter of the reference (model.py:119), no gradient on either side
            assert p.grad is None and tr[k].grad is None
            continue
        ref = tr[k].grad
        errs[k] = (rel_err(p.grad, ref), abs(p.grad.double().norm().item() - ref.double().norm().item()) / max(ref.double().norm().item(), 1e-12))
    worst = sorted(((v[0], k) for k, v in errs.items()), reverse=True)[:5]
    med = sorted(v[0] for v in errs.values())[len(errs) // 2]
    wn = max(v[1] for v in errs.values())
    print("generator training step (%s) vs oracle autograd: pet %.2e, L1 loss %.2e; gradient elements: median %.2e, worst %s; worst norm error %.2e"
          % (tag, e_pet, e_loss, med, [("%.1e" % e, k) for e, k in worst], wn))
    assert e_pet < 3e-2 and e_loss < 1e-2
    assert med < 2e-2 and worst[0][0] < 6e-2 and wn < 3e-2              # measured 9.5e-3 / 3.2e-2 / 8.9e-3


@pytest.mark.parametrize("vol,f_maps,vit", [((32, 32, 32), (8, 16, 32), dict(dim=64, depth=2, heads=2, dim_head=16, mlp_dim=128)),
                                            ((64, 64, 64), (64, 128, 256), None)])
def test_generator_training_backward_is_bit_reproducible(vol, f_maps, vit):
    """the generator-training path summed its GroupNorm-backward moments, the final conv's and the first lift's dw / db and
    the tall bias gradients with f32 atomics, so row f-1 was not run-to-run reproducible.  Round 5: per-lane LDS slots + per-block partial rows
    folded in order (csrc/gen_train.hip, gfe_colsum_f32_ws).  Two forward + backward passes from the same state: the loss and EVERY
    parameter gradient must be bit-identical."""
    import gfe_hip.det_init as det
    from gfe_hip.gen_train import generator_forward_train
    from pytorch3dunet.unet3d.model import Residual_mid_UNet3D_vit
    gen = Residual_mid_UNet3D_vit(1, 1, is_segmentation=False, f_maps=f_maps, vol_size=vol, vit_kwargs=vit)
    gen.load_state_dict(det.det_state_dict(gen.state_dict(), seed=57, prefix="gtrain3."))
    gen = gen.to(DEV).eval()
    x = det.det_inputs(2, vol, seed=57)[0].to(DEV)
    target = torch.tanh(torch.randn(2, 1, *vol, generator=torch.Generator().manual_seed(58))).to(DEV)
    runs = []
    for _ in range(2):
        for p in gen.parameters():
            p.grad = None
        loss = F.l1_loss(generator_forward_train(gen, x), target)
        loss.backward()
        runs.append((loss.detach().clone(), {k: p.grad.clone() for k, p in gen.named_parameters() if p.grad is not None}))
    assert torch.equal(runs[0][0], runs[1][0])
    diff = [k for k in runs[0][1] if not torch.equal(runs[0][1][k], runs[1][1][k])]
    assert not diff, "gradients differ between two identical passes: %s" % diff[:8]


@pytest.mark.parametrize("shape", [(2, 1, 24, 16, 16), (1, 1, 5, 7, 3), (3, 4097)])
def test_l1_loss_value_and_gradient_vs_torch(shape):
    """gfe_l1_loss (nn.L1Loss of main_gan_vit.py:72): the mean against f64, the gradient sign(pred - target) / n exactly (ties give 0), both
    run-to-run bit-identical (per-block partial sums added in order)."""
    from gfe_hip.gen_train import l1_loss
    g = torch.Generator().manual_seed(sum(shape))
    pred = torch.randn(shape, generator=g).to(DEV).requires_grad_(True)
    target = torch.randn(shape, generator=g).to(DEV)
    with torch.no_grad():
        target.view(-1)[::7] = pred.view(-1)[::7]                       # exact ties
    loss = l1_loss(pred, target)
    (loss * 3.0).backward()
    ref = (pred.detach().double() - target.double()).abs().mean()
    assert abs(loss.item() - ref.item()) < 2e-6 * max(1.0, ref.item())
    sgn = torch.sign(pred.detach() - target)
    assert torch.equal(torch.sign(pred.grad), sgn) and (sgn == 0).sum().item() >= pred.numel() // 7        # the sign pattern exactly, ties give 0
    assert torch.allclose(pred.grad, sgn * (3.0 / pred.numel()), rtol=1e-6, atol=0.0)                     # (1 / n rounded once, then x 3)
    again = l1_loss(pred.detach(), target)
    assert torch.equal(again, loss.detach())


def test_generator_train_steps_reduce_the_l1_loss_and_refresh_the_frozen_packs():
    """train_step (main_gan_vit.py:68-82 minus the third-party losses) with FlatAdam: the loss falls over a few steps, and the eval-mode
    forward (which caches packed / GroupNorm-folded weights) sees the updated parameters although the update kernel rewrites them
    without a tensor-version bump."""
    import gfe_hip.det_init as det
    from gfe_hip.gen_train import generator_forward_train, train_step
    from gfe_hip.train_ops import FlatAdam
    from pytorch3dunet.unet3d.model import Residual_mid_UNet3D_vit
    vol = (32, 32, 32)
    gen = Residual_mid_UNet3D_vit(1, 1, is_segmentation=False, f_maps=(8, 16, 32), vol_size=vol,
                                  vit_kwargs=dict(dim=64, depth=2, heads=2, dim_head=16, mlp_dim=128))
    gen.load_state_dict(det.det_state_dict(gen.state_dict(), seed=53, prefix="gtrain2."))
    gen = gen.to(DEV).eval()
    x = det.det_inputs(2, vol, seed=53)[0].to(DEV)
    target = torch.tanh(torch.randn(2, 1, *vol, generator=torch.Generator().manual_seed(54))).to(DEV)
    with torch.no_grad():
        before = gen(x).float().clone()
    opt = FlatAdam([p for p in gen.parameters()], lr=2e-3, max_norm=float("inf"))
    losses = [train_step(gen, opt, x, target).item() for _ in range(12)]
    print("generator L1 over 12 FlatAdam steps:", ["%.4f" % v for v in losses])
    assert losses[-1] < 0.9 * losses[0]
    with torch.no_grad():
        after = gen(x).float()
        again = generator_forward_train(gen, x).float()
    e_same, e_moved = rel_err(after, again), rel_err(after, before)
    print("frozen forward vs training forward on the updated weights: %.2e; vs the frozen forward before training: %.2e" % (e_same, e_moved))
    assert e_same < 4e-2                                            # frozen-forward path == training forward on the UPDATED weights (measured 1.6-1.8e-2:
    assert e_moved > 5e-2                                           # two bf16 evaluation orders of the same network)


@pytest.mark.parametrize("rows,dim", [(6, 16384), (3, 50000), (64, 262144)])
def test_long_row_layernorm_vs_torch(rows, dim):
    """gfe_layernorm_rows on rows of >= 16384 elements (the generator ViT's LayerNorm(patch_dim), vit.py:101-105): segments of a row run
    on different CUs (Chan-combined moments); forward, dx, dgamma, dbeta against torch in fp64, with a large common offset in the rows."""
    from gfe_hip.head_ops import layernorm_rows
    g = torch.Generator().manual_seed(rows)
    x = (torch.randn(rows, dim, generator=g) + 3.0 * torch.randn(rows, 1, generator=g)).to(DEV).requires_grad_(True)
    gamma = (1 + 0.1 * torch.randn(dim, generator=g)).to(DEV).requires_grad_(True)
    beta = (0.1 * torch.randn(dim, generator=g)).to(DEV).requires_grad_(True)
    w = torch.randn(rows, dim, generator=g).to(DEV)
    y = layernorm_rows(x, gamma, beta)
    (y * w).sum().backward()
    xr, gr, br = [t.detach().double().cpu().requires_grad_(True) for t in (x, gamma, beta)]
    yr = F.layer_norm(xr, (dim,), gr, br)
    (yr * w.double().cpu()).sum().backward()
    e = dict(y=rel_err(y, yr), dx=rel_err(x.grad, xr.grad), dgamma=rel_err(gamma.grad, gr.grad), dbeta=rel_err(beta.grad, br.grad))
    print("long-row LayerNorm (%d, %d): %s" % (rows, dim, {k: "%.1e" % v for k, v in e.items()}))
    assert max(e.values()) < 2e-5, e
