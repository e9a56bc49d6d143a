"""Cross_jamba_both / Jamba (cross_atten/mamba_transformer.py:135-251, cross_atten/jamba.py:258-535) -- the oracle's
restatement pinned to the reference's outputs (tests/golden/t6_jamba.npz, tools/make_golden.py t6) on CPU, and the HIP path against
the same fixture on the GPU: logits, loss, every parameter gradient (norm and 64-element slice)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import golden, rel_err, tt
from oracle import ref_ops as O

CARDS, N_CONT, DIM, DEPTH, HEADS, VOL, BN = (5, 3, 2), 6, 64, 3, 8, (8, 12, 6), 3


def _model_and_inputs():
    import gfe_hip.det_init as det
    from cross_atten.mamba_transformer import Cross_jamba_both
    ft = Cross_jamba_both(categories=CARDS, num_continuous=N_CONT, dim=DIM, depth=DEPTH, heads=HEADS, dim_head=DIM // HEADS, d_cross=VOL[0] * VOL[1])
    ft.load_state_dict(det.det_state_dict(ft.state_dict(), seed=41, prefix="jam."))
    x, x_cat, x_num, y = det.det_inputs(BN, VOL, CARDS, N_CONT, seed=41)
    return ft, x, x_cat, x_num, y


def _slices(t, n=64):
    f = t.detach().reshape(-1)
    return f[::max(1, f.numel() // n)][:n]


def test_structure_and_state_dict_keys():
    ft, *_ = _model_and_inputs()
    fx = golden("t6_jamba.npz")
    keys = {k[len("gnorm."):] for k in fx if k.startswith("gnorm.")}
    assert keys == {k for k, _ in ft.named_parameters()}                          # the reference's parameter names, all of them
    kinds = [type(l).__name__ for l in ft.transformer.layers]
    assert kinds == ["MambaLayer"] * 4 + ["AttentionLayer", "MambaLayer"]         # attention at (i - 4) % 8 == 0 (jamba.py:267)
    assert [l.moe.num_experts for l in ft.transformer.layers] == [1, 16, 1, 16, 1, 16]     # experts at odd layers (:268)
    assert ft.transformer.layers[0].mamba.dt_layernorm is not None                # inner_layernorms=True (:61)


def test_oracle_cross_jamba_both_vs_reference_fixture():
    fx = golden("t6_jamba.npz")
    ft, x, x_cat, x_num, y = _model_and_inputs()
    sd = {k: (v.double() if v.dtype.is_floating_point else v) for k, v in ft.state_dict().items()}
    tr = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.dtype.is_floating_point}
    sd2 = {k: tr.get(k, v) for k, v in sd.items()}
    pred = O.cross_jamba_both(x_cat, x_num.double(), tt(fx["feat"]).double(), [x.double(), tt(fx["pet"]).double()], sd2, depth=DEPTH, heads=HEADS)
    assert rel_err(pred, tt(fx["pred"])) < 2e-5
    loss = F.binary_cross_entropy(torch.sigmoid(pred.squeeze(1)), y.double())
    assert abs(loss.item() - float(fx["loss"])) < 2e-5
    loss.backward()
    for k, p in ft.named_parameters():
        ref = float(fx["gnorm." + k])
        if ref < 0:
            assert tr[k].grad is None or float(tr[k].grad.norm()) == 0.0          # experts no token was routed to
            continue
        got = float(tr[k].grad.norm()) if tr[k].grad is not None else 0.0
        assert abs(got - ref) <= 2e-4 * max(ref, 1e-6) + 1e-9, (k, got, ref)


@pytest.mark.gpu
def test_cross_jamba_both_hip_vs_reference_fixture():
    fx = golden("t6_jamba.npz")
    ft, x, x_cat, x_num, y = _model_and_inputs()
    ft = ft.cuda().eval()
    pred = ft(x_cat.cuda(), x_num.cuda(), tt(fx["feat"], device="cuda"), [x.cuda(), tt(fx["pet"], device="cuda")])
    e_pred = rel_err(pred, tt(fx["pred"]))
    loss = F.binary_cross_entropy(torch.sigmoid(pred.squeeze(1)), y.cuda().float())
    e_loss = abs(loss.item() - float(fx["loss"]))
    loss.backward()
    worst = (0.0, "")
    for k, p in ft.named_parameters():
        ref = float(fx["gnorm." + k])
        if ref < 1e-9:
            assert p.grad is None or float(p.grad.norm()) < 1e-6, k
            continue
        e = rel_err(_slices(p.grad), tt(fx["gslice." + k]))
        worst = max(worst, (e, k))
        assert abs(float(p.grad.double().norm()) - ref) / ref < 1e-2, k
    print("Cross_jamba_both vs reference: logits %.2e, loss %.2e, worst gradient element %.2e (%s)" % (e_pred, e_loss, *worst))
    # f32 everywhere except the K / V projections over the (bf16) image condition
    assert e_pred < 5e-3 and e_loss < 2e-3 and worst[0] < 2e-2


@pytest.mark.gpu
def test_sdpa_small_causal_vs_torch():
    from gfe_hip.head_ops import sdpa_small
    g = torch.Generator().manual_seed(3)
    for (B, H, L, dh, causal) in ((2, 8, 37, 64, True), (3, 2, 5, 8, True), (1, 4, 64, 16, False), (2, 1, 1, 4, True)):
        q, k, v, w = (torch.randn(B, L, H * dh, generator=g) for _ in range(4))
        gq, gk, gv = (t.cuda().requires_grad_(True) for t in (q, k, v))
        cq, ck, cv = (t.double().requires_grad_(True) for t in (q, k, v))
        o = sdpa_small(gq, gk, gv, H, causal)
        sp = lambda t: t.view(B, L, H, dh).transpose(1, 2)
        ro = F.scaled_dot_product_attention(sp(cq), sp(ck), sp(cv), is_causal=causal).transpose(1, 2).reshape(B, L, H * dh)
        assert rel_err(o, ro) < 1e-5
        (o * w.cuda()).sum().backward(); (ro * w.double()).sum().backward()
        for a, b in ((gq, cq), (gk, ck), (gv, cv)):
            assert rel_err(a.grad, b.grad) < 1e-5


@pytest.mark.gpu
def test_sdpa_small_attention_probability_dropout():
    """`attn = self.dropout(softmax(...))` of the generator's ViT in training (vit_pytorch_diy/vit.py:59; ADVICE r02: it used to be dropped
    silently): the mask lives inside the kernel (a hash of (seed, element), regenerated in the backward).  Recover it with V = identity
    (the output rows ARE the dropped probabilities), then check output and gradients of a second call with the same seed against torch
    autograd with that mask, and the keep rate against p."""
    from gfe_hip.head_ops import sdpa_small
    g = torch.Generator().manual_seed(9)
    B, H, L, dh, p = 3, 4, 16, 16, 0.25
    q, k, v, w = (torch.randn(B, L, H * dh, generator=g) for _ in range(4))
    eye = torch.eye(L).view(1, L, 1, dh).expand(B, L, H, dh).reshape(B, L, H * dh).contiguous()
    torch.manual_seed(1234)
    dropped = sdpa_small(q.cuda(), k.cuda(), eye.cuda(), H, causal=False, dropout_p=p)          # (B, L, H*dh): row r of head h = dropped P[r, :]
    sp = lambda t: t.view(B, L, H, dh).transpose(1, 2)
    P = torch.softmax(sp(q.double()) @ sp(k.double()).transpose(-1, -2) * dh ** -0.5, dim=-1)
    mask = (sp(dropped.cpu()) != 0).double()
    assert rel_err(sp(dropped.cpu()), P * mask / (1 - p)) < 1e-5
    keep = mask.mean().item()
    assert abs(keep - (1 - p)) < 0.03, keep                                                       # 3072 draws: 3 sigma = 0.023
    gq, gk, gv = (t.cuda().requires_grad_(True) for t in (q, k, v))
    cq, ck, cv = (t.double().requires_grad_(True) for t in (q, k, v))
    torch.manual_seed(1234)                                                                       # the same seed -> the same mask
    o = sdpa_small(gq, gk, gv, H, causal=False, dropout_p=p)
    Pc = torch.softmax(sp(cq) @ sp(ck).transpose(-1, -2) * dh ** -0.5, dim=-1) * mask / (1 - p)
    ro = (Pc @ sp(cv)).transpose(1, 2).reshape(B, L, H * dh)
    assert rel_err(o, ro) < 1e-5
    (o * w.cuda()).sum().backward(); (ro * w.double()).sum().backward()
    for a, b in ((gq, cq), (gk, ck), (gv, cv)):
        assert rel_err(a.grad, b.grad) < 1e-5
    torch.manual_seed(99)                                                                         # another seed -> another mask; p = 0 -> none
    o2 = sdpa_small(q.cuda(), k.cuda(), eye.cuda(), H, causal=False, dropout_p=p)
    assert not torch.equal(o2 != 0, dropped != 0)
    assert (sdpa_small(q.cuda(), k.cuda(), eye.cuda(), H, causal=False) != 0).all()
