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


@pytest.mark.gpu
@pytest.mark.parametrize("T,D,Fd,E,K", [(111, 64, 128, 16, 2), (7, 32, 48, 4, 1), (300, 64, 96, 16, 3), (37, 24, 40, 5, 2)])
def test_moe_mlp_vs_the_references_expert_loop(T, D, Fd, E, K):
    """gfe_hip.moe_ops.moe_mlp (device routing + expert sort, grouped GEMMs, gather combine) against the reference's algorithm written out in
    torch f64 -- softmax -> topk -> per-expert down(silu(gate x) * up x) * routing weight -> index_add (jamba.py:484-517) -- output, router
    logits and every gradient (x, router, all 3 E expert matrices; experts that receive no token get a zero gradient), ragged sizes included."""
    from gfe_hip.moe_ops import moe_mlp
    g = torch.Generator().manual_seed(T + E)
    x = torch.randn(T, D, generator=g)
    wr = torch.randn(E, D, generator=g) * 0.5
    wg = [torch.randn(Fd, D, generator=g) / D ** 0.5 for _ in range(E)]
    wu = [torch.randn(Fd, D, generator=g) / D ** 0.5 for _ in range(E)]
    wd = [torch.randn(D, Fd, generator=g) / Fd ** 0.5 for _ in range(E)]
    w = torch.randn(T, D, generator=g)
    if E == 5:
        wr[4] = -10.0 * wr[0].abs() - 5                          # an expert nobody is routed to (its logit is always far below)
        x = x.abs()
    dev = lambda t: t.clone().to("cuda").requires_grad_(True)
    gx, gwr, gwg, gwu, gwd = dev(x), dev(wr), [dev(t) for t in wg], [dev(t) for t in wu], [dev(t) for t in wd]
    out, logits = moe_mlp(gx, K, gwr, gwg, gwu, gwd)
    (out * w.cuda()).sum().backward()
    dd = lambda t: t.clone().double().requires_grad_(True)
    cx, cwr, cwg, cwu, cwd = dd(x), dd(wr), [dd(t) for t in wg], [dd(t) for t in wu], [dd(t) for t in wd]
    rl = cx @ cwr.t()
    rw, sel = torch.topk(F.softmax(rl, dim=1), K, dim=-1)
    ref = torch.zeros(T, D, dtype=torch.float64)
    for e in range(E):
        idx, top_x = torch.where(F.one_hot(sel, E).permute(2, 1, 0)[e])
        if top_x.numel():
            xe = cx[top_x]
            ref = ref.index_add(0, top_x, (F.silu(xe @ cwg[e].t()) * (xe @ cwu[e].t())) @ cwd[e].t() * rw[top_x, idx, None])
    (ref * w.double()).sum().backward()
    assert rel_err(logits, rl) < 1e-5 and rel_err(out, ref) < 1e-5
    assert rel_err(gx.grad, cx.grad) < 1e-5 and rel_err(gwr.grad, cwr.grad) < 1e-4
    used = 0
    for e in range(E):
        for a, b in ((gwg[e], cwg[e]), (gwu[e], cwu[e]), (gwd[e], cwd[e])):
            if b.grad is None or float(b.grad.abs().max()) == 0.0:
                assert float(a.grad.abs().max()) == 0.0, e
            else:
                used += 1
                assert rel_err(a.grad, b.grad) < 1e-5, e
    assert used >= 3


@pytest.mark.gpu
def test_cross_jamba_both_at_the_classify_configuration_vs_reference_fixture():
    """Fixture t8: the reference's Cross_jamba_both at classify_mamba.py's configuration (dim 512, depth 6 -> 12 layers, heads 8, 16-expert
    top-2 MoE on the odd layers, 208.5 M parameters, default d_cross = 160*160), two samples with native 160x160x96 image conditions:
    logits, loss and every parameter gradient (norm + 16-element slice).  Weights and inputs regenerate from the deterministic initialiser."""
    import zlib
    import gfe_hip.det_init as det
    from cross_atten.mamba_transformer import Cross_jamba_both
    fx = golden("t8_jamba_classify.npz")
    cards, n_cont, dim, depth, heads, vol, Bn = (11, 2, 2, 4, 4, 3, 3), 25, 512, 6, 8, (160, 160, 96), 2
    ft = Cross_jamba_both(categories=cards, num_continuous=n_cont, dim=dim, depth=depth, heads=heads, dim_head=dim // heads)
    assert sum(p.numel() for p in ft.parameters()) == int(fx["nparams"])
    ft.load_state_dict(det.det_state_dict(ft.state_dict(), seed=81, prefix="jam8."))
    x, x_cat, x_num, y = det.det_inputs(Bn, vol, cards, n_cont, seed=81)
    rnd = lambda key, shape: torch.from_numpy(np.random.Generator(np.random.Philox(key=[zlib.crc32(key.encode()), 12345])).standard_normal(shape).astype(np.float32))
    pet, feat = rnd("jam8.pet", (Bn, 1) + vol), rnd("jam8.feat", (Bn, 4, dim))
    ft = ft.cuda().eval()
    pred = ft(x_cat.cuda(), x_num.cuda(), feat.cuda(), [x.cuda(), pet.cuda()])
    e_pred = rel_err(pred, tt(fx["pred"]))
    loss = F.binary_cross_entropy(torch.sigmoid(pred.squeeze(1)), y.cuda().float())
    e_loss = abs(loss.item() - float(fx["loss"]))
    loss.backward()
    errs, unused = [], 0
    for k, p in ft.named_parameters():
        ref = float(fx["gnorm." + k])
        if ref < 1e-9:                                  # experts no token was routed to (reference gradient None); k_proj.bias (exactly
            unused += ".experts." in k                  # zero in exact arithmetic: softmax is shift-invariant; round-off on both sides)
            assert p.grad is None or float(p.grad.abs().max()) < 1e-8, k
            continue
        e_n = abs(float(p.grad.double().norm()) - ref) / ref
        e_s = rel_err(_slices(p.grad, 16), tt(fx["gslice." + k]))
        errs.append((max(e_n, e_s), k))
    errs.sort(reverse=True)
    print("Cross_jamba_both (classify configuration, 208.5 M parameters) vs reference: logits %.2e, loss %.2e, gradients: median %.2e, worst %s; "
          "%d expert matrices unused" % (e_pred, e_loss, errs[len(errs) // 2][0], [("%.1e" % e, k) for e, k in errs[:4]], unused))
    # f32 everywhere except the K / V projections over the (bf16) image condition
    assert e_pred < 5e-3 and e_loss < 2e-3
    assert errs[len(errs) // 2][0] < 5e-3 and errs[0][0] < 3e-2


@pytest.mark.gpu
def test_jamba_cached_decoding_reproduces_the_full_forward():
    """Jamba.step / MambaLayer / AttentionSDPA with caches (cross_atten/jamba.py:298-306, 373-383, 421-423; VERDICT r03 missing #2): feeding
    the tokens one at a time through the caches (Mamba conv window + state on the step kernels, a KV cache in the reference's
    (B, kv heads, T, head dim) layout with the one-query attention kernel) must reproduce the full-sequence forward position by position:
    the Mamba recurrence is causal, the cached attention of token t sees keys 0..t = the causal mask's row t, the MLPs are per token."""
    from cross_atten.jamba import Jamba, JambaLMConfig
    import gfe_hip.det_init as det
    cfg = JambaLMConfig(d_model=64, n_layers=6, mlp_size=128, num_attention_heads=8, num_key_value_heads=8, num_experts=4, num_experts_per_tok=2,
                        inner_layernorms=True)
    m = Jamba(cfg)
    m.load_state_dict(det.det_state_dict(m.state_dict(), seed=43, prefix="jamstep."))
    m = m.cuda().eval()
    kinds = [type(l).__name__ for l in m.layers]
    assert "AttentionLayer" in kinds and "MambaLayer" in kinds
    B, L = 3, 11
    x = torch.randn(B, L, 64, generator=torch.Generator().manual_seed(2)).cuda()
    with torch.no_grad():
        full, _ = m(x)
    caches = [l.get_empty_cache(B, x.device) for l in m.layers]
    outs = []
    for t in range(L):                                   # (no torch.no_grad here: the step path must not need it)
        o, caches = m.step(x[:, t:t + 1].contiguous(), caches)
        outs.append(o)
    got = torch.cat(outs, dim=1)
    e = rel_err(got, full)
    print("Jamba token-by-token cached decoding vs the full forward: %.2e" % e)
    assert e < 1e-4
    ia = kinds.index("AttentionLayer")
    k_cache, v_cache = caches[ia]
    assert k_cache.shape == (B, 8, L, 8) and v_cache.shape == (B, 8, L, 8)        # the reference's cache layout (jamba.py:366-383)
    im = kinds.index("MambaLayer")
    assert caches[im][0].shape == (B, cfg.d_inner, cfg.d_state) and caches[im][1].shape == (B, cfg.d_inner, cfg.d_conv - 1)


@pytest.mark.gpu
def test_attention_layer_takes_several_tokens_onto_a_warm_kv_cache():
    """VERDICT r04 missing #3: the reference's AttentionSDPA takes any number of new tokens with a cache (jamba.py:373-392: K / V appended, SDPA with
    is_causal=False); ours raised beyond one.  Three tokens onto a cache warmed with five: against the reference's formula in f64."""
    import math
    from cross_atten.jamba import AttentionSDPA, JambaLMConfig
    cfg = JambaLMConfig(d_model=64, n_layers=2, mlp_size=96, num_attention_heads=8, num_key_value_heads=4)
    torch.manual_seed(4)
    att = AttentionSDPA(cfg).cuda().eval()
    B, T0, L, H, Hkv, dh = 2, 5, 3, 8, 4, 8
    x0, x1 = torch.randn(B, T0, 64).cuda(), torch.randn(B, L, 64).cuda()
    with torch.no_grad():
        _, cache = att(x0, (None, None))
        o, cache2 = att(x1, cache)
    assert cache2[0].shape == (B, Hkv, T0 + L, dh)
    w = {k: v.detach().double().cpu() for k, v in att.named_parameters()}
    xa = torch.cat([x0, x1], dim=1).double().cpu()
    q = (x1.double().cpu() @ w["q_proj.weight"].t()).view(B, L, H, dh).transpose(1, 2)
    k = (xa @ w["k_proj.weight"].t()).view(B, T0 + L, Hkv, dh).transpose(1, 2).repeat_interleave(H // Hkv, dim=1)
    v = (xa @ w["v_proj.weight"].t()).view(B, T0 + L, Hkv, dh).transpose(1, 2).repeat_interleave(H // Hkv, dim=1)
    ref = (torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(dh), dim=-1) @ v).transpose(1, 2).reshape(B, L, H * dh) @ w["o_proj.weight"].t()
    assert rel_err(o, ref.float()) < 1e-5


@pytest.mark.gpu
def test_router_logits_are_differentiable_like_the_references():
    """SparseMoEBlock returns its router logits (jamba.py:517) and load_balancing_loss (jamba.py:537-556) differentiates through
    them; the fused MoE node must pass that gradient on to the router weight and the tokens (alone, and together with the output's)."""
    from cross_atten.jamba import JambaLMConfig, SparseMoEBlock, load_balancing_loss
    cfg = JambaLMConfig(d_model=64, n_layers=2, mlp_size=96, num_experts=4, num_experts_per_tok=2)
    g = torch.Generator().manual_seed(9)
    blk = SparseMoEBlock(cfg, num_experts=4, num_experts_per_tok=2).cuda()
    x = torch.randn(2, 9, 64, generator=g).cuda().requires_grad_(True)
    out, rl = blk(x)
    assert rl.requires_grad
    load_balancing_loss([rl], 4, 2).backward()                                          # the balance term ALONE
    x64 = x.detach().double().cpu().requires_grad_(True)
    w64 = blk.router.weight.detach().double().cpu().requires_grad_(True)
    rl64 = x64.reshape(-1, 64) @ w64.t()
    assert rel_err(rl, rl64.float()) < 1e-5
    load_balancing_loss([rl64], 4, 2).backward()
    assert rel_err(blk.router.weight.grad, w64.grad.float()) < 1e-4 and rel_err(x.grad, x64.grad.float()) < 1e-4
    for e in blk.experts:                                                                # nothing flows into the experts from this term
        assert e.gate_proj.weight.grad is None or e.gate_proj.weight.grad.abs().max() == 0
