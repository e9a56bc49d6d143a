"""GPU parity of the trainable path (Mamba, cross-attention, GEGLU FF, head Linear, clip+Adam, whole step) against the
reference-generated fixtures and the oracle.  The head's Linears keep f32 operands (exact-f32 MFMA GEMM, as the reference trains
the head in fp32): op-level tolerance 1e-3 (BASELINE.json north_star, fp32); only the K / V projections over the image condition
and the frozen generator are bf16 -> 1e-2-class tolerances for whatever includes them."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import golden, rel_err, sub_sd, tt
from oracle import ref_ops as O

pytestmark = pytest.mark.gpu
DEV = "cuda"
BF = torch.bfloat16


def test_mamba_stack_vs_reference_fixture():
    from cross_atten.mamba import Mamba, MambaConfig
    fx = golden("t0_mamba.npz")
    m = Mamba(MambaConfig(d_model=32, n_layers=2))
    m.load_state_dict(sub_sd(fx, "sd."))
    m = m.to(DEV)
    x = tt(fx["x"], device=DEV).requires_grad_(True)
    TOL = 1e-3                                      # fp32 tolerance of BASELINE.json: every GEMM of the block keeps f32 operands
    assert rel_err(m.layers[0].norm(x), tt(fx["y_norm0"])) < 1e-5
    e_block = rel_err(m.layers[0].mixer(x), tt(fx["y_block0"]))
    y = m(x)
    e_y = rel_err(y, tt(fx["y"]))
    (y * tt(fx["w"], device=DEV)).sum().backward()
    e_gx = rel_err(x.grad, tt(fx["gx"]))
    e_g = {k: rel_err(p.grad, tt(fx["g." + k])) for k, p in m.named_parameters()}
    kw = max(e_g, key=e_g.get)
    print("Mamba stack vs reference: block %.2e, stack %.2e, dx %.2e, worst parameter gradient %.2e (%s)" % (e_block, e_y, e_gx, e_g[kw], kw))
    assert e_block < TOL and e_y < TOL and e_gx < TOL and e_g[kw] < TOL


def test_token_by_token_step_reproduces_the_references_forward():
    """Row A11: Mamba.step / MambaBlock.step / ssm_step (cross_atten/mamba.py:342-405) on the kernels (gfe_mamba_step_conv / _ssm + the
    exact-f32 GEMM).  Fed one token at a time the stack must reproduce, position by position, the full-sequence forward that the REFERENCE
    computed (fixture t0_mamba.npz `y`) -- the known-answer relation SURVEY section 4 names for ssm_step -- and agree with our own fused
    full-sequence path."""
    from cross_atten.mamba import Mamba, MambaConfig
    fx = golden("t0_mamba.npz")
    cfg = MambaConfig(d_model=32, n_layers=2)
    m = Mamba(cfg)
    m.load_state_dict(sub_sd(fx, "sd."))
    m = m.to(DEV)
    x, y_ref = tt(fx["x"], device=DEV), tt(fx["y"])
    B, L, _ = x.shape
    # caches as the reference builds them for inference (mamba.py:330-340): (h = None, the last d_conv - 1 conv inputs = zeros)
    caches = [(None, torch.zeros(B, cfg.d_inner, cfg.d_conv - 1, device=DEV)) for _ in range(cfg.n_layers)]
    outs = []
    with torch.no_grad():
        for t in range(L):
            o, caches = m.step(x[:, t].contiguous(), caches)
            outs.append(o)
        y = torch.stack(outs, 1)
        y_full = m(x)
    e_ref, e_full = rel_err(y, y_ref), rel_err(y, y_full)
    print("token-by-token step vs the reference's forward %.2e, vs the fused full-sequence path %.2e" % (e_ref, e_full))
    assert e_ref < 1e-5 and e_full < 1e-5
    # the state the steps leave behind is the scan's final state: one more token must continue the sequence, not restart it
    warm = list(caches)                                  # (Mamba.step writes the new caches into the list it is given)
    with torch.no_grad():
        o2, caches2 = m.step(x[:, 0].contiguous(), list(warm))
        y1, h1 = m.layers[0].mixer.ssm_step(torch.ones(B, cfg.d_inner, device=DEV), None)
        y2, h2 = m.layers[0].mixer.ssm_step(torch.ones(B, cfg.d_inner, device=DEV), h1)
    assert not torch.allclose(o2, outs[0]) and caches2[0][0].shape == (B, cfg.d_inner, cfg.d_state)
    assert torch.isfinite(y2).all() and not torch.allclose(y1, y2)
    # Grad mode on and parameters that require grad: the reference's step is plain differentiable torch code, so is ours (VERDICT r05 missing #4):
    # same values as the kernel path, and a graph behind them
    o3, _ = m.step(x[:, 0].contiguous(), list(warm))
    assert rel_err(o3, o2) < 1e-5 and o3.requires_grad
    # ... through which the token-by-token pass trains: the gradients of sum(y * w) w.r.t. the input and every parameter equal the ones the
    # REFERENCE's full-sequence forward produced (fixture t0_mamba.npz: back-propagation through the caches == through the scan)
    for p_ in m.parameters():
        p_.grad = None
    xg = x.clone().requires_grad_(True)
    caches = [(None, torch.zeros(B, cfg.d_inner, cfg.d_conv - 1, device=DEV)) for _ in range(cfg.n_layers)]
    outs = []
    for t in range(L):
        o, caches = m.step(xg[:, t], caches)
        outs.append(o)
    ys = torch.stack(outs, 1)
    assert rel_err(ys, y_ref) < 1e-5
    (ys * tt(fx["w"], device=DEV)).sum().backward()
    e_gx = rel_err(xg.grad, tt(fx["gx"]))
    e_g = {k: rel_err(p_.grad, tt(fx["g." + k])) for k, p_ in m.named_parameters()}
    kw = max(e_g, key=e_g.get)
    print("token-by-token step under autograd vs the reference's gradients: dx %.2e, worst parameter %.2e (%s)" % (e_gx, e_g[kw], kw))
    assert e_gx < 1e-3 and e_g[kw] < 1e-3              # fp32 tolerance of BASELINE.json
    # with no graph to record the step kernels run and the result carries none
    frozen = [p_.requires_grad for p_ in m.parameters()]
    for p_ in m.parameters():
        p_.requires_grad_(False)
    o4, _ = m.step(x[:, 0].contiguous(), list(warm))
    assert torch.equal(o4, o2) and not o4.requires_grad
    for p_, f in zip(m.parameters(), frozen):
        p_.requires_grad_(f)


def test_cross_attention_ff_embedder_vs_reference_fixture():
    from cross_atten.corss_ft_transformer import FeedForward, NumericalEmbedder
    from cross_atten.sd_cross_atten import CrossAttention
    fx = golden("t0_head_ops.npz")
    ca = CrossAttention(n_heads=2, d_embed=16, d_cross=24)
    ca.load_state_dict(sub_sd(fx, "ca.sd."))
    ca = ca.to(DEV)
    x, y = tt(fx["ca.x"], device=DEV).requires_grad_(True), tt(fx["ca.y"], device=DEV).requires_grad_(True)
    TOL = 1e-3
    o = ca(x, y)
    assert rel_err(o, tt(fx["ca.out"])) < TOL
    (o * tt(fx["ca.w"], device=DEV)).sum().backward()
    assert rel_err(x.grad, tt(fx["ca.gx"])) < TOL and rel_err(y.grad, tt(fx["ca.gy"])) < TOL
    for k, p in ca.named_parameters():
        if k == "k_proj.bias":
            assert p.grad.abs().max() < 1e-5       # exactly zero in exact arithmetic
            continue
        assert rel_err(p.grad, tt(fx["ca.g." + k])) < TOL, k
    ff = FeedForward(16, mult=2, dropout=0.1)
    ff.load_state_dict(sub_sd(fx, "ff.sd."))
    ff = ff.to(DEV).eval()
    xf = tt(fx["ff.x"], device=DEV).requires_grad_(True)
    of = ff(xf)
    assert rel_err(of, tt(fx["ff.out"])) < TOL
    (of * tt(fx["ca.w"], device=DEV)).sum().backward()
    assert rel_err(xf.grad, tt(fx["ff.gx"])) < TOL
    ne = NumericalEmbedder(16, 5)
    ne.load_state_dict(sub_sd(fx, "ne.sd."))
    assert rel_err(ne.to(DEV)(tt(fx["ne.x"], device=DEV)), tt(fx["ne.out"])) < 1e-6


def _cross_attention_f64(params, x, y, heads):
    """sd_cross_atten.py:49-70 in f64 (K and V materialised, as the reference does)"""
    import math
    q = x @ params["q_proj.weight"].t() + params["q_proj.bias"]
    k = y @ params["k_proj.weight"].t() + params["k_proj.bias"]
    v = y @ params["v_proj.weight"].t() + params["v_proj.bias"]
    B, Lq, E = q.shape
    dh = E // heads
    q, k, v = (t.view(B, -1, heads, dh).transpose(1, 2) for t in (q, k, v))
    w = torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(dh), dim=-1)
    o = (w @ v).transpose(1, 2).reshape(B, Lq, E)
    return o @ params["out_proj.weight"].t() + params["out_proj.bias"]


def test_folded_one_query_cross_attention_vs_reference_fixture():
    """CrossAttention with one query per sample and a condition that wants no gradient takes the folded kernels (csrc/xattn_fold.hip, no K / V):
    the reference's own fixture (forward, dx, every parameter gradient) at f32-exact tolerance -- the materialised path the fixture test
    above keeps exercising (y.requires_grad) needs bf16-free sizes to reach that."""
    from cross_atten.sd_cross_atten import CrossAttention
    fx = golden("t0_head_ops.npz")
    ca = CrossAttention(n_heads=2, d_embed=16, d_cross=24)
    ca.load_state_dict(sub_sd(fx, "ca.sd."))
    ca = ca.to(DEV)
    x, y = tt(fx["ca.x"], device=DEV).requires_grad_(True), tt(fx["ca.y"], device=DEV)
    o = ca(x, y)
    assert rel_err(o, tt(fx["ca.out"])) < 2e-6
    (o * tt(fx["ca.w"], device=DEV)).sum().backward()
    assert rel_err(x.grad, tt(fx["ca.gx"])) < 5e-6
    for k, p in ca.named_parameters():
        if k == "k_proj.bias":
            assert p.grad is not None and p.grad.abs().max() == 0          # exactly zero: a per-head constant cannot move a softmax
            continue
        assert rel_err(p.grad, tt(fx["ca.g." + k])) < 5e-6, k


@pytest.mark.parametrize("B,heads,dh,hw,d3,nimg", [
    (3, 8, 64, (96, 96), 96, 2),        # the bench geometry: d_cross 9216, 192 keys (16-byte vector path)
    (9, 8, 8, (32, 32), 32, 2),         # T1's reduced head (dim 64), more samples than one register pass
    (2, 2, 8, (5, 5), 6, 3),            # nothing divisible by 4: scalar path, three images
    (1, 16, 4, (8, 20), 12, 1),         # more heads than one pass of the condition kernels
])
def test_folded_one_query_cross_attention_vs_f64_math(B, heads, dh, hw, d3, nimg):
    """gfe_cross_attn_q1_folded_{fwd,bwd} through CrossAttention on a Condition of f32 volumes against the reference's formula in f64 with K
    and V materialised: output, dx, every parameter gradient; gradients ADD into existing ones; two runs are bit-identical."""
    from cross_atten.sd_cross_atten import CrossAttention
    from gfe_hip.train_ops import Condition
    g = torch.Generator().manual_seed(B * 100 + d3)
    E, HW = heads * dh, hw[0] * hw[1]
    ca = CrossAttention(n_heads=heads, d_embed=E, d_cross=HW)
    with torch.no_grad():
        ca.k_proj.weight.mul_(3.0)                                            # scores of order 1, not 1e-2: a softmax that actually selects
    vols = [torch.randn(B, 1, hw[0], hw[1], d3, generator=g) for _ in range(nimg)]
    x = torch.randn(B, 1, E, generator=g)
    w = torch.randn(B, 1, E, generator=g)
    p64 = {k: v.detach().double().requires_grad_(True) for k, v in ca.named_parameters()}
    x64 = x.double().requires_grad_(True)
    y64 = torch.cat([v.double().view(B, HW, d3).transpose(1, 2) for v in vols], dim=1)          # 'b c h w d -> b (c d) (h w)', images concatenated
    ref = _cross_attention_f64(p64, x64, y64, heads)
    (ref * w.double()).sum().backward()
    ca = ca.to(DEV)
    xd = x.to(DEV).requires_grad_(True)
    outs = []
    for rep in range(2):
        for p in ca.parameters():
            p.grad = None
        xd.grad = None
        o = ca(xd, Condition([v.to(DEV) for v in vols]))
        (o * w.to(DEV)).sum().backward()
        outs.append([o.detach().clone(), xd.grad.clone()] + [p.grad.clone() for p in ca.parameters()])
    assert all(torch.equal(a, b) for a, b in zip(*outs)), "the folded cross-attention is not bit-reproducible"
    assert rel_err(o, ref.float()) < 5e-6
    assert rel_err(xd.grad, x64.grad.float()) < 2e-5
    for k, p in ca.named_parameters():
        if k == "k_proj.bias":
            assert p.grad.abs().max() == 0 and p64[k].grad.abs().max() < 1e-12 * max(1.0, float(p64["k_proj.weight"].grad.abs().max()))
            continue
        assert rel_err(p.grad, p64[k].grad.float()) < 2e-5, k
    g0 = {k: p.grad.clone() for k, p in ca.named_parameters()}                # accumulate semantics: a second backward doubles every gradient
    o = ca(xd, Condition([v.to(DEV) for v in vols]))
    (o * w.to(DEV)).sum().backward()
    for k, p in ca.named_parameters():
        assert rel_err(p.grad, 2 * g0[k]) < 1e-6 or g0[k].abs().max() == 0, k


@pytest.mark.parametrize("causal", [False, True])
def test_self_attention_on_the_small_sdpa_kernel_vs_f64_math(causal):
    """SelfAttention (sd_cross_atten.py:7-37) runs on gfe_sdpa_small (no torch-math attention left in the module): forward and every
    gradient against the reference's formula evaluated in f64; sizes outside the kernel raise instead of leaving the HIP path."""
    import math
    from cross_atten.sd_cross_atten import CrossAttention, SelfAttention
    g = torch.Generator().manual_seed(3)
    sa = SelfAttention(n_heads=4, d_embed=64).to(DEV)
    x = torch.randn(3, 37, 64, generator=g).to(DEV).requires_grad_(True)
    w = torch.randn(3, 37, 64, generator=g).to(DEV)
    o = sa(x, causal_mask=causal)
    (o * w).sum().backward()
    p64 = {k: v.detach().double().cpu().requires_grad_(True) for k, v in sa.named_parameters()}
    x64 = x.detach().double().cpu().requires_grad_(True)
    qkv = x64 @ p64["in_proj.weight"].t() + p64["in_proj.bias"]
    q, k, v = [t.view(3, 37, 4, 16).transpose(1, 2) for t in qkv.chunk(3, dim=-1)]
    wt = q @ k.transpose(-1, -2)
    if causal:
        wt = wt.masked_fill(torch.ones_like(wt, dtype=torch.bool).triu(1), -torch.inf)
    wt = torch.softmax(wt / math.sqrt(16), dim=-1)
    ref = (wt @ v).transpose(1, 2).reshape(3, 37, 64) @ p64["out_proj.weight"].t() + p64["out_proj.bias"]
    (ref * w.double().cpu()).sum().backward()
    assert rel_err(o, ref.float()) < 1e-5
    assert rel_err(x.grad, x64.grad.float()) < 1e-5
    for kname, pp in sa.named_parameters():
        assert rel_err(pp.grad, p64[kname].grad.float()) < 1e-5, kname
    with pytest.raises(NotImplementedError):
        sa(torch.randn(1, 65, 64, device=DEV))
    # CrossAttention stays general: several queries per sample run the one-query kernel over (sample, query) pairs
    ca = CrossAttention(n_heads=2, d_embed=16, d_cross=24).to(DEV)
    xq, yc = torch.randn(2, 3, 16, generator=g).to(DEV).requires_grad_(True), torch.randn(2, 6, 24, generator=g).to(DEV).requires_grad_(True)
    wq = torch.randn(2, 3, 16, generator=g).to(DEV)
    oq = ca(xq, yc)
    (oq * wq).sum().backward()
    pc = {k_: v_.detach().double().cpu().requires_grad_(True) for k_, v_ in ca.named_parameters()}
    x64q, y64q = xq.detach().double().cpu().requires_grad_(True), yc.detach().double().cpu().requires_grad_(True)
    refq = _cross_attention_f64(pc, x64q, y64q, 2)
    (refq * wq.double().cpu()).sum().backward()
    assert rel_err(oq, refq.float()) < 1e-5 and rel_err(xq.grad, x64q.grad.float()) < 1e-5 and rel_err(yc.grad, y64q.grad.float()) < 1e-5
    for kname, pp in ca.named_parameters():
        if kname != "k_proj.bias":
            assert rel_err(pp.grad, pc[kname].grad.float()) < 1e-5, kname


@pytest.mark.parametrize("n,dt", [(13832 * 16 + 3, torch.float32), (4096 * 7 + 5, torch.bfloat16), (3, torch.float32)])
def test_gelu_and_silu_mul_operators_vs_torch(n, dt):
    """gfe_gelu_fwd/bwd (exact erf; the training paths' nn.GELU) and silu_mul (the un-routed Jamba MLP, jamba.py:535) against torch's own
    functions in f64, forward and backward, vector body + scalar tail, f32 and bf16, also on a 4-byte-aligned (not 16-byte-aligned) view."""
    from gfe_hip.head_ops import gelu, silu_mul
    g = torch.Generator().manual_seed(n)
    x = (torch.randn(n + 1, generator=g) * 2).to(dt).to(DEV)
    for off in (0, 1):
        xi = x[off:off + n].detach().requires_grad_(True)
        w = torch.randn(n, generator=g).to(dt).to(DEV)
        y = gelu(xi)
        (y.float() * w.float()).sum().backward()
        x64 = xi.detach().double().cpu().requires_grad_(True)
        r = F.gelu(x64)
        (r * w.double().cpu()).sum().backward()
        tol = 1e-6 if dt == torch.float32 else 8e-3
        assert rel_err(y.float(), r.float()) < tol and rel_err(xi.grad.float(), x64.grad.float()) < tol
    if dt == torch.float32:
        a = torch.randn(n, generator=g).to(DEV).requires_grad_(True)
        b = torch.randn(n, generator=g).to(DEV).requires_grad_(True)
        h = silu_mul(a, b)
        (h * x[:n].float()).sum().backward()
        a64, b64 = a.detach().double().cpu().requires_grad_(True), b.detach().double().cpu().requires_grad_(True)
        r = F.silu(a64) * b64
        (r * x[:n].double().cpu()).sum().backward()
        assert rel_err(h, r.float()) < 1e-6 and rel_err(a.grad, a64.grad.float()) < 2e-6 and rel_err(b.grad, b64.grad.float()) < 1e-6


@pytest.mark.parametrize("n,dt", [(13832 * 16 + 3, torch.float32), (4096 * 7 + 5, torch.bfloat16), (5, torch.float32)])
def test_dropout_operator_mask_statistics_and_backward(n, dt):
    """gfe_dropout (nn.Dropout in the training twins, vit.py:24-27 / vit_3d.py:25-28): kept values are x / (1 - p) exactly as torch rounds them,
    the kept fraction is 1 - p within 5 sigma, the backward regenerates the SAME mask, two calls draw different masks, eval / p = 0 are the identity,
    vector body + scalar tail + a view that is not 16-byte aligned."""
    from gfe_hip.head_ops import dropout
    g = torch.Generator().manual_seed(n)
    p = 0.25
    x = (torch.randn(n + 1, generator=g).abs() + 0.5).to(dt).to(DEV)
    masks = []
    for off in (0, 1):
        xi = x[off:off + n].detach().requires_grad_(True)
        y = dropout(xi, p, True)
        w = (torch.randn(n, generator=g).abs() + 0.5).to(dt).to(DEV)
        (y.float() * w.float()).sum().backward()
        keep = y != 0
        assert torch.equal(y[keep], (xi.detach().float()[keep] * (1.0 / (1.0 - p))).to(dt))
        assert torch.equal(xi.grad != 0, keep) and torch.equal(xi.grad[keep], (w.float()[keep] * (1.0 / (1.0 - p))).to(dt))
        if n > 1000:
            assert abs(keep.float().mean().item() - (1 - p)) < 5 * (p * (1 - p) / n) ** 0.5
        masks.append(keep)
    if n > 1000:
        assert (masks[0] != masks[1]).float().mean().item() > 0.2           # another call, another mask
    assert dropout(x, p, False) is x and dropout(x, 0.0, True) is x


def test_embedding_offsets_bit_exact():
    from cross_atten.mamba_transformer import Cross_mamba_both
    fx = golden("t0_head_ops.npz")
    m = Cross_mamba_both(categories=(11, 2, 2, 4, 4, 3, 3), num_continuous=25, dim=16, depth=1, heads=2, d_cross=64)
    assert m.categories_offset.tolist() == fx["categories_offset"].tolist()


def test_condition_layouts_bit_exact():
    """'b c h w d -> (b c) (h w) d' + transpose(1,2) + cat (mamba_transformer.py:89-94), and its GEMM-transposed twin."""
    from gfe_hip.train_ops import Condition
    g = torch.Generator().manual_seed(0)
    x = torch.randn(3, 1, 8, 12, 20, generator=g)
    p = torch.randn(3, 1, 8, 12, 20, generator=g)
    c = Condition([x.to(DEV), p.to(DEV)])
    ref = O.build_condition([x, p]).to(BF)
    assert torch.equal(c.cond.cpu(), ref)
    assert torch.equal(c.condT.cpu(), ref.reshape(-1, ref.shape[-1]).t().contiguous())
    fx = golden("t0_index_maps.npz")           # the reference's own rearrange on an arange volume
    vol = torch.arange(2 * 4 * 6 * 8, dtype=torch.float32).view(2, 1, 4, 6, 8) % 251
    c2 = Condition([vol.to(DEV)])
    assert torch.equal(c2.cond.cpu().float(), (tt(fx["condition_2x1x4x6x8"]) % 251).float())


def test_mid_linear_fwd_and_wgrad():
    from gfe_hip.train_ops import mid_linear
    g = torch.Generator().manual_seed(1)
    B, H, W, C, S = 3, 24, 20, 256, 4
    a = torch.randn(B, H, W, C, generator=g).to(BF)
    b = torch.randn(B, H, W, C, generator=g).to(BF)
    w = (torch.randn(S, H * W, generator=g) / 20).requires_grad_(True)
    bias = torch.randn(S, generator=g).requires_grad_(True)
    wd, bd = w.detach().to(DEV).requires_grad_(True), bias.detach().to(DEV).requires_grad_(True)
    out = mid_linear(a.to(DEV), b.to(DEV), wd, bd)
    cat = torch.cat([a.float().permute(0, 3, 1, 2), b.float().permute(0, 3, 1, 2)], 1).flatten(2)      # classifier.py:330
    ref = cat @ w.t() + bias
    assert rel_err(out, ref) < 1e-4
    gw = torch.randn(B, 2 * C, S, generator=g)
    out.backward(gw.to(DEV))
    ref.backward(gw)
    assert rel_err(wd.grad, w.grad) < 1e-4 and rel_err(bd.grad, bias.grad) < 1e-5


def test_flat_clip_adam_matches_per_parameter_reference():
    from gfe_hip.train_ops import FlatAdam
    g = torch.Generator().manual_seed(2)
    shapes = [(512, 300), (7,), (1, 1, 64), (33, 5), (20000,)]
    ps = [torch.nn.Parameter(torch.randn(s, generator=g).to(DEV)) for s in shapes]
    ref_p = [p.detach().cpu().clone() for p in ps]
    ms = [torch.zeros_like(p) for p in ref_p]
    vs = [torch.zeros_like(p) for p in ref_p]
    opt = FlatAdam(ps, lr=1e-2)
    for step in range(1, 4):
        opt.zero_grad()
        grads = [torch.randn(s, generator=g) * (10.0 if i % 2 else 0.01) for i, s in enumerate(shapes)]
        for p, gr in zip(ps, grads):
            p.grad.add_(gr.to(DEV))
        opt.step()
        clipped = O.clip_per_param(grads)
        for i in range(len(ps)):
            ref_p[i], ms[i], vs[i] = O.adam_step(ref_p[i], clipped[i], ms[i], vs[i], step, lr=1e-2)
            assert rel_err(ps[i], ref_p[i]) < 1e-5, (step, i)
    # the bf16 shadows follow the parameters
    assert torch.equal(opt.flat_p16.float(), opt.flat_p.to(BF).float())


def _slices(t, n=256):
    """tools/make_golden.py::slices -- the strided sample the fixtures hold of every large tensor."""
    f = t.detach().reshape(-1)
    return f[::max(1, f.numel() // n)][:n]


def _sample_idx(numel, n=1024):
    """tools/make_golden.py::sample_idx -- the seeded element sample the fixtures hold of every gradient (`gsample.*`)."""
    import numpy as np
    if numel <= n:
        return torch.arange(numel)
    return torch.from_numpy(np.sort(np.random.default_rng(numel).choice(numel, size=n, replace=False)))


def _grad_sample_stats(fx, names, params):
    """The end-to-end gradient statistic with a variance (VERDICT r05 #6a; it replaces "worst of 64 strided elements", an extreme-value
    statistic of one realisation of the generator's bf16 rounding noise): every tensor is compared on up to 1 024 seeded random elements
    (`gsample.*`, from the reference's autograd), each error normalised by the tensor's largest reference element (`gamax.*`).  Returned:
    the median / 95th / 99th percentile over ALL sampled elements of all tensors (~1e5 values), and the largest per-tensor 95th percentile
    among tensors with at least 64 sampled elements."""
    pooled, per_tensor = [], []
    for k, p in zip(names, params):
        if float(fx["gnorm." + k]) < 1e-7 or ("gsample." + k) not in fx:
            continue
        idx = _sample_idx(p.grad.numel()).to(p.grad.device)
        a = p.grad.detach().reshape(-1)[idx].double().cpu()
        e = (a - tt(fx["gsample." + k]).double()).abs() / float(fx["gamax." + k])
        pooled.append(e)
        if e.numel() >= 64:                                # (a 4-element bias has no percentile: its norm is checked instead)
            per_tensor.append((torch.quantile(e, 0.95).item(), k))
    allv = torch.cat(pooled)
    per_tensor.sort(reverse=True)
    q = torch.quantile(allv, torch.tensor([0.5, 0.95, 0.99], dtype=torch.float64))
    return dict(n=allv.numel(), q50=q[0].item(), q95=q[1].item(), q99=q[2].item(), worst_tensor_q95=per_tensor[0], top=per_tensor[:4])


def _grad_and_update_errors(fx, names, params, opt):
    """Element-wise comparison of the parameter gradients (fixture `gslice.*`: 64 strided elements per tensor, from the reference's
    autograd) and of one clipped Adam step (`dslice.*`), plus the norms.  Returns the worst relative errors (max |a-b| / max |b|)."""
    worst = dict(gslice=(0.0, ""), gnorm=(0.0, ""), dslice=(0.0, ""), dnorm=(0.0, ""))
    per_tensor = []
    for k, p in zip(names, params):
        ref = tt(fx["gslice." + k])
        if float(fx["gnorm." + k]) < 1e-7:
            continue                                   # exactly-zero gradients (k_proj.bias: softmax is shift-invariant)
        e = rel_err(_slices(p.grad, 64), ref)
        per_tensor.append((e, k))
        if e > worst["gslice"][0]:
            worst["gslice"] = (e, k)
        e = abs(p.grad.double().norm().item() - float(fx["gnorm." + k])) / float(fx["gnorm." + k])
        if e > worst["gnorm"][0]:
            worst["gnorm"] = (e, k)
    before = opt.flat_p.clone()
    opt.step()
    for k, p, o, s in zip(names, params, opt.offs[:-1], opt.sizes):
        if float(fx["gnorm." + k]) < 1e-7:
            continue                                   # Adam normalises pure round-off there
        delta = opt.flat_p[int(o):int(o) + s] - before[int(o):int(o) + s]
        ref, gref = tt(fx["dslice." + k]), tt(fx["gslice." + k])
        # Adam's first step is lr * g / (|g| + eps): ~ lr * sign(g) where the clipped gradient is far above eps = 1e-8, but LINEAR in g
        # where it is not (A_log: |g| ~ 1e-8), and there it only repeats the gradient comparison above with that element's own
        # relative error.  Compare the update where the reference gradient is neither round-off nor in the linear regime.
        coef = min(1.0, 1.0 / (float(fx["gnorm." + k]) + 1e-6))
        m = (gref.abs() > 0.02 * gref.abs().max()) & (gref.abs() * coef > 1e-6)
        if m.any():
            e = ((_slices(delta, 64).double().cpu() - ref)[m].abs().max() / ref.abs().max()).item()
            if e > worst["dslice"][0]:
                worst["dslice"] = (e, k)
        e = abs(delta.double().norm().item() - float(fx["dnorm." + k])) / max(float(fx["dnorm." + k]), 1e-12)
        if e > worst["dnorm"][0]:
            worst["dnorm"] = (e, k)
    per_tensor.sort(reverse=True)
    worst["top"] = per_tensor[:6]
    worst["median"] = per_tensor[len(per_tensor) // 2][0]
    return worst


def _check_step_fixture(fx, gen_kw, vol, dim, depth, heads, seed, tol_fwd, tol_grad, tol_sens, tag, models=None, batch=2, tol_gen=1e-2, tol_q95=2e-3, tol_tq95=1e-2):
    from gfe_hip import det_init as det
    from gfe_hip.step import ClassifyStep, build_models
    gen, head, ft = models if models is not None else build_models(vol=vol, dim=dim, depth=depth, heads=heads, seed=seed, **gen_kw)
    x, x_cat, x_num, y = det.det_inputs(batch, vol, seed=seed)
    st = ClassifyStep(gen, head, ft)
    head.eval(); ft.eval()                          # fixtures were generated with dropout off
    st.opt.zero_grad()
    pred, (mi, mo, pet) = st.forward(x.to(DEV), x_cat.to(DEV), x_num.to(DEV))
    meas = {}
    for name, t in (("mid_input", mi), ("mid_output", mo), ("pet", pet)):
        ref_abs = float(fx[name + "_abssum"])
        assert abs(t.double().abs().sum().item() - ref_abs) / ref_abs < 1e-2, name
        meas[name] = rel_err(_slices(t.contiguous()), tt(fx[name + "_slice"]))
        # frozen generator: bf16 activations through 12 conv layers + the ViT.  tests/test_ladder_gpu.py attributes the budget: every stage
        # adds 4-6e-3 (a bf16 output rounding alone is up to 3.9e-3 of the maximum), 6-8e-3 at the end of the full-width chain
        assert meas[name] < tol_gen, (name, meas[name])
    meas["pred"] = rel_err(pred, tt(fx["pred"]))
    loss = F.binary_cross_entropy(torch.sigmoid(pred.squeeze(1)), y.to(DEV).float())
    meas["loss"] = abs(loss.item() - float(fx["loss"])) / max(1.0, float(fx["loss"]))
    loss.backward()
    names = ["head." + k for k, _ in head.named_parameters()] + ["ft." + k for k, _ in ft.named_parameters()]
    gstat = _grad_sample_stats(fx, names, st.all_params)      # (before the optimizer step: it reads the gradients)
    worst = _grad_and_update_errors(fx, names, st.all_params, st.opt)
    print("%s gradient elements, %d sampled over all tensors (error / largest element of the tensor): median %.2e, 95th percentile %.2e, 99th %.2e; "
          "largest per-tensor 95th percentile %.2e (%s); next %s"
          % (tag, gstat["n"], gstat["q50"], gstat["q95"], gstat["q99"], *gstat["worst_tensor_q95"], [("%.1e" % e, k) for e, k in gstat["top"][1:]]))
    print("%s measured rel errors vs the reference (fp32 CPU): %s | worst gradient element %.2e (%s), gradient norm %.2e (%s), "
          "Adam update element %.2e (%s), update norm %.2e (%s); gradient elements per tensor: median %.2e, six worst %s"
          % (tag, {k: "%.2e" % v for k, v in meas.items()}, *worst["gslice"], *worst["gnorm"], *worst["dslice"], *worst["dnorm"], worst["median"],
             [("%.1e" % e, k) for e, k in worst["top"]]))
    assert meas["pred"] < tol_fwd and meas["loss"] < tol_fwd, meas
    # Element-wise (64 strided elements per tensor, error / max |reference element| of the tensor).  The head itself is fp32-exact
    # (test_head_alone_...: 5e-3); what is measured here is the frozen generator's bf16 error (pet / mid features ~1e-2, run-to-run
    # different roundings from its atomically accumulated GroupNorm fold) carried through the head's backward: typical tensor 2-3e-3,
    # tensors that are signed sums with heavy cancellation (the 4-element bias of the image-token Linear = 1024 feature gradients each,
    # the 32-wide dt_proj, a feed-forward row) up to 6e-2 of their largest element, and WHICH tensor is worst changes from run to run.
    # What is ASSERTED since round 6 are the statistics with a variance: the pooled 95th percentile and the per-tensor 95th percentile
    # of the sampled elements, and the norms.  The worst single element of the 64-element slices is still printed above (it moved between
    # 3.4e-2 and 5.2e-2 with nothing but the number of K ranges of one generator product, profiles/r05/t2_realisations.txt) and only guarded
    # by a loose bound against gross errors.
    # measured (round 6): pooled 95th percentile 8.4e-4 (T2) / 1.3e-3 (T1), 99th 2.0e-3 / 3.6e-3 -- the bf16 tolerance of BASELINE.json
    # is 1e-2 -- largest per-tensor 95th percentile 4.6e-3 / 5.8e-3
    assert gstat["q95"] < tol_q95 and gstat["q99"] < 3 * tol_q95 and gstat["worst_tensor_q95"][0] < tol_tq95, gstat
    assert worst["median"] < tol_grad / 2 and worst["gnorm"][0] < tol_grad, worst
    for e, k in worst["top"]:
        assert e < 2 * tol_sens, (k, e)
    assert worst["dslice"][0] < 2 * tol_sens and worst["dnorm"][0] < tol_grad, worst


def test_reduced_step_vs_reference_fixture():
    """T1: reduced-width classify_mamba step (32^3) run by the reference on CPU vs the HIP path end to end: logits, loss, every
    parameter gradient ELEMENT-wise (fixture gslice.*), one clipped Adam step element-wise (dslice.*).  bf16 generator -> 2e-2."""
    fx = golden("t1_reduced_step.npz")
    _check_step_fixture(fx, dict(f_maps=(8, 16, 32), vit_kwargs=dict(dim=64, depth=2, heads=2, dim_head=16, mlp_dim=128)),
                        (32, 32, 32), 64, 2, 8, 11, tol_fwd=1e-2, tol_grad=2e-2, tol_sens=5e-2, tag="T1 (reduced, 32^3)",
                        tol_gen=2e-2, tol_q95=3e-3, tol_tq95=1.2e-2)        # 8 / 16 / 32 channels: fewer terms per sum to average the roundings out (measured 1.2e-2)


def test_full_96_step_vs_reference_fixture():
    """T2 / BASELINE config 1: the full-size model on 2 volumes of 96^3 (reference run on CPU in the build container)."""
    fx = golden("t2_full96_step.npz")
    _check_step_fixture(fx, dict(f_maps=(64, 128, 256)), (96, 96, 96), 512, 6, 8, 21, tol_fwd=2e-3, tol_grad=1.2e-2, tol_sens=4e-2, tag="T2 (config 1, 96^3)")
    # round 5 (K / V projections folded away, the condition in exact f32): pred 1.1e-3 -> 4.5e-4, worst gradient element 5.2e-2 -> 3.4e-2,
    # worst norm 1.1e-2 -> 7.0e-3 (bounds were 5e-3 / 6e-2 / 2e-2); what is left is the frozen generator's bf16 chain (tests/test_ladder_gpu.py)


def test_native_160x160x96_step_with_default_constructors_vs_reference_fixture():
    """T7: the reference's OWN geometry (config/classify_mamba_config.yaml:5-7), the three modules built with exactly the constructor
    calls of classify_mamba.py:36-56 -- no vol_size / in_features / d_cross: image (320,120) patch 40 (model.py:107-117), Linear(38 400, 4)
    (classifier.py:327), d_cross 25 600 (mamba_transformer.py:84) -- the only geometry the authors' model.pt loads at.  20x20x12 conv tile
    grids, LayerNorm(409 600), K = 409 600 skinny GEMMs, 25 600-wide K/V projections; one sample, reference run on CPU (tools/make_golden.py t7)."""
    from classify.classifier import Combine_classfier_vit_mid
    from cross_atten.mamba_transformer import Cross_mamba_both
    from gfe_hip import det_init as det
    from pytorch3dunet.unet3d.model import Residual_mid_UNet3D_vit
    fx = golden("t7_native_step.npz")
    gen = Residual_mid_UNet3D_vit(1, 1, is_segmentation=False, f_maps=(64, 128, 256))
    head = Combine_classfier_vit_mid(seq_length=4)
    ft = Cross_mamba_both(categories=(11, 2, 2, 4, 4, 3, 3), num_continuous=25, dim=512, dim_out=1, depth=6, heads=8, attn_dropout=0.1,
                          ff_dropout=0.1, dim_head=512 // 8)
    for m, pre in ((gen, "gen."), (head, "head."), (ft, "ft.")):
        m.load_state_dict(det.det_state_dict(m.state_dict(), seed=71, prefix=pre))
    models = (gen.to(DEV).eval(), head.to(DEV), ft.to(DEV))
    _check_step_fixture(fx, None, (160, 160, 96), 512, 6, 8, 71, tol_fwd=3e-3, tol_grad=1e-2, tol_sens=2.5e-2, tag="T7 (native 160x160x96)",
                        models=models, batch=1)


def test_head_alone_on_the_references_generator_outputs_meets_fp32_tolerance():
    """The trainable path in isolation: head + Cross_mamba_both fed with the REFERENCE's own generator outputs (T1 fixture holds them
    in full), so that nothing of the bf16 generator is in the comparison.  The head's Linears run on the exact-f32 MFMA GEMM and the
    cross-attention reads the f32 condition in place; what is left is the bf16 rounding of the mid features on the way into the image-token Linear."""
    from gfe_hip import det_init as det
    from gfe_hip.step import build_models
    from gfe_hip.train_ops import Condition, FlatAdam
    fx = golden("t1_reduced_step.npz")
    vol = (32, 32, 32)
    _, head, ft = build_models(vol=vol, f_maps=(8, 16, 32), dim=64, depth=2, heads=8, seed=11,
                               vit_kwargs=dict(dim=64, depth=2, heads=2, dim_head=16, mlp_dim=128))
    head.eval(); ft.eval()
    x, x_cat, x_num, y = det.det_inputs(2, vol, seed=11)
    params = list(head.parameters()) + list(ft.parameters())
    opt = FlatAdam(params, lr=1e-4, max_norm=1.0)
    opt.zero_grad()
    mi, mo, pet = (tt(fx[k], device=DEV) for k in ("mid_input", "mid_output", "pet"))
    feat = head(mi, mo)
    pred = ft(x_cat.to(DEV), x_num.to(DEV), feat, Condition([x.to(DEV), pet]))
    loss = F.binary_cross_entropy(torch.sigmoid(pred.squeeze(1)), y.to(DEV).float())
    loss.backward()
    names = ["head." + k for k, _ in head.named_parameters()] + ["ft." + k for k, _ in ft.named_parameters()]
    e_feat, e_pred = rel_err(feat, tt(fx["feat"])), rel_err(pred, tt(fx["pred"]))
    e_loss = abs(loss.item() - float(fx["loss"])) / max(1.0, float(fx["loss"]))
    worst = _grad_and_update_errors(fx, names, params, opt)
    print("head alone (reference generator outputs in): feat %.2e pred %.2e loss %.2e | worst gradient element %.2e (%s), norm %.2e (%s), "
          "Adam update element %.2e (%s)" % (e_feat, e_pred, e_loss, *worst["gslice"], *worst["gnorm"], *worst["dslice"]))
    # feat: a 512-term signed sum of bf16-rounded mid features (2^-9 each): 3.5e-3 of its maximum; everything behind it is f32 -- since
    # round 5 also the cross-attention over the image condition (folded K / V projections: pred 9e-4 -> 1.7e-4, loss 6e-5 -> 1.2e-5)
    assert e_feat < 6e-3 and e_pred < 1e-3 and e_loss < 1e-4
    assert worst["gslice"][0] < 8e-3 and worst["gnorm"][0] < 5e-3, worst


@pytest.mark.gpu
def test_residual_block_as_one_node_equals_norm_block_add(monkeypatch):
    """ResidualBlock.forward as ONE autograd node (RMSNorm with the residual copy, the fused Mamba block, out_proj accumulating into the
    copy; backward: the norm kernel adds the residual branch's gradient) against the three-node form mixer(norm(x)) + x: same forward bits
    and the same gradients for the stream and every parameter (the sums are the same sums: only two add launches per layer are gone)."""
    from cross_atten.mamba import Mamba, MambaConfig
    torch.manual_seed(11)
    m = Mamba(MambaConfig(d_model=64, n_layers=3, use_cuda=True)).to(DEV)
    x = torch.randn(4, 37, 64, device=DEV)
    outs = []
    for split in ("1", "0"):
        monkeypatch.setenv("GFE_MAMBA_SPLIT_RESIDUAL", split)
        xi = x.clone().requires_grad_(True)
        for p_ in m.parameters():
            p_.grad = None
        y = m(xi)
        (y * torch.linspace(-1, 1, y.numel(), device=DEV).view_as(y)).sum().backward()
        outs.append((y.detach().clone(), xi.grad.clone(), {n: p_.grad.clone() for n, p_ in m.named_parameters()}))
    (y0, dx0, g0), (y1, dx1, g1) = outs
    assert torch.equal(y0, y1)
    assert torch.equal(dx0, dx1)
    for n in g0:
        assert torch.equal(g0[n], g1[n]), n


@pytest.mark.gpu
@pytest.mark.parametrize("M,N,K", [(296, 512, 512), (37, 2048, 512), (300, 64, 1024), (8, 1, 512), (37, 40, 25), (5, 8, 8), (2048, 512, 296),
                                   (296, 2048, 512), (296, 32, 1024), (296, 1024, 32), (296, 1024, 64), (8, 4096, 512), (19, 36, 48), (296, 512, 2048), (40, 24, 16), (500, 96, 4096)])
def test_gemm_f32_exact_modes(M, N, K):
    """gfe_gemm_f32 (f32 MFMA): every operand layout, bias, accumulation and split-K against an f64 matmul: f32-exact (<= 2e-6 of the
    largest |sum|), any sizes / alignments (the 1-wide logit layer, the 25/37-wide test models)."""
    from gfe_hip import nn_ops as K_, call, ptr, stream
    g = torch.Generator(device="cpu").manual_seed(M * 7 + N)
    a, b, bias = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g), torch.randn(N, generator=g)
    ref = a.double() @ b.double().t()
    scale = (a.double().abs() @ b.double().abs().t()).max().item()
    for a_t in (False, True):
        for b_t in (False, True):
            aa = (a.t().contiguous() if a_t else a).to(DEV)
            bb = (b.t().contiguous() if b_t else b).to(DEV)
            out = K_.gemm_f32(aa, a_t, bb, b_t, bias=bias.to(DEV))
            assert (out.double().cpu() - ref - bias.double()).abs().max().item() <= 2e-6 * scale, (a_t, b_t)
            acc = torch.full((M, N), 0.5, device=DEV)
            K_.gemm_f32(aa, a_t, bb, b_t, accum_into=acc)
            assert (acc.double().cpu() - ref - 0.5).abs().max().item() <= 2e-6 * scale, (a_t, b_t, "accumulate")
            if not a_t:
                # a column range of a wider matrix as the target (mamba_block writes d delta_r into [d delta_r | dB | dC] in place), twice: the
                # in-block K split (16-byte aligned operands) and, from an operand shifted by one float, the staged kernel -- and bit-repeatable
                wide = torch.full((M, N + 7), -3.0, device=DEV)
                call("gfe_gemm_f32", ptr(aa), aa.stride(0), 0, ptr(bb), bb.stride(0), int(b_t), ptr(wide) + 4 * 3, N + 7, M, N, K, None, 0, 1, None, stream())
                assert (wide[:, 3:3 + N].double().cpu() - ref).abs().max().item() <= 2e-6 * scale and (wide[:, :3] == -3).all() and (wide[:, 3 + N:] == -3).all()
                again = K_.gemm_f32(aa, a_t, bb, b_t, bias=bias.to(DEV))
                assert torch.equal(out, again)
                if K > 1:
                    pad = torch.zeros(aa.numel() + 1, device=DEV)
                    pad[1:] = aa.reshape(-1)
                    shifted = pad[1:].view(aa.shape)
                    out2 = K_.gemm_f32(shifted, a_t, bb, b_t, bias=bias.to(DEV))
                    assert (out2.double().cpu() - ref - bias.double()).abs().max().item() <= 2e-6 * scale, (a_t, b_t, "unaligned")


@pytest.mark.gpu
@pytest.mark.parametrize("M,N,K", [(296, 512, 512), (37, 2048, 512), (1536, 512, 1152), (300, 64, 1024), (5, 8, 8)])
def test_gemm_ex_modes(M, N, K):
    """gfe_gemm_ex: every operand mode (bf16|f32, K-major|reduction-major) against an f32 matmul of the bf16-rounded operands."""
    from gfe_hip import nn_ops as K_
    g = torch.Generator(device="cpu").manual_seed(M * 7 + N)
    a = torch.randn(M, K, generator=g).cuda()
    b = torch.randn(N, K, generator=g).cuda()
    ref = a.bfloat16().float() @ b.bfloat16().float().t()
    for a_f32 in (False, True):
        for b_f32 in (False, True):
            for a_t in (False, True):
                for b_t in (False, True):
                    if K % 8 and not (a_t and b_t):
                        continue
                    if (a_t and M % (4 if a_f32 else 8)) or (b_t and N % (4 if b_f32 else 8)):
                        continue                 # row stride of a reduction-major operand must keep 16-byte alignment
                    aa = a if a_f32 else a.bfloat16()
                    bb = b if b_f32 else b.bfloat16()
                    aa = aa.t().contiguous() if a_t else aa
                    bb = bb.t().contiguous() if b_t else bb
                    out = K_.gemm_ex(aa, a_t, bb, b_t)
                    err = (out - ref).abs().max().item()
                    assert err <= 2e-3 * max(1.0, ref.abs().max().item()), (a_f32, b_f32, a_t, b_t, err)


_GRAPHED_STEP_CHILD = r"""
import sys, torch
sys.path[:0] = [{root!r}, {src!r}, {tests!r}]
from gfe_hip.step import ClassifyStep, build_models
import gfe_hip.det_init as det
kw = dict(vol=(32, 32, 32), f_maps=(8, 16, 32), dim=64, depth=2, heads=8, vit_kwargs=dict(dim=64, depth=2, heads=2, dim_head=16, mlp_dim=128), seed=3)
x, x_cat, x_num, y = [t.cuda() for t in det.det_inputs(2, (32, 32, 32), seed=4)]
outs = []
for graphed in (False, True):
    gen, head, ft = build_models(**kw)
    for m in ft.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    st = ClassifyStep(gen, head, ft)
    fn = st.train_step_graphed if graphed else st.train_step
    losses = [float(fn(x, x_cat, x_num, y)) for _ in range(3)]
    outs.append((losses, st.opt.flat_p.clone()))
(l0, p0), (l1, p1) = outs
assert max(abs(a - b) for a, b in zip(l0, l1)) < 1e-4, (l0, l1)
assert (p0 - p1).abs().max().item() < 2e-4           # 3 Adam steps of 1e-4 each: identical up to atomics-order noise
print("GRAPHED_STEP_OK")
# full size (96^3, 64/128/256 channels, B = 1): the captured generator must keep reproducing the eager outputs BIT FOR BIT (the
# memset-node bug returned garbage from the second replay on; the frozen generator has no atomics left: fixed-order bias-table fold
# and split-K sums)
gen, head, ft = build_models()
xf = det.det_inputs(1, (96, 96, 96), seed=5)[0].cuda()
with torch.no_grad():
    ref = [t.clone() for t in gen(xf, output_vit_mid=True)]
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        gen(xf, output_vit_mid=True)
    torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        res = gen(xf, output_vit_mid=True)
    for i in range(4):
        g.replay(); torch.cuda.synchronize()
        for name, a, b in zip(("mid_input", "mid_output", "pet"), res, ref):
            assert torch.equal(a, b), (i, name, int((a != b).sum()))
print("GRAPHED_FULL_OK")
"""


@pytest.mark.gpu
def test_graphed_step_matches_eager_step():
    """ClassifyStep.train_step_graphed (HIP-graph replay of zero_grad + forward + backward) updates the parameters exactly like
    train_step on the same inputs (dropout off so that both are deterministic up to the kernels' f32 atomics), and at FULL size the
    replayed generator keeps reproducing the eager one replay after replay.
    Round 1 saw whole-step replay abort with HSA_STATUS_ERROR_EXCEPTION in ~2 of 10 processes.  Root cause (round 2,
    tools/graph_gen_bisect.py, profiles/r02/graph_fault_sweeps.txt): `hipMemsetAsync` inside the C-ABI became memset NODES, which this
    ROCm's graph replay does not keep ordered behind the preceding kernel nodes; where the target block of the graph's private pool had
    an earlier tenant, the memset clobbered live data, the generator returned garbage from the second replay on and a downstream
    kernel died on it.  The library now zero-fills with a kernel (common.h: gfe_zero_async): 0 of 60 processes fail.  Runs in a child
    process so that a regression cannot take the session down -- but it FAILS the test."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = _GRAPHED_STEP_CHILD.format(root=root, src=os.path.join(root, "gfe-mamba_amd"), tests=os.path.join(root, "tests"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "GRAPHED_STEP_OK" in r.stdout and "GRAPHED_FULL_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])


@pytest.mark.gpu
def test_overlapped_update_matches_inline_update():
    """ClassifyStep(overlap_update=True) (optional: all-reduce + clip + Adam on a side stream underneath the next
    step's frozen-generator forward) produces the same losses and parameters as the inline update."""
    from gfe_hip.step import ClassifyStep, build_models
    import gfe_hip.det_init as det
    kw = dict(vol=(32, 32, 32), f_maps=(8, 16, 32), dim=64, depth=2, heads=8, vit_kwargs=dict(dim=64, depth=2, heads=2, dim_head=16, mlp_dim=128), seed=3)
    x, x_cat, x_num, y = [t.cuda() for t in det.det_inputs(2, (32, 32, 32), seed=4)]
    outs = []
    for overlap in (False, True):
        gen, head, ft = build_models(**kw)
        for m in ft.modules():
            if isinstance(m, torch.nn.Dropout):
                m.p = 0.0
        st = ClassifyStep(gen, head, ft, overlap_update=overlap)
        losses = [fn for fn in (st.train_step(x, x_cat, x_num, y) for _ in range(5))]     # no host sync between steps
        st.opt.wait_updated()
        outs.append(([float(l) for l in losses], st.opt.flat_p.clone(), float(st.eval_step(x, x_cat, x_num).sum())))
    (l0, p0, e0), (l1, p1, e1) = outs
    assert max(abs(a - b) for a, b in zip(l0, l1)) < 1e-4, (l0, l1)
    assert (p0 - p1).abs().max().item() < 3e-4
    assert abs(e0 - e1) < 1e-3


def test_cross_mamba_ablation_vs_reference_fixture():
    """Cross_mamba_ablation (cross_atten/mamba_transformer.py:254-385): logits, loss and per-parameter gradient norms of the four
    forward variants against the reference's own output (tests/golden/t3_ablation.npz)."""
    from test_oracle_golden import ABL_CASES, ablation_setup
    fx = golden("t3_ablation.npz")
    ft, depth, heads, x, x_cat, x_num, y, pet, feat = ablation_setup(fx, DEV)
    ft = ft.to(DEV).eval()
    for name, c in ABL_CASES.items():
        ft.zero_grad(set_to_none=True)
        pred = ft(x_cat, x_num, feat if c["feat"] else None, [x, pet] if c["cond"] else None, no_table=c["no_table"])
        loss = F.binary_cross_entropy(torch.sigmoid(pred.squeeze(1)), y.float())
        loss.backward()
        assert rel_err(pred, tt(fx[name + ".pred"])) < 2e-2, name
        assert abs(loss.item() - float(fx[name + ".loss"])) < 1e-2, name
        # the bias gradients behind the logit are sums of (sigmoid - y) terms of size 0.3 that cancel to 1e-3: absolute floor
        atol = 1e-3 * max(float(fx[name + ".gnorm." + k]) for k, _ in ft.named_parameters())
        for k, p in ft.named_parameters():
            ref = float(fx[name + ".gnorm." + k])
            if ref < 0:                                   # the reference left this parameter without a gradient
                assert p.grad is None or float(p.grad.norm()) == 0.0, (name, k)
            else:
                got = float(p.grad.double().norm())
                assert abs(got - ref) <= 1e-1 * ref + atol, (name, k, got, ref)     # norms of bf16-operand gradients, as in T1 / T2


@pytest.mark.gpu
def test_pipelined_step_matches_serial_step():
    """ClassifyStep.train_step_pipelined (head of batch k on a second stream under the frozen generator's forward for batch k+1)
    over four DIFFERENT batches: same losses and parameters as the serial train_step -- in particular every head consumes the
    generator outputs of its own batch -- with and without announcing the next batch."""
    from gfe_hip.step import ClassifyStep, build_models
    import gfe_hip.det_init as det
    kw = dict(vol=(32, 32, 32), f_maps=(8, 16, 32), dim=64, depth=2, heads=8, vit_kwargs=dict(dim=64, depth=2, heads=2, dim_head=16, mlp_dim=128), seed=3)
    batches = [[t.cuda() for t in det.det_inputs(2, (32, 32, 32), seed=50 + i)] for i in range(4)]
    outs = []
    # "pipelined_cus": head / side streams on 16 CUs of their own (two per XCD), the generator's on the other 240 (gfe_stream_create_cu_range)
    for mode in ("serial", "pipelined", "pipelined_unannounced", "pipelined_cus"):
        gen, head, ft = build_models(**kw)
        for m in ft.modules():
            if isinstance(m, torch.nn.Dropout):
                m.p = 0.0
        st = ClassifyStep(gen, head, ft, head_cus=16 if mode == "pipelined_cus" else 0)
        losses = []
        for i, b in enumerate(batches):
            if mode == "serial":
                losses.append(st.train_step(*b))
            else:
                nxt = batches[i + 1][0] if (mode != "pipelined_unannounced" and i + 1 < len(batches)) else None
                losses.append(st.train_step_pipelined(*b, x_next=nxt))
        st.join()
        torch.cuda.synchronize()
        outs.append(([float(l) for l in losses], st.opt.flat_p.clone(), float(st.eval_step(*batches[0][:3]).sum())))
    (l0, p0, e0) = outs[0]
    assert len(set(round(v, 4) for v in l0)) > 1                     # the batches really differ
    for l1, p1, e1 in outs[1:]:
        assert max(abs(a - b) for a, b in zip(l0, l1)) < 1e-4, (l0, l1)
        assert (p0 - p1).abs().max().item() < 4e-4
        assert abs(e0 - e1) < 1e-3


def test_head_small_ops_vs_torch_and_reference_fixture():
    """gfe_embed_tokens / mean_tokens / cross_attn_q1 / layernorm_rows / geglu / bce_sigmoid (include/gfe_hip.h): forward and every
    gradient against the same arithmetic in torch fp64 autograd, and the CrossAttention / FeedForward fixtures of the reference for the
    one-query geometry (the op-level fixture test above goes through them with 6 keys x 3 queries)."""
    from gfe_hip import head_ops as Hd
    g = torch.Generator().manual_seed(5)
    r = lambda *s: torch.randn(*s, generator=g)
    dd = lambda t: t.detach().double().cpu().requires_grad_(True)
    # --- embed_tokens: offsets of the synthetic table (11,2,2,4,4,3,3 with 2 special tokens -> [2,13,15,17,21,25,28])
    cards = (11, 2, 2, 4, 4, 3, 3)
    B, dim, ncont, nf = 3, 40, 5, 4
    off = torch.tensor([2, 13, 15, 17, 21, 25, 28])
    x_cat = torch.stack([torch.randint(0, c, (B,), generator=g) for c in cards], 1)
    x_cat[1, 0] = x_cat[0, 0]                                             # a shared embedding row: gradients must add
    emb, x_num, nw, nb, cls, feat = r(sum(cards) + 2, dim), r(B, ncont), r(ncont, dim), r(ncont, dim), r(1, 1, dim), r(B, nf, dim)
    w = r(B, 1 + len(cards) + ncont + nf, dim)
    gp = [t.to(DEV).requires_grad_(True) for t in (emb, nw, nb, cls, feat)]
    out = Hd.embed_tokens(x_cat.to(DEV), off.to(DEV), gp[0], x_num.to(DEV), gp[1], gp[2], gp[3], gp[4])
    cp = [dd(t) for t in (emb, nw, nb, cls, feat)]
    ref = torch.cat([cp[3].expand(B, -1, -1), cp[0][x_cat + off], x_num.double().unsqueeze(-1) * cp[1] + cp[2], cp[4]], 1)
    assert torch.equal(out.cpu().double()[:, :8], ref.detach()[:, :8])                      # gather and cls rows: exact copies
    assert rel_err(out, ref) < 1e-6
    (out * w.to(DEV)).sum().backward(); (ref * w.double()).sum().backward()
    for a, b in zip(gp, cp):
        assert rel_err(a.grad, b.grad) < 1e-5
    # without the table / without image tokens (Cross_mamba_ablation's switches)
    o2 = Hd.embed_tokens(None, None, None, None, None, None, gp[3], gp[4])
    assert torch.equal(o2.cpu(), torch.cat([cls.expand(B, -1, -1), feat], 1))
    # --- mean over tokens
    x = r(3, 37, 64)
    xg, xc = x.to(DEV).requires_grad_(True), dd(x)
    m, mr = Hd.mean_tokens(xg), xc.mean(1, keepdim=True)
    assert rel_err(m, mr) < 1e-6
    m.sum().backward(); mr.sum().backward()
    assert rel_err(xg.grad, xc.grad) < 1e-6
    # --- one-query cross attention, 8 heads x 64, 192 keys (the real geometry) and a ragged one
    for (B, H, nk, dh) in ((2, 8, 192, 64), (3, 2, 7, 8)):
        q, k, v, w = r(B, 1, H * dh), r(B, nk, H * dh), r(B, nk, H * dh), r(B, 1, H * dh)
        gq, gk, gv = (t.to(DEV).requires_grad_(True) for t in (q, k, v))
        cq, ck, cv = dd(q), dd(k), dd(v)
        o = Hd.cross_attn_q1(gq, gk, gv, H)
        sp = lambda t: t.view(B, -1, H, dh).transpose(1, 2)
        ro = (torch.softmax(sp(cq) @ sp(ck).transpose(-1, -2) / dh ** 0.5, -1) @ sp(cv)).transpose(1, 2).reshape(B, 1, H * dh)   # sd_cross_atten.py:58-68
        assert rel_err(o, ro) < 1e-5
        (o * w.to(DEV)).sum().backward(); (ro * w.double()).sum().backward()
        for a, b in ((gq, cq), (gk, ck), (gv, cv)):
            assert rel_err(a.grad, b.grad) < 1e-5
    # --- LayerNorm over rows
    x, ga, be, w = r(5, 512) * 3 + 1, r(512), r(512), r(5, 512)
    gx, gg, gb = (t.to(DEV).requires_grad_(True) for t in (x, ga, be))
    cx, cg, cb = dd(x), dd(ga), dd(be)
    y, yr = Hd.layernorm_rows(gx, gg, gb), F.layer_norm(cx, (512,), cg, cb)
    assert rel_err(y, yr) < 1e-5
    (y * w.to(DEV)).sum().backward(); (yr * w.double()).sum().backward()
    for a, b in ((gx, cx), (gg, cg), (gb, cb)):
        assert rel_err(a.grad, b.grad) < 1e-5
    # --- the same with MANY rows (the 3-D ViT's token rows): per-block partial rows + fixed-order reduction instead of atomics, ragged
    # widths (dim % 256 != 0, one to eight float4 per lane), row counts off the 8-row block grid, accumulation into existing gradients; bitwise repeatable
    for rows, dim in ((300, 128), (1027, 512), (259, 516), (4100, 2048), (256, 4)):
        x, ga, be, w = r(rows, dim) * 3 + 1, r(dim), r(dim), r(rows, dim)
        gx, gg, gb = (t.to(DEV).requires_grad_(True) for t in (x, ga, be))
        cx, cg, cb = dd(x), dd(ga), dd(be)
        (Hd.layernorm_rows(gx, gg, gb) * w.to(DEV)).sum().backward(); (F.layer_norm(cx, (dim,), cg, cb) * w.double()).sum().backward()
        for a, b in ((gx, cx), (gg, cg), (gb, cb)):
            assert rel_err(a.grad, b.grad) < 1e-5, (rows, dim)
        g1 = [t.grad.clone() for t in (gx, gg, gb)]
        (Hd.layernorm_rows(gx, gg, gb) * w.to(DEV)).sum().backward()                  # accumulates: exactly twice, in the same order
        for a, b in zip((gx, gg, gb), g1):
            assert torch.equal(a.grad, 2 * b), (rows, dim)
    # --- GEGLU (no dropout: exact), then the dropout statistics and mask consistency between forward and backward
    x, w = r(6, 2048), r(6, 1024)
    gx, cx = x.to(DEV).requires_grad_(True), dd(x)
    y = Hd.geglu_dropout(gx)
    a_, g_ = cx.chunk(2, -1)
    yr = a_ * F.gelu(g_)
    assert rel_err(y, yr) < 1e-6
    (y * w.to(DEV)).sum().backward(); (yr * w.double()).sum().backward()
    assert rel_err(gx.grad, cx.grad) < 1e-5
    gx2 = x.to(DEV).requires_grad_(True)
    yd = Hd.geglu_dropout(gx2, 0.1, training=True)
    kept = (yd != 0).float().mean().item()
    assert 0.86 < kept < 0.94                                                         # Dropout(0.1)
    assert rel_err(yd[yd != 0], (y.detach() / 0.9)[yd != 0]) < 1e-6                   # survivors scaled by 1 / (1 - p)
    yd.sum().backward()
    assert torch.equal(gx2.grad[:, :1024] == 0, yd == 0)                              # the backward regenerates the same mask
    yd2 = Hd.geglu_dropout(x.to(DEV), 0.1, training=True)
    assert not torch.equal(yd2 == 0, yd == 0)                                         # a new mask on every call
    # --- BCE(sigmoid): value and gradient, saturated logits included (torch clamps the logs at -100)
    z = torch.tensor([0.3, -2.0, 5.0, -30.0, 30.0, 120.0, -120.0, 0.0])
    t = torch.tensor([1.0, 0.0, 1.0, 1.0, 0.0, 0.0, 1.0, 1.0])
    gz = z.to(DEV).requires_grad_(True)
    cz = z.clone().requires_grad_(True)
    loss, lref = Hd.bce_sigmoid(gz, t.to(DEV)), F.binary_cross_entropy(torch.sigmoid(cz), t)
    assert abs(loss.item() - lref.item()) < 1e-5 * max(1.0, lref.item())
    loss.backward(); lref.backward()
    assert (gz.grad.cpu() - cz.grad).abs().max() < 1e-6


@pytest.mark.gpu
def test_dropout_masks_change_between_graph_replays():
    """GEGLU + dropout under a HIP graph: the host-drawn seed is baked into the node, so the mask varies through the device step counter
    (head_ops.set_dropout_step_counter) that the captured step increments -- two replays drop different units, forward and backward of
    one replay use the same mask, and without the counter the replays would be identical."""
    from gfe_hip import head_ops as Hd
    x = torch.randn(64, 512, device=DEV, requires_grad=True)
    ctr = torch.zeros(1, dtype=torch.int64, device=DEV)

    def run():
        ctr.add_(1)
        x.grad = None
        y = Hd.geglu_dropout(x, 0.3, training=True)
        y.sum().backward()
        return y.detach(), x.grad

    Hd.set_dropout_step_counter(ctr)
    try:
        side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            run()
        torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            y, gx = run()
    finally:
        Hd.set_dropout_step_counter(None)
    g.replay(); torch.cuda.synchronize()
    y1, g1 = y.clone(), gx.clone()
    g.replay(); torch.cuda.synchronize()
    y2, g2 = y.clone(), gx.clone()
    drop1, drop2 = y1 == 0, y2 == 0
    assert 0.2 < drop1.float().mean() < 0.4 and 0.2 < drop2.float().mean() < 0.4
    assert (drop1 != drop2).float().mean() > 0.2                                  # fresh mask per replay
    assert torch.equal(g1[:, :256] == 0, drop1) and torch.equal(g2[:, :256] == 0, drop2)       # backward of a replay uses its forward's mask
