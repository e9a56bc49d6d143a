"""GPU parity of the trainable path (Mamba, cross-attention, GEGLU FF, head Linear, clip+Adam, whole step) against the
reference-generated fixtures and the oracle.  GEMM operands are bf16 -> 1e-2-class tolerances (BASELINE.json north_star)."""
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
    assert rel_err(m.layers[0].norm(x), tt(fx["y_norm0"])) < 1e-5
    assert rel_err(m.layers[0].mixer(x), tt(fx["y_block0"])) < 2e-2
    y = m(x)
    assert rel_err(y, tt(fx["y"])) < 2e-2
    (y * tt(fx["w"], device=DEV)).sum().backward()
    assert rel_err(x.grad, tt(fx["gx"])) < 3e-2
    for k, p in m.named_parameters():
        assert rel_err(p.grad, tt(fx["g." + k])) < 4e-2, k


def test_cross_attention_ff_embedder_vs_reference_fixture():
    from cross_atten.corss_ft_transformer import FeedForward, NumericalEmbedder
    from cross_atten.sd_cross_atten import CrossAttention
    fx = golden("t0_head_ops.npz")
    ca = CrossAttention(n_heads=2, d_embed=16, d_cross=24)
    ca.load_state_dict(sub_sd(fx, "ca.sd."))
    ca = ca.to(DEV)
    x, y = tt(fx["ca.x"], device=DEV).requires_grad_(True), tt(fx["ca.y"], device=DEV).requires_grad_(True)
    o = ca(x, y)
    assert rel_err(o, tt(fx["ca.out"])) < 2e-2
    (o * tt(fx["ca.w"], device=DEV)).sum().backward()
    assert rel_err(x.grad, tt(fx["ca.gx"])) < 3e-2 and rel_err(y.grad, tt(fx["ca.gy"])) < 3e-2
    for k, p in ca.named_parameters():
        if k == "k_proj.bias":
            assert p.grad.abs().max() < 1e-3       # exactly zero in exact arithmetic
            continue
        assert rel_err(p.grad, tt(fx["ca.g." + k])) < 4e-2, k
    ff = FeedForward(16, mult=2, dropout=0.1)
    ff.load_state_dict(sub_sd(fx, "ff.sd."))
    ff = ff.to(DEV).eval()
    xf = tt(fx["ff.x"], device=DEV).requires_grad_(True)
    of = ff(xf)
    assert rel_err(of, tt(fx["ff.out"])) < 2e-2
    (of * tt(fx["ca.w"], device=DEV)).sum().backward()
    assert rel_err(xf.grad, tt(fx["ff.gx"])) < 3e-2
    ne = NumericalEmbedder(16, 5)
    ne.load_state_dict(sub_sd(fx, "ne.sd."))
    assert rel_err(ne.to(DEV)(tt(fx["ne.x"], device=DEV)), tt(fx["ne.out"])) < 1e-6


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


def _check_step_fixture(fx, gen_kw, vol, dim, depth, heads, seed, tol_fwd, tol_grad):
    from gfe_hip import det_init as det
    from gfe_hip.step import ClassifyStep, build_models
    gen, head, ft = build_models(vol=vol, dim=dim, depth=depth, heads=heads, seed=seed, **gen_kw)
    x, x_cat, x_num, y = det.det_inputs(2, vol, seed=seed)
    st = ClassifyStep(gen, head, ft)
    head.eval(); ft.eval()                          # fixtures were generated with dropout off
    st.opt.zero_grad()
    pred, (mi, mo, pet) = st.forward(x.to(DEV), x_cat.to(DEV), x_num.to(DEV))
    for name, t in (("mid_input", mi), ("mid_output", mo), ("pet", pet)):
        ref_sum, ref_abs = float(fx[name + "_sum"]), float(fx[name + "_abssum"])
        assert abs(t.double().abs().sum().item() - ref_abs) / ref_abs < tol_fwd, name
        f = t.contiguous().reshape(-1) if name == "pet" else t.contiguous().reshape(-1)
        step = max(1, f.numel() // 256)
        assert rel_err(f[::step][:256], tt(fx[name + "_slice"])) < 5 * tol_fwd, name
    assert rel_err(pred, tt(fx["pred"])) < 5 * tol_fwd
    loss = F.binary_cross_entropy(torch.sigmoid(pred.squeeze(1)), y.to(DEV).float())
    assert abs(loss.item() - float(fx["loss"])) < 5 * tol_fwd * max(1.0, float(fx["loss"]))
    loss.backward()
    names = ["head." + k for k, _ in head.named_parameters()] + ["ft." + k for k, _ in ft.named_parameters()]
    bad = []
    for k, p in zip(names, st.all_params):
        ref = float(fx["gnorm." + k])
        got = p.grad.double().norm().item()
        if ref < 1e-7:
            continue
        if abs(got - ref) / ref > tol_grad:
            bad.append((k, got, ref))
    assert not bad, bad
    before = st.opt.flat_p.clone()
    st.opt.step()
    for (k, p), o, s in zip(zip(names, st.all_params), st.opt.offs[:-1], st.opt.sizes):
        if k.endswith("k_proj.bias"):     # gradient is exactly zero in exact arithmetic; Adam normalises pure round-off
            continue
        ref = float(fx["dnorm." + k])
        got = (st.opt.flat_p[int(o):int(o) + s] - before[int(o):int(o) + s]).double().norm().item()
        assert abs(got - ref) / max(ref, 1e-12) < 0.1, (k, got, ref)     # Adam's first step is ~lr*sign(g): robust to bf16 noise


def test_reduced_step_vs_reference_fixture():
    """T1: reduced-width classify_mamba step (32^3) run by the reference on CPU vs the HIP path end to end."""
    fx = golden("t1_reduced_step.npz")
    _check_step_fixture(fx, dict(f_maps=(8, 16, 32), vit_kwargs=dict(dim=64, depth=2, heads=2, dim_head=16, mlp_dim=128)),
                        (32, 32, 32), 64, 2, 8, 11, tol_fwd=2e-2, tol_grad=8e-2)


def test_full_96_step_vs_reference_fixture():
    """T2 / BASELINE config 1: the full-size model on 2 volumes of 96^3 (reference run on CPU in the build container)."""
    fx = golden("t2_full96_step.npz")
    _check_step_fixture(fx, dict(f_maps=(64, 128, 256)), (96, 96, 96), 512, 6, 8, 21, tol_fwd=2e-2, tol_grad=1e-1)


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
"""


@pytest.mark.gpu
def test_graphed_step_matches_eager_step():
    """ClassifyStep.train_step_graphed (HIP-graph replay of zero_grad + forward + backward) updates the parameters exactly like
    train_step on the same inputs (dropout off so that both are deterministic up to the kernels' f32 atomics).
    Runs in a child process: whole-step graph replay at full size hits an intermittent `HSA_STATUS_ERROR_EXCEPTION` on this ROCm
    (about 2 in 10 processes, already at the commit that introduced the feature, never with AMD_SERIALIZE_KERNEL=3, never in the
    eager path) -- an abort of that kind is reported as an expected failure of this optional feature instead of
    taking the test session down; a numerical mismatch still fails."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = _GRAPHED_STEP_CHILD.format(root=root, src=os.path.join(root, "gfe-mamba_amd"), tests=os.path.join(root, "tests"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    if "HSA_STATUS_ERROR_EXCEPTION" in r.stderr:
        pytest.xfail("HIP-graph replay aborted with HSA_STATUS_ERROR_EXCEPTION (known intermittent runtime fault, optional feature)")
    assert r.returncode == 0 and "GRAPHED_STEP_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])


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
    for mode in ("serial", "pipelined", "pipelined_unannounced"):
        gen, head, ft = build_models(**kw)
        for m in ft.modules():
            if isinstance(m, torch.nn.Dropout):
                m.p = 0.0
        st = ClassifyStep(gen, head, ft)
        losses = []
        for i, b in enumerate(batches):
            if mode == "serial":
                losses.append(st.train_step(*b))
            else:
                nxt = batches[i + 1][0] if (mode == "pipelined" and i + 1 < len(batches)) else None
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
