"""GPU parity of the scan kernels (through the C-ABI) against the oracle and the reference-generated fixtures.
Tolerances (BASELINE.json north_star): <= 1e-3 rel for fp32 I/O, <= 1e-2 rel for bf16 I/O."""
import pytest
import torch

from conftest import golden, rel_err, tt
from oracle import ref_ops as O

pytestmark = pytest.mark.gpu
DEV = "cuda"
TOL32, TOL16 = 1e-3, 1e-2


@pytest.mark.parametrize("L", [1, 2, 3, 4, 5, 37, 64, 100])
@pytest.mark.parametrize("chunk", [0, 8])
def test_pscan_vs_reference_fixture(L, chunk):
    from gfe_hip.scan_ops import pscan
    fx = golden("t0_pscan.npz")
    A = tt(fx[f"pscan_L{L}_A"], device=DEV).requires_grad_(True)
    X = tt(fx[f"pscan_L{L}_X"], device=DEV).requires_grad_(True)
    H = pscan(A, X, chunk)
    assert rel_err(H, tt(fx[f"pscan_L{L}_H"])) < 1e-5
    H.backward(tt(fx[f"pscan_L{L}_gH"], device=DEV))
    assert rel_err(X.grad, tt(fx[f"pscan_L{L}_gX"])) < 1e-5
    gA_ref = tt(fx[f"pscan_L{L}_gA"])
    assert (A.grad.cpu().double() - gA_ref).abs().max() <= 1e-5 * max(1.0, gA_ref.abs().max().item())
    assert torch.all(A.grad[:, 0] == 0)            # pscan.py:221-222
    assert A.grad_fn is None and X.grad_fn is None  # inputs untouched


def test_pscan_bf16_and_mixed_dtype():
    from gfe_hip.scan_ops import pscan
    g = torch.Generator().manual_seed(3)
    A = (torch.rand(2, 70, 16, 16, generator=g) * 0.9 + 0.05)
    X = torch.randn(2, 70, 16, 16, generator=g)
    ref = O.pscan(A.bfloat16().double(), X.bfloat16().double())
    H = pscan(A.bfloat16().to(DEV), X.bfloat16().to(DEV), 16)
    assert H.dtype == torch.bfloat16 and rel_err(H, ref) < TOL16
    H2 = pscan(A.to(DEV), X.bfloat16().to(DEV))          # fp32 A with bf16 X promotes (mamba.py:232, 275-278)
    assert H2.dtype == torch.float32 and rel_err(H2, O.pscan(A.double(), X.bfloat16().double())) < 1e-5


def _ss_inputs(fx, dtype=torch.float32):
    g = lambda k: tt(fx["ss_" + k], device=DEV)
    lp = lambda t: t.to(dtype).requires_grad_(True)
    return dict(x=lp(g("x")), delta=lp(g("delta")), A=g("A").requires_grad_(True), B=lp(g("B")), C=lp(g("C")),
                D=g("D").requires_grad_(True))


@pytest.mark.parametrize("chunk", [0, 8, 16])
def test_selective_scan_vs_reference_fixture(chunk):
    from gfe_hip.scan_ops import selective_scan_tm
    fx = golden("t0_selective_scan.npz")
    i = _ss_inputs(fx)
    y = selective_scan_tm(i["x"], i["delta"], i["A"], i["B"], i["C"], i["D"], chunk=chunk)
    assert rel_err(y, tt(fx["ss_y"])) < TOL32
    assert rel_err(y, tt(fx["ss_y_seq"])) < TOL32
    (y * tt(fx["ss_w"], device=DEV)).sum().backward()
    for k in ("x", "delta", "A", "B", "C", "D"):
        assert rel_err(i[k].grad, tt(fx["ss_g" + k])) < TOL32, k


@pytest.mark.parametrize("chunk", [0, 8])
@pytest.mark.parametrize("dtype,tol", [(torch.float32, TOL32), (torch.bfloat16, TOL16)])
def test_selective_scan_fused_gate_bias_softplus(chunk, dtype, tol):
    """softplus(delta+bias) and y*silu(z) fused (the plug-in contract, mamba.py:251) -- forward vs the reference
    fixture, gradients vs autograd through the oracle."""
    from gfe_hip.scan_ops import selective_scan_tm
    fx = golden("t0_selective_scan.npz")
    i = _ss_inputs(fx, dtype)
    z = tt(fx["ss_z"], device=DEV).to(dtype).requires_grad_(True)
    draw = tt(fx["ss_draw"], device=DEV).to(dtype).requires_grad_(True)
    dbias = tt(fx["ss_dbias"], device=DEV).requires_grad_(True)
    w = tt(fx["ss_w"], device=DEV)
    y = selective_scan_tm(i["x"], draw, i["A"], i["B"], i["C"], i["D"], z=z, delta_bias=dbias, delta_softplus=True, chunk=chunk)
    assert y.dtype == dtype
    if dtype == torch.float32:
        assert rel_err(y, tt(fx["ss_yfn"])) < tol
    (y.float() * w).sum().backward()
    # oracle on the same (possibly bf16-rounded) inputs, fp64
    c = lambda t: t.detach().cpu().double().requires_grad_(True)
    ox, od, oA, oB, oC, oD, oz, ob = c(i["x"]), c(draw), c(i["A"]), c(i["B"]), c(i["C"]), c(i["D"]), c(z), c(dbias)
    oy = O.selective_scan(ox, torch.nn.functional.softplus(od + ob), oA, oB, oC, oD) * torch.nn.functional.silu(oz)
    assert rel_err(y, oy) < tol
    (oy * w.cpu().double()).sum().backward()
    for name, got, ref in (("x", i["x"], ox), ("delta", draw, od), ("A", i["A"], oA), ("B", i["B"], oB), ("C", i["C"], oC),
                           ("D", i["D"], oD), ("z", z, oz), ("bias", dbias, ob)):
        assert rel_err(got.grad, ref.grad) < tol, name


def test_selective_scan_fn_plugin_layout():
    """Channel-major layouts of the reference's slot (mamba.py:245-252): (B,ED,L) / (B,N,L)."""
    from gfe_hip.scan_ops import selective_scan_fn
    fx = golden("t0_selective_scan.npz")
    g = lambda k: tt(fx["ss_" + k], device=DEV)
    tr = lambda t: t.transpose(1, 2)
    delta_cm = tr(g("draw")).contiguous()          # the reference computes delta channel-major (mamba.py:238)
    y = selective_scan_fn(tr(g("x")), delta_cm, g("A"), tr(g("B")), tr(g("C")), g("D"), z=tr(g("z")),
                          delta_bias=g("dbias"), delta_softplus=True)
    assert y.shape == (2, 64, 37)
    assert rel_err(tr(y), tt(fx["ss_yfn"])) < TOL32


def test_selective_scan_fn_plugin_layout_backward():
    """The slot is used in training (mamba.py:243-252 inside MambaBlock.ssm under autograd): gradients w.r.t. every argument THROUGH the
    channel-major views must equal autograd through the oracle's sequential definition on the same values (VERDICT r03 missing #4)."""
    from gfe_hip.scan_ops import selective_scan_fn
    from oracle import ref_ops as O
    fx = golden("t0_selective_scan.npz")
    names = ("x", "draw", "A", "B", "C", "D", "z", "dbias")
    leaf = {k: tt(fx["ss_" + k], device=DEV).requires_grad_(True) for k in names}
    tr = lambda t: t.transpose(1, 2)
    delta_cm = tr(leaf["draw"]).contiguous()                 # a non-leaf: its gradient flows back through the copy into `draw`
    y = selective_scan_fn(tr(leaf["x"]), delta_cm, leaf["A"], tr(leaf["B"]), tr(leaf["C"]), leaf["D"], z=tr(leaf["z"]),
                          delta_bias=leaf["dbias"], delta_softplus=True)
    w = torch.randn(y.shape, generator=torch.Generator().manual_seed(5)).to(DEV)
    (y * w).sum().backward()
    ref = {k: tt(fx["ss_" + k]).double().requires_grad_(True) for k in names}
    # the oracle's restatement of the slot's contract, in the slot's own channel-major layouts
    oy = O.selective_scan_fn(tr(ref["x"]), tr(ref["draw"]), ref["A"], tr(ref["B"]), tr(ref["C"]), ref["D"], z=tr(ref["z"]),
                             delta_bias=ref["dbias"], delta_softplus=True)
    assert rel_err(y, oy.float()) < TOL32
    (oy * w.cpu().double()).sum().backward()
    for k in names:
        assert leaf[k].grad is not None and rel_err(leaf[k].grad, ref[k].grad.float()) < TOL32, k


@pytest.mark.parametrize("dtype,tol", [(torch.float32, TOL32), (torch.bfloat16, TOL16)])
def test_selective_scan_bench_shape_properties(dtype, tol):
    """BASELINE config 2 size (B=1, L=4096, ED=1024, N=16): (a) a 64-channel slab against the oracle's sequential
    definition, (b) chunked == unchunked, (c) linearity in u (C-side), all through the fused kernel."""
    from gfe_hip.scan_ops import selective_scan_tm
    B, L, ED, N = 1, 4096, 1024, 16
    g = torch.Generator().manual_seed(0)
    u = torch.randn(B, L, ED, generator=g)
    draw = torch.randn(B, L, ED, generator=g) * 0.1
    dt = torch.exp(torch.rand(ED, generator=g) * (torch.log(torch.tensor(0.1)) - torch.log(torch.tensor(0.001))) + torch.log(torch.tensor(0.001))).clamp(min=1e-4)
    bias = dt + torch.log(-torch.expm1(-dt))
    A = -(torch.arange(1, N + 1, dtype=torch.float32)).repeat(ED, 1)
    Bm, Cm = torch.randn(B, L, N, generator=g), torch.randn(B, L, N, generator=g)
    D = torch.ones(ED)
    z = torch.randn(B, L, ED, generator=g)
    lp = lambda t: t.to(dtype).to(DEV)
    args = (lp(draw), A.to(DEV), lp(Bm), lp(Cm), D.to(DEV))
    kw = dict(z=lp(z), delta_bias=bias.to(DEV), delta_softplus=True)
    y = selective_scan_tm(lp(u), *args, **kw)                          # automatic chunking (nchunks > 1)
    y1 = selective_scan_tm(lp(u), *args, chunk=L, **kw)                # single chunk
    assert rel_err(y, y1) < (1e-5 if dtype == torch.float32 else 1e-2)
    sl = slice(128, 192)
    r = lambda t: t.to(dtype).double()
    ref = O.selective_scan(r(u)[..., sl], torch.nn.functional.softplus(r(draw)[..., sl] + bias[sl].double()), A[sl].double(),
                           r(Bm), r(Cm), D[sl].double()) * torch.nn.functional.silu(r(z)[..., sl])
    assert rel_err(y[..., sl], ref) < tol
    y2 = selective_scan_tm(lp(2 * u), *args, **kw)
    assert rel_err(y2.float(), 2 * y.float()) < (1e-5 if dtype == torch.float32 else 2e-2)


def test_selective_scan_ragged_and_edge_shapes():
    """L=1, L not a multiple of the sub-chunk, N in {4, 8}, no D / no z / no bias."""
    from gfe_hip.scan_ops import selective_scan_tm
    g = torch.Generator().manual_seed(5)
    for (B, L, ED, N) in [(1, 1, 64, 16), (3, 7, 128, 8), (2, 13, 64, 4), (2, 33, 192, 16)]:
        u = torch.randn(B, L, ED, generator=g).requires_grad_(True)
        d = (torch.rand(B, L, ED, generator=g) * 0.2 + 0.01).requires_grad_(True)
        A = (-torch.rand(ED, N, generator=g) * 4 - 0.1).requires_grad_(True)
        Bm = torch.randn(B, L, N, generator=g).requires_grad_(True)
        Cm = torch.randn(B, L, N, generator=g).requires_grad_(True)
        w = torch.randn(B, L, ED, generator=g)
        ins = [u, d, A, Bm, Cm]
        gi = [t.detach().to(DEV).requires_grad_(True) for t in ins]
        y = selective_scan_tm(*gi, chunk=5 if L > 5 else 0)
        ref = O.selective_scan(u.double(), d.double(), A.double(), Bm.double(), Cm.double(), torch.zeros(ED).double())
        assert rel_err(y, ref) < TOL32, (B, L, ED, N)
        (y * w.to(DEV)).sum().backward()
        gr = torch.autograd.grad((ref * w.double()).sum(), ins)
        for a, b in zip(gi, gr):
            assert rel_err(a.grad, b) < TOL32, (B, L, ED, N)


def test_mambablock_selective_scan_methods_match_reference_fixture():
    """MambaBlock.selective_scan / selective_scan_seq (mamba.py:265-318) keep their names and signatures and agree with the
    reference's outputs for both."""
    from cross_atten.mamba import MambaBlock, MambaConfig
    fx = golden("t0_selective_scan.npz")
    i = _ss_inputs(fx)
    blk = MambaBlock(MambaConfig(d_model=32, n_layers=1)).to(DEV)
    with torch.no_grad():
        y = blk.selective_scan(i["x"], i["delta"], i["A"], i["B"], i["C"], i["D"])
        ys = blk.selective_scan_seq(i["x"], i["delta"], i["A"], i["B"], i["C"], i["D"])
    assert rel_err(y, tt(fx["ss_y"])) < TOL32 and rel_err(ys, tt(fx["ss_y_seq"])) < TOL32


@pytest.mark.gpu
@pytest.mark.parametrize("chunk", [0, 32, 64])
def test_selective_scan_tile_ring_edge_lengths(chunk):
    """Round 6 kernels (csrc/sscan2.hip, N = 16): the input tiles form a ring of three read across the tile barrier, the staging lanes finish a
    tile two barriers after they parked it, the state passes skip the chunk nobody folds.  Lengths around every tile count from one to five
    (1 ... 161 steps: 1, 2, 3, 4, 5+ tiles with full and ragged last tiles), one launch and chunked (chunks of one and two tiles, ragged last
    chunk), fused bias + softplus + D + gate, against the oracle in f64: output and all eight gradients."""
    from gfe_hip.scan_ops import selective_scan_tm
    g = torch.Generator().manual_seed(41 + chunk)
    B, ED, N = 2, 64, 16
    for L in (1, 2, 31, 32, 33, 63, 64, 65, 95, 96, 97, 128, 129, 161):
        if chunk and L <= chunk:
            continue
        u = torch.randn(B, L, ED, generator=g).requires_grad_(True)
        d = (torch.randn(B, L, ED, generator=g) * 0.5).requires_grad_(True)
        A = (-torch.rand(ED, N, generator=g) * 5 - 0.1).requires_grad_(True)
        Bm = torch.randn(B, L, N, generator=g).requires_grad_(True)
        Cm = torch.randn(B, L, N, generator=g).requires_grad_(True)
        D = torch.randn(ED, generator=g).requires_grad_(True)
        z = torch.randn(B, L, ED, generator=g).requires_grad_(True)
        bias = (torch.randn(ED, generator=g) - 2).requires_grad_(True)
        w = torch.randn(B, L, ED, generator=g)
        ins = [u, d, A, Bm, Cm, D, z, bias]
        gi = [t.detach().to(DEV).requires_grad_(True) for t in ins]
        y = selective_scan_tm(gi[0], gi[1], gi[2], gi[3], gi[4], gi[5], z=gi[6], delta_bias=gi[7], delta_softplus=True, chunk=chunk)
        dd = [t.double() for t in ins]
        ref = O.selective_scan(dd[0], torch.nn.functional.softplus(dd[1] + dd[7]), dd[2], dd[3], dd[4], dd[5]) * torch.nn.functional.silu(dd[6])
        assert rel_err(y, ref) < TOL32, (L, chunk, rel_err(y, ref))
        (y * w.to(DEV)).sum().backward()
        gr = torch.autograd.grad((ref * w.double()).sum(), dd)
        for k, (a, b) in enumerate(zip(gi, gr)):
            assert rel_err(a.grad, b) < TOL32, (L, chunk, k, rel_err(a.grad, b))
