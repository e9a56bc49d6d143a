"""Host-side pieces of the Mamba API that need no GPU:

  * npo2 / pad_npo2 (cross_atten/pscan.py:13-33) -- kept for callers of the reference, checked against its semantics and against the
    sequence lengths the reference pads (37 -> 64, 4096 -> 4096);
  * the single-token inference path Mamba.step / MambaBlock.step / ssm_step (cross_atten/mamba.py:342-405): fed one token at a time it
    must reproduce, position by position, the full-sequence forward that the REFERENCE computed (fixture t0_mamba.npz `y`) -- the
    known-answer relation SURVEY section 4 names for ssm_step."""
import torch

from conftest import golden, rel_err, sub_sd, tt


def test_npo2_and_pad_npo2():
    from cross_atten.pscan import npo2, pad_npo2
    assert [npo2(n) for n in (1, 2, 3, 4, 5, 37, 64, 65, 100, 4096)] == [1, 2, 4, 4, 8, 64, 64, 128, 128, 4096]     # pscan.py:13-18
    g = torch.Generator().manual_seed(0)
    for L in (1, 3, 37, 64):
        X = torch.randn(2, L, 5, 4, generator=g)
        P = pad_npo2(X)
        assert P.shape == (2, npo2(L), 5, 4)                                     # pscan.py:20-33: pads dim 1 only
        assert torch.equal(P[:, :L], X) and torch.count_nonzero(P[:, L:]) == 0   # ... with zeros, the data untouched
        assert X.shape[1] == L                                                   # the input is not modified


def test_token_by_token_step_reproduces_the_references_forward():
    from cross_atten.mamba import Mamba, MambaConfig
    fx = golden("t0_mamba.npz")
    cfg = MambaConfig(d_model=32, n_layers=2)
    m = Mamba(cfg)
    m.load_state_dict(sub_sd(fx, "sd."))
    x, y_ref = tt(fx["x"]), tt(fx["y"])
    B, L, _ = x.shape
    # caches as the reference builds them for inference (mamba.py:330-340): (h = None, the last d_conv - 1 conv inputs = zeros)
    caches = [(None, torch.zeros(B, cfg.d_inner, cfg.d_conv - 1)) for _ in range(cfg.n_layers)]
    outs = []
    with torch.no_grad():
        for t in range(L):
            o, caches = m.step(x[:, t], caches)
            outs.append(o)
    y = torch.stack(outs, 1)
    assert rel_err(y, y_ref) < 1e-5
    # the state the steps leave behind is the scan's final state: one more token must continue the sequence, not restart it
    with torch.no_grad():
        o2, _ = m.step(x[:, 0], caches)
    assert not torch.allclose(o2, outs[0])
