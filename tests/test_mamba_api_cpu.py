"""Host-side pieces of the Mamba API that need no GPU: npo2 / pad_npo2 (cross_atten/pscan.py:13-33) -- kept for
callers of the reference, checked against its semantics and against the sequence lengths the reference pads (37 -> 64, 4096 -> 4096).
(Row A11, the single-token step, runs on the kernels since round 3: tests/test_head_gpu.py.)  The modules themselves refuse CPU tensors."""
import pytest
import torch


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


def test_modules_refuse_cpu_tensors():
    """there is no CPU fallback -- a CPU tensor raises instead of silently taking a slow path."""
    from cross_atten.mamba import Mamba, MambaConfig, RMSNorm
    m = Mamba(MambaConfig(d_model=32, n_layers=1))
    with pytest.raises(RuntimeError):
        RMSNorm(32)(torch.zeros(2, 3, 32))
    with pytest.raises(RuntimeError):
        m(torch.zeros(2, 3, 32))
    with pytest.raises(RuntimeError), torch.no_grad():
        m.step(torch.zeros(2, 32), [(None, torch.zeros(2, 64, 3))])
