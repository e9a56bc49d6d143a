"""CPU-side checks of the C-ABI library: it loads, exports every symbol include/gfe_hip.h declares, and
rejects bad arguments with error codes before touching the GPU (no compute calls here)."""
import ctypes

import pytest

import gfe_hip


def test_library_exports_every_declared_symbol():
    protos = gfe_hip.parse_header()
    assert len(protos) >= 6
    L = ctypes.CDLL(gfe_hip.LIB_PATH)
    missing = [n for n in protos if not hasattr(L, n)]
    assert not missing, missing


def test_abi_version_and_arch():
    L = gfe_hip.lib()
    assert L.gfe_abi_version() == gfe_hip._abi_from_header()
    assert L.gfe_build_arch() == b"gfx950"


def test_argument_validation_returns_error_codes():
    L = gfe_hip.lib()
    # NULL pointers
    assert L.gfe_selective_scan_fwd(None, None, None, None, None, None, None, None, None, None, None,
                                    1, 8, 64, 16, 8, 0, 0, None) == -1
    assert L.gfe_pscan_fwd(None, None, None, None, 1, 8, 64, 8, 0, None) == -1
    # bad state size is refused by the planner
    T, nc = ctypes.c_int(), ctypes.c_int()
    assert L.gfe_sscan_plan(1, 8, 64, 5, 0, 0, ctypes.byref(T), ctypes.byref(nc)) == -2
    assert L.gfe_sscan_plan(1, 4096, 1024, 16, 0, 0, ctypes.byref(T), ctypes.byref(nc)) == 0
    assert T.value * nc.value >= 4096 and nc.value > 1


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(gfe_hip, "_lib", None)
    monkeypatch.setattr(gfe_hip, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(gfe_hip.GfeError):
        gfe_hip.lib()


def test_cpu_tensors_are_refused():
    import torch
    from gfe_hip.scan_ops import pscan, selective_scan_tm
    x = torch.zeros(1, 4, 64, 4)
    with pytest.raises(RuntimeError):
        pscan(x, x)
    with pytest.raises(RuntimeError):
        selective_scan_tm(torch.zeros(1, 4, 64), torch.zeros(1, 4, 64), torch.zeros(64, 4), torch.zeros(1, 4, 4), torch.zeros(1, 4, 4))
