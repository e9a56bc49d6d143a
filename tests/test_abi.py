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


def test_f32_gemm_inblock_rule_is_a_host_side_function_of_shape_and_alignment():
    """gfe_gemm_f32_inblock (host only, no GPU call): the in-block K split takes the head's own shapes -- K-major A, 16-byte aligned operands,
    K % 16 == 0, at most 512 rows -- and nothing the frozen generator computes (its f32 products have 27 * Cout rows: their K cut must not
    depend on the batch), nor the few-tiles / long-K shapes that measured faster on the staged kernel."""
    L = gfe_hip.lib()
    A, B = 0x7f0000000000, 0x7f0000100000                        # fake 16-byte aligned device addresses (never dereferenced)
    q = lambda M, N, K, a_tr=0, b_tr=0, a=A, b=B, lda=None, ldb=None: L.gfe_gemm_f32_inblock(
        a, lda if lda is not None else (M if a_tr else K), a_tr, b, ldb if ldb is not None else (N if b_tr else K), b_tr, M, N, K)
    assert q(296, 512, 1024) == 1 and q(296, 1024, 32) == 1 and q(8, 512, 512) == 1 and q(8, 4096, 512) == 1           # forward products
    assert q(296, 1024, 512, b_tr=1) == 1 and q(296, 512, 2048, b_tr=1) == 1 and q(296, 1024, 64, b_tr=1) == 1        # dgrad products
    assert q(296, 512, 500) == 0 and q(296, 512, 512, a=A + 4) == 0 and q(296, 512, 512, lda=514) == 0                # K % 16, alignment, ld % 4
    assert q(296, 512, 512, a_tr=1) == 0 and q(296, 30, 512, b_tr=1) == 0                                             # reduction-major A; N % 4
    assert q(6912, 1024, 256, b_tr=1) == 0 and q(513, 512, 512) == 0                                                  # the generator's row counts
    assert q(296, 64, 1024) == 0 and q(8, 512, 2048) == 0                                                             # K-major B, few tiles, long K
    assert q(296, 32, 1024, b_tr=1) == 1                                                                              # dt_proj's dgrad (no split-K workspace at its call site)
    assert q(37, 2048, 512) == 1 and q(296, 2048, 512) == 0                                                          # >= 512 tiles of 32 x 32: one staged launch


def test_pointwise_conv_plan_and_shape_rules_are_host_side():
    """gfe_conv1x1 (ABI 49): the GroupNorm-partial slot count is a function of the SAMPLE's voxel count alone (<= 64 blocks of >= 256 voxels, so
    a volume's statistics do not depend on the batch it rides in), the Python gate `conv1x1_ok` names exactly the shapes the entry point
    takes, and bad arguments come back as error codes before anything is launched."""
    from gfe_hip import nn_ops as K
    L = gfe_hip.lib()
    assert L.gfe_conv1x1_stat_slots(48 ** 3) == 64 and L.gfe_conv1x1_stat_slots(24 ** 3) == 54 and L.gfe_conv1x1_stat_slots(1) == 1
    assert L.gfe_conv1x1_stat_slots(256) == 1 and L.gfe_conv1x1_stat_slots(257) == 2 and L.gfe_conv1x1_stat_slots(41 ** 3) <= 64
    assert K.conv1x1_ok(64, 128) and K.conv1x1_ok(128, 256) and K.conv1x1_ok(64, 64) and K.conv1x1_ok(128, 128) and K.conv1x1_ok(64, 512)
    assert not K.conv1x1_ok(1, 64) and not K.conv1x1_ok(32, 64) and not K.conv1x1_ok(64, 96) and not K.conv1x1_ok(128, 512) and not K.conv1x1_ok(256, 128)
    X = 0x7f0000000000                                             # fake device addresses: validation fails first, nothing is dereferenced
    assert L.gfe_conv1x1(None, X, None, X, 1, 8, 64, 128, None, 0, 0, None) == -1                   # NULL x
    assert L.gfe_conv1x1(X, X, None, X, 1, 8, 32, 128, None, 0, 0, None) == -2                      # Cin
    assert L.gfe_conv1x1(X, X, None, X, 1, 8, 128, 512, None, 0, 0, None) == -2                     # eight channel groups
    assert L.gfe_conv1x1(X, X, None, X, 1, 1 << 24, 64, 128, None, 0, 0, None) == -2                # a sample beyond the 32-bit buffer offsets
    assert L.gfe_conv1x1(X, X, None, X, 1, 48 ** 3, 64, 128, X, 10, 0, None) == -2                  # statistics workspace with too few slots


def test_committed_traffic_counters_belong_to_the_kernels_in_the_tree():
    """bench.py's roofline.traffic is a COMMITTED rocprofv3 measurement (profiles/r06/traffic_r06.json), so it must
    not outlive the kernel it was taken on.  The collect script stores the sha256 of the kernels' sources next to the numbers;
    measured_traffic() returns None on a mismatch, and this test fails until tools/collect_profiles_r06.sh has been re-run."""
    import os
    from gfe_hip.step_bench import TRAFFIC_JSON, measured_traffic, traffic_is_current
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not os.path.exists(os.path.join(root, *TRAFFIC_JSON)):
        import pytest
        pytest.skip("no committed traffic file yet")
    for key in ("conv_igemm_64to64_96cubed_b8", "attn_fwd_b8_h8_n1729", "scan_b8"):
        assert traffic_is_current(key) is True, f"{key}: the kernel source changed after the counters were taken -- re-run tools/collect_profiles_r06.sh"
        assert measured_traffic(key) and measured_traffic(key) > 0
