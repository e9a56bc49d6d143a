"""Compile-time guard for the hot kernels (CPU only: hipcc cross-compiles gfx950 without a GPU).

A VGPR spill in the conv main loop cost 30 % in round 1 without failing any parity test, so the register budget of the kernels that
DESIGN.md quotes is asserted here from hipcc's own resource remarks."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "gfe-mamba_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"


def _resources(src, tmp_path):
    out = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=fast", "-c", os.path.join(CSRC, src), *(["-fno-honor-nans"] if src in ("attn.hip", "sscan2.hip") else []),
                          "-o", str(tmp_path / "x.o"), "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-2000:]
    res, cur = {}, None
    for line in out.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = res.setdefault(m.group(1), {})
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
        if m and cur is not None:
            cur[m.group(1).strip()] = int(m.group(2))
    return res


pytestmark = pytest.mark.skipif(not os.path.exists(HIPCC) or shutil.which("make") is None, reason="needs the ROCm toolchain")


def test_conv_hot_variants_do_not_spill(tmp_path):
    res = _resources("conv3d.hip", tmp_path)
    hot = [k for k in res if "conv_igemm_kernelILi4ELi3ELb1" in k]          # 27-tap variants: plain, with GroupNorm partials, with the lift residual, with the fused final 1x1x1 conv
    assert len(hot) == 4
    for k in hot:
        assert res[k]["VGPRs Spill"] == 0 and res[k]["VGPRs"] <= 256, (k, res[k])
    for k in res:                                                          # 8 waves per CU need <= 256 VGPRs everywhere
        if "conv_igemm_kernel" in k:
            assert res[k]["VGPRs"] <= 256


def test_attention_keeps_four_waves_per_simd(tmp_path):
    """8-wave blocks, two per CU: all 448 blocks of the bench shape are resident at once only at <= 128 registers (the LDS image is dynamic
    shared memory so that the launch bound binds: with a static 48 KB array hipcc lowers its occupancy target and ignores it)."""
    res = _resources("attn.hip", tmp_path)
    ks = [k for k in res if "attn_fwd_kernel" in k]                      # the plain kernel and its dropout instantiation
    assert len(ks) == 2
    for k in ks:
        # round 5: the retry loop around the tile loop (untracked pass, tracked fallback) leaves a handful of prologue / epilogue values in
        # scratch -- none inside the tile loop (checked on the ISA when the bound was set: the 48 MFMAs sit between the spill stores and reloads)
        assert res[k]["VGPRs"] <= 128 and res[k]["VGPRs Spill"] <= 8, res[k]


def test_scan_kernels_keep_their_two_waves_per_simd(tmp_path):
    """csrc/sscan2.hip runs 8-wave blocks (4 scan + 4 staging waves), one block per CU (round 6: the per-pair partial rows and the ring of
    three input tiles make the forward's LDS image 116 KB): every instantiation must stay within 256 registers (two waves per SIMD)
    without spills -- the backward keeps a 32-step segment's a_t / h_t pairs in 128 of them -- and within the CU's 160 KiB of LDS."""
    res = _resources("sscan2.hip", tmp_path)
    fwd = [k for k in res if "sscan2_fwd_kernel" in k]
    bwd = [k for k in res if "sscan2_bwd_kernel" in k]
    assert len(fwd) == 8 and len(bwd) == 12          # T x B/C row type x {state pass, full pass} (x {atomics, fixed-order partials} for the full backward)
    for k in fwd + bwd:
        assert res[k]["VGPRs"] + res[k].get("AGPRs", 0) <= 256 and res[k]["VGPRs Spill"] == 0, (k, res[k])
        assert res[k]["LDS Size"] <= 160 * 1024, (k, res[k])


def test_pointwise_conv_streams_from_registers_without_spills_or_lds_in_the_loop(tmp_path):
    """csrc/conv1x1.hip: four-wave blocks, two per CU (`__launch_bounds__(256, 2)`): every instantiation within 256 registers, nothing in
    scratch, and the only LDS is the 1-2 KB the GroupNorm partials are folded through at the block's end (operands go from global memory
    straight into the MFMA registers)."""
    res = _resources("conv1x1.hip", tmp_path)
    ks = [k for k in res if "conv1x1_kernel" in k]
    assert len(ks) == 6                               # (KS, NCT) in {(2, 8), (2, 4), (4, 4)} x {with, without statistics}
    for k in ks:
        assert res[k]["VGPRs"] <= 256 and res[k]["VGPRs Spill"] == 0 and res[k]["ScratchSize"] == 0, (k, res[k])
        assert res[k]["LDS Size"] <= 2048, (k, res[k])


def test_product_library_carries_no_diagnostic_switch(tmp_path):
    """the timing / ablation switches of csrc/ (GFE_EXP_*, CONVT_EXP_*, *_STAMPS) build diagnostic libraries only.
    (1) every switch the sources test is listed in csrc/diag_guard.h; (2) one of them without -DGFE_DIAG does not compile; (3) `make all`
    refuses flags that name one; (4) the in-tree product library does not export the marker a diagnostic library carries."""
    import ctypes
    import re
    guard = open(os.path.join(CSRC, "diag_guard.h")).read()
    used = set()
    for f in os.listdir(CSRC):
        if f.endswith((".hip", ".h")) and f != "diag_guard.h":
            for m in re.finditer(r"#\s*(?:if|ifdef|ifndef|elif)[^\n]*?\b((?:GFE_EXP_|CONVT_EXP_)[A-Z0-9_]+|GFE_[A-Z0-9]*_STAMPS|GFE_ATTN_ALWAYS_TRACK)\b", open(os.path.join(CSRC, f)).read()):
                used.add(m.group(1))
    assert used, "no switches found: the pattern no longer matches the sources"
    missing = sorted(u for u in used if ("defined(%s)" % u) not in guard)
    assert not missing, "switches not listed in diag_guard.h: %s" % missing
    tu = tmp_path / "tu.hip"
    tu.write_text('#include "common.h"\nint main() { return 0; }\n')
    base = [HIPCC, "--offload-arch=gfx950", "-std=c++17", "-fsyntax-only", "--cuda-host-only", "-I", CSRC, str(tu)]
    bad = subprocess.run(base + ["-DGFE_EXP_NOMFMA"], capture_output=True, text=True)
    assert bad.returncode != 0 and "GFE_DIAG" in bad.stderr, bad.stderr[-400:]
    ok = subprocess.run(base + ["-DGFE_EXP_NOMFMA", "-DGFE_DIAG"], capture_output=True, text=True)
    assert ok.returncode == 0, ok.stderr[-400:]
    mk = subprocess.run(["make", "-n", "-C", CSRC, "CXXFLAGS=-DCONVT_EXP_NOW"], capture_output=True, text=True)
    assert mk.returncode != 0 and "diagnostic switches" in mk.stderr
    lib = os.path.join(os.path.dirname(CSRC), "gfe_hip", "libgfe_hip.so")
    if os.path.exists(lib):
        assert not hasattr(ctypes.CDLL(lib), "gfe_diag_build"), "gfe_hip/libgfe_hip.so was built with -DGFE_DIAG"
        exp = os.path.join(os.path.dirname(os.path.dirname(CSRC)), "exp_build", "lib_stamps.so")
        if os.path.exists(exp):
            assert hasattr(ctypes.CDLL(exp), "gfe_diag_build")


def test_scan_forward_counted_barrier_wait_matches_the_emitted_lds_stream(tmp_path):
    """csrc/sscan2.hip, forward scan waves: the tile barrier is preceded by `s_waitcnt lgkmcnt(12)`, not lgkmcnt(0): the next tile's first
    twelve fragment reads stay in flight across the barrier, and -- LDS operations of a wave complete in order -- "at most twelve outstanding"
    means every partial-row WRITE of the tile has landed.  That argument holds only if exactly those twelve reads, and nothing else, were
    issued behind the last write.  The count is a source constant; what hipcc emits is checked here on the ISA of every instantiation."""
    asm = tmp_path / "sscan2.s"
    out = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=fast", "-fno-honor-nans", "-S", "--cuda-device-only",
                          os.path.join(CSRC, "sscan2.hip"), "-o", str(asm)], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-2000:]
    s = asm.read_text()
    seen = 0
    for m in re.finditer(r"^(_ZN[^\n]*sscan2_fwd_kernel[^\n]*Lb0[^\n]*):", s, re.M):
        body = s[m.end():s.index(".Lfunc_end", m.end())]
        lines = [l.strip() for l in body.splitlines()]
        waits = [i for i, l in enumerate(lines) if re.match(r"s_waitcnt lgkmcnt\((\d+)\)", l) and i + 1 < len(lines) and lines[i + 1].startswith("s_barrier")]
        counted = [i for i in waits if not lines[i].startswith("s_waitcnt lgkmcnt(0)")]
        assert len(counted) == 1, (m.group(1), [lines[i] for i in waits])
        i = counted[0]
        n = int(re.match(r"s_waitcnt lgkmcnt\((\d+)\)", lines[i]).group(1))
        reads, j = 0, i - 1
        while j >= 0 and not lines[j].startswith("ds_write"):
            if lines[j].startswith("ds_read"):
                reads += 1
            else:
                assert not lines[j].startswith("ds_"), (m.group(1), lines[j])
            j -= 1
        assert j >= 0 and reads == n, (m.group(1), reads, n)
        seen += 1
    assert seen == 4
