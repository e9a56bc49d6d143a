"""Diagnostic: per-phase s_memtime cycles of the sscan2 kernels (wave 0 of block 0) from a -DGFE_S2_STAMPS build.
    tools/build_exp.sh stamps "-DGFE_S2_STAMPS" sscan2.hip && GFE_HIP_LIB=exp_build/lib_stamps.so python tools/scan_stamps.py [batch]"""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gfe-mamba_amd"))
import torch
import bench, gfe_hip
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
wl = bench.ScanWorkload(B)
for _ in range(3):
    wl.step()
torch.cuda.synchronize()
L = ctypes.CDLL(gfe_hip.LIB_PATH)
buf = (ctypes.c_ulonglong * 48)()
assert L.gfe_dbg_s2_stamps(buf) == 0
v = list(buf)
nt = 4096 // 32
print("fwd  per tile: top/store %d  grp-prologue %d  scan %d  park %d  barrier %d" % tuple(x // nt for x in (v[1], 0, v[2], v[3], v[4])), " [loop-top %d]" % (v[0] // nt))
print("bwd  per segment: loop-top %d  park %d  barrier1 %d  fetch %d  phase1a %d  phase1b %d  phase2a %d  phase2b %d  barrier2 %d  coop %d" % tuple(v[16 + i] // nt for i in (0, 1, 2, 7, 8, 3, 9, 4, 5, 6)))
print("total per step: fwd %.1f  bwd %.1f cycles" % (sum(v[0:12]) / 4096, sum(v[16:28]) / 4096))
