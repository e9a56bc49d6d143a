#!/usr/bin/env python3
"""In-kernel phase stamps of the DMA GEMM (build: tools/build_exp.sh gdma_stamps "-DGFE_GDMA_STAMPS" gemm_dma.hip; run with
GFE_HIP_LIB=exp_build/lib_gdma_stamps.so).  Prints cycles per unit and phase for wave 0 (group 0) and wave 4 (group 1) of block 0."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gfe-mamba_amd")]
import torch
import gfe_hip
from gfe_hip import nn_ops as K
BF = torch.bfloat16
M, N, Kd = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (4096, 4096, 4096)
g = torch.Generator().manual_seed(1)
a = torch.randn(M, Kd, generator=g).to(BF).cuda()
b = (torch.randn(N, Kd, generator=g) / Kd ** 0.5).to(BF).cuda()
out = torch.empty(M, N, dtype=BF, device="cuda")
for _ in range(20):
    K.gemm_nt(a, b, out=out)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 32)()
L = ctypes.CDLL(gfe_hip.LIB_PATH)
assert L.gfe_dbg_gdma_stamps(buf) == 0
tiles = -(-M // 256) * (N // 128)
units = (Kd // 64) * len(range(0, max(1, -(-tiles // 256))))     # units of block 0 (approx.: tiles per block x nk)
names = ["epilogue+zero", "ds_issue", "lgkm_wait", "barrier1", "mfma", "barrier2", "-", "loop"]
for w in (0, 1):
    v = [buf[w * 8 + i] for i in range(8)]
    tot = sum(v)
    print("group %d: total %d ticks over ~%d units = %.0f per unit" % (w, tot, units, tot / max(1, units)))
    print("   " + "  ".join("%s=%.0f" % (n, x / max(1, units)) for n, x in zip(names, v)))
