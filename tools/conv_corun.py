#!/usr/bin/env python3
"""What a co-running kernel on another stream does to the 64->64 conv launch (every CU wanted by 256 persistent blocks), with the static
tile shares (GFE_CONV_STATIC=1) and with the per-XCD ticket scheduler: `nblk` blocks of tools/probes/occupier.hip hold their CUs while
the conv launches run.  Stands in for RCCL's all-reduce kernel in a multi-rank step (bench.py --gpus N), which a 1-GPU box cannot run.
    python tools/conv_corun.py [nblk ...]"""
import ctypes
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gfe-mamba_amd"))
out = os.path.join(ROOT, "gpurun_out", "occupier.so")
os.makedirs(os.path.dirname(out), exist_ok=True)
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-shared", "-fPIC", os.path.join(ROOT, "tools", "probes", "occupier.hip"), "-o", out])
import torch
from gfe_hip import nn_ops as K

occ = ctypes.CDLL(out)
occ.occupy.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_void_p]
C, D, B = 64, 96, 8
g = torch.Generator().manual_seed(0)
x = torch.randn(B, D, D, D, C, generator=g).to(torch.bfloat16).cuda()
w32 = K.pack_conv3((torch.randn(C, C, 3, 3, 3, generator=g) / (27 * C) ** 0.5).cuda(), torch.float32)
ss = K.groupnorm_scale_shift(x, torch.ones(C, device="cuda"), torch.zeros(C, device="cuda"), 8)
w, tab = K.fold_groupnorm(w32, ss[0], ss[1], K.CONV3_TAPS, C, C)
y = K.conv_igemm(x, w, K.CONV3_TAPS, C, bias_tab=tab, relu=True)
side = torch.cuda.Stream()
run = lambda: K.conv_igemm(x, w, K.CONV3_TAPS, C, bias_tab=tab, relu=True, out=y)
for _ in range(3):
    run()
torch.cuda.synchronize()
ref = y.clone()
mode = "static shares  " if os.environ.get("GFE_CONV_STATIC") == "1" else "ticket scheduler"
for nblk in [int(a) for a in sys.argv[1:]] or [0, 16, 32, 64]:
    iters = 10
    if nblk:
        occ.occupy(nblk, 256, 40000.0, side.cuda_stream)         # holds its CUs for 40 ms: longer than the timed region
        time.sleep(0.003)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        run()
    e1.record(); e1.synchronize()
    torch.cuda.synchronize()
    assert torch.equal(y, ref)
    print(f"{mode}: conv 64->64 @96^3 B=8 beside {nblk:3d} occupied CUs: {e0.elapsed_time(e1) / iters:.3f} ms per launch")
