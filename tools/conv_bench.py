#!/usr/bin/env python3
"""Micro-benchmark of the implicit-GEMM conv kernel alone (for rocprofv3 --pmc passes and A/B timing).
    python tools/conv_bench.py [C] [D] [B] [iters] [Cin] [stats 0|1] [data random|zeros|const]
data: what the activations hold -- the clock the chip keeps under matrix load depends on how many operand bits toggle"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gfe-mamba_amd"))
import torch
from gfe_hip import nn_ops as K

C = int(sys.argv[1]) if len(sys.argv) > 1 else 64
D = int(sys.argv[2]) if len(sys.argv) > 2 else 96
B = int(sys.argv[3]) if len(sys.argv) > 3 else 8
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 5
CI = int(sys.argv[5]) if len(sys.argv) > 5 else C          # input channels (default: square)
ST = bool(int(sys.argv[6])) if len(sys.argv) > 6 else False  # 1: the variant that also writes the GroupNorm partials of its output
DATA = sys.argv[7] if len(sys.argv) > 7 else "random"
g = torch.Generator().manual_seed(0)
x = torch.randn(B, D, D, D, CI, generator=g).to(torch.bfloat16).cuda()
if DATA == "zeros":
    x.zero_()
elif DATA == "const":
    x.fill_(0.5)
elif DATA == "relu":
    x.clamp_(min=0)
w32 = K.pack_conv3((torch.randn(C, CI, 3, 3, 3, generator=g) / (27 * CI) ** 0.5).cuda(), torch.float32)
ss = K.groupnorm_scale_shift(x, torch.ones(CI, device="cuda"), torch.zeros(CI, device="cuda"), 8)
w, tab = K.fold_groupnorm(w32, ss[0], ss[1], K.CONV3_TAPS, CI, C)
y = K.conv_igemm(x, w, K.CONV3_TAPS, C, bias_tab=tab, relu=True)
stats = (K.new_gn_partials(B, K.conv_stat_slots(B, D, D, D, C), C, x.device), 0) if ST else None
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters):
    K.conv_igemm(x, w, K.CONV3_TAPS, C, bias_tab=tab, relu=True, out=y, stats=stats)
e1.record()
e1.synchronize()
ms = e0.elapsed_time(e1) / iters
fl = 2.0 * 27 * C * CI * B * D ** 3
print(f"conv Cin={CI} C={C} D={D} B={B}{' +stats' if ST else ''} data={DATA}: {ms:.3f} ms  {fl / ms / 1e9:.1f} TFLOP/s")
