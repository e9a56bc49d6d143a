#!/usr/bin/env python3
"""Micro-benchmark of the transposed-conv upsampling (8 parity-class launches + resize + skip-sum + GroupNorm partials).
    python tools/convt_bench.py [Cin] [Cout] [D_in] [B] [iters]       default: decoders.1 of the 96^3 model (128 -> 64, 48^3 -> 96^3)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gfe-mamba_amd"))
import torch
from pytorch3dunet.unet3d.buildingblocks import TransposeConvUpsampling

CI = int(sys.argv[1]) if len(sys.argv) > 1 else 128
CO = int(sys.argv[2]) if len(sys.argv) > 2 else 64
D = int(sys.argv[3]) if len(sys.argv) > 3 else 48
B = int(sys.argv[4]) if len(sys.argv) > 4 else 8
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 10
g = torch.Generator().manual_seed(0)
up = TransposeConvUpsampling(CI, CO).cuda()
x = torch.randn(B, D, D, D, CI, generator=g).to(torch.bfloat16).cuda()
skip = torch.randn(B, 2 * D, 2 * D, 2 * D, CO, generator=g).to(torch.bfloat16).cuda()
with torch.no_grad():
    for _ in range(2):
        up(skip, x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        up(skip, x)
    e1.record()
    e1.synchronize()
ms = e0.elapsed_time(e1) / iters
fl = 2.0 * 27 * CI * CO * B * D ** 3
gb = (x.numel() + 2 * skip.numel()) * 2 / 1e9
print(f"convT {CI}->{CO} {D}^3->{2*D}^3 B={B}: {ms:.3f} ms  {fl / ms / 1e9:.1f} TFLOP/s  minimal traffic {gb:.2f} GB = {gb / ms:.2f} TB/s")
