#!/usr/bin/env python3
"""Latency probe of the small-M GEMMs of the head (M = B*37 = 296): time vs K (fixed overhead vs per-k-tile cost) and vs split-K."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gfe-mamba_amd"))
import torch
from gfe_hip import nn_ops as K
def bench(f, iters=100):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): f()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
g = torch.Generator().manual_seed(0)
for (M, N) in [(296, 512), (296, 2048), (8, 512)]:
    res = []
    for Kd in (64, 128, 256, 512, 1024, 2048):
        a = torch.randn(M, Kd, generator=g).to(torch.bfloat16).cuda()
        b = torch.randn(N, Kd, generator=g).to(torch.bfloat16).cuda()
        res.append(f"K{Kd}:{bench(lambda: K.gemm_nt(a, b, out_dtype=torch.float32, split_k=1)):.1f}")
    print(M, N, " ".join(res), "us")
x = torch.randn(296, 512, device="cuda")
print("empty-ish kernel (cast 296x512):", f"{bench(lambda: K.cast(x, torch.bfloat16)):.1f} us")
