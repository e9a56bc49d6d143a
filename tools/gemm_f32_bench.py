#!/usr/bin/env python3
"""The head's exact-f32 products at their own shapes (config 2 at 8 volumes: 296 token rows; the 8-row cross-attention / feed-forward
products), timed back to back with HIP events: python tools/gemm_f32_bench.py  (GFE_F32_NO_KS=1: the staged kernel + split-K reduction
instead of the in-block K split).  Prints us per product (launches included) and the f32-MFMA fraction (157 TFLOP/s dense)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gfe-mamba_amd"))
import torch
from gfe_hip import nn_ops as K

SHAPES = [  # name, M, N, K, b reduction-major, accumulate
    ("in_proj fwd", 296, 2048, 512, False, False), ("x_proj fwd", 296, 64, 1024, False, False), ("dt_proj fwd", 296, 1024, 32, False, False),
    ("out_proj fwd", 296, 512, 1024, False, False), ("out_proj dgrad", 296, 1024, 512, True, False), ("dt_proj dgrad", 296, 32, 1024, True, False),
    ("x_proj dgrad (+=)", 296, 1024, 64, True, True), ("in_proj dgrad", 296, 512, 2048, True, False),
    ("q proj (8 rows)", 8, 512, 512, False, False), ("ff1 (8 rows)", 8, 4096, 512, False, False), ("ff2 (8 rows)", 8, 512, 2048, False, False),
]
dev = "cuda"
tot = 0.0
for name, M, N, Kd, btr, acc in SHAPES:
    a = torch.randn(M, Kd, device=dev)
    b = torch.randn((Kd, N) if btr else (N, Kd), device=dev)
    tgt = torch.zeros(M, N, device=dev) if acc else None
    run = lambda: K.gemm_f32(a, False, b, btr, accum_into=tgt)
    for _ in range(20):
        run()
    # replayed from a HIP graph: the host side of an eager call (two ctypes calls + an allocation) is longer than these kernels
    n = 50
    gr = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        with torch.cuda.graph(gr, stream=side):
            for _ in range(n):
                run()
    torch.cuda.current_stream().wait_stream(side)
    gr.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(4):
        gr.replay()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (4 * n)
    tot += us
    print(f"{name:20s} {M:4d} x {N:4d} x {Kd:4d}  {us:7.2f} us  {2.0 * M * N * Kd / us / 1e6 / 157.0:.3f} of f32 MFMA")
print(f"sum {tot:.1f} us")
