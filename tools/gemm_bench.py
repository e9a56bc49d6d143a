#!/usr/bin/env python3
"""A/B of the two bf16 GEMM main loops on the 3-D ViT's projection shapes (vit_pytorch_diy/vit_3d.py:41-46, 50; B = 8 -> 13 832 token rows):
the persistent LDS-DMA loop (csrc/gemm_dma.hip) against gemm_nt_kernel<128,0,0> (GFE_GEMM_NO_DMA=1), interleaved rounds in ONE process
(cdna_hip_programming.md 5.4 rule 24), random operands (rule 25).  Prints TFLOP/s and the fraction of the 2.5 PFLOP/s bf16 peak.

    python tools/gemm_bench.py [rounds] [iters]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gfe-mamba_amd")]
import torch

os.environ.setdefault("GFE_GEMM_DMA_ALL", "1")      # A/B every shape, also those the dispatch keeps on the old kernel
from gfe_hip import nn_ops as K

BF = torch.bfloat16
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
M = 13832
shapes = [("qkv   ", 1536, 512, dict()), ("out   ", 512, 512, dict(bias=True, res=True, f32=True)),
          ("ff1   ", 2048, 512, dict(bias=True, act=1)), ("ff2   ", 512, 2048, dict(bias=True, res=True, f32=True)),
          ("square", 4096, 4096, dict())]


def timeit(fn):
    st = torch.cuda.current_stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn()
    e0.record(st)
    for _ in range(iters):
        fn()
    e1.record(st)
    e1.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for name, N, Kd, o in shapes:
    m = 4096 if name == "square" else M
    g = torch.Generator().manual_seed(N + Kd)
    a = torch.randn(m, Kd, generator=g).to(BF).cuda()
    b = (torch.randn(N, Kd, generator=g) / Kd ** 0.5).to(BF).cuda()
    bias = torch.randn(N, generator=g).cuda() if o.get("bias") else None
    res = torch.randn(m, N, generator=g).cuda() if o.get("res") else None
    out = torch.empty(m, N, dtype=torch.float32 if o.get("f32") else BF, device="cuda")
    fn = lambda: K.gemm_nt(a, b, bias=bias, res=res, act=o.get("act", 0), out=out)
    arms = ("dma256", "dma128", "old")
    t = {a_: [] for a_ in arms}
    for _ in range(rounds):
        for arm in arms:
            os.environ.pop("GFE_GEMM_NO_DMA", None)
            os.environ.pop("GFE_GEMM_DMA_NJ", None)
            if arm == "old":
                os.environ["GFE_GEMM_NO_DMA"] = "1"
            else:
                os.environ["GFE_GEMM_DMA_NJ"] = "4" if arm == "dma256" else "2"
            t[arm].append(timeit(fn))
    os.environ.pop("GFE_GEMM_NO_DMA", None)
    os.environ.pop("GFE_GEMM_DMA_NJ", None)
    fl = 2.0 * m * N * Kd
    for arm in arms:
        v = sorted(t[arm])
        med = v[len(v) // 2]
        print("%s M=%5d N=%4d K=%4d %s: median %7.1f us (min %7.1f)  %6.0f TFLOP/s = %.3f of peak" % (name, m, N, Kd, arm, med, v[0], fl / med / 1e6, fl / med / 1e6 / 2500))
