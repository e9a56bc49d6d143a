#!/usr/bin/env python3
"""Chunk-length sweep of the N = 16 selective scan at small batches (config 2 shapes): ms forward / forward + backward per chunk length.
    python tools/scan_chunk_sweep.py [B ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gfe-mamba_amd")]
import torch
from gfe_hip.scan_ops import selective_scan_tm

L, ED, N = 4096, 1024, 16


def timeit(fn, iters=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / iters


for B in [int(a) for a in sys.argv[1:]] or [1, 2, 4]:
    g = torch.Generator().manual_seed(0)
    mk = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(torch.bfloat16).cuda().requires_grad_(True)
    u, d, z = mk(B, L, ED), mk(B, L, ED, sc=0.1), mk(B, L, ED)
    Bm, Cm = mk(B, L, N), mk(B, L, N)
    A = (-(torch.arange(1, N + 1, dtype=torch.float32)).repeat(ED, 1)).cuda().requires_grad_(True)
    D = torch.ones(ED, device="cuda", requires_grad=True)
    bias = torch.full((ED,), -3.0, device="cuda", requires_grad=True)
    dy = torch.randn(B, L, ED, generator=g).to(torch.bfloat16).cuda()
    for chunk in (0, 128, 256, 512, 1024, 2048):
        def fwd():
            return selective_scan_tm(u, d, A, Bm, Cm, D, z=z, delta_bias=bias, delta_softplus=True, chunk=chunk)

        def step():
            y = fwd()
            for t in (u, d, z, Bm, Cm, A, D, bias):
                t.grad = None
            y.backward(dy)
        with torch.no_grad():
            tf = timeit(fwd)
        ts = min(timeit(step) for _ in range(3))
        print(f"B={B} chunk={chunk:5d}: fwd {tf:.4f} ms  fwd+bwd {ts:.4f} ms", flush=True)
