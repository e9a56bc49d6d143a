#!/usr/bin/env python3
"""Uninitialised-read screen: fill the caching allocator's blocks with a poison pattern, free them, then run the generator at batch 8 and
at batch 1 out of the poisoned pool and compare (every sample must come out bit for bit the same: tests/test_configs_gpu.py).  A kernel
that reads memory nobody wrote shows up as a mismatch or a NaN here on ANY box, not only on one whose memory happens to be dirty.
    python tools/poison_probe.py [pattern nan|big|neg]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gfe-mamba_amd")); sys.path.insert(0, ROOT)
import torch
from gfe_hip.step import build_models
import gfe_hip.det_init as det
pat = sys.argv[1] if len(sys.argv) > 1 else "nan"
val = {"nan": float("nan"), "big": 3.0e4, "neg": -7.5}[pat]


def poison():
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    junk = [torch.full((1 << 28,), val, device="cuda") for _ in range(24)]       # 24 GB of poison in 1 GB blocks
    sizes = [1 << 12, 1 << 16, 1 << 20, 1 << 22, 1 << 24, 1 << 26]
    small = [torch.full((s,), val, device="cuda") for s in sizes for _ in range(64)]
    del junk, small
    torch.cuda.synchronize()                                                      # blocks go back to the caching allocator, contents stay


gen, head, ft = build_models(vol=(96, 96, 96), seed=0)
x = det.det_inputs(8, (96, 96, 96), seed=77)[0].cuda()
poison()
with torch.no_grad():
    o8 = [t.float().clone() for t in gen(x, output_vit_mid=True)]
    worst = [0.0, 0.0, 0.0]
    for b in range(8):
        poison()
        o1 = gen(x[b:b + 1], output_vit_mid=True)
        for j in range(3):
            d = (o8[j][b:b + 1] - o1[j].float()).abs().max().item() / max(1e-30, o8[j][b:b + 1].abs().max().item())
            worst[j] = max(worst[j], d if d == d else float("inf"))
print("poison %s: batch-8 vs batch-1 rel differences (mid_input, mid_output, pet): %.3e %.3e %.3e; finite: %s" % (
    pat, *worst, all(bool(torch.isfinite(t).all()) for t in o8)))
