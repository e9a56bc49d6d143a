#!/usr/bin/env python3
"""Host enqueue time vs GPU time of one eager generator forward.   python tools/gen_host_probe.py [B]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gfe-mamba_amd")); sys.path.insert(0, ROOT)
import torch
from gfe_hip.step import build_models
import gfe_hip.det_init as det
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
gen, _, _ = build_models()
x = det.det_inputs(B, (96, 96, 96), seed=1)[0].cuda()
with torch.no_grad():
    for _ in range(5):
        gen(x, output_vit_mid=True)
    torch.cuda.synchronize()
    n = 20
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record()
    for _ in range(n):
        gen(x, output_vit_mid=True)
    t1 = time.perf_counter(); e1.record(); e1.synchronize()
    print("B=%d: host enqueue %.2f ms per forward, GPU span %.2f ms per forward" % (B, (t1 - t0) / n * 1e3, e0.elapsed_time(e1) / n))
