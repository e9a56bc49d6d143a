#!/usr/bin/env python3
"""Is the pipelined step host-bound?  Host enqueue time per step (no synchronisation inside the loop) vs. the time until the GPU is done."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gfe-mamba_amd")]
import torch
from gfe_hip.step_bench import StepWorkload
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
for pipeline, graph in ((True, False), (True, True), (False, False)):
    wl = StepWorkload(B, pipeline=pipeline, graph=graph)
    for _ in range(5):
        wl.step()
    torch.cuda.synchronize()
    n = 30
    t0 = time.perf_counter()
    for _ in range(n):
        wl.step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"B={B} pipeline={pipeline} graph_head={graph}: host enqueue {1e3 * (t1 - t0) / n:.2f} ms/step, GPU done {1e3 * (t2 - t0) / n:.2f} ms/step", flush=True)
