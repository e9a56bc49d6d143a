#!/usr/bin/env python3
"""GPU time of the pieces of the small-batch pipelined step: the head's HIP graph (zero_grad + forward + loss + backward) replayed alone,
the optimizer alone, the generator alone.     python tools/head_graph_probe.py [B]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gfe-mamba_amd")); sys.path.insert(0, ROOT)
import torch
from gfe_hip.step_bench import StepWorkload
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
wl = StepWorkload(B, graph=True)
for _ in range(5):
    wl.step()
wl.step_obj.join()
torch.cuda.synchronize()
st = wl.step_obj
x = wl.inputs[0]


def t(fn, n=30):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


H = st._head_stream
with torch.cuda.stream(H):
    print("head graph replay  %8.1f us  (%d nodes n/a)" % (t(lambda: st._hgraph.replay()), 0))
    print("optimizer          %8.1f us" % t(lambda: st.opt.step(1, None)))
with torch.no_grad():
    print("generator forward  %8.1f us" % t(lambda: st.gen(x, output_vit_mid=True), 10))
print("pipelined step     %8.1f us" % t(lambda: wl.step(), 20))
