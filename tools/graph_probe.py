#!/usr/bin/env python3
"""Experiment: capture zero_grad + forward + backward of the classify step in a HIP graph and time replay + eager optimiser."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gfe-mamba_amd")); sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F
from gfe_hip.step_bench import StepWorkload
wl = StepWorkload(8)
st = wl.step_obj
x, xc, xn, y = wl.inputs
def fwd_bwd():
    st.opt.zero_grad()
    pred, _ = st.forward(x, xc, xn)
    loss = F.binary_cross_entropy(torch.sigmoid(pred.squeeze(1)), y.float())
    loss.backward()
    return loss
st.head.train(); st.ft.train()
for _ in range(3):
    wl.step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    wl.step()
torch.cuda.synchronize()
print("eager ms/step", (time.perf_counter() - t0) * 100)
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        fwd_bwd(); st.opt.step(1, None)
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    static_loss = fwd_bwd()
torch.cuda.synchronize()
for _ in range(3):
    g.replay(); st.opt.step(1, None)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    g.replay(); st.opt.step(1, None)
torch.cuda.synchronize()
print("graph ms/step", (time.perf_counter() - t0) * 100, "loss", float(static_loss))
