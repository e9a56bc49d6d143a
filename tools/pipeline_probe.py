#!/usr/bin/env python3
"""Experiment: software-pipeline the step across batches -- the frozen generator's forward for batch k+1 on one stream while the head's
forward/backward/update for batch k runs on another (the generator does not depend on the update).   python tools/pipeline_probe.py [B] [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gfe-mamba_amd"))
import torch
import torch.nn.functional as F
from gfe_hip import det_init as det
from gfe_hip.step import ClassifyStep, build_models
from gfe_hip.train_ops import Condition
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
N = int(sys.argv[2]) if len(sys.argv) > 2 else 10
gen, head, ft = build_models()
st = ClassifyStep(gen, head, ft)
x, x_cat, x_num, y = [t.cuda() for t in det.det_inputs(B, (96, 96, 96), seed=1)]
head.train(); ft.train()

def gen_fwd():
    with torch.no_grad():
        return gen(x, output_vit_mid=True)

def head_step(outs):
    mi, mo, pet = outs
    st.opt.zero_grad()
    pred = ft(x_cat, x_num, head(mi, mo), Condition([x, pet]))
    loss = F.binary_cross_entropy(torch.sigmoid(pred.squeeze(1)), y.float())
    loss.backward()
    st.opt.step()
    return loss.detach()

def run_serial(n):
    for _ in range(n):
        head_step(gen_fwd())

def run_pipelined(n):
    G, H = torch.cuda.current_stream(), run_pipelined.H
    outs = gen_fwd()
    ev = torch.cuda.Event(); ev.record(G)
    for i in range(n):
        H.wait_event(ev)
        with torch.cuda.stream(H):
            for t in outs: t.record_stream(H)
            head_step(outs)
        if i + 1 < n:
            outs = gen_fwd()                       # batch i+1 under the head of batch i
            ev = torch.cuda.Event(); ev.record(G)
    G.wait_stream(H)
run_pipelined.H = torch.cuda.Stream()

for name, fn in (("serial", run_serial), ("pipelined", run_pipelined), ("serial", run_serial), ("pipelined", run_pipelined)):
    fn(3); torch.cuda.synchronize()
    t0 = time.perf_counter(); fn(N); torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / N
    print(f"{name}: {dt * 1e3:.2f} ms/step  {B / dt:.1f} volumes/s", flush=True)
