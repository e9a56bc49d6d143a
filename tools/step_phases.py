#!/usr/bin/env python3
"""Wall / GPU time of the phases of one classify step (generator forward, head forward, backward, optimiser), and the CPU enqueue
time of each phase (how far the host runs ahead of the GPU).  python tools/step_phases.py [B]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gfe-mamba_amd")); sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F
from gfe_hip.step_bench import StepWorkload
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
wl = StepWorkload(B)
for _ in range(3):
    wl.step()
torch.cuda.synchronize()
st = wl.step_obj
x, xc, xn, y = wl.inputs
names = ["zero_grad", "generator", "head fwd", "loss+backward", "optimizer"]
acc_cpu = [0.0] * 5; acc_gpu = [0.0] * 5
N = 10
for it in range(N):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
    cpu = []
    torch.cuda.synchronize()
    t = time.perf_counter(); ev[0].record()
    st.head.train(); st.ft.train(); st.opt.zero_grad()
    cpu.append(time.perf_counter() - t); t = time.perf_counter(); ev[1].record()
    with torch.no_grad():
        mi, mo, pet = st.gen(x, output_vit_mid=True)
    cpu.append(time.perf_counter() - t); t = time.perf_counter(); ev[2].record()
    from gfe_hip.train_ops import Condition
    pred = st.ft(xc, xn, st.head(mi, mo), Condition([x, pet]))
    cpu.append(time.perf_counter() - t); t = time.perf_counter(); ev[3].record()
    loss = F.binary_cross_entropy(torch.sigmoid(pred.squeeze(1)), y.float())
    loss.backward()
    cpu.append(time.perf_counter() - t); t = time.perf_counter(); ev[4].record()
    st.opt.step(st.world_size, st.group)
    cpu.append(time.perf_counter() - t); ev[5].record()
    torch.cuda.synchronize()
    for i in range(5):
        acc_cpu[i] += cpu[i]; acc_gpu[i] += ev[i].elapsed_time(ev[i + 1])
for i, n in enumerate(names):
    print(f"{n:14s} cpu enqueue {acc_cpu[i] / N * 1e3:7.2f} ms   gpu span {acc_gpu[i] / N:7.2f} ms")
print("sum cpu", sum(acc_cpu) / N * 1e3, "sum gpu", sum(acc_gpu) / N)
