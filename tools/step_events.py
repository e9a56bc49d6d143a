#!/usr/bin/env python3
"""Where the pipelined step's time goes, in events (no profiler: the host keeps its normal pace): timing events around every generator forward
(caller's stream) and every head step (head stream) of ClassifyStep.train_step_pipelined, steady state.   python tools/step_events.py [B] [steps] [graph]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gfe-mamba_amd"))
import torch
from gfe_hip import det_init as det
from gfe_hip.step import ClassifyStep, build_models
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
N = int(sys.argv[2]) if len(sys.argv) > 2 else 20
graph = len(sys.argv) > 3 and sys.argv[3] == "graph"
gen, head, ft = build_models()
st = ClassifyStep(gen, head, ft)
ins = [t.cuda() for t in det.det_inputs(B, (96, 96, 96), seed=1)]
for _ in range(6):
    st.train_step_pipelined(*ins, x_next=ins[0], graph_head=graph)
st.join(); torch.cuda.synchronize()
st.trace = []
ref = torch.cuda.Event(enable_timing=True); ref.record()
for _ in range(N):
    st.train_step_pipelined(*ins, x_next=ins[0], graph_head=graph)
st.join(); torch.cuda.synchronize()
tr, st.trace = st.trace, None
gens = [(ref.elapsed_time(a), ref.elapsed_time(b)) for k, a, b in tr if k == "gen"]
heads = [(ref.elapsed_time(a), ref.elapsed_time(b)) for k, a, b in tr if k == "head"]
print("call   gen start    end   (dur) | head start    end   (dur) | head end - gen end | gen start - previous head end")
for i in range(N):
    (g0, g1), (h0, h1) = gens[i], heads[i]
    gap = g0 - heads[i - 1][1] if i else float("nan")
    print("%3d  %9.2f %8.2f (%5.2f) | %9.2f %8.2f (%5.2f) | %6.2f | %6.2f" % (i, g0, g1, g1 - g0, h0, h1, h1 - h0, h1 - g1, gap))
per = (heads[-1][1] - heads[4][1]) / (N - 5)
print("period %.3f ms = %.1f volumes/s; generator %.2f ms, head span %.2f ms, head end behind the generator's end by %.2f ms (median over the last %d calls)"
      % (per, B / per * 1e3, sorted(g1 - g0 for g0, g1 in gens[5:])[(N - 5) // 2], sorted(h1 - h0 for h0, h1 in heads[5:])[(N - 5) // 2],
         sorted(heads[i][1] - gens[i][1] for i in range(5, N))[(N - 5) // 2], N - 5))
