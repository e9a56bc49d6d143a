#!/usr/bin/env python3
"""Single-GPU stand-in for the multi-rank step: an RCCL process group of ONE rank (watchdog thread and communicator alive), the pipelined
step with the head replayed from a HIP graph, and a real all_reduce of the flat gradient buffer on the head stream in every step."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gfe-mamba_amd")]
import torch
import torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
PG = os.environ.get("PROBE_PG", "1") == "1"
MODE = os.environ.get("PROBE_PG_MODE", "eager")          # eager: device_id given (communicator created now); lazy: on the first collective; gloo
if PG:
    if MODE == "eager":
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    elif MODE == "lazy":
        dist.init_process_group("nccl", rank=0, world_size=1)
    elif MODE == "lazy_used":
        dist.init_process_group("nccl", rank=0, world_size=1)
        t_ = torch.zeros(8, device="cuda"); dist.all_reduce(t_); torch.cuda.synchronize()
    else:
        dist.init_process_group("gloo", rank=0, world_size=1)
import gfe_hip.step as S
orig = S.allreduce_grads_
FORCE = os.environ.get("PROBE_ALLREDUCE", "1") == "1" and os.environ.get("PROBE_PG", "1") == "1"
GRAPH = os.environ.get("PROBE_GRAPH", "1") == "1"
def forced(flat_g, world_size, group):            # world_size 1 would skip the collective: force it
    if FORCE:
        dist.all_reduce(flat_g, group=group)
    return orig(flat_g, 1, group)
S.allreduce_grads_ = forced
from gfe_hip.step_bench import StepWorkload
if not GRAPH:
    os.environ["GFE_NO_AUTO_GRAPH"] = "1"
wl = StepWorkload(8, world=1, graph=GRAPH)
assert wl.graph_head == GRAPH
for _ in range(10):
    wl.step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(40):
    loss = wl.step()
torch.cuda.synchronize()
print("dp graph probe ok (graph=%s, allreduce=%s): %.2f ms/step, loss %.4f" % (GRAPH, FORCE, (time.perf_counter() - t0) / 40 * 1e3, float(loss)))
if PG:
    dist.destroy_process_group()
