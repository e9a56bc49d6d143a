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
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
import gfe_hip.step as S
orig = S.allreduce_grads_
def forced(flat_g, world_size, group):            # world_size 1 would skip the collective: force it
    dist.all_reduce(flat_g, group=group)
    return orig(flat_g, 1, group)
S.allreduce_grads_ = forced
from gfe_hip.step_bench import StepWorkload
wl = StepWorkload(8, world=1, graph=True)
assert wl.graph_head
for _ in range(5):
    wl.step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    loss = wl.step()
torch.cuda.synchronize()
print("dp graph probe ok: %.2f ms/step, loss %.4f" % ((time.perf_counter() - t0) / 20 * 1e3, float(loss)))
dist.destroy_process_group()
