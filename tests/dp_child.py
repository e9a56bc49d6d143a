#!/usr/bin/env python3
"""Child process of tests/test_multirank_gpu.py (a FRESH process: never an exec from one that has touched the GPU).

    dp_child.py steps <nogroup|group> <out.pt> [port]

runs 5 pipelined training steps of the classify_mamba path with the head replayed from a HIP graph -- the configuration
`bench.py --gpus N` uses for N > 1 -- on 2 volumes of 96^3 and saves the trainable parameters and the losses.  `group`: under a
ONE-rank RCCL process group (lazily created communicator, barrier(device_ids=...), watchdog thread alive) with a real all_reduce
of the flat gradient buffer on the head stream in every step; `nogroup`: no process group at all."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gfe-mamba_amd")]


def main():
    what, mode, out = sys.argv[1], sys.argv[2], sys.argv[3]
    assert what == "steps" and mode in ("group", "nogroup")
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    if mode == "group":
        os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", sys.argv[4]
        dist.init_process_group("nccl", rank=0, world_size=1)         # no device_id: lazy communicator, as bench.py does
        dist.barrier(device_ids=[0])
    torch.manual_seed(1234)
    torch.cuda.manual_seed(1234)
    from gfe_hip.step_bench import StepWorkload
    wl = StepWorkload(2, world=1, rank=0, graph=True, distributed=(mode == "group"))
    assert wl.graph_head and wl.step_obj.opt.force_collective == (mode == "group")
    if mode == "group":
        calls, real = [0], dist.all_reduce

        def counted(t, *a, **k):
            calls[0] += 1
            return real(t, *a, **k)
        dist.all_reduce = counted
    losses = []
    for _ in range(5):
        losses.append(wl.step().clone())
    wl.step_obj.join()
    torch.cuda.synchronize()
    if mode == "group":
        dist.all_reduce = real
        assert calls[0] == 5, f"expected one all_reduce of the gradient buffer per step, saw {calls[0]}"
        dist.barrier(device_ids=[0])
    torch.save({"p": wl.step_obj.opt.flat_p.cpu(), "g": wl.step_obj.opt.flat_g.cpu(), "loss": torch.stack(losses).cpu()}, out)
    if mode == "group":
        dist.destroy_process_group()
    print("dp_child ok", mode, [round(float(l), 6) for l in losses])


if __name__ == "__main__":
    main()
