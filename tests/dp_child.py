#!/usr/bin/env python3
"""Child process of tests/test_multirank_gpu.py (a FRESH process: never an exec from one that has touched the GPU).

    dp_child.py steps <nogroup|group> <out.pt> [port]
    dp_child.py dp <rank> <world> <out.pt> <port>

runs 5 pipelined training steps of the classify_mamba path with the head replayed from a HIP graph -- the configuration
`bench.py --gpus N` uses for N > 1 -- on 2 volumes of 96^3 and saves the trainable parameters and the losses.  `group`: under a
ONE-rank RCCL process group (lazily created communicator, barrier(device_ids=...), watchdog thread alive) with a real all_reduce
of the flat gradient buffer on the head stream in every step; `nogroup`: no process group at all.

`dp`: BASELINE config 5's code path with REAL ranks on the one GPU a test box has: `world` processes share cuda:0 under a gloo group
(RCCL refuses two ranks on one device; the product's all-reduce helper stages the flat gradient buffer through host memory there), each
runs ClassifyStep(world_size=world) with the head replayed from a HIP graph for 3 pipelined steps on ITS contiguous shard of one global
batch of 4 volumes of 96^3 (gfe_hip.step.shard_batch); world = 1 is the single process stepping the whole batch (no group).  Dropout is
off in both (the GEGLU mask is a hash of the element's position in the LOCAL batch), everything else is the training configuration."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gfe-mamba_amd")]


def dp_main():
    rank, world, out, port = int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5]
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    if world > 1:
        os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", port
        dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(1234)
    torch.cuda.manual_seed(1234)
    import gfe_hip.det_init as det
    from gfe_hip.step import ClassifyStep, barrier, build_models, shard_batch
    vol = (96, 96, 96)
    gen, head, ft = build_models(vol=vol, seed=0)
    for m in ft.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    st = ClassifyStep(gen, head, ft, world_size=world)
    full = [t.cuda() for t in det.det_inputs(4, vol, seed=77)]
    x, x_cat, x_num, y = [t.contiguous() for t in shard_batch(full, rank, world)]
    if world > 1:
        calls, real = [0], dist.all_reduce

        def counted(t, *a, **k):
            calls[0] += 1
            return real(t, *a, **k)
        dist.all_reduce = counted
        barrier(0)
    losses = []
    for _ in range(3):
        losses.append(st.train_step_pipelined(x, x_cat, x_num, y, x_next=x, graph_head=True).clone())
    st.join()
    torch.cuda.synchronize()
    if world > 1:
        dist.all_reduce = real
        assert calls[0] == 3 * st.dp_buckets, f"expected {st.dp_buckets} all_reduce(s) of the gradient buffer per step, saw {calls[0]} in 3 steps"
        barrier(0)
    # flat_g keeps the SUMMED, unscaled gradient of the last step (the 1/world mean lives in the clip/Adam kernel's grad_scale)
    torch.save({"p": st.opt.flat_p.cpu(), "g": st.opt.flat_g.cpu(), "loss": torch.stack(losses).cpu()}, out)
    if world > 1:
        dist.destroy_process_group()
    print("dp_child ok rank %d/%d" % (rank, world), [round(float(l), 6) for l in losses])


def main():
    if sys.argv[1] == "dp":
        return dp_main()
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
