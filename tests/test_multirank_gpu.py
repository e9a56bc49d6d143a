"""The multi-rank code path (BASELINE config 5: batch-sharded DP, RCCL gradient all-reduce; classify_mamba.py:94-109)
on the ONE GPU a test box has.

* a one-rank `nccl` process group stands in for the N-rank one: lazily created communicator, `barrier(device_ids=...)`, watchdog
  thread alive, a real `all_reduce` of the flat gradient buffer on the head stream in every step, head replayed from a HIP graph --
  the exact configuration `bench.py --gpus N` runs -- and must give the same parameters as the plain step;
* `python -m torch.distributed.run --nproc-per-node 1 bench.py --gpus 1` (how the driver starts N ranks) must print the JSON line
  with the `allreduce` object and the per-rank step times;
* `FlatAdam._step(world_size=2)` semantics with the PRODUCT's kernels: the SUM of two half-batch gradients, scaled by 1/2 inside
  the clip/Adam kernel, must reproduce the full-batch step (what tests/test_dp_gloo.py checks with the oracle's gradient on CPU).

Children are fresh processes started with subprocess (a child, never an exec of this GPU-initialised process)."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEV = "cuda"


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _env():
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    return env


def _run(cmd, timeout=900):
    r = subprocess.run(cmd, env=_env(), capture_output=True, text=True, timeout=timeout, cwd=ROOT)
    assert r.returncode == 0, "child failed (%d)\n--- stdout ---\n%s\n--- stderr ---\n%s" % (r.returncode, r.stdout[-3000:], r.stderr[-6000:])
    return r


def test_one_rank_rccl_group_graphed_pipelined_step_equals_the_plain_step(tmp_path):
    child = os.path.join(ROOT, "tests", "dp_child.py")
    a, b = str(tmp_path / "nogroup.pt"), str(tmp_path / "group.pt")
    _run([sys.executable, child, "steps", "nogroup", a])
    _run([sys.executable, child, "steps", "group", b, str(_free_port())])
    ra, rb = torch.load(a), torch.load(b)
    dl = (ra["loss"] - rb["loss"]).abs().max().item()
    dp = (ra["p"] - rb["p"]).abs()
    dg = ((ra["g"] - rb["g"]).abs().max() / ra["g"].abs().max()).item()
    frac = (dp > 1e-6).float().mean().item()
    print("one-rank RCCL group vs no group, 5 graphed pipelined steps: max |dloss| %.2e, last gradient rel err %.2e, parameters: %.2e of the "
          "elements differ by > 1e-6 (max %.2e), losses %s" % (dl, dg, frac, dp.max().item(), [round(v, 6) for v in rb["loss"].tolist()]))
    assert torch.isfinite(rb["loss"]).all() and torch.isfinite(rb["p"]).all()
    # Round 4: the trainable half is bit-reproducible -- every gradient sum has one owner and one order (per-chunk norm partials for the clip
    # factor, column-owner blocks for the norm weights, per-sample / per-channel-group partial rows for the conv1d and scan gradients, a plain
    # embedding scatter): two PROCESSES running the same five steps end with identical bits, with or without a process group.
    # (Round 3 lived with f32 atomics here: gradient 3.7e-6 ... 1e-5 apart, 1e-4 of the parameters beyond 1e-6.)
    assert dl == 0.0 and dg == 0.0, (dl, dg)
    assert torch.equal(ra["p"], rb["p"]), (frac, dp.max().item())


def test_bench_under_torchrun_with_one_rank_reports_the_allreduce():
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1", "--nproc-per-node=1",
           os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "2", "--batch", "2", "--no-cpu-baseline"]
    r = _run(cmd)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    print("bench.py under torchrun (1 rank):", {k: out[k] for k in ("value", "ms_per_step", "allreduce", "ms_per_step_by_rank")})
    assert out["n_gpus"] == 1 and out["steps"] == 3 and out["value"] > 0
    assert out["config"]["hip_graph"] is True and out["config"]["parallelism"] == "dp1"
    ar = out["allreduce"]
    assert ar["ranks"] == 1 and ar["bytes"] > 80e6 and ar["ms"] > 0             # the 21.8 M-parameter flat gradient buffer (87 MB)
    assert len(out["ms_per_step_by_rank"]["ranks"]) == 1
    assert out["roofline"]["frac"] > 0


def test_two_ranks_on_one_gpu_equal_one_process_on_the_whole_batch(tmp_path):
    """BASELINE config 5's code path with two REAL ranks: two fresh processes share cuda:0 under a gloo group (host-staged all-reduce of
    the flat gradient buffer: gfe_hip.step.all_reduce_), each runs ClassifyStep(world_size=2) with the head replayed from a HIP graph for
    3 pipelined steps on its half of a batch of 4 x 96^3; a third process steps the whole batch alone.  DDP semantics of
    classify_mamba.py:69-73, 104-109: mean over ranks of the shard-mean losses' gradients == gradient of the batch-mean loss."""
    child = os.path.join(ROOT, "tests", "dp_child.py")
    port = str(_free_port())
    outs = [str(tmp_path / ("r%d.pt" % r)) for r in range(2)]
    procs = [subprocess.Popen([sys.executable, child, "dp", str(r), "2", outs[r], port], env=_env(), cwd=ROOT,
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(2)]
    logs = []
    try:
        for pr in procs:
            logs.append(pr.communicate(timeout=1500))
    finally:
        for pr in procs:
            if pr.poll() is None:
                pr.kill()                                    # the exact children started above
    for r, (pr, (so, se)) in enumerate(zip(procs, logs)):
        assert pr.returncode == 0, "rank %d failed (%d)\n--- stdout ---\n%s\n--- stderr ---\n%s" % (r, pr.returncode, so[-3000:], se[-6000:])
    one = str(tmp_path / "one.pt")
    _run([sys.executable, child, "dp", "0", "1", one, port])
    r0, r1, ref = torch.load(outs[0]), torch.load(outs[1]), torch.load(one)
    # the bucketed variant (GFE_DP_BUCKETS=4: the flat gradient all-reduced and updated as four slices of whole tensors, last parameters
    # first -- FlatAdam.set_buckets) must leave the SAME bits: every element sees the same sum, every tensor the same norm
    port2 = str(_free_port())
    outs_b = [str(tmp_path / ("b%d.pt" % r)) for r in range(2)]
    env_b = dict(_env(), GFE_DP_BUCKETS="4")
    procs_b = [subprocess.Popen([sys.executable, child, "dp", str(r), "2", outs_b[r], port2], env=env_b, cwd=ROOT,
                                stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(2)]
    logs_b = []
    try:
        for pr in procs_b:
            logs_b.append(pr.communicate(timeout=1500))
    finally:
        for pr in procs_b:
            if pr.poll() is None:
                pr.kill()
    for r, (pr, (so, se)) in enumerate(zip(procs_b, logs_b)):
        assert pr.returncode == 0, "bucketed rank %d failed (%d)\n--- stdout ---\n%s\n--- stderr ---\n%s" % (r, pr.returncode, so[-3000:], se[-6000:])
    b0, b1 = torch.load(outs_b[0]), torch.load(outs_b[1])
    assert torch.equal(b0["p"], r0["p"]) and torch.equal(b1["p"], r1["p"]) and torch.equal(b0["g"], r0["g"]) and torch.equal(b0["loss"], r0["loss"])
    # every rank applies the same update to its replica: bit for bit (the clip factor's norm is a fixed-order sum since round 4)
    assert torch.equal(r0["p"], r1["p"]), "replicas drifted apart: %.3e" % (r0["p"] - r1["p"]).abs().max().item()
    assert torch.equal(r0["g"], r1["g"])
    l_dp = 0.5 * (r0["loss"] + r1["loss"])                                   # BCELoss is a batch mean (classify_mamba.py:67)
    dl = (l_dp - ref["loss"]).abs().max().item()
    dg = ((0.5 * r0["g"] - ref["g"]).abs().max() / ref["g"].abs().max()).item()
    dp = (r0["p"] - ref["p"]).abs()
    frac = (dp > 1e-6).float().mean().item()
    print("2 ranks on one GPU (gloo, graphed pipelined head) vs 1 process on the whole batch, 3 steps: max |dloss| %.2e, last gradient rel err %.2e, "
          "parameters: %.2e of the elements differ by > 1e-6 (max %.2e); losses %s vs %s"
          % (dl, dg, frac, dp.max().item(), [round(v, 6) for v in l_dp.tolist()], [round(v, 6) for v in ref["loss"].tolist()]))
    assert torch.isfinite(r0["loss"]).all() and torch.isfinite(r0["p"]).all()
    assert dl < 5e-6 and dg < 5e-5
    # Adam turns a gradient element that is pure rounding noise into a +-lr step of either sign (see the one-rank test above)
    assert frac < 3e-3, frac
    assert dp.max().item() <= 3 * 2 * 1e-4 * 1.01


def test_bench_under_torchrun_with_two_ranks_on_one_gpu():
    """`torchrun --nproc-per-node 2 bench.py --gpus 2` -- the driver's N = 2 launch -- on a one-GPU box: GFE_DIST_BACKEND=gloo lets both
    ranks share the device.  Exercises bench.py's rank > 0 paths: per-rank inputs and seeds, barriers, max-over-ranks timing, the per-rank
    step times, the roofline leg on rank 0 only while rank 1 waits in the collective's barrier, the all-reduce timing leg."""
    env = _env()
    env["GFE_DIST_BACKEND"] = "gloo"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1", "--nproc-per-node=2",
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2", "--batch", "2"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1500, cwd=ROOT)
    assert r.returncode == 0, "bench failed (%d)\n--- stdout ---\n%s\n--- stderr ---\n%s" % (r.returncode, r.stdout[-3000:], r.stderr[-6000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                       # rank 0 prints the ONE line, rank 1 nothing
    out = json.loads(lines[0])
    print("bench.py under torchrun (2 ranks, one GPU, gloo):", {k: out[k] for k in ("value", "ms_per_step", "allreduce", "ms_per_step_by_rank")})
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["value"] > 0 and out["scaling"] == "weak"
    assert out["config"]["parallelism"] == "dp2" and out["config"]["global_batch"] == 4 and out["config"]["hip_graph"] is True
    assert out["config"]["dist_backend"] == "gloo"
    assert len(out["ms_per_step_by_rank"]["ranks"]) == 2
    ar = out["allreduce"]
    assert ar["ranks"] == 2 and ar["bytes"] > 80e6 and ar["ms"] > 0
    assert out["roofline"]["frac"] > 0 and out["cpu_baseline"] is None       # the CPU leg runs at N = 1 only
    # value = the units ALL ranks processed / the max-over-ranks time
    assert abs(out["value"] - 4 / (out["ms_per_step"] * 1e-3)) / out["value"] < 1e-3


class _FrozenOutputs:
    """Stands in for the frozen generator: hands out precomputed outputs for the rows the test selects, so that the comparison below
    isolates the data-parallel arithmetic from the generator's own dependence on the batch it is run with."""

    def __init__(self, outs):
        self.outs, self.rows = outs, slice(None)

    def eval(self):
        return self

    def __call__(self, x, output_vit_mid=True):
        return tuple(o[self.rows].contiguous() for o in self.outs)


def test_world_size_2_update_from_summed_half_batch_gradients_equals_the_full_batch_update(monkeypatch):
    import gfe_hip.det_init as det
    import gfe_hip.step as S
    from gfe_hip.step import ClassifyStep, build_models
    vol, kw = (32, 32, 32), dict(f_maps=(8, 16, 32), dim=64, depth=2, heads=8, vit_kwargs=dict(dim=64, depth=2, heads=2, dim_head=16, mlp_dim=128), seed=5)
    x, x_cat, x_num, y = [t.to(DEV) for t in det.det_inputs(4, vol, seed=9)]
    gen, _, _ = build_models(vol=vol, **kw)
    with torch.no_grad():
        outs = gen(x, output_vit_mid=True)
    fake = _FrozenOutputs(outs)

    def make(world):
        _, head, ft = build_models(vol=vol, **kw)
        return ClassifyStep(fake, head, ft, world_size=world)

    def drop_off(st):
        for m in st.ft.modules():
            if isinstance(m, torch.nn.Dropout):
                m.p = 0.0                                                    # the GEGLU dropout mask depends on the batch layout

    full, r0, r1 = make(1), make(2), make(2)
    for st in (full, r0, r1):
        drop_off(st)
    other = {}

    def fake_allreduce(flat_grad, world_size, group=None, force=False):      # rank 0's view of all_reduce(SUM) over two ranks
        if world_size == 2:
            flat_grad.add_(other["g"])
        return S.dp_mean_scale(world_size)
    monkeypatch.setattr(S, "allreduce_grads_", fake_allreduce)

    for step in range(2):
        fake.rows = slice(None)
        l_full = full.train_step(x, x_cat, x_num, y)
        # "rank 1": gradient of its shard only (forward + backward, no update)
        fake.rows = slice(2, 4)
        r1.head.train(); r1.ft.train()
        r1.opt.zero_grad()
        pred, _ = r1.forward(x[2:], x_cat[2:], x_num[2:])
        l1 = S.bce_sigmoid(pred.squeeze(1), y[2:])
        l1.backward()
        other["g"] = r1.opt.flat_g.clone()
        # "rank 0": its shard, then the (faked) all-reduce + clip + Adam with world_size = 2
        fake.rows = slice(0, 2)
        l0 = r0.train_step(x[:2], x_cat[:2], x_num[:2], y[:2])
        assert abs(0.5 * (l0.item() + l1.item()) - l_full.item()) < 1e-5     # BCELoss is a batch mean (classify_mamba.py:67)
        g_dp, g_full = 0.5 * r0.opt.flat_g, full.opt.flat_g                  # flat_g keeps the summed, unscaled gradient
        e_g = ((g_dp - g_full).abs().max() / g_full.abs().max()).item()
        dp = (r0.opt.flat_p - full.opt.flat_p).abs()
        e_p, frac = dp.max().item(), (dp > 1e-6).float().mean().item()
        print("step %d: DP(2) vs full batch: gradient rel err %.2e, parameters: max |d| %.2e, %.2e of the elements beyond 1e-6" % (step, e_g, e_p, frac))
        assert e_g < 1e-5, e_g
        # (a DIFFERENT summation order, not a different run: two half-batch sums against one full-batch sum)  Adam divides by sqrt(v): an
        # element whose gradient is rounding noise (k_proj.bias: exactly zero in exact arithmetic) moves by up to lr in either direction
        assert frac < 3e-3 and e_p <= (step + 1) * 2 * 1e-4 * 1.01, (frac, e_p)
        # keep rank 1's replica in step with rank 0's (every rank applies the same update)
        r1.opt.flat_p.copy_(r0.opt.flat_p); r1.opt.flat_p16.copy_(r0.opt.flat_p16)
