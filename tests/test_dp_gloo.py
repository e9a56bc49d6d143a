"""Data-parallel path on CPU: 2 processes, gloo.  The product's DP rule (gfe_hip/step.py) is: shard the batch, all-reduce
(SUM) the flat gradient buffer, scale by 1/world BEFORE the per-parameter clip.  Checked here against the oracle: the
result must equal the single-process gradient of the global batch (BCELoss is a batch mean, classify_mamba.py:67,104),
and the clipped Adam update computed from it must match on every rank."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from gfe_hip.step import allreduce_grads_, dp_mean_scale, shard_batch
from oracle import ref_ops as O

CARDS, NCONT, DIM, DEPTH, HEADS, DCROSS, KEYS = (3, 2, 4), 5, 16, 1, 2, 24, 6


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _weights():
    g = torch.Generator().manual_seed(7)
    r = lambda *s: torch.randn(*s, generator=g) * 0.3
    ED, N, R = 2 * DIM, 16, 1
    sd = {"cls_token": r(1, 1, DIM), "categories_offset": O.categories_offset(CARDS), "categorical_embeds.weight": r(sum(CARDS) + 2, DIM),
          "numerical_embedder.weights": r(NCONT, DIM), "numerical_embedder.biases": r(NCONT, DIM)}
    p = "transformer.layers.0."
    sd.update({p + "norm.weight": torch.ones(DIM), p + "mixer.in_proj.weight": r(2 * ED, DIM), p + "mixer.conv1d.weight": r(ED, 1, 4),
               p + "mixer.conv1d.bias": r(ED), p + "mixer.x_proj.weight": r(R + 2 * N, ED), p + "mixer.dt_proj.weight": r(ED, R),
               p + "mixer.dt_proj.bias": r(ED) - 3, p + "mixer.A_log": torch.log(torch.arange(1, N + 1).float()).repeat(ED, 1),
               p + "mixer.D": torch.ones(ED), p + "mixer.out_proj.weight": r(DIM, ED)})
    for n, (o, i) in dict(q_proj=(DIM, DIM), k_proj=(DIM, DCROSS), v_proj=(DIM, DCROSS), out_proj=(DIM, DIM)).items():
        sd[f"final_cross.{n}.weight"], sd[f"final_cross.{n}.bias"] = r(o, i), r(o)
    sd.update({"final_feed.0.weight": torch.ones(DIM), "final_feed.0.bias": torch.zeros(DIM), "final_feed.1.weight": r(4 * DIM, DIM),
               "final_feed.1.bias": r(4 * DIM), "final_feed.4.weight": r(DIM, 2 * DIM), "final_feed.4.bias": r(DIM),
               "to_logits.0.weight": torch.ones(DIM), "to_logits.0.bias": torch.zeros(DIM), "to_logits.1.weight": r(1, DIM), "to_logits.1.bias": r(1)})
    return sd


def _batch(B):
    g = torch.Generator().manual_seed(11)
    x_cat = torch.stack([torch.randint(0, c, (B,), generator=g) for c in CARDS], 1)
    x_num = torch.randn(B, NCONT, generator=g)
    feat = torch.randn(B, 4, DIM, generator=g)
    mri, pet = torch.randn(B, 1, 4, 6, 3, generator=g), torch.randn(B, 1, 4, 6, 3, generator=g)    # d_cross = 24, keys = 6
    y = torch.randint(0, 2, (B,), generator=g)
    return x_cat, x_num, feat, mri, pet, y


def _flat_grad(sd, batch):
    x_cat, x_num, feat, mri, pet, y = batch
    names = [k for k, v in sd.items() if v.dtype.is_floating_point]
    leaves = {k: sd[k].clone().requires_grad_(True) for k in names}
    sd2 = {k: leaves.get(k, v) for k, v in sd.items()}
    pred = O.cross_mamba_both(x_cat, x_num, feat, [mri, pet], sd2, depth=DEPTH, heads=HEADS)
    O.bce_sigmoid(pred, y).backward()
    return names, [leaves[k].grad for k in names]


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    sd = _weights()
    full = _batch(4)
    shard = shard_batch(full, rank, world)
    names, grads = _flat_grad(sd, shard)
    flat = torch.cat([g.reshape(-1) for g in grads])
    scale = allreduce_grads_(flat, world)            # the product's collective
    flat = flat * scale
    # reference: single-process gradient of the global batch
    _, gfull = _flat_grad(sd, full)
    ref = torch.cat([g.reshape(-1) for g in gfull])
    err = ((flat - ref).abs().max() / ref.abs().max()).item()
    # per-parameter clip + Adam from the reduced gradient: identical on every rank
    sizes = [g.numel() for g in grads]
    upd = torch.cat([O.adam_step(sd[k].reshape(-1), c, torch.zeros(n), torch.zeros(n), 1)[0]
                     for k, c, n in zip(names, O.clip_per_param(list(torch.split(flat, sizes))), sizes)])
    gathered = [torch.zeros_like(upd) for _ in range(world)]
    dist.all_gather(gathered, upd)
    same = all(torch.equal(gathered[0], t) for t in gathered)
    if rank == 0:
        out.put((err, same, scale))
    dist.destroy_process_group()


def test_dp_allreduce_matches_global_batch_gradient():
    assert dp_mean_scale(2) == 0.5
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    err, same, scale = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert scale == 0.5
    assert err < 1e-5, err
    assert same, "ranks diverged after the update"


def test_shard_batch_is_a_partition():
    t = torch.arange(24).view(8, 3)
    parts = [shard_batch([t], r, 4)[0] for r in range(4)]
    assert torch.equal(torch.cat(parts), t)


def test_bucket_plan_covers_every_tensor_once_last_parameters_first():
    """FlatAdam.set_buckets without a GPU: the plan is pure host arithmetic.  Buckets are contiguous ranges of whole tensors,
    ordered from the last parameter to the first (the order in which the backward finishes them), they cover every element and every chunk of
    the table exactly once, and n = 1 / n > #tensors degenerate sensibly."""
    import numpy as np
    from gfe_hip.train_ops import FlatAdam
    sizes = [40000, 7, 16384, 300, 8192, 123456, 64, 9000]
    opt = FlatAdam.__new__(FlatAdam)
    opt.params, opt.sizes = [None] * len(sizes), sizes
    opt.offs = np.concatenate([[0], np.cumsum([(s + 7) // 8 * 8 for s in sizes])])
    tids = []
    for tid, s in enumerate(sizes):
        tids += [tid] * ((s + 8191) // 8192)
    opt._chunk_tids = np.asarray(tids, dtype=np.int64)
    for n in (1, 2, 4, 8, 100):
        nb = opt.set_buckets(n)
        assert 1 <= nb <= min(n, len(sizes))
        b = opt.buckets
        assert b[0][1] == int(opt.offs[-1]) and b[-1][0] == 0 and b[0][3] == len(tids) and b[-1][2] == 0
        for (o0, o1, c0, c1), (p0, p1, d0, d1) in zip(b[:-1], b[1:]):
            assert o0 == p1 and c0 == d1                                   # contiguous, descending
        for o0, o1, c0, c1 in b:
            assert o0 < o1 and c0 < c1 and o0 in set(int(v) for v in opt.offs) and o1 in set(int(v) for v in opt.offs)
            assert len(set(tids[c0:c1])) == sum(1 for t in range(len(sizes)) if o0 <= int(opt.offs[t]) < o1)
    if True:
        opt.set_buckets(4)
        el = [o1 - o0 for o0, o1, _, _ in opt.buckets]
        assert max(el) < 0.75 * int(opt.offs[-1])                          # no bucket swallows the buffer when the sizes allow a split
