"""The classify_mamba training / validation step (reference: classify_mamba.py:94-109 train, :129-135 val) on MI355X.

One process per GPU; with world_size > 1 the batch is sharded over ranks and the flat gradient buffer is all-reduced
(RCCL over xGMI) BEFORE the per-parameter clip, which makes the update identical to a single-process step on the
global batch (BCELoss is a batch mean).
"""
import os

import torch
import torch.nn.functional as F

from . import call, lib
from .head_ops import bce_sigmoid
from .train_ops import Condition, FlatAdam, side_wgrads


# CUs the pipelined step hands to the head's streams (ClassifyStep.head_cus; GFE_HEAD_CUS overrides; 0 = no split)
HEAD_CUS_DEFAULT = 0


def dp_mean_scale(world_size):
    """all_reduce(SUM) followed by this scale == gradient of the global-batch mean loss (equal shards)."""
    return 1.0 / world_size


def shard_batch(tensors, rank, world_size):
    """Contiguous equal shards of the leading (batch) dimension: rank r gets rows [r*B/W, (r+1)*B/W)."""
    out = []
    for t in tensors:
        B = t.shape[0]
        assert B % world_size == 0, "global batch must divide evenly over ranks (BCELoss mean == mean of shard means)"
        n = B // world_size
        out.append(t[rank * n:(rank + 1) * n])
    return out


def _is_gloo(group=None):
    import torch.distributed as dist
    return dist.is_initialized() and str(dist.get_backend(group)).lower() == "gloo"


def all_reduce_(t, op=None, group=None):
    """dist.all_reduce on the backend the process group was created with.  RCCL (backend "nccl") is the product path.  `gloo` exists so that a
    ONE-GPU box can run the N-rank code path with N processes sharing the device (RCCL refuses two ranks on one device): there a device
    tensor is staged through pinned host memory -- stream-ordered copy out, host all-reduce, copy back -- a test / dry-run transport
    (GFE_DIST_BACKEND=gloo in bench.py, tests/test_multirank_gpu.py), never chosen by itself."""
    import torch.distributed as dist
    op = dist.ReduceOp.SUM if op is None else op
    if t.is_cuda and _is_gloo(group):
        h = torch.empty(t.shape, dtype=t.dtype, device="cpu", pin_memory=True)
        h.copy_(t, non_blocking=True)
        torch.cuda.current_stream().synchronize()
        dist.all_reduce(h, op=op, group=group)
        t.copy_(h, non_blocking=True)
        torch.cuda.current_stream().synchronize()          # (h is freed on return)
        return t
    dist.all_reduce(t, op=op, group=group)
    return t


def barrier(local_device=0, group=None):
    """dist.barrier; RCCL wants the device it should use (no eagerly bound communicator), gloo takes none."""
    import torch.distributed as dist
    if _is_gloo(group):
        dist.barrier(group=group)
    else:
        dist.barrier(group=group, device_ids=[local_device])


def allreduce_grads_(flat_grad, world_size, group=None, force=False):
    """The step's only collective: SUM all-reduce of the flat gradient buffer (RCCL on GPUs; gloo in the CPU tests and in the
    two-processes-on-one-GPU dry run).  Returns the scale the optimiser must apply (folded into the clip/Adam kernel instead of a
    separate pass).  `force`: issue the collective for a ONE-rank group too (the identity; lets a one-GPU box run the multi-rank code path)."""
    if world_size > 1 or force:
        all_reduce_(flat_grad, group=group)
    return dp_mean_scale(world_size)


class ClassifyStep:
    def __init__(self, gen, head, ft, lr=1e-4, max_norm=1.0, world_size=1, group=None, overlap_update=None, force_collective=False, head_cus=None, dp_buckets=None):
        self.gen, self.head, self.ft = gen.eval(), head, ft
        # train_step_pipelined: CUs the head's streams own (a multiple of 8 = the same number from every XCD; 0 = both streams share the
        # chip and the head only runs in the gaps between conv launches).  GFE_HEAD_CUS overrides.
        env = os.environ.get("GFE_HEAD_CUS")
        self.head_cus = int(env) if env not in (None, "") else (HEAD_CUS_DEFAULT if head_cus is None else int(head_cus))
        self._gen_stream = None
        self.trace = None                          # tools/step_events.py: a list that collects (generator start, end, head start, end) timing events per pipelined call
        self.all_params = list(head.parameters()) + list(ft.parameters())          # classify_mamba.py:57-61
        self.opt = FlatAdam(self.all_params, lr=lr, max_norm=max_norm)             # Adam(lr=1e-4) + per-parameter clip
        self.opt.force_collective = bool(force_collective)                         # all_reduce even in a one-rank group
        self.world_size, self.group = world_size, group
        # The generator is frozen (classify_mamba.py:53,100), so its forward for step k+1 does not depend on update k: with
        # overlap_update the gradient all-reduce + Adam of step k run on a side stream underneath it (same arithmetic, same order
        # of updates).  On ONE rank there is nothing to hide and the update kernel only takes CUs from the persistent conv kernels
        # (round 1: -3 %), so the mode chooses itself: off at world_size 1, ON from two ranks up, where the step would otherwise wait
        # for an 87 MB ring all-reduce (point-to-point xGMI: ~1-2 ms at 8 ranks) before the next generator may start.
        # GFE_OVERLAP_UPDATE=0|1 and overlap_update=True|False override; GFE_DP_BUCKETS=n (or dp_buckets) splits collective + update
        # into n buckets (FlatAdam.set_buckets).  Both are A/B switches for the first 8-GPU run: bench.py prints what was chosen.
        env_ov = os.environ.get("GFE_OVERLAP_UPDATE")
        if env_ov in ("0", "1"):
            overlap_update = env_ov == "1"
        self.overlap_update = (world_size > 1) if overlap_update is None else bool(overlap_update)
        nb = int(os.environ.get("GFE_DP_BUCKETS", "0") or 0) or int(dp_buckets or 1)
        self.dp_buckets = self.opt.set_buckets(nb) if nb > 1 else 1

    def dp_modes(self):
        """What the multi-rank switches resolved to (bench.py logs it with every line)."""
        return {"world_size": self.world_size, "overlap_update": self.overlap_update, "dp_buckets": self.dp_buckets,
                "collective": "forced" if self.opt.force_collective else ("all_reduce" if self.world_size > 1 else "none")}

    # ---- cross-batch software pipeline -------------------------------------------------------------------------------
    # The generator is frozen (classify_mamba.py:53, 100), so its forward for batch k+1 depends on nothing batch k's head computes
    # or updates.  train_step_pipelined runs the head (forward, backward, all-reduce, clip + Adam) of batch k on a second stream
    # while the caller's stream already runs the generator for batch k+1: the head's ~150 small, latency-bound launches fill the
    # gaps of the generator's persistent conv kernels instead of having the chip to themselves.  Same arithmetic, same order of
    # updates, one generator forward and one head step per call -- measured 15.9 -> 13.8 ms/step at 8 volumes (tools/pipeline_probe.py).
    def _generate_async(self, x):
        """Generator forward on the current stream -- or, with head_cus > 0, on the stream that owns the other 256 - head_cus CUs, ordered
        behind the current stream; returns (x, outputs, event recorded behind them, (version, address) of x)."""
        GS = self._gen_stream
        if GS is None:
            if self.trace is not None:
                g0 = torch.cuda.Event(enable_timing=True)
                g0.record(torch.cuda.current_stream())
            with torch.no_grad():
                outs = self.gen(x, output_vit_mid=True)
            ev = torch.cuda.Event(enable_timing=self.trace is not None)
            ev.record(torch.cuda.current_stream())
            if self.trace is not None:
                self.trace.append(("gen", g0, ev))
            return x, outs, ev, (x._version, x.data_ptr())
        GS.wait_stream(torch.cuda.current_stream())
        old = lib().gfe_conv_reserve_cus(self.head_cus)            # the persistent conv kernels launch one block per CU of THIS stream
        try:
            with torch.cuda.stream(GS), torch.no_grad():
                x.record_stream(GS)
                outs = self.gen(x, output_vit_mid=True)
                ev = torch.cuda.Event()
                ev.record(GS)
        finally:
            lib().gfe_conv_reserve_cus(old)
        return x, outs, ev, (x._version, x.data_ptr())

    def _split_streams(self):
        """Head / side / generator streams on disjoint CU sets (gfe_stream_create_cu_range): the head's chain of ~230 small launches gets
        head_cus CUs of its own -- the same slots of every XCD -- and advances all the time; the persistent conv kernels fill the rest.
        Without the split a conv block occupies a whole CU (496 of 512 registers per SIMD lane, 155 of 160 KB LDS), so the head only ran in
        the gaps between conv launches and cost the step the length of its dependent chain."""
        import ctypes
        from . import train_ops

        def masked(lo, hi):
            h = ctypes.c_void_p()
            call("gfe_stream_create_cu_range", lo, hi, ctypes.byref(h))
            return torch.cuda.ExternalStream(h.value)

        n = self.head_cus
        self._head_stream = masked(0, n)
        train_ops._Side.stream, train_ops._Side.masked = masked(0, n), True    # weight-gradient leaves and the image condition: the head's CUs too
        self._gen_stream = masked(n, 256)

    def join(self):
        """Make the current stream wait for a head step that train_step_pipelined left running on the head stream."""
        ev = getattr(self, "_head_done", None)
        if ev is not None:
            torch.cuda.current_stream().wait_event(ev)
            self._head_done = None

    def _head_step_graphed(self, x, x_cat, x_num, y, mid_input, mid_output, pet):
        """zero_grad + head forward + loss + backward replayed from a HIP graph on the CURRENT stream (the pipeline's head stream): one
        graph launch instead of ~450 kernel launches, so the host stays far ahead of the GPU (eager: 9 ms of enqueue per 10.8 ms step).
        Inputs are copied into the captured buffers; dropout masks change from replay to replay through a device step counter."""
        from . import head_ops as Hd
        ins = (x, x_cat, x_num, y, mid_input, mid_output, pet)
        sig = tuple((tuple(t.shape), t.dtype) for t in ins)
        if getattr(self, "_hgraph", None) is None or self._hsig != sig:
            cur = torch.cuda.current_stream()
            self._hin = [t.clone() for t in ins]
            self._drop_ctr = torch.zeros(1, dtype=torch.int64, device=x.device)
            hin = self._hin

            def fwd_bwd():
                self._drop_ctr.add_(1)
                self.opt.zero_grad()
                mid_feature = self.head(hin[4], hin[5])
                pred = self.ft(hin[1], hin[2], mid_feature, [hin[0], hin[6]])          # (the condition is built inside, beside the token stack)
                loss = bce_sigmoid(pred.squeeze(1), hin[3])
                with side_wgrads():                                # weight gradients beside the chain, joined before the region ends
                    loss.backward()
                return loss.detach()

            Hd.set_dropout_step_counter(self._drop_ctr)
            try:
                for _ in range(2):                                 # warm-up (allocator, lazy packs, function attributes) on this stream
                    fwd_bwd()
                cur.synchronize()
                self._hgraph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self._hgraph, stream=cur, capture_error_mode="thread_local"):     # (RCCL's watchdog thread may touch the runtime meanwhile)
                    self._hloss = fwd_bwd()
            finally:
                Hd.set_dropout_step_counter(None)
            self._hsig = sig
        for dst, src in zip(self._hin, ins):
            dst.copy_(src)
        self._hgraph.replay()
        return self._hloss.clone()            # the captured loss buffer is overwritten by the next replay: hand out a copy

    def train_step_pipelined(self, x, x_cat, x_num, y, x_next=None, graph_head=False):
        """One training step on (x, x_cat, x_num, y); if `x_next` (the next batch's volumes) is given, its generator forward is
        enqueued on the caller's stream right away and overlaps this batch's head.  Returns this batch's loss: produced on the head
        stream, and the caller's stream is ordered behind the kernel that writes it; the PARAMETERS are only safe to read from another
        stream after `join()` (or a device synchronisation)."""
        G = torch.cuda.current_stream()
        if getattr(self, "_head_stream", None) is None:
            if self.head_cus > 0:
                self._split_streams()
            else:
                from . import train_ops
                if train_ops._Side.masked:                      # an earlier step object confined the side stream to its head CUs
                    train_ops._Side.stream, train_ops._Side.masked = None, False
                self._head_stream = torch.cuda.Stream()
        H = self._head_stream
        # Whatever the caller enqueues on its stream from here on (refilling the previous call's input buffers, the usual
        # double-buffered loader) is ordered behind the previous head step, which still read them.  Inputs of THIS call must stay
        # unmodified until the next call returns, or join().
        prev = getattr(self, "_head_done", None)
        if prev is not None:
            G.wait_event(prev)
            if self._gen_stream is not None:
                self._gen_stream.wait_event(prev)
        pf = getattr(self, "_prefetched", None)
        # the announced batch is recognised by identity AND content version AND address: a loader that refills the same tensor in place
        # (or a new tensor at a recycled address) gets a fresh generator forward, never the previous batch's outputs
        if pf is None or pf[0] is not x or pf[3] != (x._version, x.data_ptr()):
            pf = self._generate_async(x)                      # pipeline prologue (or a batch nobody announced)
        self._prefetched = None
        _, (mid_input, mid_output, pet), ev, _ = pf
        H.wait_stream(G)                                      # inputs, and everything else the caller queued before this call
        H.wait_event(ev)
        # Batch k+1's generator goes to the caller's stream BEFORE this batch's head is enqueued (the head waits for the caller's stream only up
        # to the line above).  In steady state the host runs a step ahead and the order of the two enqueues does not matter; right after a
        # synchronisation it does: the head is ~450 launches = 6-8 ms of host time, and a generator enqueued behind them starts that much late
        # (bench.py --steps 20: 748-759 volumes/s against 770-772 at --steps 100 on one box, profiles/r05/step_kw_headfirst.txt; with this order 783-790 at both, step_kw_genfirst_box*.txt).
        nxt_pf = self._generate_async(x_next) if x_next is not None else None
        with torch.cuda.stream(H):
            for t in (x, x_cat, x_num, y, mid_input, mid_output, pet):
                t.record_stream(H)                            # allocated on the caller's stream, read here
            self.head.train(); self.ft.train()
            loss_ready = None
            if self.trace is not None:
                h0 = torch.cuda.Event(enable_timing=True)
                h0.record(H)
            if graph_head:
                loss = self._head_step_graphed(x, x_cat, x_num, y, mid_input, mid_output, pet)
            else:
                self.opt.zero_grad()
                mid_feature = self.head(mid_input, mid_output)
                pred = self.ft(x_cat, x_num, mid_feature, [x, pet])
                loss = bce_sigmoid(pred.squeeze(1), y)                   # classify_mamba.py:104, value + gradient in one launch
                loss_ready = torch.cuda.Event()
                loss_ready.record(H)
                with side_wgrads():
                    loss.backward()
            if loss_ready is None:
                loss_ready = torch.cuda.Event()
                loss_ready.record(H)
            self.opt.step(self.world_size, self.group)
            self._head_done = torch.cuda.Event(enable_timing=self.trace is not None)
            self._head_done.record(H)
            if self.trace is not None:
                self.trace.append(("head", h0, self._head_done))
        self._prefetched = nxt_pf                             # batch k+1's generator, running under batch k's head
        G.wait_event(loss_ready)                              # the loss may be read on the caller's stream (whose next generator is already enqueued)
        return loss.detach()

    def forward(self, x, x_cat, x_num, _before_head=None):
        self.join()
        with torch.no_grad():
            mid_input, mid_output, pet = self.gen(x, output_vit_mid=True)          # classify_mamba.py:100-101
        self.opt.wait_updated()                                                    # first reader of the trainable parameters
        if _before_head is not None:
            _before_head()
        mid_feature = self.head(mid_input, mid_output)                             # :102
        pred = self.ft(x_cat, x_num, mid_feature, [x, pet])             # :103
        return pred, (mid_input, mid_output, pet)

    def train_step(self, x, x_cat, x_num, y):
        self.head.train(); self.ft.train()
        pred, _ = self.forward(x, x_cat, x_num, _before_head=self.opt.zero_grad)   # optimizer.zero_grad() (:109), moved behind the frozen generator
        loss = bce_sigmoid(pred.squeeze(1), y)                                     # :104
        with side_wgrads():                                                        # (weight gradients on a second stream, joined before the optimizer)
            loss.backward()                                                        # :105
        self.opt.step(self.world_size, self.group, overlap=self.overlap_update)   # all-reduce, clip (:106-107), Adam (:108)
        return loss.detach()

    def train_step_graphed(self, x, x_cat, x_num, y):
        """The same step with zero_grad + forward + backward replayed from a HIP graph (the optimiser, with its all-reduce, stays
        eager).  At 8 volumes per GPU the step is GPU-bound and the graph buys ~1 %; at 1-4 volumes the ~8 ms of host enqueue
        dominate and replay removes them.  Shapes must stay fixed; inputs are copied into the captured buffers."""
        g = getattr(self, "_graph", None)
        if g is None or tuple(self._gin[0].shape) != tuple(x.shape):
            self.head.train(); self.ft.train()
            self._gin = [t.clone() for t in (x, x_cat, x_num, y)]

            from . import head_ops as Hd
            self._drop_ctr = torch.zeros(1, dtype=torch.int64, device=x.device)      # fresh dropout masks on every replay (head_ops)
            Hd.set_dropout_step_counter(self._drop_ctr)

            def fwd_bwd():
                self._drop_ctr.add_(1)
                self.opt.zero_grad()
                pred, _ = self.forward(self._gin[0], self._gin[1], self._gin[2])
                loss = bce_sigmoid(pred.squeeze(1), self._gin[3])
                with side_wgrads():
                    loss.backward()
                return loss.detach()

            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):                      # warm-up off the capturing stream (allocator, lazy packs, attributes)
                for _ in range(2):
                    fwd_bwd()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            self._graph = torch.cuda.CUDAGraph()
            try:
                with torch.cuda.graph(self._graph, capture_error_mode="thread_local"):
                    self._gloss = fwd_bwd()
            finally:
                Hd.set_dropout_step_counter(None)
        for dst, src in zip(self._gin, (x, x_cat, x_num, y)):
            dst.copy_(src)
        self.opt.wait_updated()
        self._graph.replay()
        loss = self._gloss.clone()            # (the graph's private buffer is overwritten by the next replay)
        self.opt.step(self.world_size, self.group)
        return loss

    @torch.no_grad()
    def eval_step(self, x, x_cat, x_num):
        self.head.eval(); self.ft.eval()
        pred, _ = self.forward(x, x_cat, x_num)
        return torch.sigmoid(pred)                                                 # :135


def build_models(vol=(96, 96, 96), f_maps=(64, 128, 256), dim=512, depth=6, heads=8, cards=(11, 2, 2, 4, 4, 3, 3), n_cont=25,
                 vit_kwargs=None, seed=0, device="cuda"):
    """Generator + head + classifier with geometry derived from `vol`, weights from the deterministic initialiser."""
    from . import det_init as det
    from classify.classifier import Combine_classfier_vit_mid
    from cross_atten.mamba_transformer import Cross_mamba_both
    from pytorch3dunet.unet3d.model import Residual_mid_UNet3D_vit, vit_geometry
    gen = Residual_mid_UNet3D_vit(1, 1, is_segmentation=False, f_maps=f_maps, vol_size=vol, vit_kwargs=vit_kwargs)
    (H, W), _ = vit_geometry(vol)
    head = Combine_classfier_vit_mid(seq_length=4, in_features=H * W)
    ft = Cross_mamba_both(categories=cards, num_continuous=n_cont, dim=dim, depth=depth, heads=heads, dim_head=dim // heads,
                          d_cross=vol[0] * vol[1])
    gen.load_state_dict(det.det_state_dict(gen.state_dict(), seed=seed, prefix="gen."))
    head.load_state_dict(det.det_state_dict(head.state_dict(), seed=seed, prefix="head."))
    ft.load_state_dict(det.det_state_dict(ft.state_dict(), seed=seed, prefix="ft."))
    return gen.to(device).eval(), head.to(device), ft.to(device)


def smoke():
    """One tiny train step of the whole path (reduced widths, 32^3 volumes) checked against the oracle's forward."""
    from . import det_init as det
    from oracle import ref_ops as O
    gen, head, ft = build_models(vol=(32, 32, 32), f_maps=(8, 16, 32), dim=64, depth=2, heads=8,
                                 vit_kwargs=dict(dim=64, depth=2, heads=2, dim_head=16, mlp_dim=128), seed=11)
    sd_g = {k: v.float().cpu() for k, v in gen.state_dict().items()}
    sd_h = {k: v.float().cpu() for k, v in head.state_dict().items()}
    sd_f = {k: (v.float() if v.dtype.is_floating_point else v).cpu() for k, v in ft.state_dict().items()}
    x, x_cat, x_num, y = det.det_inputs(2, (32, 32, 32), seed=11)
    step = ClassifyStep(gen, head, ft)
    head.eval(); ft.eval()
    pred, _ = step.forward(x.cuda(), x_cat.cuda(), x_num.cuda())
    mi, mo, pet = O.generator(x, sd_g, vit_heads=2, vit_depth=2)
    ref = O.cross_mamba_both(x_cat, x_num, O.combine_classifier_vit_mid(mi, mo, sd_h), [x, pet], sd_f, depth=2, heads=8)
    err = (pred.cpu() - ref).abs().max().item() / max(ref.abs().max().item(), 1e-6)
    assert err < 5e-2, f"smoke: logits differ from the oracle by {err}"
    loss = step.train_step(x.cuda(), x_cat.cuda(), x_num.cuda(), y.cuda())
    assert torch.isfinite(loss).item()
    print("smoke ok: reduced classify_mamba step, logits rel err %.2e, loss %.4f" % (err, loss.item()))
