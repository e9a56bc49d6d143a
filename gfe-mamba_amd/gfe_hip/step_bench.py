"""bench.py workload: the classify_mamba training step at 96^3, 8 volumes per GPU (BASELINE config 5's per-GPU share)."""
import time

import torch

from . import det_init as det
from .step import ClassifyStep, build_models

MFMA_BF16_PEAK_TFLOPS = 2500.0


TRAFFIC_JSON = ("profiles", "r06", "traffic_r06.json")
_TRAFFIC_KERNEL_SOURCE = {"conv_igemm_64to64_96cubed_b8": "conv3d.hip", "attn_fwd_b8_h8_n1729": "attn.hip", "attn_bwd_b8_h8_n1729": "attn_bwd.hip", "scan_b8": "sscan2.hip"}


def traffic_is_current(key, rel=TRAFFIC_JSON):
    """True / False: the committed counters were / were not taken on the kernel source that is in the tree now (the collect script stores
    the sha256 of the kernels' .hip files next to the numbers); None: no file, or a file from before the hashes were kept."""
    import hashlib
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    try:
        want = json.load(open(os.path.join(root, *rel)))["kernel_sources"]["sha256"][_TRAFFIC_KERNEL_SOURCE[key]]
        have = hashlib.sha256(open(os.path.join(root, "gfe-mamba_amd", "csrc", _TRAFFIC_KERNEL_SOURCE[key]), "rb").read()).hexdigest()
    except (OSError, KeyError, ValueError):
        return None
    return want == have


def measured_traffic(key, rel=TRAFFIC_JSON):
    """HBM-side bytes per launch from the committed rocprofv3 PMC passes (FETCH_SIZE x2 as the gfx950 correction prescribes + WRITE_SIZE,
    separate passes: tools/collect_profiles_r06.sh); None when the file does not travel with the tree -- or when the kernel's source has
    changed since the counters were taken (a committed number must not silently outlive the kernel it describes)."""
    import json
    import os
    f = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), *rel)
    if traffic_is_current(key, rel) is False:
        return None
    try:
        return float(json.load(open(f))[key]["traffic_bytes"])          # a committed rocprofv3 measurement, not this run's
    except (OSError, KeyError, ValueError):
        return None
# generator conv FLOPs per 96^3 volume: 3x3x3 convs + transposed convs, multiply-add = 2 flop
CONV_K3_GFLOP_PER_VOL = 4 * 195.7 + 4 * 97.8 + 2 * 48.9
GEN_GFLOP_PER_VOL = 1356.1
HEAD_GFLOP_PER_SAMPLE = 11.64


class StepWorkload:
    name = "classify_mamba train step (frozen generator fwd + head fwd/bwd + per-param clip + Adam), 96^3, synthetic"

    def __init__(self, batch, world=1, rank=0, vol=(96, 96, 96), graph=False, pipeline=True, distributed=None):
        self.batch, self.world, self.vol, self.graph, self.pipeline = batch, world, vol, graph, pipeline
        self.vol_tag = "96^3" if tuple(vol) == (96, 96, 96) else "x".join(map(str, vol))
        self.name = StepWorkload.name.replace("96^3", self.vol_tag)
        # The head (~450 launches: zero_grad, forward, loss, backward) can replay from a HIP graph inside the pipeline: host enqueue 8.5 ->
        # 3.8 ms per 11 ms step at B=8.  On one GPU at B=8 the eager head is 0.8 % faster (742-744 vs 734-739 volumes/s) and stays the
        # default; the graph is the default where the host is the risk: batches of 1-4 volumes (host-bound: 1.2-2.5x) and multi-rank runs
        # (N processes enqueue at once and the step time is the MAX over ranks, so one rank's host hiccup costs every rank).
        import os as _os
        # `distributed`: a process group is live (torchrun started us) -- true for a ONE-rank group too, which then runs the multi-rank
        # code path: graphed head, all_reduce of the flat gradient buffer on the head stream in every step
        self.distributed = (world > 1) if distributed is None else bool(distributed)
        self.graph_head = pipeline and (graph or ((batch <= 4 or self.distributed) and _os.environ.get("GFE_NO_AUTO_GRAPH") != "1"))
        gen, head, ft = build_models(vol=vol, seed=0)
        import os
        ov = os.environ.get("GFE_OVERLAP_UPDATE")          # default off (see ClassifyStep); 1 turns it on for A/B runs
        self.step_obj = ClassifyStep(gen, head, ft, world_size=world, overlap_update=None if ov is None else ov == "1",
                                     force_collective=self.distributed)
        x, x_cat, x_num, y = det.det_inputs(batch, vol, seed=1000 + rank)
        self.inputs = [t.cuda() for t in (x, x_cat, x_num, y)]
        self.units = batch

    def step(self):
        if self.graph and not self.pipeline:             # HIP-graph replay of zero_grad + forward + backward (small batches are host-bound)
            return self.step_obj.train_step_graphed(*self.inputs)
        if self.pipeline and self.graph_head:            # the pipeline with the head's ~450 launches replayed from a graph
            return self.step_obj.train_step_pipelined(*self.inputs, x_next=self.inputs[0], graph_head=True)
        if self.pipeline:
            # one head step (this batch) + one generator forward (the next batch; synthetic: the same volumes) per call, on two
            # streams: the frozen generator does not depend on the update (ClassifyStep.train_step_pipelined)
            return self.step_obj.train_step_pipelined(*self.inputs, x_next=self.inputs[0])
        return self.step_obj.train_step(*self.inputs)

    def roofline(self, iters=50):
        """Dominant kernel = conv_igemm (3x3x3 convs of the generator): algorithmic FLOPs / its measured device time, averaged over 50
        back-to-back launches AFTER the timed steps have run (a warm chip at the clock it holds under matrix load: the figure the
        rocprofv3 summary of the same command reproduces; a 10-launch burst on a cool chip read 8 % high in round 2)."""
        from . import nn_ops as K
        gen = self.step_obj.gen
        blk = gen.encoders[0].basic_module
        x = self.inputs[0]
        with torch.no_grad():
            r = blk.lift(x)
            conv = blk.conv2
            w32 = K.pack_conv3(conv.conv.weight, torch.float32)
            g, b = conv.groupnorm.weight.detach().float().contiguous(), conv.groupnorm.bias.detach().float().contiguous()
            ss = K.groupnorm_scale_shift(r, g, b, 8)
            w, tab = K.fold_groupnorm(w32, ss[0], ss[1], K.CONV3_TAPS, 64, 64)
            out = K.conv_igemm(r, w, K.CONV3_TAPS, 64, bias_tab=tab, relu=True)
            run = lambda: K.conv_igemm(r, w, K.CONV3_TAPS, 64, bias_tab=tab, relu=True, out=out)     # the launch alone: no allocation inside the timed loop
            for _ in range(3):
                run()
            torch.cuda.synchronize()
            st = torch.cuda.current_stream()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            for _ in range(iters):
                run()
            e1.record(st)
            e1.synchronize()
            ms = e0.elapsed_time(e1) / iters
        flops = 2.0 * 27 * 64 * 64 * self.batch * self.vol[0] * self.vol[1] * self.vol[2]
        tf = flops / (ms * 1e-3) / 1e12
        return {"bound": "mfma", "achieved": round(tf, 1), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(tf / MFMA_BF16_PEAK_TFLOPS, 4),
                "traffic": measured_traffic("conv_igemm_64to64_96cubed_b8") if (self.batch == 8 and self.vol_tag == "96^3") else None,
                "traffic_source": "profiles/r06/traffic_r06.json (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE passes of this launch, committed with the kernel source's hash; not this run)" if (self.batch == 8 and self.vol_tag == "96^3") else None,
                "kernel": "conv_igemm_kernel<4,3,true> (GroupNorm-folded Conv3d 3x3x3 64->64 @%s, ReLU): the launch the step makes three times per "
                          "forward (encoders.0 conv3, decoders.1 conv2 / conv3), timed alone on operands built from encoders.0's lifted tensor and "
                          "conv2 weights -- the step itself collapses encoders.0 conv2 to a one-channel conv (DESIGN 4.1)" % self.vol_tag,
                "launch_ms": round(ms, 4), "launches_timed": iters,
                "algorithmic_flops": flops}

    def cpu_baseline(self):
        """The oracle (plain torch CPU restatement of the reference path) forward+backward on 1 volume (bounded sample)."""
        from oracle import ref_ops as O
        st = self.step_obj
        f32 = lambda m: {k: (v.float() if v.dtype.is_floating_point else v).cpu() for k, v in m.state_dict().items()}
        sd_g, sd_h, sd_f = f32(st.gen), f32(st.head), f32(st.ft)
        x, x_cat, x_num, y = [t[:1].cpu() for t in self.inputs]
        t0 = time.perf_counter()
        with torch.no_grad():
            mi, mo, pet = O.generator(x, sd_g)
        tr = {k: v.clone().requires_grad_(True) for k, v in list(sd_h.items()) + list(sd_f.items()) if v.dtype.is_floating_point}
        sd_h2 = {k: tr.get(k, v) for k, v in sd_h.items()}
        sd_f2 = {k: tr.get(k, v) for k, v in sd_f.items()}
        pred = O.cross_mamba_both(x_cat, x_num, O.combine_classifier_vit_mid(mi, mo, sd_h2), [x, pet], sd_f2, depth=6, heads=8)
        loss = O.bce_sigmoid(pred, y)
        loss.backward()
        dt = time.perf_counter() - t0
        return {"value": round(1.0 / dt, 4), "unit": "volumes/s", "cores": torch.get_num_threads(), "kind": "port",
                "sample": "oracle/ref_ops.py (torch CPU fp32): generator fwd + head fwd/bwd on 1 volume of %s (no optimiser)" % self.vol_tag}

    def allreduce_stats(self, local=0, iters=10):
        """The step's only collective, timed alone on every rank (max over ranks): SUM all-reduce of the flat f32 gradient buffer.
        bus GB/s = 2 (N-1)/N x bytes / time (the ring all-reduce's per-link traffic).  Under the gloo dry-run transport
        (GFE_DIST_BACKEND=gloo: N processes on one GPU) the figure is that of the host-staged stand-in, and says so."""
        import torch.distributed as dist
        from .step import _is_gloo, all_reduce_, barrier
        g = self.step_obj.opt.flat_g
        gloo = _is_gloo()
        iters = 3 if gloo else iters
        for _ in range(2):
            all_reduce_(g)
        torch.cuda.synchronize()
        barrier(local)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            all_reduce_(g)
        torch.cuda.synchronize()
        dt = torch.tensor([(time.perf_counter() - t0) / iters], dtype=torch.float64, device="cpu" if gloo else "cuda")
        dist.all_reduce(dt, op=dist.ReduceOp.MAX)
        g.zero_()
        n, nbytes = self.world, g.numel() * 4
        return {"bytes": nbytes, "ms": round(dt.item() * 1e3, 4), "algbw_GBs": round(nbytes / dt.item() / 1e9, 1),
                "busbw_GBs": round(2 * (n - 1) / n * nbytes / dt.item() / 1e9, 1), "ranks": n,
                "collective": "all_reduce(SUM) of the flat f32 gradient buffer, " + ("gloo through pinned host memory (dry-run transport, not RCCL)" if gloo else "RCCL")}

    def extra(self):
        if self.vol_tag != "96^3":         # conv FLOPs scale with the voxel count; the ViT / head GEMMs with d_cross and the patch size (not tabulated)
            r = self.vol[0] * self.vol[1] * self.vol[2] / 96.0 ** 3
            return {"generator_conv_gflop_per_volume": round(CONV_K3_GFLOP_PER_VOL * r, 1)}
        return {"generator_gflop_per_volume": GEN_GFLOP_PER_VOL, "head_gflop_per_sample": HEAD_GFLOP_PER_SAMPLE}
