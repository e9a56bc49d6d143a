#!/usr/bin/env python3
"""bench.py -- measures BASELINE.json's metric on MI355X and prints ONE JSON line (rank 0).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload step|scan|pscan|vit3d|vit3dtrain|normalise|gen128|gentrain]

workload `step` (default): one classify_mamba training step (frozen generator fwd + head fwd/bwd + per-parameter
clip + Adam) on a synthetic batch of 8 volumes of 96^3 per GPU (BASELINE config 5's per-GPU share == config 3 + bwd).
By default the step is software-pipelined across batches (ClassifyStep.train_step_pipelined): every timed step runs the head
(fwd, bwd, all-reduce, clip + Adam) of batch k on one stream and the frozen generator's forward for batch k+1 on another -- one
generator forward and one head step per step, nothing skipped or cached, identical updates; `--no-pipeline` runs them back to back.
workload `scan`: BASELINE config 2, the fused selective scan alone (L=4096, ED=1024, N=16, bf16), fwd+bwd.
workload `vit3d`: the synthetic 3-D ViT of SURVEY 8-d (96^3, 8^3 patches -> 1729 tokens, dim 512, depth 4, 8 heads x 64), forward;
its roofline object is the flash-attention kernel against the bf16 MFMA peak.

workload `gentrain`: SURVEY 8-f1, one generator training step (L1 + Adam) on 128^3 volumes (gfe_hip/gen_train.py).
workload `gen128`: BASELINE config 4, the generator forward alone (main_gan_vit.py:69) on 2 volumes of 128^3.

N > 1: one rank per GPU over RCCL.  Either the caller starts the ranks (`python -m torch.distributed.run --nproc-per-node N
bench.py --gpus N ...`: RANK / LOCAL_RANK / WORLD_SIZE in the environment), or `python bench.py --gpus N` alone does: with no
WORLD_SIZE in the environment the process starts `torch.distributed.run` with N ranks as a CHILD process -- before this
process has made any GPU call; it never re-executes itself -- and exits with the child's code.  The timed region is bracketed
by barrier + synchronize on both sides; the reported time is the max over ranks.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(ROOT, "gfe-mamba_amd")
for p in (ROOT, SRC):
    if p not in sys.path:
        sys.path.insert(0, p)



def launch_command(n_gpus, argv):
    """The command `bench.py --gpus N` starts when nobody has started the ranks for it: torch.distributed.run, one rank per GPU
    of this node.  `--standalone` lets torchrun bind its rendezvous store to a free port ITSELF (no bind-close-reuse race with
    another bench or profile run on the host); `--local-addr 127.0.0.1` because the container hostname may not resolve."""
    return [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1",
            f"--nproc-per-node={n_gpus}", os.path.abspath(__file__)] + list(argv)


def maybe_self_launch(argv=None):
    """`python bench.py --gpus N` (N > 1) without WORLD_SIZE: start the N ranks as a child process.  Nothing in this process has
    touched the GPU at this point (torch is not even imported), and the child is a child -- no exec of a GPU-initialised process."""
    argv = sys.argv[1:] if argv is None else argv
    if "WORLD_SIZE" in os.environ:
        return None
    n = 1
    for i, a in enumerate(argv):
        if a == "--gpus" and i + 1 < len(argv):
            n = int(argv[i + 1])
        elif a.startswith("--gpus="):
            n = int(a.split("=", 1)[1])
    if n <= 1:
        return None
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: RCCL across processes needs it on this host driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // n)))    # N ranks share the host's cores
    return subprocess.call(launch_command(n, argv), env=env)


if __name__ == "__main__":
    _rc = maybe_self_launch()
    if _rc is not None:
        sys.exit(_rc)

import torch
import torch.distributed as dist

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)
MFMA_BF16_PEAK_TFLOPS = 2500.0


def time_region(fn, iters):
    """Average duration (ms) of fn() measured with HIP events on the stream the kernels are launched on."""
    st = torch.cuda.current_stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(iters):
        fn()
    e1.record(st)
    e1.synchronize()
    return e0.elapsed_time(e1) / iters


class ScanWorkload:
    """BASELINE config 2: u, delta, z ~ N(0,1) (delta*0.1), dt-bias from the reference init
    (mamba.py:150-155), A[e,n] = -(n+1) (mamba.py:160-161,232), B, C ~ N(0,1), D = 1; bf16 I/O, f32 state."""
    name = "selective_scan fwd+bwd, L=4096 ED=1024 N=16 bf16 (config 2)"

    def __init__(self, batch, dtype=torch.bfloat16, L=4096, ED=1024, N=16, graph=None):
        self.B, self.L, self.ED, self.N, self.dtype = batch, L, ED, N, dtype
        self.graph = (batch <= 2) if graph is None else bool(graph)          # host-bound sizes replay from a HIP graph (see _capture)
        self._g_step = self._g_fwd = None
        g = torch.Generator().manual_seed(0)
        dev = "cuda"
        lp = lambda t: t.to(dtype).to(dev)
        self.u = lp(torch.randn(batch, L, ED, generator=g)).requires_grad_(True)
        self.draw = lp(torch.randn(batch, L, ED, generator=g) * 0.1).requires_grad_(True)
        dt = torch.exp(torch.rand(ED, generator=g) * (torch.log(torch.tensor(0.1)) - torch.log(torch.tensor(0.001)))
                       + torch.log(torch.tensor(0.001))).clamp(min=1e-4)
        self.bias = (dt + torch.log(-torch.expm1(-dt))).to(dev).requires_grad_(True)
        self.A = (-(torch.arange(1, N + 1, dtype=torch.float32)).repeat(ED, 1)).to(dev).requires_grad_(True)
        self.Bm = lp(torch.randn(batch, L, N, generator=g)).requires_grad_(True)
        self.Cm = lp(torch.randn(batch, L, N, generator=g)).requires_grad_(True)
        self.D = torch.ones(ED, device=dev).requires_grad_(True)
        self.z = lp(torch.randn(batch, L, ED, generator=g)).requires_grad_(True)
        self.dy = lp(torch.randn(batch, L, ED, generator=g))
        self.units = batch * L          # tokens per step
        s = 2 if dtype == torch.bfloat16 else 4
        # algorithmic bytes: fwd = u,delta,z read + y write + B,C read + A,D,bias; bwd = 7 streams + 4 B/C streams
        self.bytes_fwd = 4 * batch * L * ED * s + 2 * batch * L * N * s + (ED * N + 2 * ED) * 4
        self.bytes_bwd = 7 * batch * L * ED * s + 4 * batch * L * N * s + (2 * ED * N + 4 * ED) * 4

    def fwd(self):
        from gfe_hip.scan_ops import selective_scan_tm
        self.y = selective_scan_tm(self.u, self.draw, self.A, self.Bm, self.Cm, self.D, z=self.z, delta_bias=self.bias,
                                   delta_softplus=True, chunk=int(os.environ.get("GFE_SSCAN_CHUNK", "0")))
        return self.y

    def _eager_step(self):
        y = self.fwd()
        for t in (self.u, self.draw, self.A, self.Bm, self.Cm, self.D, self.z, self.bias):
            t.grad = None
        y.backward(self.dy)

    def _capture(self, fn):
        """fn() replayed from a HIP graph: at B <= 2 the ~25 host-side operations of one forward + backward (allocations, the autograd
        graph, 8 launches) take longer than the GPU needs for them (0.28 ms against 0.15 ms at B = 1, profiles/r02/scan_b1_*), so the
        eager loop measures Python, not the kernels.  Same kernels, same arguments; static input buffers."""
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                fn()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            fn()
        return g

    def step(self):
        if not self.graph:
            return self._eager_step()
        if self._g_step is None:
            self._g_step = self._capture(self._eager_step)
        self._g_step.replay()

    def roofline(self, iters=20):
        with torch.no_grad():
            self.fwd()
            if self.graph:
                if self._g_fwd is None:
                    self._g_fwd = self._capture(self.fwd)
                t_f = time_region(self._g_fwd.replay, iters)
            else:
                t_f = time_region(lambda: self.fwd(), iters)
        t_s = time_region(self.step, iters)
        t_b = max(t_s - t_f, 1e-6)
        gbs = (self.bytes_fwd + self.bytes_bwd) / (t_s * 1e-3) / 1e9
        from gfe_hip.step_bench import measured_traffic
        traffic = measured_traffic("scan_b8") if (self.B, self.L, self.ED) == (8, 4096, 1024) else None
        return {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
                "traffic": traffic, "traffic_source": None if traffic is None else "profiles/r06/traffic_r06.json (committed rocprofv3 PMC passes of this command, not this run)",
                "kernel": "sscan2_fwd + sscan2_bwd (fused selective scan: state-pair lanes, 4 scan + 4 staging waves per block; one launch each way from B = 8, three each way when L is chunked)",
                "fwd_ms": round(t_f, 4), "bwd_ms": round(t_b, 4),
                "fwd_GBs": round(self.bytes_fwd / (t_f * 1e-3) / 1e9, 1), "bwd_GBs": round(self.bytes_bwd / (t_b * 1e-3) / 1e9, 1),
                "algorithmic_bytes": self.bytes_fwd + self.bytes_bwd}

    def cpu_baseline(self):
        """oracle/scan_ref.c (plain C port of mamba.py:288-318), 1 core, forward only, on a bounded sample:
        B=1, L=4096, 256 of the 1024 channels."""
        from oracle import c_oracle
        ch = 256
        f = lambda t: t.detach()[:1, :, :ch].float().cpu().numpy() if t.dim() == 3 and t.shape[-1] == self.ED else t.detach()[:1].float().cpu().numpy()
        t0 = time.perf_counter()
        c_oracle.selective_scan(f(self.u), f(self.draw), self.A.detach()[:ch].cpu().numpy(), f(self.Bm), f(self.Cm),
                                self.D.detach()[:ch].cpu().numpy(), z=f(self.z), bias=self.bias.detach()[:ch].cpu().numpy(), softplus=True)
        dt = time.perf_counter() - t0
        tok_s = self.L / (dt * self.ED / ch)
        return {"value": round(tok_s, 1), "unit": "tokens/s (forward only)", "cores": 1, "kind": "port",
                "sample": f"oracle/scan_ref.c, B=1 L=4096, {ch}/{self.ED} channels, forward only, scaled to ED={self.ED}"}


class PscanWorkload:
    """Operator boundary 1: the MATERIALISED drop-in `pscan(A, X)` of cross_atten/pscan.py:226 at config 2's shape,
    (B, L, D, N) = (B, 4096, 1024, 16) bf16: A in (0.05, 0.95), X ~ N(0,1).  Algorithmic bytes: forward = read A, X + write
    H = 3 B L D N s (403 MB at B = 1), backward = read A, H, gH + write gA, gX = 5 B L D N s (671 MB)."""
    name = "pscan(A, X) fwd+bwd (materialised drop-in), L=4096 D=1024 N=16 bf16"

    def __init__(self, batch, L=4096, D=1024, N=16):
        g = torch.Generator().manual_seed(0)
        self.B, self.L, self.D, self.N = batch, L, D, N
        self.A = (torch.rand(batch, L, D, N, generator=g) * 0.9 + 0.05).to(torch.bfloat16).cuda().requires_grad_(True)
        self.X = torch.randn(batch, L, D, N, generator=g).to(torch.bfloat16).cuda().requires_grad_(True)
        self.gH = torch.randn(batch, L, D, N, generator=g).to(torch.bfloat16).cuda()
        self.units = batch * L
        n = batch * L * D * N * 2
        self.bytes_fwd, self.bytes_bwd = 3 * n, 5 * n

    def fwd(self):
        from cross_atten.pscan import pscan
        return pscan(self.A, self.X)

    def step(self):
        H = self.fwd()
        self.A.grad = self.X.grad = None
        H.backward(self.gH)

    def roofline(self, iters=20):
        with torch.no_grad():
            self.fwd()
            t_f = time_region(self.fwd, iters)
        self.step()
        t_s = time_region(self.step, iters)
        t_b = max(t_s - t_f, 1e-6)
        gbs = (self.bytes_fwd + self.bytes_bwd) / (t_s * 1e-3) / 1e9
        return {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": None,
                "kernel": "pscan_fwd + pscan_bwd (csrc/pscan.hip: streaming first-order recurrence over (B, L, D*N), no power-of-two padding)",
                "fwd_ms": round(t_f, 4), "bwd_ms": round(t_b, 4), "fwd_GBs": round(self.bytes_fwd / (t_f * 1e-3) / 1e9, 1),
                "bwd_GBs": round(self.bytes_bwd / (t_b * 1e-3) / 1e9, 1), "algorithmic_bytes": self.bytes_fwd + self.bytes_bwd}

    def cpu_baseline(self):
        """oracle.ref_ops.pscan + pscan_grads (torch CPU restatement of pscan.py:36-224) on a slab: B=1, L=4096, 64 of the 1024 channels."""
        from oracle import ref_ops as O
        ch = 64
        A, X, gH = (t.detach()[:1, :, :ch].float().cpu() for t in (self.A, self.X, self.gH))
        t0 = time.perf_counter()
        H = O.pscan(A, X)
        O.pscan_grads(A, H, gH)
        dt = time.perf_counter() - t0
        return {"value": round(self.L / (dt * self.D / ch), 1), "unit": "tokens/s", "cores": torch.get_num_threads(), "kind": "port",
                "sample": f"oracle.ref_ops.pscan + pscan_grads, B=1 L=4096, {ch}/{self.D} channels, fp32 torch CPU, scaled to D={self.D}"}


class Vit3dWorkload:
    """SURVEY 8-d "MFMA 3D-ViT attention row": vit_3d.ViT(image_size=96, image_patch_size=8, frames=96, frame_patch_size=8,
    channels=1, dim=512, depth=4, heads=8, dim_head=64, mlp_dim=2048, num_classes=1), B volumes of 96^3, bf16 GEMM/attention
    operands, deterministic random-init weights.  Synthetic: the reference never instantiates this module."""
    name = "vit_3d.ViT forward, 96^3 / 8^3 patches = 1729 tokens, dim 512 depth 4 heads 8x64 (synthetic, SURVEY 8-d)"

    def __init__(self, batch):
        import gfe_hip.det_init as det
        from vit_pytorch_diy.vit_3d import ViT
        self.B = batch
        m = ViT(image_size=96, image_patch_size=8, frames=96, frame_patch_size=8, channels=1, dim=512, depth=4, heads=8,
                dim_head=64, mlp_dim=2048, num_classes=1)
        m.load_state_dict(det.det_state_dict(m.state_dict(), seed=21, prefix="vit3d."))
        self.m = m.cuda().eval()
        g = torch.Generator().manual_seed(0)
        self.x = torch.randn(batch, 1, 96, 96, 96, generator=g).clamp_(-1, 1).cuda()
        self.units = batch
        self.n, self.H, self.dh = 1729, 8, 64

    def step(self):
        with torch.no_grad():
            return self.m(self.x)

    def roofline(self, iters=50):
        from gfe_hip import nn_ops as K
        B, n, H, dh = self.B, self.n, self.H, self.dh
        g = torch.Generator().manual_seed(1)
        qkv = torch.randn(B * n, 3 * H * dh, generator=g).to(torch.bfloat16).cuda()
        inner = H * dh
        f = lambda: K.attention_fwd(qkv[:, :inner], qkv[:, inner:2 * inner], qkv[:, 2 * inner:], B, H, n, dh, dh ** -0.5)
        f()
        ms = time_region(f, iters)
        flops = 4.0 * B * H * n * n * dh                 # attention FLOPs per layer = 4 B h n^2 d
        tf = flops / (ms * 1e-3) / 1e12
        return {"bound": "mfma", "achieved": round(tf, 1), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": round(tf / MFMA_BF16_PEAK_TFLOPS, 4), "traffic": self._traffic(),
                "traffic_source": "profiles/r06/traffic_r06.json (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE passes of this launch, committed; not this run)" if self.B == 8 else None,
                "kernel": "attn_fwd_kernel (one layer, B x 8 heads x 1729 x 64)", "launches_timed": iters,
                "launch_ms": round(ms, 4), "algorithmic_flops": flops}

    def _traffic(self):
        from gfe_hip.step_bench import measured_traffic
        return measured_traffic("attn_fwd_b8_h8_n1729") if self.B == 8 else None

    def cpu_baseline(self):
        """oracle.ref_ops.vit3d (torch CPU restatement of vit_3d.py:113-128) on ONE volume."""
        from oracle import ref_ops as O
        sd = {k: v.detach().float().cpu() for k, v in self.m.state_dict().items()}
        x = self.x[:1].cpu()
        t0 = time.perf_counter()
        with torch.no_grad():
            O.vit3d(x, sd, "", frame_patch=8, patch=8, heads=8, depth=4)
        dt = time.perf_counter() - t0
        return {"value": round(1.0 / dt, 3), "unit": "volumes/s", "cores": torch.get_num_threads(), "kind": "port",
                "sample": "oracle.ref_ops.vit3d, 1 volume of 96^3, fp32, torch CPU"}


class Vit3dTrainWorkload(Vit3dWorkload):
    """The same module under autograd, as the reference would train it (vit_3d.py:47-57 through autograd): forward_train, mean of the
    logits as a stand-in loss, backward, FlatAdam update.  Roofline object: the attention BACKWARD (gfe_attention_bwd = prep + dK/dV + dQ
    launches) of one layer against the bf16 MFMA peak, algorithmic flops = 5 matrix products = 10 B h n^2 d."""
    name = "vit_3d.ViT training step (forward + backward + Adam), 96^3 / 8^3 patches = 1729 tokens, dim 512 depth 4 heads 8x64 (synthetic)"

    def __init__(self, batch):
        super().__init__(batch)
        from gfe_hip.train_ops import FlatAdam
        self.m.train()
        self.opt = FlatAdam(list(self.m.parameters()), lr=1e-4, max_norm=float("inf"))
        self.loss = None

    def step(self):
        self.opt.zero_grad()
        loss = self.m(self.x).mean()
        loss.backward()
        self.opt.step()
        self.loss = loss.detach()

    def roofline(self, iters=50):
        from gfe_hip import nn_ops as K
        B, n, H, dh = self.B, self.n, self.H, self.dh
        g = torch.Generator().manual_seed(1)
        inner = H * dh
        qkv = torch.randn(B * n, 3 * inner, generator=g).to(torch.bfloat16).cuda()
        dout = torch.randn(B * n, inner, generator=g).to(torch.bfloat16).cuda()
        q, k, v = qkv[:, :inner], qkv[:, inner:2 * inner], qkv[:, 2 * inner:]
        o, nlse = K.attention_fwd(q, k, v, B, H, n, dh, dh ** -0.5, with_lse=True)
        dqkv = torch.empty_like(qkv)
        f = lambda: K.attention_bwd(q, k, v, o, dout, nlse, B, H, n, dh, dh ** -0.5, dqkv=dqkv)
        f()
        ms = time_region(f, iters)
        flops = 10.0 * B * H * n * n * dh
        tf = flops / (ms * 1e-3) / 1e12
        return {"bound": "mfma", "achieved": round(tf, 1), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": round(tf / MFMA_BF16_PEAK_TFLOPS, 4), "traffic": None,
                "kernel": "gfe_attention_bwd: attn_bwd_prep + attn_bwd_dkdv + attn_bwd_dq kernels (one layer, B x 8 heads x 1729 x 64; "
                          "S and dP are recomputed in both main kernels: 28 MFMAs executed per 20 algorithmic)",
                "launches_timed": iters, "launch_ms": round(ms, 4), "algorithmic_flops": flops}

    def cpu_baseline(self):
        """autograd through oracle.ref_ops.vit3d (torch CPU restatement of vit_3d.py:113-128) on ONE volume: forward + backward."""
        from oracle import ref_ops as O
        sd = {k: v.detach().float().cpu().clone().requires_grad_(True) for k, v in self.m.state_dict().items()}
        x = self.x[:1].cpu()
        t0 = time.perf_counter()
        out, _ = O.vit3d(x, sd, "", frame_patch=8, patch=8, heads=8, depth=4)
        out.mean().backward()
        dt = time.perf_counter() - t0
        return {"value": round(1.0 / dt, 3), "unit": "volumes/s", "cores": torch.get_num_threads(), "kind": "port",
                "sample": "autograd through oracle.ref_ops.vit3d, 1 volume of 96^3, fp32, torch CPU (forward + backward, no optimizer)"}


class NormWorkload:
    """adaptive_normal (utils/data_normalization.py:20-48) over a batch of native-size 160x160x96 f32 volumes with an
    MRI-like histogram (35 % exact-zero background, skewed positive tissue intensities, some negatives)."""
    name = "adaptive_normal (0.1%/99.9% quantile normalisation), 160x160x96 f32 volumes, synthetic"

    def __init__(self, batch, vol=(160, 160, 96)):
        g = torch.Generator().manual_seed(0)
        x = torch.randn((batch,) + vol, generator=g).abs() ** 3 * 400 - 20
        x[torch.rand((batch,) + vol, generator=g) < 0.35] = 0.0
        self.x = x.cuda()
        self.units = batch
        self.nvox = x[0].numel()

    def step(self):
        from utils.data_normalization import adaptive_normal
        self.y = adaptive_normal(self.x, batched=True, check=False)
        return self.y

    def roofline(self, iters=20):
        self.step()
        t = time_region(self.step, iters)
        alg = 8.0 * self.nvox * self.units                       # read x once + write y once; the select's three extra reads are overhead
        gbs = alg / (t * 1e-3) / 1e9
        return {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": None,
                "kernel": "an_hist x3 + an_select x3 + an_apply (7 launches per batch; 20 B/voxel moved for 8 B/voxel algorithmic)",
                "launch_ms": round(t, 4), "algorithmic_bytes": alg}

    def cpu_baseline(self):
        """oracle/ref_ops.adaptive_normal (the reference's sort-based algorithm, torch CPU) on 2 volumes."""
        from oracle import ref_ops as O
        xs = self.x[:2].cpu()
        t0 = time.perf_counter()
        for v in xs:
            O.adaptive_normal(v.clone())
        dt = time.perf_counter() - t0
        return {"value": round(len(xs) / dt, 3), "unit": "volumes/s", "cores": torch.get_num_threads(), "kind": "port",
                "sample": "oracle/ref_ops.adaptive_normal (torch CPU sort) on 2 volumes of 160x160x96"}


class Gen128Workload:
    """BASELINE config 4: the MRI->PET generator forward alone (main_gan_vit.py:69: `model(condition)`, output_vit_mid=False) on
    volumes of 128^3 (ViT image (256,128), patch 32), batch 2, bf16 activations, deterministic random-init weights."""
    name = "Residual_mid_UNet3D_vit forward (main_gan_vit generator), 128^3, batch 2, synthetic (config 4)"
    GFLOP_PER_VOL = 3214.4          # ~3.21 TFLOP per 128^3 volume (conv 96^3 figures x (128/96)^3 + ViT GEMMs)

    def __init__(self, batch):
        import gfe_hip.det_init as det
        from pytorch3dunet.unet3d.model import Residual_mid_UNet3D_vit
        self.vol = (128, 128, 128)
        gen = Residual_mid_UNet3D_vit(1, 1, is_segmentation=False, f_maps=(64, 128, 256), vol_size=self.vol)
        gen.load_state_dict(det.det_state_dict(gen.state_dict(), seed=32, prefix="gen128."))
        self.gen = gen.cuda().eval()
        self.x = det.det_inputs(batch, self.vol, seed=6)[0].cuda()
        self.batch = self.units = batch

    def step(self):
        with torch.no_grad():
            return self.gen(self.x)

    def roofline(self, iters=10):
        """Dominant kernel: the GroupNorm-folded 27-tap 64->64 conv at 128^3 (three launches per forward)."""
        from gfe_hip import nn_ops as K
        blk = self.gen.encoders[0].basic_module
        with torch.no_grad():
            r = blk.lift(self.x)
            conv = blk.conv2
            w32 = K.pack_conv3(conv.conv.weight, torch.float32)
            g, b = conv.groupnorm.weight.detach().float().contiguous(), conv.groupnorm.bias.detach().float().contiguous()
            ss = K.groupnorm_scale_shift(r, g, b, 8)
            w, tab = K.fold_groupnorm(w32, ss[0], ss[1], K.CONV3_TAPS, 64, 64)
            out = K.conv_igemm(r, w, K.CONV3_TAPS, 64, bias_tab=tab, relu=True)
            run = lambda: K.conv_igemm(r, w, K.CONV3_TAPS, 64, bias_tab=tab, relu=True, out=out)
            for _ in range(3):
                run()
            ms = time_region(run, iters)
        flops = 2.0 * 27 * 64 * 64 * self.batch * 128 ** 3
        tf = flops / (ms * 1e-3) / 1e12
        return {"bound": "mfma", "achieved": round(tf, 1), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(tf / MFMA_BF16_PEAK_TFLOPS, 4),
                "traffic": None, "kernel": "conv_igemm_kernel<4,3,true> (GroupNorm-folded Conv3d 3x3x3 64->64 @128^3, ReLU)",
                "launch_ms": round(ms, 4), "algorithmic_flops": flops}

    def cpu_baseline(self):
        """oracle.ref_ops.generator (torch CPU fp32 restatement of model.py:137-175) on ONE 128^3 volume."""
        from oracle import ref_ops as O
        sd = {k: v.detach().float().cpu() for k, v in self.gen.state_dict().items()}
        x = self.x[:1].cpu()
        t0 = time.perf_counter()
        with torch.no_grad():
            O.generator(x, sd)
        dt = time.perf_counter() - t0
        return {"value": round(1.0 / dt, 4), "unit": "volumes/s", "cores": torch.get_num_threads(), "kind": "port",
                "sample": "oracle.ref_ops.generator, 1 volume of 128^3, fp32, torch CPU"}

    def extra(self):
        return {"generator_gflop_per_volume": self.GFLOP_PER_VOL}


class GenTrainWorkload(Gen128Workload):
    """one generator TRAINING step of main_gan_vit.py:68-82 (L1(model(mri), pet) -> backward -> Adam) on 128^3 volumes,
    train mode (dropout on), bf16 activations / f32 parameter gradients, FlatAdam without clipping."""
    name = "Residual_mid_UNet3D_vit training step (main_gan_vit: L1 + Adam), 128^3, synthetic (row f-1)"

    def __init__(self, batch):
        super().__init__(batch)
        from gfe_hip.train_ops import FlatAdam
        self.gen.train()
        self.opt = FlatAdam([p for p in self.gen.parameters()], lr=1e-4, max_norm=float("inf"))
        self.target = torch.tanh(torch.randn(batch, 1, *self.vol, generator=torch.Generator().manual_seed(7))).cuda()
        self.loss = None

    def step(self):
        from gfe_hip.gen_train import train_step
        self.loss = train_step(self.gen, self.opt, self.x, self.target)

    def roofline(self, iters=5):
        """Dominant kernel of the backward: the fused weight gradient (27 taps x (64 x V) @ (V x 64), reduction over voxels; csrc/conv_wgrad.hip)."""
        from gfe_hip import gen_train as GT
        B = self.batch
        xh = torch.randn(B, 128, 128, 128, 64, device="cuda").to(torch.bfloat16)
        d = torch.randn(B, 128, 128, 128, 64, device="cuda").to(torch.bfloat16)
        run = lambda: GT.conv_wgrad(xh, d, GT.K.CONV3_TAPS)
        run()
        ms = time_region(run, iters)
        flops = 2.0 * 27 * 64 * 64 * B * 128 ** 3
        tf = flops / (ms * 1e-3) / 1e12
        return {"bound": "mfma", "achieved": round(tf, 1), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(tf / MFMA_BF16_PEAK_TFLOPS, 4),
                "traffic": None, "kernel": "conv_wgrad_kernel<7> + sum_splits (Conv3d 3x3x3 64->64 @128^3 weight gradient, all 27 taps in one launch)",
                "launch_ms": round(ms, 4), "algorithmic_flops": flops}

    def cpu_baseline(self):
        """torch CPU fp32 autograd through oracle.ref_ops.generator + L1 on ONE 128^3 volume (forward + backward, no optimizer)."""
        import torch.nn.functional as F
        from oracle import ref_ops as O
        sd = {k: v.detach().float().cpu().clone().requires_grad_(v.dtype.is_floating_point) for k, v in self.gen.state_dict().items()}
        x, tgt = self.x[:1].cpu(), self.target[:1].cpu()
        t0 = time.perf_counter()
        _, _, pet = O.generator(x, sd)
        F.l1_loss(pet, tgt).backward()
        dt = time.perf_counter() - t0
        return {"value": round(1.0 / dt, 4), "unit": "volumes/s", "cores": torch.get_num_threads(), "kind": "port",
                "sample": "autograd through oracle.ref_ops.generator + L1, 1 volume of 128^3, fp32, torch CPU"}

    def extra(self):
        return {"generator_gflop_per_volume_fwd": self.GFLOP_PER_VOL, "l1_loss": None if self.loss is None else round(float(self.loss), 6)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--workload", default=os.environ.get("GFE_BENCH_WORKLOAD", "step"), choices=["step", "scan", "pscan", "vit3d", "vit3dtrain", "normalise", "gen128", "gentrain"])
    ap.add_argument("--volume", default="96", choices=["96", "native"], help="step workload: 96 = 96^3 (BASELINE's metric); native = the reference's own 160x160x96 (config/classify_mamba_config.yaml:5-7; --batch 2 = its train_bc)")
    ap.add_argument("--batch", type=int, default=None, help="per-GPU batch (volumes for `step`, sequences for `scan`; default 8, gen128: 2)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--graph", action="store_true", help="step workload: with --no-pipeline, replay the whole serial step (generator included) from one HIP graph; inside the default two-stream pipeline it replays the head from a graph (already the default for batches of 1-4 volumes and for --gpus N > 1; GFE_NO_AUTO_GRAPH=1 turns that off; DESIGN.md 6)")
    ap.add_argument("--no-pipeline", action="store_true", help="step workload: strictly serial step (generator, then head) on one stream")
    a = ap.parse_args()
    if a.batch is None:
        a.batch = 2 if a.workload in ("gen128", "gentrain") else 1 if a.workload == "pscan" else 8

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # started by torch.distributed.run (any N, ALSO N = 1): the process group, the barrier, the gradient all-reduce on the head stream
    # and the graphed head are all live, so that a one-GPU box exercises exactly the code path the 8-GPU run takes (tests/test_multirank_gpu.py)
    distributed = "WORLD_SIZE" in os.environ
    # stdout carries the ONE JSON line and nothing else: everything printed while the job runs (RCCL announces its version / library path on
    # stdout when the first communicator is created, from C stdio) goes to stderr; the descriptor comes back just before the line is printed
    sys.stdout.flush()
    stdout_fd = os.dup(1)
    os.dup2(2, 1)
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    # GFE_DIST_BACKEND=gloo: dry run of the N-rank code path on a box with fewer GPUs than ranks (ranks share devices round-robin; the
    # gradient all-reduce is staged through host memory, gfe_hip.step.all_reduce_).  RCCL refuses two ranks on one device, so this is how
    # `torchrun --nproc-per-node 2 bench.py --gpus 2` runs on a one-GPU box (tests/test_multirank_gpu.py).  Default and product path: nccl = RCCL.
    backend = os.environ.get("GFE_DIST_BACKEND", "nccl").lower()
    assert backend in ("nccl", "gloo"), backend
    gloo = distributed and backend == "gloo"
    if gloo:
        local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    if distributed:
        torch.set_num_threads(max(1, min(torch.get_num_threads(), (os.cpu_count() or 8) // max(world, 1))))    # N ranks share the host
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # NOT device_id=...: binding the group to the device at init (eager communicator) makes every kernel of the step slower on this
        # stack -- 10.9 -> 12.8-13.0 ms per step with a ONE-rank group and no collective at all (tools/dp_graph_probe.py: PROBE_PG_MODE=eager
        # vs lazy); with the communicator created lazily on the first collective the step time is unchanged.
        dist.init_process_group(backend)
    n_gpus = world
    assert a.gpus == n_gpus or "WORLD_SIZE" in os.environ, "internal: --gpus N > 1 without WORLD_SIZE is started by maybe_self_launch()"
    # per-rank RNG stream for everything drawn on the device during the step (the GEGLU feed-forward's dropout mask,
    # mamba_transformer.py:85): each rank must drop different units of its own shard
    torch.manual_seed(1234 + rank)
    torch.cuda.manual_seed(1234 + rank)

    if a.workload == "gen128":
        wl = Gen128Workload(a.batch)
        steps, warmup = a.steps or 20, a.warmup if a.warmup is not None else 3
        metric, unit, dtype = "MRI volumes/sec (128^3 bf16) MRI->PET generator forward (main_gan_vit, config 4)", "volumes/s", "bf16"
        cfg = {"workload": wl.name, "global_batch": a.batch * n_gpus, "batch_per_gpu": a.batch, "volume": "128x128x128",
               "parallelism": f"replicas x{n_gpus}"}
    elif a.workload == "gentrain":
        wl = GenTrainWorkload(a.batch)
        steps, warmup = a.steps or 10, a.warmup if a.warmup is not None else 2
        metric, unit, dtype = "MRI volumes/sec (128^3 bf16) MRI->PET generator training step (main_gan_vit, L1 + Adam) [row f-1]", "volumes/s", "bf16"
        cfg = {"workload": wl.name, "global_batch": a.batch * n_gpus, "batch_per_gpu": a.batch, "volume": "128x128x128",
               "parallelism": f"replicas x{n_gpus}"}
    elif a.workload == "scan":
        wl = ScanWorkload(a.batch, graph=True if a.graph else None)
        steps, warmup = a.steps or 50, a.warmup if a.warmup is not None else 10
        metric, unit, dtype = "selective-scan tokens/sec (L=4096 ED=1024 N=16 bf16) fwd+bwd", "tokens/s", "bf16"
        cfg = {"workload": wl.name, "batch_per_gpu": a.batch, "parallelism": f"replicas x{n_gpus}", "hip_graph": wl.graph}
    elif a.workload == "pscan":
        wl = PscanWorkload(a.batch)
        steps, warmup = a.steps or 30, a.warmup if a.warmup is not None else 5
        metric, unit, dtype = "pscan tokens/sec (L=4096 D=1024 N=16 bf16, materialised A/X) fwd+bwd [operator boundary 1]", "tokens/s", "bf16"
        cfg = {"workload": wl.name, "batch_per_gpu": wl.B, "parallelism": f"replicas x{n_gpus}"}
    elif a.workload == "vit3d":
        wl = Vit3dWorkload(a.batch)
        steps, warmup = a.steps or 20, a.warmup if a.warmup is not None else 5
        metric, unit, dtype = "3-D ViT volumes/sec (96^3, 1729 tokens, bf16) forward [synthetic MFMA-attention row]", "volumes/s", "bf16"
        cfg = {"workload": wl.name, "batch_per_gpu": a.batch, "parallelism": f"replicas x{n_gpus}"}
    elif a.workload == "vit3dtrain":
        wl = Vit3dTrainWorkload(a.batch)
        steps, warmup = a.steps or 10, a.warmup if a.warmup is not None else 3
        metric, unit, dtype = "3-D ViT volumes/sec (96^3, 1729 tokens, bf16) training step fwd+bwd+Adam [synthetic MFMA-attention row]", "volumes/s", "bf16"
        cfg = {"workload": wl.name, "batch_per_gpu": a.batch, "parallelism": f"replicas x{n_gpus}"}
    elif a.workload == "normalise":
        wl = NormWorkload(a.batch)
        steps, warmup = a.steps or 50, a.warmup if a.warmup is not None else 10
        metric, unit, dtype = "MRI volumes/sec normalised (adaptive_normal, 160x160x96 f32) [input-pipeline row]", "volumes/s", "f32"
        cfg = {"workload": wl.name, "batch_per_gpu": a.batch, "parallelism": f"replicas x{n_gpus}"}
    else:
        from gfe_hip.step_bench import StepWorkload
        vol = (160, 160, 96) if a.volume == "native" else (96, 96, 96)
        wl = StepWorkload(a.batch, world=world, rank=rank, vol=vol, graph=a.graph, pipeline=not a.no_pipeline, distributed=distributed)
        steps, warmup = a.steps or 40, a.warmup if a.warmup is not None else 10     # (a fresh box needs a few steps before clocks / page-ins settle)
        metric, unit, dtype = "MRI volumes/sec (%s bf16) classify_mamba fwd+bwd" % ("96^3" if a.volume == "96" else "160x160x96"), "volumes/s", "bf16"
        cfg = {"workload": wl.name, "global_batch": a.batch * n_gpus, "batch_per_gpu": a.batch, "volume": "x".join(map(str, vol)),
               "parallelism": f"dp{n_gpus}", "dist_backend": (backend if distributed else None), "hip_graph": bool(a.graph or getattr(wl, "graph_head", False)),
               "pipeline": ("generator(batch k+1) || head(batch k), 2 streams" + (", head replayed from a HIP graph" if getattr(wl, "graph_head", False) else "")) if wl.pipeline else "none",
               "dp": wl.step_obj.dp_modes()}       # how the multi-rank switches resolved (GFE_OVERLAP_UPDATE / GFE_DP_BUCKETS / GFE_NO_AUTO_GRAPH override)

    def barrier():
        if distributed:
            from gfe_hip.step import barrier as _barrier
            _barrier(local)

    for _ in range(warmup):
        wl.step()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        wl.step()
    torch.cuda.synchronize()
    el_own = time.perf_counter() - t0           # this rank's own K steps (before it waits for the slowest rank)
    barrier()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    per_rank = None
    if distributed:
        cdev = "cpu" if gloo else "cuda"
        t = torch.tensor([el], device=cdev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = t.item()
        own = torch.tensor([el_own / steps * 1e3], device=cdev, dtype=torch.float64)
        allr = [torch.zeros_like(own) for _ in range(world)]
        dist.all_gather(allr, own)
        per_rank = [round(v.item(), 4) for v in allr]       # a host-bound or straggling rank shows up here

    roof = wl.roofline() if rank == 0 else None              # one GPU's kernel figure; ranks > 0 go straight to the collective's barrier
    comm = wl.allreduce_stats(local) if distributed and hasattr(wl, "allreduce_stats") else None
    cpu = None
    if rank == 0 and n_gpus == 1 and not a.no_cpu_baseline:
        cpu = wl.cpu_baseline()
    if rank == 0:
        ms = el / steps * 1e3
        out = {"metric": metric, "value": round(wl.units * n_gpus / (el / steps), 3), "unit": unit, "n_gpus": n_gpus,
               "steps": steps, "warmup": warmup, "ms_per_step": round(ms, 4), "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": dtype, "data": "synthetic", "config": cfg, "roofline": roof, "cpu_baseline": cpu}
        extra = getattr(wl, "extra", None)
        if extra:
            out["extra"] = extra() if callable(extra) else extra
        if comm:
            out["allreduce"] = comm
        if per_rank is not None:
            out["ms_per_step_by_rank"] = {"min": min(per_rank), "max": max(per_rank), "ranks": per_rank}
        import ctypes
        sys.stdout.flush()
        ctypes.CDLL(None).fflush(None)                # C-side buffers (RCCL's banner) leave through stderr too
        os.dup2(stdout_fd, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)                                 # (anything the teardown prints is not part of the contract either)
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
