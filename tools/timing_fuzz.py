#!/usr/bin/env python3
"""Timing-fuzz harness for the generator's hand-synchronised kernels (the batch-8 vs batch-1 mismatch that one
box of the pool showed twice and 18 others never).

The conv / transposed-conv / one-channel conv / LDS-DMA GEMM kernels order their LDS-DMA transfers, LDS reads, ticket hand-offs and stores
with hand-counted `s_waitcnt vmcnt` waits and raw `s_barrier`s.  If one of those orders is wrong, the result depends on the relative timing
of a block's waves -- which a normal run on a normal box never varies.  `make -C gfe-mamba_amd/csrc fuzz` builds the same sources with
-DGFE_TIMING_FUZZ: every wait, barrier, DMA burst and ticket access is preceded by a per-wave pseudo-random `s_sleep` (common.h), so
every launch sees another interleaving of its waves, including ones no box's clocks would produce.

    python tools/timing_fuzz.py [--iters 200] [--vol 96] [--env K=V ...]

1. child "ref": the PRODUCT library runs the generator at batch 8 and at batch 1 for each of the 8 samples and stores bit-exact
   fingerprints of every stage output (tools/gen_stages.py: one 64-bit word per conv tile and sample); it also asserts batch 8 == batch 1.
2. child "fuzz": GFE_HIP_LIB = the fuzz library; --iters times: batch 8, then batch 1 of sample (i mod 8); every stage of every run must
   reproduce the reference fingerprints bit for bit.  A mismatch prints the first diverging stage in network order and the tiles, and the
   run goes on (the count and the distinct stages are what localises a race).  --env switches (GFE_CONV_STATIC=1, GFE_CONV_BRICK=0,
   GFE_GEMM_NO_DMA=1, GFE_CONVT_STREAMED=1) apply to both children, for bisecting.
--workload vit3d: the other hand-synchronised kernels (flash attention forward / dK-dV / dQ with their LDS-DMA rings and counted waits, the
LDS-DMA GEMM under the 3-D ViT's projections, the fused conv weight gradient): the synthetic vit_3d.ViT forward, its training forward +
backward (logits, dx, every parameter gradient) and one gfe_conv3d_wgrad launch, fingerprinted per 4096-word chunk -- all of them are
deterministic kernels (no atomics), so every fuzzed run must again be bit-identical.
Exit code 0 = every fuzzed run bit-identical to the un-fuzzed library."""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FUZZ_LIB = os.path.join(ROOT, "gfe-mamba_amd", "gfe_hip", "libgfe_hip_fuzz.so")


def child(role, args):
    sys.path.insert(0, os.path.join(ROOT, "gfe-mamba_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
    import torch
    import gfe_hip
    from gfe_hip.step import build_models
    import gfe_hip.det_init as det
    import gen_stages as G
    vol = (args.vol,) * 3
    lib = os.path.basename(gfe_hip.LIB_PATH)
    if args.workload == "vit3d":
        return child_vit3d(role, args, lib, torch, G, det)
    if args.workload == "scan":
        return child_scan(role, args, lib, torch, G, det)
    gen, _, _ = build_models(vol=vol, seed=0)
    x = det.det_inputs(8, vol, seed=77)[0].cuda()

    def run(xb):
        st = G.staged_forward(gen, xb)
        fp = G.fingerprints(st)
        del st
        return fp

    if role == "ref":
        assert "fuzz" not in lib, lib
        ref8 = run(x)
        ref1 = [run(x[b:b + 1]) for b in range(8)]
        bad = []
        for b in range(8):
            d = G.first_divergence({k: v[b:b + 1] for k, v in ref8.items()}, ref1[b])
            if d:
                bad.append((b, d))
        again = G.first_divergence(ref8, run(x))
        torch.save({"ref8": ref8, "ref1": ref1}, args.ref)
        print(json.dumps({"role": "ref", "lib": lib, "device": G.device_report(), "batch8_vs_batch1_mismatches": [(b, d[0], d[2]) for b, d in bad],
                          "batch8_repeat_mismatch": again and (again[0], again[2])}), flush=True)
        return 1 if (bad or again) else 0

    assert "fuzz" in lib, "the fuzz child must load the fuzz library (GFE_HIP_LIB): " + lib
    ref = torch.load(args.ref)
    ref8, ref1 = ref["ref8"], ref["ref1"]
    fails, t0, ms8, ms1 = [], time.time(), [], []
    for it in range(args.iters):
        for batch in (8, 1):
            b = it % 8
            torch.cuda.synchronize(); t = time.time()
            fp = run(x if batch == 8 else x[b:b + 1])
            torch.cuda.synchronize(); (ms8 if batch == 8 else ms1).append((time.time() - t) * 1e3)
            d = G.first_divergence(ref8 if batch == 8 else ref1[b], fp)
            if d:
                fails.append({"iter": it, "batch": batch, "sample": None if batch == 8 else b, "stage": d[0], "cells": d[1], "ncells": d[2]})
                print("MISMATCH", json.dumps(fails[-1]), flush=True)
    ms8.sort(); ms1.sort()
    print(json.dumps({"role": "fuzz", "lib": lib, "iters": args.iters, "runs": 2 * args.iters, "mismatching_runs": len(fails),
                      "stages": sorted({f["stage"] for f in fails}), "seconds": round(time.time() - t0, 1),
                      "median_ms_incl_fingerprints": {"batch8": round(ms8[len(ms8) // 2], 1), "batch1": round(ms1[len(ms1) // 2], 1)}}), flush=True)
    return 1 if fails else 0


def child_scan(role, args, lib, torch, G, det):
    """--workload scan (round 6): the fused selective scan's forward and backward (csrc/sscan2.hip: 4 scan + 4 staging waves per block, a ring of
    three input tiles read across the tile barrier, per-pair partial rows handed over through LDS, a counted lgkmcnt wait in front of the
    barrier) at four shapes -- one launch each way (B = 8, L = 300), the chunked two-pass plan (B = 1, L = 1000, chunk 256), f32 I/O with ragged
    L = 77 and config 2's own length (L = 4096, 128 tiles per block) -- with the fixed-order accumulation (GFE_SCAN_DETERMINISTIC=1): y and all eight gradients must come out bit-identical."""
    os.environ["GFE_SCAN_DETERMINISTIC"] = "1"
    from gfe_hip.scan_ops import selective_scan_tm
    g = torch.Generator().manual_seed(3)
    cases = []
    for (B, L, ED, dt_, chunk) in ((8, 300, 128, torch.bfloat16, 0), (1, 1000, 256, torch.bfloat16, 256), (2, 77, 64, torch.float32, 0), (2, 4096, 256, torch.bfloat16, 0)):
        mk = lambda *s_, sc=1.0: (torch.randn(*s_, generator=g) * sc).to(dt_).cuda().requires_grad_(True)
        u, d, z, Bm, Cm = mk(B, L, ED), mk(B, L, ED, sc=0.3), mk(B, L, ED), mk(B, L, 16), mk(B, L, 16)
        A = (-(torch.rand(ED, 16, generator=g) * 6 + 0.2)).cuda().requires_grad_(True)
        D = torch.randn(ED, generator=g).cuda().requires_grad_(True)
        bias = (torch.randn(ED, generator=g) - 3).cuda().requires_grad_(True)
        dy = torch.randn(B, L, ED, generator=g).to(dt_).cuda()
        cases.append(((u, d, A, Bm, Cm, D, z, bias), dy, chunk))

    def run():
        out = {}
        for i, (ins, dy, chunk) in enumerate(cases):
            for t in ins:
                t.grad = None
            u, d, A, Bm, Cm, D, z, bias = ins
            y = selective_scan_tm(u, d, A, Bm, Cm, D, z=z, delta_bias=bias, delta_softplus=True, chunk=chunk)
            y.backward(dy)
            out["y%d" % i] = y.detach()
            for n, t in zip(("du", "ddelta", "dA", "dB", "dC", "dD", "dz", "dbias"), ins):
                out["%s%d" % (n, i)] = t.grad
        return {k_: G.fingerprint(v_.detach().float().reshape(1, -1)).cpu() for k_, v_ in out.items()}

    if role == "ref":
        assert "fuzz" not in lib, lib
        ref, again = run(), run()
        order = tuple(ref.keys())
        d = G.first_divergence(ref, again, order)
        torch.save({"ref": ref}, args.ref)
        print(json.dumps({"role": "ref", "workload": "scan", "lib": lib, "device": G.device_report(), "tensors": len(ref), "repeat_mismatch": d and (d[0], d[2])}), flush=True)
        return 1 if d else 0
    assert "fuzz" in lib, "the fuzz child must load the fuzz library (GFE_HIP_LIB): " + lib
    ref = torch.load(args.ref)["ref"]
    order = tuple(ref.keys())
    fails, t0 = [], time.time()
    for it in range(args.iters):
        d = G.first_divergence(ref, run(), order)
        if d:
            fails.append({"iter": it, "tensor": d[0], "cells": d[1], "ncells": d[2]})
            print("MISMATCH", json.dumps(fails[-1]), flush=True)
    print(json.dumps({"role": "fuzz", "workload": "scan", "lib": lib, "iters": args.iters, "mismatching_runs": len(fails),
                      "tensors": sorted({f["tensor"] for f in fails}), "seconds": round(time.time() - t0, 1)}), flush=True)
    return 1 if fails else 0


def child_vit3d(role, args, lib, torch, G, det):
    from vit_pytorch_diy.vit_3d import ViT
    from gfe_hip.gen_train import conv_wgrad
    from gfe_hip import nn_ops as K
    B = 4
    m = ViT(image_size=96, image_patch_size=8, frames=96, frame_patch_size=8, channels=1, dim=512, depth=4, heads=8, dim_head=64, mlp_dim=2048, num_classes=1)
    m.load_state_dict(det.det_state_dict(m.state_dict(), seed=21, prefix="vit3d."))
    m = m.cuda()
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B, 1, 96, 96, 96, generator=g).clamp_(-1, 1).cuda()
    wi = torch.randn(2, 32, 32, 32, 64, generator=g).to(torch.bfloat16).cuda()
    wd = torch.randn(2, 32, 32, 32, 64, generator=g).to(torch.bfloat16).cuda()

    def run():
        out = {}
        m.eval()
        with torch.no_grad():
            out["inference_logits"] = m(x)
        m.train()                                               # (dropout 0: the training twin = flash attention with the row statistic + backward)
        for p_ in m.parameters():
            p_.grad = None
        xi = x.clone().requires_grad_(True)
        y = m(xi)
        (y * torch.linspace(0.5, 1.5, y.numel(), device=y.device).view_as(y)).sum().backward()
        out["train_logits"], out["dx"] = y.detach(), xi.grad
        for k_, p_ in m.named_parameters():
            out["grad." + k_] = p_.grad
        out["conv_wgrad"] = conv_wgrad(wi, wd, K.CONV3_TAPS)
        return {k_: G.fingerprint(v_.detach().float().reshape(1, -1)).cpu() for k_, v_ in out.items()}

    order = None
    if role == "ref":
        assert "fuzz" not in lib, lib
        ref = run()
        again = run()
        order = tuple(ref.keys())
        d = G.first_divergence(ref, again, order)
        torch.save({"ref": ref}, args.ref)
        print(json.dumps({"role": "ref", "workload": "vit3d", "lib": lib, "device": G.device_report(), "tensors": len(ref), "repeat_mismatch": d and (d[0], d[2])}), flush=True)
        return 1 if d else 0
    assert "fuzz" in lib, "the fuzz child must load the fuzz library (GFE_HIP_LIB): " + lib
    ref = torch.load(args.ref)["ref"]
    order = tuple(ref.keys())
    fails, t0 = [], time.time()
    for it in range(args.iters):
        d = G.first_divergence(ref, run(), order)
        if d:
            fails.append({"iter": it, "tensor": d[0], "cells": d[1], "ncells": d[2]})
            print("MISMATCH", json.dumps(fails[-1]), flush=True)
    print(json.dumps({"role": "fuzz", "workload": "vit3d", "lib": lib, "iters": args.iters, "mismatching_runs": len(fails),
                      "tensors": sorted({f["tensor"] for f in fails}), "seconds": round(time.time() - t0, 1)}), flush=True)
    return 1 if fails else 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--vol", type=int, default=96)
    ap.add_argument("--env", action="append", default=[], help="K=V for both children (bisect switches)")
    ap.add_argument("--ref", default=os.path.join(ROOT, "gpurun_out", "timing_fuzz_ref.pt"))
    ap.add_argument("--role", default=None)
    ap.add_argument("--workload", default="generator", choices=["generator", "vit3d", "scan"])
    ap.add_argument("--heavy", action="store_true", help="the heavier fuzz level (make fuzz_heavy: every other site visit sleeps, long sleeps at 1 of 32)")
    args = ap.parse_args()
    if args.role:
        sys.exit(child(args.role, args))
    # always through make (incremental): a fuzz library older than the sources fails to load as soon as the header gains a symbol
    subprocess.check_call(["make", "-s", "-j", "8", "-C", os.path.join(ROOT, "gfe-mamba_amd", "csrc"), "fuzz_heavy" if args.heavy else "fuzz"])
    os.makedirs(os.path.dirname(args.ref), exist_ok=True)
    env = dict(os.environ)
    for kv in args.env:
        k, v = kv.split("=", 1)
        env[k] = v
    base = [sys.executable, os.path.abspath(__file__), "--iters", str(args.iters), "--vol", str(args.vol), "--ref", args.ref, "--workload", args.workload]
    env.pop("GFE_HIP_LIB", None)
    rc_ref = subprocess.call(base + ["--role", "ref"], env=env)                  # children, never an exec of a GPU-initialised process
    env["GFE_HIP_LIB"] = FUZZ_LIB.replace("_fuzz.so", "_fuzz_heavy.so") if args.heavy else FUZZ_LIB
    rc_fuzz = subprocess.call(base + ["--role", "fuzz"], env=env)
    print("timing_fuzz [%s]: ref rc %d (the product library repeats itself / batch 8 == batch 1), fuzz rc %d (%s)" % (args.workload, 
        rc_ref, rc_fuzz, "every fuzzed run bit-identical" if rc_fuzz == 0 else "MISMATCHES, see above"))
    sys.exit(1 if (rc_ref or rc_fuzz) else 0)


if __name__ == "__main__":
    main()
