#!/usr/bin/env python3
"""Summaries of one tools/collect_profiles_r06.sh run: scan / attention / conv HBM traffic from the separate --pmc passes (FETCH_SIZE x2 on
gfx950 + WRITE_SIZE), the roofline leg's launches inside the step trace, traffic_r06.json with the kernel sources' hashes.
    python tools/summarise_profiles_r06.py <gpurun_out/r06> <repo root>"""
import csv, glob, collections, os, json
import sys
O, R = sys.argv[1], sys.argv[2]
out = open(os.path.join(O, "scan_traffic.txt"), "w")
res = {}
for d, scale in (("pmc_scan_fetch", 2.0), ("pmc_scan_write", 1.0)):      # gfx950: FETCH_SIZE reports half of wide streaming reads (MI355X_MICROARCH.md)
    f = glob.glob(O + "/" + d + "/**/*counter_collection.csv", recursive=True)
    if not f: continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"]
        if "sscan2" not in k: continue
        agg[k[k.index("sscan2"):][:40]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        kb = sum(v) / len(v) * scale            # counters are in KB
        res.setdefault(k, {})[d] = kb * 1024
        line = f"{d:16s} {k:42s} per launch {kb * 1024 / 1e6:9.1f} MB (x{scale:g} applied)"
        print(line); out.write(line + "\n")
out.close()
json.dump(res, open(os.path.join(O, "scan_traffic.json"), "w"), indent=1)
tot = sum(sum(v.values()) for v in res.values())
attn = {}
for d, scale in (("pmc_attn_fetch", 2.0), ("pmc_attn_write", 1.0)):
    f = glob.glob(O + "/" + d + "/**/*counter_collection.csv", recursive=True)
    if not f: continue
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f[0])) if "attn_fwd" in r["Kernel_Name"]]
    if v: attn[d] = sum(v) / len(v) * scale * 1024
attnb = {}
for kern in ("attn_bwd_dkdv", "attn_bwd_dq", "attn_bwd_prep"):
    for d, scale in (("pmc_attn_fetch", 2.0), ("pmc_attn_write", 1.0)):
        f = glob.glob(O + "/" + d + "/**/*counter_collection.csv", recursive=True)
        if not f: continue
        v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f[0])) if kern in r["Kernel_Name"]]
        if v: attnb.setdefault(kern, {})[d] = sum(v) / len(v) * scale * 1024
conv = {}
for d, scale in (("pmc_conv_fetch", 2.0), ("pmc_conv_write", 1.0)):
    f = glob.glob(O + "/" + d + "/**/*counter_collection.csv", recursive=True)
    if not f: continue
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f[0])) if "conv_igemm_kernel<4, 3, true, false" in r["Kernel_Name"]]
    if v: conv[d] = sum(v) / len(v) * scale * 1024
extra = {}
if len(conv) == 2:
    extra["conv_igemm_64to64_96cubed_b8"] = {"traffic_bytes": sum(conv.values()), "fetch_x2_bytes": conv["pmc_conv_fetch"], "write_bytes": conv["pmc_conv_write"],
                                              "method": "rocprofv3 --pmc FETCH_SIZE (x2) / WRITE_SIZE, separate passes, mean per launch of conv_igemm_kernel<4,3,true> in tools/conv_bench.py 64 96 8"}
    print("conv 64->64 @96^3 B=8 traffic per launch: fetch %.2f GB (x2 applied) + write %.2f GB" % (conv["pmc_conv_fetch"] / 1e9, conv["pmc_conv_write"] / 1e9))
if len(attn) == 2:
    extra["attn_fwd_b8_h8_n1729"] = {"traffic_bytes": sum(attn.values()), "fetch_x2_bytes": attn["pmc_attn_fetch"], "write_bytes": attn["pmc_attn_write"],
                                     "method": "rocprofv3 --pmc FETCH_SIZE (x2) / WRITE_SIZE, separate passes, mean per launch of attn_fwd_kernel in tools/attn_bench.py 8 8 1729"}
    print("attention B=8 H=8 n=1729 traffic per launch: fetch %.1f MB (x2 applied) + write %.1f MB" % (attn["pmc_attn_fetch"] / 1e6, attn["pmc_attn_write"] / 1e6))
if attnb:
    extra["attn_bwd_b8_h8_n1729"] = {"traffic_bytes": sum(sum(v.values()) for v in attnb.values()), "per_kernel": attnb,
                                     "method": "rocprofv3 --pmc FETCH_SIZE (x2) / WRITE_SIZE, separate passes, mean per launch of the three gfe_attention_bwd kernels in tools/attn_bench.py 8 8 1729"}
    print("attention backward traffic per launch:", {k: {kk: round(vv / 1e6, 1) for kk, vv in v.items()} for k, v in attnb.items()})
# the roofline leg's own launches inside the step profile: the bench line's roofline.launch_ms must agree with rocprofv3's view of them
tr = glob.glob(O + "/prof_step/**/*kernel_trace.csv", recursive=True)
if tr:
    d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(tr[0])) if "conv_igemm_kernel<4, 3, true, false, false, false, false>" in r["Kernel_Name"]]
    last = d[-50:]
    if last:
        mean = sum(last) / len(last) / 1e3
        with open(os.path.join(O, "step_b8_roofline_launches.txt"), "w") as fh:
            fh.write("rocprofv3 --kernel-trace of python bench.py --no-cpu-baseline --steps 10 --warmup 3 (profiles/r06/step_b8_kernel_stats.csv):\n")
            fh.write("conv_igemm_kernel<4,3,true,false,...> launches in the process: %d; the roofline leg = the last 50 of them:\n" % len(d))
            fh.write("  mean %.1f us, min %.1f, max %.1f  ->  %.1f TFLOP/s = %.4f of 2500 (algorithmic 1565.5 GFLOP per launch)\n" % (
                mean, min(last) / 1e3, max(last) / 1e3, 1565.515579392 / mean * 1e3, 1565.515579392 / mean * 1e3 / 2500))
import hashlib
srcs = {k: hashlib.sha256(open(os.path.join(R, "gfe-mamba_amd", "csrc", k), "rb").read()).hexdigest() for k in ("conv3d.hip", "attn.hip", "attn_bwd.hip", "sscan2.hip")}
extra["kernel_sources"] = {"sha256": srcs, "note": "the kernels these counters were taken on: gfe_hip.step_bench.measured_traffic() returns None (and tests/test_abi.py fails) once a source differs"}
json.dump({**extra, "scan_b8": {"traffic_bytes": tot, "per_kernel": res,
                       "method": "rocprofv3 --pmc FETCH_SIZE (x2: gfx950 reports half of coalesced reads at 4, 8 and 16 B per lane, tools/probes/fetch_calib.hip) and --pmc WRITE_SIZE, separate passes, mean per launch of sscan2_fwd + sscan2_bwd at B=8 L=4096 ED=1024 N=16 bf16"}},
          open(os.path.join(O, "traffic_r06.json"), "w"), indent=1)
