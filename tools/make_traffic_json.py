#!/usr/bin/env python3
"""profiles/r01/pmc_summary_TAG.txt (tools/collect_profiles.sh) -> profiles/r01/traffic_TAG.json, the per-launch HBM-side bytes bench.py
reports as roofline.traffic.     python tools/make_traffic_json.py v6
FETCH_SIZE / WRITE_SIZE are in KiB; FETCH_SIZE is doubled on gfx950 for 16-B-per-lane streaming reads (MI355X_MICROARCH.md, HBM section)."""
import json, os, re, sys
tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rows = {}
for line in open(os.path.join(root, "profiles", "r01", f"pmc_summary_{tag}.txt")):
    m = re.match(r"(pmc_\w+)\s.*?(FETCH_SIZE|WRITE_SIZE)=([0-9.e+]+)", line)
    if m:
        rows[m.group(1)] = float(m.group(3)) * 1024.0
def entry(kind, alg=None, note=None, mult=1):
    f, w = 2.0 * rows[f"pmc_{kind}_fetch"] * mult, rows[f"pmc_{kind}_write"] * mult
    e = dict(fetch_bytes=f, write_bytes=w, traffic_bytes=f + w)
    if alg is not None:
        e["algorithmic_bytes"] = alg
    if note:
        e["note"] = note
    return e
out = {
    "source": f"profiles/r01/pmc_summary_{tag}.txt (rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE, separate passes, kernel-trace only)",
    "correction": "FETCH_SIZE x2 on gfx950 for 16-B-per-lane streaming reads (MI355X_MICROARCH.md, HBM); WRITE_SIZE as reported; counters are L2 memory-side requests (Infinity-Cache hits included)",
    "conv_igemm_64to64_96cubed_b8": entry("conv", 1811939328),
    "attn_fwd_b8_h8_n1729": entry("attn", 56655872),
    "sscan_all_kernels_b8_avg_per_launch": entry("scan"),
    "sscan_fwd_bwd_step_b8": entry("scan", 744710144, "6 launches per fwd+bwd step (state pass, carry, full pass; adjoint state pass, carry, adjoint) x the per-launch averages above", 6),
}
json.dump(out, open(os.path.join(root, "profiles", "r01", f"traffic_{tag}.json"), "w"), indent=1)
print(json.dumps({k: v.get("traffic_bytes") for k, v in out.items() if isinstance(v, dict)}))
