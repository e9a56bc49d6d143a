"""Summarises rocprofv3 --pmc passes of the scan kernels: tools/pmc_summ.py gpurun_out/DIR -> per kernel role averages."""
import csv, glob, collections, os, sys
O = sys.argv[1]
lines = []
for d in sorted(glob.glob(O + "/pmc_*")):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    if not f: continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list)); dur = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"]
        if "sscan2_" not in k: continue
        targs = k.split("<")[1].split(">")[0] if "<" in k else ""
        role = "fwd" if "sscan2_fwd" in k else "bwd" if "sscan2_bwd" in k else "fold"
        a = [x.strip() for x in targs.split(",")]
        if role in ("fwd", "bwd") and len(a) >= 3 and a[2] == "true": role += "_state"
        agg[role][r["Counter_Name"]].append(float(r["Counter_Value"])); dur[role].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    for k in sorted(agg):
        lines.append(f"{os.path.basename(d):8s} {k:10s} avg_us {sum(dur[k]) / len(dur[k]) / 1e3:8.1f} " + " ".join(f"{c}={sum(v) / len(v):.4g}" for c, v in sorted(agg[k].items())))
print("\n".join(lines))
open(os.path.join(O, "pmc_summary.txt"), "w").write("\n".join(lines) + "\n")
