#!/bin/bash
# Kernel-time (rocprofv3) comparison of chunk lengths for the N = 16 scan at one batch size: tools/scan_chunk_prof.sh B "chunks..."
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02/chunkprof; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
B=$1
for c in $2; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/b${B}_c$c -o s -- python3 $R/tools/scan_one.py $B $c > /dev/null 2>&1
  python3 - <<PY
import csv
rows = list(csv.DictReader(open("$O/b${B}_c$c/s_kernel_stats.csv")))
tot = {}
for r in rows:
    if "sscan2" in r["Name"]:
        k = r["Name"][r["Name"].index("sscan2"):][:44]
        tot[k] = float(r["AverageNs"]) / 1e3
print("B=$B chunk=$c:", " ".join("%s %.1f" % (k.replace("unsigned short", "bf16"), v) for k, v in sorted(tot.items())), "| sum %.1f us" % sum(tot.values()))
PY
done
