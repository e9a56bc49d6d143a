#!/bin/bash
# Runs on the GPU box: kernel stats + SQ issue/stall counters of the scan bench.  usage: tools/pmc_scan.sh OUTDIR BATCH
O=$GRAFT_REPO_ROOT/gpurun_out/$1; B=${2:-8}
R=$GRAFT_REPO_ROOT
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_scan -o scan -- python3 $R/bench.py --workload scan --batch $B --no-cpu-baseline --steps 20 --warmup 5 > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS --kernel-trace --output-format csv -d $O/pmc_sq -o p -- python3 $R/bench.py --workload scan --batch $B --no-cpu-baseline --steps 4 --warmup 2 > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/pmc_sq2 -o p -- python3 $R/bench.py --workload scan --batch $B --no-cpu-baseline --steps 4 --warmup 2 > /dev/null 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_grbm -o p -- python3 $R/bench.py --workload scan --batch $B --no-cpu-baseline --steps 4 --warmup 2 > /dev/null 2>&1
cd $R
python3 - <<PY
import csv, glob, collections, os
O = "$O"
out = open(os.path.join(O, "pmc_summary.txt"), "w")
f = glob.glob(O + "/prof_scan/**/*kernel_stats.csv", recursive=True)
if f:
    for r in list(csv.DictReader(open(f[0])))[:12]:
        line = "%-90s calls %5d avg_us %9.1f" % (r["Name"][:90], int(r["Calls"]), float(r["AverageNs"]) / 1e3)
        print(line); out.write(line + "\n")
for d in sorted(glob.glob(O + "/pmc_*")):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    if not f: continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list)); dur = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"]
        if "sscan" not in k: continue
        k = k.split("(")[0][-50:]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"])); dur[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    for k in agg:
        line = f"{os.path.basename(d):10s} {k:52s} avg_us {sum(dur[k]) / len(dur[k]) / 1e3:8.1f} " + " ".join(f"{c}={sum(v) / len(v):.4g}" for c, v in agg[k].items())
        print(line); out.write(line + "\n")
out.close()
PY
