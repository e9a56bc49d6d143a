#!/bin/bash
# Runs on the GPU box: SQ issue / stall / LDS counters of the bf16 GEMM main loops.  usage: tools/pmc_gemm.sh OUTDIR [extra env assignments...]
O=$GRAFT_REPO_ROOT/gpurun_out/$1; R=$GRAFT_REPO_ROOT; shift
mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for e in "$@"; do export "$e"; done
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS --kernel-trace --output-format csv -d $O/pmc_a -o p -- python3 $R/tools/gemm_bench.py 1 5 > /dev/null 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_WAVES --kernel-trace --output-format csv -d $O/pmc_b -o p -- python3 $R/tools/gemm_bench.py 1 5 > /dev/null 2>&1
cd $R
python3 - <<PY
import csv, glob, collections, os
O = "$O"
for d in sorted(glob.glob(O + "/pmc_*")):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    if not f: continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list)); dur = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"]
        if "gemm" not in k: continue
        key = (("dma" if "gemm_dma" in k else "old") + " grid=" + r.get("Grid_Size", "?"))
        agg[key][r["Counter_Name"]].append(float(r["Counter_Value"])); dur[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    for key in sorted(agg):
        print(os.path.basename(d), key, "avg_us %.1f" % (sum(dur[key]) / max(1, len(dur[key])) / 1e3), " ".join(f"{c}={sum(v)/len(v):.4g}" for c, v in agg[key].items()))
PY
