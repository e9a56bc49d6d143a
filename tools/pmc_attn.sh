#!/bin/bash
# Runs on the GPU box: SQ issue / stall counters of the flash-attention kernels.  usage: tools/pmc_attn.sh OUTDIR [kernel-name filter ...]
O=$GRAFT_REPO_ROOT/gpurun_out/$1; R=$GRAFT_REPO_ROOT; shift; KERNELS="${*:-attn_fwd}"
mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS --kernel-trace --output-format csv -d $O/pmc_a -o p -- python3 $R/tools/attn_bench.py 8 8 1729 10 > /dev/null 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_WAVES --kernel-trace --output-format csv -d $O/pmc_b -o p -- python3 $R/tools/attn_bench.py 8 8 1729 10 > /dev/null 2>&1
cd $R
python3 - <<PY
import csv, glob, collections, os
O = "$O"
for kern in "$KERNELS".split():
  for d in sorted(glob.glob(O + "/pmc_*")):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    if not f: continue
    agg = collections.defaultdict(list); dur = []
    for r in csv.DictReader(open(f[0])):
        if kern not in r["Kernel_Name"]: continue
        agg[r["Counter_Name"]].append(float(r["Counter_Value"])); dur.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    print(kern, os.path.basename(d), "avg_us %.1f" % (sum(dur) / max(1, len(dur)) / 1e3), " ".join(f"{c}={sum(v)/len(v):.4g}" for c, v in agg.items()))
PY
