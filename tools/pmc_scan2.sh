#!/bin/bash
# On the GPU box: SQ issue/stall/LDS counters of the scan kernels, forward and backward apart.  usage: tools/pmc_scan2.sh OUTDIR [LIB] [BATCH]
O=$GRAFT_REPO_ROOT/gpurun_out/$1; LIB=$2; B=${3:-8}
R=$GRAFT_REPO_ROOT
mkdir -p $O
if [ -n "$LIB" ]; then export GFE_HIP_LIB=$R/$LIB; fi
cd /tmp && export TMPDIR=/tmp
P="python3 $R/tools/scan_exp/time_scan.py $B 4"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS --kernel-trace --output-format csv -d $O/pmc_sq -o p -- $P > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/pmc_sq2 -o p -- $P > /dev/null 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU_TRANS SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_LEVEL_LDS SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS --kernel-trace --output-format csv -d $O/pmc_sq3 -o p -- $P > /dev/null 2>&1
cd $R
python3 - <<PY
import csv, glob, collections, os
O = "$O"
out = open(os.path.join(O, "pmc_summary.txt"), "w")
for d in sorted(glob.glob(O + "/pmc_*")):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    if not f: continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list)); dur = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"]
        if "sscan2_" not in k: continue
        k = ("fwd" if "sscan2_fwd" in k else "bwd" if "sscan2_bwd" in k else "fold") + ("_state" if "true" in k.split("(")[0].split(",")[-2 if "bwd" in k else -1] else "")
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"])); dur[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    for k in sorted(agg):
        line = f"{os.path.basename(d):8s} {k:10s} avg_us {sum(dur[k]) / len(dur[k]) / 1e3:8.1f} " + " ".join(f"{c}={sum(v) / len(v):.4g}" for c, v in sorted(agg[k].items()))
        print(line); out.write(line + "\n")
out.close()
PY
