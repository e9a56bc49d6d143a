#!/bin/bash
# On the GPU box: the clock a kernel actually runs at, and how busy its matrix pipe is.  usage: tools/pmc_conv_clock.sh OUTDIR [KERNEL-SUBSTRING PROGRAM ARGS...]
# (default: the 64 -> 64 @96^3 conv launch of tools/conv_bench.py)
# GRBM_GUI_ACTIVE = shader-clock cycles of the launch (/ its duration = GHz); SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CYCLES: matrix-pipe duty.
O=$GRAFT_REPO_ROOT/gpurun_out/$1; R=$GRAFT_REPO_ROOT
mkdir -p $O
shift; KSUB=${1:-"conv_igemm_kernel<4, 3, true"}; if [ $# -gt 0 ]; then shift; fi
if [ $# -eq 0 ]; then set -- $R/tools/conv_bench.py 64 96 8 20; fi
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 --kernel-trace --output-format csv -d $O/pmc_clk -o p -- python3 "$@" > /dev/null 2>&1
cd $R
python3 - $O "$KSUB" <<'PY'
import csv, glob, sys, collections
O = sys.argv[1]
f = glob.glob(O + "/pmc_clk/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(list); dur = []
for r in csv.DictReader(open(f[0])):
    if sys.argv[2] not in r["Kernel_Name"]: continue
    agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    if r["Counter_Name"] == "GRBM_GUI_ACTIVE": dur.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
m = {k: sum(v) / len(v) for k, v in agg.items()}
d = sum(dur) / len(dur)
out = ["%s under rocprofv3 --pmc (mean of %d launches): duration %.1f us" % (sys.argv[2], len(dur), d / 1e3)]
out.append("  " + "  ".join("%s=%.4g" % kv for kv in sorted(m.items())))
if "GRBM_GUI_ACTIVE" in m: out.append("  shader clock = GRBM_GUI_ACTIVE / 8 XCDs / duration = %.3f GHz;  matrix pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / cycles = %.3f" % (m["GRBM_GUI_ACTIVE"] / 8 / d, m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 1024 / (m["GRBM_GUI_ACTIVE"] / 8)))
print("\n".join(out)); open(O + "/conv_clock.txt", "w").write("\n".join(out) + "\n")
PY
