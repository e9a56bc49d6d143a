#!/bin/bash
# Runs on the GPU box: bench JSON lines + rocprofv3 kernel stats + PMC passes -> gpurun_out/r01/ (copied to profiles/r01/ afterwards).
# usage: tools/collect_profiles.sh TAG
TAG=${1:-v3}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r01
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $O/step_b8_${TAG}_bench.json 2> /dev/null
python3 $R/bench.py --workload vit3d > $O/vit3d_b8_${TAG}_bench.json 2> /dev/null
python3 $R/bench.py --workload normalise > $O/normalise_b8_${TAG}_bench.json 2> /dev/null
for b in 1 8 64; do python3 $R/bench.py --workload scan --batch $b > $O/scan_b${b}_${TAG}_bench.json 2> /dev/null; done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_step -o step -- python3 $R/bench.py --no-cpu-baseline --steps 10 --warmup 3 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_scan -o scan -- python3 $R/bench.py --workload scan --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_vit3d -o vit3d -- python3 $R/bench.py --workload vit3d --no-cpu-baseline > /dev/null 2>&1
# the roofline kernel alone (one shape: 64->64 @96^3, B=8), so that its average duration can be read off directly
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_conv64 -o conv64 -- python3 $R/tools/conv_bench.py 64 96 8 20 > /dev/null 2>&1
cp $O/prof_conv64/conv64_kernel_stats.csv $O/conv64_b8_${TAG}_kernel_stats.csv
cp $O/prof_step/step_kernel_stats.csv $O/step_b8_${TAG}_kernel_stats.csv
cp $O/prof_scan/scan_kernel_stats.csv $O/scan_b8_${TAG}_kernel_stats.csv
cp $O/prof_vit3d/vit3d_kernel_stats.csv $O/vit3d_b8_${TAG}_kernel_stats.csv
# PMC passes (separate runs, kernel-trace only): HBM traffic of the conv / scan / attention kernels
for what in conv attn; do
  tool=$R/tools/${what}_bench.py
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_${what}_fetch -o p -- python3 $tool > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_${what}_write -o p -- python3 $tool > /dev/null 2>&1
  rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $O/pmc_${what}_sq -o p -- python3 $tool > /dev/null 2>&1
done
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_scan_fetch -o p -- python3 $R/bench.py --workload scan --no-cpu-baseline --steps 5 --warmup 2 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_scan_write -o p -- python3 $R/bench.py --workload scan --no-cpu-baseline --steps 5 --warmup 2 > /dev/null 2>&1
cd $R
python3 - <<PY
import csv, glob, collections, os
O = "$O"
out = open(os.path.join(O, "pmc_summary_${TAG}.txt"), "w")
for d in sorted(glob.glob(O + "/pmc_*")):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    if not f: continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list)); dur = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"]
        if not any(t in k for t in ("conv_igemm", "attn_fwd", "sscan")): continue
        k = k.split("(")[0][-60:]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"])); dur[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    for k in agg:
        line = f"{os.path.basename(d):18s} {k:62s} launches {len(dur[k]) // max(1, len(agg[k])):4d} avg_us {sum(dur[k]) / len(dur[k]) / 1e3:9.1f} " + \
               " ".join(f"{c}={sum(v) / len(v):.4g}" for c, v in agg[k].items())
        print(line); out.write(line + "\n")
out.close()
PY
for f in $O/*_bench.json; do echo "$(basename $f): $(cut -c1-260 $f)"; done
