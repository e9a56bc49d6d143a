#!/bin/bash
# The PMC passes of tools/collect_profiles_r06.sh alone (separate --pmc runs with --kernel-trace only) + the summaries: re-run after a kernel
# source changed, so that profiles/r06/traffic_r06.json describes the kernels in the tree (tests/test_abi.py checks the stored hashes).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_step -o step -- python3 $R/bench.py --no-cpu-baseline --steps 10 --warmup 3 > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_scan_fetch -o p -- python3 $R/bench.py --workload scan --no-cpu-baseline --steps 5 --warmup 2 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_scan_write -o p -- python3 $R/bench.py --workload scan --no-cpu-baseline --steps 5 --warmup 2 > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_conv_fetch -o p -- python3 $R/tools/conv_bench.py 64 96 8 10 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_conv_write -o p -- python3 $R/tools/conv_bench.py 64 96 8 10 > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_attn_fetch -o p -- python3 $R/tools/attn_bench.py 8 8 1729 20 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_attn_write -o p -- python3 $R/tools/attn_bench.py 8 8 1729 20 > /dev/null 2>&1
bash $R/tools/pmc_attn.sh r06/attn_pmc attn_fwd attn_bwd_dkdv attn_bwd_dq > $O/attn_pmc.txt 2>&1
cd $R
cp $O/prof_step/step_kernel_stats.csv $O/step_b8_kernel_stats.csv
python3 $R/tools/summarise_profiles_r06.py $O $R
cp $O/traffic_r06.json $R/profiles/r06/traffic_r06.json      # (the box's copy: the bench lines below cite the counters just taken)
python3 $R/bench.py > $O/step_b8_bench.json 2> /dev/null
python3 $R/bench.py --workload vit3d > $O/vit3d_b8_bench.json 2> /dev/null
