#!/bin/bash
# Runs on the GPU box: bench JSON lines + rocprofv3 kernel stats (+ scan PMC) -> gpurun_out/r06/ (copied to profiles/r06/ afterwards).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --steps 20 --warmup 5 > $O/step_b8_bench.json 2> /dev/null                      # (the driver's arguments)
python3 $R/bench.py --no-pipeline > $O/step_b8_serial_bench.json 2> /dev/null
python3 $R/bench.py --workload gen128 > $O/gen128_b2_bench.json 2> /dev/null
python3 $R/bench.py --workload vit3d > $O/vit3d_b8_bench.json 2> /dev/null
python3 $R/bench.py --workload gentrain > $O/gentrain_b2_bench.json 2> /dev/null
python3 $R/bench.py --workload vit3dtrain > $O/vit3dtrain_b8_bench.json 2> /dev/null
for b in 1 8 64; do python3 $R/bench.py --workload scan --batch $b > $O/scan_b${b}_bench.json 2> /dev/null; done
python3 $R/bench.py --workload pscan > $O/pscan_b1_bench.json 2> /dev/null
python3 $R/bench.py --volume native --batch 2 > $O/step_native_b2_bench.json 2> /dev/null
python3 $R/bench.py --batch 1 > $O/step_b1_bench.json 2> /dev/null
python3 -m torch.distributed.run --standalone --local-addr 127.0.0.1 --nnodes=1 --nproc-per-node=1 $R/bench.py --gpus 1 --no-cpu-baseline > $O/step_b8_torchrun1_bench.json 2> /dev/null
# two REAL ranks on the one GPU (gloo dry-run transport: the N-rank code path, not a scaling figure)
GFE_DIST_BACKEND=gloo python3 -m torch.distributed.run --standalone --local-addr 127.0.0.1 --nnodes=1 --nproc-per-node=2 $R/bench.py --gpus 2 --batch 4 --steps 10 --warmup 3 > $O/step_b4_2ranks_gloo_bench.json 2> /dev/null
GFE_NO_SIDE_WGRAD=1 python3 $R/bench.py --no-cpu-baseline > $O/step_b8_noside_bench.json 2> /dev/null
python3 $R/tools/gemm_bench.py 5 20 > $O/gemm_ab.txt 2>&1
bash $R/tools/pmc_gemm.sh r06/gemm_pmc > $O/gemm_pmc.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_step -o step -- python3 $R/bench.py --no-cpu-baseline --steps 10 --warmup 3 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_scan8 -o scan -- python3 $R/bench.py --workload scan --batch 8 --no-cpu-baseline --steps 20 --warmup 5 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_scan1 -o scan -- python3 $R/bench.py --workload scan --batch 1 --no-cpu-baseline --steps 20 --warmup 5 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_gen128 -o gen -- python3 $R/bench.py --workload gen128 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_conv64 -o conv64 -- python3 $R/tools/conv_bench.py 64 96 8 20 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_gentrain -o gentrain -- python3 $R/bench.py --workload gentrain --no-cpu-baseline --steps 3 --warmup 1 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_vit3d -o vit -- python3 $R/bench.py --workload vit3d --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_vit3dtrain -o vt -- python3 $R/bench.py --workload vit3dtrain --no-cpu-baseline --steps 5 --warmup 2 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_attnb -o attnb -- python3 $R/tools/attn_bench.py > /dev/null 2>&1
cp $O/prof_vit3dtrain/vt_kernel_stats.csv $O/vit3dtrain_b8_kernel_stats.csv
cp $O/prof_attnb/attnb_kernel_stats.csv $O/attn_bwd_kernel_stats.csv
cp $O/prof_gentrain/gentrain_kernel_stats.csv $O/gentrain_b2_kernel_stats.csv
cp $O/prof_vit3d/vit_kernel_stats.csv $O/vit3d_b8_kernel_stats.csv
cp $O/prof_step/step_kernel_stats.csv $O/step_b8_kernel_stats.csv
cp $O/prof_scan8/scan_kernel_stats.csv $O/scan_b8_kernel_stats.csv
cp $O/prof_scan1/scan_kernel_stats.csv $O/scan_b1_kernel_stats.csv
cp $O/prof_gen128/gen_kernel_stats.csv $O/gen128_b2_kernel_stats.csv
cp $O/prof_conv64/conv64_kernel_stats.csv $O/conv64_b8_kernel_stats.csv
# scan traffic (separate PMC passes, kernel trace only)
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_scan_fetch -o p -- python3 $R/bench.py --workload scan --no-cpu-baseline --steps 5 --warmup 2 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_scan_write -o p -- python3 $R/bench.py --workload scan --no-cpu-baseline --steps 5 --warmup 2 > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_conv_fetch -o p -- python3 $R/tools/conv_bench.py 64 96 8 10 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_conv_write -o p -- python3 $R/tools/conv_bench.py 64 96 8 10 > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_attn_fetch -o p -- python3 $R/tools/attn_bench.py 8 8 1729 20 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_attn_write -o p -- python3 $R/tools/attn_bench.py 8 8 1729 20 > /dev/null 2>&1
bash $R/tools/pmc_attn.sh r06/attn_pmc attn_fwd attn_bwd_dkdv attn_bwd_dq > $O/attn_pmc.txt 2>&1
bash $R/tools/pmc_scan2.sh r06/scan_pmc > /dev/null 2>&1; python3 $R/tools/pmc_summ.py $O/scan_pmc > $O/scan_pmc.txt 2>&1
cd /tmp
cd $R
python3 $R/tools/summarise_profiles_r06.py $O $R
python3 $R/bench.py --steps 3000 --warmup 50 --no-cpu-baseline > $O/step_b8_sustained.json 2> /dev/null      # ~30 s of steps: the sustained figure next to the driver-style one
for f in $O/*_bench.json $O/step_b8_sustained.json; do echo "$(basename $f): $(cut -c1-300 $f)"; done
