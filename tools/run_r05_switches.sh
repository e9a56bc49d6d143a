#!/bin/bash
# every documented A/B switch and bench mode still runs on the final tree (a few steps each; the value is printed for orientation only)
run() { echo "$1: $(env $2 python bench.py --no-cpu-baseline --steps 6 --warmup 3 $3 2>&1 | tail -1 | python -c 'import sys,json
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["unit"])
except Exception as e: print("FAILED", e)')"; }
run "default" "" ""
run "GFE_XATTN_MATERIALISED=1" "GFE_XATTN_MATERIALISED=1" ""
run "GFE_NO_SIDE_WGRAD=1" "GFE_NO_SIDE_WGRAD=1" ""
run "GFE_CONV_STATIC=1" "GFE_CONV_STATIC=1" ""
run "GFE_CONV_BRICK=0" "GFE_CONV_BRICK=0" ""
run "GFE_GEMM_NO_DMA=1" "GFE_GEMM_NO_DMA=1" ""
run "GFE_CONVT_STREAMED=1" "GFE_CONVT_STREAMED=1" ""
run "GFE_F32_NO_KS=1" "GFE_F32_NO_KS=1" ""
run "GFE_CONV_RESERVE_CUS=8" "GFE_CONV_RESERVE_CUS=8" ""
run "--graph" "" "--graph"
run "--no-pipeline" "" "--no-pipeline"
run "--no-pipeline --graph" "" "--no-pipeline --graph"
run "--batch 1" "" "--batch 1"
run "--batch 4" "" "--batch 4"
run "--volume native --batch 2" "" "--volume native --batch 2"
run "scan GFE_SCAN_DETERMINISTIC=1" "GFE_SCAN_DETERMINISTIC=1" "--workload scan"
run "scan B=1" "" "--workload scan --batch 1"
run "pscan" "" "--workload pscan"
run "normalise" "" "--workload normalise"
run "gen128" "" "--workload gen128"
run "gentrain" "" "--workload gentrain"
run "vit3dtrain" "" "--workload vit3dtrain"
