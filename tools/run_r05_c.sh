#!/bin/bash
# round 5, GPU call 3: folded vs materialised cross-attention, same box
mkdir -p gpurun_out/r05
python -m pytest tests/test_head_gpu.py -x -q -m gpu -k "folded or cross_attention or pipelined or graphed" 2>&1 | tail -4
O=$GRAFT_REPO_ROOT/gpurun_out/r05
for i in 1 2; do
  for m in 0 1; do GFE_XATTN_MATERIALISED=$m python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('materialised=$m step', d['value'], d['ms_per_step'])"; done
done 2>&1 | tee $O/xattn_fold_step_ab.txt
for m in 0 1; do echo "materialised=$m"; GFE_XATTN_MATERIALISED=$m python tools/head_graph_probe.py 8; done 2>&1 | grep -v amdgpu.ids | tee -a $O/xattn_fold_step_ab.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_step -o step -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 10 --warmup 3 > /dev/null 2>&1
cp $O/prof_step/*/step_kernel_stats.csv $O/step_b8_kernel_stats.csv 2>/dev/null || cp $O/prof_step/step_kernel_stats.csv $O/step_b8_kernel_stats.csv
grep -E "xf_|Name" $O/step_b8_kernel_stats.csv | head
