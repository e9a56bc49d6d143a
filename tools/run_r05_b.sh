#!/bin/bash
# round 5, GPU call 2: folded one-query cross-attention -- parity + step bench
mkdir -p gpurun_out/r05
python -m pytest tests/test_head_gpu.py -x -q -m gpu -s 2>&1 | grep -v "^$" | tail -40 > gpurun_out/r05/test_head_fold.txt; tail -15 gpurun_out/r05/test_head_fold.txt
for i in 1 2; do python bench.py --steps 30 --warmup 5 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('step', d['value'], d['ms_per_step'], d['roofline']['frac'])"; done
