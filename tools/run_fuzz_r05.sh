#!/bin/bash
# round 5, GPU call 1: timing-fuzz harness + the diagnosable invariance test + the touched GEMM test
mkdir -p gpurun_out/r05
python tools/timing_fuzz.py --iters ${1:-200} > gpurun_out/r05/timing_fuzz.txt 2>&1; echo "fuzz rc $?" >> gpurun_out/r05/timing_fuzz.txt
tail -5 gpurun_out/r05/timing_fuzz.txt
python -m pytest tests/test_configs_gpu.py tests/test_unet_gpu.py -x -q -m gpu -k "batch8 or skinny or gemm" 2>&1 | tail -5
python bench.py --steps 20 --warmup 5 2>&1 | tail -1 | tee gpurun_out/r05/step_b8_bench_start.json
