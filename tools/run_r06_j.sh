#!/bin/bash
O=gpurun_out/r06j; mkdir -p $O; rm -f $O/plan.txt
timeout 900 python -m pytest tests/test_scan_gpu.py tests/test_configs_gpu.py -x -q -m gpu 2>&1 | tail -2
for b in 1 2 4 8; do python tools/scan_exp/time_scan.py $b 20 2>/dev/null | tee -a $O/plan.txt; done
for ch in 128 256 512; do echo "chunk $ch"; GFE_SSCAN_CHUNK=$ch python tools/scan_exp/time_scan.py 1 20 2>/dev/null | tee -a $O/plan.txt; done
for ch in 512 1024; do echo "chunk $ch B=2"; GFE_SSCAN_CHUNK=$ch python tools/scan_exp/time_scan.py 2 20 2>/dev/null | tee -a $O/plan.txt; done
for ch in 1024 2048; do echo "chunk $ch B=4"; GFE_SSCAN_CHUNK=$ch python tools/scan_exp/time_scan.py 4 20 2>/dev/null | tee -a $O/plan.txt; done
