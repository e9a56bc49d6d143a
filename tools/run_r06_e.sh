#!/bin/bash
O=gpurun_out/r06e; mkdir -p $O
timeout 900 python -m pytest tests/test_scan_gpu.py tests/test_configs_gpu.py -x -q -m gpu > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
python tools/scan_exp/time_scan.py 8 20 2>/dev/null
python tools/scan_exp/time_scan.py 8 20 2>/dev/null
python tools/scan_exp/time_scan.py 1 20 2>/dev/null
GFE_HIP_LIB=exp_build/lib_stamps.so python tools/scan_stamps.py 8 2>&1 | tail -3
