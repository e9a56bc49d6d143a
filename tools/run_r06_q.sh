#!/bin/bash
O=gpurun_out/r06q; mkdir -p $O
bash tools/run_convt_ab.sh r06q ct_base ct_ahead
GFE_HIP_LIB=exp_build/lib_ct_ahead.so timeout 1200 python -m pytest tests/test_unet_gpu.py tests/test_configs_gpu.py -x -q -m gpu 2>&1 | tail -3
