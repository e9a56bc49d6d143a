#!/bin/bash
O=gpurun_out/r06k; mkdir -p $O
timeout 1500 python -m pytest tests/test_head_gpu.py -x -q -m gpu -s -k "step_vs_reference or 160x160" > $O/pytest.txt 2>&1; grep -E "gradient elements|passed|failed|Error" $O/pytest.txt | cut -c1-600
