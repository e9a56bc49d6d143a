#!/bin/bash
O=gpurun_out/r06j; mkdir -p $O
for b in 1 2 4 8 16; do python tools/scan_exp/time_scan.py $b 20 2>/dev/null | tee -a $O/plan.txt; done
GFE_SSCAN_CHUNK=256 python tools/scan_exp/time_scan.py 1 20 2>/dev/null | tee -a $O/plan.txt
GFE_SSCAN_CHUNK=1024 python tools/scan_exp/time_scan.py 1 20 2>/dev/null | tee -a $O/plan.txt
