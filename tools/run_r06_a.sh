#!/bin/bash
# round 6, call A: today's baseline of the scan on this box (bench lines + in-kernel stamps) before the kernels change
O=gpurun_out/r06a; mkdir -p $O
python bench.py --workload scan --batch 8 --steps 30 --warmup 5 > $O/scan_b8.json 2> $O/scan_b8.err
python bench.py --workload scan --batch 1 --steps 30 --warmup 5 > $O/scan_b1.json 2> $O/scan_b1.err
GFE_HIP_LIB=exp_build/lib_stamps.so python tools/scan_stamps.py 8 > $O/stamps_b8.txt 2>&1
cat $O/scan_b8.json $O/scan_b1.json $O/stamps_b8.txt
