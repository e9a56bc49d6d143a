#!/bin/bash
O=gpurun_out/r06fuzz; mkdir -p $O
python tools/timing_fuzz.py --workload scan --iters 3000 > $O/timing_fuzz_scan_3000.txt 2>&1; tail -3 $O/timing_fuzz_scan_3000.txt | cut -c1-300
python tools/timing_fuzz.py --workload scan --iters 1000 --heavy > $O/timing_fuzz_scan_heavy_1000.txt 2>&1; tail -3 $O/timing_fuzz_scan_heavy_1000.txt | cut -c1-300
