#!/bin/bash
# on the GPU box: times every exp_build/lib_<name>.so given on the command line (interleaved twice, so that clock drift shows)
O=gpurun_out/$1; shift; mkdir -p $O
for rep in 1 2; do
  for n in "$@"; do
    GFE_HIP_LIB=exp_build/lib_$n.so python tools/scan_exp/time_scan.py 8 20 2>/dev/null | tee -a $O/ab.txt
  done
done
