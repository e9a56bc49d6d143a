#!/bin/bash
# on the GPU box: times the 64->64 @96^3 conv launch for every exp_build/lib_<name>.so given (interleaved, twice)
O=gpurun_out/$1; shift; mkdir -p $O
for rep in 1 2; do
  for n in "$@"; do
    echo -n "$n: " | tee -a $O/conv_ab.txt
    GFE_HIP_LIB=exp_build/lib_$n.so python tools/conv_bench.py 64 96 8 30 2>/dev/null | tail -1 | tee -a $O/conv_ab.txt
  done
done
