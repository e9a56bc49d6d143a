#!/bin/bash
# on the GPU box: tools/lift_bench.py for every exp_build/lib_<name>.so given (interleaved)
O=gpurun_out/$1; shift; mkdir -p $O
for n in "$@"; do
  echo "## $n" | tee -a $O/lift_ab.txt
  GFE_HIP_LIB=exp_build/lib_$n.so python tools/lift_bench.py 2>/dev/null | tail -4 | tee -a $O/lift_ab.txt
done
