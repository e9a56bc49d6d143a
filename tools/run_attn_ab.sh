#!/bin/bash
# on the GPU box: tools/attn_bench.py (forward line) for every exp_build/lib_<name>.so given (interleaved, twice)
O=gpurun_out/$1; shift; mkdir -p $O
for rep in 1 2; do
  for n in "$@"; do
    echo -n "$n: " | tee -a $O/attn_ab.txt
    GFE_HIP_LIB=exp_build/lib_$n.so python tools/attn_bench.py 8 8 1729 50 2>/dev/null | grep -v backward | tail -1 | tee -a $O/attn_ab.txt
  done
done
