#!/bin/bash
# on the GPU box: times the decoders.1 transposed conv (128 -> 64, 48^3 -> 96^3, B = 8) for every exp_build/lib_<name>.so given (interleaved, twice)
O=gpurun_out/$1; shift; mkdir -p $O
for rep in 1 2; do
  for n in "$@"; do
    echo -n "$n: " | tee -a $O/convt_ab.txt
    GFE_HIP_LIB=exp_build/lib_$n.so python tools/convt_bench.py 128 64 48 8 20 2>/dev/null | tail -1 | tee -a $O/convt_ab.txt
  done
done
