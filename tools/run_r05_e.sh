#!/bin/bash
# round 5: fuzz 500 (fresh fuzz library) + attention ablation builds on one box
mkdir -p gpurun_out/r05
python tools/timing_fuzz.py --iters 500 > gpurun_out/r05/timing_fuzz_500.txt 2>&1; tail -3 gpurun_out/r05/timing_fuzz_500.txt
R=$GRAFT_REPO_ROOT
{
for v in product track novalu nomfma nodma neither; do
  if [ $v = product ]; then L=""; else L=$R/exp_build/lib_attn_$v.so; fi
  for b in 8 32; do echo -n "$v B=$b: "; env ${L:+GFE_HIP_LIB=$L} python tools/attn_bench.py $b 8 1729 60 2>/dev/null | grep "attention B" | head -1; done
done
} 2>&1 | tee gpurun_out/r05/attn_ablation.txt
