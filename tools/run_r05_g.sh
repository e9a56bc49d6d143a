#!/bin/bash
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
python -m pytest tests/test_unet_gpu.py -x -q -m gpu -k "attention or vit3d or vit_3d" 2>&1 | tail -3
{
for rep in 1 2 3; do
for v in product rt kfirst; do
  if [ $v = product ]; then L=""; else L=$R/exp_build/lib_attn_$v.so; fi
  for b in 8 32; do echo -n "$v B=$b: "; env ${L:+GFE_HIP_LIB=$L} python tools/attn_bench.py $b 8 1729 60 2>/dev/null | grep "attention B" | head -1; done
done; done
} 2>&1 | tee gpurun_out/r05/attn_two_instances_ab.txt
