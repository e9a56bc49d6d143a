#!/bin/bash
# round 5, GPU call: untracked attention forward -- parity + A/B against the tracked-only build
mkdir -p gpurun_out/r05
python -m pytest tests/test_unet_gpu.py -x -q -m gpu -k "attention or vit3d or vit_3d" -s 2>&1 | grep -E "passed|failed|Error|error|untracked|flash" | tail -30
for i in 1 2 3; do
  echo "--- untracked (product)"; python tools/attn_bench.py 8 8 1729 100 2>&1 | grep "attention B" | head -1
  echo "--- tracked only"; GFE_HIP_LIB=$GRAFT_REPO_ROOT/exp_build/lib_attn_track.so python tools/attn_bench.py 8 8 1729 100 2>&1 | grep "attention B" | head -1
done 2>&1 | tee gpurun_out/r05/attn_untracked_ab.txt
echo "--- B=32"; python tools/attn_bench.py 32 8 1729 30 2>&1 | grep "attention B" | head -1 | tee -a gpurun_out/r05/attn_untracked_ab.txt
GFE_HIP_LIB=$GRAFT_REPO_ROOT/exp_build/lib_attn_track.so python tools/attn_bench.py 32 8 1729 30 2>&1 | grep "attention B" | head -1 | tee -a gpurun_out/r05/attn_untracked_ab.txt
