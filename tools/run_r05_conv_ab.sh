#!/bin/bash
# same-box A/B of two builds of the library: $1 = the other library (e.g. gfe-mamba_amd/gfe_hip/libgfe_hip_old.so); conv kernel alone, then the step
OTHER=$GRAFT_REPO_ROOT/$1
for i in 1 2 3; do
  echo "new:  $(python tools/conv_bench.py 64 96 8 20 2>/dev/null | tail -1)"
  echo "other: $(GFE_HIP_LIB=$OTHER python tools/conv_bench.py 64 96 8 20 2>/dev/null | tail -1)"
done
for i in 1 2 3; do
  echo "new:  $(python bench.py --no-cpu-baseline --steps 30 --warmup 8 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["roofline"]["launch_ms"])')"
  echo "other: $(GFE_HIP_LIB=$OTHER python bench.py --no-cpu-baseline --steps 30 --warmup 8 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["roofline"]["launch_ms"])')"
done
