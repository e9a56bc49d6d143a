#!/bin/bash
# on the GPU box: the headline step with / without gfe_conv1x1, alternating (3 x)
O=gpurun_out/$1; mkdir -p $O
for rep in 1 2 3; do
  echo -n "conv1x1: " | tee -a $O/step_lift_ab.txt; python bench.py --no-cpu-baseline --steps 40 --warmup 10 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" | tee -a $O/step_lift_ab.txt
  echo -n "one-tap: " | tee -a $O/step_lift_ab.txt; python tools/bench_old_lift.py --no-cpu-baseline --steps 40 --warmup 10 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" | tee -a $O/step_lift_ab.txt
done
