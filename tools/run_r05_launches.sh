#!/bin/bash
# what a launch on the head's stream costs the pipelined step: N extra one-element launches per step behind the optimiser, alternating runs on one box
for i in 1 2; do for n in 0 100 200 400; do echo "extra head launches $n: $(GFE_EXP_HEAD_LAUNCHES=$n python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"])')"; done; done
