#!/bin/bash
# what each part of the head takes out of the pipelined step (timing experiment: GFE_EXP_HEAD_PART != all does NOT train): alternating runs on one box
for i in 1 2; do for part in all fwdbwd fwd none; do echo "head part $part: $(GFE_EXP_HEAD_PART=$part python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"])')"; done; done
