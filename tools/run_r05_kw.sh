#!/bin/bash
# does the headline depend on the driver's --steps 20 --warmup 5?  alternating runs on one box
for i in 1 2 3; do
  for kw in "20 5" "40 10" "20 10" "100 10"; do
    set -- $kw
    echo "steps $1 warmup $2: $(python bench.py --no-cpu-baseline --steps $1 --warmup $2 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d["roofline"]["launch_ms"])')"
  done
done
