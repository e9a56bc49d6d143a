for rep in 1 2 3; do
  echo -n "conv1x1 serial: "; python bench.py --no-pipeline --no-cpu-baseline --steps 40 --warmup 10 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
  echo -n "one-tap serial: "; python tools/bench_old_lift.py --no-pipeline --no-cpu-baseline --steps 40 --warmup 10 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done
