#!/bin/bash
# full GPU suite + driver-style headline + sustained headline
O=gpurun_out/r06full; mkdir -p $O
timeout 2400 python -m pytest tests -x -q -m gpu > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt
python bench.py --steps 20 --warmup 5 > $O/step_b8_bench.json 2> $O/step.err; cut -c1-400 $O/step_b8_bench.json
python bench.py --steps 3000 --warmup 50 --no-cpu-baseline > $O/step_b8_sustained.json 2>> $O/step.err; cut -c1-300 $O/step_b8_sustained.json
