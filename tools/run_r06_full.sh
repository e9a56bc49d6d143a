#!/bin/bash
# full GPU suite + smoke + driver-style headline
O=gpurun_out/r06full; mkdir -p $O
timeout 2400 python -m pytest tests -x -q -m gpu > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
python bench.py --steps 20 --warmup 5 > $O/step_b8_bench.json 2> $O/step.err; cut -c1-300 $O/step_b8_bench.json
