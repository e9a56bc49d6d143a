#!/bin/bash
O=gpurun_out/r06m; mkdir -p $O
timeout 2400 python -m pytest tests/test_unet_gpu.py tests/test_configs_gpu.py tests/test_gen_train_gpu.py tests/test_ladder_gpu.py -x -q -m gpu > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
python tools/timing_fuzz.py --iters 300 > $O/timing_fuzz_gen_300.txt 2>&1; tail -2 $O/timing_fuzz_gen_300.txt | cut -c1-300
for i in 1 2; do python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('step', d['value'], d['ms_per_step'], d['roofline']['frac'])"; done
