#!/bin/bash
# after the collection: the bench lines that cite profiles/r06/traffic_r06.json, and the new scan edge test
O=gpurun_out/r06n; mkdir -p $O
timeout 900 python -m pytest tests/test_scan_gpu.py -x -q -m gpu 2>&1 | tail -2
python bench.py --steps 20 --warmup 5 > $O/step_b8_bench.json 2>/dev/null
python bench.py --steps 3000 --warmup 50 --no-cpu-baseline > $O/step_b8_sustained.json 2>/dev/null
python bench.py --workload scan --batch 8 > $O/scan_b8_bench.json 2>/dev/null
python bench.py --workload vit3d > $O/vit3d_b8_bench.json 2>/dev/null
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r06n/*.json')):
    d = json.loads(open(f).read()); r = d.get('roofline', {})
    print(f.split('/')[-1], d['value'], d['ms_per_step'], r.get('frac'), r.get('traffic'))
PY
