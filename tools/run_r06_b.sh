#!/bin/bash
# round 6, call B: new forward (old backward): parity tests of the scan, then bench lines + stamps
O=gpurun_out/r06b; mkdir -p $O
timeout 900 python -m pytest tests/test_scan_gpu.py tests/test_configs_gpu.py -x -q -m gpu > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt
python bench.py --workload scan --batch 8 --steps 30 --warmup 5 > $O/scan_b8.json 2> $O/scan_b8.err
python bench.py --workload scan --batch 1 --steps 30 --warmup 5 > $O/scan_b1.json 2> $O/scan_b1.err
GFE_HIP_LIB=exp_build/lib_stamps.so python tools/scan_stamps.py 8 > $O/stamps_b8.txt 2>&1
python - <<'PY'
import json
for f in ("scan_b8", "scan_b1"):
    try:
        d = json.loads(open("gpurun_out/r06b/%s.json" % f).read()); r = d["roofline"]
        print(f, d["ms_per_step"], r["frac"], r["fwd_ms"], r["bwd_ms"])
    except Exception as e: print(f, "failed", e)
PY
cat $O/stamps_b8.txt | tail -4
