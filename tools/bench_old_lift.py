"""A/B arm: bench.py with the encoders' 1x1x1 lift convs on the one-tap implicit-GEMM path of rounds 1-5 instead of gfe_conv1x1 (round 6).
    python tools/bench_old_lift.py --steps 20 --warmup 5      (same arguments as bench.py)"""
import os, runpy, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gfe-mamba_amd"))
from gfe_hip import nn_ops as K
K.conv1x1_ok = lambda cin, cout: False
sys.argv[0] = os.path.join(ROOT, "bench.py")
runpy.run_path(sys.argv[0], run_name="__main__")
