"""Times the fused selective scan (config 2) forward and forward+backward with HIP events for the library GFE_HIP_LIB names.
    GFE_HIP_LIB=exp_build/lib_X.so python tools/scan_exp/time_scan.py [batch] [iters]   -> one line: tag fwd_ms bwd_ms
Used by tools/scan_exp/ab.sh for ablation builds (results of ablated kernels are wrong by construction; only the time is read)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gfe-mamba_amd")]
import torch
import bench
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 30
wl = bench.ScanWorkload(B)
for _ in range(5):
    wl.step()
torch.cuda.synchronize()
with torch.no_grad():
    tf = min(bench.time_region(lambda: wl.fwd(), iters) for _ in range(3))
ts = min(bench.time_region(wl.step, iters) for _ in range(3))
print("%-28s B=%d fwd %.4f ms  bwd %.4f ms  total %.4f" % (os.path.basename(os.environ.get("GFE_HIP_LIB", "product")), B, tf, ts - tf, ts), flush=True)
