"""The encoders' 1x1x1 lift convs (64 -> 128 @48^3, 128 -> 256 @24^3, B = 8) with their GroupNorm partials: the one-tap implicit-GEMM path of rounds
1-5 against the streaming product of round 6 (gfe_conv1x1), alternating, same operands.   python tools/lift_bench.py"""
import os, sys
sys.path.insert(0, "gfe-mamba_amd")
import torch
from gfe_hip import nn_ops as K
g = torch.Generator().manual_seed(0)


def med(fn, n=30):
    for _ in range(3):
        fn()
    ts = []
    for _ in range(n):
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); e.record(); e.synchronize()
        ts.append(a.elapsed_time(e) * 1e3)
    ts.sort()
    return ts[n // 2]


for (cin, cout, D) in ((64, 128, 48), (128, 256, 24)):
    x = torch.randn(8, D, D, D, cin, generator=g).to(torch.bfloat16).cuda()
    w5 = (torch.randn(cout, cin, 1, 1, 1, generator=g) / cin ** 0.5).cuda()
    w = K.pack_conv1(w5)
    wb = w5.reshape(cout, cin).to(torch.bfloat16).contiguous()
    b = torch.randn(cout, generator=g).cuda()
    mb = (x.numel() + 8 * D ** 3 * cout) * 2 / 1e6
    for rep in range(2):
        t0 = med(lambda: K.conv_igemm(x, w, [(0, 0, 0)], cout, bias=b, stats=True))
        t1 = med(lambda: K.conv1x1(x, wb, b, stats=True))
        print("lift conv %d->%d @%d^3 (%.0f MB in+out): one-tap implicit GEMM %.1f us = %.2f TB/s | streaming product %.1f us = %.2f TB/s" % (
            cin, cout, D, mb, t0, mb / t0, t1, mb / t1))
