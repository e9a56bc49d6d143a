import os, sys
sys.path.insert(0, "gfe-mamba_amd")
import torch
from gfe_hip import nn_ops as K
g = torch.Generator().manual_seed(0)
for (cin, cout, D) in ((64, 128, 48), (128, 256, 24)):
    x = torch.randn(8, D, D, D, cin, generator=g).to(torch.bfloat16).cuda()
    w = K.pack_conv1((torch.randn(cout, cin, 1, 1, 1, generator=g) / cin ** 0.5).cuda())
    b = torch.randn(cout, generator=g).cuda()
    for _ in range(3):
        K.conv_igemm(x, w, [(0, 0, 0)], cout, bias=b, stats=True)
    ts = []
    for _ in range(30):
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); K.conv_igemm(x, w, [(0, 0, 0)], cout, bias=b, stats=True); e.record(); e.synchronize()
        ts.append(a.elapsed_time(e) * 1e3)
    ts.sort()
    mb = (x.numel() + 8 * D ** 3 * cout) * 2 / 1e6
    print("lift conv %d->%d @%d^3: median %.1f us (%.0f MB in+out = %.2f TB/s)" % (cin, cout, D, ts[15], mb, mb / ts[15]))
