import os, sys
ROOT = "/root/repo"
sys.path[:0] = [ROOT, os.path.join(ROOT, "gfe-mamba_amd")]
import torch
import gfe_hip.gen_train as GT
B = 2
xh = torch.randn(B, 128, 128, 128, 64, device="cuda").to(torch.bfloat16)
d = torch.randn(B, 128, 128, 128, 64, device="cuda").to(torch.bfloat16)
for _ in range(3): GT.conv_wgrad(xh, d, GT.K.CONV3_TAPS)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): GT.conv_wgrad(xh, d, GT.K.CONV3_TAPS)
e1.record(); e1.synchronize()
print("wgrad 64->64 @128^3 B=2: %.3f ms" % (e0.elapsed_time(e1) / 10))
