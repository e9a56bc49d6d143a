import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gfe-mamba_amd")]
import torch
from gfe_hip.scan_ops import selective_scan_tm
B, chunk = int(sys.argv[1]), int(sys.argv[2])
L, ED, N = 4096, 1024, 16
g = torch.Generator().manual_seed(0)
mk = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(torch.bfloat16).cuda().requires_grad_(True)
u, d, z, Bm, Cm = mk(B, L, ED), mk(B, L, ED, sc=0.1), mk(B, L, ED), mk(B, L, N), mk(B, L, N)
A = (-(torch.arange(1, N + 1, dtype=torch.float32)).repeat(ED, 1)).cuda().requires_grad_(True)
D = torch.ones(ED, device="cuda", requires_grad=True)
bias = torch.full((ED,), -3.0, device="cuda", requires_grad=True)
dy = torch.randn(B, L, ED, generator=g).to(torch.bfloat16).cuda()
for _ in range(20):
    y = selective_scan_tm(u, d, A, Bm, Cm, D, z=z, delta_bias=bias, delta_softplus=True, chunk=chunk)
    y.backward(dy)
torch.cuda.synchronize()
