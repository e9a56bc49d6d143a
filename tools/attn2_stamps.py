#!/usr/bin/env python3
"""In-kernel cycle stamps of the two-sub-tile attention experiment (diagnostic build exp_build/lib_a2_stamp.so).
Slots: 1 kernel entry, 2 first tile landed, 3 / 4 start of phase 1 / 2 of a block, 5 / 6 before / after the tile barrier + DMA issue, 7 loop end, 8 verdict done, 9 exit."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gfe-mamba_amd"))
import torch, numpy as np
from gfe_hip import nn_ops as K
B, H, n, dh = 8, 8, 1729, 64
g = torch.Generator().manual_seed(0)
qkv = (torch.randn(B * n, 3 * H * dh, generator=g)).to(torch.bfloat16).cuda()
inner = H * dh
f = lambda: K.attention_fwd(qkv[:, :inner], qkv[:, inner:2 * inner], qkv[:, 2 * inner:], B, H, n, dh, dh ** -0.5)
for _ in range(3): f()
buf = torch.zeros(8 * 1024 * 2, dtype=torch.int64, device="cuda")
L = ctypes.CDLL(os.environ["GFE_HIP_LIB"])
L.gfe_debug_set_a2_stamp.argtypes = [ctypes.c_void_p]
assert L.gfe_debug_set_a2_stamp(buf.data_ptr()) == 0
torch.cuda.synchronize(); f(); torch.cuda.synchronize()
st = buf.view(8, 1024, 2).cpu().numpy()
for wv in range(8):
    rec = st[wv]; nrec = int((rec[:, 0] != 0).sum()); rec = rec[:nrec]
    seg = {}
    for i in range(1, nrec):
        seg.setdefault((int(rec[i - 1, 0]), int(rec[i, 0])), []).append(int(rec[i, 1] - rec[i - 1, 1]))
    print(f"wave {wv}: {nrec} stamps, total {int(rec[-1, 1] - rec[0, 1])} cycles; " + "  ".join(f"{a}->{b}: n={len(v)} avg={np.mean(v):.0f} med={np.median(v):.0f} max={max(v)}" for (a, b), v in sorted(seg.items())))
