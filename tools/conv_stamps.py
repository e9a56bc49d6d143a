#!/usr/bin/env python3
"""In-kernel cycle stamps of the conv kernel (diagnostic build exp_build/lib_STAMP.so, see tools/build_exp.sh):
    GFE_HIP_LIB=exp_build/lib_STAMP.so python tools/conv_stamps.py [C] [D]
Stamp slots per stage: 1 before the wait+barrier, 2 after the barrier, 3 after the DMA issue; per unit: 4 after the last MFMA issued,
5 after the epilogue.  Prints per-wave average cycles of each segment for one block."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gfe-mamba_amd"))
import torch
from gfe_hip import nn_ops as K, lib
C = int(sys.argv[1]) if len(sys.argv) > 1 else 64
D = int(sys.argv[2]) if len(sys.argv) > 2 else 96
B = 8
g = torch.Generator().manual_seed(0)
x = torch.randn(B, D, D, D, C, generator=g).to(torch.bfloat16).cuda()
w32 = K.pack_conv3((torch.randn(C, C, 3, 3, 3, generator=g) / (27 * C) ** 0.5).cuda(), torch.float32)
ss = K.groupnorm_scale_shift(x, torch.ones(C, device="cuda"), torch.zeros(C, device="cuda"), 8)
w, tab = K.fold_groupnorm(w32, ss[0], ss[1], K.CONV3_TAPS, C, C)
for _ in range(3):
    y = K.conv_igemm(x, w, K.CONV3_TAPS, C, bias_tab=tab, relu=True)
buf = torch.zeros(8 * 4096 * 2, dtype=torch.int64, device="cuda")
L = ctypes.CDLL(os.environ["GFE_HIP_LIB"])
L.gfe_debug_set_stamp_buffer.argtypes = [ctypes.c_void_p]
assert L.gfe_debug_set_stamp_buffer(buf.data_ptr()) == 0
torch.cuda.synchronize()
y = K.conv_igemm(x, w, K.CONV3_TAPS, C, bias_tab=tab, relu=True)
torch.cuda.synchronize()
st = buf.view(8, 4096, 2).cpu().numpy()
import numpy as np
for wv in range(8):
    rec = st[wv]; n = int((rec[:, 0] != 0).sum())
    rec = rec[:n]
    seg = {}
    for i in range(1, n):
        key = (int(rec[i - 1, 0]), int(rec[i, 0]))
        seg.setdefault(key, []).append(int(rec[i, 1] - rec[i - 1, 1]))
    tot = int(rec[-1, 1] - rec[0, 1])
    print(f"wave {wv}: {n} stamps, total {tot} cyc; " + "  ".join(f"{a}->{b}: n={len(v)} avg={np.mean(v):.0f} med={np.median(v):.0f}" for (a, b), v in sorted(seg.items())))
# detailed timeline of one unit for waves 0 and 4
for wv in (0, 4):
    rec = st[wv]; n = int((rec[:, 0] != 0).sum())
    i0 = 200
    print("wave", wv, " ".join(f"{int(rec[i,0])}:{int(rec[i,1]-rec[i0,1])}" for i in range(i0, min(n, i0 + 64))))
