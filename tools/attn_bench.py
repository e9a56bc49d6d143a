#!/usr/bin/env python3
"""Micro-benchmark of the flash-attention kernel on the synthetic 3-D ViT shape of SURVEY 8-d (B=8, heads 8, n=1729, d=64).
    python tools/attn_bench.py [B] [H] [n] [iters]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gfe-mamba_amd"))
import torch
from gfe_hip import nn_ops as K

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
H = int(sys.argv[2]) if len(sys.argv) > 2 else 8
n = int(sys.argv[3]) if len(sys.argv) > 3 else 1729
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 50
dh, inner = 64, H * 64
g = torch.Generator().manual_seed(0)
qkv = torch.randn(B * n, 3 * inner, generator=g).to(torch.bfloat16).cuda()
q, k, v = qkv[:, :inner], qkv[:, inner:2 * inner], qkv[:, 2 * inner:]
for _ in range(3):
    K.attention_fwd(q, k, v, B, H, n, dh, dh ** -0.5)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters):
    K.attention_fwd(q, k, v, B, H, n, dh, dh ** -0.5)
e1.record()
e1.synchronize()
ms = e0.elapsed_time(e1) / iters
fl = 4.0 * B * H * n * n * dh
print(f"attention B={B} H={H} n={n}: {ms * 1e3:.1f} us  {fl / ms / 1e9:.1f} TFLOP/s  ({fl / ms / 1e9 / 2500:.3f} of 2.5 PFLOP/s)")

# backward (gfe_attention_bwd: prep + dK/dV + dQ launches): 2.5x the forward's algorithmic flops (S, dP, dV, dK, dQ)
o, nlse = K.attention_fwd(q, k, v, B, H, n, dh, dh ** -0.5, with_lse=True)
dout = torch.randn(B * n, inner, generator=g).to(torch.bfloat16).cuda()
dqkv = torch.empty_like(qkv)
for _ in range(3):
    K.attention_bwd(q, k, v, o, dout, nlse, B, H, n, dh, dh ** -0.5, dqkv=dqkv)
torch.cuda.synchronize()
e0.record()
for _ in range(iters):
    K.attention_bwd(q, k, v, o, dout, nlse, B, H, n, dh, dh ** -0.5, dqkv=dqkv)
e1.record()
e1.synchronize()
ms = e0.elapsed_time(e1) / iters
print(f"attention backward B={B} H={H} n={n}: {ms * 1e3:.1f} us  {2.5 * fl / ms / 1e9:.1f} TFLOP/s  ({2.5 * fl / ms / 1e9 / 2500:.3f} of 2.5 PFLOP/s, "
      f"algorithmic 10 n^2 d flops per head)")
