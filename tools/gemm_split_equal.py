"""Is a K-split product on the LDS-DMA main loop bit-identical to the staged kernel for the same number of ranges?  (yes: profiles/r05/gemm_split_equal.txt)"""
import os, sys
sys.path.insert(0, "gfe-mamba_amd")
import torch
from gfe_hip import nn_ops as K
import gfe_hip
g = torch.Generator().manual_seed(3)
Kd, N = 147456, 512
for M in (200, 25, 50):
    a = torch.randn(M, Kd, generator=g).to(torch.bfloat16).cuda()
    b = (torch.randn(N, Kd, generator=g) * Kd ** -0.5).to(torch.bfloat16).cuda()
    bias = torch.randn(N, generator=g).cuda()
    for split in (48, 64, 32):
        os.environ["GFE_GEMM_NO_DMA"] = "1"                  # the staged kernel (gemm_nt_kernel) for the K ranges
        y0 = K.gemm_nt(a, b, bias=bias, out_dtype=torch.float32, split_k=split)
        os.environ.pop("GFE_GEMM_NO_DMA")                       # the persistent LDS-DMA main loop
        n0 = gfe_hip.lib().gfe_gemm_dma_launches()
        y1 = K.gemm_nt(a, b, bias=bias, out_dtype=torch.float32, split_k=split)
        took = gfe_hip.lib().gfe_gemm_dma_launches() - n0
        print("M", M, "split", split, "dma launches", took, "equal", torch.equal(y0, y1), "max diff", (y0 - y1).abs().max().item())
