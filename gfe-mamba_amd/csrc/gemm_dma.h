// Interface between gemm.hip (dispatch, operand modes, split-K) and gemm_dma.hip (the persistent LDS-DMA main loop for the plain
// bf16 x bf16 K-major shape: every nn.Linear forward of the inference pipelines, vit_pytorch_diy/vit_3d.py:41-46, 50).
#pragma once
#include <stdint.h>
#include <hip/hip_runtime.h>

struct GemmDmaArgs {
    const void* A; const void* B; void* C; const float* bias; const void* res;
    int64_t lda, ldb, ldc, ldres;
    int M, N, K;
    int out_f32, res_f32, act;
    // K cut into nsplit equal ranges (K % (64 * nsplit) == 0), range z's tile stored RAW (f32, no bias / activation / residual) to part[z][M][N];
    // the caller's fixed-order reduction finishes the product (gemm.hip, splitk_reduce_kernel).  nsplit <= 1: part is ignored.
    int nsplit = 1; float* part = nullptr;
};

// true: shape / alignment / size limits of the DMA kernel hold (gemm_dma_launch may be called)
bool gemm_dma_usable(const GemmDmaArgs& a);
int gemm_dma_launch(const GemmDmaArgs& a, hipStream_t st);
