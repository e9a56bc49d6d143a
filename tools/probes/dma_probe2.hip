// Probe: cost of issuing buffer_load..lds pieces (1 KiB / wave instruction) for different source address patterns.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((address_space(3))) void* lds_void_t;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void* base, unsigned bytes) {
    const uint64_t a = (uint64_t)base;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a), hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
    return __builtin_amdgcn_make_buffer_rsrc((void*)(((uint64_t)hi << 32) | lo), 0, (int)__builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}
// pattern 0: contiguous 1 KiB; 1: 16 segments of 64 B, 128 B apart (one slab of a 64-channel voxel row); 2: 8 segments of 128 B contiguous lines spaced 256
template <int NP, int PAT, bool SOFF>
__global__ __launch_bounds__(512) void probe(const char* src, size_t bytes_per_block, long long* out, int iters, int nwaves_issue) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const char* base = src + (size_t)blockIdx.x * bytes_per_block;
    const __amdgpu_buffer_rsrc_t rs = rsrc(base, (unsigned)bytes_per_block);
    unsigned lane_off;
    if (PAT == 0) lane_off = lane * 16;
    else if (PAT == 1) lane_off = (lane >> 2) * 128 + (lane & 3) * 16;
    else lane_off = (lane >> 3) * 256 + (lane & 7) * 16;
    const unsigned piece_stride = PAT == 0 ? 1024 : 2048;
    long long t_issue = 0, t_wait = 0;
    unsigned pos = wave * NP * piece_stride;
    for (int it = 0; it < iters; ++it) {
        __builtin_amdgcn_s_barrier();
        if (wave < nwaves_issue) {
            const long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
            for (int j = 0; j < NP; ++j)
                if (SOFF) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t)(smem + (wave * NP + j) * 1024), 16, lane_off + j * piece_stride, __builtin_amdgcn_readfirstlane(pos), 0, 0);
                else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t)(smem + (wave * NP + j) * 1024), 16, lane_off + pos + j * piece_stride, 0, 0, 0);
            const long long t1 = __builtin_amdgcn_s_memtime();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const long long t2 = __builtin_amdgcn_s_memtime();
            t_issue += t1 - t0; t_wait += t2 - t1;
            pos += 8 * NP * piece_stride;
            if (pos + NP * piece_stride + 4096 > bytes_per_block) pos = wave * NP * piece_stride;
        }
    }
    if (lane == 0 && wave < nwaves_issue) { out[(blockIdx.x * 8 + wave) * 2] = t_issue; out[(blockIdx.x * 8 + wave) * 2 + 1] = t_wait; }
#endif
}
template <int NP, int PAT, bool SOFF = false>
void run(const char* name, const char* d, size_t bpb, long long* dout, int nw, int iters, int grid = 256) {
    hipFuncSetAttribute((const void*)probe<NP, PAT, SOFF>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipMemset(dout, 0, 256 * 8 * 2 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((probe<NP, PAT, SOFF>), dim3(grid), dim3(512), 131072, 0, d, bpb, dout, iters, nw);
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe<NP, PAT, SOFF>), dim3(grid), dim3(512), 131072, 0, d, bpb, dout, iters, nw);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(256 * 8 * 2); hipMemcpy(h.data(), dout, h.size() * 8, hipMemcpyDeviceToHost);
    double is = 0, wt = 0; int n = 0;
    for (int b = 0; b < grid; ++b) for (int w = 0; w < nw; ++w) { is += h[(b * 8 + w) * 2]; wt += h[(b * 8 + w) * 2 + 1]; ++n; }
    const double pieces = (double)iters * NP;
    printf("grid %3d bpb %4zu KiB %-20s NP=%2d waves=%d: issue %.0f cyc/piece, wait %.0f cyc/iter, %.3f ms, %.2f TB/s\n", grid, bpb >> 10, name, NP, nw, is / n / pieces, wt / n / iters, ms,
           (double)grid * nw * pieces * 1024 / ms / 1e9);
}
int main() {
    const size_t bpb = 8u << 20;            // 8 MiB per block: 2 GiB total, beyond L2 + MALL
    char* d; hipMalloc(&d, 256 * bpb); hipMemset(d, 1, 256 * bpb);
    long long* dout; hipMalloc(&dout, 256 * 8 * 2 * 8);
    for (int grid : {64, 256}) {
        for (size_t sz : {(size_t)8 << 20, (size_t)256 << 10}) {
            run<16, 1, false>("64B@128 voffset", d, sz, dout, 4, 200, grid);
            run<16, 1, true>("64B@128 soffset", d, sz, dout, 4, 200, grid);
            run<16, 0, false>("contig voffset", d, sz, dout, 4, 200, grid);
            run<16, 0, true>("contig soffset", d, sz, dout, 4, 200, grid);
        }
    }
    return 0;
}
