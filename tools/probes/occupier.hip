// Probe for the conv kernel's tile scheduler: `nblk` blocks that do nothing but hold their CUs for `usec` microseconds (what a communication
// kernel on another stream does to a compute launch that wants every CU).  Built by tools/conv_corun.py into gpurun_out/, never part of the library.
#include <hip/hip_runtime.h>
#include <stdint.h>
__global__ void occupy_kernel(long long ticks) {
    extern __shared__ int lds[];                                     // 64 KiB: a conv block (135 KiB) cannot share the CU, as with a register-hungry co-runner
    if (ticks < 0) lds[threadIdx.x] = 1;
    const long long t0 = __builtin_amdgcn_s_memrealtime();          // 100 MHz
    while ((long long)__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}
extern "C" int occupy(int nblk, int threads, double usec, void* stream) {
    hipLaunchKernelGGL(occupy_kernel, dim3(nblk), dim3(threads), 65536, (hipStream_t)stream, (long long)(usec * 100.0));
    return (int)hipGetLastError();
}
