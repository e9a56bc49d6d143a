// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 against known byte counts: a streaming read of N bytes with 4, 8 and 16 B
// per lane (global_load_dword / dwordx2 / dwordx4), and a streaming write.  MI355X_MICROARCH.md states that FETCH_SIZE reports half the
// bytes of a wide (16 B/lane) coalesced read; this shows what it reports for the narrower loads the scan kernels use.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/fetch_calib.hip -o exp_build/fetch_calib
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out -o p -- exp_build/fetch_calib
#include <hip/hip_runtime.h>
#include <cstdio>
template <typename T>
__global__ void rd(const T* __restrict__ p, T* __restrict__ out, size_t n) {
    T acc = T();
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        T v = p[i];
        unsigned* a = reinterpret_cast<unsigned*>(&acc); const unsigned* b = reinterpret_cast<const unsigned*>(&v);
        for (unsigned k = 0; k < sizeof(T) / 4; ++k) a[k] ^= b[k];
    }
    if (reinterpret_cast<unsigned*>(&acc)[0] == 0x12345678u) out[0] = acc;
}
__global__ void wr(uint4* __restrict__ p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = make_uint4(1, 2, 3, (unsigned)i);
}
int main() {
    const size_t bytes = 1ull << 30;      // 1 GiB: beyond L2 and the 256 MB Infinity Cache
    void *a, *o;
    hipMalloc(&a, bytes); hipMalloc(&o, 64);
    hipMemset(a, 1, bytes);
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(rd<unsigned>, dim3(4096), dim3(256), 0, 0, (const unsigned*)a, (unsigned*)o, bytes / 4);
        hipLaunchKernelGGL(rd<uint2>, dim3(4096), dim3(256), 0, 0, (const uint2*)a, (uint2*)o, bytes / 8);
        hipLaunchKernelGGL(rd<uint4>, dim3(4096), dim3(256), 0, 0, (const uint4*)a, (uint4*)o, bytes / 16);
        hipLaunchKernelGGL(wr, dim3(4096), dim3(256), 0, 0, (uint4*)a, bytes / 16);
    }
    hipDeviceSynchronize();
    printf("each kernel moves %zu bytes\n", bytes);
    return 0;
}
