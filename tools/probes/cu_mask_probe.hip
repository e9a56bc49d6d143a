// Which CUs does a CU-masked stream (hipExtStreamCreateWithCUMask) run on?  Per mask: launch many short spinning blocks, record every
// block's (XCC_ID, SE, SH, CU) from the hardware registers and print the distinct set.  hipcc --offload-arch=gfx950 cu_mask_probe.hip -o cu_mask_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <set>
#include <vector>
#include <string>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__global__ void where_kernel(unsigned* out, int spin) {
    const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20);      // HW_REG_XCC_ID[3:0]
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);       // HW_REG_HW_ID
    const long long t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < spin) {}
    if (threadIdx.x == 0) out[blockIdx.x] = (xcc << 16) | (hw & 0xffff);
}
static void run(const char* name, const std::vector<uint32_t>& mask, bool use_mask) {
    hipStream_t st;
    if (use_mask) CK(hipExtStreamCreateWithCUMask(&st, (uint32_t)mask.size(), mask.data()));
    else CK(hipStreamCreate(&st));
    const int nb = 8192;
    unsigned* d; CK(hipMalloc(&d, nb * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(where_kernel, dim3(nb), dim3(256), 0, st, d, 2000);
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    hipLaunchKernelGGL(where_kernel, dim3(nb), dim3(256), 0, st, d, 2000);
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned> h(nb); CK(hipMemcpy(h.data(), d, nb * 4, hipMemcpyDeviceToHost));
    std::set<unsigned> cus; int per_xcc[16] = {0};
    for (unsigned v : h) { const unsigned key = (v >> 16) << 16 | (v & 0xff00); cus.insert(key); }
    for (unsigned k : cus) per_xcc[(k >> 16) & 15]++;
    printf("%-28s %3zu distinct CUs, %.3f ms; per XCC:", name, cus.size(), ms);
    for (int i = 0; i < 8; ++i) printf(" %d", per_xcc[i]);
    printf("\n");
    if (cus.size() <= 40) { printf("   (xcc,se,sh,cu):"); for (unsigned k : cus) printf(" (%u,%u,%u,%u)", k >> 16, (k >> 13) & 7, (k >> 12) & 1, (k >> 8) & 15); printf("\n"); }
    CK(hipFree(d)); CK(hipStreamDestroy(st));
}
int main() {
    run("no mask", {}, false);
    auto bits = [](int lo, int hi) { std::vector<uint32_t> m(8, 0); for (int i = lo; i < hi; ++i) m[i / 32] |= 1u << (i % 32); return m; };
    run("bits 0..15", bits(0, 16), true);
    run("bits 0..7", bits(0, 8), true);
    run("bits 8..15", bits(8, 16), true);
    run("bits 16..255", bits(16, 256), true);
    run("bits 0..31", bits(0, 32), true);
    run("bits 32..255", bits(32, 256), true);
    run("bits 240..255", bits(240, 256), true);
    // two streams at once: a long kernel on the big mask, short ones on the small mask
    return 0;
}
