// Library identity + tiny device utilities shared by the C-ABI (include/gfe_hip.h).
#include "common.h"

extern "C" {
int gfe_abi_version(void) { return GFE_ABI_VERSION; }
const char* gfe_build_arch(void) { return "gfx950"; }

// A HIP stream whose kernels run only on the CUs of `mask` (hipExtStreamCreateWithCUMask).  Bit i of the mask = CU slot i / 8 of XCD i % 8
// (measured with tools/probes/cu_mask_probe.hip: bits 0..15 are two CUs of every XCD), so gfe_stream_create_cu_range(lo, hi) with
// multiples of 8 takes the same CUs from every XCD.  The two-stream step gives the head CUs of its own this way (gfe_hip/step.py).
int gfe_stream_create_cu_range(int lo, int hi, void** stream) {
    GFE_REQUIRE(stream && lo >= 0 && hi > lo && hi <= 256, GFE_ERR_SHAPE);
    uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = lo; i < hi; ++i) mask[i >> 5] |= 1u << (i & 31);
    hipStream_t st = nullptr;
    if (hipExtStreamCreateWithCUMask(&st, 8, mask) != hipSuccess) { (void)hipGetLastError(); return GFE_ERR_HIP; }
    *stream = (void*)st;
    return 0;
}
int gfe_stream_destroy(void* stream) {
    GFE_REQUIRE(stream, GFE_ERR_NULL);
    return hipStreamDestroy((hipStream_t)stream) == hipSuccess ? 0 : GFE_ERR_HIP;
}
}
