// Dropout mask of the flash-attention kernels (attn.hip, attn_bwd.hip): element (bh, q, key) of the probability matrix is KEPT iff
// attn_drop_hash(attn_drop_seed(seed, bh), q * npad + key) >= thr, thr = p * 2^32 -- a counter-based 32-bit hash (two multiplies), the same
// in the forward and in both backward kernels, so no mask is ever stored.  tests/test_unet_gpu.py restates it in numpy.
#pragma once
#include <stdint.h>

__host__ __device__ __forceinline__ uint32_t attn_drop_hash(uint32_t seed_bh, uint32_t e) {
    uint32_t x = e ^ seed_bh;
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
__host__ __device__ __forceinline__ uint32_t attn_drop_seed(uint32_t seed_lo, uint32_t seed_hi, uint32_t bh) {
    return attn_drop_hash(seed_hi, seed_lo ^ (bh * 0x9E3779B9u));
}
static inline uint32_t attn_drop_threshold(float p_drop) {          // P(hash < thr) = p
    const double t = (double)p_drop * 4294967296.0;
    return t <= 0.0 ? 0u : (t >= 4294967295.0 ? 4294967295u : (uint32_t)t);
}
