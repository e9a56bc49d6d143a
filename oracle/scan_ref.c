/* ORACLE -- test infrastructure, not product code.
 *
 * Plain-C restatement of the reference's *sequential* selective scan, the definition of correctness for the
 * fused kernel: cross_atten/mamba.py:288-318 (MambaBlock.selective_scan_seq) with the softplus(delta + bias)
 * of mamba.py:255-256 and the y * silu(z) gate of mamba.py:220-222.  Token-major layout:
 *   u, delta, z, y: (B, L, ED)   Bm, Cm: (B, L, N)   A: (ED, N)   D, bias: (ED).
 * Pinned by tests/test_oracle_golden.py::test_c_scan_oracle against tests/golden/t0_selective_scan.npz
 * (outputs of the reference).  Used by tests/ and as bench.py's cpu_baseline ("port", 1 core).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

static float softplus_(float x) { return x > 20.0f ? x : log1pf(expf(x)); }

int gfe_oracle_selective_scan(const float* u, const float* delta, const float* A, const float* Bm, const float* Cm,
                              const float* D, const float* z, const float* bias, float* y,
                              int64_t B, int64_t L, int64_t ED, int64_t N, int softplus) {
    float* h = (float*)malloc(sizeof(float) * (size_t)N);
    if (!h) return -1;
    for (int64_t b = 0; b < B; ++b)
        for (int64_t e = 0; e < ED; ++e) {
            for (int64_t n = 0; n < N; ++n) h[n] = 0.0f;
            for (int64_t t = 0; t < L; ++t) {
                const int64_t o = (b * L + t) * ED + e;
                float dt = delta[o] + (bias ? bias[e] : 0.0f);
                if (softplus) dt = softplus_(dt);
                const float x = u[o];
                float acc = 0.0f;
                for (int64_t n = 0; n < N; ++n) {
                    const float dA = expf(dt * A[e * N + n]);                 /* mamba.py:300 */
                    const float bx = dt * Bm[(b * L + t) * N + n] * x;        /* mamba.py:301-303 */
                    h[n] = dA * h[n] + bx;                                    /* mamba.py:309 */
                    acc += h[n] * Cm[(b * L + t) * N + n];                    /* mamba.py:314 */
                }
                float yv = acc + (D ? D[e] : 0.0f) * x;                      /* mamba.py:316 */
                if (z) { const float zv = z[o]; yv *= zv / (1.0f + expf(-zv)); }
                y[o] = yv;
            }
        }
    free(h);
    return 0;
}
