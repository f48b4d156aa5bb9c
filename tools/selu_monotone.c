/* tools/selu_monotone.c -- exhaustive check that the canonical SELU (oracle/cv_oracle.c cvo_selu = csrc/cv_math.hpp
 * cvm::selu bit for bit) is monotone non-decreasing over ALL fp32 inputs.
 *
 * Why it matters: max-pooling commutes with a monotone activation, max_j selu(a_j + b) == selu(max_j a_j + b)
 * (x -> x + b is monotone in IEEE arithmetic), so the conv kernels may pool the raw accumulators and evaluate the
 * 19-instruction SELU once per POOLED row instead of once per conv row -- bit-identical iff no pair a < b exists with
 * selu(a) > selu(b).  x >= 0 is a rounded multiplication by a positive constant (monotone); the 2^31 negative inputs
 * are swept here.  Also reports how often selu(a) == selu(b) for neighbours (plateaus are fine) and checks the seam
 * selu(-tiny) <= selu(+0).
 *
 *   gcc -O2 -fopenmp -ffp-contract=off tools/selu_monotone.c -Loracle -lcv_oracle -Wl,-rpath,$PWD/oracle -o /tmp/selu_monotone
 */
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <math.h>

float cvo_selu_scalar(float x);

static inline float as_float(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

int main(void)
{
    /* negative floats by increasing magnitude: 0x80000000 (-0) .. 0xFF800000 (-inf); selu must be non-increasing */
    const uint64_t lo = 0x80000000ull, hi = 0xFF800000ull;
    const uint64_t chunk = 1u << 20;
    long long viol = 0, worst_ulps = 0;
    uint32_t first_bad = 0;
#pragma omp parallel for schedule(dynamic) reduction(+ : viol) reduction(max : worst_ulps)
    for (uint64_t c = lo; c <= hi; c += chunk) {
        uint64_t end = c + chunk <= hi ? c + chunk : hi;
        float prev = cvo_selu_scalar(as_float((uint32_t)c));
        for (uint64_t u = c + 1; u <= end; u++) {          /* overlaps the next chunk's first element */
            float cur = cvo_selu_scalar(as_float((uint32_t)u));
            if (cur > prev) {                               /* x decreased, selu increased */
                viol++;
                uint32_t a, b; memcpy(&a, &cur, 4); memcpy(&b, &prev, 4);
                long long d = (long long)(a & 0x7fffffff) - (long long)(b & 0x7fffffff);
                if (d < 0) d = -d;
                if (d > worst_ulps) worst_ulps = d;
#pragma omp critical
                if (!first_bad) first_bad = (uint32_t)u;
            }
            prev = cur;
        }
    }
    float seam_neg = cvo_selu_scalar(as_float(0x80000001u)), seam_zero = cvo_selu_scalar(0.0f), seam_mzero = cvo_selu_scalar(-0.0f);
    printf("negative inputs swept: %llu\n", (unsigned long long)(hi - lo + 1));
    printf("monotonicity violations: %lld (worst %lld ulp)\n", viol, worst_ulps);
    if (first_bad) printf("one violation at x = %.9g (0x%08x)\n", as_float(first_bad), first_bad);
    printf("seam: selu(-denorm_min) = %g, selu(-0) = %g, selu(+0) = %g, selu(-inf) = %.9g\n", seam_neg, seam_mzero, seam_zero,
           cvo_selu_scalar(-INFINITY));
    return viol != 0 || seam_neg > seam_zero;
}
