/* Host check of the division sequence the streaming kernels use for launch-uniform divisors:
 *   y = RN(1/b) (host), q0 = a*y, r0 = fma(-b,q0,a), q1 = fma(r0,y,q0), r1 = fma(-b,q1,a), q = fma(r1,y,q1)
 * against the correctly rounded a / b, over every significand of a (two binades) for a list of divisors,
 * plus random pairs inside the exponent window the library admits (|a| in [2^-60, 2^64], |b| in [2^-40, 2^40]).  gcc -O2 -fopenmp -ffp-contract=off divcheck.c -lm */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static float from_bits(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static float refined(float a, float b, float y)
{
    float q = a * y;
    float r = fmaf(-b, q, a);
    q = fmaf(r, y, q);
    r = fmaf(-b, q, a);
    return fmaf(r, y, q);
}
static uint64_t splitmix(uint64_t* s) { uint64_t z = (*s += 0x9E3779B97F4A7C15ull); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }

int main(int argc, char** argv)
{
    long bad = 0;
    float divisors[64]; int nd = 0;
    const float fixed[] = {20.0f, 1.0f, 0.5f, 2.0f, 1.5f, 0.75f, 0.01f, 10.0f, 3.0f, 7.0f, 1.9999999f, 1.0000001f, 0.1f, 5.0f, 0.25f, 6.2831855f};
    for (unsigned i = 0; i < sizeof fixed / sizeof *fixed; ++i) divisors[nd++] = fixed[i];
    uint64_t seed = 12345;
    while (nd < 48) { uint32_t m = (uint32_t)splitmix(&seed) & 0x7FFFFFu; divisors[nd++] = from_bits(0x3F800000u | m); }
    for (int d = 0; d < nd; ++d) {
        const float b = divisors[d], y = (float)(1.0 / (double)b);
        long local = 0;
#pragma omp parallel for reduction(+ : local)
        for (uint32_t m = 0; m < (1u << 24); ++m) {
            const float a = from_bits(0x3F000000u + m);   /* [0.5, 2): every significand in two binades */
            const float want = a / b, got = refined(a, b, y);
            if (memcmp(&want, &got, 4) != 0) ++local;
        }
        if (local) printf("divisor %.9g: %ld mismatches\n", b, local);
        bad += local;
    }
    /* random pairs: |a| in [2^-60, 2^64], |b| in [2^-40, 2^40], both signs */
    long n_rand = argc > 1 ? atol(argv[1]) : 200000000L, local = 0;
#pragma omp parallel for reduction(+ : local)
    for (long i = 0; i < n_rand; ++i) {
        uint64_t s = 0xABCDEF12345ull + (uint64_t)i * 0x9E3779B97F4A7C15ull;
        const uint64_t r1 = splitmix(&s), r2 = splitmix(&s);
        const uint32_t ea = 127 - 60 + (uint32_t)(r1 >> 40) % 125, eb = 127 - 40 + (uint32_t)(r2 >> 40) % 81;
        const float a = from_bits(((uint32_t)(r1 >> 63) << 31) | (ea << 23) | ((uint32_t)r1 & 0x7FFFFFu));
        const float b = from_bits(((uint32_t)(r2 >> 63) << 31) | (eb << 23) | ((uint32_t)r2 & 0x7FFFFFu));
        const float y = (float)(1.0 / (double)b);
        const float want = a / b, got = refined(a, b, y);
        if (memcmp(&want, &got, 4) != 0) ++local;
    }
    if (local) printf("random pairs: %ld mismatches of %ld\n", local, n_rand);
    bad += local;
    printf("%s: %d divisors x 2^24 numerators, %ld random pairs, %ld mismatches\n", bad ? "FAIL" : "ok", nd, n_rand, bad);
    return bad != 0;
}
