/* tools/sincos_sweep.c -- how far is the sin/cos parity contract from the platform's libm?
 *
 * ORBextractor.cc:112-113 calls cos(float)/sin(float) = glibc cosf/sinf; the project's contract
 * (oracle/orb_oracle.c orc_sincos_f == pilotguru_amd/csrc/describe.hip pg_sincos_f) is a fixed double-precision
 * sequence rounded once.  This tool measures the difference against THIS machine's glibc:
 *   (1) every float in [0, 2*pi] (all 1.09e9 bit patterns): how many sin / cos results differ, by how many ulp;
 *   (2) what matters: every float angle in [0, 360) degrees -- a superset of what cv::fastAtan2 can return --
 *       taken through computeOrbDescriptor's arithmetic (angle * factorPI in float, ORBextractor.cc:110-120):
 *       for how many angles does ANY of the 512 pattern points land on a different pixel, and how many taps move.
 * Build:  gcc -O2 -fopenmp -ffp-contract=off -I oracle tools/sincos_sweep.c oracle/orb_oracle.c oracle/bow_oracle.c \
 *             oracle/match_oracle.c oracle/post_oracle.c oracle/calib_oracle.c -lm -o /tmp/sincos_sweep
 * The result of the round-2 run is kept in profiles/r02_sincos_sweep.txt.  TEST INFRASTRUCTURE (links the oracle). */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <gnu/libc-version.h>
#include "orb_oracle.h"

static const int8_t k_pat[1024] = {
#include "orb_pattern31.inc"
};

static inline int cv_round_d(double v) { return (int)lrint(v); }

int main(void)
{
    union { uint32_t u; float f; } lim;
    lim.f = 6.2831855f;                                   /* (float)(2*pi), rounded up */
    const uint32_t n1 = lim.u + 1;
    long long dsin = 0, dcos = 0, maxulp = 0;
#pragma omp parallel for schedule(static) reduction(+ : dsin, dcos) reduction(max : maxulp)
    for (uint32_t i = 0; i < n1; i++) {
        union { uint32_t u; float f; } in, s1, c1, s2, c2;
        in.u = i;
        orc_sincos_f(in.f, &s1.f, &c1.f);
        s2.f = sinf(in.f); c2.f = cosf(in.f);
        if (s1.u != s2.u) { dsin++; long long d = llabs((long long)(int32_t)s1.u - (long long)(int32_t)s2.u); if (d > maxulp && d < 1000) maxulp = d; }
        if (c1.u != c2.u) { dcos++; long long d = llabs((long long)(int32_t)c1.u - (long long)(int32_t)c2.u); if (d > maxulp && d < 1000) maxulp = d; }
    }
    printf("glibc %s\n", gnu_get_libc_version());
    printf("(1) floats in [0, 2*pi]: %u values; sinf differs from the contract for %lld (%.3g), cosf for %lld (%.3g); largest difference %lld ulp\n",
           n1, dsin, (double)dsin / n1, dcos, (double)dcos / n1, maxulp);

    lim.f = 360.0f;
    const uint32_t n2 = lim.u;                            /* every float in [0, 360) */
    const float factorPI = (float)(3.1415926535897932384626433832795 / 180.f);
    long long angDiff = 0, angMoved = 0, tapsMoved = 0;
#pragma omp parallel for schedule(static) reduction(+ : angDiff, angMoved, tapsMoved)
    for (uint32_t i = 0; i < n2; i++) {
        union { uint32_t u; float f; } deg, s1, c1, s2, c2;
        deg.u = i;
        const float angle = deg.f * factorPI;
        orc_sincos_f(angle, &s1.f, &c1.f);
        s2.f = sinf(angle); c2.f = cosf(angle);
        if (s1.u == s2.u && c1.u == c2.u) continue;
        angDiff++;
        int moved = 0;
        for (int k = 0; k < 512; k++) {
            const int px = k_pat[2 * k], py = k_pat[2 * k + 1];
            const float r1 = px * s1.f, r1b = py * c1.f, q1 = px * c1.f, q1b = py * s1.f;
            const float r2 = px * s2.f, r2b = py * c2.f, q2 = px * c2.f, q2b = py * s2.f;
            if (cv_round_d((double)(r1 + r1b)) != cv_round_d((double)(r2 + r2b))) moved++;
            if (cv_round_d((double)(q1 - q1b)) != cv_round_d((double)(q2 - q2b))) moved++;
        }
        if (moved) { angMoved++; tapsMoved += moved; }
    }
    printf("(2) float angles in [0, 360) deg: %u values; sin or cos differs for %lld (%.3g); a pattern tap lands on another pixel for %lld angles "
           "(%.3g of all angles), %lld tap coordinates in total (of %u x 1024)\n",
           n2, angDiff, (double)angDiff / n2, angMoved, (double)angMoved / n2, tapsMoved, n2);
    return 0;
}
