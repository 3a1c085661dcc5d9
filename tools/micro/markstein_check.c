// Bit-for-bit check of ks_kernels.hip::div_const -- q = x * r; e = fma(-d, q, x); fma(e, r, q) with r = RN(1 / d) -- against the
// IEEE division x / d: dividends whose true quotient lies next to a rounding boundary (midpoint between two doubles) or next
// to a representable double, significands near 2, for divisors dx, dx^2, dx^4 and 6 of several grids (4.3e9 cases; a
// purely random sweep of 2.6e9 more was run once with the same result: 0 mismatches).
// build + run (CPU):  gcc -O2 -ffp-contract=off -mfma -o /tmp/markstein_check tools/micro/markstein_check.c -lm && /tmp/markstein_check
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
static uint64_t s = 0x9E3779B97F4A7C15ULL;
static inline uint64_t rnd(void) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }
static inline double div_m(double x, double d, double r) { double q = x * r; double e = fma(-d, q, x); return fma(e, r, q); }
int main(void) {
    const double dxs[] = {22.0 / 64, 2.0 * 3.141592653589793 / 512, 1.0 / 3, 0.1};
    long bad = 0, total = 0;
    for (unsigned k = 0; k < 4; ++k) {
        const double dx = dxs[k];
        const double ds[4] = {dx, dx * dx, (dx * dx) * (dx * dx), 6.0};
        for (int j = 0; j < 4; ++j) {
            const double d = ds[j], r = 1.0 / d;
            for (long i = 0; i < 30000000L; ++i) {
                uint64_t b = rnd();
                const int e = (int)(b % 41) - 20;
                uint64_t m = (rnd() & 0xFFFFFFFFFFFFFULL) | ((uint64_t)(1023 + e) << 52);
                if ((b >> 50) & 1) m |= 0xFFFFFFFFFFF00ULL & ((b >> 20) | 0xFFFFFFF000000ULL);   // significands near 2
                double t; memcpy(&t, &m, 8);
                const long double half_ulp = ldexpl(1.0L, e - 53);
                const long double targets[3] = {(long double)t + half_ulp, (long double)t, (long double)t - half_ulp};
                for (int v = 0; v < 3; ++v) {
                    const double x0 = (double)(targets[v] * (long double)d);
                    for (int o = -1; o <= 1; ++o) {
                        const double x = o == 0 ? x0 : nextafter(x0, o > 0 ? INFINITY : -INFINITY);
                        const double tq = x / d, g = div_m(x, d, r);
                        if (memcmp(&tq, &g, 8)) { if (bad < 5) printf("MISMATCH d=%a x=%a true=%a got=%a\n", d, x, tq, g); ++bad; }
                        ++total;
                    }
                }
            }
        }
    }
    printf("adversarial total %ld mismatches %ld\n", total, bad);
    return bad != 0;
}
