// Host check of smpow64::pow_halfint (simplemath_amd/csrc/sm_pow64.h): scalar exponents -8 ... 8 in steps of one half,
// against glibc pow (< 1 ULP itself): max ULP distance and the share of identical results over random bases from the whole
// range (any bit pattern, config 4's range, near 1, subnormals, a log-uniform sweep), negative bases, and the lattice of
// zeros / infinities / NaN.  The seeds stand in for v_rsq_f64 / v_rcp_f64 with float-rounded values (coarser than the device's).
// Prints "max_ulp <n> over <count> identical <share>" and "lattice_mismatches <n>".
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <limits>

#include "sm_pow64.h"

static int64_t ord(double f) { int64_t u; memcpy(&u, &f, 8); return u < 0 ? std::numeric_limits<int64_t>::min() - u : u; }
static uint64_t mix(uint64_t x) { x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ULL; x ^= x >> 27; x *= 0x94D049BB133111EBULL; x ^= x >> 31; return x; }

int main(int argc, char **argv) {
    const uint64_t per = argc > 1 ? strtoull(argv[1], nullptr, 10) : 120000;
    int64_t worst = 0; uint64_t count = 0, same = 0; double wx = 0, wy = 0;
    for (int m2 = -16; m2 <= 16; ++m2) {
        if (m2 == 0) continue;
        const double y = m2 * 0.5;
        int back = 0;
        if (!smpow64::halfint_exponent(y, &back) || back != m2) { printf("halfint_exponent(%g) wrong\n", y); return 1; }
        for (uint64_t i = 0; i < per; ++i) {
            uint64_t h = mix(i * 0x9E3779B97F4A7C15ULL + (uint64_t)(m2 + 100));
            double x;
            switch (i % 6) {
                case 0: { uint64_t u = h & 0x7fffffffffffffffULL; memcpy(&x, &u, 8); break; }
                case 1: x = 0.01 + (double)(h >> 11) * 0x1.0p-53 * 99.99; break;
                case 2: x = 1.0 + ((double)(h >> 11) * 0x1.0p-53 - 0.5) * 1e-3; break;
                case 3: { uint64_t u = h & 0x000fffffffffffffULL; memcpy(&x, &u, 8); break; }
                case 4: x = -(0.01 + (double)(h >> 11) * 0x1.0p-53 * 99.99); break;                         // negative bases
                default: x = std::exp2((double)((int64_t)(h % 4200) - 2100) / 2.0) * (1.0 + (double)(h >> 40) * 0x1p-24); break;  // results across the whole exponent range, overflow and underflow included
            }
            if (!(x == x)) continue;
            const double got = smpow64::pow_halfint_switch<-16, 16>(x, m2), want = std::pow(x, y);  // the compile-time forms the kernels use
            const double loop = smpow64::pow_halfint(x, m2);                                        // the same chain with a run-time exponent
            if (memcmp(&got, &loop, 8) != 0 && !(got != got && loop != loop)) { printf("forms differ x=%a y=%g %a %a\n", x, y, got, loop); return 1; }
            if (got != got || want != want) { if ((got != got) != (want != want)) { printf("nan mismatch x=%a y=%g got=%a want=%a\n", x, y, got, want); return 1; } continue; }
            int64_t d = ord(got) - ord(want); if (d < 0) d = -d;
            if (d > worst) { worst = d; wx = x; wy = y; }
            same += d == 0;
            ++count;
        }
    }
    printf("max_ulp %lld over %llu identical %.6f (x=%a y=%g)\n", (long long)worst, (unsigned long long)count, (double)same / (double)count, wx, wy);
    const double inf = std::numeric_limits<double>::infinity(), nan = std::numeric_limits<double>::quiet_NaN();
    const double sp[] = {0.0, -0.0, 1.0, -1.0, inf, -inf, nan, 0.5, -0.5, 2.0, -2.0, 3.0, -3.0, 4.0, -4.0, 5e-324, -5e-324, 1.7976931348623157e308,
                         -1.7976931348623157e308, 2.2250738585072014e-308, -2.2250738585072014e-308, 1.5, -1.5, 1e10, -1e10, 1e-200, 1e200, 0.9999999999999999, 1.0000000000000002};
    int bad = 0;
    for (double x : sp) for (int m2 = -16; m2 <= 16; ++m2) {
        if (m2 == 0) continue;
        const double got = smpow64::pow_halfint_switch<-16, 16>(x, m2), want = std::pow(x, m2 * 0.5);
        int64_t d = ord(got) - ord(want); if (d < 0) d = -d;
        const bool ok = (got != got && want != want) || (ord(got) == ord(want) && std::signbit(got) == std::signbit(want)) ||
                        (d <= 1 && std::isfinite(want) && want != 0.0 && std::signbit(got) == std::signbit(want));
        if (!ok) { ++bad; printf("lattice x=%a y=%g got=%a want=%a\n", x, m2 * 0.5, got, want); }
    }
    printf("lattice_mismatches %d\n", bad);
    return 0;
}
