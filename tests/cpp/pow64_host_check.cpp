// Host check of simplemath_amd/csrc/sm_pow64.h: max ULP distance from glibc pow (< 1 ULP itself) over a
// dense sweep incl. subnormals, values near 1 with huge exponents, results near overflow / underflow; the
// special-case lattice against libm.  Prints "max_ulp <n> over <count>" and "lattice_mismatches <n>".
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>

#include "sm_pow64.h"

static int64_t ord(double f) { int64_t u; memcpy(&u, &f, 8); return u < 0 ? std::numeric_limits<int64_t>::min() - u : u; }
static uint64_t mix(uint64_t x) { x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ULL; x ^= x >> 27; x *= 0x94D049BB133111EBULL; x ^= x >> 31; return x; }

int main() {
    const double ys[] = {2.5, 2.0, 3.0, 0.5, -1.0, -2.5, 1.5, 7.0, -3.0, 1.0 / 3, 10.25, 100.0, -100.0, 1e-3, 37.75, -0.001, 1e6, -1e6,
                         4503599627370496.0, 1.0000000000000002, 0.9999999999999999, 123456.7, 700.0, -700.0, 1e-300, 0.1,
                         // exponents for which y * ln x overflows or the product's low part is garbage; subnormal exponents
                         1e300, -1e300, 1.7976931348623157e308, -1.7976931348623157e308, 18446744073709551616.0, 5e-324, -1e-310, 1e17, -3e15};
    int64_t worst = 0; uint64_t count = 0; double wx = 0, wy = 0;
    for (double y : ys) {
        for (uint64_t i = 0; i < 150000; ++i) {
            uint64_t h = mix(i * 0x9E3779B97F4A7C15ULL + (uint64_t)(int64_t)(y * 1000));
            double x;
            switch (i % 5) {
                case 0: { uint64_t u = h & 0x7fffffffffffffffULL; memcpy(&x, &u, 8); break; }                 // any positive bit pattern
                case 1: x = 0.01 + (double)(h >> 11) * 0x1.0p-53 * 99.99; break;                             // config-4 range
                case 2: x = 1.0 + ((double)(h >> 11) * 0x1.0p-53 - 0.5) * 1e-3; break;                       // near 1
                case 3: { uint64_t u = h & 0x000fffffffffffffULL; memcpy(&x, &u, 8); break; }                 // subnormal
                default: x = std::exp2((double)((int64_t)(h % 2000) - 1000) / 10.0) * (1.0 + (double)(h >> 40) * 0x1p-24); break;
            }
            if (!(x == x) || std::isinf(x)) continue;
            const double got = smpow64::pow(x, y, smpow64::kLogTab, smpow64::kExpTab), want = std::pow(x, y);
            int64_t d = ord(got) - ord(want); if (d < 0) d = -d;
            if (d > worst) { worst = d; wx = x; wy = y; }
            ++count;
        }
    }
    printf("max_ulp %lld over %llu (x=%a y=%a)\n", (long long)worst, (unsigned long long)count, wx, wy);
    const double inf = std::numeric_limits<double>::infinity(), nan = std::numeric_limits<double>::quiet_NaN();
    const double sp[] = {0.0, -0.0, 1.0, -1.0, inf, -inf, nan, 0.5, -0.5, 2.0, -2.0, 3.0, -3.0, 4.0, -4.0, 5e-324, -5e-324, 1.7976931348623157e308,
                         -1.7976931348623157e308, 1.5, -1.5, 4503599627370496.0, 4503599627370497.0, 9007199254740992.0, 9007199254740994.0,
                         -4503599627370497.0, 1e10, -1e10, 0.9999999999999999, 1.0000000000000002, -0.9999999999999999, -1.0000000000000002};
    int bad = 0;
    for (double x : sp) for (double y : sp) {
        const double got = smpow64::pow(x, y, smpow64::kLogTab, smpow64::kExpTab), want = std::pow(x, y);
        int64_t d = ord(got) - ord(want); if (d < 0) d = -d;
        const bool ok = (got != got && want != want) || (ord(got) == ord(want) && std::signbit(got) == std::signbit(want)) ||
                        (d <= 1 && std::isfinite(want) && want != 0.0);
        if (!ok) { ++bad; printf("lattice x=%a y=%a got=%a want=%a\n", x, y, got, want); }
    }
    printf("lattice_mismatches %d\n", bad);

    // One exponent of moderate magnitude for the whole array (smpow64::pow_scalar, LEVEL 2: |y| <= 16, LEVEL 1: |y| <= 1024): against
    // glibc, and the TRUE error in ULP against long double powl (64-bit significand) where the result is a normal number.
    const double ys1[] = {2.7, 0.3333, 7.5, 8.5, 16.0, -16.0, 1e-3, 0.1, 15.99, -13.37, 3.141592653589793, 1.0 / 3, -0.001, 1e-300, 2.2250738585072014e-308,
                          1.0000000000000002, 0.9999999999999999, 10.25, -7.77, 37.75, 100.0, -100.0, 700.0, -700.0, 1024.0, -1000.5, 123.456, 16.000000000000004,
                          511.3, -64.1, 17.0};
    int64_t sworst = 0; uint64_t scount = 0; double strue = 0, swx = 0, swy = 0; int levels[3] = {0, 0, 0}, sbad = 0;
    for (double y : ys1) {
        const int level = smpow64::scalar_level(y);
        ++levels[level];
        for (uint64_t i = 0; i < 120000; ++i) {
            uint64_t h = mix(i * 0x9E3779B97F4A7C15ULL + (uint64_t)(int64_t)(y * 1000));
            double x;
            switch (i % 5) {
                case 0: { uint64_t u = h & 0x7fffffffffffffffULL; memcpy(&x, &u, 8); break; }
                case 1: x = 0.01 + (double)(h >> 11) * 0x1.0p-53 * 99.99; break;
                case 2: x = 1.0 + ((double)(h >> 11) * 0x1.0p-53 - 0.5) * ((h & 1) ? 1e-3 : 0.05); break;
                case 3: { uint64_t u = h & 0x000fffffffffffffULL; memcpy(&x, &u, 8); break; }
                default: x = std::exp2((double)((int64_t)(h % 2000) - 1000) / 10.0) * (1.0 + (double)(h >> 40) * 0x1p-24); break;
            }
            if (!(x == x) || std::isinf(x)) continue;
            const double got = smpow64::pow_scalar(x, y, level, smpow64::kLogTab, smpow64::kExpTab), want = std::pow(x, y);
            int64_t d = ord(got) - ord(want); if (d < 0) d = -d;
            if (d > sworst) { sworst = d; swx = x; swy = y; }
            if (std::isfinite(want) && want > 1e-290) {
                const long double t = powl((long double)x, (long double)y);
                const double u = std::nextafter(want, INFINITY) - want;
                const double err = (double)fabsl(((long double)got - t) / u);
                if (err > strue) strue = err;
            }
            ++scount;
        }
        for (double x : sp) {  // the lattice's bases with this exponent
            const double got = smpow64::pow_scalar(x, y, level, smpow64::kLogTab, smpow64::kExpTab), want = std::pow(x, y);
            int64_t d = ord(got) - ord(want); if (d < 0) d = -d;
            const bool ok = (got != got && want != want) || (ord(got) == ord(want) && std::signbit(got) == std::signbit(want)) ||
                            (d <= 1 && std::isfinite(want) && want != 0.0);
            if (!ok) { ++sbad; printf("scalar lattice x=%a y=%a got=%a want=%a\n", x, y, got, want); }
        }
    }
    printf("scalar_max_ulp %lld over %llu (x=%a y=%a) true_err %.3f levels %d %d %d lattice_mismatches_scalar %d\n", (long long)sworst,
           (unsigned long long)scount, swx, swy, strue, levels[0], levels[1], levels[2], sbad);
    return 0;
}
