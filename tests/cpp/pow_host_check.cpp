// Host check of simplemath_amd/csrc/sm_pow.h (the same source the gfx950
// kernels inline): max ULP distance from the correctly rounded x^y over a
// dense sweep, plus the C99 special-case lattice against the host libm.
// Prints "max_ulp <n> over <count>" and "lattice_mismatches <n>".
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>

#include "sm_pow.h"

static int64_t ord(float f) { uint32_t u; memcpy(&u, &f, 4); return (u & 0x80000000u) ? -(int64_t)(u & 0x7fffffffu) : (int64_t)u; }
static uint64_t mix(uint64_t x) { x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ULL; x ^= x >> 27; x *= 0x94D049BB133111EBULL; x ^= x >> 31; return x; }

int main() {
    const float ys[] = {2.5f, 2.0f, 3.0f, 0.5f, -1.0f, -2.5f, 1.5f, 7.0f, -3.0f, 0.33333334f, 10.25f, 100.0f, -100.0f, 1e-3f,
                        37.75f, -0.001f, 1e6f, -1e6f, 8388608.0f, 1.0000001f, 0.99999994f, 123456.7f};
    int64_t worst = 0; uint64_t count = 0; float wx = 0, wy = 0;
    for (float y : ys) {
        for (uint64_t i = 0; i < 400000; ++i) {
            uint64_t h = mix(i * 0x9E3779B97F4A7C15ULL + (uint64_t)(y * 1000));
            float x;
            switch (i & 3) {
                case 0: { uint32_t u = (uint32_t)h & 0x7fffffffu; memcpy(&x, &u, 4); break; }          // any positive bit pattern
                case 1: x = 0.01f + (float)((h >> 40) * 0x1.0p-24) * 99.99f; break;                   // config-4 range
                case 2: x = 1.0f + ((float)((h >> 40) * 0x1.0p-24) - 0.5f) * 1e-3f; break;            // near 1
                default: { uint32_t u = ((uint32_t)h & 0x007fffffu); memcpy(&x, &u, 4); break; }      // subnormal
            }
            if (!(x == x) || std::isinf(x)) continue;
            float got = smpow::powf(x, y);
            float want = (float)std::pow((double)x, (double)y);
            int64_t d = llabs(ord(got) - ord(want));
            if (d > worst) { worst = d; wx = x; wy = y; }
            ++count;
        }
    }
    printf("max_ulp %lld over %llu (x=%a y=%a)\n", (long long)worst, (unsigned long long)count, wx, wy);

    const float inf = std::numeric_limits<float>::infinity(), nan = std::numeric_limits<float>::quiet_NaN();
    const float sp[] = {0.0f, -0.0f, 1.0f, -1.0f, inf, -inf, nan, 0.5f, -0.5f, 2.0f, -2.0f, 3.0f, -3.0f, 4.0f, -4.0f, 1e-45f, -1e-45f,
                        3.4028235e38f, -3.4028235e38f, 1.5f, -1.5f, 8388608.0f, 8388609.0f, 16777216.0f, 16777218.0f, -8388609.0f, 1e10f, -1e10f,
                        0.99999994f, 1.0000001f, -0.99999994f, -1.0000001f};
    int bad = 0;
    for (float x : sp) for (float y : sp) {
        float got = smpow::powf(x, y), want = std::pow(x, y);
        bool ok = (got != got && want != want) || (ord(got) == ord(want) && std::signbit(got) == std::signbit(want)) ||
                  (llabs(ord(got) - ord(want)) <= 1 && std::isfinite(want) && want != 0.0f);
        if (!ok) { ++bad; printf("lattice x=%a y=%a got=%a want=%a\n", x, y, got, want); }
    }
    printf("lattice_mismatches %d\n", bad);
    return 0;
}
