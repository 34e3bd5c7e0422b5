// sm_pow.h -- float pow evaluated in registers, shared by the gfx950 kernels
// and (compiled for the host by tests/test_pow_host.py) by the CPU check of
// the algorithm itself.
//
// Stands in for PowOp<float>::apply = std::pow(float, float)
// (reference include/math/pow.h:8-10); the reference has no vector body for it
// (pow.h:12-13 undefined, :16-32 commented out).  Parity bar: <= 4 ULP of the
// correctly rounded result (BASELINE north_star); this evaluation stays within
// 1 ULP, so it is interchangeable with glibc powf under that bar.
//
// Method: x^y = 2^(y * log2|x|) with the whole exponent chain in fp64.
//   log2|x| = e + t * P(t^2),  t = (m-1)/(m+1),  m in [~sqrt(1/2), ~sqrt(2))
//             P = (2/ln2) * sum_{k<=6} w^k/(2k+1): the factor t is exact, so the
//             error is RELATIVE to log2(m) (< 2^-39) -- it stays harmless when
//             x is near 1 and y is huge;
//             1/(m+1) = v_rcp_f64 seed (~2^-23) + one Newton step (~2^-46)
//   2^E     = 2^n * exp(f*ln2),  n = rint(E), |f| <= 1/2, degree-9 Taylor (< 2^-36)
//   one rounding to f32 at the end (v_cvt_f32_f64), which also yields
//   subnormal results, 0 and +inf correctly.
// No tables, no MFMA (nothing to contract): ~31 fp64-rate VALU ops + 1 rcp per
// element.  Everything is straight-line: W elements are evaluated side by side
// (pow_n<W>) so each polynomial constant is materialised once per step and the
// special cases are selects, not branches.  The special-case lattice is C99
// F.9.4.4 / IEEE 754-2008 9.2.1 as glibc implements it (x^0 = 1 and 1^y = 1
// even for quiet NaN, not for signalling NaN).
#pragma once

#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define SM_POW_FN __host__ __device__ __forceinline__
#else
#define SM_POW_FN static inline
#endif
#if defined(__HIP_DEVICE_COMPILE__)
#define SM_POW_FMA(a, b, c) __builtin_fma((a), (b), (c))
#define SM_POW_RINT(x) __builtin_rint(x)
// v_rcp_f64: ~2^-23 relative; one Newton step brings it to ~2^-46
#define SM_POW_RCP_SEED(x) __builtin_amdgcn_rcp(x)
#else
#include <math.h>
#define SM_POW_FMA(a, b, c) fma((a), (b), (c))
#define SM_POW_RINT(x) rint(x)
// host stand-in with the hardware seed's accuracy (24 bits), so the CPU check
// exercises the same Newton refinement
#define SM_POW_RCP_SEED(x) ((double)(float)(1.0 / (x)))
#endif

namespace smpow {

// Polynomial constants.  On the device they sit in constant memory: a uniform
// s_load puts them in SGPRs, and v_fma_f64 takes an SGPR pair as its addend,
// so a Horner step is ONE VALU instruction (as VGPR immediates each step would
// cost two extra v_mov_b32 -- as much issue time as the fma itself).
//   [0..6]  (2/ln2) / (2k+1), k = 6 .. 0      log2 series in w = t^2
//   [7]     ln2
//   [8..15] 1/9!, 1/8!, ... 1/2!               exp series in g = f*ln2
// (not `const` on the device, or the compiler folds the values back into immediates)
#if defined(__HIPCC__)
inline __constant__ double kC[16] = {
#else
static const double kC[16] = {
#endif
    0.22195308321368667, 0.2623081892525388, 0.3205988979753252, 0.4121985831111324,
    0.5770780163555853,  0.9617966939259756, 2.8853900817779268, 0.6931471805599453,
    2.7557319223985893e-06, 2.48015873015873e-05, 0.0001984126984126984, 0.001388888888888889,
    0.008333333333333333, 0.041666666666666664, 0.16666666666666666, 0.5};

SM_POW_FN uint32_t f32_bits(float x) { uint32_t u; memcpy(&u, &x, 4); return u; }
SM_POW_FN float bits_f32(uint32_t u) { float x; memcpy(&x, &u, 4); return x; }
SM_POW_FN uint64_t f64_bits(double x) { uint64_t u; memcpy(&u, &x, 8); return u; }
SM_POW_FN double bits_f64(uint64_t u) { double x; memcpy(&x, &u, 8); return x; }
SM_POW_FN double make_f64(uint32_t hi, uint32_t lo) { return bits_f64(((uint64_t)hi << 32) | lo); }

// 0: not an integer, 1: odd integer, 2: even integer (y finite, non-zero). Branch-free.
SM_POW_FN int int_class(uint32_t iy) {
    const int e = (int)((iy >> 23) & 0xff);
    int sh = 150 - e;                       // fractional bits of |y| when 127 <= e <= 150
    sh = sh < 0 ? 0 : (sh > 31 ? 31 : sh);
    const uint32_t frac = iy & ((1u << sh) - 1u);
    const int odd = (int)((iy >> sh) & 1u);
    const int whole = frac ? 0 : (odd ? 1 : 2);
    return e < 127 ? 0 : (e > 150 ? 2 : whole);
}

// 2^(y * log2(ax)) for W finite positive ax (subnormals included) and finite y,
// rounded to f32.  Straight-line; the loops are over the W independent elements.
template <int W>
SM_POW_FN void pow_core_n(const float (&ax)[W], const float (&y)[W], float (&out)[W]) {
    double m[W], t[W], w[W], p[W], E[W], g[W], q[W];
    int e[W];
#pragma unroll
    for (int k = 0; k < W; ++k) {
        // exact widening; f32 subnormals become normal doubles
        const uint64_t db = f64_bits((double)ax[k]);
        const uint32_t hi = (uint32_t)(db >> 32), lo = (uint32_t)db;
        const uint32_t mant = hi & 0x000fffffu;
        const uint32_t big = mant > 0x0006a09eu ? 1u : 0u;       // m > ~sqrt(2): use m/2, e+1
        e[k] = (int)(hi >> 20) - 1023 + (int)big;
        m[k] = make_f64(mant | (big ? 0x3fe00000u : 0x3ff00000u), lo);
    }
#pragma unroll
    for (int k = 0; k < W; ++k) {
        const double den = m[k] + 1.0;
        double r = SM_POW_RCP_SEED(den);
        r = SM_POW_FMA(r, SM_POW_FMA(-den, r, 1.0), r);
        t[k] = (m[k] - 1.0) * r;
        w[k] = t[k] * t[k];
    }
#define SM_POW_STEP(acc, x, c) for (int k = 0; k < W; ++k) acc[k] = SM_POW_FMA(acc[k], x[k], c)
#pragma unroll
    for (int k = 0; k < W; ++k) p[k] = kC[0];
#pragma unroll
    SM_POW_STEP(p, w, kC[1]);
#pragma unroll
    SM_POW_STEP(p, w, kC[2]);
#pragma unroll
    SM_POW_STEP(p, w, kC[3]);
#pragma unroll
    SM_POW_STEP(p, w, kC[4]);
#pragma unroll
    SM_POW_STEP(p, w, kC[5]);
#pragma unroll
    SM_POW_STEP(p, w, kC[6]);
#pragma unroll
    for (int k = 0; k < W; ++k) {
        const double lg = SM_POW_FMA(t[k], p[k], (double)e[k]);  // log2(ax)
        E[k] = (double)y[k] * lg;
        const double n = SM_POW_RINT(E[k]);
        g[k] = (E[k] - n) * kC[7];                               // f * ln2, |g| <= 0.3466
        E[k] = n;
    }
#pragma unroll
    for (int k = 0; k < W; ++k) q[k] = kC[8];
#pragma unroll
    SM_POW_STEP(q, g, kC[9]);
#pragma unroll
    SM_POW_STEP(q, g, kC[10]);
#pragma unroll
    SM_POW_STEP(q, g, kC[11]);
#pragma unroll
    SM_POW_STEP(q, g, kC[12]);
#pragma unroll
    SM_POW_STEP(q, g, kC[13]);
#pragma unroll
    SM_POW_STEP(q, g, kC[14]);
#pragma unroll
    SM_POW_STEP(q, g, kC[15]);
#pragma unroll
    SM_POW_STEP(q, g, 1.0);
#pragma unroll
    SM_POW_STEP(q, g, 1.0);
#undef SM_POW_STEP
#pragma unroll
    for (int k = 0; k < W; ++k) {
        // |E| beyond +-300 is 0 / inf in f32 anyway; the clamp keeps 2^n * q a normal double
        const double nc = E[k] > 300.0 ? 300.0 : (E[k] < -300.0 ? -300.0 : E[k]);
        const int n = (int)nc;
        const uint64_t qb = f64_bits(q[k]);                      // q in [0.70, 1.42]
        const double scaled = make_f64((uint32_t)(qb >> 32) + ((uint32_t)n << 20), (uint32_t)qb);
        out[k] = (float)scaled;
    }
}

// x^y for W independent (x, y) pairs.
template <int W>
SM_POW_FN void pow_n(const float (&x)[W], const float (&y)[W], float (&out)[W]) {
    const uint32_t ONE = 0x3f800000u, INF = 0x7f800000u, QNAN = 0x7fc00000u;
    float axc[W], yc_f[W], core[W];
#pragma unroll
    for (int k = 0; k < W; ++k) {
        const uint32_t ax = f32_bits(x[k]) & 0x7fffffffu, ay = f32_bits(y[k]) & 0x7fffffffu;
        // keep the core's inputs finite and positive; special lanes are overwritten below
        axc[k] = bits_f32((ax == 0 || ax >= INF) ? ONE : ax);
        yc_f[k] = ay >= INF ? 1.0f : y[k];
    }
    pow_core_n<W>(axc, yc_f, core);
#pragma unroll
    for (int k = 0; k < W; ++k) {
        const uint32_t ix = f32_bits(x[k]), iy = f32_bits(y[k]);
        const uint32_t ax = ix & 0x7fffffffu, ay = iy & 0x7fffffffu;
        const bool x_neg = (ix >> 31) != 0, y_neg = (iy >> 31) != 0;
        const bool x_nan = ax > INF, y_nan = ay > INF;
        const bool x_one = ix == ONE, y_zero = ay == 0;
        const int yc = int_class(iy);
        const uint32_t sign = (x_neg && yc == 1) ? 0x80000000u : 0u;
        uint32_t r = f32_bits(core[k]) | sign;
        r = (x_neg && yc == 0) ? QNAN : r;                                   // negative base, non-integer y
        r = ax == INF ? (sign | (y_neg ? 0u : INF)) : r;                     // (+-inf)^y
        r = ax == 0 ? (sign | (y_neg ? INF : 0u)) : r;                       // (+-0)^y
        r = ay == INF ? (ax == ONE ? ONE : (((ax < ONE) == y_neg) ? INF : 0u)) : r;  // x^(+-inf)
        r = (x_one || y_zero) ? ONE : r;                                     // 1^y = x^0 = 1
        const bool snan = (x_nan && !(ix & 0x00400000u)) || (y_nan && !(iy & 0x00400000u));
        const uint32_t nan_r = snan ? QNAN : ((x_one || y_zero) ? ONE : QNAN);
        r = (x_nan || y_nan) ? nan_r : r;
        out[k] = bits_f32(r);
    }
}

SM_POW_FN float powf(float x, float y) {
    const float xs[1] = {x}, ys[1] = {y};
    float r[1];
    pow_n<1>(xs, ys, r);
    return r[0];
}

}  // namespace smpow
