// sm_pow.h -- float pow evaluated in registers, shared by the gfx950 kernels
// and (compiled for the host by tests/test_pow_host.py) by the CPU check of
// the algorithm itself.
//
// Stands in for PowOp<float>::apply = std::pow(float, float)
// (reference include/math/pow.h:8-10); the reference has no vector body for it
// (pow.h:12-13 undefined, :16-32 commented out).  Parity bar: <= 4 ULP of the
// correctly rounded result (BASELINE north_star); this evaluation stays within
// 2 ULP, so it is interchangeable with glibc powf under that bar.
//
// Method: x^y = 2^(y * log2|x|) with the whole exponent chain in fp64.
//   |x| = 2^e * m, m in [~sqrt(1/2), ~sqrt(2));  c = k/64 the breakpoint nearest m
//   log2|x| = e + logc[k] + r * Q(r),   r = m * invc[k] - 1,  |r| <= 2^-6.5
//             {invc, logc} from a 47-entry table (752 B, staged in LDS by the kernels);
//             the k = 64 entry is exactly {1, 0}, so around x = 1 the result
//             r * Q(r) keeps its RELATIVE accuracy (< 2^-41) -- harmless even
//             when x is near 1 and y is huge.  No division, no reciprocal.
//   E = y * log2|x| = n + f,  n = rint(E), |f| <= 1/2        (still fp64: this is where
//             a float exponent chain would lose the result's low bits)
//   2^f     = 1 + g*P(g),  g = f*ln2, degree-6 P, in f32 (the fraction only needs the
//             result's own precision), then v_ldexp_f32 by n: correct subnormals, 0, +inf.
// Error budget: E carries < 2^-36 relative, the f32 stage <= ~1.2 ULP worst case
// (final fma 0.5 + polynomial + g's rounding), total measured <= 2 ULP against the
// 4 ULP bar.  An all-fp64 exp stage holds 1 ULP but costs 11 more fp64 ops/element
// and leaves the kernel VALU-bound (profiles/r01_sweep_pow.txt).
// No MFMA (nothing to contract): 15 fp64-rate VALU ops (4.1 cycles per wave-instruction,
// tools/ubench_valu.hip) + ~20 f32/integer ops per element.  Everything is straight-line:
// W elements are evaluated side by side (pow_n<W>), polynomial constants come
// from constant memory into SGPRs (so a Horner step is one v_fma_f64 with an
// SGPR addend instead of two v_mov + v_fmac), and the special-case lattice is
// skipped wave-uniformly when no lane needs it.  The lattice is C99 F.9.4.4 /
// IEEE 754-2008 9.2.1 as glibc implements it (x^0 = 1 and 1^y = 1 even for
// quiet NaN, not for signalling NaN).
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
#define SM_POW_FMAF(a, b, c) __builtin_fmaf((a), (b), (c))
#define SM_POW_LDEXPF(x, n) __builtin_ldexpf((x), (n))
#else
#include <math.h>
#define SM_POW_FMA(a, b, c) fma((a), (b), (c))
#define SM_POW_RINT(x) rint(x)
#define SM_POW_FMAF(a, b, c) fmaf((a), (b), (c))
#define SM_POW_LDEXPF(x, n) ldexpf((x), (n))
#endif

namespace smpow {

constexpr int kTabFirst = 45, kTabLast = 91, kTabN = kTabLast - kTabFirst + 1;  // c = k/64

// Device: constant memory (not `const`, or the compiler folds the values back
// into 64-bit immediates that cost two v_mov_b32 per use).  Host: plain tables.
#if defined(__HIPCC__)
#define SM_POW_TABLE inline __constant__ double
#else
#define SM_POW_TABLE static const double
#endif

// {invc, logc} pairs: invc = fl(64/k), logc = fl(log2(1/invc)); tools/gen_pow_table.py
SM_POW_TABLE kLogTab[2 * kTabN] = {
    0x1.6c16c16c16c17p+0, -0x1.042bd4b9a7c99p-1,  // k=45
    0x1.642c8590b2164p+0, -0x1.e7df5fe538ab3p-2,  // k=46
    0x1.5c9882b931057p+0, -0x1.c819dc2d45fe4p-2,  // k=47
    0x1.5555555555555p+0, -0x1.a8ff971810a5dp-2,  // k=48
    0x1.4e5e0a72f0539p+0, -0x1.8a8980abfbd30p-2,  // k=49
    0x1.47ae147ae147bp+0, -0x1.6cb0f6865c8ebp-2,  // k=50
    0x1.4141414141414p+0, -0x1.4f6fbb2cec598p-2,  // k=51
    0x1.3b13b13b13b14p+0, -0x1.32bfee370ee6ap-2,  // k=52
    0x1.3521cfb2b78c1p+0, -0x1.169c05363f157p-2,  // k=53
    0x1.2f684bda12f68p+0, -0x1.f5fd8a9063e32p-3,  // k=54
    0x1.29e4129e4129ep+0, -0x1.bfc67a7fff4cap-3,  // k=55
    0x1.2492492492492p+0, -0x1.8a8980abfbd30p-3,  // k=56
    0x1.1f7047dc11f70p+0, -0x1.563dc29ffacafp-3,  // k=57
    0x1.1a7b9611a7b96p+0, -0x1.22dadc2ab3496p-3,  // k=58
    0x1.15b1e5f75270dp+0, -0x1.e0b1ae8f2fd56p-4,  // k=59
    0x1.1111111111111p+0, -0x1.7d60496cfbb4bp-4,  // k=60
    0x1.0c9714fbcda3bp+0, -0x1.1bb32a60054a2p-4,  // k=61
    0x1.0842108421084p+0, -0x1.77394c9d958d0p-5,  // k=62
    0x1.0410410410410p+0, -0x1.743ee861f353fp-6,  // k=63
    0x1.0000000000000p+0, 0x0.0p+0,  // k=64
    0x1.f81f81f81f820p-1, 0x1.6e79685c2d212p-6,  // k=65
    0x1.f07c1f07c1f08p-1, 0x1.6bad3758efd81p-5,  // k=66
    0x1.e9131abf0b767p-1, 0x1.0eb389fa29f9dp-4,  // k=67
    0x1.e1e1e1e1e1e1ep-1, 0x1.663f6fac91318p-4,  // k=68
    0x1.dae6076b981dbp-1, 0x1.bc84240adabb9p-4,  // k=69
    0x1.d41d41d41d41dp-1, 0x1.08c588cda79e5p-3,  // k=70
    0x1.cd85689039b0bp-1, 0x1.32ae9e278ae19p-3,  // k=71
    0x1.c71c71c71c71cp-1, 0x1.5c01a39fbd68bp-3,  // k=72
    0x1.c0e070381c0e0p-1, 0x1.84c2bd02f03b6p-3,  // k=73
    0x1.bacf914c1bad0p-1, 0x1.acf5e2db4ec91p-3,  // k=74
    0x1.b4e81b4e81b4fp-1, 0x1.d49ee4c32596cp-3,  // k=75
    0x1.af286bca1af28p-1, 0x1.fbc16b902680dp-3,  // k=76
    0x1.a98ef606a63bep-1, 0x1.11307dad30b74p-2,  // k=77
    0x1.a41a41a41a41ap-1, 0x1.24407ab0e073ap-2,  // k=78
    0x1.9ec8e951033d9p-1, 0x1.37124cea4cdedp-2,  // k=79
    0x1.999999999999ap-1, 0x1.49a784bcd1b8ap-2,  // k=80
    0x1.948b0fcd6e9e0p-1, 0x1.5c01a39fbd689p-2,  // k=81
    0x1.8f9c18f9c18fap-1, 0x1.6e221cd9d0cddp-2,  // k=82
    0x1.8acb90f6bf3aap-1, 0x1.800a563161c53p-2,  // k=83
    0x1.8618618618618p-1, 0x1.91bba891f170ap-2,  // k=84
    0x1.8181818181818p-1, 0x1.a33760a7f6051p-2,  // k=85
    0x1.7d05f417d05f4p-1, 0x1.b47ebf73882a1p-2,  // k=86
    0x1.78a4c8178a4c8p-1, 0x1.c592fad295b57p-2,  // k=87
    0x1.745d1745d1746p-1, 0x1.d6753e032ea0ep-2,  // k=88
    0x1.702e05c0b8170p-1, 0x1.e726aa1e754d3p-2,  // k=89
    0x1.6c16c16c16c17p-1, 0x1.f7a8568cb06cep-2,  // k=90
    0x1.6816816816817p-1, 0x1.03fda8b97997ep-1,  // k=91
};

//   [0..5]  (-1)^i / ((i+1) ln2)              Q(r) = log2(1+r)/r
SM_POW_TABLE kC[6] = {
    0x1.71547652b82fep+0,  // 1.4426950408889634
    -0x1.71547652b82fep-1,  // -0.7213475204444817
    0x1.ec709dc3a03fdp-2,  // 0.4808983469629878
    -0x1.71547652b82fep-2,  // -0.36067376022224085
    0x1.2776c50ef9bfep-2,  // 0.28853900817779266
    -0x1.ec709dc3a03fdp-3,  // -0.2404491734814939
};
#undef SM_POW_TABLE

SM_POW_FN uint32_t f32_bits(float x) { uint32_t u; memcpy(&u, &x, 4); return u; }
SM_POW_FN float bits_f32(uint32_t u) { float x; memcpy(&x, &u, 4); return x; }
SM_POW_FN uint64_t f64_bits(double x) { uint64_t u; memcpy(&u, &x, 8); return u; }
SM_POW_FN double bits_f64(uint64_t u) { double x; memcpy(&x, &u, 8); return x; }
SM_POW_FN double make_f64(uint32_t hi, uint32_t lo) { return bits_f64(((uint64_t)hi << 32) | lo); }

// double -> int32 with saturation (v_cvt_i32_f64 saturates in hardware; in C++ an
// out-of-range conversion is undefined, so the device form is spelled as the instruction).
SM_POW_FN int sat_i32(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
    int r;
    asm("v_cvt_i32_f64 %0, %1" : "=v"(r) : "v"(x));
    return r;
#else
    return x >= 2147483647.0 ? 2147483647 : (x <= -2147483648.0 ? (-2147483647 - 1) : (int)x);
#endif
}

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
// rounded to f32.  `tab` = kLogTab's layout (LDS copy on the device).
// Straight-line; the loops are over the W independent elements.
template <int W>
SM_POW_FN void pow_core_n(const float (&ax)[W], const float (&y)[W], float (&out)[W], const double *tab) {
    double m[W], r[W], p[W], le[W];
#pragma unroll
    for (int k = 0; k < W; ++k) {
        // exact widening; f32 subnormals become normal doubles
        const uint64_t db = f64_bits((double)ax[k]);
        const uint32_t hi = (uint32_t)(db >> 32), lo = (uint32_t)db;
        const uint32_t mant = hi & 0x000fffffu;
        const bool big = mant > 0x0006a09eu;                     // m > ~sqrt(2): use m/2, e+1
        const int e = (int)(hi >> 20) - 1023 + (big ? 1 : 0);
        m[k] = make_f64(mant | (big ? 0x3fe00000u : 0x3ff00000u), lo);
        // k = round(64 m): m = 1+f -> 64 + round(64 f);  m = (1+f)/2 -> 32 + round(32 f)
        const uint32_t idx = big ? (32u - kTabFirst) + ((mant + 0x4000u) >> 15) : (64u - kTabFirst) + ((mant + 0x2000u) >> 14);
        const double invc = tab[2 * idx], logc = tab[2 * idx + 1];
        r[k] = SM_POW_FMA(m[k], invc, -1.0);                     // exact when invc == 1
        le[k] = logc + (double)e;
    }
#define SM_POW_STEP(acc, x, c) for (int k = 0; k < W; ++k) acc[k] = SM_POW_FMA(acc[k], x[k], c)
#pragma unroll
    for (int k = 0; k < W; ++k) p[k] = kC[5];
#pragma unroll
    SM_POW_STEP(p, r, kC[4]);
#pragma unroll
    SM_POW_STEP(p, r, kC[3]);
#pragma unroll
    SM_POW_STEP(p, r, kC[2]);
#pragma unroll
    SM_POW_STEP(p, r, kC[1]);
#pragma unroll
    SM_POW_STEP(p, r, kC[0]);
    float gf[W], qf[W];
    int ni[W];
#pragma unroll
    for (int k = 0; k < W; ++k) {
        const double lg = SM_POW_FMA(r[k], p[k], le[k]);         // log2(ax)
        const double E = (double)y[k] * lg;
        const double n = SM_POW_RINT(E);
        gf[k] = (float)(E - n) * 0.693147182f;                   // g = f * ln2, |g| <= 0.3466
        ni[k] = sat_i32(n);                                      // |E| can exceed int range (huge y)
    }
    // 2^f = e^g = 1 + g * P(g),  P(g) = 1 + g/2 + g^2/6 + ... + g^6/5040  (truncation < 2^-27)
#define SM_POW_STEPF(acc, x, c) for (int k = 0; k < W; ++k) acc[k] = SM_POW_FMAF(acc[k], x[k], c)
#pragma unroll
    for (int k = 0; k < W; ++k) qf[k] = 1.98412698e-4f;          // 1/7!
#pragma unroll
    SM_POW_STEPF(qf, gf, 1.38888889e-3f);                        // 1/6!
#pragma unroll
    SM_POW_STEPF(qf, gf, 8.33333333e-3f);                        // 1/5!
#pragma unroll
    SM_POW_STEPF(qf, gf, 4.16666667e-2f);                        // 1/4!
#pragma unroll
    SM_POW_STEPF(qf, gf, 1.66666667e-1f);                        // 1/3!
#pragma unroll
    SM_POW_STEPF(qf, gf, 0.5f);
#pragma unroll
    SM_POW_STEPF(qf, gf, 1.0f);
#pragma unroll
    SM_POW_STEPF(qf, gf, 1.0f);                                  // q = 1 + g*P(g) in [0.70, 1.42]
#undef SM_POW_STEPF
#undef SM_POW_STEP
#pragma unroll
    for (int k = 0; k < W; ++k) {
        // beyond +-300 the result is 0 / inf anyway; ldexp rounds subnormal results correctly
        const int n = ni[k] < -300 ? -300 : (ni[k] > 300 ? 300 : ni[k]);
        out[k] = SM_POW_LDEXPF(qf[k], n);
    }
}

// True when any lane of the wavefront (device) / the value itself (host) is set.
SM_POW_FN bool any_lane(bool v) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_ballot_w64(v) != 0;
#else
    return v;
#endif
}

// x^y for W independent (x, y) pairs.
template <int W>
SM_POW_FN void pow_n(const float (&x)[W], const float (&y)[W], float (&out)[W], const double *tab) {
    const uint32_t ONE = 0x3f800000u, INF = 0x7f800000u, QNAN = 0x7fc00000u;
    // Ordinary operands -- x positive, finite, non-zero; y finite, non-zero -- need none of
    // the special-case lattice.  The test is wave-uniform (one ballot), so the usual case
    // (e.g. BASELINE config 4: a in (0.01, 100), y = 2.5) runs the bare exp2/log2 chain and
    // only wavefronts that actually hold a special lane pay for the selects below.
    // x ordinary <=> ix - 1 < INF - 1 (unsigned: 0 wraps to the top, negatives have the sign bit)
    uint32_t worst_x = 0, worst_y = 0;
#pragma unroll
    for (int k = 0; k < W; ++k) {
        const uint32_t tx = f32_bits(x[k]) - 1u, ty = (f32_bits(y[k]) & 0x7fffffffu) - 1u;
        worst_x = tx > worst_x ? tx : worst_x;
        worst_y = ty > worst_y ? ty : worst_y;
    }
    const bool special = worst_x >= INF - 1u || worst_y >= INF - 1u;
    if (!any_lane(special)) {
        pow_core_n<W>(x, y, out, tab);
        return;
    }
    float axc[W], yc_f[W], core[W];
#pragma unroll
    for (int k = 0; k < W; ++k) {
        const uint32_t ax = f32_bits(x[k]) & 0x7fffffffu, ay = f32_bits(y[k]) & 0x7fffffffu;
        // keep the core's inputs finite and positive; special lanes are overwritten below
        axc[k] = bits_f32((ax == 0 || ax >= INF) ? ONE : ax);
        yc_f[k] = ay >= INF ? 1.0f : y[k];
    }
    pow_core_n<W>(axc, yc_f, core, tab);
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

// One element, table read from where it lives (constant memory on the device):
// the per-element kernels and vector tails.
SM_POW_FN float powf(float x, float y) {
    const float xs[1] = {x}, ys[1] = {y};
    float r[1];
    pow_n<1>(xs, ys, r, kLogTab);
    return r[0];
}

}  // namespace smpow
