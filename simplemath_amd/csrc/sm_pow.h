// sm_pow.h -- float pow evaluated in registers, shared by the gfx950 kernels
// and (compiled for the host by tests/cpp/pow_host_check.cpp) by the CPU check of
// the algorithm itself.
//
// Stands in for PowOp<float>::apply = std::pow(float, float)
// (reference include/math/pow.h:8-10); the reference has no vector body for it
// (pow.h:12-13 undefined, :16-32 commented out).  Parity bar: <= 4 ULP of the
// correctly rounded result (BASELINE north_star); this evaluation stays within
// 2 ULP (measured: 1), so it is interchangeable with glibc powf under that bar.
//
// Method: x^y = 2^(y * log2 x), the exponent chain in fp64, everything else in f32 / integers.
//   Range reduction on the float's BITS, one straight-line path (round 1 widened to fp64 first and chose between m and
//   m/2 with five selects per element):  tmp = ix - OFF;  i = (tmp >> 16) & 127;  e = (int)tmp >> 23;
//   z = float(ix - (tmp & 0xff800000)) in [OFF, 2 OFF), OFF = 0x3f328000 ~ 0.6973.
//   log2 x = e + logc[i] + r * Q(r),   r = z * invc[i] - 1 (one fp64 fma, exact to 2^-53),  |r| <= 2^-8
//             {invc, logc} from a 128-entry table (2 KiB, staged in LDS by the kernels).  OFF puts 1.0 in the
//             middle of interval 77, whose entry is exactly {1, 0}: around x = 1 the result r * Q(r) keeps its
//             RELATIVE accuracy (2^-37) -- harmless even when x is near 1 and y is huge.  Q: degree 3.
//   E = y * log2 x = n + f,  n = rint(E), |f| <= 1/2        (still fp64: this is where a float exponent chain
//             would lose the result's low bits)
//   2^f     = 1 + f * P(f),  degree-5 P in f32 (the fraction only needs the result's own precision), on the device
//             as PACKED f32 (v_pk_fma_f32: two elements per instruction), then v_ldexp_f32 by n: correct
//             subnormals, 0, +inf.
// Subnormal x (no implicit bit to strip) and everything else out of the ordinary go through the special path, which
// pre-scales by 2^24 and carries -24 into e.
// Cost per element on gfx950: 12 fp64-rate VALU ops (2 cvt in, fma, add, 4 fma, mul, rint, sub, cvt out), 1 cvt_i32,
// 6 integer ops, 1 ds_read_b128, 3 packed-f32 + 1 ldexp (+ a clamp) -- 27, 30 with addressing and the special-case test, against round 1's 45
// (profiles/r02_pmc_sq_cycles.txt: 3.19e7 VALU wave-instructions per 2^26-element launch, r01: 4.76e7).
// No MFMA: there is no contraction.  W elements are evaluated side by side (pow_n<W>), polynomial constants come
// from constant memory into SGPRs, and the special-case lattice is skipped wave-uniformly when no lane needs it.  The
// lattice is C99 F.9.4.4 / IEEE 754-2008 9.2.1 as glibc implements it (x^0 = 1 and 1^y = 1 even for quiet NaN, not
// for signalling NaN).
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

constexpr uint32_t kOff = 0x3f328000u;  // z = x / 2^e in [0.6973, 1.3945); see tools/gen_pow_table.py
constexpr int kTabBits = 7, kTabN = 1 << kTabBits;

// Device: constant memory (not `const`, or the compiler folds the values back
// into 64-bit immediates that cost two v_mov_b32 per use).  Host: plain tables.
#if defined(__HIPCC__)
#define SM_POW_TABLE inline __constant__ double
#define SM_POW_TABLEF inline __constant__ float
#else
#define SM_POW_TABLE static const double
#define SM_POW_TABLEF static const float
#endif

// {invc, logc} per interval: invc = fl(1 / midpoint), logc = fl(log2(1 / invc)); tools/gen_pow_table.py
SM_POW_TABLE kLogTab[2 * kTabN] = {
    0x1.6e1f76b4337c7p+0, -0x1.08494c66b8ef0p-1,  // 0
    0x1.6c16c16c16c17p+0, -0x1.042bd4b9a7c99p-1,  // 1
    0x1.6a13cd1537290p+0, -0x1.0014332be0032p-1,  // 2
    0x1.6816816816817p+0, -0x1.f804ae8d0cd04p-2,  // 3
    0x1.661ec6a5122f9p+0, -0x1.efec61b011f85p-2,  // 4
    0x1.642c8590b2164p+0, -0x1.e7df5fe538ab3p-2,  // 5
    0x1.623fa77016240p+0, -0x1.dfdd89d586e2cp-2,  // 6
    0x1.6058160581606p+0, -0x1.d7e6c0abc357bp-2,  // 7
    0x1.5e75bb8d015e7p+0, -0x1.cffae611ad12ap-2,  // 8
    0x1.5c9882b931057p+0, -0x1.c819dc2d45fe4p-2,  // 9
    0x1.5ac056b015ac0p+0, -0x1.c043859e2fdb2p-2,  // 10
    0x1.58ed2308158edp+0, -0x1.b877c57b1b06fp-2,  // 11
    0x1.571ed3c506b3ap+0, -0x1.b0b67f4f46812p-2,  // 12
    0x1.5555555555555p+0, -0x1.a8ff971810a5dp-2,  // 13
    0x1.5390948f40febp+0, -0x1.a152f142981b5p-2,  // 14
    0x1.51d07eae2f815p+0, -0x1.99b072a96c6b2p-2,  // 15
    0x1.5015015015015p+0, -0x1.921800924dd3bp-2,  // 16
    0x1.4e5e0a72f0539p+0, -0x1.8a8980abfbd30p-2,  // 17
    0x1.4cab88725af6ep+0, -0x1.8304d90c11fd1p-2,  // 18
    0x1.4afd6a052bf5bp+0, -0x1.7b89f02cf2aafp-2,  // 19
    0x1.49539e3b2d067p+0, -0x1.7418acebbf18fp-2,  // 20
    0x1.47ae147ae147bp+0, -0x1.6cb0f6865c8ebp-2,  // 21
    0x1.460cbc7f5cf9ap+0, -0x1.6552b49986277p-2,  // 22
    0x1.446f86562d9fbp+0, -0x1.5dfdcf1eeae0fp-2,  // 23
    0x1.42d6625d51f87p+0, -0x1.56b22e6b578e5p-2,  // 24
    0x1.4141414141414p+0, -0x1.4f6fbb2cec598p-2,  // 25
    0x1.3fb013fb013fbp+0, -0x1.48365e695d797p-2,  // 26
    0x1.3e22cbce4a902p+0, -0x1.4106017c3eca0p-2,  // 27
    0x1.3c995a47babe7p+0, -0x1.39de8e1559f6ep-2,  // 28
    0x1.3b13b13b13b14p+0, -0x1.32bfee370ee6ap-2,  // 29
    0x1.3991c2c187f63p+0, -0x1.2baa0c34be1ebp-2,  // 30
    0x1.3813813813814p+0, -0x1.249cd2b13cd6fp-2,  // 31
    0x1.3698df3de0748p+0, -0x1.1d982c9d5270ap-2,  // 32
    0x1.3521cfb2b78c1p+0, -0x1.169c05363f157p-2,  // 33
    0x1.33ae45b57bcb2p+0, -0x1.0fa848044b352p-2,  // 34
    0x1.323e34a2b10bfp+0, -0x1.08bce0d95fa36p-2,  // 35
    0x1.30d190130d190p+0, -0x1.01d9bbcfa61d4p-2,  // 36
    0x1.2f684bda12f68p+0, -0x1.f5fd8a9063e32p-3,  // 37
    0x1.2e025c04b8097p+0, -0x1.e857d3d361368p-3,  // 38
    0x1.2c9fb4d812ca0p+0, -0x1.dac22d3e441d6p-3,  // 39
    0x1.2b404ad012b40p+0, -0x1.cd3c712d31106p-3,  // 40
    0x1.29e4129e4129ep+0, -0x1.bfc67a7fff4cap-3,  // 41
    0x1.288b01288b013p+0, -0x1.b2602497d534ap-3,  // 42
    0x1.27350b8812735p+0, -0x1.a5094b54d2828p-3,  // 43
    0x1.25e22708092f1p+0, -0x1.97c1cb13c7ec0p-3,  // 44
    0x1.2492492492492p+0, -0x1.8a8980abfbd30p-3,  // 45
    0x1.23456789abcdfp+0, -0x1.7d60496cfbb4cp-3,  // 46
    0x1.21fb78121fb78p+0, -0x1.7046031c79f84p-3,  // 47
    0x1.20b470c67c0d9p+0, -0x1.633a8bf437ce6p-3,  // 48
    0x1.1f7047dc11f70p+0, -0x1.563dc29ffacafp-3,  // 49
    0x1.1e2ef3b3fb874p+0, -0x1.494f863b8df32p-3,  // 50
    0x1.1cf06ada2811dp+0, -0x1.3c6fb650cde51p-3,  // 51
    0x1.1bb4a4046ed29p+0, -0x1.2f9e32d5bfdd1p-3,  // 52
    0x1.1a7b9611a7b96p+0, -0x1.22dadc2ab3496p-3,  // 53
    0x1.19453808ca29cp+0, -0x1.162593186da70p-3,  // 54
    0x1.1811811811812p+0, -0x1.097e38ce6064ep-3,  // 55
    0x1.16e0689427379p+0, -0x1.f9c95dc1d1167p-4,  // 56
    0x1.15b1e5f75270dp+0, -0x1.e0b1ae8f2fd56p-4,  // 57
    0x1.1485f0e0acd3bp+0, -0x1.c7b528b70f1bcp-4,  // 58
    0x1.135c81135c811p+0, -0x1.aed391ab6674ap-4,  // 59
    0x1.12358e75d3033p+0, -0x1.960caf9abb7c1p-4,  // 60
    0x1.1111111111111p+0, -0x1.7d60496cfbb4bp-4,  // 61
    0x1.0fef010fef011p+0, -0x1.64ce26c067157p-4,  // 62
    0x1.0ecf56be69c90p+0, -0x1.4c560fe68af8bp-4,  // 63
    0x1.0db20a88f4696p+0, -0x1.33f7cde14cf63p-4,  // 64
    0x1.0c9714fbcda3bp+0, -0x1.1bb32a60054a2p-4,  // 65
    0x1.0b7e6ec259dc8p+0, -0x1.0387efbca86a7p-4,  // 66
    0x1.0a6810a6810a7p+0, -0x1.d6ebd1f1fec14p-5,  // 67
    0x1.0953f39010954p+0, -0x1.a6f9c377dd31dp-5,  // 68
    0x1.0842108421084p+0, -0x1.77394c9d958d0p-5,  // 69
    0x1.073260a47f7c6p+0, -0x1.47aa07357703cp-5,  // 70
    0x1.0624dd2f1a9fcp+0, -0x1.184b8e4c56afcp-5,  // 71
    0x1.05197f7d73404p+0, -0x1.d23afc49139f1p-6,  // 72
    0x1.0410410410410p+0, -0x1.743ee861f353fp-6,  // 73
    0x1.03091b51f5e1ap+0, -0x1.16a21e20a0a29p-6,  // 74
    0x1.0204081020408p+0, -0x1.72c7ba20f731cp-7,  // 75
    0x1.0101010101010p+0, -0x1.720d9c06a8348p-8,  // 76
    0x1.0000000000000p+0, 0x0.0p+0,  // 77
    0x1.fc07f01fc07f0p-1, 0x1.6fe50b6ef085dp-7,  // 78
    0x1.f81f81f81f820p-1, 0x1.6e79685c2d212p-6,  // 79
    0x1.f44659e4a4271p-1, 0x1.11cd1d513341bp-5,  // 80
    0x1.f07c1f07c1f08p-1, 0x1.6bad3758efd81p-5,  // 81
    0x1.ecc07b301ecc0p-1, 0x1.c4dfab90aab6ap-5,  // 82
    0x1.e9131abf0b767p-1, 0x1.0eb389fa29f9dp-4,  // 83
    0x1.e573ac901e574p-1, 0x1.3aa2fdd27f1bfp-4,  // 84
    0x1.e1e1e1e1e1e1ep-1, 0x1.663f6fac91318p-4,  // 85
    0x1.de5d6e3f8868ap-1, 0x1.918a16e46335ep-4,  // 86
    0x1.dae6076b981dbp-1, 0x1.bc84240adabb9p-4,  // 87
    0x1.d77b654b82c34p-1, 0x1.e72ec117fa5adp-4,  // 88
    0x1.d41d41d41d41dp-1, 0x1.08c588cda79e5p-3,  // 89
    0x1.d0cb58f6ec074p-1, 0x1.1dcd197552b7dp-3,  // 90
    0x1.cd85689039b0bp-1, 0x1.32ae9e278ae19p-3,  // 91
    0x1.ca4b3055ee191p-1, 0x1.476a9f983f74dp-3,  // 92
    0x1.c71c71c71c71cp-1, 0x1.5c01a39fbd68bp-3,  // 93
    0x1.c3f8f01c3f8f0p-1, 0x1.70742d4ef0280p-3,  // 94
    0x1.c0e070381c0e0p-1, 0x1.84c2bd02f03b6p-3,  // 95
    0x1.bdd2b899406f7p-1, 0x1.98edd077e70e1p-3,  // 96
    0x1.bacf914c1bad0p-1, 0x1.acf5e2db4ec91p-3,  // 97
    0x1.b7d6c3dda338bp-1, 0x1.c0db6cdd94defp-3,  // 98
    0x1.b4e81b4e81b4fp-1, 0x1.d49ee4c32596cp-3,  // 99
    0x1.b2036406c80d9p-1, 0x1.e840be74e6a4dp-3,  // 100
    0x1.af286bca1af28p-1, 0x1.fbc16b902680dp-3,  // 101
    0x1.ac5701ac5701bp-1, 0x1.0790adbb03009p-2,  // 102
    0x1.a98ef606a63bep-1, 0x1.11307dad30b74p-2,  // 103
    0x1.a6d01a6d01a6dp-1, 0x1.1ac05b291f070p-2,  // 104
    0x1.a41a41a41a41ap-1, 0x1.24407ab0e073ap-2,  // 105
    0x1.a16d3f97a4b02p-1, 0x1.2db10fc4d9aaep-2,  // 106
    0x1.9ec8e951033d9p-1, 0x1.37124cea4cdedp-2,  // 107
    0x1.9c2d14ee4a102p-1, 0x1.406463b1b0448p-2,  // 108
    0x1.999999999999ap-1, 0x1.49a784bcd1b8ap-2,  // 109
    0x1.970e4f80cb872p-1, 0x1.52dbdfc4c96b5p-2,  // 110
    0x1.948b0fcd6e9e0p-1, 0x1.5c01a39fbd689p-2,  // 111
    0x1.920fb49d0e229p-1, 0x1.6518fe4677ba6p-2,  // 112
    0x1.8f9c18f9c18fap-1, 0x1.6e221cd9d0cddp-2,  // 113
    0x1.8d3018d3018d3p-1, 0x1.771d2ba7efb3cp-2,  // 114
    0x1.8acb90f6bf3aap-1, 0x1.800a563161c53p-2,  // 115
    0x1.886e5f0abb04ap-1, 0x1.88e9c72e0b224p-2,  // 116
    0x1.8618618618618p-1, 0x1.91bba891f170ap-2,  // 117
    0x1.83c977ab2beddp-1, 0x1.9a802391e2330p-2,  // 118
    0x1.8181818181818p-1, 0x1.a33760a7f6051p-2,  // 119
    0x1.7f405fd017f40p-1, 0x1.abe18797f1f4ap-2,  // 120
    0x1.7d05f417d05f4p-1, 0x1.b47ebf73882a1p-2,  // 121
    0x1.7ad2208e0ecc3p-1, 0x1.bd0f2e9e79032p-2,  // 122
    0x1.78a4c8178a4c8p-1, 0x1.c592fad295b57p-2,  // 123
    0x1.767dce434a9b1p-1, 0x1.ce0a4923a587dp-2,  // 124
    0x1.745d1745d1746p-1, 0x1.d6753e032ea0ep-2,  // 125
    0x1.724287f46debcp-1, 0x1.ded3fd442364cp-2,  // 126
    0x1.702e05c0b8170p-1, 0x1.e726aa1e754d3p-2,  // 127
};

// Q(r) = log2(1 + r) / r on |r| <= 2^-8, degree 3, max relative error 2^-37.3
SM_POW_TABLE kC[4] = {
    0x1.71547652aef42p+0,  // 1.4426950408805657
    -0x1.7154765291b6fp-1,  // -0.7213475204269865
    0x1.ec71c53b2b2a5p-2,  // 0.48090274976383746
    -0x1.7155aa1a0d2a5p-2,  // -0.3606783464813737
};
// P(f) = (2^f - 1) / f on |f| <= 1/2, degree 5, for f32 evaluation: 1 + f P(f) is within 2^-26.8 of 2^f
SM_POW_TABLEF kP[6] = {
    0x1.62e430p-1f,  // 0.6931471824645996
    0x1.ebfbe0p-3f,  // 0.24022650718688965
    0x1.c6af6cp-5f,  // 0.055503569543361664
    0x1.3b2a1cp-7f,  // 0.009618056938052177
    0x1.5f0896p-10f,  // 0.0013390866806730628
    0x1.444004p-13f,  // 0.00015461447765119374
};
#undef SM_POW_TABLE
#undef SM_POW_TABLEF

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

// 2^(y * log2(ax)) for W finite positive NORMAL ax (eadj[k] added to the exponent: the special path passes -24 with
// a pre-scaled subnormal) and finite y, rounded to f32.  `tab` = kLogTab's layout (LDS copy on the device).
// Straight-line; the loops are over the W independent elements.
template <int W, bool ADJ>
SM_POW_FN void pow_core_n(const float (&ax)[W], const float (&y)[W], float (&out)[W], const double *tab, const int (&eadj)[W]) {
    double r[W], p[W], le[W];
#pragma unroll
    for (int k = 0; k < W; ++k) {
        const uint32_t ix = f32_bits(ax[k]);
        const uint32_t tmp = ix - kOff;
        const uint32_t idx = (tmp >> (23 - kTabBits)) & (uint32_t)(kTabN - 1);
        const int e = ((int32_t)tmp >> 23) + (ADJ ? eadj[k] : 0);
        const double z = (double)bits_f32(ix - (tmp & 0xff800000u));
        const double invc = tab[2 * idx], logc = tab[2 * idx + 1];
        r[k] = SM_POW_FMA(z, invc, -1.0);                        // exact when invc == 1
        le[k] = logc + (double)e;
    }
#define SM_POW_STEP(acc, x, c) for (int k = 0; k < W; ++k) acc[k] = SM_POW_FMA(acc[k], x[k], c)
#pragma unroll
    for (int k = 0; k < W; ++k) p[k] = kC[3];
#pragma unroll
    SM_POW_STEP(p, r, kC[2]);
#pragma unroll
    SM_POW_STEP(p, r, kC[1]);
#pragma unroll
    SM_POW_STEP(p, r, kC[0]);
#undef SM_POW_STEP
    float ff[W];
    int ni[W];
#pragma unroll
    for (int k = 0; k < W; ++k) {
        const double lg = SM_POW_FMA(r[k], p[k], le[k]);         // log2(ax)
        const double E = (double)y[k] * lg;
        const double n = SM_POW_RINT(E);
        ff[k] = (float)(E - n);                                  // |f| <= 1/2
        ni[k] = sat_i32(n);                                      // |E| can exceed int range (huge y)
    }
    // 2^f = 1 + f * P(f)
    float qf[W];
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (W % 2 == 0) {  // packed f32: two elements per v_pk_fma_f32
        typedef float f2 __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int k = 0; k < W; k += 2) {
            const f2 f = {ff[k], ff[k + 1]};
            f2 q = {kP[5], kP[5]};
            q = __builtin_elementwise_fma(q, f, (f2){kP[4], kP[4]});
            q = __builtin_elementwise_fma(q, f, (f2){kP[3], kP[3]});
            q = __builtin_elementwise_fma(q, f, (f2){kP[2], kP[2]});
            q = __builtin_elementwise_fma(q, f, (f2){kP[1], kP[1]});
            q = __builtin_elementwise_fma(q, f, (f2){kP[0], kP[0]});
            q = __builtin_elementwise_fma(q, f, (f2){1.0f, 1.0f});
            qf[k] = q[0];
            qf[k + 1] = q[1];
        }
    } else
#endif
    {
#pragma unroll
        for (int k = 0; k < W; ++k) {
            float q = kP[5];
            q = SM_POW_FMAF(q, ff[k], kP[4]);
            q = SM_POW_FMAF(q, ff[k], kP[3]);
            q = SM_POW_FMAF(q, ff[k], kP[2]);
            q = SM_POW_FMAF(q, ff[k], kP[1]);
            q = SM_POW_FMAF(q, ff[k], kP[0]);
            qf[k] = SM_POW_FMAF(q, ff[k], 1.0f);                 // in [0.70, 1.42]
        }
    }
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

template <int W>
SM_POW_FN void pow_special_n(const float (&x)[W], const float (&y)[W], float (&out)[W], const double *tab);

// x^y for W independent (x, y) pairs.
template <int W>
SM_POW_FN void pow_n(const float (&x)[W], const float (&y)[W], float (&out)[W], const double *tab) {
    const uint32_t INF = 0x7f800000u;
    // Ordinary operands -- x positive, finite, non-zero; y finite, non-zero -- need none of
    // the special-case lattice.  The test is wave-uniform (one ballot), so the usual case
    // (e.g. BASELINE config 4: a in (0.01, 100), y = 2.5) runs the bare exp2/log2 chain and
    // only wavefronts that actually hold a special lane pay for the selects below.
    // x ordinary <=> positive, finite and NORMAL <=> ix - MINNORM < INF - MINNORM (unsigned: zero and subnormals wrap
    // to the top, negatives have the sign bit)
    const uint32_t MINNORM = 0x00800000u;
    uint32_t worst_x = 0, worst_y = 0;
#pragma unroll
    for (int k = 0; k < W; ++k) {
        const uint32_t tx = f32_bits(x[k]) - MINNORM, ty = (f32_bits(y[k]) & 0x7fffffffu) - 1u;
        worst_x = tx > worst_x ? tx : worst_x;
        worst_y = ty > worst_y ? ty : worst_y;
    }
    const bool special = worst_x >= INF - MINNORM || worst_y >= INF - 1u;
    int eadj[W];
    if (!any_lane(special)) {
        pow_core_n<W, false>(x, y, out, tab, eadj);
        return;
    }
    // From here on the operands are opaque to the optimiser.  Without this, a kernel that evaluates several vectors
    // against the SAME exponents (the row kernel: two rows, one row-constant operand) sees the exponent-only part of
    // the lattice below as common to both evaluations and hoists it ABOVE the branch -- ~70 VALU instructions per wave
    // executed on the ordinary path for nothing (row pow: 393 instructions per wave instead of 315,
    // profiles/r02_pmc_sq_pow_shapes.txt).
    float xs[W], ys[W];
#pragma unroll
    for (int k = 0; k < W; ++k) {
        xs[k] = x[k];
        ys[k] = y[k];
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("" : "+v"(xs[k]), "+v"(ys[k]));
#endif
    }
    pow_special_n<W>(xs, ys, out, tab);
}

// The special path of pow_n: at least one lane of the wavefront holds a zero, subnormal, negative, infinite or NaN base,
// or a zero, infinite or NaN exponent.
template <int W>
SM_POW_FN void pow_special_n(const float (&x)[W], const float (&y)[W], float (&out)[W], const double *tab) {
    const uint32_t ONE = 0x3f800000u, INF = 0x7f800000u, QNAN = 0x7fc00000u, MINNORM = 0x00800000u;
    int eadj[W];
    float axc[W], yc_f[W], core[W];
#pragma unroll
    for (int k = 0; k < W; ++k) {
        const uint32_t ax = f32_bits(x[k]) & 0x7fffffffu, ay = f32_bits(y[k]) & 0x7fffffffu;
        // keep the core's inputs finite, positive and normal; special lanes are overwritten below
        const bool sub = ax < MINNORM;                            // subnormal: scale by 2^24 (exact), carry -24 into e
        const float a = (ax == 0 || ax >= INF) ? 1.0f : bits_f32(ax);
        axc[k] = sub ? a * 16777216.0f : a;
        eadj[k] = (sub && ax != 0) ? -24 : 0;
        yc_f[k] = ay >= INF ? 1.0f : y[k];
    }
    pow_core_n<W, true>(axc, yc_f, core, tab, eadj);
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
