// sm_pow64.h -- double-precision pow evaluated in registers (PowOp<double>), shared by the gfx950
// kernels and its host check (tests/cpp/pow64_host_check.cpp).
//
// Stands in for PowOp<double>::apply = std::pow(double, double) (reference include/math/pow.h:8-10; the
// reference has no vector body: pow.h:34-52 is commented out).  Bar: the float bar of BASELINE north_star
// (4 ULP) applied to double; measured <= 1 ULP against glibc pow on the host sweep.
//
// Method (double-double where it matters):
//   x = 2^k * z, z in [OFF, 2 OFF);  interval i of 128:  r = z * invc[i] - 1 as r + rlo (invc has 9
//   significant bits, |r| < 2^-7.59, so the product needs up to 54 bits);  ln x = k ln2 + logc[i] + ln(1 + r) accumulated as hi + lo:
//     k*Ln2hi + logc_hi is exact (trailing zeros), r and -r^2/2 enter through error-free sums / an fma
//     residual, the cubic-and-higher terms (< 2^-24 relative) in plain fp64; the interval holding 1 has
//     {invc, logc} = {1, 0}, so ln x keeps its relative accuracy near x = 1.
//   E = y * ln x as ehi + elo (fma residual);  E = (k' + j/128) ln2 + r',  |r'| < ln2/256;
//   e^E = 2^k' * T[j] * (1 + tail[j] + r' + r'^2/2 + ... + r'^5/120);  v_ldexp_f64 applies 2^k' with
//   correct subnormal / overflow behaviour.
// Tables (3 KiB + 2 KiB) are staged in LDS by the kernels (OpCtx<PowOp<double>>); tools/gen_pow64_tables.py
// generates them.  ~54 fp64 + ~11 integer / conversion vector instructions per element on the ordinary path
// (pow_core_t<false>; operands that are not ordinary -- anything but a positive normal base and an exponent of normal
// magnitude -- take pow_general).  Special cases: C99 F.9.4.4 as for the float form.
#pragma once

#include <stdint.h>
#include <string.h>
#include <type_traits>

#include "sm_pow.h"  // SM_POW_FN, SM_POW_FMA, SM_POW_RINT, sat_i32, any_lane, f64_bits / bits_f64

#if defined(__HIP_DEVICE_COMPILE__)
#define SM_POW_LDEXP(x, n) __builtin_ldexp((x), (n))
#else
#define SM_POW_LDEXP(x, n) ldexp((x), (n))
#endif
#if defined(__HIPCC__)
#define SM_POW64_TABLE inline __constant__ double
#else
#define SM_POW64_TABLE static const double
#endif

namespace smpow64 {

using smpow::bits_f64;
using smpow::f64_bits;

constexpr int kN = 128;
constexpr uint64_t kOff = 0x3fe6955500000000ULL;
constexpr int kLogTabDoubles = 3 * kN, kExpTabDoubles = 2 * kN;

// {invc, logc_hi, logc_lo} per interval
SM_POW64_TABLE kLogTab[kLogTabDoubles] = {
    0x1.6a00000000000p+0, -0x1.62c82f2a00000p-2, -0x1.9c7952f6f5f23p-34,  // 0
    0x1.6800000000000p+0, -0x1.5d1bdbf400000p-2, -0x1.809ca508d8e0fp-34,  // 1
    0x1.6600000000000p+0, -0x1.5767717400000p-2, -0x1.569b1526adb28p-36,  // 2
    0x1.6400000000000p+0, -0x1.51aad87200000p-2, -0x1.bf05a13927ac2p-35,  // 3
    0x1.6200000000000p+0, -0x1.4be5f95600000p-2, -0x1.778a0db4c994ap-34,  // 4
    0x1.6000000000000p+0, -0x1.4618bc2000000p-2, -0x1.c5ec27d0b7b38p-34,  // 5
    0x1.5e00000000000p+0, -0x1.4043086800000p-2, -0x1.a9f8ef43049f8p-36,  // 6
    0x1.5c00000000000p+0, -0x1.3a64c55600000p-2, -0x1.28bd38e5e6b9bp-35,  // 7
    0x1.5a00000000000p+0, -0x1.347dd9a800000p-2, -0x1.87d54d6456750p-34,  // 8
    0x1.5900000000000p+0, -0x1.31871c9400000p-2, -0x1.44184fab94cedp-34,  // 9
    0x1.5700000000000p+0, -0x1.2b9303aa00000p-2, -0x1.89d249da5280ap-34,  // 10
    0x1.5500000000000p+0, -0x1.2596010c00000p-2, -0x1.f7639ef0893a9p-34,  // 11
    0x1.5300000000000p+0, -0x1.1f8ff9e400000p-2, -0x1.145e51b010330p-35,  // 12
    0x1.5200000000000p+0, -0x1.1c898c1600000p-2, -0x1.333f5f78d1cebp-35,  // 13
    0x1.5000000000000p+0, -0x1.1675caba00000p-2, -0x1.74c1c07398fabp-35,  // 14
    0x1.4e00000000000p+0, -0x1.1058bf9a00000p-2, -0x1.c95aa313f4157p-35,  // 15
    0x1.4c00000000000p+0, -0x1.0a324e2600000p-2, -0x1.390e35f73f7a0p-34,  // 16
    0x1.4b00000000000p+0, -0x1.071b85fc00000p-2, -0x1.ab21a3a2e0ff3p-35,  // 17
    0x1.4900000000000p+0, -0x1.00e6c45a00000p-2, -0x1.aa0398d1aa5c0p-35,  // 18
    0x1.4700000000000p+0, -0x1.f550a56400000p-3, -0x1.6f66e0e2fb6ffp-36,  // 19
    0x1.4600000000000p+0, -0x1.ef0adcbc00000p-3, -0x1.c59365218de54p-35,  // 20
    0x1.4400000000000p+0, -0x1.e27076e200000p-3, -0x1.5e5cbd3d51000p-36,  // 21
    0x1.4300000000000p+0, -0x1.dc1bca0a00000p-3, -0x1.7d8fac1a628cdp-36,  // 22
    0x1.4100000000000p+0, -0x1.cf6354e000000p-3, -0x1.38bb891cd03ebp-36,  // 23
    0x1.3f00000000000p+0, -0x1.c296855800000p-3, -0x1.8318146108e3bp-36,  // 24
    0x1.3e00000000000p+0, -0x1.bc28674200000p-3, -0x1.b19ac53f39d12p-36,  // 25
    0x1.3c00000000000p+0, -0x1.af3c94e800000p-3, -0x1.7fe5b19cc0327p-40,  // 26
    0x1.3b00000000000p+0, -0x1.a8becfc800000p-3, -0x1.05e3185cf21bap-36,  // 27
    0x1.3900000000000p+0, -0x1.9bb362e600000p-3, -0x1.dfb8355d78c7cp-35,  // 28
    0x1.3800000000000p+0, -0x1.9525a9ce00000p-3, -0x1.456b476413075p-35,  // 29
    0x1.3600000000000p+0, -0x1.87fa065200000p-3, -0x1.9221204012030p-40,  // 30
    0x1.3500000000000p+0, -0x1.815c0a1400000p-3, -0x1.abf56b41b7f8cp-38,  // 31
    0x1.3300000000000p+0, -0x1.740f8f5400000p-3, -0x1.bd264d9bf9d58p-42,  // 32
    0x1.3200000000000p+0, -0x1.6d60fe7000000p-3, -0x1.9d21c8d54765cp-35,  // 33
    0x1.3100000000000p+0, -0x1.66acd42600000p-3, -0x1.2ad50dedfe364p-35,  // 34
    0x1.2f00000000000p+0, -0x1.59338d9800000p-3, -0x1.82085d345baabp-35,  // 35
    0x1.2e00000000000p+0, -0x1.526e5e3a00000p-3, -0x1.b437a2e401d6ep-39,  // 36
    0x1.2c00000000000p+0, -0x1.44d2b6cc00000p-3, -0x1.6fa3ccfa7b2a2p-36,  // 37
    0x1.2b00000000000p+0, -0x1.3dfc2b0e00000p-3, -0x1.98c5395315c61p-36,  // 38
    0x1.2a00000000000p+0, -0x1.371fc20000000p-3, -0x1.e8f743bcd96c5p-35,  // 39
    0x1.2800000000000p+0, -0x1.29552f8000000p-3, -0x1.ff5234c05dc71p-35,  // 40
    0x1.2700000000000p+0, -0x1.2266f19000000p-3, -0x1.4b596faa3df8cp-36,  // 41
    0x1.2600000000000p+0, -0x1.1b72ad5200000p-3, -0x1.ecf40520c08d2p-36,  // 42
    0x1.2400000000000p+0, -0x1.0d77e7cc00000p-3, -0x1.08e596697717ap-35,  // 43
    0x1.2300000000000p+0, -0x1.0671512c00000p-3, -0x1.4b2dc543191fbp-36,  // 44
    0x1.2200000000000p+0, -0x1.fec9131c00000p-4, -0x1.beabaaa2e519ap-36,  // 45
    0x1.2000000000000p+0, -0x1.e27076e200000p-4, -0x1.5e5cbd3d51000p-37,  // 46
    0x1.1f00000000000p+0, -0x1.d4313d6600000p-4, -0x1.966babc86eca9p-37,  // 47
    0x1.1e00000000000p+0, -0x1.c5e548f400000p-4, -0x1.bc74315d617f0p-36,  // 48
    0x1.1d00000000000p+0, -0x1.b78c82ba00000p-4, -0x1.0eda10843c678p-36,  // 49
    0x1.1c00000000000p+0, -0x1.a926d3a400000p-4, -0x1.5aac6ca17a455p-37,  // 50
    0x1.1a00000000000p+0, -0x1.8c345d6200000p-4, -0x1.19b20f5acb42ap-36,  // 51
    0x1.1900000000000p+0, -0x1.7da766d600000p-4, -0x1.b12cc844480c9p-36,  // 52
    0x1.1800000000000p+0, -0x1.6f0d28ae00000p-4, -0x1.5ad2e6f9266e8p-38,  // 53
    0x1.1700000000000p+0, -0x1.60658a9200000p-4, -0x1.750c3b1dee9c5p-36,  // 54
    0x1.1500000000000p+0, -0x1.42edcbea00000p-4, -0x1.91bc0eeea7c9bp-38,  // 55
    0x1.1400000000000p+0, -0x1.341d796000000p-4, -0x1.bd1d092998376p-36,  // 56
    0x1.1300000000000p+0, -0x1.253f62f000000p-4, -0x1.4282df1f6d34ep-37,  // 57
    0x1.1200000000000p+0, -0x1.16536eea00000p-4, -0x1.bd7074312e0bap-39,  // 58
    0x1.1100000000000p+0, -0x1.0759835800000p-4, -0x1.8e471301b4a66p-36,  // 59
    0x1.1000000000000p+0, -0x1.f0a30c0000000p-5, -0x1.162a6617cc971p-37,  // 60
    0x1.0f00000000000p+0, -0x1.d276b8ac00000p-5, -0x1.b0b5211e3c532p-37,  // 61
    0x1.0e00000000000p+0, -0x1.b42dd71000000p-5, -0x1.971bec28d14c8p-37,  // 62
    0x1.0c00000000000p+0, -0x1.77458f6200000p-5, -0x1.2dcfc4634f2a2p-37,  // 63
    0x1.0b00000000000p+0, -0x1.58a5bafc00000p-5, -0x1.1c9a918d51ea6p-38,  // 64
    0x1.0a00000000000p+0, -0x1.39e87b9e00000p-5, -0x1.ebd5fa9015b20p-37,  // 65
    0x1.0900000000000p+0, -0x1.1b0d989200000p-5, -0x1.ecbfe16517764p-40,  // 66
    0x1.0800000000000p+0, -0x1.f829b0e600000p-6, -0x1.833004cf8fc14p-38,  // 67
    0x1.0700000000000p+0, -0x1.b9fc027a00000p-6, -0x1.f232ff7a8cb6fp-39,  // 68
    0x1.0600000000000p+0, -0x1.7b91b07c00000p-6, -0x1.5b11aa927f54cp-38,  // 69
    0x1.0500000000000p+0, -0x1.3cea443400000p-6, -0x1.a95d3bcd295bfp-40,  // 70
    0x1.0400000000000p+0, -0x1.fc0a8b0e00000p-7, -0x1.c03e3cf9eda75p-39,  // 71
    0x1.0300000000000p+0, -0x1.7dc475f800000p-7, -0x1.0a76dd2512f06p-43,  // 72
    0x1.0200000000000p+0, -0x1.fe02a6b000000p-8, -0x1.06788fc376904p-40,  // 73
    0x1.0100000000000p+0, -0x1.ff00aa2a00000p-9, -0x1.10bc04a086b57p-41,  // 74
    0x1.0000000000000p+0, 0x0.0p+0, 0x0.0p+0,  // 75
    0x1.fb00000000000p-1, 0x1.41929f9600000p-7, 0x1.065df1d57404ep-40,  // 76
    0x1.f700000000000p-1, 0x1.228fb1fe00000p-6, 0x1.45c4f8ca12648p-39,  // 77
    0x1.f400000000000p-1, 0x1.8492528c00000p-6, 0x1.1957d173697cfp-39,  // 78
    0x1.f000000000000p-1, 0x1.0415d89e00000p-5, 0x1.d1111c05cf1d7p-39,  // 79
    0x1.ec00000000000p-1, 0x1.466aed4200000p-5, 0x1.bc7d319148406p-38,  // 80
    0x1.e800000000000p-1, 0x1.894aa14800000p-5, 0x1.fb3433517d2edp-37,  // 81
    0x1.e500000000000p-1, 0x1.bbcebfc600000p-5, 0x1.1e8407973ce84p-38,  // 82
    0x1.e100000000000p-1, 0x1.ffa6911a00000p-5, 0x1.7260119307035p-38,  // 83
    0x1.de00000000000p-1, 0x1.1973bd1400000p-4, 0x1.9559b4553e4c3p-38,  // 84
    0x1.da00000000000p-1, 0x1.3bdf5a7c00000p-4, 0x1.1ee642f52eda7p-36,  // 85
    0x1.d700000000000p-1, 0x1.55e1005000000p-4, 0x1.c07075d0314f2p-37,  // 86
    0x1.d400000000000p-1, 0x1.700d30ae00000p-4, 0x1.581c1e8da99dfp-37,  // 87
    0x1.d000000000000p-1, 0x1.9335e5d400000p-4, 0x1.94988ae1d5ea4p-36,  // 88
    0x1.cd00000000000p-1, 0x1.adc77ee400000p-4, 0x1.aea8c4df63ce7p-36,  // 89
    0x1.ca00000000000p-1, 0x1.c885801a00000p-4, 0x1.c4b2368e32d56p-36,  // 90
    0x1.c700000000000p-1, 0x1.e3707ee200000p-4, 0x1.0487b42733b35p-36,  // 91
    0x1.c300000000000p-1, 0x1.03cdc0a400000p-3, 0x1.1ec0d4e78b4fep-35,  // 92
    0x1.c000000000000p-1, 0x1.1178e82200000p-3, 0x1.f91ef78ce2d08p-37,  // 93
    0x1.bd00000000000p-1, 0x1.1f3b925e00000p-3, 0x1.25d41162c9ef9p-35,  // 94
    0x1.ba00000000000p-1, 0x1.2d1610c800000p-3, 0x1.a04e75b32e06dp-37,  // 95
    0x1.b700000000000p-1, 0x1.3b08b67400000p-3, 0x1.7f2a90b86b670p-35,  // 96
    0x1.b400000000000p-1, 0x1.4913d83200000p-3, 0x1.3b560de553f6ep-35,  // 97
    0x1.b200000000000p-1, 0x1.527e5e4a00000p-3, 0x1.b58cfa395a5f7p-39,  // 98
    0x1.af00000000000p-1, 0x1.60b3100a00000p-3, 0x1.09475d49b3b84p-35,  // 99
    0x1.ac00000000000p-1, 0x1.6f0128b600000p-3, 0x1.56abb9c8698f8p-35,  // 100
    0x1.a900000000000p-1, 0x1.7d6903ca00000p-3, 0x1.eb59fca741e7fp-36,  // 101
    0x1.a600000000000p-1, 0x1.8beafeb200000p-3, 0x1.8fe8c2ab5516dp-35,  // 102
    0x1.a400000000000p-1, 0x1.95a5adce00000p-3, 0x1.7017f22858a10p-35,  // 103
    0x1.a100000000000p-1, 0x1.a454082e00000p-3, 0x1.aac14ef903ee3p-37,  // 104
    0x1.9e00000000000p-1, 0x1.b31d857400000p-3, 0x1.bce3ca72b1532p-35,  // 105
    0x1.9c00000000000p-1, 0x1.bd08738200000p-3, 0x1.bd8ad0ee9aafbp-35,  // 106
    0x1.9900000000000p-1, 0x1.cc000c9c00000p-3, 0x1.b3c5254f4550ap-35,  // 107
    0x1.9700000000000p-1, 0x1.d60a17f800000p-3, 0x1.035148fc81ef9p-35,  // 108
    0x1.9400000000000p-1, 0x1.e530effe00000p-3, 0x1.c4048489d8108p-37,  // 109
    0x1.9200000000000p-1, 0x1.ef5ade4c00000p-3, 0x1.cffe5deea9a44p-35,  // 110
    0x1.8f00000000000p-1, 0x1.feb2233e00000p-3, 0x1.40f9a0c6f004ap-36,  // 111
    0x1.8d00000000000p-1, 0x1.047e60cc00000p-2, 0x1.e83b7be21a730p-34,  // 112
    0x1.8a00000000000p-1, 0x1.0c42d67600000p-2, 0x1.62e31162c79d6p-38,  // 113
    0x1.8800000000000p-1, 0x1.1178e82200000p-2, 0x1.f91ef78ce2d08p-36,  // 114
    0x1.8600000000000p-1, 0x1.16b5ccba00000p-2, 0x1.9f6e6b37de946p-35,  // 115
    0x1.8300000000000p-1, 0x1.1e9e167800000p-2, 0x1.133e8a8961ba5p-35,  // 116
    0x1.8100000000000p-1, 0x1.23ec599000000p-2, 0x1.eba4906edd747p-34,  // 117
    0x1.7f00000000000p-1, 0x1.2941afb000000p-2, 0x1.86b7bcf5233c7p-34,  // 118
    0x1.7d00000000000p-1, 0x1.2e9e2bce00000p-2, 0x1.2286018251a3cp-38,  // 119
    0x1.7a00000000000p-1, 0x1.36b6776a00000p-2, 0x1.e1116ecdb0f17p-34,  // 120
    0x1.7800000000000p-1, 0x1.3c25277200000p-2, 0x1.33183b54b606cp-34,  // 121
    0x1.7600000000000p-1, 0x1.419b423c00000p-2, 0x1.5e8c721b76487p-34,  // 122
    0x1.7400000000000p-1, 0x1.4718dc2600000p-2, 0x1.1c41b063ed305p-34,  // 123
    0x1.7200000000000p-1, 0x1.4c9e09e000000p-2, 0x1.72c3beedc9ea5p-34,  // 124
    0x1.7000000000000p-1, 0x1.522ae07200000p-2, 0x1.8a3d7ce102c99p-34,  // 125
    0x1.6e00000000000p-1, 0x1.57bf753c00000p-2, 0x1.1a3f5bdbdcba8p-35,  // 126
    0x1.6c00000000000p-1, 0x1.5d5bddf400000p-2, 0x1.95f2fa6afbaddp-34,  // 127
};
// {2^(j/128), tail / 2^(j/128)}
SM_POW64_TABLE kExpTab[kExpTabDoubles] = {
    0x1.0000000000000p+0, 0x0.0p+0,  // 0
    0x1.0163da9fb3335p+0, 0x1.b3b4f1a88bf6ep-54,  // 1
    0x1.02c9a3e778061p+0, -0x1.160139cd8dc5cp-56,  // 2
    0x1.04315e86e7f85p+0, -0x1.05e7a108766d1p-54,  // 3
    0x1.059b0d3158574p+0, 0x1.cd2523567f613p-55,  // 4
    0x1.0706b29ddf6dep+0, -0x1.bce8023f98efap-55,  // 5
    0x1.0874518759bc8p+0, 0x1.0f74e61e6c861p-57,  // 6
    0x1.09e3ecac6f383p+0, 0x1.0a3e45b33d399p-54,  // 7
    0x1.0b5586cf9890fp+0, 0x1.79aa65d837b6dp-54,  // 8
    0x1.0cc922b7247f7p+0, 0x1.eb51a92fdeffbp-55,  // 9
    0x1.0e3ec32d3d1a2p+0, 0x1.ebe3d702f9cd2p-60,  // 10
    0x1.0fb66affed31bp+0, -0x1.a033489906e0bp-57,  // 11
    0x1.11301d0125b51p+0, -0x1.556522a2fbd0ep-54,  // 12
    0x1.12abdc06c31ccp+0, -0x1.080ef8c4eea54p-58,  // 13
    0x1.1429aaea92de0p+0, -0x1.1c923b9d5f415p-54,  // 14
    0x1.15a98c8a58e51p+0, 0x1.0d3e3e95c55afp-55,  // 15
    0x1.172b83c7d517bp+0, -0x1.01b15eaa59348p-55,  // 16
    0x1.18af9388c8deap+0, -0x1.f1ff055de323dp-55,  // 17
    0x1.1a35beb6fcb75p+0, 0x1.b898c3f1353bfp-55,  // 18
    0x1.1bbe084045cd4p+0, -0x1.6d99c7611eb26p-54,  // 19
    0x1.1d4873168b9aap+0, 0x1.aecf73e3a2f60p-54,  // 20
    0x1.1ed5022fcd91dp+0, -0x1.fe782cb86389ep-55,  // 21
    0x1.2063b88628cd6p+0, 0x1.a6f4144a6c38dp-55,  // 22
    0x1.21f49917ddc96p+0, 0x1.07a05b0e4047dp-55,  // 23
    0x1.2387a6e756238p+0, 0x1.68efde3a8a894p-54,  // 24
    0x1.251ce4fb2a63fp+0, 0x1.75e18f274487dp-55,  // 25
    0x1.26b4565e27cddp+0, 0x1.0472b981fe7f2p-55,  // 26
    0x1.284dfe1f56381p+0, -0x1.6b87b3f71085ep-54,  // 27
    0x1.29e9df51fdee1p+0, 0x1.2f7e16d09ab31p-55,  // 28
    0x1.2b87fd0dad990p+0, -0x1.d219b1a6fbffbp-60,  // 29
    0x1.2d285a6e4030bp+0, 0x1.b3782720c0ab4p-55,  // 30
    0x1.2ecafa93e2f56p+0, 0x1.e149289cecb8ep-57,  // 31
    0x1.306fe0a31b715p+0, 0x1.34d754db0abb6p-55,  // 32
    0x1.32170fc4cd831p+0, 0x1.64201e2ac744cp-55,  // 33
    0x1.33c08b26416ffp+0, 0x1.fdd395dd3f84bp-55,  // 34
    0x1.356c55f929ff1p+0, -0x1.6a3803b8e5b04p-55,  // 35
    0x1.371a7373aa9cbp+0, -0x1.24aedcc4b5069p-54,  // 36
    0x1.38cae6d05d866p+0, -0x1.907f81b512d8ep-54,  // 37
    0x1.3a7db34e59ff7p+0, -0x1.1d1e83e9436d2p-56,  // 38
    0x1.3c32dc313a8e5p+0, -0x1.91919b3ce1b15p-54,  // 39
    0x1.3dea64c123422p+0, 0x1.59f48a72a4c6dp-55,  // 40
    0x1.3fa4504ac801cp+0, -0x1.312607a28698ap-54,  // 41
    0x1.4160a21f72e2ap+0, -0x1.8a78f4817895bp-58,  // 42
    0x1.431f5d950a897p+0, -0x1.c2c9b67499a1cp-56,  // 43
    0x1.44e086061892dp+0, 0x1.363ed60c2ac12p-59,  // 44
    0x1.46a41ed1d0057p+0, 0x1.666093b0664efp-54,  // 45
    0x1.486a2b5c13cd0p+0, 0x1.ecce1daa10378p-57,  // 46
    0x1.4a32af0d7d3dep+0, 0x1.3ff8e3f0f1230p-54,  // 47
    0x1.4bfdad5362a27p+0, 0x1.690cebb7aafb0p-56,  // 48
    0x1.4dcb299fddd0dp+0, 0x1.31dbdeb54e077p-54,  // 49
    0x1.4f9b2769d2ca7p+0, -0x1.f94340071a38ep-55,  // 50
    0x1.516daa2cf6642p+0, -0x1.7deccdc93a349p-55,  // 51
    0x1.5342b569d4f82p+0, -0x1.8dec6bd0f3860p-56,  // 52
    0x1.551a4ca5d920fp+0, -0x1.61246ec7b5cf6p-55,  // 53
    0x1.56f4736b527dap+0, 0x1.3350518fdd78ep-54,  // 54
    0x1.58d12d497c7fdp+0, 0x1.b98b72f8a9b06p-56,  // 55
    0x1.5ab07dd485429p+0, 0x1.063e1e21c5409p-54,  // 56
    0x1.5c9268a5946b7p+0, 0x1.4c7855019c6eap-60,  // 57
    0x1.5e76f15ad2148p+0, 0x1.432e62b64c036p-54,  // 58
    0x1.605e1b976dc09p+0, -0x1.ce44a6199769fp-55,  // 59
    0x1.6247eb03a5585p+0, -0x1.c33c53bef4da8p-55,  // 60
    0x1.6434634ccc320p+0, -0x1.45378892be9aep-55,  // 61
    0x1.6623882552225p+0, -0x1.3cedd78565858p-54,  // 62
    0x1.68155d44ca973p+0, 0x1.710aa807e1964p-58,  // 63
    0x1.6a09e667f3bcdp+0, -0x1.3b3efbf5e2229p-54,  // 64
    0x1.6c012750bdabfp+0, -0x1.a12ad8734b982p-57,  // 65
    0x1.6dfb23c651a2fp+0, -0x1.367efb86da9eep-57,  // 66
    0x1.6ff7df9519484p+0, -0x1.0dc3d54e08851p-55,  // 67
    0x1.71f75e8ec5f74p+0, -0x1.81f647e5a3ecep-56,  // 68
    0x1.73f9a48a58174p+0, -0x1.6ee4ac08b7db0p-55,  // 69
    0x1.75feb564267c9p+0, -0x1.619321e55e68ap-55,  // 70
    0x1.780694fde5d3fp+0, 0x1.09ccb5e09d4d3p-54,  // 71
    0x1.7a11473eb0187p+0, -0x1.b32dcb94da51dp-56,  // 72
    0x1.7c1ed0130c132p+0, 0x1.4ecfd5467c06cp-54,  // 73
    0x1.7e2f336cf4e62p+0, 0x1.5ebe1abd66c55p-57,  // 74
    0x1.80427543e1a12p+0, -0x1.8a1c52fb3cf42p-55,  // 75
    0x1.82589994cce13p+0, -0x1.369b6f13b3734p-54,  // 76
    0x1.8471a4623c7adp+0, -0x1.05e843a19ff1ep-55,  // 77
    0x1.868d99b4492edp+0, -0x1.4d450d872576ep-54,  // 78
    0x1.88ac7d98a6699p+0, 0x1.0ad675b0e8a00p-54,  // 79
    0x1.8ace5422aa0dbp+0, 0x1.db72fc1f0eab5p-55,  // 80
    0x1.8cf3216b5448cp+0, -0x1.5b6609cc5e7ffp-57,  // 81
    0x1.8f1ae99157736p+0, 0x1.bf68359f35f44p-56,  // 82
    0x1.9145b0b91ffc6p+0, -0x1.3091fa71e3d83p-54,  // 83
    0x1.93737b0cdc5e5p+0, -0x1.da9b88b6c1e29p-58,  // 84
    0x1.95a44cbc8520fp+0, -0x1.c23f97c90b959p-57,  // 85
    0x1.97d829fde4e50p+0, -0x1.2434322f4f9aap-54,  // 86
    0x1.9a0f170ca07bap+0, -0x1.5ca6cd7668e4bp-55,  // 87
    0x1.9c49182a3f090p+0, 0x1.1affc2b91ce27p-56,  // 88
    0x1.9e86319e32323p+0, 0x1.dd235e10a73bbp-57,  // 89
    0x1.a0c667b5de565p+0, -0x1.7c50422622263p-55,  // 90
    0x1.a309bec4a2d33p+0, 0x1.b1c86e3e231d5p-55,  // 91
    0x1.a5503b23e255dp+0, -0x1.1bbd1d3bcbb15p-54,  // 92
    0x1.a799e1330b358p+0, 0x1.0cc319cee31d2p-54,  // 93
    0x1.a9e6b5579fdbfp+0, 0x1.469846e735ab3p-55,  // 94
    0x1.ac36bbfd3f37ap+0, -0x1.2dfcd978e9db4p-55,  // 95
    0x1.ae89f995ad3adp+0, 0x1.c1a7792cb3386p-55,  // 96
    0x1.b0e07298db666p+0, -0x1.07b8f4ad1d9fap-54,  // 97
    0x1.b33a2b84f15fbp+0, -0x1.5c3d956dcaebap-58,  // 98
    0x1.b59728de5593ap+0, -0x1.0a40e3da6f640p-54,  // 99
    0x1.b7f76f2fb5e47p+0, -0x1.8d6f438ad9334p-57,  // 100
    0x1.ba5b030a1064ap+0, -0x1.1eee26b588a35p-54,  // 101
    0x1.bcc1e904bc1d2p+0, 0x1.4ffd70a5fddcdp-56,  // 102
    0x1.bf2c25bd71e09p+0, -0x1.1bdfbfa9298acp-54,  // 103
    0x1.c199bdd85529cp+0, 0x1.36eae30af0cb3p-56,  // 104
    0x1.c40ab5fffd07ap+0, 0x1.ee3325c9ffd93p-55,  // 105
    0x1.c67f12e57d14bp+0, 0x1.4e08fd10959acp-55,  // 106
    0x1.c8f6d9406e7b5p+0, 0x1.3cdaf384e1a67p-57,  // 107
    0x1.cb720dcef9069p+0, 0x1.76b2c6c921967p-57,  // 108
    0x1.cdf0b555dc3fap+0, -0x1.08a1883ccb5d2p-55,  // 109
    0x1.d072d4a07897cp+0, -0x1.fad5d3ffffa6ep-55,  // 110
    0x1.d2f87080d89f2p+0, -0x1.00dae3875a949p-54,  // 111
    0x1.d5818dcfba487p+0, 0x1.4a385a63d07a8p-56,  // 112
    0x1.d80e316c98398p+0, -0x1.2919e2040220ep-55,  // 113
    0x1.da9e603db3285p+0, 0x1.e5a50d5c192acp-55,  // 114
    0x1.dd321f301b460p+0, 0x1.43a59ac016b4bp-55,  // 115
    0x1.dfc97337b9b5fp+0, -0x1.2d52107b43e20p-55,  // 116
    0x1.e264614f5a129p+0, -0x1.92ab93b470dc8p-55,  // 117
    0x1.e502ee78b3ff6p+0, 0x1.4b604603a88d3p-56,  // 118
    0x1.e7a51fbc74c83p+0, 0x1.3c5ec519d7271p-55,  // 119
    0x1.ea4afa2a490dap+0, -0x1.ff7128fd391f0p-55,  // 120
    0x1.ecf482d8e67f1p+0, -0x1.dae98e223747dp-55,  // 121
    0x1.efa1bee615a27p+0, 0x1.ec3bc41aa2008p-55,  // 122
    0x1.f252b376bba97p+0, 0x1.42b94c3a9eb32p-55,  // 123
    0x1.f50765b6e4540p+0, 0x1.a64a931d185eep-55,  // 124
    0x1.f7bfdad9cbe14p+0, -0x1.e37bae43be3edp-55,  // 125
    0x1.fa7c1819e90d8p+0, 0x1.7893b4d91cd9dp-56,  // 126
    0x1.fd3c22b8f71f1p+0, 0x1.305c14160cc89p-58,  // 127
};
#undef SM_POW64_TABLE

// 0: not an integer, 1: odd integer, 2: even integer (y finite, non-zero)
SM_POW_FN int int_class(uint64_t iy) {
    const int e = (int)((iy >> 52) & 0x7ff);
    int sh = 1075 - e;                       // fractional bits of |y| when 1023 <= e <= 1075
    sh = sh < 0 ? 0 : (sh > 63 ? 63 : sh);
    const uint64_t frac = iy & ((1ULL << sh) - 1ULL);
    const int odd = (int)((iy >> sh) & 1ULL);
    const int whole = frac ? 0 : (odd ? 1 : 2);
    return e < 1023 ? 0 : (e > 1075 ? 2 : whole);
}

// p * r + c for a compile-time constant c.  On the device the constant rides in an SGPR pair: left to itself the
// compiler forms a two-address v_fmac_f64 and first copies each constant into the VGPR pair it overwrites -- two
// v_mov_b32 per polynomial step, 14 of the ~95 vector instructions of one evaluation.
SM_POW_FN double fma_c(double p, double r, double c) {
#if defined(__HIP_DEVICE_COMPILE__)
    double d;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(p), "v"(r), "s"(c));
    return d;
#else
    return SM_POW_FMA(p, r, c);
#endif
}
// 4 r + c (4.0 is an inline operand)
SM_POW_FN double fma4_c(double r, double c) {
#if defined(__HIP_DEVICE_COMPILE__)
    double d;
    asm("v_fma_f64 %0, 4.0, %1, %2" : "=v"(d) : "v"(r), "s"(c));
    return d;
#else
    return SM_POW_FMA(4.0, r, c);
#endif
}
#if defined(__HIP_DEVICE_COMPILE__)
#define SM_POW_FMAX(a, b) __builtin_fmax((a), (b))
#define SM_POW_FMIN(a, b) __builtin_fmin((a), (b))
#else
#define SM_POW_FMAX(a, b) fmax((a), (b))
#define SM_POW_FMIN(a, b) fmin((a), (b))
#endif

// Where the evaluation finds its table entries.
//   TabAoS     the tables as generated: {invc, logc_hi, logc_lo} / {T, tail} per entry (kLogTab / kExpTab, or an LDS copy
//              of them): three LDS reads per element, and across a wave's scattered indices they collide -- entry i starts
//              in bank 6 i mod 32, so 64 random lookups share 16 starting banks (43 % of the LDS-active cycles of the
//              scalar-exponent kernel were bank conflicts, profiles/r02_pow64_rate.txt).
//   TabBanked  sixteen replicas of every value, one per PAIR OF BANKS: value (array a, entry i) for replica c lives at
//              double index (a * 128 + i) * 16 + c, lane l reads replica l mod 16 -- whatever the indices, sixteen
//              neighbouring lanes hit sixteen different bank pairs: no lookup can conflict.  80 KiB of LDS, so the kernel
//              that uses it runs 1024-thread workgroups, two per CU (ops.hip.h: PowBanked).
#if defined(__HIPCC__)
#define SM_POW_MEMFN __host__ __device__ __forceinline__
#else
#define SM_POW_MEMFN inline
#endif
struct TabAoS {
    const double *logtab, *exptab;
    SM_POW_MEMFN double invc(int i) const { return logtab[3 * i]; }
    SM_POW_MEMFN double logc(int i) const { return logtab[3 * i + 1]; }
    SM_POW_MEMFN double logctail(int i) const { return logtab[3 * i + 2]; }
    SM_POW_MEMFN double th(int j) const { return exptab[2 * j]; }
    SM_POW_MEMFN double trel(int j) const { return exptab[2 * j + 1]; }
};
constexpr int kBankedReplicas = 16, kBankedArrays = 5, kBankedDoubles = kBankedArrays * kN * kBankedReplicas;
struct TabBanked {
    const double *mine;  // the LDS copy + (lane & 15)
    SM_POW_MEMFN double invc(int i) const { return mine[(0 * kN + i) * kBankedReplicas]; }
    SM_POW_MEMFN double logc(int i) const { return mine[(1 * kN + i) * kBankedReplicas]; }
    SM_POW_MEMFN double logctail(int i) const { return mine[(2 * kN + i) * kBankedReplicas]; }
    SM_POW_MEMFN double th(int j) const { return mine[(3 * kN + j) * kBankedReplicas]; }
    SM_POW_MEMFN double trel(int j) const { return mine[(4 * kN + j) * kBankedReplicas]; }
};
// value `a` (0 invc, 1 logc_hi, 2 logc_lo, 3 T, 4 tail) of entry i, from the generated tables
SM_POW_FN double table_value(int a, int i) { return a < 3 ? kLogTab[3 * i + a] : kExpTab[2 * i + (a - 3)]; }

// ax finite > 0, y finite.  SUB = false: ax is known to be normal (the callers route subnormal bases through the general path).
template <bool SUB, typename TAB>
SM_POW_FN double pow_core_t(double ax, double y, const TAB &tab) {
    uint32_t hi = (uint32_t)(f64_bits(ax) >> 32), lw = (uint32_t)f64_bits(ax);
    int sub = 0;
    if constexpr (SUB) {
        if (hi < 0x00100000u) {  // subnormal: normalise
            const uint64_t nx = f64_bits(ax * 0x1p52);
            hi = (uint32_t)(nx >> 32);
            lw = (uint32_t)nx;
            sub = 52;
        }
    }
    static_assert((kOff & 0xffffffffULL) == 0, "the interval arithmetic below works on the high word alone");
    const uint32_t tmp = hi - (uint32_t)(kOff >> 32);
    const int i = (int)((tmp >> (20 - 7)) & (kN - 1));
    const int k = ((int32_t)tmp >> 20) - sub;
    const double z = smpow::make_f64(hi - (tmp & 0xfff00000u), lw);
    const double invc = tab.invc(i), logc = tab.logc(i), logctail = tab.logctail(i);
    const double kd = (double)k;
    const double Ln2hi = 0x1.62e42f8000000p-1, Ln2lo = 0x1.be8e7bcd5e4f2p-27;
    // r = z*invc - 1 needs up to 54 bits (53 + 9 - the ~8 that cancel): carry the last one as rlo
    const double ph = z * invc, pl = SM_POW_FMA(z, invc, -ph);  // z*invc = ph + pl exactly
    const double rm = ph - 1.0;                                 // exact (Sterbenz)
    const double r = rm + pl, rlo = (rm - r) + pl;
    const double t1 = SM_POW_FMA(kd, Ln2hi, logc);  // exact
    const double t2 = t1 + r;
    const double lo1 = SM_POW_FMA(kd, Ln2lo, logctail);
    const double lo2 = (t1 - t2) + r;
    const double ar = -0.5 * r, ar2 = r * ar;
    const double hi2 = t2 + ar2;
    const double lo3 = SM_POW_FMA(ar, r, -ar2);
    const double lo4 = (t2 - hi2) + ar2;
    // cubic and higher terms, r^3 (1/3 - r/4 + r^2/5 - r^3/6 + r^4/7 - r^5/8), as a polynomial in s = -r/2 (= ar):
    // r^3 (1/3 + s/2 + 4 s^2/5 + 4 s^3/3 + 16 s^4/7 + 4 s^5) -- the leading step then multiplies by an inline constant
    double p = fma4_c(ar, 0x1.2492492492492p+1);               // 16/7
    p = fma_c(p, ar, 0x1.5555555555555p+0);                    //  4/3
    p = fma_c(p, ar, 0x1.999999999999ap-1);                    //  4/5
    p = SM_POW_FMA(p, ar, 0.5);
    p = fma_c(p, ar, 0x1.5555555555555p-2);                    //  1/3
    const double r3 = (r * r) * r;
    const double lo = SM_POW_FMA(p, r3, (((lo1 + lo2) + lo3) + lo4) + rlo);  // d ln(1+r)/dr ~ 1: rlo enters at first order
    const double lhi = hi2 + lo, ltail = (hi2 - lhi) + lo;     // ln(ax) = lhi + ltail

    const double ehi = y * lhi;
    // beyond +-1500 the result is 0 / inf whatever the fraction (and ehi itself may have overflowed): clamp.  The low
    // part is below 2^-40 for any in-range ehi; out of range it is garbage (up to NaN: inf - inf), and confining it to
    // [-1, 1] (fmax / fmin return the other operand for a NaN) keeps 1 + q positive, which is all 2^+-2164 needs.
    const double elo = SM_POW_FMIN(SM_POW_FMAX(SM_POW_FMA(y, ltail, SM_POW_FMA(y, lhi, -ehi)), -1.0), 1.0);
    const double InvLn2N = 0x1.71547652b82fep+7, Ln2hiN = 0x1.62e42f8000000p-8, Ln2loN = 0x1.be8e7bcd5e4f2p-34;
    const double ec = SM_POW_FMIN(SM_POW_FMAX(ehi, -1500.0), 1500.0);  // keeps kd2 * Ln2hiN exact
    const double kd2 = SM_POW_RINT(ec * InvLn2N);
    const int ki = smpow::sat_i32(kd2);
    double rr = SM_POW_FMA(-kd2, Ln2hiN, ec);
    rr = SM_POW_FMA(-kd2, Ln2loN, rr);
    rr += elo;
    const int j = ki & (kN - 1), e = ki >> 7;  // arithmetic shift: floor
    const double th = tab.th(j), trel = tab.trel(j);
    const double r2 = rr * rr;
    // trel + rr + rr^2 (1/2 + rr (1/6 + rr (1/24 + rr/120))); the innermost step as (rr + 5)/120: one constant per instruction
    double u = (rr + 5.0) * 0x1.1111111111111p-7;
    u = fma_c(u, rr, 0x1.5555555555555p-3);
    u = SM_POW_FMA(u, rr, 0.5);
    const double q = SM_POW_FMA(u, r2, trel + rr);
    return SM_POW_LDEXP(SM_POW_FMA(th, q, th), e);
}

// Every operand class: the core on stand-in operands, then the C99 F.9.4.4 lattice.
template <typename TAB>
SM_POW_FN double pow_general(double x, double y, const TAB &tab) {
    const uint64_t ONE = 0x3ff0000000000000ULL, INF = 0x7ff0000000000000ULL, QNAN = 0x7ff8000000000000ULL;
    const uint64_t ix = f64_bits(x), iy = f64_bits(y);
    const uint64_t ax = ix & 0x7fffffffffffffffULL, ay = iy & 0x7fffffffffffffffULL;
    const double axc = bits_f64((ax == 0 || ax >= INF) ? ONE : ax);
    const double yc_d = ay >= INF ? 1.0 : y;
    const double core = pow_core_t<true>(axc, yc_d, tab);
    const bool x_neg = (ix >> 63) != 0, y_neg = (iy >> 63) != 0;
    const bool x_nan = ax > INF, y_nan = ay > INF;
    const bool x_one = ix == ONE, y_zero = ay == 0;
    const int yc = int_class(iy);
    const uint64_t sign = (x_neg && yc == 1) ? 0x8000000000000000ULL : 0ULL;
    uint64_t r = f64_bits(core) | sign;
    r = (x_neg && yc == 0) ? QNAN : r;
    r = ax == INF ? (sign | (y_neg ? 0ULL : INF)) : r;
    r = ax == 0 ? (sign | (y_neg ? INF : 0ULL)) : r;
    r = ay == INF ? (ax == ONE ? ONE : (((ax < ONE) == y_neg) ? INF : 0ULL)) : r;
    r = (x_one || y_zero) ? ONE : r;
    const bool snan = (x_nan && !(ix & 0x0008000000000000ULL)) || (y_nan && !(iy & 0x0008000000000000ULL));
    const uint64_t nan_r = snan ? QNAN : ((x_one || y_zero) ? ONE : QNAN);
    r = (x_nan || y_nan) ? nan_r : r;
    return bits_f64(r);
}

// How far an operand's high word is from "ordinary": x positive and normal, y finite and of normal magnitude.  Both
// classes are visible in the high 32 bits; anything else (zeros, subnormals, negatives, infinities, NaNs) yields a value
// >= kOrdinarySpan and takes pow_general, which is correct for every operand.
constexpr uint32_t kMinNormalHi = 0x00100000u, kOrdinarySpan = 0x7ff00000u - kMinNormalHi;
SM_POW_FN uint32_t oddness(double x, double y) {
    const uint32_t hx = (uint32_t)(f64_bits(x) >> 32), hy = (uint32_t)(f64_bits(y) >> 32) & 0x7fffffffu;
    const uint32_t dx = hx - kMinNormalHi, dy = hy - kMinNormalHi;
    return dx > dy ? dx : dy;
}

SM_POW_FN double pow(double x, double y, const double *logtab, const double *exptab) {
    const TabAoS tab{logtab, exptab};
    if (!smpow::any_lane(oddness(x, y) >= kOrdinarySpan)) return pow_core_t<false>(x, y, tab);
    return pow_general(x, y, tab);
}

// x^y for W independent pairs held in registers: one test (and one branch) for the group.
template <int W, typename TAB>
SM_POW_FN void pow_n(const double (&x)[W], const double (&y)[W], double (&out)[W], const TAB &tab) {
    uint32_t worst = 0;
#pragma unroll
    for (int k = 0; k < W; ++k) {
        const uint32_t o = oddness(x[k], y[k]);
        worst = o > worst ? o : worst;
    }
    if (!smpow::any_lane(worst >= kOrdinarySpan)) {
#pragma unroll
        for (int k = 0; k < W; ++k) out[k] = pow_core_t<false>(x[k], y[k], tab);
        return;
    }
    // operands opaque from here on, so that nothing of the lattice is hoisted above the branch (sm_pow.h: pow_n)
#pragma unroll
    for (int k = 0; k < W; ++k) {
        double xs = x[k], ys = y[k];
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("" : "+v"(xs), "+v"(ys));
#endif
        out[k] = pow_general(xs, ys, tab);
    }
}
template <int W>
SM_POW_FN void pow_n(const double (&x)[W], const double (&y)[W], double (&out)[W], const double *logtab, const double *exptab) {
    pow_n<W, TabAoS>(x, y, out, TabAoS{logtab, exptab});
}


// ---------------------------------------------------------------------------------------------------------------------
// ONE exponent of moderate magnitude for a whole array (sm::pow(a, 2.7): array_scalar_op, calculate.h:137-169, with
// PowOp<double>::apply, pow.h:8-10).  The evaluation above is built for ANY pair (x, y): |y ln x| may overflow, y may be
// 2^60, and every error term of ln x is carried because it is multiplied by an unknown y.  The general form is bound by its
// vector instructions (profiles/r02_pow64_rate.txt: the VALU busy 79 % of the kernel at ~75 instructions per element), so
// knowing the exponent on the host buys time directly:
//   LEVEL 1, |y| <= 1024:  E = y ln x stays below 2^20, so nothing overflows and nothing needs clamping (4 instructions);
//            ln x need not be renormalised before the product (y * hi, then y * lo joins the low part: 3); rounding E N / ln 2
//            to an integer is one fma against 1.5 * 2^52 whose low word IS the integer (glibc's exp idiom: 2 instead of
//            mul + rndne + cvt); the residual of r * (-r/2), < 2^-69, times 1024 is 2^-59: dropped (1).
//   LEVEL 2, |y| <= 16:  what ln x must carry shrinks with y.  r = fma(z, invc, -1) in ONE rounding (the bit it can lose,
//            <= 2^-61.6, times 16 is 0.04 ULP of the result) instead of the exact product and its three-term
//            renormalisation (5); r - r^2/2 summed first and added to k ln 2 + log c with ONE error-free step instead of
//            two (2).  37 fp64 instructions instead of 55.
// Bases that are not positive normal numbers go through pow_general like everywhere else.  The kernels read the tables as five
// plain arrays (TabSoA).  Host check: tests/cpp/pow64_host_check.cpp
// (each level over its range of exponents against glibc's pow, <= 1 ULP).
struct TabSoA {  // five arrays of kN doubles in LDS: invc | logc | logctail | T | tail
    // Entry i of every array starts in bank pair i mod 16, so the sixteen lanes of one LDS pass meet sixteen different starting
    // positions (the generated {invc, logc, logctail} triples: the same sixteen; triples padded to 32 bytes for one b128 read: FOUR --
    // 70 % of the LDS-active cycles were bank conflicts, profiles/r04_pow64_rate.txt).  The interval's BYTE offset is a shift and a
    // mask of the high word; the compiler pairs the reads 1 KiB apart into ds_read2st64_b64.
    const double *base;
    SM_POW_MEMFN double invc(int i) const { return base[i]; }
    SM_POW_MEMFN double logc(int i) const { return base[kN + i]; }
    SM_POW_MEMFN double logctail(int i) const { return base[2 * kN + i]; }
    SM_POW_MEMFN double th(int j) const { return base[3 * kN + j]; }
    SM_POW_MEMFN double trel(int j) const { return base[4 * kN + j]; }
    SM_POW_MEMFN const double *log_entry(uint32_t tmp) const {  // the interval encoded in tmp's bits 13..19
        return reinterpret_cast<const double *>(reinterpret_cast<const char *>(base) + ((tmp >> (20 - 7 - 3)) & ((uint32_t)(kN - 1) << 3)));
    }
};
constexpr int kSoADoubles = 5 * kN;  // value `a` of entry i at a * kN + i (table_value)
// 0: the general evaluation; 1: |y| <= 1024; 2: |y| <= 16 (y finite and of normal magnitude in both)
inline int scalar_level(double y) {
    const uint64_t ay = f64_bits(y) & 0x7fffffffffffffffULL;
    if (ay < 0x0010000000000000ULL || ay > 0x4090000000000000ULL) return 0;  // zero / subnormal, or beyond 1024 (infinities, NaNs)
    return ay <= 0x4030000000000000ULL ? 2 : 1;
}
template <int LEVEL, typename TAB>
SM_POW_FN double pow_core_u(double ax, double y, const TAB &tab) {
    static_assert(LEVEL == 1 || LEVEL == 2, "see scalar_level");
    const uint32_t hi = (uint32_t)(f64_bits(ax) >> 32), lw = (uint32_t)f64_bits(ax);
    const uint32_t tmp = hi - (uint32_t)(kOff >> 32);
    const int i = (int)((tmp >> (20 - 7)) & (kN - 1));
    const int k = (int32_t)tmp >> 20;
    const double z = smpow::make_f64(hi - (tmp & 0xfff00000u), lw);
    double invc, logc, logctail;
    if constexpr (std::is_same<TAB, TabSoA>::value) {
        const double *e = tab.log_entry(tmp);
        invc = e[0]; logc = e[kN]; logctail = e[2 * kN];
    } else {
        invc = tab.invc(i); logc = tab.logc(i); logctail = tab.logctail(i);
    }
    const double kd = (double)k;
    const double Ln2hi = 0x1.62e42f8000000p-1, Ln2lo = 0x1.be8e7bcd5e4f2p-27;
    const double t1 = SM_POW_FMA(kd, Ln2hi, logc);  // exact
    const double lo1 = SM_POW_FMA(kd, Ln2lo, logctail);
    double r, lhi, low;
    if constexpr (LEVEL == 2) {
        r = SM_POW_FMA(z, invc, -1.0);
        const double ar2 = r * (-0.5 * r);
        const double s = r + ar2;
        lhi = t1 + s;
        low = lo1 + ((t1 - lhi) + s);
    } else {
        const double ph = z * invc, pl = SM_POW_FMA(z, invc, -ph);
        const double rm = ph - 1.0;
        r = rm + pl;
        const double rlo = (rm - r) + pl;
        const double t2 = t1 + r;
        const double lo2 = (t1 - t2) + r;
        const double ar2 = r * (-0.5 * r);
        lhi = t2 + ar2;
        const double lo4 = (t2 - lhi) + ar2;
        low = ((lo1 + lo2) + lo4) + rlo;
    }
    const double ar = -0.5 * r;
    double p = fma4_c(ar, 0x1.2492492492492p+1);               // 16/7   (the polynomial of pow_core_t)
    p = fma_c(p, ar, 0x1.5555555555555p+0);                    //  4/3
    p = fma_c(p, ar, 0x1.999999999999ap-1);                    //  4/5
    p = SM_POW_FMA(p, ar, 0.5);
    p = fma_c(p, ar, 0x1.5555555555555p-2);                    //  1/3
    const double r3 = (r * r) * r;
    const double lo = SM_POW_FMA(p, r3, low);                  // ln(ax) = lhi + lo, lo up to 2^-23: not renormalised

    const double ehi = y * lhi;
    const double elo = SM_POW_FMA(y, lo, SM_POW_FMA(y, lhi, -ehi));  // |y lo| <= 2^-13
    const double InvLn2N = 0x1.71547652b82fep+7, Ln2hiN = 0x1.62e42f8000000p-8, Ln2loN = 0x1.be8e7bcd5e4f2p-34;
    const double Shift = 0x1.8p52;
    const double ks = SM_POW_FMA(ehi, InvLn2N, Shift);        // |ehi InvLn2N| < 2^28: the integer sits in the low word
    const int ki = (int32_t)(uint32_t)f64_bits(ks);
    const double kd2 = ks - Shift;
    double rr = SM_POW_FMA(-kd2, Ln2hiN, ehi);
    rr = SM_POW_FMA(-kd2, Ln2loN, rr);
    rr += elo;
    const int j = ki & (kN - 1), e = ki >> 7;
    const double th = tab.th(j), trel = tab.trel(j);
    const double r2 = rr * rr;
    double u = (rr + 5.0) * 0x1.1111111111111p-7;
    u = fma_c(u, rr, 0x1.5555555555555p-3);
    u = SM_POW_FMA(u, rr, 0.5);
    const double q = SM_POW_FMA(u, r2, trel + rr);
    return SM_POW_LDEXP(SM_POW_FMA(th, q, th), e);             // e up to +-2^13: v_ldexp_f64 saturates to 0 / inf
}
// W bases held in registers, one exponent for which scalar_level(y) == LEVEL
template <int LEVEL, int W, typename TAB>
SM_POW_FN void pow_scalar_n(const double (&x)[W], double y, double (&out)[W], const TAB &tab) {
    uint32_t worst = 0;
#pragma unroll
    for (int k = 0; k < W; ++k) {
        const uint32_t o = (uint32_t)(f64_bits(x[k]) >> 32) - kMinNormalHi;
        worst = o > worst ? o : worst;
    }
    if (!smpow::any_lane(worst >= kOrdinarySpan)) {
#pragma unroll
        for (int k = 0; k < W; ++k) out[k] = pow_core_u<LEVEL>(x[k], y, tab);
        return;
    }
#pragma unroll
    for (int k = 0; k < W; ++k) {
        double xs = x[k], ys = y;
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("" : "+v"(xs), "+v"(ys));
#endif
        out[k] = pow_general(xs, ys, tab);
    }
}
// by run-time exponent (host check; a kernel's scalar tail)
SM_POW_FN double pow_scalar(double x, double y, int level, const double *logtab, const double *exptab) {
    const TabAoS tab{logtab, exptab};
    const double xs[1] = {x};
    double r[1];
    if (level == 2) pow_scalar_n<2, 1>(xs, y, r, tab);
    else if (level == 1) pow_scalar_n<1, 1>(xs, y, r, tab);
    else r[0] = pow(x, y, logtab, exptab);
    return r[0];
}

// ---------------------------------------------------------------------------------------------------------------------
// Scalar exponents that are a multiple of one half, |y| <= 8 (sm::pow(a, 2.5), a cube, an inverse square root ...):
// x = m 2^e with e even when y is not an integer, so that x^y = m^y 2^(e y) with an INTEGER e y.  m^|y| is a product
// chain in double-double arithmetic (squarings and multiplications of (hi, lo) pairs, each exact to ~2^-104; the square
// root to ~2^-73: a seed, two Newton steps, one exact residual), a negative exponent one double-double reciprocal, and
// the power of two goes on with v_ldexp_f64.  No table, no logarithm: ~30 vector instructions per element for y = 2.5
// against ~65 plus five scattered LDS reads for the general form -- which makes the operator HBM-bound.  The value
// rounded at the end is within 2^-72 of the true power, so the result is the correctly rounded power except once in
// ~2^19 elements, where it is its neighbour: <= 1 ULP from libm's pow like the general form (tests/cpp/pow_halfint_host_check.cpp:
// every exponent -8 ... 8 in steps of one half over the whole range of x, subnormals and the special-case lattice included).
struct DD { double hi, lo; };
SM_POW_FN DD dd_renorm(double p, double e) { const double hi = p + e; return DD{hi, e - (hi - p)}; }  // |e| <= |p| ulp-wise
SM_POW_FN DD dd_sqr(DD a) {
    const double p = a.hi * a.hi;
    double e = SM_POW_FMA(a.hi, a.hi, -p);
    e = SM_POW_FMA(a.hi + a.hi, a.lo, e);
    return dd_renorm(p, e);
}
SM_POW_FN DD dd_mul(DD a, DD b) {
    const double p = a.hi * b.hi;
    double e = SM_POW_FMA(a.hi, b.hi, -p);
    e = SM_POW_FMA(a.hi, b.lo, e);
    e = SM_POW_FMA(a.lo, b.hi, e);
    return dd_renorm(p, e);
}
// seeds good to 2^-20 or better: the device's v_rsq_f64 / v_rcp_f64; on the host a float-rounded value stands in for them
SM_POW_FN double seed_rsq(double m) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rsq(m);
#else
    return (double)(float)(1.0 / sqrt(m));
#endif
}
SM_POW_FN double seed_rcp(double m) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rcp(m);
#else
    return (double)(float)(1.0 / m);
#endif
}
SM_POW_FN DD dd_sqrt(double m) {  // m in [0.5, 2)
    const double rs = seed_rsq(m), h = 0.5 * rs;
    double s = m * rs;
    s = SM_POW_FMA(SM_POW_FMA(-s, s, m), h, s);
    s = SM_POW_FMA(SM_POW_FMA(-s, s, m), h, s);
    return DD{s, SM_POW_FMA(-s, s, m) * h};
}
SM_POW_FN DD dd_recip(DD a) {  // a.hi normal, far from the ends of the range
    double q = seed_rcp(a.hi);
    q = SM_POW_FMA(q, SM_POW_FMA(-a.hi, q, 1.0), q);
    q = SM_POW_FMA(q, SM_POW_FMA(-a.hi, q, 1.0), q);
    const double r = SM_POW_FMA(-a.lo, q, SM_POW_FMA(-a.hi, q, 1.0));
    return dd_renorm(q, q * r);
}
// Is y one of these exponents?  *m2 = 2 y.  (y = 0 and the exponents whose power is one IEEE operation are served elsewhere.)
SM_POW_FN bool halfint_exponent(double y, int *m2) {
    const double t = y + y;
    if (!(t >= -16.0 && t <= 16.0) || t != SM_POW_RINT(t) || t == 0.0) return false;
    *m2 = (int)t;
    return true;
}
// W bases held in registers, one exponent: the control flow depends on the exponent alone.
// CM2 != 0: the exponent (times two) as a compile-time constant -- the chain becomes straight-line code and the compiler
// interleaves the W (and the caller's U x W) independent chains; this is the form the kernels use (one instantiation per
// exponent).  With a run-time exponent the same code is a loop with branches: pow(a, 2.5) at N = 2^26 174 us against
// 218 us with wave-uniform branches per element and 183 us with the W chains inside one pass of the branches.
template <int W, int CM2 = 0>
SM_POW_FN void pow_halfint_n(const double (&x)[W], int m2, double (&out)[W]) {
    if (CM2 != 0) m2 = CM2;
    const int am = m2 < 0 ? -m2 : m2, n = am >> 1;
    const bool half = (am & 1) != 0, neg_y = m2 < 0;
    double ax[W], m[W];
    int e[W];
#pragma unroll
    for (int w = 0; w < W; ++w) {
        ax[w] = bits_f64(f64_bits(x[w]) & 0x7fffffffffffffffULL);
#if defined(__HIP_DEVICE_COMPILE__)
        m[w] = __builtin_amdgcn_frexp_mant(ax[w]);  // [0.5, 1), subnormals included
        e[w] = __builtin_amdgcn_frexp_exp(ax[w]);
#else
        m[w] = frexp(ax[w], &e[w]);
#endif
        if (half) {  // an even e for half-integer exponents: m in [0.5, 2)
            const int adj = e[w] & 1;
            m[w] = adj ? m[w] + m[w] : m[w];
            e[w] -= adj;
        }
    }
    // m^n: square-and-multiply over the bits of n
    DD b[W], p[W];
#pragma unroll
    for (int w = 0; w < W; ++w) { b[w] = DD{m[w], 0.0}; p[w] = DD{1.0, 0.0}; }
    bool have = false;
    for (int k = n; k; k >>= 1) {
        if (k & 1) {
#pragma unroll
            for (int w = 0; w < W; ++w) p[w] = have ? dd_mul(p[w], b[w]) : b[w];
            have = true;
        }
        if (k >> 1) {
#pragma unroll
            for (int w = 0; w < W; ++w) b[w] = dd_sqr(b[w]);
        }
    }
    if (half) {
#pragma unroll
        for (int w = 0; w < W; ++w) { const DD s = dd_sqrt(m[w]); p[w] = have ? dd_mul(p[w], s) : s; }
    }
    if (neg_y) {
#pragma unroll
        for (int w = 0; w < W; ++w) p[w] = dd_recip(p[w]);
    }
    const double inf = bits_f64(0x7ff0000000000000ULL);
#pragma unroll
    for (int w = 0; w < W; ++w) {
        const int scale = (e[w] * m2) / 2;  // exact: e is even whenever m2 is odd
#if defined(__HIP_DEVICE_COMPILE__)
        double r = __builtin_ldexp(p[w].hi + p[w].lo, scale);
#else
        double r = ldexp(p[w].hi + p[w].lo, scale);
#endif
        // zeros, infinities, negative bases (C99 F.9.4.4); NaN went through the arithmetic
        const bool x_zero = ax[w] == 0.0, x_inf = ax[w] == inf, x_neg = (f64_bits(x[w]) >> 63) != 0;
        r = x_zero ? (neg_y ? inf : 0.0) : r;
        r = x_inf ? (neg_y ? 0.0 : inf) : r;
        if (!half) r = (x_neg && (n & 1)) ? -r : r;                                          // integer exponent: the sign of an odd power
        else r = (x_neg && !x_zero && !x_inf) ? bits_f64(0x7ff8000000000000ULL) : r;           // a negative base to a fractional power
        out[w] = r;
    }
}
SM_POW_FN double pow_halfint(double x, int m2) {
    const double xs[1] = {x};
    double r[1];
    pow_halfint_n<1>(xs, m2, r);
    return r[0];
}
// the compile-time forms by run-time exponent (host check; a kernel's scalar tail)
template <int LO, int HI>
SM_POW_FN double pow_halfint_switch(double x, int m2) {
    if constexpr (LO == HI) {
        const double xs[1] = {x};
        double r[1];
        pow_halfint_n<1, LO == 0 ? 1 : LO>(xs, m2, r);
        return r[0];
    } else {
        constexpr int MID = LO + (HI - LO) / 2;
        return m2 <= MID ? pow_halfint_switch<LO, MID>(x, m2) : pow_halfint_switch<MID + 1, HI>(x, m2);
    }
}

}  // namespace smpow64
