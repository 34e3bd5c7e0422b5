// helpers.h -- rank limit, contiguity predicate and register-group traits
// (drop-in for the reference's include/math/helpers.h).
//
// The reference's SimdTraits<T> describes x86 registers (__m128/__m256/__m512
// and their load/store/set1).  On MI355X the unit a lane moves is one 16-byte
// register group -- float4 / double2 / int4 / long2 -- and the loads live in
// libsmhip's kernels, so the trait keeps only what host code can ask about.
#pragma once

#include <cstddef>
#include <cstdint>
#include <type_traits>
#include <vector>

// The reference's helpers.h:2 includes <immintrin.h>, and plugin headers written against it (README.md:105-117:
// `inline __m256 MyOp<float>::apply_simd<__m256>(...)`) rely on that for __m256 and the _mm256_* intrinsics.  On an x86
// host they keep parsing; the specialisations are simply never called -- the loop runs on the MI355X.
#if defined(__x86_64__) || defined(__i386__) || defined(_M_X64)
#include <immintrin.h>
#define SM_HAVE_X86_SIMD 1
#endif

#ifndef MAX_NDIM
#define MAX_NDIM 6  // same limit as the reference (helpers.h:4) and SMHIP_MAX_NDIM
#endif

template <typename>
struct dependent_false : std::false_type {};

template <typename T>
struct SimdTraits {
    static constexpr std::size_t register_bytes = 16;                    // one dwordx4 per lane
    static constexpr std::size_t simd_width = register_bytes / sizeof(T);  // elements per lane per access
    static constexpr std::size_t wave_width = 64 * simd_width;           // elements per wavefront access (1 KiB)
};
#ifdef SM_HAVE_X86_SIMD
// The x86 register names the reference's traits publish (helpers.h:24-27, 58-61, 90-93), for plugin headers that spell
// their specialisations through them.
template <> struct SimdTraits<float> { using m128 = __m128; using m256 = __m256; using m512 = __m512;
    static constexpr std::size_t register_bytes = 16, simd_width = 4, wave_width = 256; };
template <> struct SimdTraits<double> { using m128 = __m128d; using m256 = __m256d; using m512 = __m512d;
    static constexpr std::size_t register_bytes = 16, simd_width = 2, wave_width = 128; };
template <> struct SimdTraits<std::int32_t> { using m128 = __m128i; using m256 = __m256i; using m512 = __m512i;
    static constexpr std::size_t register_bytes = 16, simd_width = 4, wave_width = 256; };
#endif

// Dense row-major test (reference helpers.h:130-139).
inline bool is_contiguous(const std::vector<std::size_t> &shape, const std::vector<std::size_t> &stride) {
    std::size_t expect = 1;
    for (std::size_t i = shape.size(); i-- > 0;) {
        if (stride[i] != expect) return false;
        expect *= shape[i];
    }
    return true;
}
