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

// Dense row-major test (reference helpers.h:130-139).
inline bool is_contiguous(const std::vector<std::size_t> &shape, const std::vector<std::size_t> &stride) {
    std::size_t expect = 1;
    for (std::size_t i = shape.size(); i-- > 0;) {
        if (stride[i] != expect) return false;
        expect *= shape[i];
    }
    return true;
}
