// product.h -- dot_product<T>, the library's reduction (drop-in for the
// reference's include/math/product.h:8-224; reached through SMArray::operator%).
//
// Host pointers in, one T out, synchronous -- computed by libsmhip's reduction
// kernels (smhip_dot): wavefront shuffle tree + LDS across waves, accumulating in
// fp64 for float/double (the reference's 8 f32 lane accumulators stop absorbing
// addends at 2^24 each) and in wrapping integers for every integer type -- int32 /
// int64 and the generic template's 8- / 16-bit and unsigned types (product.h:8-20) --
// bit-identical to the reference in any order.  std::complex<double> (product.h:168-224) runs
// smhip_dot_c64: sum a[i]*b[i], unconjugated, separate fp64 fma chains for the real
// and imaginary parts.
#pragma once

#include <complex>
#include <cstddef>
#include <stdexcept>

#include "calculate.h"

namespace sm::hip {
template <typename T>
T dot_device(const T *a, const T *b, std::size_t n) {
    static_assert(dot_dtype_of<T>::id >= 0, "dot_product: element type has no kernels");
    T out{};
    check(smhip_dot(dot_dtype_of<T>::id, a, b, n, &out));
    return out;
}
// std::complex<double>: device pointers to n {re, im} pairs
inline std::complex<double> dot_device_c64(const std::complex<double> *a, const std::complex<double> *b, std::size_t n) {
    double out[2] = {0, 0};
    check(smhip_dot_c64(a, b, n, out));
    return {out[0], out[1]};
}
// std::complex<float> (the generic template's instantiation): device pointers to n {re, im} float pairs
inline std::complex<float> dot_device_c32(const std::complex<float> *a, const std::complex<float> *b, std::size_t n) {
    float out[2] = {0, 0};
    check(smhip_dot_c32(a, b, n, out));
    return {out[0], out[1]};
}
}  // namespace sm::hip

template <typename T>
T dot_product(const T *a, const T *b, std::size_t n) {
    using namespace sm::hip;
    if constexpr (std::is_same_v<T, std::complex<float>>) {
        if (n == 0) return T{};
        DeviceBuffer da(n * sizeof(T)), db(n * sizeof(T));
        check(smhip_upload(da.get(), a, n * sizeof(T)));
        check(smhip_upload(db.get(), b, n * sizeof(T)));
        return dot_device_c32(da.template as<T>(), db.template as<T>(), n);
    } else if constexpr (std::is_same_v<T, std::complex<double>>) {
        if (n == 0) return T{};
        DeviceBuffer da(n * sizeof(T)), db(n * sizeof(T));
        check(smhip_upload(da.get(), a, n * sizeof(T)));
        check(smhip_upload(db.get(), b, n * sizeof(T)));
        return dot_device_c64(da.template as<T>(), db.template as<T>(), n);
    } else if constexpr (dot_dtype_of<T>::id >= 0) {
        if (n == 0) return T{};
        DeviceBuffer da(n * sizeof(T)), db(n * sizeof(T));
        check(smhip_upload(da.get(), a, n * sizeof(T)));
        check(smhip_upload(db.get(), b, n * sizeof(T)));
        return dot_device<T>(da.template as<T>(), db.template as<T>(), n);
    } else {
        throw std::runtime_error("dot_product: this element type has no gfx950 kernels yet (no CPU fallback)");
    }
}
