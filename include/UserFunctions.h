// UserFunctions.h -- free functions of the drop-in surface: sm::empty / ones /
// zeros, sm::pow, operator<< (reference include/UserFunctions.h:8-57), plus
// sm::sum and sm::synchronize for the MI355X side.
//
// Creation happens in HBM: ones/zeros are device fills (smhip_fill), empty
// reserves pooled device memory; a host mirror appears only if host code later
// touches `data` or operator()(i...).
#pragma once

#include <ostream>
#include <vector>

#include "SMArray.h"

namespace sm {

template <typename T, typename... Args>
SMArray<T> empty(Args... args) {
    return SMArray<T>::device_empty({static_cast<std::size_t>(args)...});
}

template <typename T, typename... Args>
SMArray<T> ones(Args... args) {
    return SMArray<T>::device_full({static_cast<std::size_t>(args)...}, T{1});
}

template <typename T, typename... Args>
SMArray<T> zeros(Args... args) {
    return SMArray<T>::device_full({static_cast<std::size_t>(args)...}, T{0});
}

// Element-wise arr ^ val (reference UserFunctions.h:42-48 -> array_scalar_op<T, PowOp<T>>).
// int32/int64: the reference's square-and-multiply with wrapping products
// (crafted_pow.h:54-103) on every element; float: in-register exp2/log2 chain.
template <typename T>
SMArray<T> pow(const SMArray<T> &arr, T val) {
    return arr.template apply_scalar<PowOp<T>>(val);
}

// Sum of all elements in fp64 (BASELINE config 5's reduction; no reference counterpart).
template <typename T>
double sum(const SMArray<T> &arr) {
    return arr.sum();
}

// Block until every queued kernel has finished (operators are asynchronous;
// anything that reads values on the host synchronises by itself).
inline void synchronize() { hip::check(smhip_synchronize()); }

}  // namespace sm

template <typename T>
std::ostream &operator<<(std::ostream &os, const sm::SMArray<T> &arr) {
    return os << arr.toString();
}
