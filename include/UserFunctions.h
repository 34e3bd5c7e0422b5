// UserFunctions.h -- free functions of the drop-in surface: sm::empty / ones /
// zeros, sm::pow, operator<< (reference include/UserFunctions.h:8-57), plus
// sm::sum and sm::synchronize for the MI355X side.
//
// Creation happens in HBM: ones/zeros are device fills (smhip_fill), empty
// reserves pooled device memory; a host mirror appears only if host code later
// touches `data` or operator()(i...).
#pragma once

#include <initializer_list>
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
// ... of a temporary: sm::pow(a - b, 2.0f) continues the expression's operator chain (SMArray::pow_of) instead of reading
// its result back from HBM -- the squared difference is one kernel.
template <typename T>
SMArray<T> pow(SMArray<T> &&arr, T val) {
    return SMArray<T>::pow_of(arr, true, val);
}

// Fusion hook: (a Op1 b) Op2 c in ONE pass over HBM (the reference makes two passes and a
// temporary).  Dense, equal-shaped operands take the two-Op kernel; operands that broadcast against each other (a row, a
// column, a scalar-like array, the reference tests' (1,224,1,3)) take the chain kernel (smhip_chain) -- what the operators
// themselves do for `(a + b) * c` written as one expression; user-defined Ops and pow are evaluated as the two operator
// calls.  Same values every way.
//   auto r = sm::fused<AddOp<float>, MultiplyOp<float>>(a, b, c);   // (a + b) * c
template <typename Op1, typename Op2, typename T>
SMArray<T> fused(const SMArray<T> &a, const SMArray<T> &b, const SMArray<T> &c) {
    if constexpr (hip::on_device_v<T, Op1> && hip::on_device_v<T, Op2>) {
        const int o1 = hip::device_op<Op1>::id(), o2 = hip::device_op<Op2>::id();
        if (o1 <= SMHIP_OP_DIV && o2 <= SMHIP_OP_DIV && a.shape() == b.shape() && a.shape() == c.shape() && a.is_dense() &&
            b.is_dense() && c.is_dense()) {
            SMArray<T>::common_device(a, c);
            hip::DeviceGuard on(SMArray<T>::common_device(a, b));
            SMArray<T> out = SMArray<T>::device_empty(std::vector<std::size_t>(a.shape()));
            hip::check(smhip_fused_contiguous(o1, o2, hip::dtype_of<T>::id, a.device_data(), b.device_data(), c.device_data(),
                                              nullptr, out.device_data_mut(), a.totalSize));
            return out;
        }
        if (o1 <= SMHIP_OP_DIV && o2 <= SMHIP_OP_DIV) return SMArray<T>::template chain_of<Op1, Op2>(a, b, &c, T{});
    }
    return a.template apply<Op1>(b).template apply<Op2>(c);
}
template <typename Op1, typename Op2, typename T>
SMArray<T> fused(const SMArray<T> &a, const SMArray<T> &b, T c) {
    if constexpr (hip::on_device_v<T, Op1> && hip::on_device_v<T, Op2>) {
        const int o1 = hip::device_op<Op1>::id(), o2 = hip::device_op<Op2>::id();
        if (o1 <= SMHIP_OP_DIV && o2 <= SMHIP_OP_DIV && a.shape() == b.shape() && a.is_dense() && b.is_dense()) {
            hip::DeviceGuard on(SMArray<T>::common_device(a, b));
            SMArray<T> out = SMArray<T>::device_empty(std::vector<std::size_t>(a.shape()));
            hip::check(smhip_fused_contiguous(o1, o2, hip::dtype_of<T>::id, a.device_data(), b.device_data(), nullptr, &c,
                                              out.device_data_mut(), a.totalSize));
            return out;
        }
        if (o1 <= SMHIP_OP_DIV && o2 <= SMHIP_OP_DIV) return SMArray<T>::template chain_of<Op1, Op2>(a, b, nullptr, c);
    }
    return a.template apply<Op1>(b).template apply_scalar<Op2>(c);
}

// A whole expression in ONE pass over HBM: sm::expr("(a0 + a1) * a2 - 3 * a3", a, b, c, d).  The operands appear as
// a0 .. a7 in a HIP expression of the element type and broadcast against each other like the operators' (NumPy rule): a
// row, a column, a per-channel value or a one-element array is read through the caches inside the pass, a transposed or
// stepped view is copied dense first.  Each operation rounds as the separate operators do, so the values equal the operator
// chain's; the traffic is (k + 1) * sizeof(T) bytes per element for k full-size operands instead of 3 * sizeof(T) per
// operator.  Compiled by hipRTC on first use, cached.
// Run-time scalars appear as s0 .. s3 and are passed at launch (changing them does not recompile):
//     sm::expr("a0 * s0 + a1", {alpha}, x, y)        // axpy in one pass
template <typename T, typename... Rest>
SMArray<T> expr(const char *expression, std::initializer_list<T> scalars, const SMArray<T> &first, const Rest &...rest) {
    static_assert(hip::dtype_of<T>::id >= 0, "sm::expr: element type has no kernels");
    static_assert(sizeof...(Rest) <= 7, "sm::expr: at most 8 operands");
    static_assert((std::is_same_v<Rest, SMArray<T>> && ...), "sm::expr: operands must be SMArray<T> of one element type");
    if (scalars.size() > 4) throw std::runtime_error("sm::expr: at most 4 scalars");
    const SMArray<T> *arrays[] = {&first, &rest...};
    constexpr int n = 1 + static_cast<int>(sizeof...(Rest));
    const void *ptrs[8] = {};
    for (int k = 1; k < n; ++k) SMArray<T>::common_device(first, *arrays[k]);  // all operands on one GPU, or it throws
    hip::DeviceGuard on(first.device());
    bool flat = true;
    for (int k = 0; k < n; ++k) flat = flat && arrays[k]->shape() == first.shape() && arrays[k]->is_dense();
    if (flat) {
        for (int k = 0; k < n; ++k) ptrs[k] = arrays[k]->device_data();
        SMArray<T> out = SMArray<T>::device_empty(std::vector<std::size_t>(first.shape()));
        hip::check(smhip_fused_expr(expression, hip::dtype_of<T>::id, ptrs, n, scalars.size() ? scalars.begin() : nullptr,
                                    static_cast<int>(scalars.size()), out.device_data_mut(), first.totalSize));
        return out;
    }
    // the common shape (sm::broadcast throws the reference's "Cannot broadcast shapes" on a mismatch), then every operand's
    // strides against it
    std::vector<std::size_t> shape = first.shape();
    const std::vector<std::size_t> none;
    for (int k = 1; k < n; ++k) shape = sm::broadcast(shape, std::vector<std::size_t>(shape.size(), 0), arrays[k]->shape(), arrays[k]->strides()).resultShape;
    if (shape.size() > MAX_NDIM) throw std::runtime_error("rank exceeds MAX_NDIM");
    if (shape.empty()) throw std::runtime_error("sm::expr: 0-d operands");
    const std::size_t nd = shape.size();
    std::vector<std::int64_t> strides(static_cast<std::size_t>(n) * nd, 0), sh(shape.begin(), shape.end());
    for (int k = 0; k < n; ++k) {
        const auto &ls = arrays[k]->shape();
        const auto &lt = arrays[k]->strides();
        const std::size_t shift = nd - ls.size();
        for (std::size_t i = 0; i < ls.size(); ++i) strides[k * nd + shift + i] = (ls[i] == 1 && shape[shift + i] != 1) ? 0 : static_cast<std::int64_t>(lt[i]);
        ptrs[k] = arrays[k]->device_data();
    }
    SMArray<T> out = SMArray<T>::device_empty(std::move(shape));
    hip::check(smhip_fused_expr_bcast(expression, hip::dtype_of<T>::id, ptrs, strides.data(), n, scalars.size() ? scalars.begin() : nullptr,
                                      static_cast<int>(scalars.size()), sh.data(), static_cast<int>(nd), out.device_data_mut()));
    return out;
}
template <typename T, typename... Rest>
SMArray<T> expr(const char *expression, const SMArray<T> &first, const Rest &...rest) {
    return expr<T>(expression, std::initializer_list<T>{}, first, rest...);
}

// The sum of an expression's results in the same single pass, nothing stored: sm::expr_sum("(a0 - a1) * (a0 - a1)", x, y)
// is a squared distance at 8 bytes per element and no temporary (fp64 accumulation, deterministic).
template <typename T, typename... Rest>
double expr_sum(const char *expression, std::initializer_list<T> scalars, const SMArray<T> &first, const Rest &...rest) {
    static_assert(hip::dtype_of<T>::id >= 0, "sm::expr_sum: element type has no kernels");
    static_assert(sizeof...(Rest) <= 7, "sm::expr_sum: at most 8 operands");
    static_assert((std::is_same_v<Rest, SMArray<T>> && ...), "sm::expr_sum: operands must be SMArray<T> of one element type");
    if (scalars.size() > 4) throw std::runtime_error("sm::expr_sum: at most 4 scalars");
    const SMArray<T> *arrays[] = {&first, &rest...};
    constexpr int n = 1 + static_cast<int>(sizeof...(Rest));
    std::vector<SMArray<T>> dense;
    dense.reserve(n);
    const void *ptrs[8] = {};
    for (int k = 1; k < n; ++k) SMArray<T>::common_device(first, *arrays[k]);
    hip::DeviceGuard on(first.device());
    for (int k = 0; k < n; ++k) {
        if (arrays[k]->shape() != first.shape()) throw std::runtime_error("sm::expr_sum: operands must have the same shape");
        if (arrays[k]->is_dense()) {
            ptrs[k] = arrays[k]->device_data();
        } else {
            dense.push_back(arrays[k]->contiguous());
            ptrs[k] = dense.back().device_data();
        }
    }
    hip::DeviceBuffer result(sizeof(double));
    hip::check(smhip_fused_expr_sum_async(expression, hip::dtype_of<T>::id, ptrs, n, scalars.size() ? scalars.begin() : nullptr,
                                          static_cast<int>(scalars.size()), nullptr, first.totalSize, result.template as<double>()));
    double total = 0;
    hip::check(smhip_download(&total, result.get(), sizeof total));
    return total;
}
template <typename T, typename... Rest>
double expr_sum(const char *expression, const SMArray<T> &first, const Rest &...rest) {
    return expr_sum<T>(expression, std::initializer_list<T>{}, first, rest...);
}

// Sum of all elements in fp64 (BASELINE config 5's reduction; no reference counterpart).
template <typename T>
double sum(const SMArray<T> &arr) {
    return arr.sum();
}
template <typename T>
double sum(SMArray<T> &&arr) {  // sm::sum(sm::pow(a - b, 2.0f)): the expression's temporary, summed in its chain's own pass
    return std::move(arr).sum();
}

// Block until every queued kernel has finished (operators are asynchronous;
// anything that reads values on the host synchronises by itself).
inline void synchronize() { hip::check(smhip_synchronize()); }

}  // namespace sm

template <typename T>
std::ostream &operator<<(std::ostream &os, const sm::SMArray<T> &arr) {
    return os << arr.toString();
}
