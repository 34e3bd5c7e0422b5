// calculate.h -- the loop entry points of the drop-in host side.
//
// Same three global templates, same signatures, same meaning as the reference
// (include/math/calculate.h:5-8, 101-102, 137-138):
//
//   element_wise_op<T,Op>(a, stride_a, b, stride_b, n, result, shape)
//   handle_contiguous_arrays<T,Op>(a, b, result, n)
//   array_scalar_op<T,Op>(a, value, n, result)
//
// They take HOST pointers, block until `result` is filled, return void -- and
// run the loop on the MI355X: operands are staged into pooled device buffers,
// the Op's gfx950 functor runs (libsmhip: smhip_elementwise / smhip_contiguous /
// smhip_array_scalar), the result is copied back.  That keeps code that calls
// these directly (the README's "add the operator in SMArray.h" recipe) working,
// at PCIe speed.  sm::SMArray's own operators do NOT come through here: they
// keep their data resident in HBM and call the device-pointer forms in
// sm::hip below, which is the hot path.
//
// Errors: the reference has no error channel here (void, UB on bad input);
// a failing device call throws std::runtime_error, the library's single
// exception type (SMUtils.h:76-78).
//
// User-defined Ops: SM_DEVICE_OP(MyOp, "(a + b) * 2") or SM_DEFINE_OP(MyOp, (a + b) * 2) (math/ops.h) gives an Op its
// device form -- the expression is compiled for gfx950 with hipRTC on first use -- and it then runs through the same
// entry points as the built-ins.  An Op WITHOUT a device form (the README recipe as written: apply() plus an x86
// apply_simd<__m256> body, README.md:86-117) is a COMPILE-TIME error whose message says which line to add: nothing in
// this library silently computes on the host.  A build that really wants such an Op evaluated by its host apply() --
// e.g. to bring an existing plugin up unmodified before writing its device string -- opts in with
// -DSM_ALLOW_HOST_USER_OPS: the three templates below then run the reference's scalar loop (calculate.h:52-63,96 /
// :131-133 / :166-168) for USER Ops on HOST pointers.  The five built-in Ops never take that path, with or without it.
#pragma once

#include <cstddef>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <typeinfo>
#include <vector>

#include "helpers.h"
#include "ops.h"
#include "smhip.h"

namespace sm::hip {

inline void check(int rc) {
    if (rc < 0) throw std::runtime_error(std::string("smhip: ") + smhip_last_error());
}

// Makes `device` the calling thread's current GPU for a scope (a no-op when it already is).
class DeviceGuard {
public:
    explicit DeviceGuard(int device) {
        smhip_get_device(&prev_);
        if (device != prev_) { check(smhip_set_device(device)); switched_ = true; }
    }
    DeviceGuard(const DeviceGuard &) = delete;
    DeviceGuard &operator=(const DeviceGuard &) = delete;
    ~DeviceGuard() { if (switched_) smhip_set_device(prev_); }
private:
    int prev_ = 0;
    bool switched_ = false;
};

// RAII pooled device buffer (smhip_alloc / smhip_free).
class DeviceBuffer {
public:
    DeviceBuffer() = default;
    explicit DeviceBuffer(std::size_t bytes) { check(smhip_alloc(&ptr_, bytes ? bytes : 1)); }
    DeviceBuffer(const DeviceBuffer &) = delete;
    DeviceBuffer &operator=(const DeviceBuffer &) = delete;
    DeviceBuffer(DeviceBuffer &&o) noexcept : ptr_(o.ptr_) { o.ptr_ = nullptr; }
    DeviceBuffer &operator=(DeviceBuffer &&o) noexcept {
        if (this != &o) { reset(); ptr_ = o.ptr_; o.ptr_ = nullptr; }
        return *this;
    }
    ~DeviceBuffer() { reset(); }
    void reset() { if (ptr_) { smhip_free(ptr_); ptr_ = nullptr; } }
    void *get() const { return ptr_; }
    template <typename T> T *as() const { return static_cast<T *>(ptr_); }
private:
    void *ptr_ = nullptr;
};

inline std::vector<std::int64_t> to_i64(const std::vector<std::size_t> &v) { return {v.begin(), v.end()}; }

// Elements an operand spans: 1 + sum (extent-1) * stride.
inline std::size_t span_of(const std::vector<std::size_t> &shape, const std::vector<std::size_t> &stride) {
    std::size_t last = 0;
    for (std::size_t i = 0; i < shape.size(); ++i) {
        if (shape[i] == 0) return 0;
        last += (shape[i] - 1) * stride[i];
    }
    return last + 1;
}

// ---- device-pointer forms: what SMArray's operators call (the hot path) ----
template <typename T, typename Op>
void element_wise_op_device(const T *a, const std::vector<std::size_t> &stride_a, const T *b,
                            const std::vector<std::size_t> &stride_b, T *result, const std::vector<std::size_t> &shape) {
    static_assert(on_device_v<T, Op>, "no gfx950 functor for this Op / element type");
    const auto sa = to_i64(stride_a), sb = to_i64(stride_b), sh = to_i64(shape);
    check(smhip_elementwise(device_op<Op>::id(), dtype_of<T>::id, a, sa.data(), b, sb.data(), sh.data(),
                            static_cast<int>(sh.size()), result));
}

template <typename T, typename Op>
void array_scalar_op_device(const T *a, T value, std::size_t n, T *result) {
    static_assert(on_device_v<T, Op>, "no gfx950 functor for this Op / element type");
    check(smhip_array_scalar(device_op<Op>::id(), dtype_of<T>::id, a, &value, n, result));
}

template <typename Op> struct is_builtin_op : std::false_type {};
template <typename T> struct is_builtin_op<AddOp<T>> : std::true_type {};
template <typename T> struct is_builtin_op<SubtractOp<T>> : std::true_type {};
template <typename T> struct is_builtin_op<MultiplyOp<T>> : std::true_type {};
template <typename T> struct is_builtin_op<DivideOp<T>> : std::true_type {};
template <typename T> struct is_builtin_op<PowOp<T>> : std::true_type {};

// What happens to an (element type, Op) pair that has no gfx950 kernel.
template <typename T, typename Op>
constexpr void no_device_form() {
    static_assert(dtype_of<T>::id >= 0 || !is_builtin_op<Op>::value,
                  "simpleMath/MI355X: this element type has no gfx950 kernels (float, double and signed 32/64-bit integers do; "
                  "std::complex has a dot product only) and the library has no CPU arithmetic path");
#ifndef SM_ALLOW_HOST_USER_OPS
    static_assert(dependent_false<Op>::value,
                  "simpleMath/MI355X: this Op has no device form. After the Op add  SM_DEVICE_OP(MyOp, \"<its arithmetic as a HIP "
                  "expression in a and b>\")  -- or define it in one line with  SM_DEFINE_OP(MyOp, (a + b) * 2)  (math/ops.h). "
                  "To run an unmodified plugin through its host apply() instead, compile with -DSM_ALLOW_HOST_USER_OPS.");
#endif
}

}  // namespace sm::hip

template <typename T, typename Operation>
void handle_contiguous_arrays(const T *a, const T *b, T *result, std::size_t n);

template <typename T, typename Operation>
void element_wise_op(const T *a, const std::vector<std::size_t> &stride_a, const T *b,
                     const std::vector<std::size_t> &stride_b, std::size_t n, T *result,
                     const std::vector<std::size_t> &shape) {
    using namespace sm::hip;
    if constexpr (on_device_v<T, Operation>) {
        if (shape.size() > MAX_NDIM) throw std::runtime_error("element_wise_op: rank exceeds MAX_NDIM");
        if (n == 0) return;
        const std::size_t na = span_of(shape, stride_a), nb = span_of(shape, stride_b);
        DeviceBuffer da(na * sizeof(T)), db(nb * sizeof(T)), dr(n * sizeof(T));
        check(smhip_upload(da.get(), a, na * sizeof(T)));
        check(smhip_upload(db.get(), b, nb * sizeof(T)));
        element_wise_op_device<T, Operation>(da.template as<T>(), stride_a, db.template as<T>(), stride_b,
                                             dr.template as<T>(), shape);
        check(smhip_download(result, dr.get(), n * sizeof(T)));
    } else {
        no_device_form<T, Operation>();
        // opt-in only (SM_ALLOW_HOST_USER_OPS), user Ops only: the reference's scalar statement, result[linear] =
        // Op::apply(a[offA], b[offB]) with the row-major unravel of `linear` (calculate.h:52-63, :96), as an odometer
        std::vector<std::size_t> idx(shape.size(), 0);
        std::size_t offA = 0, offB = 0;
        for (std::size_t linear = 0; linear < n; ++linear) {
            result[linear] = Operation::apply(a[offA], b[offB]);
            for (std::size_t k = shape.size(); k-- > 0;) {
                offA += stride_a[k];
                offB += stride_b[k];
                if (++idx[k] < shape[k]) break;
                offA -= stride_a[k] * shape[k];
                offB -= stride_b[k] * shape[k];
                idx[k] = 0;
            }
        }
    }
}

template <typename T, typename Operation>
void handle_contiguous_arrays(const T *a, const T *b, T *result, std::size_t n) {
    using namespace sm::hip;
    if constexpr (on_device_v<T, Operation>) {
        if (n == 0) return;
        DeviceBuffer da(n * sizeof(T)), db(n * sizeof(T)), dr(n * sizeof(T));
        check(smhip_upload(da.get(), a, n * sizeof(T)));
        check(smhip_upload(db.get(), b, n * sizeof(T)));
        check(smhip_contiguous(device_op<Operation>::id(), dtype_of<T>::id, da.get(), db.get(), dr.get(), n));
        check(smhip_download(result, dr.get(), n * sizeof(T)));
    } else {
        no_device_form<T, Operation>();
        for (std::size_t i = 0; i < n; ++i) result[i] = Operation::apply(a[i], b[i]);  // opt-in only, user Ops only
    }
}

template <typename T, typename Operation>
void array_scalar_op(const T *a, T value, const std::size_t n, T *result) {
    using namespace sm::hip;
    if constexpr (on_device_v<T, Operation>) {
        if (n == 0) return;
        DeviceBuffer da(n * sizeof(T)), dr(n * sizeof(T));
        check(smhip_upload(da.get(), a, n * sizeof(T)));
        array_scalar_op_device<T, Operation>(da.template as<T>(), value, n, dr.template as<T>());
        check(smhip_download(result, dr.get(), n * sizeof(T)));
    } else {
        no_device_form<T, Operation>();
        for (std::size_t i = 0; i < n; ++i) result[i] = Operation::apply(a[i], value);  // opt-in only, user Ops only
    }
}
