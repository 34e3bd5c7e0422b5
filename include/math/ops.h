// ops.h -- the Op policy structs (the plugin contract) and their device mapping.
//
// Contract kept from the reference (include/math/add.h:5-14 and siblings,
// README "Extending with Custom Operations"): a stateless struct template in
// the global namespace with
//     static T apply(const T& a, const T& b);                  // scalar meaning
//     template<class R> static R apply_simd(const R&, const R&);  // register form, declared only
// `apply` stays the definition of what the Op means.  What changes is where the
// loop runs: sm::hip::device_op<Op>::id names the functor libsmhip's gfx950
// kernels instantiate for it (simplemath_amd/csrc/ops.hip.h).  The five
// built-in Ops always run on the device; a user Op gets its device form from
// SM_DEVICE_OP (below) and is refused loudly without one -- nothing computes on the host.
#pragma once

#include <cmath>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <type_traits>

#include "helpers.h"
#include "smhip.h"

template <typename T>
struct AddOp {
    static T apply(const T &a, const T &b) { return a + b; }
    template <typename SIMD_T> static SIMD_T apply_simd(const SIMD_T &a, const SIMD_T &b);
};

template <typename T>
struct SubtractOp {
    static T apply(const T &a, const T &b) { return a - b; }
    template <typename SIMD_T> static SIMD_T apply_simd(const SIMD_T &a, const SIMD_T &b);
};

template <typename T>
struct MultiplyOp {
    static T apply(const T &a, const T &b) { return a * b; }
    template <typename SIMD_T> static SIMD_T apply_simd(const SIMD_T &a, const SIMD_T &b);
};

template <typename T>
struct DivideOp {
    static T apply(const T &a, const T &b) { return a / b; }
    template <typename SIMD_T> static SIMD_T apply_simd(const SIMD_T &a, const SIMD_T &b);
};

template <typename T>
struct PowOp {
    static T apply(const T &base, const T &exponent) { return static_cast<T>(std::pow(base, exponent)); }
    template <typename SIMD_T> static SIMD_T apply_simd(const SIMD_T &base, const SIMD_T &exponent);
};

namespace sm::hip {

// Element types the kernels are built for (any signed 32/64-bit integer type maps by size).
template <typename T> struct dtype_of {
    static constexpr int id = std::is_same_v<T, float>    ? static_cast<int>(SMHIP_F32)
                              : std::is_same_v<T, double> ? static_cast<int>(SMHIP_F64)
                              : (std::is_integral_v<T> && std::is_signed_v<T> && sizeof(T) == 4) ? static_cast<int>(SMHIP_I32)
                              : (std::is_integral_v<T> && std::is_signed_v<T> && sizeof(T) == 8) ? static_cast<int>(SMHIP_I64)
                                                                                                 : -1;
};

// Op -> device functor.  `available` says whether the Op can run on the GPU, `id()` names the
// functor: a constant for the five built-ins (AOT kernels in libsmhip), a registered id for a user
// Op whose arithmetic was handed over as a HIP expression (SM_DEVICE_OP below).
template <typename Op> struct device_op {
    static constexpr bool available = false;
    static int id() { return -1; }
};
template <typename T> struct device_op<AddOp<T>> { static constexpr bool available = true; static int id() { return SMHIP_OP_ADD; } };
template <typename T> struct device_op<SubtractOp<T>> { static constexpr bool available = true; static int id() { return SMHIP_OP_SUB; } };
template <typename T> struct device_op<MultiplyOp<T>> { static constexpr bool available = true; static int id() { return SMHIP_OP_MUL; } };
template <typename T> struct device_op<DivideOp<T>> { static constexpr bool available = true; static int id() { return SMHIP_OP_DIV; } };
template <typename T> struct device_op<PowOp<T>> { static constexpr bool available = true; static int id() { return SMHIP_OP_POW; } };

inline int register_expr(const char *hip_expression) {
    int id = -1;
    if (smhip_register_op(hip_expression, &id) < 0) throw std::runtime_error(std::string("smhip: ") + smhip_last_error());
    return id;
}

template <typename T, typename Op>
inline constexpr bool on_device_v = (dtype_of<T>::id >= 0) && device_op<Op>::available;

}  // namespace sm::hip

// Give a user-defined Op template its device form (the gfx950 counterpart of writing an
// apply_simd<__m256> specialisation, README.md:105-117): the same arithmetic as apply(), as a HIP
// expression in `a` and `b`.  At global scope, after the Op:
//     template <typename T> struct MyOp { static T apply(const T& a, const T& b) { return (a + b) * 2; } ... };
//     SM_DEVICE_OP(MyOp, "(a + b) * 2")
// The expression is compiled for gfx950 by hipRTC on first use and cached.
#define SM_DEVICE_OP(OpTemplate, hip_expression)                                   \
    namespace sm::hip {                                                            \
    template <typename T> struct device_op<OpTemplate<T>> {                        \
        static constexpr bool available = true;                                    \
        static int id() {                                                          \
            static const int v = register_expr(hip_expression);                    \
            return v;                                                              \
        }                                                                          \
    };                                                                             \
    }
