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
// SM_DEVICE_OP / SM_DEFINE_OP (below).  Without one it does not compile (a static_assert says what to add), unless the
// build opts in to evaluating USER Ops with their host apply() by defining SM_ALLOW_HOST_USER_OPS (math/calculate.h).
#pragma once

#include <cmath>
#include <complex>
#include <cstdint>
#include <cstdlib>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <typeinfo>

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

// Element types SMArray<T>::operator% has a kernel for: the four above plus the other integer types the reference's GENERIC
// dot_product<T> (product.h:8-20) serves -- 8- / 16-bit and unsigned 32- / 64-bit integers (bool excepted: its `sum += a * b`
// is a logical OR, not arithmetic).
template <typename T> struct dot_dtype_of {
    static constexpr int id = dtype_of<T>::id >= 0 ? dtype_of<T>::id
                              : (!std::is_integral_v<T> || std::is_same_v<T, bool>) ? -1
                              : sizeof(T) == 1 ? static_cast<int>(std::is_signed_v<T> ? SMHIP_I8 : SMHIP_U8)
                              : sizeof(T) == 2 ? static_cast<int>(std::is_signed_v<T> ? SMHIP_I16 : SMHIP_U16)
                              : sizeof(T) == 4 ? static_cast<int>(SMHIP_U32)
                              : sizeof(T) == 8 ? static_cast<int>(SMHIP_U64)
                                               : -1;
};

// How an element type MOVES on the device -- dense copies of views, repeat(), assignment into views: no arithmetic --
// as `lanes` consecutive elements of a type the copy kernels know: the kernel types as themselves, unsigned 32- / 64-bit
// integers as their signed twins, std::complex<float | double> as pairs of floats / doubles (an extra innermost axis of
// extent 2).  id < 0: the type has no device copy kernels (8- and 16-bit integers: their views are gathered on the host).
template <typename T> struct transport_of {
    static constexpr int id = dtype_of<T>::id >= 0 ? dtype_of<T>::id
                              : (std::is_integral_v<T> && !std::is_same_v<T, bool> && sizeof(T) == 4) ? static_cast<int>(SMHIP_I32)
                              : (std::is_integral_v<T> && sizeof(T) == 8) ? static_cast<int>(SMHIP_I64)
                                                                          : -1;
    static constexpr int lanes = 1;
};
template <> struct transport_of<std::complex<double>> { static constexpr int id = SMHIP_F64, lanes = 2; };
template <> struct transport_of<std::complex<float>> { static constexpr int id = SMHIP_F32, lanes = 2; };

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

// SM_DEVICE_OP's string is a second statement of what Op::apply() says.  On an Op's first use per element type the two are
// compared: 4096 sample pairs (both signs, magnitudes in [1/4, 4], never zero) go through the compiled kernel and through
// the host Op::apply(); integers must agree exactly, floats to a relative 1e-5 / 1e-12 (the point is to catch a different
// FORMULA, not a contracted multiply-add).  A mismatch throws std::runtime_error naming the Op.  SMHIP_VERIFY_USER_OPS=0
// in the environment skips the check.
template <typename T, typename Op>
inline void verify_user_op(int id, const char *hip_expression) {
    if constexpr (dtype_of<T>::id >= 0) {
        if (const char *env = std::getenv("SMHIP_VERIFY_USER_OPS"))
            if (env[0] == '0') return;
        constexpr std::size_t kN = 4096;
        T a[kN], b[kN], want[kN], got[kN];
        std::uint64_t state = 0x243F6A8885A308D3ull;
        auto next = [&state] {  // splitmix64
            std::uint64_t z = (state += 0x9E3779B97F4A7C15ull);
            z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
            z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
            return z ^ (z >> 31);
        };
        auto sample = [&next]() -> T {
            const std::uint64_t r = next();
            if constexpr (std::is_integral_v<T>) {
                const T m = static_cast<T>(1 + (r >> 8) % 64);
                return (r & 1) ? m : static_cast<T>(-m);
            } else {
                const double m = 0.25 * std::exp2(4.0 * static_cast<double>(r >> 11) / 9007199254740992.0);  // [1/4, 4)
                return static_cast<T>((r & 1) ? m : -m);
            }
        };
        for (std::size_t i = 0; i < kN; ++i) { a[i] = sample(); b[i] = sample(); want[i] = Op::apply(a[i], b[i]); }
        void *da = nullptr, *db = nullptr, *dr = nullptr;
        auto ok = [](int rc) { if (rc < 0) throw std::runtime_error(std::string("smhip: ") + smhip_last_error()); };
        ok(smhip_alloc(&da, sizeof a)); ok(smhip_alloc(&db, sizeof b)); ok(smhip_alloc(&dr, sizeof got));
        int rc = smhip_upload(da, a, sizeof a);
        if (rc >= 0) rc = smhip_upload(db, b, sizeof b);
        if (rc >= 0) rc = smhip_contiguous(id, dtype_of<T>::id, da, db, dr, kN);
        if (rc >= 0) rc = smhip_download(got, dr, sizeof got);
        smhip_free(da); smhip_free(db); smhip_free(dr);
        ok(rc);
        for (std::size_t i = 0; i < kN; ++i) {
            bool same;
            if constexpr (std::is_integral_v<T>) {
                same = got[i] == want[i];
            } else {
                const double g = static_cast<double>(got[i]), w = static_cast<double>(want[i]);
                const double tol = (sizeof(T) == 4 ? 1e-5 : 1e-12) * std::fmax(std::fabs(w), 1e-30);
                same = (std::isnan(g) && std::isnan(w)) || g == w || std::fabs(g - w) <= tol;
            }
            if (!same)
                throw std::runtime_error(std::string("simpleMath/MI355X: user Op '") + typeid(Op).name() + "': its device form \"" +
                                         hip_expression + "\" disagrees with Op::apply() -- apply(" + std::to_string(a[i]) + ", " +
                                         std::to_string(b[i]) + ") = " + std::to_string(want[i]) + " on the host, " +
                                         std::to_string(got[i]) + " from the kernel. Fix the SM_DEVICE_OP string (or define the Op once "
                                         "with SM_DEFINE_OP); SMHIP_VERIFY_USER_OPS=0 disables this check.");
        }
    }
}

template <typename T, typename Op>
inline int register_and_verify(const char *hip_expression) {
    const int id = register_expr(hip_expression);
    verify_user_op<T, Op>(id, hip_expression);
    return id;
}

}  // namespace sm::hip

// Give a user-defined Op template its device form (the gfx950 counterpart of writing an
// apply_simd<__m256> specialisation, README.md:105-117): the same arithmetic as apply(), as a HIP
// expression in `a` and `b` (the element type is `T`).  At global scope, after the Op:
//     template <typename T> struct MyOp { static T apply(const T& a, const T& b) { return (a + b) * 2; } ... };
//     SM_DEVICE_OP(MyOp, "(a + b) * 2")
// The expression is compiled for gfx950 by hipRTC on first use and cached; that first use also checks it against
// apply() on sample values (sm::hip::verify_user_op above).
#define SM_DEVICE_OP(OpTemplate, hip_expression)                                             \
    namespace sm::hip {                                                                      \
    template <typename T> struct device_op<OpTemplate<T>> {                                  \
        static constexpr bool available = true;                                              \
        static int id() {                                                                    \
            static const int v = register_and_verify<T, OpTemplate<T>>(hip_expression);      \
            return v;                                                                        \
        }                                                                                    \
    };                                                                                       \
    }

// One source of truth: the whole Op -- the struct with apply() / apply_simd (the reference's contract,
// include/math/add.h:5-14) AND its device form -- from one expression in `a`, `b` and `T`:
//     SM_DEFINE_OP(MyOp, (a + b) * 2)
// apply() is the expression compiled by the host compiler, the gfx950 kernel is the same tokens compiled by hipRTC.
#define SM_DEFINE_OP(OpName, ...)                                                            \
    template <typename T> struct OpName {                                                    \
        static T apply(const T &a, const T &b) { return static_cast<T>(__VA_ARGS__); }       \
        template <typename SIMD_T> static SIMD_T apply_simd(const SIMD_T &a, const SIMD_T &b); \
    };                                                                                       \
    SM_DEVICE_OP(OpName, #__VA_ARGS__)
