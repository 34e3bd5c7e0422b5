// contiguous.hip -- the dense streaming kernels.
//
// Replaces handle_contiguous_arrays<T,Op> (reference include/math/calculate.h:
// 101-134, an 8-wide AVX2 loop on ONE host thread) and array_scalar_op<T,Op>
// (calculate.h:137-169) with HBM-streaming gfx950 kernels.
//
// Roofline: HBM-bound.  Algorithmic traffic 3 * sizeof(T) B/elem for a op b
// (2 reads + 1 write; 12 B/elem for f32), 2 * sizeof(T) for a op scalar.
// Shape of the launch (profiles/r01_sweep_stream_add.txt, N = 2^28 f32 add):
//   - one 16-byte vector per lane (float4 / double2 / int4 / long2): a wave
//     moves 1 KiB per memory instruction, fully coalesced;
//   - NO grid-stride loop: one vector per thread, >= 65k workgroups, so the
//     hardware dispatcher (not a software loop) keeps all 256 CUs / 8 XCDs
//     fed and every wave retires after 2 loads + 1 store.  Persistent
//     grid-stride variants with 2-8 vectors in flight per lane measured
//     5-12 % slower on this 2R+1W stream;
//   - non-temporal loads and stores (each byte is touched once): +8 %.
// Blocks are dealt round-robin over the 8 XCDs, consecutive blocks touch
// consecutive 16 KiB spans, so every XCD streams from all HBM stacks at once;
// there is no reuse for an XCD-affine mapping to exploit.
#include <type_traits>

#include "internal.h"
#include "ops.hip.h"

namespace smhip {
namespace {

using namespace dev;

constexpr int kBlockBig = 1024;   // n_vec >= kBigThreshold
constexpr int kBlockSmall = 256;
constexpr size_t kBigThreshold = (size_t)1 << 20;

// out[i] = a[i] op b[i], vector body + <= width-1 scalar tail elements done
// by the first thread past the body.
template <typename T, typename Op, int BLOCK>
__global__ __launch_bounds__(BLOCK) void contiguous_vec_kernel(const T *__restrict__ a, const T *__restrict__ b,
                                                               T *__restrict__ out, size_t n_vec, int tail) {
    typedef typename VecTraits<T>::vec_t V;
    constexpr int W = VecTraits<T>::width;
    OpCtx<Op> ctx;
    ctx.init();
    const size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i < n_vec) {
        const V va = load_stream(reinterpret_cast<const V *>(a) + i);
        const V vb = load_stream(reinterpret_cast<const V *>(b) + i);
        store_stream(reinterpret_cast<V *>(out) + i, apply_vec<Op, T>(ctx, va, vb));
    } else if (i == n_vec) {
        for (int k = 0; k < tail; ++k) out[n_vec * W + k] = Op::apply(a[n_vec * W + k], b[n_vec * W + k]);
    }
}

// out[i] = a[i] op s  (SWAPPED: s op a[i]); s is a kernel argument, i.e. it
// lives in SGPRs -- the `set1` of calculate.h:141-146 costs nothing here.
template <typename T, typename Op, int BLOCK, bool SWAPPED>
__global__ __launch_bounds__(BLOCK) void scalar_vec_kernel(const T *__restrict__ a, T s, T *__restrict__ out,
                                                           size_t n_vec, int tail) {
    typedef typename VecTraits<T>::vec_t V;
    constexpr int W = VecTraits<T>::width;
    OpCtx<Op> ctx;
    ctx.init();
    const size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i < n_vec) {
        const V va = load_stream(reinterpret_cast<const V *>(a) + i);
        store_stream(reinterpret_cast<V *>(out) + i, apply_vec_scalar<Op, T, SWAPPED>(ctx, va, s));
    } else if (i == n_vec) {
        for (int k = 0; k < tail; ++k) {
            const T x = a[n_vec * W + k];
            out[n_vec * W + k] = SWAPPED ? Op::apply(s, x) : Op::apply(x, s);
        }
    }
}

// Same, with the scalar fetched from device memory (a fully broadcast array
// operand, e.g. `A + one_element_array`): a wave-uniform load.
template <typename T, typename Op, int BLOCK, bool SWAPPED>
__global__ __launch_bounds__(BLOCK) void devscalar_vec_kernel(const T *__restrict__ a, const T *__restrict__ sp,
                                                              T *__restrict__ out, size_t n_vec, int tail) {
    typedef typename VecTraits<T>::vec_t V;
    constexpr int W = VecTraits<T>::width;
    OpCtx<Op> ctx;
    ctx.init();
    const T s = *sp;
    const size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i < n_vec) {
        const V va = load_stream(reinterpret_cast<const V *>(a) + i);
        store_stream(reinterpret_cast<V *>(out) + i, apply_vec_scalar<Op, T, SWAPPED>(ctx, va, s));
    } else if (i == n_vec) {
        for (int k = 0; k < tail; ++k) {
            const T x = a[n_vec * W + k];
            out[n_vec * W + k] = SWAPPED ? Op::apply(s, x) : Op::apply(x, s);
        }
    }
}

// Arithmetic-heavy Ops (pow): the streaming kernels above give each wave one load, a
// long stretch of VALU work, one store -- memory and VALU time add up instead of
// overlapping.  This form is persistent (grid = CUs x 32 workgroups of 512) and
// software-pipelined one vector deep: the next vector's loads are in flight while
// the current one is evaluated, and the per-workgroup LDS table of PowOp<float> is
// staged once per workgroup instead of once per 16 KiB.  profiles/r01_sweep_pow.txt:
// 91 us vs 104-147 us for config 4, a plain copy of the same bytes being 85 us.
// KIND 0: a[i] op b[i];  1: a[i] op s;  2: s op a[i].
constexpr int kHeavyBlock = 512;
constexpr int kHeavyGridPerCU = 32;

template <typename T, typename Op, int KIND>
__global__ __launch_bounds__(kHeavyBlock) void heavy_vec_kernel(const T *__restrict__ a, const T *__restrict__ b, T s,
                                                                T *__restrict__ out, size_t n_vec, int tail) {
    typedef typename VecTraits<T>::vec_t V;
    constexpr int W = VecTraits<T>::width;
    OpCtx<Op> ctx;
    ctx.init();
    const V *av = reinterpret_cast<const V *>(a), *bv = reinterpret_cast<const V *>(b);
    V *ov = reinterpret_cast<V *>(out);
    const size_t stride = (size_t)gridDim.x * kHeavyBlock;
    size_t i = (size_t)blockIdx.x * kHeavyBlock + threadIdx.x;
    if (i == 0) {  // the n % W scalar tail rides with the first lane
        for (int k = 0; k < tail; ++k) {
            const T x = a[n_vec * W + k];
            const T y = KIND == 0 ? b[n_vec * W + k] : s;
            out[n_vec * W + k] = KIND == 2 ? Op::apply(y, x) : Op::apply(x, y);
        }
    }
    if (i >= n_vec) return;
    V ca = load_stream(av + i), cb;
    if constexpr (KIND == 0) cb = load_stream(bv + i);
    for (;;) {
        const size_t nx = i + stride;
        const bool more = nx < n_vec;
        V na, nb;
        if (more) {
            na = load_stream(av + nx);
            if constexpr (KIND == 0) nb = load_stream(bv + nx);
        }
        V r;
        if constexpr (KIND == 0) r = apply_vec<Op, T>(ctx, ca, cb);
        else r = apply_vec_scalar<Op, T, KIND == 2>(ctx, ca, s);
        store_stream(ov + i, r);
        if (!more) break;
        ca = na;
        if constexpr (KIND == 0) cb = nb;
        i = nx;
    }
}

template <typename Op> struct IsHeavy : std::false_type {};
// float / double pow only: integer pow is a short square-and-multiply loop, and the plain launch beats the pipelined one on
// it for every exponent distribution tried (tools/ipow_exp.py: 80 % vs 64 % of peak for exponents < 32, 42 % vs 40 % for
// 20-bit exponents)
template <typename T> struct IsHeavy<PowOp<T>> : std::integral_constant<bool, std::is_floating_point<T>::value> {};

inline unsigned heavy_grid(size_t n_vec) {
    const size_t want = (n_vec + kHeavyBlock - 1) / kHeavyBlock;
    const size_t cap = (size_t)compute_units() * kHeavyGridPerCU;
    return (unsigned)(want < cap ? (want ? want : 1) : cap);
}

inline int grid_for(size_t threads, int block, unsigned *grid) {
    const size_t g = (threads + block - 1) / block;
    if (g > 0x7fffffffu) return fail(SMHIP_ERR_UNSUPPORTED, "array too large for one launch (%zu workgroups)", g);
    *grid = (unsigned)g;
    return SMHIP_OK;
}

template <typename T, typename Op>
int run_contiguous(const void *a, const void *b, void *out, size_t n, hipStream_t s) {
    constexpr int W = VecTraits<T>::width;
    const T *pa = static_cast<const T *>(a), *pb = static_cast<const T *>(b);
    T *po = static_cast<T *>(out);
    unsigned grid;
    const size_t n_vec = n / W;
    const int tail = (int)(n % W);
    const size_t threads = n_vec + (tail ? 1 : 0);
    if constexpr (IsHeavy<Op>::value) {
        hipLaunchKernelGGL((heavy_vec_kernel<T, Op, 0>), dim3(heavy_grid(n_vec)), dim3(kHeavyBlock), 0, s, pa, pb, T{}, po, n_vec, tail);
    } else if (n_vec >= kBigThreshold) {
        if (int rc = grid_for(threads, kBlockBig, &grid)) return rc;
        hipLaunchKernelGGL((contiguous_vec_kernel<T, Op, kBlockBig>), dim3(grid), dim3(kBlockBig), 0, s, pa, pb, po, n_vec, tail);
    } else {
        if (int rc = grid_for(threads, kBlockSmall, &grid)) return rc;
        hipLaunchKernelGGL((contiguous_vec_kernel<T, Op, kBlockSmall>), dim3(grid), dim3(kBlockSmall), 0, s, pa, pb, po, n_vec, tail);
    }
    SMHIP_LAUNCH_CHECK("contiguous");
    return SMHIP_OK;
}

// sm::pow(a, s) with an exponent whose power is ONE correctly rounded IEEE operation: x^2 = x*x, x^1 = x, x^-1 = 1/x,
// x^0.5 = sqrt(x) (with pow's own answers for -0 and -inf).  These are exact-rounded -- never worse than the general
// 2^(y log2 x) path's <= 1 ULP -- and stream at the array-scalar rate (81 % of peak instead of 71 %).  Squares are what
// the reference's own float pow benchmark raises to (benchmark/pow.cpp:33-49).
template <typename T> struct PowSquare { static __device__ __forceinline__ T apply(T a, T) { return a * a; } };
template <typename T> struct PowIdentity { static __device__ __forceinline__ T apply(T a, T) { return a * T(1); } };  // quiets a signalling NaN
template <typename T> struct PowReciprocal { static __device__ __forceinline__ T apply(T a, T) { return T(1) / a; } };
template <typename T> struct PowSqrt {
    static __device__ __forceinline__ T apply(T a, T) {
        if (a == T(0)) return T(0);                               // pow(-0, 0.5) = +0 (sqrt keeps the sign)
        if (a == -__builtin_huge_val()) return (T)__builtin_huge_val();  // pow(-inf, 0.5) = +inf (sqrt: NaN)
        if constexpr (sizeof(T) == 4) return __builtin_sqrtf(a);
        else return __builtin_sqrt(a);
    }
};

template <typename T, typename Op, bool SWAPPED>
int run_scalar(const void *a, T value, size_t n, void *out, hipStream_t s) {
    constexpr int W = VecTraits<T>::width;
    const T *pa = static_cast<const T *>(a);
    T *po = static_cast<T *>(out);
    unsigned grid;
    const size_t n_vec = n / W;
    const int tail = (int)(n % W);
    const size_t threads = n_vec + (tail ? 1 : 0);
    if constexpr (IsHeavy<Op>::value && std::is_floating_point<T>::value && !SWAPPED) {
        if (value == T(2) || value == T(1) || value == T(-1) || value == T(0.5)) {
            if (int rc = grid_for(threads, kBlockSmall, &grid)) return rc;
            const dim3 g(grid), b(kBlockSmall);
            if (value == T(2)) hipLaunchKernelGGL((scalar_vec_kernel<T, PowSquare<T>, kBlockSmall, false>), g, b, 0, s, pa, value, po, n_vec, tail);
            else if (value == T(1)) hipLaunchKernelGGL((scalar_vec_kernel<T, PowIdentity<T>, kBlockSmall, false>), g, b, 0, s, pa, value, po, n_vec, tail);
            else if (value == T(-1)) hipLaunchKernelGGL((scalar_vec_kernel<T, PowReciprocal<T>, kBlockSmall, false>), g, b, 0, s, pa, value, po, n_vec, tail);
            else hipLaunchKernelGGL((scalar_vec_kernel<T, PowSqrt<T>, kBlockSmall, false>), g, b, 0, s, pa, value, po, n_vec, tail);
            SMHIP_LAUNCH_CHECK("array_scalar pow (exact form)");
            return SMHIP_OK;
        }
    }
    constexpr bool kHeavy = IsHeavy<Op>::value;
    if constexpr (kHeavy) {
        hipLaunchKernelGGL((heavy_vec_kernel<T, Op, SWAPPED ? 2 : 1>), dim3(heavy_grid(n_vec)), dim3(kHeavyBlock), 0, s, pa,
                           static_cast<const T *>(nullptr), value, po, n_vec, tail);
    } else {
        // one read + one write stream: workgroups of 256 at every size (tools/sweep_scalar.hip, profiles/r01_sweep_scalar.txt:
        // 81.7 % of peak at N = 2^28 against 78.7 % with 1024, and two or more vectors per lane lose 4-10 %)
        if (int rc = grid_for(threads, kBlockSmall, &grid)) return rc;
        hipLaunchKernelGGL((scalar_vec_kernel<T, Op, kBlockSmall, SWAPPED>), dim3(grid), dim3(kBlockSmall), 0, s, pa, value, po, n_vec, tail);
    }
    SMHIP_LAUNCH_CHECK("array_scalar");
    return SMHIP_OK;
}

template <typename T, typename Op, bool SWAPPED>
int run_devscalar(const void *a, const void *sp, size_t n, void *out, hipStream_t s) {
    constexpr int W = VecTraits<T>::width;
    const T *pa = static_cast<const T *>(a);
    const T *ps = static_cast<const T *>(sp);
    T *po = static_cast<T *>(out);
    unsigned grid;
    const size_t n_vec = n / W;
    const int tail = (int)(n % W);
    if (int rc = grid_for(n_vec + (tail ? 1 : 0), kBlockSmall, &grid)) return rc;
    hipLaunchKernelGGL((devscalar_vec_kernel<T, Op, kBlockSmall, SWAPPED>), dim3(grid), dim3(kBlockSmall), 0, s, pa, ps, po, n_vec, tail);
    SMHIP_LAUNCH_CHECK("array_devscalar");
    return SMHIP_OK;
}

// op x dtype dispatch: F(T, Op) expands to a call.
#define SMHIP_DISPATCH_OP(T, F)                                  \
    switch (op) {                                                \
        case SMHIP_OP_ADD: return F(T, AddOp<T>);                \
        case SMHIP_OP_SUB: return F(T, SubtractOp<T>);           \
        case SMHIP_OP_MUL: return F(T, MultiplyOp<T>);           \
        case SMHIP_OP_DIV: return F(T, DivideOp<T>);             \
        case SMHIP_OP_POW: return F(T, PowOp<T>);                \
        case SMHIP_OP_LEFT: return F(T, LeftOp<T>);                \
    }                                                            \
    break;
#define SMHIP_DISPATCH(F)                                        \
    switch (dtype) {                                             \
        case SMHIP_F32: SMHIP_DISPATCH_OP(float, F)              \
        case SMHIP_F64: SMHIP_DISPATCH_OP(double, F)             \
        case SMHIP_I32: SMHIP_DISPATCH_OP(int32_t, F)            \
        case SMHIP_I64: SMHIP_DISPATCH_OP(int64_t, F)            \
    }

}  // namespace

int launch_contiguous(int op, int dtype, const void *a, const void *b, void *out, size_t n, hipStream_t s) {
    if (n == 0) return SMHIP_OK;
#define F(T, OP) run_contiguous<T, OP>(a, b, out, n, s)
    SMHIP_DISPATCH(F)
#undef F
    return fail(SMHIP_ERR_INVALID, "contiguous: bad op %d / dtype %d", op, dtype);
}

int launch_array_scalar(int op, int dtype, const void *a, const void *value_host, size_t n, void *out, hipStream_t s) {
    if (n == 0) return SMHIP_OK;
#define F(T, OP) run_scalar<T, OP, false>(a, *static_cast<const T *>(value_host), n, out, s)
    SMHIP_DISPATCH(F)
#undef F
    return fail(SMHIP_ERR_INVALID, "array_scalar: bad op %d / dtype %d", op, dtype);
}

int launch_array_devscalar(int op, int dtype, const void *a, const void *value_dev, size_t n, void *out, bool swapped,
                           hipStream_t s) {
    if (n == 0) return SMHIP_OK;
    if (swapped) {
#define F(T, OP) run_devscalar<T, OP, true>(a, value_dev, n, out, s)
        SMHIP_DISPATCH(F)
#undef F
    } else {
#define F(T, OP) run_devscalar<T, OP, false>(a, value_dev, n, out, s)
        SMHIP_DISPATCH(F)
#undef F
    }
    return fail(SMHIP_ERR_INVALID, "array_devscalar: bad op %d / dtype %d", op, dtype);
}

}  // namespace smhip
