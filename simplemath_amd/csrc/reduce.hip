// reduce.hip -- whole-array reductions.
//
// Replaces dot_product<T> (reference include/math/product.h:8-224, reached
// through SMArray::operator% at include/SMArray.h:213-215): one host thread,
// 8 f32 lane accumulators that stop absorbing addends past 2^24 (SURVEY 0:
// the dot of 2^28 ones returns 2^27).  Adds sum() and the fused a op b + sum
// of BASELINE config 5, which the reference lacks.
//
// Structure: each lane streams 16-byte vectors and accumulates privately
// (fp64 for f32/f64, wrapping 64-bit integers for i32/i64 -- so integer results
// are bit-identical to the reference in any order), then a 64-lane wavefront
// shuffle tree (__shfl_down), then LDS across the workgroup's waves, one partial
// per workgroup, and a second single-workgroup pass over the partials in a
// fixed order: results are deterministic run to run.
// Roofline: HBM-bound; sizeof(T) B/elem (sum), 2*sizeof(T) (dot),
// 3*sizeof(T) (fused op+sum: the sum adds no traffic).
#include <type_traits>

#include "internal.h"
#include "ops.hip.h"

namespace smhip {
namespace {

using namespace dev;

constexpr int kBlock = 256;
// 16-byte vectors per lane and operand (tools/reduce_rates.py, profiles/r01_reduce_rates.txt).  Read-only streams want
// more loads in flight than the 2R+1W streams do: a plain sum runs at 86 % of peak with two vectors per lane (65-69 %
// with one, 82-84 % with four or eight), a dot at 84 % with two per operand (79 % with one).  The fused op+sum writes as
// well and keeps ONE per operand like the streaming kernels (79.5 %; 75 % with two; profiles/r01_sweep_fused_sum.txt).
#ifndef SMHIP_SUM_VECTORS
#define SMHIP_SUM_VECTORS 2
#endif
#ifndef SMHIP_DOT_VECTORS
#define SMHIP_DOT_VECTORS 2
#endif
#ifndef SMHIP_FUSED_VECTORS
#define SMHIP_FUSED_VECTORS 1
#endif
constexpr int kSumVectors = SMHIP_SUM_VECTORS, kDotVectors = SMHIP_DOT_VECTORS;
constexpr int vec_per_thread(int mode) { return mode == 0 /* kSum */ ? kSumVectors : mode == 1 /* kDot */ ? kDotVectors : SMHIP_FUSED_VECTORS; }
constexpr int kFinalBlock = 1024;

template <typename T> struct AccOf { typedef double type; };
template <> struct AccOf<int32_t> { typedef uint64_t type; };
template <> struct AccOf<int64_t> { typedef uint64_t type; };

template <typename A> __device__ __forceinline__ A wave_reduce(A v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}
__device__ __forceinline__ uint64_t shfl_down_u64(uint64_t v, int off) {
    const uint32_t lo = __shfl_down((uint32_t)v, off, 64), hi = __shfl_down((uint32_t)(v >> 32), off, 64);
    return ((uint64_t)hi << 32) | lo;
}
template <> __device__ __forceinline__ uint64_t wave_reduce<uint64_t>(uint64_t v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += shfl_down_u64(v, off);
    return v;
}

// Workgroup total in thread 0.
template <typename A, int BLOCK> __device__ __forceinline__ A block_reduce(A v) {
    __shared__ A lds[BLOCK / 64];
    v = wave_reduce(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) lds[wave] = v;
    __syncthreads();
    if (wave == 0) {
        v = lane < BLOCK / 64 ? lds[lane] : A(0);
        v = wave_reduce(v);
    }
    return v;
}

template <typename T, typename A> __device__ __forceinline__ A widen(T x) { return (A)x; }
template <> __device__ __forceinline__ uint64_t widen<int32_t, uint64_t>(int32_t x) { return (uint64_t)(int64_t)x; }
template <> __device__ __forceinline__ uint64_t widen<int64_t, uint64_t>(int64_t x) { return (uint64_t)x; }

// acc += x*y.  f32: the product of two 24-bit significands is exact in fp64, so the
// whole dot is an fp64 fma chain (the reference's -mfma build fuses too, but into f32
// lanes).  f64: fused, one rounding per term.  i32/i64: product wraps in T like
// _mm256_mullo_epi32, then accumulates mod 2^64.
template <typename T, typename A> __device__ __forceinline__ void add_prod(A &acc, T x, T y) { acc += widen<T, A>(MultiplyOp<T>::apply(x, y)); }
template <> __device__ __forceinline__ void add_prod<float, double>(double &acc, float x, float y) { acc = __builtin_fma((double)x, (double)y, acc); }
template <> __device__ __forceinline__ void add_prod<double, double>(double &acc, double x, double y) { acc = __builtin_fma(x, y, acc); }

enum Mode { kSum = 0, kDot = 1, kFused = 2 };

// One vector's contribution: MODE kSum: acc += a;  kDot: acc += a*b;  kFused: out = a op b, acc += out.
template <typename T, typename Op, int MODE, typename A>
__device__ __forceinline__ void consume(const OpCtx<Op> &ctx, A &acc, typename VecTraits<T>::vec_t va, typename VecTraits<T>::vec_t vb,
                                        typename VecTraits<T>::vec_t *out_slot) {
    typedef typename VecTraits<T>::vec_t V;
    constexpr int W = VecTraits<T>::width;
    if constexpr (MODE == kFused) {
        const V r = apply_vec<Op, T>(ctx, va, vb);
        store_stream(out_slot, r);
#pragma unroll
        for (int k = 0; k < W; ++k) acc += widen<T, A>(r[k]);
    } else if constexpr (MODE == kDot) {
#pragma unroll
        for (int k = 0; k < W; ++k) add_prod<T, A>(acc, va[k], vb[k]);
    } else {
#pragma unroll
        for (int k = 0; k < W; ++k) acc += widen<T, A>(va[k]);
    }
}

// What a reduction hands back.  *out8 receives 8 bytes:
//   floats            the fp64 total;
//   ints, AS_DOUBLE   (double) of the exact 64-bit total         (sum, fused op+sum);
//   ints, !AS_DOUBLE  the total wrapped to T, sign-extended to int64 (dot: per-rank partials add up mod 2^32 / 2^64
//                     exactly like the reference's _mm256_add_epi32 accumulators, in any order).
// *out_native (optional) receives the value narrowed to T (dot's return type).
template <typename T, bool AS_DOUBLE>
__device__ __forceinline__ void write_result(typename AccOf<T>::type acc, void *__restrict__ out8, T *__restrict__ out_native) {
    if constexpr (std::is_floating_point<T>::value) {
        if (out8) *static_cast<double *>(out8) = acc;
        if (out_native) *out_native = (T)acc;
    } else {
        const T wrapped = (T)acc;
        if (out8) {
            if constexpr (AS_DOUBLE) *static_cast<double *>(out8) = (double)(int64_t)acc;
            else *static_cast<int64_t *>(out8) = (int64_t)wrapped;
        }
        if (out_native) *out_native = wrapped;
    }
}

// Each workgroup owns one tile of kBlock * kVecPerThread vectors.  Full tiles (all
// but possibly the last) take a guard-free path: every load of the tile is issued
// before the first use, so kVecPerThread (x2 operands) 16-byte loads are in flight
// per lane.  (Per-vector bounds guards made the compiler wait on each load pair
// in turn -- 2.6 TB/s instead of 6.)
template <typename T, typename Op, int MODE>
__global__ __launch_bounds__(kBlock, 8) void reduce_kernel(const T *__restrict__ a, const T *__restrict__ b, T *__restrict__ out,
                                                           size_t n_vec, size_t n, typename AccOf<T>::type *__restrict__ partials,
                                                           void *__restrict__ out8, T *__restrict__ out_native) {
    typedef typename AccOf<T>::type A;
    typedef typename VecTraits<T>::vec_t V;
    constexpr int W = VecTraits<T>::width;
    constexpr int kVecPerThread = vec_per_thread(MODE);
    constexpr size_t kTile = (size_t)kBlock * kVecPerThread;
    const V *av = reinterpret_cast<const V *>(a), *bv = reinterpret_cast<const V *>(b);
    V *ov = reinterpret_cast<V *>(out);
    const size_t tile0 = (size_t)blockIdx.x * kTile + threadIdx.x;
    OpCtx<Op> ctx;
    ctx.init();
    A acc = A(0);
    if ((size_t)blockIdx.x * kTile + kTile <= n_vec) {
        V va[kVecPerThread], vb[kVecPerThread];
#pragma unroll
        for (int u = 0; u < kVecPerThread; ++u) {
            va[u] = load_stream(av + tile0 + (size_t)u * kBlock);
            if constexpr (MODE != kSum) vb[u] = load_stream(bv + tile0 + (size_t)u * kBlock);
            else vb[u] = va[u];
        }
#pragma unroll
        for (int u = 0; u < kVecPerThread; ++u) consume<T, Op, MODE, A>(ctx, acc, va[u], vb[u], ov + tile0 + (size_t)u * kBlock);
    } else {
        for (int u = 0; u < kVecPerThread; ++u) {
            const size_t i = tile0 + (size_t)u * kBlock;
            if (i < n_vec) {
                const V va = load_stream(av + i);
                const V vb = MODE != kSum ? load_stream(bv + i) : va;
                consume<T, Op, MODE, A>(ctx, acc, va, vb, ov + i);
            }
        }
        // scalar tail (n % W elements): the last workgroup's first lane
        if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) {
            for (size_t k = n_vec * W; k < n; ++k) {
                if constexpr (MODE == kFused) {
                    const T r = Op::apply(a[k], b[k]);
                    out[k] = r;
                    acc += widen<T, A>(r);
                } else if constexpr (MODE == kDot) add_prod<T, A>(acc, a[k], b[k]);
                else acc += widen<T, A>(a[k]);
            }
        }
    }
    acc = block_reduce<A, kBlock>(acc);
    if (threadIdx.x == 0) {
        if (gridDim.x == 1) write_result<T, MODE != kDot>(acc, out8, out_native);  // a small array: no second launch
        else partials[blockIdx.x] = acc;
    }
}

// Intermediate pass when there are many partials (one vector per lane means one partial per 4 KiB
// of each operand): every workgroup folds kFoldSpan of them into one, in a fixed order.
constexpr int kFoldSpan = 1024;  // 2^18 partials (2^28 f32) -> 256 workgroups here, 256 values for the last pass
template <typename A>
__global__ __launch_bounds__(kBlock) void fold_kernel(const A *__restrict__ in, size_t count, A *__restrict__ out) {
    const size_t base = (size_t)blockIdx.x * kFoldSpan;
    A acc = A(0);
    for (int k = threadIdx.x; k < kFoldSpan; k += kBlock)
        if (base + k < count) acc += in[base + k];
    acc = block_reduce<A, kBlock>(acc);
    if (threadIdx.x == 0) out[blockIdx.x] = acc;
}

// Last pass: fixed-order sum of the partials (write_result above says what is handed back).
template <typename T, bool AS_DOUBLE>
__global__ __launch_bounds__(kFinalBlock) void finalize_kernel(const typename AccOf<T>::type *__restrict__ partials, size_t count,
                                                               void *__restrict__ out8, T *__restrict__ out_native) {
    typedef typename AccOf<T>::type A;
    A acc = A(0);
    for (size_t i = threadIdx.x; i < count; i += kFinalBlock) acc += partials[i];
    acc = block_reduce<A, kFinalBlock>(acc);
    if (threadIdx.x == 0) write_result<T, AS_DOUBLE>(acc, out8, out_native);
}

inline bool aligned16(const void *p) { return p == nullptr || (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// std::complex<double> dot (reference product.h:168-224): sum a[i] * b[i], unconjugated, as the
// reference's scalar tail defines it (its AVX body adds every real/imaginary product twice --
// _mm256_permute_pd(va, 0x0) duplicates lanes, rbuf[0..3] are then all summed -- which is taken as a
// bug, not as semantics).  One complex = one 16-byte vector {re, im}; separate fp64 fma chains for
// the real and imaginary sums, grid-stride, then the same wave / LDS / partials tree.
typedef double dbl2 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(kBlock) void cdot_kernel(const dbl2 *__restrict__ a, const dbl2 *__restrict__ b, size_t n,
                                                      double *__restrict__ partials, size_t blocks) {
    // one-shot like reduce_kernel: a workgroup owns kBlock * 2 consecutive elements, two per lane and operand in flight;
    // its partial sums go to partials[block] (real) and partials[blocks + block] (imaginary)
    double re = 0.0, im = 0.0;
    auto acc = [&](dbl2 x, dbl2 y) {
        re = __builtin_fma(x[0], y[0], re);
        re = __builtin_fma(-x[1], y[1], re);
        im = __builtin_fma(x[0], y[1], im);
        im = __builtin_fma(x[1], y[0], im);
    };
    const size_t i0 = (size_t)blockIdx.x * (kBlock * 2) + threadIdx.x, i1 = i0 + kBlock;
    if (i1 < n) {
        const dbl2 x0 = load_stream(a + i0), y0 = load_stream(b + i0), x1 = load_stream(a + i1), y1 = load_stream(b + i1);
        acc(x0, y0);
        acc(x1, y1);
    } else if (i0 < n) {
        acc(load_stream(a + i0), load_stream(b + i0));
    }
    __shared__ double lds[2][kBlock / 64];
    re = wave_reduce(re);
    im = wave_reduce(im);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { lds[0][wave] = re; lds[1][wave] = im; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double r = 0.0, m = 0.0;
        for (int w = 0; w < kBlock / 64; ++w) { r += lds[0][w]; m += lds[1][w]; }
        partials[blockIdx.x] = r;
        partials[blocks + blockIdx.x] = m;
    }
}
__global__ __launch_bounds__(64) void cdot_finalize_kernel(const double *__restrict__ re_parts, const double *__restrict__ im_parts,
                                                           size_t count, double *__restrict__ out2) {
    double re = 0.0, im = 0.0;
    for (size_t i = threadIdx.x; i < count; i += 64) { re += re_parts[i]; im += im_parts[i]; }
    re = wave_reduce(re);
    im = wave_reduce(im);
    if (threadIdx.x == 0) { out2[0] = re; out2[1] = im; }
}

template <typename T, typename Op, int MODE>
int run_reduce(const void *a_, const void *b_, void *out_, size_t n, void *out8, void *out_native, hipStream_t s) {
    typedef typename AccOf<T>::type A;
    constexpr int W = VecTraits<T>::width;
    const T *a = static_cast<const T *>(a_), *b = static_cast<const T *>(b_);
    T *out = static_cast<T *>(out_);
    const size_t n_vec = n / W;
    const size_t tile = (size_t)kBlock * vec_per_thread(MODE);
    size_t blocks;
    blocks = n_vec / tile + 1;  // the last workgroup takes the partial tile and the n % W tail (maybe empty)
    if (blocks > 0x7fffffffu) return fail(SMHIP_ERR_UNSUPPORTED, "reduction too large (%zu workgroups)", blocks);
    const size_t folded = (blocks + kFoldSpan - 1) / kFoldSpan;
    double *scratch;
    ScratchLease lease;
    if (int rc = lease.take(blocks + folded, &scratch)) return rc;
    A *partials = reinterpret_cast<A *>(scratch);
    hipLaunchKernelGGL((reduce_kernel<T, Op, MODE>), dim3((unsigned)blocks), dim3(kBlock), 0, s, a, b, out, n_vec, n, partials, out8,
                       static_cast<T *>(out_native));
    SMHIP_LAUNCH_CHECK("reduce");
    if (blocks == 1) return SMHIP_OK;  // the single workgroup wrote the result itself
    if (blocks > (size_t)kFoldSpan) {
        hipLaunchKernelGGL(fold_kernel<A>, dim3((unsigned)folded), dim3(kBlock), 0, s, partials, blocks, partials + blocks);
        SMHIP_LAUNCH_CHECK("reduce fold");
        partials += blocks;
        blocks = folded;
    }
    hipLaunchKernelGGL((finalize_kernel<T, MODE != kDot>), dim3(1), dim3(kFinalBlock), 0, s, partials, blocks, out8, static_cast<T *>(out_native));
    SMHIP_LAUNCH_CHECK("reduce finalize");
    return SMHIP_OK;
}

}  // namespace

int launch_sum(int dtype, const void *a, size_t n, double *out_dev, hipStream_t s) {
    switch (dtype) {
        case SMHIP_F32: return run_reduce<float, AddOp<float>, kSum>(a, nullptr, nullptr, n, out_dev, nullptr, s);
        case SMHIP_F64: return run_reduce<double, AddOp<double>, kSum>(a, nullptr, nullptr, n, out_dev, nullptr, s);
        case SMHIP_I32: return run_reduce<int32_t, AddOp<int32_t>, kSum>(a, nullptr, nullptr, n, out_dev, nullptr, s);
        case SMHIP_I64: return run_reduce<int64_t, AddOp<int64_t>, kSum>(a, nullptr, nullptr, n, out_dev, nullptr, s);
    }
    return fail(SMHIP_ERR_INVALID, "sum: bad dtype %d", dtype);
}

int launch_dot(int dtype, const void *a, const void *b, size_t n, double *out8_dev, void *out_native_dev, hipStream_t s) {
    switch (dtype) {
        case SMHIP_F32: return run_reduce<float, AddOp<float>, kDot>(a, b, nullptr, n, out8_dev, out_native_dev, s);
        case SMHIP_F64: return run_reduce<double, AddOp<double>, kDot>(a, b, nullptr, n, out8_dev, out_native_dev, s);
        case SMHIP_I32: return run_reduce<int32_t, AddOp<int32_t>, kDot>(a, b, nullptr, n, out8_dev, out_native_dev, s);
        case SMHIP_I64: return run_reduce<int64_t, AddOp<int64_t>, kDot>(a, b, nullptr, n, out8_dev, out_native_dev, s);
    }
    return fail(SMHIP_ERR_INVALID, "dot: bad dtype %d", dtype);
}

int launch_cdot(const void *a, const void *b, size_t n, double *out2_dev, hipStream_t s) {
    const size_t blocks = n / (kBlock * 2) + 1;
    if (blocks > 0x7fffffffu) return fail(SMHIP_ERR_UNSUPPORTED, "complex dot too large (%zu workgroups)", blocks);
    const size_t folded = (blocks + kFoldSpan - 1) / kFoldSpan;
    double *scratch;
    ScratchLease lease;
    if (int rc = lease.take(2 * blocks + 2 * folded, &scratch)) return rc;
    hipLaunchKernelGGL(cdot_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, s, static_cast<const dbl2 *>(a), static_cast<const dbl2 *>(b), n, scratch, blocks);
    SMHIP_LAUNCH_CHECK("cdot");
    const double *re = scratch, *im = scratch + blocks;
    size_t count = blocks;
    if (blocks > (size_t)kFoldSpan) {  // fixed-order fold of the two partial arrays, as in run_reduce
        double *fre = scratch + 2 * blocks, *fim = fre + folded;
        hipLaunchKernelGGL(fold_kernel<double>, dim3((unsigned)folded), dim3(kBlock), 0, s, re, blocks, fre);
        hipLaunchKernelGGL(fold_kernel<double>, dim3((unsigned)folded), dim3(kBlock), 0, s, im, blocks, fim);
        SMHIP_LAUNCH_CHECK("cdot fold");
        re = fre; im = fim; count = folded;
    }
    hipLaunchKernelGGL(cdot_finalize_kernel, dim3(1), dim3(64), 0, s, re, im, count, out2_dev);
    SMHIP_LAUNCH_CHECK("cdot finalize");
    return SMHIP_OK;
}

int launch_contiguous_sum(int op, int dtype, const void *a, const void *b, void *out, size_t n, double *sum_dev, hipStream_t s) {
#define SMHIP_DISPATCH_OP(T)                                                                                          \
    switch (op) {                                                                                                     \
        case SMHIP_OP_ADD: return run_reduce<T, AddOp<T>, kFused>(a, b, out, n, sum_dev, nullptr, s);                 \
        case SMHIP_OP_SUB: return run_reduce<T, SubtractOp<T>, kFused>(a, b, out, n, sum_dev, nullptr, s);            \
        case SMHIP_OP_MUL: return run_reduce<T, MultiplyOp<T>, kFused>(a, b, out, n, sum_dev, nullptr, s);            \
        case SMHIP_OP_DIV: return run_reduce<T, DivideOp<T>, kFused>(a, b, out, n, sum_dev, nullptr, s);              \
        case SMHIP_OP_POW: return run_reduce<T, PowOp<T>, kFused>(a, b, out, n, sum_dev, nullptr, s);                 \
        case SMHIP_OP_LEFT: return run_reduce<T, LeftOp<T>, kFused>(a, b, out, n, sum_dev, nullptr, s);                 \
    }                                                                                                                 \
    break;
    switch (dtype) {
        case SMHIP_F32: SMHIP_DISPATCH_OP(float)
        case SMHIP_F64: SMHIP_DISPATCH_OP(double)
        case SMHIP_I32: SMHIP_DISPATCH_OP(int32_t)
        case SMHIP_I64: SMHIP_DISPATCH_OP(int64_t)
    }
#undef SMHIP_DISPATCH_OP
    return fail(SMHIP_ERR_INVALID, "contiguous_sum: bad op %d / dtype %d", op, dtype);
}

// For reductions whose first pass is compiled at run time (jit.hip: fused expression + sum): `partials` holds one
// accumulator per workgroup of that pass (double for float types, uint64 for integer types, as AccOf<T>), with room
// for blocks / kFoldSpan + 1 more behind them; this runs the fixed-order fold and the final pass into *out8 (fp64).
int reduce_finish(int dtype, void *partials_, size_t blocks, double *out8, hipStream_t s) {
    const size_t folded = (blocks + kFoldSpan - 1) / kFoldSpan;
    auto go = [&](auto tag) {
        typedef decltype(tag) T;
        typedef typename AccOf<T>::type A;
        A *partials = static_cast<A *>(partials_);
        size_t count = blocks;
        if (blocks > (size_t)kFoldSpan) {
            hipLaunchKernelGGL(fold_kernel<A>, dim3((unsigned)folded), dim3(kBlock), 0, s, partials, blocks, partials + blocks);
            partials += blocks;
            count = folded;
        }
        hipLaunchKernelGGL((finalize_kernel<T, true>), dim3(1), dim3(kFinalBlock), 0, s, partials, count, out8, static_cast<T *>(nullptr));
    };
    switch (dtype) {
        case SMHIP_F32: go(float{}); break;
        case SMHIP_F64: go(double{}); break;
        case SMHIP_I32: go(int32_t{}); break;
        case SMHIP_I64: go(int64_t{}); break;
        default: return fail(SMHIP_ERR_INVALID, "reduce_finish: bad dtype %d", dtype);
    }
    SMHIP_LAUNCH_CHECK("reduce finish");
    return SMHIP_OK;
}

}  // namespace smhip
