/*
 * sm_oracle.c -- CPU restatement of simpleMath's element_wise_op hot path.
 * TEST INFRASTRUCTURE ONLY (see sm_oracle.h for the rules and parity status).
 *
 * Build: gcc -O3 -std=c11 -mavx2 -mfma -ffp-contract=off -fopenmp -fPIC -shared
 * (-ffp-contract=off: every float op below is the single IEEE operation the
 * reference's intrinsic performs; the places where the reference build fuses
 * are written with explicit fma()).
 *
 * Citations are file:line in the reference tree (/root/reference).
 */
#include "sm_oracle.h"

#include <limits.h>
#include <math.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define SMO_CHUNK 1024          /* include/macros.h:16 CHUNK_SIZE */
#define SMO_OMP_THRESHOLD 100000 /* calculate.h:47, :152 */

/* ---------------------------------------------------------------- shapes */

/* include/SMUtils.h:34-99 */
int smo_broadcast(int nd1, const size_t *shape1, const size_t *strides1,
                  int nd2, const size_t *shape2, const size_t *strides2,
                  size_t *result_shape, size_t *new_strides1,
                  size_t *new_strides2, size_t *total_size) {
    const int nd = nd1 > nd2 ? nd1 : nd2;
    const int off1 = nd - nd1, off2 = nd - nd2;
    size_t total = 1;
    for (int i = 0; i < nd; ++i) {
        /* right-align; missing leading dims are size 1 / stride 0 (:51-72) */
        size_t d1 = i < off1 ? 1 : shape1[i - off1];
        size_t s1 = i < off1 ? 0 : strides1[i - off1];
        size_t d2 = i < off2 ? 1 : shape2[i - off2];
        size_t s2 = i < off2 ? 0 : strides2[i - off2];
        if (d1 != d2 && d1 != 1 && d2 != 1) return -1; /* :76-78 throws */
        result_shape[i] = d1 > d2 ? d1 : d2;
        total *= result_shape[i];
        if (d1 == 1 && d2 > 1) s1 = 0; /* :83-88 */
        if (d2 == 1 && d1 > 1) s2 = 0;
        new_strides1[i] = s1;
        new_strides2[i] = s2;
    }
    *total_size = total;
    return nd;
}

/* include/math/helpers.h:130-139 */
int smo_is_contiguous(int ndim, const size_t *shape, const size_t *stride) {
    size_t expected = 1;
    for (int i = ndim - 1; i >= 0; --i) {
        if (stride[i] != expected) return 0;
        expected *= shape[i];
    }
    return 1;
}

/* ------------------------------------------------------- scalar Op::apply */

/* int32 + - * wrap (two's complement), as _mm256_{add,sub,mullo}_epi32 do:
 * add.h:64-82, subtract.h:65-83, multiply.h:68-86. */
static inline int32_t i32_add(int32_t a, int32_t b) { return (int32_t)((uint32_t)a + (uint32_t)b); }
static inline int32_t i32_sub(int32_t a, int32_t b) { return (int32_t)((uint32_t)a - (uint32_t)b); }
static inline int32_t i32_mul(int32_t a, int32_t b) { return (int32_t)((uint32_t)a * (uint32_t)b); }
static inline int64_t i64_add(int64_t a, int64_t b) { return (int64_t)((uint64_t)a + (uint64_t)b); }
static inline int64_t i64_sub(int64_t a, int64_t b) { return (int64_t)((uint64_t)a - (uint64_t)b); }
static inline int64_t i64_mul(int64_t a, int64_t b) { return (int64_t)((uint64_t)a * (uint64_t)b); }

/* division.h:67-70, :93-106: C `/`, truncating toward zero.  x/0 and
 * INT_MIN/-1 trap (SIGFPE) in the reference; a GPU cannot trap, so both sides
 * of the parity suite use this definition: x/0 = 0, INT_MIN/-1 = INT_MIN
 * (DESIGN.md "defined where the reference traps"). */
static inline int32_t i32_div(int32_t a, int32_t b) {
    if (b == 0) return 0;
    if (a == INT32_MIN && b == -1) return INT32_MIN;
    return a / b;
}
static inline int64_t i64_div(int64_t a, int64_t b) {
    if (b == 0) return 0;
    if (a == INT64_MIN && b == -1) return INT64_MIN;
    return a / b;
}

/* include/math/simd/crafted_pow.h:54-103, one lane. */
int32_t smo_powi32(int32_t base, int32_t exponent) {
    /* :60 _mm256_abs_epi32 (INT_MIN stays 0x80000000), :79 logical shift */
    uint32_t e = exponent < 0 ? (uint32_t)0 - (uint32_t)exponent : (uint32_t)exponent;
    int32_t cur = base, pos = 1;
    while (e != 0) {                       /* :65 (per-lane view of the loop) */
        if (e & 1u) pos = i32_mul(pos, cur); /* :67-72 */
        cur = i32_mul(cur, cur);           /* :76 */
        e >>= 1;                           /* :79 */
    }
    if (base == 0 && exponent > 0) pos = 0; /* :81-84 */
    int32_t neg = 0;                        /* :85 */
    if (base == 1) neg = 1;                 /* :88-89 */
    if (base == -1) neg = (exponent & 1) ? -1 : 1; /* :92-95 */
    return exponent < 0 ? neg : pos;        /* :99-102 */
}

static inline int64_t powi64(int64_t base, int64_t exponent) {
    uint64_t e = exponent < 0 ? (uint64_t)0 - (uint64_t)exponent : (uint64_t)exponent;
    int64_t cur = base, pos = 1;
    while (e != 0) {
        if (e & 1u) pos = i64_mul(pos, cur);
        cur = i64_mul(cur, cur);
        e >>= 1;
    }
    if (base == 0 && exponent > 0) pos = 0;
    int64_t neg = 0;
    if (base == 1) neg = 1;
    if (base == -1) neg = (exponent & 1) ? -1 : 1;
    return exponent < 0 ? neg : pos;
}

/* PowOp<int>::apply, pow.h:8-10: std::pow(int,int) promotes to double, the
 * result converts back to int (cvttsd2si: out-of-range -> INT_MIN on x86). */
static inline int32_t i32_pow_libm(int32_t base, int32_t exponent) {
    double r = pow((double)base, (double)exponent);
    if (!(r > -2147483649.0 && r < 2147483648.0)) return INT32_MIN;
    return (int32_t)r;
}

#define DEFINE_APPLY(NAME, T, ADD, SUB, MUL, DIV, POW)                       \
    static inline T NAME(int op, T a, T b) {                                 \
        switch (op) {                                                        \
            case SMO_ADD: return ADD;                                        \
            case SMO_SUB: return SUB;                                        \
            case SMO_MUL: return MUL;                                        \
            case SMO_DIV: return DIV;                                        \
            default: return POW;                                             \
        }                                                                    \
    }
/* add.h:7-9, subtract.h:7-9, multiply.h:9-11, division.h:10-12, pow.h:8-10 */
DEFINE_APPLY(apply_f32, float, a + b, a - b, a * b, a / b, powf(a, b))
DEFINE_APPLY(apply_f64, double, a + b, a - b, a * b, a / b, pow(a, b))
DEFINE_APPLY(apply_i32, int32_t, i32_add(a, b), i32_sub(a, b), i32_mul(a, b), i32_div(a, b), smo_powi32(a, b))
DEFINE_APPLY(apply_i64, int64_t, i64_add(a, b), i64_sub(a, b), i64_mul(a, b), i64_div(a, b), powi64(a, b))

/* ------------------------------------------ handle_contiguous_arrays<T,Op> */

/* calculate.h:101-134: 8-wide body + scalar tail, single thread.  The body
 * and the tail perform the same IEEE / wrapping operation per element, so a
 * plain loop (auto-vectorised to the same AVX2 instructions) is equivalent. */
#define DEFINE_CONTIG(NAME, T, APPLY)                                        \
    static void NAME(int op, const T *a, const T *b, T *r, size_t n) {       \
        switch (op) {                                                        \
            case SMO_ADD: for (size_t i = 0; i < n; ++i) r[i] = APPLY(SMO_ADD, a[i], b[i]); break; \
            case SMO_SUB: for (size_t i = 0; i < n; ++i) r[i] = APPLY(SMO_SUB, a[i], b[i]); break; \
            case SMO_MUL: for (size_t i = 0; i < n; ++i) r[i] = APPLY(SMO_MUL, a[i], b[i]); break; \
            case SMO_DIV: for (size_t i = 0; i < n; ++i) r[i] = APPLY(SMO_DIV, a[i], b[i]); break; \
            default:      for (size_t i = 0; i < n; ++i) r[i] = APPLY(SMO_POW, a[i], b[i]); break; \
        }                                                                    \
    }
DEFINE_CONTIG(contig_f32, float, apply_f32)
DEFINE_CONTIG(contig_f64, double, apply_f64)
DEFINE_CONTIG(contig_i32, int32_t, apply_i32)
DEFINE_CONTIG(contig_i64, int64_t, apply_i64)

int smo_contiguous(int op, int dtype, const void *a, const void *b, void *r, size_t n) {
    if (op < SMO_ADD || op > SMO_POW) return -1;
    switch (dtype) {
        case SMO_F32: contig_f32(op, a, b, r, n); return 0;
        case SMO_F64: contig_f64(op, a, b, r, n); return 0;
        case SMO_I32: contig_i32(op, a, b, r, n); return 0;
        case SMO_I64: contig_i64(op, a, b, r, n); return 0;
    }
    return -1;
}

/* Best-effort all-core forms for the CPU baseline beside the GPU: the same loops cut into one
 * contiguous block per thread (64-element aligned so vector bodies stay aligned), threads spread over the cores.
 * The thread count is whatever smo_set_threads() last set -- bench.py caps it at the physical cores this process may
 * actually use (cgroup quota, affinity): 256 unbound SMT threads under a 16-CPU quota ran SLOWER than one core. */
static void mt_block(size_t n, size_t *lo, size_t *hi) {
#ifdef _OPENMP
    const size_t nt = (size_t)omp_get_num_threads(), t = (size_t)omp_get_thread_num();
#else
    const size_t nt = 1, t = 0;
#endif
    size_t per = ((n + nt - 1) / nt + 63) & ~(size_t)63;
    *lo = t * per;
    *hi = *lo + per;
    if (*lo > n) *lo = n;
    if (*hi > n) *hi = n;
}

int smo_contiguous_mt(int op, int dtype, const void *a, const void *b, void *r, size_t n) {
    if (op < SMO_ADD || op > SMO_POW) return -1;
    const size_t esz = (dtype == SMO_F64 || dtype == SMO_I64) ? 8 : 4;
    int rc = 0;
#pragma omp parallel proc_bind(spread)
    {
        size_t lo, hi;
        mt_block(n, &lo, &hi);
        if (hi > lo) {
            int c = smo_contiguous(op, dtype, (const char *)a + lo * esz,
                                   (const char *)b + lo * esz, (char *)r + lo * esz, hi - lo);
            if (c) {
#pragma omp atomic write
                rc = c;
            }
        }
    }
    return rc;
}

/* ------------------------------------------------- element_wise_op<T,Op> */

/* calculate.h:16-96, the general N-D loop.  `canVectorize` is identically
 * false (:43-46), so only the scalar statement at :96 ever runs. */
#define DEFINE_GENERAL(NAME, T, APPLY)                                        \
    static void NAME(int op, const T *a, const size_t *sa, const T *b,        \
                     const size_t *sb, size_t n, T *result,                   \
                     const size_t *shape, int ndim) {                         \
        size_t prod_shape[SMO_MAX_NDIM], sal[SMO_MAX_NDIM], sbl[SMO_MAX_NDIM]; \
        for (int i = 0; i < ndim; ++i) { sal[i] = sa[i]; sbl[i] = sb[i]; }    \
        prod_shape[ndim - 1] = 1;                       /* :27-30 */          \
        for (int k = ndim - 2; k >= 0; --k)                                   \
            prod_shape[k] = shape[k + 1] * prod_shape[k + 1];                 \
        /* :47-49 static schedule over 1024-element chunks when n > 100000 */ \
        _Pragma("omp parallel for schedule(static) if (n > SMO_OMP_THRESHOLD)") \
        for (int64_t chunk_start = 0; chunk_start < (int64_t)n; chunk_start += SMO_CHUNK) { \
            size_t chunk_end = (size_t)chunk_start + SMO_CHUNK;               \
            if (chunk_end > n) chunk_end = n;                                 \
            for (size_t linear = (size_t)chunk_start; linear < chunk_end; ++linear) { \
                size_t offA = 0, offB = 0, rem = linear;                      \
                for (int k = 0; k < ndim; ++k) {        /* :58-63 */          \
                    size_t idx = rem / prod_shape[k];                         \
                    rem %= prod_shape[k];                                     \
                    offA += idx * sal[k];                                     \
                    offB += idx * sbl[k];                                     \
                }                                                             \
                result[linear] = APPLY(op, a[offA], b[offB]); /* :96 */       \
            }                                                                 \
        }                                                                     \
    }
DEFINE_GENERAL(general_f32, float, apply_f32)
DEFINE_GENERAL(general_f64, double, apply_f64)
DEFINE_GENERAL(general_i32, int32_t, apply_i32)
DEFINE_GENERAL(general_i64, int64_t, apply_i64)

int smo_elementwise(int op, int dtype, const void *a, const size_t *sa,
                    const void *b, const size_t *sb, size_t n, void *result,
                    const size_t *shape, int ndim, int quirk_1d) {
    if (op < SMO_ADD || op > SMO_POW || ndim < 1 || ndim > SMO_MAX_NDIM) return -1;
    /* calculate.h:10-11 fast-path predicate */
    int same = 1;
    for (int i = 0; i < ndim; ++i) same &= (sa[i] == sb[i]);
    int fast = (sa[ndim - 1] == 1 && sb[ndim - 1] == 1 && same &&
                smo_is_contiguous(ndim, shape, sa));
    if (ndim == 1 && quirk_1d) fast = 1; /* :10 `ndim == 1 ||` */
    if (fast) return smo_contiguous(op, dtype, a, b, result, n);
    switch (dtype) {
        case SMO_F32: general_f32(op, a, sa, b, sb, n, result, shape, ndim); return 0;
        case SMO_F64: general_f64(op, a, sa, b, sb, n, result, shape, ndim); return 0;
        case SMO_I32: general_i32(op, a, sa, b, sb, n, result, shape, ndim); return 0;
        case SMO_I64: general_i64(op, a, sa, b, sb, n, result, shape, ndim); return 0;
    }
    return -1;
}

/* ------------------------------------------------- array_scalar_op<T,Op> */

/* calculate.h:137-169: SIMD body over [0, n - n % width) under
 * `omp parallel for if (simd_end > 100000)`, scalar tail after it.  Flat over
 * n, strides ignored. */
#define DEFINE_SCALAR(NAME, T, APPLY)                                         \
    static void NAME(int op, const T *a, T v, size_t n, T *r) {               \
        _Pragma("omp parallel for schedule(static) if (n > SMO_OMP_THRESHOLD)") \
        for (int64_t i = 0; i < (int64_t)n; ++i) r[i] = APPLY(op, a[i], v);   \
    }
DEFINE_SCALAR(scalar_f32, float, apply_f32)
DEFINE_SCALAR(scalar_f64, double, apply_f64)
DEFINE_SCALAR(scalar_i32, int32_t, apply_i32)
DEFINE_SCALAR(scalar_i64, int64_t, apply_i64)

int smo_array_scalar(int op, int dtype, const void *a, const void *value,
                     size_t n, void *result, int int_pow_tail_libm) {
    if (op < SMO_ADD || op > SMO_POW) return -1;
    switch (dtype) {
        case SMO_F32: scalar_f32(op, a, *(const float *)value, n, result); return 0;
        case SMO_F64: scalar_f64(op, a, *(const double *)value, n, result); return 0;
        case SMO_I64: scalar_i64(op, a, *(const int64_t *)value, n, result); return 0;
        case SMO_I32: {
            const int32_t v = *(const int32_t *)value;
            scalar_i32(op, a, v, n, result);
            if (op == SMO_POW && int_pow_tail_libm) {
                /* calculate.h:140,166-168: elements past simd_end go through
                 * PowOp<int>::apply = std::pow (pow.h:8-10). AVX2 width 8. */
                const int32_t *ai = a;
                int32_t *ri = result;
                for (size_t i = n - n % 8; i < n; ++i) ri[i] = i32_pow_libm(ai[i], v);
            }
            return 0;
        }
    }
    return -1;
}

/* ------------------------------------------------------------ dot_product */

/* product.h:88-115 as built with -mavx2 -mfma: GCC contracts
 * _mm256_add_ps(vsum, _mm256_mul_ps(va, vb)) into vfmadd (its intrinsics are
 * plain vector expressions under the default -ffp-contract=fast), and the
 * tail `result += a[i] * b[i]` likewise; hence fmaf/fma here. */
static float dot_f32_lanes(const float *a, const float *b, size_t n) {
    float lane[8] = {0};
    size_t i = 0;
    for (; i + 7 < n; i += 8)
        for (int l = 0; l < 8; ++l) lane[l] = fmaf(a[i + l], b[i + l], lane[l]);
    float s4[4];
    for (int l = 0; l < 4; ++l) s4[l] = lane[l] + lane[l + 4]; /* :95-97 low+high */
    float result = 0.0f;
    result += ((s4[0] + s4[1]) + s4[2]) + s4[3];               /* :100 */
    for (; i < n; ++i) result = fmaf(a[i], b[i], result);      /* :113-114 */
    return result;
}

/* product.h:135-162 */
static double dot_f64_lanes(const double *a, const double *b, size_t n) {
    double lane[4] = {0};
    size_t i = 0;
    for (; i + 3 < n; i += 4)
        for (int l = 0; l < 4; ++l) lane[l] = fma(a[i + l], b[i + l], lane[l]);
    double s2[2] = {lane[0] + lane[2], lane[1] + lane[3]};     /* :142-144 */
    double result = 0.0;
    result += s2[0] + s2[1];                                   /* :147 */
    for (; i < n; ++i) result = fma(a[i], b[i], result);       /* :160-161 */
    return result;
}

/* product.h:26-69: wrapping arithmetic is associative and commutative, so the
 * lane order is unobservable. */
static int32_t dot_i32(const int32_t *a, const int32_t *b, size_t n) {
    uint32_t s = 0;
    for (size_t i = 0; i < n; ++i) s += (uint32_t)a[i] * (uint32_t)b[i];
    return (int32_t)s;
}
static int64_t dot_i64(const int64_t *a, const int64_t *b, size_t n) { /* :16-19 */
    uint64_t s = 0;
    for (size_t i = 0; i < n; ++i) s += (uint64_t)a[i] * (uint64_t)b[i];
    return (int64_t)s;
}

/* Neumaier-compensated fp64 accumulation: the "true" value oracle. */
typedef struct { double s, c; } ksum_t;
static inline void ksum_add(ksum_t *k, double x) {
    double t = k->s + x;
    if (fabs(k->s) >= fabs(x)) k->c += (k->s - t) + x;
    else k->c += (x - t) + k->s;
    k->s = t;
}

int smo_dot(int dtype, const void *a, const void *b, size_t n, void *out, int lane_order) {
    switch (dtype) {
        case SMO_F32:
            if (lane_order) *(float *)out = dot_f32_lanes(a, b, n);
            else {
                const float *x = a, *y = b;
                ksum_t k = {0, 0};
                for (size_t i = 0; i < n; ++i) ksum_add(&k, (double)x[i] * (double)y[i]);
                *(float *)out = (float)(k.s + k.c);
            }
            return 0;
        case SMO_F64:
            if (lane_order) *(double *)out = dot_f64_lanes(a, b, n);
            else {
                const double *x = a, *y = b;
                ksum_t k = {0, 0};
                for (size_t i = 0; i < n; ++i) {
                    double p = x[i] * y[i];
                    ksum_add(&k, p);
                    ksum_add(&k, fma(x[i], y[i], -p)); /* exact product error */
                }
                *(double *)out = k.s + k.c;
            }
            return 0;
        case SMO_I32: *(int32_t *)out = dot_i32(a, b, n); return 0;
        case SMO_I64: *(int64_t *)out = dot_i64(a, b, n); return 0;
    }
    return -1;
}

/* dot_product<std::complex<double>>, include/math/product.h:168-224.  a, b: n {re, im} pairs.
 *   avx_body == 0: the DEFINITION -- the scalar statement `result += a[i] * b[i]` (:221-222) for every i.  With the only
 *     flags the reference compiles under (g++ -O3 -mavx2 -mfma) GCC expands the complex product inline and
 *     contracts it: re = fma(ar, br, -(ai * bi)), im = fma(ar, bi, ai * br) (checked bit for bit against the compiled
 *     reference at n = 1, tests/test_oracle.py::test_golden_dot_extra); the sums are plain additions.
 *   avx_body != 0: the function as shipped -- for i + 1 < n the AVX body (:175-199) handles two elements per iteration:
 *     _mm256_permute_pd(va, 0x0) / (va, 0xF) DUPLICATE the real / imaginary part into both slots of each 128-bit lane, so
 *     all four lanes of `real` / `imag` carry products and rbuf[0] + rbuf[1] + rbuf[2] + rbuf[3] counts every product
 *     TWICE; then the scalar tail.  Kept as data:
 *     it is what the reference returns, and it is not what a dot product is. */
int smo_dot_c64(const double *a, const double *b, size_t n, double *out2, int avx_body) {
    double re = 0.0, im = 0.0;
    size_t i = 0;
    if (avx_body) {
        double vr[4] = {0, 0, 0, 0}, vi[4] = {0, 0, 0, 0};
        for (; i + 1 < n; i += 2) {
            for (int l = 0; l < 4; ++l) {
                const size_t e = i + (size_t)(l >> 1);  /* lanes 0,1: element i; lanes 2,3: element i + 1 */
                const double ar = a[2 * e], ai = a[2 * e + 1], br = b[2 * e], bi = b[2 * e + 1];
                /* GCC's _mm256_mul_pd / _mm256_sub_pd are plain vector operators, so -mfma contracts them: here it rounds
                 * the a_r products first and fuses the a_i ones (vmulpd, vmulpd, vfnmadd231pd, vfmadd132pd in the compiled
                 * reference) -- the scalar tail below came out the other way round */
                vr[l] += fma(-ai, bi, ar * br);
                vi[l] += fma(ai, br, ar * bi);
            }
        }
        re = ((vr[0] + vr[1]) + vr[2]) + vr[3];
        im = ((vi[0] + vi[1]) + vi[2]) + vi[3];
    }
    for (; i < n; ++i) {
        const double ar = a[2 * i], ai = a[2 * i + 1], br = b[2 * i], bi = b[2 * i + 1];
        re += fma(ar, br, -(ai * bi));
        im += fma(ar, bi, ai * br);
    }
    out2[0] = re;
    out2[1] = im;
    return 0;
}

/* The generic dot_product<T> (product.h:8-20) instantiated with std::complex<float>: `T sum = 0; sum += a[i] * b[i]`, n
 * {re, im} pairs of floats.
 *   as_shipped != 0: the reference's bits -- GCC (-O3 -mfma) expands the complex product inline and contracts it,
 *     re = fmaf(ar, br, -(ai * bi)), im = fmaf(ar, bi, ai * br) (vmulss, vmulss, vfmsub231ss, vfmadd231ss in the compiled
 *     reference), and the sums are sequential float additions.
 *   as_shipped == 0: the numerically meaningful value the HIP kernel is compared with: exact products, compensated fp64 sums. */
int smo_dot_c32(const float *a, const float *b, size_t n, float *out2, int as_shipped) {
    if (as_shipped) {
        float re = 0.0f, im = 0.0f;
        for (size_t i = 0; i < n; ++i) {
            const float ar = a[2 * i], ai = a[2 * i + 1], br = b[2 * i], bi = b[2 * i + 1];
            re += fmaf(ar, br, -(ai * bi));
            im += fmaf(ar, bi, ai * br);
        }
        out2[0] = re;
        out2[1] = im;
        return 0;
    }
    ksum_t kr = {0, 0}, ki = {0, 0};
    for (size_t i = 0; i < n; ++i) {
        const double ar = a[2 * i], ai = a[2 * i + 1], br = b[2 * i], bi = b[2 * i + 1];
        ksum_add(&kr, ar * br);  /* products of two floats are exact in fp64 */
        ksum_add(&kr, -(ai * bi));
        ksum_add(&ki, ar * bi);
        ksum_add(&ki, ai * br);
    }
    out2[0] = (float)(kr.s + kr.c);
    out2[1] = (float)(ki.s + ki.c);
    return 0;
}

/* The generic dot_product<T> (product.h:8-20) for the integer types the specialisations do not cover: `T sum = 0; sum +=
 * a[i] * b[i]` -- the product is formed in the promoted type and the sum is cut back to T every step, i.e. the exact sum
 * of products modulo 2^(8 sizeof T) (unsigned int is routed to the int32 kernel, :10-15: the same value).
 * kind: 4 = int8, 5 = uint8, 6 = int16, 7 = uint16, 8 = uint32, 9 = uint64 (smhip.h's SMHIP_I8 ... SMHIP_U64); `out`
 * receives one T. */
int smo_dot_int(int kind, const void *a, const void *b, size_t n, void *out) {
    uint64_t s = 0;
    for (size_t i = 0; i < n; ++i) {
        uint64_t x, y;
        switch (kind) {
            case 4: x = (uint64_t)(int64_t)((const int8_t *)a)[i]; y = (uint64_t)(int64_t)((const int8_t *)b)[i]; break;
            case 5: x = ((const uint8_t *)a)[i]; y = ((const uint8_t *)b)[i]; break;
            case 6: x = (uint64_t)(int64_t)((const int16_t *)a)[i]; y = (uint64_t)(int64_t)((const int16_t *)b)[i]; break;
            case 7: x = ((const uint16_t *)a)[i]; y = ((const uint16_t *)b)[i]; break;
            case 8: x = ((const uint32_t *)a)[i]; y = ((const uint32_t *)b)[i]; break;
            case 9: x = ((const uint64_t *)a)[i]; y = ((const uint64_t *)b)[i]; break;
            default: return -1;
        }
        s += x * y;
    }
    switch (kind) {
        case 4: case 5: *(uint8_t *)out = (uint8_t)s; break;
        case 6: case 7: *(uint16_t *)out = (uint16_t)s; break;
        case 8: *(uint32_t *)out = (uint32_t)s; break;
        default: *(uint64_t *)out = s; break;
    }
    return 0;
}

double smo_sum_f64acc(int dtype, const void *a, size_t n) {
    ksum_t k = {0, 0};
    switch (dtype) {
        case SMO_F32: { const float *x = a; for (size_t i = 0; i < n; ++i) ksum_add(&k, x[i]); break; }
        case SMO_F64: { const double *x = a; for (size_t i = 0; i < n; ++i) ksum_add(&k, x[i]); break; }
        case SMO_I32: { const int32_t *x = a; for (size_t i = 0; i < n; ++i) ksum_add(&k, x[i]); break; }
        case SMO_I64: { const int64_t *x = a; for (size_t i = 0; i < n; ++i) ksum_add(&k, (double)x[i]); break; }
        default: return NAN;
    }
    return k.s + k.c;
}

double smo_contiguous_sum(int op, int dtype, const void *a, const void *b, void *result, size_t n) {
    if (smo_contiguous(op, dtype, a, b, result, n)) return NAN;
    return smo_sum_f64acc(dtype, result, n);
}

/* array_scalar_op over all cores, as the reference runs it (`#pragma omp parallel for` over the SIMD iterations,
 * calculate.h:152); float pow is std::pow per element (pow.h:8-10). */
int smo_array_scalar_mt(int op, int dtype, const void *a, const void *value, size_t n, void *result) {
    if (op < SMO_ADD || op > SMO_POW) return -1;
    const size_t esz = (dtype == SMO_F64 || dtype == SMO_I64) ? 8 : 4;
    int rc = 0;
#pragma omp parallel proc_bind(spread)
    {
        size_t lo, hi;
        mt_block(n, &lo, &hi);
        if (hi > lo) {
            int c = smo_array_scalar(op, dtype, (const char *)a + lo * esz, value, hi - lo, (char *)result + lo * esz, 0);
            if (c) {
#pragma omp atomic write
                rc = c;
            }
        }
    }
    return rc;
}

/* a op b stored + the fp64 sum of the results, all cores: per-thread partial sums added in thread order. */
double smo_contiguous_sum_mt(int op, int dtype, const void *a, const void *b, void *result, size_t n) {
    const size_t esz = (dtype == SMO_F64 || dtype == SMO_I64) ? 8 : 4;
    double partial[1024] = {0};
    int bad = 0, used = 1;
#pragma omp parallel proc_bind(spread)
    {
#ifdef _OPENMP
        const int t = omp_get_thread_num();
#pragma omp single
        used = omp_get_num_threads();
#else
        const int t = 0;
#endif
        size_t lo, hi;
        mt_block(n, &lo, &hi);
        if (hi > lo && t < 1024) {
            if (smo_contiguous(op, dtype, (const char *)a + lo * esz, (const char *)b + lo * esz, (char *)result + lo * esz, hi - lo)) {
#pragma omp atomic write
                bad = 1;
            }
            partial[t] = smo_sum_f64acc(dtype, (const char *)result + lo * esz, hi - lo);
        }
    }
    if (bad || used > 1024) return NAN;
    double s = 0;
    for (int t = 0; t < used; ++t) s += partial[t];
    return s;
}

/* Threads the *_mt forms (and every later OpenMP region of this thread, the compiled reference's included) use. */
void smo_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* ------------------------------------------------------ synthetic inputs */

static inline uint64_t mix64(uint64_t x) { /* splitmix64 finaliser */
    x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ULL;
    x ^= x >> 27; x *= 0x94D049BB133111EBULL;
    x ^= x >> 31;
    return x;
}

float smo_uniform_f32(uint64_t seed, uint64_t i, float lo, float hi) {
    uint64_t h = mix64(i + seed * 0x9E3779B97F4A7C15ULL);
    float u = (float)(h >> 40) * 0x1.0p-24f; /* exact: 24 bits */
    return fmaf(u, hi - lo, lo);
}

void smo_fill_uniform_f32(float *dst, size_t n, uint64_t seed, uint64_t first, float lo, float hi) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < (int64_t)n; ++i) dst[i] = smo_uniform_f32(seed, first + (uint64_t)i, lo, hi);
}

int smo_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
