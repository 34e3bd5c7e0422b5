/*
 * sm_oracle.h -- CPU restatement of simpleMath's element_wise_op hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing outside tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may include, link or call this.  The product
 * (include/smhip.h, simplemath_amd/csrc) never routes through it.
 *
 * Parity status: PINNED.  Every function here is checked (tests/test_oracle.py)
 * against (a) the reference's own known-answer tests, restated as data in
 * tests/golden/reference_kat.json, and (b) randomised vectors produced by the
 * reference's own headers compiled into oracle/_ref/libsmref.so
 * (tests/golden/make_golden.py writes the .npz fixtures beside it).  The one exception is
 * float pow: the reference has no array path for it (pow.h:12-13 is an
 * undefined symbol), its scalar arithmetic is glibc powf, and its own float
 * pow tests are commented out -- that function is "parity unpinned" by the
 * reference and pinned here by correctly-rounded fp64 golden vectors.
 *
 * All file:line citations are relative to the reference tree.
 */
#ifndef SM_ORACLE_H
#define SM_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SMO_MAX_NDIM 6 /* include/math/helpers.h:4 */

enum { SMO_ADD = 0, SMO_SUB = 1, SMO_MUL = 2, SMO_DIV = 3, SMO_POW = 4 };
enum { SMO_F32 = 0, SMO_F64 = 1, SMO_I32 = 2, SMO_I64 = 3 };

/* sm::broadcast, include/SMUtils.h:34-99.  Returns the broadcast rank, or -1
 * when the shapes are incompatible (the reference throws std::runtime_error at
 * SMUtils.h:76-78).  Outputs hold max(nd1, nd2) entries. */
int smo_broadcast(int nd1, const size_t *shape1, const size_t *strides1,
                  int nd2, const size_t *shape2, const size_t *strides2,
                  size_t *result_shape, size_t *new_strides1,
                  size_t *new_strides2, size_t *total_size);

/* is_contiguous, include/math/helpers.h:130-139. */
int smo_is_contiguous(int ndim, const size_t *shape, const size_t *stride);

/* element_wise_op<T,Op>, include/math/calculate.h:5-99 (dispatch + general
 * N-D loop) and handle_contiguous_arrays, calculate.h:101-134.
 * `quirk_1d` != 0 reproduces calculate.h:10 literally (every 1-D call is
 * treated as dense regardless of strides); 0 walks the strides (the corrected
 * behaviour the HIP path implements).
 * Returns 0, or -1 for an unsupported op/dtype/ndim. */
int smo_elementwise(int op, int dtype, const void *a, const size_t *stride_a,
                    const void *b, const size_t *stride_b, size_t n,
                    void *result, const size_t *shape, int ndim, int quirk_1d);

/* handle_contiguous_arrays alone (calculate.h:101-134): single thread, like
 * the reference. */
int smo_contiguous(int op, int dtype, const void *a, const void *b,
                   void *result, size_t n);

/* Best-effort variant for the CPU baseline's second leg: same arithmetic,
 * OpenMP over all cores. Not reference behaviour. */
int smo_contiguous_mt(int op, int dtype, const void *a, const void *b,
                      void *result, size_t n);

/* array_scalar_op<T,Op>, calculate.h:137-169.  `value` points at one T.
 * For int32 pow, `int_pow_tail_libm` != 0 evaluates the last n % 8 elements as
 * the reference's tail does (std::pow(int,int) -> double -> int, pow.h:8-10);
 * 0 uses the vector body's square-and-multiply everywhere (the HIP path's
 * defined behaviour). */
int smo_array_scalar(int op, int dtype, const void *a, const void *value,
                     size_t n, void *result, int int_pow_tail_libm);

/* __sm256_powi_ps for one lane, include/math/simd/crafted_pow.h:54-103. */
int32_t smo_powi32(int32_t base, int32_t exponent);

/* dot_product<T>, include/math/product.h.  `lane_order` != 0 reproduces the
 * AVX2 accumulation order (8 f32 / 4 f64 / 8 i32 lane accumulators, mul then
 * add, low+high, buf[0..3], scalar tail: product.h:39-66, 88-115, 135-162);
 * 0 accumulates in fp64 (f32/f64) -- the numerically meaningful oracle the
 * HIP reduction is compared with.  Result written as one T at `out`. */
int smo_dot(int dtype, const void *a, const void *b, size_t n, void *out,
            int lane_order);

/* dot_product<std::complex<double>> (product.h:168-224): n {re, im} pairs each; {re, im} to out2.  avx_body == 0: the
 * scalar statement `result += a[i] * b[i]` for every element (the definition); != 0: as shipped, AVX body first (its
 * permutes count every product twice), scalar tail after. */
int smo_dot_c64(const double *a, const double *b, size_t n, double *out2, int avx_body);
/* The generic dot_product<T> (product.h:8-20) with T = std::complex<float>: n {re, im} float pairs; as_shipped != 0: the
 * reference's bits (contracted products, sequential float sums); 0: exact products, compensated fp64 sums. */
int smo_dot_c32(const float *a, const float *b, size_t n, float *out2, int as_shipped);
/* The generic dot_product<T> (product.h:8-20) for int8/uint8/int16/uint16/uint32/uint64 (kind 4..9): the sum of products
 * modulo 2^(8 sizeof T); one T to `out`. */
int smo_dot_int(int kind, const void *a, const void *b, size_t n, void *out);

/* Whole-array sum in fp64 (Neumaier-compensated).  No reference counterpart
 * (SURVEY 8c "Oracle for global sum"). */
double smo_sum_f64acc(int dtype, const void *a, size_t n);

/* result = a op b elementwise AND returns sum(result) in compensated fp64:
 * oracle for the fused add+sum of BASELINE config 5. */
double smo_contiguous_sum(int op, int dtype, const void *a, const void *b,
                          void *result, size_t n);

/* Counter-based uniform generator shared bit-for-bit with the HIP fill kernel
 * (simplemath_amd/csrc/fill.hip): element i of stream `seed` in [lo, hi). */
float smo_uniform_f32(uint64_t seed, uint64_t i, float lo, float hi);
void smo_fill_uniform_f32(float *dst, size_t n, uint64_t seed, uint64_t first,
                          float lo, float hi);

int smo_num_threads(void);

/* All-core forms for the CPU baseline timed beside the GPU (bench.py cpu_baseline); results are
 * those of the single-thread functions.  smo_set_threads: threads every later OpenMP region of the calling thread
 * uses, including the compiled reference's (same libgomp). */
void smo_set_threads(int n);
int smo_array_scalar_mt(int op, int dtype, const void *a, const void *value, size_t n, void *result);
double smo_contiguous_sum_mt(int op, int dtype, const void *a, const void *b, void *result, size_t n);

#ifdef __cplusplus
}
#endif
#endif
