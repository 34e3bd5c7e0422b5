/*
 * smhip.h -- C ABI of libsmhip.so, the MI355X (gfx950) implementation of
 * simpleMath's element_wise_op hot path.
 *
 * This is the drop-in boundary: plain C, plain pointers and sizes, no C++ or
 * torch types.  Each entry point names the reference interface it stands in
 * for (file:line in alielmorsy/simpleMath @ 2025-10-10).  The header-only C++
 * host side (include/sm.h, include/SMArray.h ...) calls these exactly where
 * the reference calls its loop templates; INTEGRATION.md shows the binding a
 * reference maintainer would add.
 *
 * Conventions
 *   - Every function returns SMHIP_OK (0) or a negative smhip_status; nothing
 *     throws across the boundary.  smhip_last_error() gives the text for the
 *     calling thread's last failure.
 *   - `a`, `b`, `out`, `dptr` are DEVICE pointers on the calling thread's
 *     current device (smhip_set_device) unless a parameter says `_host`.
 *     The caller owns every buffer; the library keeps no pointer past return.
 *   - Shapes and strides are in ELEMENTS (like the reference's
 *     std::vector<size_t>), passed as int64_t; strides must be >= 0.
 *   - Work is enqueued on the calling thread's stream (smhip_set_stream; the
 *     library's own per-device stream by default) and is asynchronous unless a
 *     function returns a value to host memory.
 *   - Callable from any host thread; device and stream selection are per thread.
 *   - There is no CPU fallback: without a HIP device every compute entry point
 *     fails with SMHIP_ERR_NO_DEVICE.
 */
#ifndef SMHIP_H
#define SMHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SMHIP_MAX_NDIM 6 /* MAX_NDIM, include/math/helpers.h:4 */

typedef enum smhip_status {
    SMHIP_OK = 0,
    SMHIP_ERR_INVALID = -1,     /* bad op / dtype / ndim / null pointer / negative stride */
    SMHIP_ERR_HIP = -2,         /* a HIP runtime call failed; see smhip_last_error() */
    SMHIP_ERR_NO_DEVICE = -3,   /* no usable gfx950 device */
    SMHIP_ERR_UNSUPPORTED = -4, /* combination not implemented */
    SMHIP_ERR_BROADCAST = -5    /* shapes not broadcastable (SMUtils.h:76-78 throws) */
} smhip_status;

/* Op policy ids: AddOp, SubtractOp, MultiplyOp, DivideOp, PowOp
 * (include/math/{add,subtract,multiply,division,pow}.h). */
typedef enum smhip_op {
    SMHIP_OP_ADD = 0, SMHIP_OP_SUB = 1, SMHIP_OP_MUL = 2, SMHIP_OP_DIV = 3, SMHIP_OP_POW = 4,
    /* out = a, b ignored: the dense copy of a strided / broadcast view (gather). No reference
     * Op; it is what SMArray::contiguous() and repeat() are made of. */
    SMHIP_OP_LEFT = 5
} smhip_op;

/* User-defined Ops (the README's "Extending with Custom Operations", README.md:86-133): on gfx950 an
 * Op's device form is its arithmetic as a HIP expression in `a` and `b` of the element type, e.g.
 * "(a + b) * 2".  smhip_register_op returns an id >= SMHIP_OP_USER_BASE that smhip_elementwise,
 * smhip_contiguous and smhip_array_scalar accept as `op`; kernels are compiled for gfx950 with hipRTC on
 * first use per element type and kernel variant, kept loaded for the life of the process, and their code objects are
 * cached on disk ($SMHIP_JIT_CACHE, default $XDG_CACHE_HOME/smhip or ~/.cache/smhip; "off" disables) so later
 * processes skip the compile.  A string that does not compile fails that first use with SMHIP_ERR_INVALID and the
 * compiler's message. */
#define SMHIP_OP_USER_BASE 100
int smhip_register_op(const char *hip_expression, int *op_id);

/* Element types: the SimdTraits<T> specialisations (helpers.h:23-119) plus
 * int64 (declared TODO at helpers.h:122-127). */
typedef enum smhip_dtype {
    SMHIP_F32 = 0, SMHIP_F64 = 1, SMHIP_I32 = 2, SMHIP_I64 = 3,
    /* Accepted by smhip_dot / smhip_dot_async ONLY: the other integer element types the reference's generic
     * dot_product<T> (product.h:8-20) is instantiated with through SMArray<T>::operator% (SMArray.h:22-27 admits any
     * arithmetic T).  They have no elementwise kernels -- the reference cannot compile those either (no SimdTraits). */
    SMHIP_I8 = 4, SMHIP_U8 = 5, SMHIP_I16 = 6, SMHIP_U16 = 7, SMHIP_U32 = 8, SMHIP_U64 = 9
} smhip_dtype;

/* ------------------------------------------------------------- context */
const char *smhip_version(void);
const char *smhip_last_error(void);
int smhip_device_count(int *count);
int smhip_set_device(int device);
int smhip_get_device(int *device);
/* Use the caller's hipStream_t (e.g. torch's current stream) for this thread;
 * NULL restores the library's own stream.  The new stream is ordered after the
 * thread's previous one (buffers used before the switch may be freed after it).
 * The stream must stay alive until the thread has switched away from it. */
int smhip_set_stream(void *hip_stream);
int smhip_get_stream(void **hip_stream);
int smhip_synchronize(void);

/* -------------------------------------------------------------- memory */
/* Operators on tiny arrays (<= 4096 results; + - * / of every element type, integer pow, smhip_fill, smhip_copy / _copy_strided,
 * smhip_upload of <= 256 bytes; operands of any strides or host-built;
 * the library's own queue) are RECORDED by smhip_elementwise / _inline / smhip_contiguous / smhip_array_scalar and go out several
 * to a launch: independent ones side by side, dependent ones in call order on one workgroup -- the launch goes out when 30 are
 * recorded, when a new one would tie two long lists of recorded ones together, and before ANY other call on the device touches
 * the stream (operators, copies, smhip_synchronize, events, smhip_get_stream): what a caller can observe is what call order
 * promises.  A buffer freed while a recorded operator refers to it returns to the pool after that launch.  Results are bit-identical
 * to the one-launch path (the same Op::apply).  SMHIP_TINY_BATCH=0 turns it off.  This reports, for the calling thread's device,
 * the launches made so far and the operators they carried (csrc/tiny.hip; replaces one launch per operator on the arrays of
 * benchmark/add.cpp:4-19 and benchmark/pow.cpp:5-28). */
int smhip_tiny_stats(unsigned long long *launches, unsigned long long *operators);

/* Replaces the per-operator `new T[n]` / `delete[]` (SMArray.h:219, :342-346)
 * with a pooled device allocator. */
int smhip_alloc(void **dptr, size_t bytes);
int smhip_free(void *dptr);
int smhip_pool_trim(void);
int smhip_pool_stats(size_t *bytes_in_use, size_t *bytes_cached);
int smhip_upload(void *dst, const void *src_host, size_t bytes);
int smhip_download(void *dst_host, const void *src, size_t bytes);
int smhip_copy(void *dst, const void *src, size_t bytes);
/* sm::ones / sm::zeros fill (UserFunctions.h:18-40) on device. */
int smhip_fill(int dtype, void *dst, const void *value_host, size_t n);
/* Synthetic input: element i = uniform[lo,hi) from a counter-based hash of
 * (seed, first + i); bit-identical to the CPU checker generator. */
int smhip_fill_uniform_f32(float *dst, size_t n, uint64_t seed, uint64_t first, float lo, float hi);

/* --------------------------------------------------------- shape layer */
/* sm::broadcast (SMUtils.h:34-99), host only.  Outputs hold max(nd1, nd2)
 * entries.  Returns the broadcast rank (>= 0) or SMHIP_ERR_BROADCAST. */
int smhip_broadcast(int nd1, const int64_t *shape1, const int64_t *strides1,
                    int nd2, const int64_t *shape2, const int64_t *strides2,
                    int64_t *result_shape, int64_t *new_strides1, int64_t *new_strides2,
                    int64_t *total_size);
/* is_contiguous (helpers.h:130-139): 1 / 0. */
int smhip_is_contiguous(int ndim, const int64_t *shape, const int64_t *strides);

/* ------------------------------------------------------------ hot path */
/* element_wise_op<T,Op> (calculate.h:5-99): out[linear] = a[sum idx_k*sa_k]
 * op b[sum idx_k*sb_k], idx from the row-major unravel of `linear` over
 * `shape`; `out` is dense row-major with prod(shape) elements.  Unlike
 * calculate.h:10 a 1-D call walks its strides, and
 * ndim > SMHIP_MAX_NDIM is rejected instead of overflowing (quirk 6).
 * `out` may BE an operand (same first element, the operand dense in the output's order: in-place a = a op b) -- every
 * element is then read before it is written and computed once; any other overlap of `out` with an operand is undefined,
 * as it is in the reference (calculate.h:96 writes result[linear] while other threads still read a and b). */
int smhip_elementwise(int op, int dtype, const void *a, const int64_t *stride_a,
                      const void *b, const int64_t *stride_b,
                      const int64_t *shape, int ndim, void *out);
/* element_wise_op for TINY operands that exist only in host memory (the reference's simple_check / BM_SMArrayPow_1D/2D
 * build 10-25 element arrays on the host and apply one operator, benchmark/add.cpp:4-19, benchmark/pow.cpp:5-28).
 * `a_host_bytes` > 0 says `a` is a HOST pointer to that many bytes (the operand's whole span, <= SMHIP_INLINE_MAX_BYTES),
 * which are copied into the kernel's argument block -- the launch packet carries the operand, so there is no upload
 * packet (~2.7 us each on this stack) and no device buffer for it; 0 says `a` is a device pointer as in
 * smhip_elementwise.  Likewise b; a scalar is an inline operand of one element with all strides 0.  At most
 * SMHIP_INLINE_MAX_OUTPUTS results, built-in Ops only (SMHIP_ERR_UNSUPPORTED otherwise: use smhip_elementwise). */
#define SMHIP_INLINE_MAX_BYTES 1024
#define SMHIP_INLINE_MAX_OUTPUTS 4096
int smhip_elementwise_inline(int op, int dtype, const void *a, size_t a_host_bytes, const int64_t *stride_a,
                             const void *b, size_t b_host_bytes, const int64_t *stride_b,
                             const int64_t *shape, int ndim, void *out);
/* A whole expression in one pass: out[i] = EXPR(a0[i], ..., a{k-1}[i]) over k <= 8 dense operands of n elements each,
 * EXPR a HIP expression in a0..a7 and the scalars s0..s3 (n_scalars <= 4 values of the element type in host memory, passed
 * at launch: changing them does not recompile), e.g. "(a0 + a1) * a2 - s0 * a3".  (k + 1) * sizeof(T) bytes per
 * element instead of 3 * sizeof(T) per operator of the chain it replaces (the reference evaluates such a chain as one
 * operator call and one temporary per step, SMArray.h:217-305).  Compiled by hipRTC on first use, cached like
 * smhip_register_op's kernels; each operation rounds as the separate operators do (no contraction). */
int smhip_fused_expr(const char *hip_expression, int dtype, const void *const *operands, int n_operands, const void *scalars_host,
                     int n_scalars, void *out, size_t n);
/* smhip_fused_expr over operands that BROADCAST against the result: operands[k] is operand k's first element and
 * strides[k * ndim .. ) its strides against `shape` (0 where it broadcasts, as smhip_broadcast returns them); `out` is dense
 * row-major over `shape`.  A row, a column / per-row value, a per-channel value or any operand periodic in the output is
 * read through the caches inside the one pass; a transposed or stepped view is copied dense first. */
int smhip_fused_expr_bcast(const char *hip_expression, int dtype, const void *const *operands, const int64_t *strides, int n_operands,
                           const void *scalars_host, int n_scalars, const int64_t *shape, int ndim, void *out);
/* The same expression with the sum of its results, in the same single pass: *sum_dev (device memory, fp64; integer types:
 * the exact 64-bit total as a double, like smhip_sum) = sum_i EXPR(...).  out_or_null: also store the elementwise result, or
 * reduce only -- e.g. "(a0 - a1) * (a0 - a1)" with out_or_null = NULL is a squared distance at 8 bytes per element and no
 * temporary.  Generalises smhip_contiguous_sum_async (BASELINE config 5's fused add + sum).  Asynchronous. */
int smhip_fused_expr_sum_async(const char *hip_expression, int dtype, const void *const *operands, int n_operands,
                               const void *scalars_host, int n_scalars, void *out_or_null, size_t n, double *sum_dev);
/* SMArray's element-copy assignment `dst_view = src` (SMArray.h:89-97, a host loop in the reference):
 * dst[sum idx_k*dst_strides_k] = src[sum idx_k*src_strides_k] over `shape`.  A source stride may be 0 (broadcast);
 * a destination stride may not (for extents > 1).  Source and destination must not overlap. */
int smhip_copy_strided(int dtype, const void *src, const int64_t *src_strides, void *dst, const int64_t *dst_strides,
                       const int64_t *shape, int ndim);
/* handle_contiguous_arrays<T,Op> (calculate.h:101-134): out[i] = a[i] op b[i]. */
int smhip_contiguous(int op, int dtype, const void *a, const void *b, void *out, size_t n);
/* array_scalar_op<T,Op> (calculate.h:137-169): out[i] = a[i] op value; `value_host`
 * points at one T in host memory.  Also sm::pow (UserFunctions.h:42-48). */
int smhip_array_scalar(int op, int dtype, const void *a, const void *value_host, size_t n, void *out);
/* dot_product<T> (product.h:8-224 via SMArray.h:213-215): one T to *out_host.
 * f32/f64 accumulate in fp64 (the reference's f32 lane accumulators saturate,
 * SURVEY section 0); integer types -- i32/i64 and SMHIP_I8 ... SMHIP_U64, the generic
 * template's `T sum; sum += a[i] * b[i]` -- wrap exactly like the reference (the sum of
 * products modulo 2^(8 sizeof T)). Synchronous. */
int smhip_dot(int dtype, const void *a, const void *b, size_t n, void *out_host);
/* dot_product<std::complex<double>> (product.h:168-224): a, b hold n {re, im} pairs of doubles (16-byte
 * aligned); writes {re, im} of sum a[i]*b[i] (unconjugated, as the reference's scalar tail computes
 * it) to out2_host.  Synchronous. */
int smhip_dot_c64(const void *a, const void *b, size_t n, double *out2_host);
/* The generic dot_product<T> (product.h:8-20) with T = std::complex<float>: a, b hold n {re, im} pairs of floats (8-byte
 * aligned -- the kernel's 16-byte loads are legal at any element-aligned address on gfx950 and run within 2 % of aligned
 * ones, profiles/r01_sweep_unaligned.txt, so a view that starts on an odd complex element needs no peeling);
 * {re, im} of sum a[i]*b[i] to out2_host.  Accumulated in fp64 (the reference adds in float, sequentially), rounded
 * to float once.  Synchronous; the _async form leaves the fp64 {re, im} in device memory (out2_dev: 2 doubles). */
int smhip_dot_c32(const void *a, const void *b, size_t n, float *out2_host);
int smhip_dot_c32_async(const void *a, const void *b, size_t n, double *out2_dev);
/* Whole-array sum in fp64 (no reference counterpart; BASELINE config 5). Synchronous. */
int smhip_sum(int dtype, const void *a, size_t n, double *out_host);
/* Asynchronous forms leaving the fp64 result in device memory so a multi-GPU
 * caller can all-reduce it (RCCL) without a host round trip. */
int smhip_sum_async(int dtype, const void *a, size_t n, double *out_dev);
int smhip_dot_async(int dtype, const void *a, const void *b, size_t n, double *out_dev);
/* smhip_dot_c64 leaving {re, im} in device memory (out2_dev: 2 doubles). */
int smhip_dot_c64_async(const void *a, const void *b, size_t n, double *out2_dev);
/* Fused out = a op b and *sum_dev = sum(out) in one pass (config 5: 12 B/elem). */
int smhip_contiguous_sum_async(int op, int dtype, const void *a, const void *b, void *out, size_t n,
                               double *sum_dev);

/* Fusion hook: out[i] = (a[i] op1 b[i]) op2 c[i] in ONE pass over dense arrays; pass c = NULL
 * and c_scalar_host -> one T for (a op1 b) op2 scalar.  op1, op2 in {ADD, SUB, MUL, DIV}; each
 * stage rounds as the separate Ops do, so the result is bit-identical to two operator calls
 * (SMArray.h:217-305 x 2) at 2/3 (array c) or 1/2 (scalar c) of their HBM traffic. */
int smhip_fused_contiguous(int op1, int op2, int dtype, const void *a, const void *b, const void *c,
                           const void *c_scalar_host, void *out, size_t n);

/* A CHAIN of broadcasted operators in as few passes as possible -- one, when every operand is dense, a row, a column /
 * per-row value, periodic in the output, or a scalar:
 *     r = x[0];   r = r ops[k] x[k+1]   (swapped[k] != 0:  r = x[k+1] ops[k] r),   k = 0 .. n_operands - 2;   out = r
 * What the reference evaluates as n_operands - 1 operator calls with a fresh temporary each (SMArray.h:217-305, `new T[n]`
 * at :219; element_wise_op / array_scalar_op, calculate.h:5-169) -- `(A * row + B) * 0.5f` is 28 bytes per f32 element
 * there and 12 here.  operands[k] is a device pointer to operand k's first element with strides[k * ndim .. ) its strides
 * broadcast against `shape` (0 where it broadcasts; as smhip_broadcast returns them), or NULL for a scalar whose value is
 * element k of scalars_host (n_operands elements of the element type in host memory; entries of array operands are
 * ignored).  operands[0] must be an array.  ops[k] in {ADD, SUB, MUL, DIV}, or POW with a SCALAR x[k+1] and swapped[k] == 0
 * (sm::pow(<expression>, s), UserFunctions.h:42-48: ^2 is a stage of the one-pass kernel, any other exponent cuts the chain and runs
 * smhip_array_scalar's evaluation on the value so far -- the same bits either way).  `out` is dense row-major over `shape` and
 * must not overlap an operand.  Each stage is the single rounded / wrapping operation the separate operator performs, so
 * the result is bit-identical to the operator chain.  An operand the one-pass kernel has no index form for (a transposed
 * or stepped view) cuts the chain: that operator runs through smhip_elementwise's kernels, the rest stays fused. */
#define SMHIP_CHAIN_MAX_OPERANDS 16
int smhip_chain(int dtype, int n_operands, const void *const *operands, const int64_t *strides, const void *scalars_host,
                const int *ops, const int *swapped, const int64_t *shape, int ndim, void *out);

/* The SUM of a chain's value -- sm::pow(a - b, 2.0f).sum(), a squared error -- without writing the value: the arguments of
 * smhip_chain minus `out`; accumulation as smhip_sum (fp64 for the float types, wrapping 64-bit for the integer types, each
 * element first rounded to the element type stage by stage as the operators round it).  One pass over the operands when
 * every one of them is dense or a scalar (8 bytes per f32 element for two operands, against 12 + 4 for the chain and the sum
 * of its result); otherwise the chain into a pooled temporary and smhip_sum of that.  The _async form leaves the fp64 result in
 * device memory (stream-ordered, like smhip_sum_async); the other waits and returns it. */
int smhip_chain_sum_async(int dtype, int n_operands, const void *const *operands, const int64_t *strides, const void *scalars_host,
                          const int *ops, const int *swapped, const int64_t *shape, int ndim, double *sum_dev);
int smhip_chain_sum(int dtype, int n_operands, const void *const *operands, const int64_t *strides, const void *scalars_host, const int *ops,
                    const int *swapped, const int64_t *shape, int ndim, double *sum_host);

/* ----------------------------------------------------------- multi-GPU */
/* The reference's only fan-out is the OpenMP `parallel for` over chunks of the output (calculate.h:47, :152).  Its
 * MI355X counterpart is the RESULT's outermost dimension cut into one block per GPU of the node: elementwise blocks
 * are independent (no data-path collective at all); a whole-array reduction ends in ONE RCCL all-reduce of an 8-byte
 * scalar over xGMI.  Two ways to drive it:
 *   (1) one process, n devices:  smhip_set_devices(n) makes devices 0..n-1 a group (one library stream per device,
 *       communicators from ncclCommInitAll) and the smhip_sharded_* entry points take PER-DEVICE POINTER TABLES
 *       (entry g = a buffer on device g, e.g. from smhip_set_device(g) + smhip_alloc).  This is what sm::set_devices /
 *       sm::Sharded<T> (include/Sharded.h) use.
 *   (2) one process per GPU (torchrun-style):  rank 0 calls smhip_comm_unique_id, every rank smhip_comm_init_rank
 *       with its device selected; each rank then runs the ordinary single-device entry points on its shard and
 *       smhip_allreduce_sum_async on its partial.
 * RCCL (librccl.so.1) is loaded when one of these is first called; a process that never does needs no RCCL. */

/* Near-equal contiguous blocks of `n` items over `world` ranks: the first n % world ranks get one extra. Host only. */
int smhip_split_range(int64_t n, int world, int rank, int64_t *start, int64_t *count);
/* Rank `rank`'s block of the broadcast problem (shape, stride_a, stride_b as smhip_broadcast returns them), cut along
 * dim 0: its result shape, the element offsets of its block in a, b and the dense result, and whether an operand is
 * broadcast along dim 0 and therefore needed whole on every device (bit 0: a, bit 1: b).  Host only. */
int smhip_shard_outer(const int64_t *shape, const int64_t *stride_a, const int64_t *stride_b, int ndim, int world, int rank,
                      int64_t *shard_shape, int64_t *offset_a, int64_t *offset_b, int64_t *offset_out, int *replicated_mask);

/* Devices 0..n-1 become the group (n >= 1; n = 1 still goes through RCCL with a one-rank communicator).  Replaces an
 * earlier group; n = 0 dissolves it.  Not to be called while sharded work of another thread is in flight. */
int smhip_set_devices(int n);
int smhip_get_devices(int *n);
/* Blocks until every device of the group has finished its queued work. */
int smhip_sharded_synchronize(void);
/* What RCCL ITSELF reports for the group's communicator on device `index` (ncclCommCount / ncclCommUserRank /
 * ncclCommCuDevice), so a caller -- bench.py's "rccl" field -- can check that the collective library saw as many ranks
 * as it was asked for.  Any output may be NULL. */
int smhip_group_info(int index, int *nranks, int *rank, int *device);
/* ncclGetVersion (e.g. 22203); loads RCCL. */
int smhip_rccl_version(int *version);
/* `bytes` from device src_device's memory to device dst_device's, device to device (hipMemcpyPeerAsync, over xGMI when the
 * pair has peer access, which is enabled on first use).  Asynchronous and stream-ordered on BOTH sides: the copy waits for
 * what is queued on the source device's library stream, runs on the destination device's library stream, and the source's
 * stream continues after it.  How sm::Sharded<T>::scatter / replicate / gather move device-resident arrays (the
 * reference has no counterpart: its threads share one address space, calculate.h:47). */
int smhip_copy_peer(void *dst, int dst_device, const void *src, int src_device, size_t bytes);

/* handle_contiguous_arrays per device: out[g][i] = a[g][i] op b[g][i], i < n[g].  Asynchronous, no collective. */
int smhip_sharded_contiguous(int op, int dtype, const void *const *a, const void *const *b, void *const *out, const size_t *n);
/* array_scalar_op per device. */
int smhip_sharded_array_scalar(int op, int dtype, const void *const *a, const void *value_host, const size_t *n, void *const *out);
/* element_wise_op over the group: `shape` is the GLOBAL result shape; device g computes block g of dim 0
 * (smhip_shard_outer).  a[g] / b[g] point at the FIRST ELEMENT OF DEVICE g's BLOCK of that operand when its dim-0
 * stride is non-zero (the operand is sharded like the result), or at device g's full copy when the stride is 0 (the
 * operand is broadcast along dim 0 and replicated, e.g. BASELINE config 3's 16 KiB row); the inner strides are the
 * same on every device.  out[g] is device g's dense block.  Asynchronous, no collective. */
int smhip_sharded_elementwise(int op, int dtype, const void *const *a, const int64_t *stride_a, const void *const *b,
                              const int64_t *stride_b, const int64_t *shape, int ndim, void *const *out);
/* BASELINE config 5: out[g] = a[g] op b[g] and the sum of ALL results, fused per device (12 B/elem for f32), then ONE
 * ncclAllReduce(1 x fp64, sum) inside ncclGroupStart/End; the total is written to *sum_host (every device also holds
 * it).  Synchronous for the scalar only: the other devices' streams are not waited for. */
int smhip_sharded_contiguous_sum(int op, int dtype, const void *const *a, const void *const *b, void *const *out, const size_t *n,
                                 double *sum_host);
/* Whole-array sum / dot of a sharded array: per-device partial + one all-reduce.  dot: the element type's value to
 * *out_host (f32/f64 via fp64 partials; i32/i64 wrap exactly like the single-device and the reference's result). */
int smhip_sharded_sum(int dtype, const void *const *a, const size_t *n, double *sum_host);
int smhip_sharded_dot(int dtype, const void *const *a, const void *const *b, const size_t *n, void *out_host);

/* One process per GPU.  id128: NCCL_UNIQUE_ID_BYTES (128) bytes, produced on one rank and carried to the others by
 * the launcher's own channel (bench.py: torch.distributed broadcast; MPI_Bcast; a file).  smhip_comm_init_rank binds
 * the communicator to the calling thread's current device and is collective over the ranks. */
int smhip_comm_unique_id(void *id128);
int smhip_comm_init_rank(int nranks, int rank, const void *id128);
/* The communicator as RCCL reports it (ncclCommCount / ncclCommUserRank); 0 / -1 when there is none. */
int smhip_comm_info(int *nranks, int *rank);
int smhip_comm_destroy(void);
/* In-place sum over the ranks of `count` values in device memory, on the calling thread's stream (so it is ordered
 * after the kernel that produced them): SMHIP_F32 / SMHIP_F64 as floats, SMHIP_I32 / SMHIP_I64 wrapping (modulo 2^32 /
 * 2^64).  Asynchronous. */
int smhip_allreduce_sum_async(int dtype, void *inout_dev, size_t count);

/* --------------------------------------------------------- diagnostics */
/* The stream-policy word the library would give a launch that reads [a, a + a_bytes) and [b, b + b_bytes) (either may be
 * NULL / 0) and writes [out, out + out_bytes) on the calling thread's device: bit 0 = its reads carry the non-temporal
 * hint, bit 1 = its results are stored to stay in the Infinity Cache (read side, write side, cold
 * operands).  Like a launch it records the spans as touched.  Host-only arithmetic on the pointer VALUES (nothing is
 * dereferenced, no device is needed): the residency rule's test hook. */
int smhip_policy_probe(const void *a, size_t a_bytes, const void *b, size_t b_bytes, const void *out, size_t out_bytes, int *policy);
/* The same answer WITHOUT recording the spans as touched: looking does not change what the next launch is told. */
int smhip_policy_peek(const void *a, size_t a_bytes, const void *b, size_t b_bytes, const void *out, size_t out_bytes, int *policy);
/* For the test-suite: honour the failure-injection variables SMHIP_TEST_FAIL_COMM_INIT / SMHIP_TEST_FAIL_ALLREDUCE (the
 * group's error paths on a one-GPU box).  Off by default: the variables alone do nothing. */
int smhip_enable_test_hooks(int on);
/* The calling thread's device runs the library's operators on two hardware queues (independent operators overlap their
 * tails and heads; an operator that depends on another's result, or overwrites what another still reads, is ordered behind
 * it by queue order or an event edge -- the reference's threads likewise run whatever chunk is ready, calculate.h:47).
 * *queues: 2, or 1 once the second queue is off for the device (SMHIP_QUEUES=1, smhip_get_stream handed the stream out, the
 * device group is in use); *alternations: operators that took "the other" queue because they depended on nothing
 * unfinished; *edges: cross-queue event edges issued so far.  Any output may be NULL. */
int smhip_queue_stats(int *queues, unsigned long long *alternations, unsigned long long *edges);
/* How a dense streaming operator over operands of `bytes_per_operand` bytes is launched: *pieces = the number of kernel
 * launches it goes out as.  `streams` = the full-size streams it moves: 3 for smhip_contiguous and the fused op+sum (two
 * reads, one write), 2 for smhip_array_scalar and smhip_dot, 1 for smhip_sum.  Three-stream operators are cut above
 * 512 MiB per operand, the others above 2 GiB (DESIGN.md section 3 "Very large arrays").  Host-only.  bench.py prices its
 * roofline per LAUNCH with it. */
int smhip_launch_pieces(size_t bytes_per_operand, int streams, int *pieces);

/* -------------------------------------------------------------- timing */
/* HIP events on the calling thread's stream (what bench.py brackets the
 * timed region with). */
int smhip_event_create(void **event);
int smhip_event_record(void *event);
int smhip_event_synchronize(void *event);
int smhip_event_elapsed_ms(void *start, void *stop, float *ms);
int smhip_event_destroy(void *event);

#ifdef __cplusplus
}
#endif
#endif /* SMHIP_H */
