// internal.h -- shared plumbing of libsmhip.so (not installed).
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include <initializer_list>

#include "smhip.h"

namespace smhip {

// Records the message for smhip_last_error() on this thread; returns `code`.
int fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));

// Makes sure this thread has a usable gfx950 device selected and returns the
// stream its work goes to.  SMHIP_OK or SMHIP_ERR_NO_DEVICE / SMHIP_ERR_HIP.
// Outside an operator's scope (OpScope below) it also switches the device to ONE library queue: a caller that asks for "the"
// stream -- the sharded entry points -- relies on stream order.
int acquire(hipStream_t *stream);
// The same without touching the queues: the caller's stream, or the device's first library queue.
int acquire_stream(hipStream_t *stream);
// tiny.hip: operators on tiny arrays are recorded and go out several to a launch; acquire_stream() flushes them first.
constexpr int64_t kTinyMaxResults = 4096;  // operators with at most this many results may be recorded
bool tiny_any_recorded();
int tiny_flush_device(int dev);
bool tiny_defer_free(int dev, void *p, size_t bytes);
bool tiny_context(int *dev);  // runtime.hip: may the calling thread record a tiny operator now?
int tiny_try_enqueue(int op, int dtype, const void *a, size_t a_host_bytes, const int64_t *sa, const void *b, size_t b_host_bytes,
                     const int64_t *sb, const int64_t *shape, int ndim, const void *scalar_host, void *out, bool *taken, int kind = 0);
void tiny_stats(int dev, unsigned long long *launches, unsigned long long *operators);
// Second library queue of `dev` off (true) / on again (false); see runtime.hip "two queues per device".
void dispatch_single_queue(int dev, bool single);

struct Span { const void *p; size_t bytes; };
// One operator of the C ABI: which library queue it runs on and what it is ordered behind (runtime.hip: two queues per
// device).  begin() with the operator's operand spans lets independent operators overlap; begin_barrier() is for entry
// points whose spans are not declared: ordered behind everything, everything later behind them.  The scope keeps the
// device's dispatcher locked until the operator's launches are queued.
class OpScope {
public:
    OpScope() = default;
    OpScope(const OpScope &) = delete;
    OpScope &operator=(const OpScope &) = delete;
    ~OpScope();
    int begin(const Span *reads, size_t n_reads, Span write, hipStream_t *stream, bool barrier = false);
    int begin_barrier(hipStream_t *stream);
private:
    void *locked_ = nullptr;
};

// A recycled (or new) timing-disabled event of device `dev` (nullptr if none can be made) / its return to the pool.
hipEvent_t pool_event_take(int dev);
void pool_event_give(int dev, hipEvent_t e);

// Compute units of the calling thread's current device (256 on MI355X); valid after acquire().
int compute_units();

// Device the calling thread's work goes to.
int current_device();

// Runs the enclosed calls on device `device`'s library stream (whatever stream the thread brought), then puts the
// thread back: how the sharded entry points walk the device group from one host thread.
class ThreadDeviceScope {
public:
    explicit ThreadDeviceScope(int device);
    ~ThreadDeviceScope();
    ThreadDeviceScope(const ThreadDeviceScope &) = delete;
    ThreadDeviceScope &operator=(const ThreadDeviceScope &) = delete;
private:
    int prev_device_;
    bool prev_use_user_;
    void *bridged_ = nullptr;  // the library stream that was ordered after the caller's own stream on entry (and back on exit)
};

// The partials of ONE reduction call: `count` doubles from the pool, handed back when the lease ends.  The pool's
// stream-ordered reuse makes that safe while the kernels are still queued, and two reductions queued on different
// streams never share a buffer.
class ScratchLease {
public:
    ScratchLease() = default;
    ScratchLease(const ScratchLease &) = delete;
    ScratchLease &operator=(const ScratchLease &) = delete;
    ~ScratchLease() { if (p_) smhip_free(p_); }
    int take(size_t count, double **ptr) {
        if (int rc = smhip_alloc(&p_, (count ? count : 1) * sizeof(double))) return rc;
        *ptr = static_cast<double *>(p_);
        return SMHIP_OK;
    }
private:
    void *p_ = nullptr;
};
int reduce_finish(int dtype, void *partials, size_t blocks, double *out8, hipStream_t s);
constexpr int kReduceFoldSpan = 1024;  // = reduce.hip's kGroupTarget: partial buffers need blocks + blocks / span + 2 slots

#define SMHIP_TRY(expr)                                                                        \
    do {                                                                                       \
        hipError_t smhip_e_ = (expr);                                                          \
        if (smhip_e_ != hipSuccess)                                                            \
            return ::smhip::fail(SMHIP_ERR_HIP, "%s: %s", #expr, hipGetErrorString(smhip_e_)); \
    } while (0)

#define SMHIP_LAUNCH_CHECK(what)                                                              \
    do {                                                                                       \
        hipError_t smhip_e_ = hipGetLastError();                                               \
        if (smhip_e_ != hipSuccess)                                                            \
            return ::smhip::fail(SMHIP_ERR_HIP, "launch %s: %s", what, hipGetErrorString(smhip_e_)); \
    } while (0)

// Read streams of a launch get the non-temporal hint when together they exceed the Infinity Cache (ops.hip.h: load_stream_if).
constexpr size_t kInfinityCacheBytes = (size_t)256 << 20;
inline int stream_reads(size_t bytes_read) { return bytes_read > kInfinityCacheBytes ? 1 : 0; }
// ... and results are stored so that they stay in it (ops.hip.h: store_stream_if, an `sc1` store; bit 1 of the same
// word) when the launch's whole footprint fits: the next operator then finds them there.  Small footprints keep the
// non-temporal store: at 16 and 32 MiB (1R+1W of 8 / 16 MiB per array -- about what the eight 4 MiB L2s hold) it was the
// faster one in every setting of tools/sweep_store_sc1.hip; from 48 MiB (2R+1W of 16 MiB) on the `sc1` store wins.
// SMHIP_STORE_POLICY=nt switches the rule off (experiments).
constexpr size_t kStoreKeepFloor = (size_t)40 << 20;
constexpr int kPolicyLoadNt = 1, kPolicyStoreKeep = 2;  // the word's bits; ops.hip.h (dev::kLoadNt / kStoreKeep) and jit.hip's preludes mirror them
int stream_policy(size_t bytes_read, size_t bytes_written);
// The same word for a launch whose operands are known: reads of spans that the library has NOT touched recently --
// nothing of theirs can be in the Infinity Cache -- get the non-temporal hint at any size.  Round 2's rule (plain loads
// whenever a launch reads <= 256 MiB) is tuned to operands that repeat or were just produced; on cold operands plain loads
// cost 3-12 % (tools/sweep_cold.hip -> profiles/r03_sweep_cold.txt: 2R+1W at 64 MiB per array 35.7 us plain, 32.5 us nt).
// The library cannot see the cache, but it knows what it launched: a per-device ring of the spans its recent launches read
// and wrote, each stamped with the device's running byte count (runtime.hip: residency tracker).  A span is WARM when a
// launch touched it within the last kWarmWindow bytes of library traffic on that device.  Also records this launch's
// touches.  A hint only: a wrong guess costs a few percent, never a wrong result.  SMHIP_RESIDENCY=off restores round 2's rule.
constexpr size_t kWarmWindow = (size_t)192 << 20, kTrackFloor = (size_t)2 << 20;
int stream_policy(std::initializer_list<Span> reads, Span write);
// The same refinement for a policy word that was planned from sizes alone (broadcast.hip's plans are shared with the
// run-time compiled kernels and know no pointers): ORs in the non-temporal read hint for cold operands, records the touches.
int refine_policy(int policy, std::initializer_list<Span> reads, Span write);
int refine_policy(int policy, const Span *reads, size_t n_reads, Span write);
// [p, p + bytes) on device `dev` no longer holds what a library kernel left there (freed, uploaded, peer-copied into)
void residency_forget(int dev, const void *p, size_t bytes);

// Large operands go out as several launches (contiguous.hip explains why).  piece_for(n_vec, streams) = the piece size in
// 16-byte vectors for operands of n_vec of them, 0 for "one launch"; `streams` = the full-size streams the kernel moves
// (3: a op b -> out, the fused op+sum; 2: a op s -> out, dot; 1: sum).
//   three streams   n_vec <= 2^25 (512 MiB)        one launch (256 MiB: 84.0 % whole, 82.5 % halved; 512 MiB: 82.8 / 82.6 %)
//                   2^25 < n_vec <= 2^26 (1 GiB)   halves of 2^25: the headline's 2^28 floats as TWO launches, 496.6 -> 491.8 us
//                                                  (81.1 -> 81.9 %, five alternating rounds each within 0.1 %; four pieces 81.6, eight 80.7 %)
//                   larger                         pieces of 2^24 (256 MiB): 2^29 80.0 -> 82.6 %, 2^30 78.4 -> 81.5 %, 2^31 76-81 -> 82-83 %
//   one / two       n_vec <= 2^27 (2 GiB)          one launch (pieces change nothing up to there: a*s 81.4 / 81.5 %, sum 83.2 / 83.4 %,
//                                                  dot 83.3 / 82.0 % whole / in pieces at 2^28-2^29 elements)
//                   larger                         pieces of 2^25: a*s 80.5 -> 81.2 %, sum 82.1 -> 83.7 %, dot 80.9 -> 82.7 % at 2^30
//   (profiles/r03_headline_pieces.txt, r03_mid_pieces.txt, r03_big_add.txt, r03_mix_pieces.txt)
// SMHIP_PIECE_LOG2VEC=<k>: every operand above 2^k vectors in pieces of 2^k, whatever the mix (tests run the piece loops at
// small sizes with it); 0: never split.
size_t piece_for(size_t n_vec, int streams);

inline size_t dtype_size(int dtype) {
    switch (dtype) {
        case SMHIP_F64: case SMHIP_I64: case SMHIP_U64: return 8;
        case SMHIP_I8: case SMHIP_U8: return 1;
        case SMHIP_I16: case SMHIP_U16: return 2;
        default: return 4;
    }
}
inline bool valid_dtype(int dtype) { return dtype >= SMHIP_F32 && dtype <= SMHIP_I64; }
inline bool valid_dot_dtype(int dtype) { return dtype >= SMHIP_F32 && dtype <= SMHIP_U64; }  // + the generic dot's integer types
inline bool valid_op(int op) { return op >= SMHIP_OP_ADD && op <= SMHIP_OP_LEFT; }

// Kernel launchers (one translation unit each).
int launch_contiguous(int op, int dtype, const void *a, const void *b, void *out, size_t n, hipStream_t s);
int launch_array_scalar(int op, int dtype, const void *a, const void *value_host, size_t n, void *out, hipStream_t s);
// a dense (rows x cols) a against one row (b_is_row) or one column of b; cols a multiple of the 16-byte vector width,
// rows * cols / width < 2^32
int launch_flat_rows(int op, int dtype, const void *a, const void *b, void *out, size_t rows, size_t cols, bool b_is_row, hipStream_t s);
// b is a device pointer to ONE element (a fully broadcast operand); `swapped`
// computes value op a[i] instead of a[i] op value.
int launch_array_devscalar(int op, int dtype, const void *a, const void *value_dev, size_t n, void *out,
                           bool swapped, hipStream_t s);
int launch_broadcast(int op, int dtype, const void *a, const int64_t *sa, const void *b, const int64_t *sb,
                     const int64_t *shape, int ndim, void *out, hipStream_t s);
int launch_inline(int op, int dtype, const void *a, size_t a_host_bytes, const int64_t *sa, const void *b, size_t b_host_bytes,
                  const int64_t *sb, const int64_t *shape, int ndim, void *out, hipStream_t s);
int launch_copy_strided(int dtype, const void *src, const int64_t *src_strides, void *dst, const int64_t *dst_strides,
                        const int64_t *shape, int ndim, hipStream_t s);
// run-time compiled user Ops (jit.hip)
int jit_register(const char *expr, int *op_id);
int jit_fused_expr(const char *expr, int dtype, const void *const *operands, int n_operands, const void *scalars_host, int n_scalars,
                   void *out, size_t n, double *sum_dev, hipStream_t s);
int jit_contiguous(int op, int dtype, const void *a, const void *b, void *out, size_t n, hipStream_t s);
int jit_array_scalar(int op, int dtype, const void *a, const void *value_host, size_t n, void *out, hipStream_t s);
inline bool user_op(int op) { return op >= SMHIP_OP_USER_BASE; }
int launch_fused(int op1, int op2, int dtype, const void *a, const void *b, const void *c, const void *c_scalar_host, void *out,
                 size_t n, hipStream_t s);
// chain.hip: r = x0; r = r op[k] x[k+1] (swapped[k]: x[k+1] op[k] r) over broadcast operands, as few passes as possible
int launch_chain(int dtype, int n_operands, const void *const *operands, const int64_t *strides, const void *scalars_host, const int *ops,
                 const int *swapped, const int64_t *shape, int ndim, void *out, hipStream_t s);
int launch_chain_sum(int dtype, int n_operands, const void *const *operands, const int64_t *strides, const void *scalars_host, const int *ops,
                     const int *swapped, const int64_t *shape, int ndim, double *sum_dev, hipStream_t s);
// An operand of smhip_fused_expr_bcast in the index form chain.hip found for it: kind 0 dense a[i], 1 row a[i mod P] (P a
// whole number of vectors), 2 splat a[(i / R) mod C]
struct ExprLeaf { int kind; const void *ptr; uint64_t P, R, C; };
int launch_expr_bcast(const char *expr, int dtype, const void *const *operands, const int64_t *strides, int n_operands,
                      const void *scalars_host, int n_scalars, const int64_t *shape, int ndim, void *out, hipStream_t s);
int jit_fused_expr_bcast(const char *expr, int dtype, const ExprLeaf *leaves, int n_operands, const void *scalars_host, int n_scalars,
                         void *out, size_t n, hipStream_t s);
int launch_fill(int dtype, void *dst, const void *value_host, size_t n, hipStream_t s);
int launch_fill_uniform_f32(float *dst, size_t n, uint64_t seed, uint64_t first, float lo, float hi, hipStream_t s);
int launch_sum(int dtype, const void *a, size_t n, double *out_dev, hipStream_t s);
int launch_dot(int dtype, const void *a, const void *b, size_t n, double *out_dev, void *out_native_dev, hipStream_t s);
int launch_cdot(const void *a, const void *b, size_t n, double *out2_dev, hipStream_t s);
int launch_cdot32(const void *a, const void *b, size_t n, double *out2_dev, hipStream_t s);  // n {re, im} float pairs; fp64 {re, im} out
int launch_contiguous_sum(int op, int dtype, const void *a, const void *b, void *out, size_t n, double *sum_dev,
                          hipStream_t s);

}  // namespace smhip
