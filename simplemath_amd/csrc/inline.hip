// inline.hip -- tiny operands that ride in the kernel's argument block.
//
// The reference's smallest benchmarks build a 5 x 5 or a 10-element array on the host and apply one operator
// (benchmark/add.cpp:4-19 simple_check, benchmark/pow.cpp:5-28 BM_SMArrayPow_1D/2D): ~0.3-2.6 us on a CPU.  On the GPU
// every DEPENDENT packet on a stream -- a copy or a kernel -- costs ~2.7 us (profiles/r01_small_array_breakdown.txt), and
// round 1 spent two per operator on such arrays: the upload of the host-built operand, then the kernel.  Here an operand
// of <= 1 KiB that exists only on the host is copied into the kernel's ARGUMENT BLOCK (the launch packet carries it),
// and the kernel reads it from the kernarg segment: one packet, no device buffer for the operand at all.
// One output per lane, plain div/mod unravel: at <= 4096 outputs nothing else matters.
#include <string.h>

#include "internal.h"
#include "ops.hip.h"

namespace smhip {
namespace {

using namespace dev;

constexpr size_t kInlineBytes = SMHIP_INLINE_MAX_BYTES;

// CAP bytes per operand.  Two capacities: the launch writes the whole argument block into the kernarg ring -- device
// memory, over the bus -- so a 2 KiB block costs a 100-byte operand about a microsecond it does not need
// (simple_check 4.3-4.7 us with the 1 KiB slots alone, profiles/r02_cpp_benchmarks.txt).
template <int CAP> struct alignas(16) InlineBlock {  // FIRST kernel argument: it sits at offset 0 of the kernarg segment
    unsigned char a[CAP], b[CAP];
};
#ifndef SMHIP_INLINE_SMALL_CAP
#define SMHIP_INLINE_SMALL_CAP 128
#endif
constexpr int kSmallCap = SMHIP_INLINE_SMALL_CAP;
struct InlineParams {
    int64_t sa[SMHIP_MAX_NDIM], sb[SMHIP_MAX_NDIM];  // innermost first
    uint32_t shape[SMHIP_MAX_NDIM];                   // innermost first
    int ndim;
    uint32_t n;
    uint32_t a_inline, b_inline;
};

template <typename T, typename Op, int CAP>
__global__ __launch_bounds__(256) void inline_kernel(InlineBlock<CAP> blk, const T *__restrict__ a_dev, const T *__restrict__ b_dev,
                                                     T *__restrict__ out, InlineParams p) {
    (void)blk;  // read through the kernarg pointer: indexing the by-value copy per lane would spill it to scratch
    const char *kernarg = (const char *)__builtin_amdgcn_kernarg_segment_ptr();
    const T *a = p.a_inline ? reinterpret_cast<const T *>(kernarg) : a_dev;
    const T *b = p.b_inline ? reinterpret_cast<const T *>(kernarg + CAP) : b_dev;
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= p.n) return;
    uint32_t rem = i;
    int64_t offA = 0, offB = 0;
    for (int d = 0; d < p.ndim; ++d) {
        const uint32_t idx = d == p.ndim - 1 ? rem : rem % p.shape[d];
        rem /= p.shape[d];
        offA += (int64_t)idx * p.sa[d];
        offB += (int64_t)idx * p.sb[d];
    }
    out[i] = Op::apply(a[offA], b[offB]);
}

// The common shapes -- every operand either dense in output order or a single value -- with WHERE each operand lives as a
// template parameter: the inline operands' addresses are then compile-time offsets into the kernarg segment, so their
// loads are issued at once, next to (not behind) the scalar loads of the other arguments.  The kernarg segment is host
// memory: each dependent access is a trip over the fabric (~1.3 us), and the generic kernel above makes two in a row.
// A_INL: a rides in the block.  B_MODE: 0 device array, 1 inline array, 2 inline scalar (element 0).
template <typename T, typename Op, bool A_INL, int B_MODE, int CAP>
__global__ __launch_bounds__(256) void inline_dense_kernel(InlineBlock<CAP> blk, const T *__restrict__ a_dev, const T *__restrict__ b_dev,
                                                           T *__restrict__ out, uint32_t n) {
    (void)blk;
    const char *kernarg = (const char *)__builtin_amdgcn_kernarg_segment_ptr();
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const T x = A_INL ? reinterpret_cast<const T *>(kernarg)[i] : a_dev[i];
    const T y = B_MODE == 2 ? reinterpret_cast<const T *>(kernarg + CAP)[0]
                            : (B_MODE == 1 ? reinterpret_cast<const T *>(kernarg + CAP)[i] : b_dev[i]);
    out[i] = Op::apply(x, y);
}

template <typename T, typename Op, int CAP>
int run_inline(const void *a, size_t a_host_bytes, const void *b, size_t b_host_bytes, void *out, const InlineParams &p, hipStream_t s) {
    InlineBlock<CAP> blk;
    if (a_host_bytes) memcpy(blk.a, a, a_host_bytes);
    if (b_host_bytes) memcpy(blk.b, b, b_host_bytes);
    // dense in output order <=> strides are the running products of the extents; a single value <=> all strides 0
    bool a_dense = true, b_dense = true, b_scalar = true;
    int64_t run = 1;
    for (int d = 0; d < p.ndim; ++d) {
        if (p.shape[d] > 1) {
            a_dense &= p.sa[d] == run;
            b_dense &= p.sb[d] == run;
            b_scalar &= p.sb[d] == 0;
        }
        run *= p.shape[d];
    }
    const dim3 grid((p.n + 255) / 256), block(256);
    const T *ad = static_cast<const T *>(a), *bd = static_cast<const T *>(b);
    T *od = static_cast<T *>(out);
    if (a_dense && (b_dense || (b_scalar && p.b_inline))) {
        const int mode = (p.a_inline ? 3 : 0) + (b_scalar && p.b_inline ? 2 : (p.b_inline ? 1 : 0));
        switch (mode) {
            case 1: hipLaunchKernelGGL((inline_dense_kernel<T, Op, false, 1, CAP>), grid, block, 0, s, blk, ad, bd, od, p.n); break;
            case 2: hipLaunchKernelGGL((inline_dense_kernel<T, Op, false, 2, CAP>), grid, block, 0, s, blk, ad, bd, od, p.n); break;
            case 3: hipLaunchKernelGGL((inline_dense_kernel<T, Op, true, 0, CAP>), grid, block, 0, s, blk, ad, bd, od, p.n); break;
            case 4: hipLaunchKernelGGL((inline_dense_kernel<T, Op, true, 1, CAP>), grid, block, 0, s, blk, ad, bd, od, p.n); break;
            case 5: hipLaunchKernelGGL((inline_dense_kernel<T, Op, true, 2, CAP>), grid, block, 0, s, blk, ad, bd, od, p.n); break;
            default: goto generic;  // nothing inline: the caller should have used smhip_elementwise, but it still works
        }
        SMHIP_LAUNCH_CHECK("inline (dense)");
        return SMHIP_OK;
    }
generic:
    hipLaunchKernelGGL((inline_kernel<T, Op, CAP>), dim3((p.n + 255) / 256), dim3(256), 0, s, blk, static_cast<const T *>(a),
                       static_cast<const T *>(b), static_cast<T *>(out), p);
    SMHIP_LAUNCH_CHECK("inline");
    return SMHIP_OK;
}

}  // namespace

int launch_inline(int op, int dtype, const void *a, size_t a_host_bytes, const int64_t *sa, const void *b, size_t b_host_bytes,
                  const int64_t *sb, const int64_t *shape, int ndim, void *out, hipStream_t s) {
    InlineParams p{};
    size_t n = 1;
    for (int d = 0; d < ndim; ++d) {
        const int src = ndim - 1 - d;
        p.shape[d] = (uint32_t)shape[src];
        p.sa[d] = sa[src];
        p.sb[d] = sb[src];
        n *= (size_t)shape[src];
    }
    p.ndim = ndim;
    p.n = (uint32_t)n;
    p.a_inline = a_host_bytes != 0;
    p.b_inline = b_host_bytes != 0;
    const bool small = a_host_bytes <= (size_t)kSmallCap && b_host_bytes <= (size_t)kSmallCap;
#define SMHIP_INLINE_OP(T, OP) return small ? run_inline<T, OP<T>, kSmallCap>(a, a_host_bytes, b, b_host_bytes, out, p, s) \
                                            : run_inline<T, OP<T>, (int)kInlineBytes>(a, a_host_bytes, b, b_host_bytes, out, p, s)
#define SMHIP_INLINE_OPS(T)                                                              \
    switch (op) {                                                                        \
        case SMHIP_OP_ADD: SMHIP_INLINE_OP(T, AddOp);    \
        case SMHIP_OP_SUB: SMHIP_INLINE_OP(T, SubtractOp);    \
        case SMHIP_OP_MUL: SMHIP_INLINE_OP(T, MultiplyOp);    \
        case SMHIP_OP_DIV: SMHIP_INLINE_OP(T, DivideOp);    \
        case SMHIP_OP_POW: SMHIP_INLINE_OP(T, PowOp);    \
        case SMHIP_OP_LEFT: SMHIP_INLINE_OP(T, LeftOp);    \
    }                                                                                    \
    break;
    switch (dtype) {
        case SMHIP_F32: SMHIP_INLINE_OPS(float)
        case SMHIP_F64: SMHIP_INLINE_OPS(double)
        case SMHIP_I32: SMHIP_INLINE_OPS(int32_t)
        case SMHIP_I64: SMHIP_INLINE_OPS(int64_t)
    }
#undef SMHIP_INLINE_OPS
    return fail(SMHIP_ERR_INVALID, "elementwise_inline: bad op %d / dtype %d", op, dtype);
}

}  // namespace smhip
