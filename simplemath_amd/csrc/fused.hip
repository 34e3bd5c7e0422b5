// fused.hip -- two chained Ops in one pass: out = (a op1 b) op2 c.
//
// The reference evaluates `(a + b) * c` as two operator calls, each a full
// pass with a freshly allocated intermediate (SMArray.h:217-305): 24 B/elem of
// traffic for f32 plus the temporary.  Fused, the intermediate never leaves
// registers: 16 B/elem with an array `c`, 12 B/elem with a scalar `c`
// ("a fusion hook so (a + b) * c does one pass").  Each stage
// rounds exactly as the separate Ops do (-ffp-contract=off), so the result is
// bit-identical to the two-pass evaluation.
// Same streaming shape as contiguous.hip: one 16-byte vector per lane, no loop, nt.
#include "internal.h"
#include "ops.hip.h"

namespace smhip {
namespace {

using namespace dev;

constexpr int kBlock = 1024;

template <typename T, typename Op1, typename Op2, bool SCALAR_C>
__global__ __launch_bounds__(kBlock) void fused_vec_kernel(const T *__restrict__ a, const T *__restrict__ b,
                                                           const T *__restrict__ c, T cs, T *__restrict__ out, size_t n_vec,
                                                           int tail, int nt) {
    typedef typename VecTraits<T>::vec_t V;
    constexpr int W = VecTraits<T>::width;
    const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < n_vec) {
        const V va = load_stream_if(T, reinterpret_cast<const V *>(a) + i, nt);
        const V vb = load_stream_if(T, reinterpret_cast<const V *>(b) + i, nt);
        V vc;
        if constexpr (!SCALAR_C) vc = load_stream_if(T, reinterpret_cast<const V *>(c) + i, nt);
        V r;
#pragma unroll
        for (int k = 0; k < W; ++k) r[k] = Op2::apply(Op1::apply(va[k], vb[k]), SCALAR_C ? cs : vc[k]);
        store_stream_if(T, reinterpret_cast<V *>(out) + i, r, nt);
    } else if (i == n_vec) {
        for (int k = 0; k < tail; ++k) {
            const size_t e = n_vec * W + k;
            out[e] = Op2::apply(Op1::apply(a[e], b[e]), SCALAR_C ? cs : c[e]);
        }
    }
}

template <typename T, typename Op1, typename Op2>
int run(const void *a_, const void *b_, const void *c_, const void *cs_host, void *out_, size_t n, hipStream_t s) {
    constexpr int W = VecTraits<T>::width;
    const T *a = static_cast<const T *>(a_), *b = static_cast<const T *>(b_), *c = static_cast<const T *>(c_);
    T *out = static_cast<T *>(out_);
    const bool scalar = c_ == nullptr;
    const T cs = scalar ? *static_cast<const T *>(cs_host) : T{};
    const size_t n_vec = n / W, threads = n_vec + (n % W ? 1 : 0);
    const size_t g = (threads + kBlock - 1) / kBlock;
    if (g > 0x7fffffffu) return fail(SMHIP_ERR_UNSUPPORTED, "fused: array too large for one launch");
    if (scalar) hipLaunchKernelGGL((fused_vec_kernel<T, Op1, Op2, true>), dim3((unsigned)g), dim3(kBlock), 0, s, a, b, c, cs, out, n_vec, (int)(n % W), stream_policy({{a, n * sizeof(T)}, {b, n * sizeof(T)}}, {out, n * sizeof(T)}));
    else hipLaunchKernelGGL((fused_vec_kernel<T, Op1, Op2, false>), dim3((unsigned)g), dim3(kBlock), 0, s, a, b, c, cs, out, n_vec, (int)(n % W), stream_policy({{a, n * sizeof(T)}, {b, n * sizeof(T)}, {c, n * sizeof(T)}}, {out, n * sizeof(T)}));
    SMHIP_LAUNCH_CHECK("fused");
    return SMHIP_OK;
}

template <typename T, typename Op1>
int pick2(int op2, const void *a, const void *b, const void *c, const void *cs, void *out, size_t n, hipStream_t s) {
    switch (op2) {
        case SMHIP_OP_ADD: return run<T, Op1, AddOp<T>>(a, b, c, cs, out, n, s);
        case SMHIP_OP_SUB: return run<T, Op1, SubtractOp<T>>(a, b, c, cs, out, n, s);
        case SMHIP_OP_MUL: return run<T, Op1, MultiplyOp<T>>(a, b, c, cs, out, n, s);
        case SMHIP_OP_DIV: return run<T, Op1, DivideOp<T>>(a, b, c, cs, out, n, s);
    }
    return fail(SMHIP_ERR_UNSUPPORTED, "fused: second op %d not fusable (add, sub, mul, div are)", op2);
}

template <typename T>
int pick1(int op1, int op2, const void *a, const void *b, const void *c, const void *cs, void *out, size_t n, hipStream_t s) {
    switch (op1) {
        case SMHIP_OP_ADD: return pick2<T, AddOp<T>>(op2, a, b, c, cs, out, n, s);
        case SMHIP_OP_SUB: return pick2<T, SubtractOp<T>>(op2, a, b, c, cs, out, n, s);
        case SMHIP_OP_MUL: return pick2<T, MultiplyOp<T>>(op2, a, b, c, cs, out, n, s);
        case SMHIP_OP_DIV: return pick2<T, DivideOp<T>>(op2, a, b, c, cs, out, n, s);
    }
    return fail(SMHIP_ERR_UNSUPPORTED, "fused: first op %d not fusable (add, sub, mul, div are)", op1);
}

}  // namespace

int launch_fused(int op1, int op2, int dtype, const void *a, const void *b, const void *c, const void *c_scalar_host, void *out,
                 size_t n, hipStream_t s) {
    if (n == 0) return SMHIP_OK;
    switch (dtype) {
        case SMHIP_F32: return pick1<float>(op1, op2, a, b, c, c_scalar_host, out, n, s);
        case SMHIP_F64: return pick1<double>(op1, op2, a, b, c, c_scalar_host, out, n, s);
        case SMHIP_I32: return pick1<int32_t>(op1, op2, a, b, c, c_scalar_host, out, n, s);
        case SMHIP_I64: return pick1<int64_t>(op1, op2, a, b, c, c_scalar_host, out, n, s);
    }
    return fail(SMHIP_ERR_INVALID, "fused: bad dtype %d", dtype);
}

}  // namespace smhip
