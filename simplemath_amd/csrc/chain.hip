// chain.hip -- a CHAIN of broadcasted elementwise operators in one pass over HBM.
//
// The reference's idiom is a chain of operator calls, `(A * row + B) * 0.5f`: each operator is one full pass with a
// freshly allocated temporary (SMArray.h:217-305, `new T[n]` at :219; element_wise_op / array_scalar_op,
// calculate.h:5-169).  For f32 that chain moves 8 + 12 + 8 = 28 bytes per element; evaluated in ONE kernel the
// temporaries never leave registers and it moves 12 (A, B, out; the 16 KiB row comes from the caches).
//
//     r = x0;   r = r op1 x1  (or x1 op1 r);   r = r op2 x2;  ...      out[i] = r
//
// Every stage is the SAME single IEEE / wrapping operation the separate operators perform (-ffp-contract=off, the device
// functors of ops.hip.h), so the result is bit-identical to the operator chain.
//
// Leaves.  Against the dense row-major output every operand has one of four index forms (classified on the host):
//     dense      a[i]                    -- a full-size stream: 16-byte vectors, the launch's streaming policy
//     row        a[i mod P]              -- the operand ignores the leading axes (config 3's (1, 4096) row; (1,1,1,C) channel
//                                           values): one cached 16-byte load, P a whole number of vectors (a period that
//                                           is not -- rows of 3 -- is written out lcm(P, W) / P times first, a few bytes)
//     splat      a[(i / R) mod C]        -- constant along the trailing axes (a column, a per-row value, NCHW's (1,C,1,1)):
//                                           one cached element per vector (per element when R is not whole vectors)
//     scalar     a kernel argument
// An operand that is periodic but not dense within its period ((1,224,1,3) against (N,224,224,3), the reference tests'
// pattern, tests/add.cpp:59-92) has its period written out once (SMHIP_OP_LEFT through the broadcast kernels) and is a row.
// Anything else -- transposed or stepped views -- is not a leaf: the chain is cut there, that one operator runs through the
// broadcast kernels (tile / strided-row / gather: their rates, not a gather's), and the chain continues from its result.
//
// Kernel.  The Ops and the operand of each stage are RUN-TIME, wave-uniform codes in the argument block (a scalar branch per
// stage); what is compiled per variant is the load structure -- how many dense streams, rows and splats -- so that every
// load of a lane is issued before the first use, in one basic block, like the contiguous kernels.  One vector per lane,
// no loop over the data; 44 variants per element type (0-4 dense streams, 0-2 rows, 0-2 splats -- `(x - mean) / std` with
// per-row or per-channel statistics is two splats) instead of 4^k Op combinations times the operand forms.
// Roofline: HBM; algorithmic bytes = sizeof(T) * (dense leaves + 1) per element + the small operands once.
#include <stdlib.h>
#include <string.h>

#include <numeric>
#include <vector>

#include "internal.h"
#include "ops.hip.h"

namespace smhip {
namespace {

using namespace dev;

constexpr int kMaxDense = 4, kMaxRow = 2, kMaxSplat = 2, kMaxStages = 8;
constexpr int kSlotRow = kMaxDense, kSlotSplat = kMaxDense + kMaxRow, kSlotScalar = kMaxDense + kMaxRow + kMaxSplat;

template <typename T>
struct ChainArgs {
    const T *dense[kMaxDense];
    const T *row[kMaxRow];
    const T *spl[kMaxSplat];
    FastDiv row_p[kMaxRow];      // the period in vectors
    uint32_t row_ph[kMaxRow];    // this launch's first vector within the period
    FastDiv spl_r[kMaxSplat];    // elements (or, whole-vector form, vectors) per value
    FastDiv spl_c[kMaxSplat];    // values before the operand repeats
    uint32_t spl_r0[kMaxSplat];  // this launch's first element (vector) within its run of R
    uint32_t spl_q0[kMaxSplat];  // ... and the run's number, mod C
    uint32_t spl_elem[kMaxSplat];  // 1: R is not a whole number of vectors -- one index per element; 2: one element per WAVE (R a whole number of waves' worth)
    uint32_t n_stages, head;     // head: the slot r starts from
    uint32_t stage[kMaxStages];  // op | slot << 8 | swapped << 16: whole words, so that the stages are scalar loads at fixed offsets
    T scalar[kMaxStages];
};

template <typename T> __device__ __forceinline__ typename VecTraits<T>::full_t splat_of(T y) {
    typename VecTraits<T>::full_t v;
#pragma unroll
    for (int k = 0; k < VecTraits<T>::width; ++k) v[k] = y;
    return v;
}

// r = r op x (SWAP: x op r) for one vector; op is wave-uniform
template <typename T>
__device__ __forceinline__ typename VecTraits<T>::full_t chain_step(uint32_t op, bool swap, typename VecTraits<T>::full_t r,
                                                                    typename VecTraits<T>::full_t x) {
    typedef typename VecTraits<T>::full_t V;
    constexpr int W = VecTraits<T>::width;
    const V lhs = swap ? x : r, rhs = swap ? r : x;
    V o;
    switch (op) {
        case SMHIP_OP_ADD:
#pragma unroll
            for (int k = 0; k < W; ++k) o[k] = AddOp<T>::apply(lhs[k], rhs[k]);
            break;
        case SMHIP_OP_SUB:
#pragma unroll
            for (int k = 0; k < W; ++k) o[k] = SubtractOp<T>::apply(lhs[k], rhs[k]);
            break;
        case SMHIP_OP_MUL:
#pragma unroll
            for (int k = 0; k < W; ++k) o[k] = MultiplyOp<T>::apply(lhs[k], rhs[k]);
            break;
        case SMHIP_OP_POW:  // only ever r ^ 2 (launch_chain): the one multiplication sm::pow(a, 2) is (contiguous.hip: PowSquare; integer pow wraps the same way)
#pragma unroll
            for (int k = 0; k < W; ++k) o[k] = MultiplyOp<T>::apply(r[k], r[k]);
            break;
        default:
#pragma unroll
            for (int k = 0; k < W; ++k) o[k] = DivideOp<T>::apply(lhs[k], rhs[k]);
            break;
    }
    return o;
}

// The small operands' vectors for output vector v (v < 2^31: pieces, run_segment).
template <typename T, int NR, int NS>
__device__ __forceinline__ void chain_small_loads(const ChainArgs<T> &A, uint32_t v, typename VecTraits<T>::full_t (&r)[NR > 0 ? NR : 1],
                                                  typename VecTraits<T>::full_t (&s)[NS > 0 ? NS : 1]) {
    typedef typename VecTraits<T>::vec_t V;
    constexpr int W = VecTraits<T>::width;
#pragma unroll
    for (int k = 0; k < NR; ++k) {
        uint32_t q, m;
        A.row_p[k].divmod(A.row_ph[k] + v, q, m);
        r[k] = reinterpret_cast<const V *>(A.row[k])[m];  // the period is whole vectors: in bounds for the tail's lane too
    }
#pragma unroll
    for (int k = 0; k < NS; ++k) {
        if (A.spl_elem[k] == 2) {  // wave-uniform (run_segment): no vector instruction until the splat itself
            uint32_t q, m;
            A.spl_c[k].divmod(A.spl_q0[k] + A.spl_r[k].div(A.spl_r0[k] + (uint32_t)__builtin_amdgcn_readfirstlane((int)(v & ~63u))), q, m);
            s[k] = splat_of<T>(A.spl[k][__builtin_amdgcn_readfirstlane((int)m)]);
        } else if (A.spl_elem[k]) {
#pragma unroll
            for (int e = 0; e < W; ++e) {
                uint32_t q, m;
                A.spl_c[k].divmod(A.spl_q0[k] + A.spl_r[k].div(A.spl_r0[k] + v * W + e), q, m);
                s[k][e] = A.spl[k][m];
            }
        } else {
            uint32_t q, m;
            A.spl_c[k].divmod(A.spl_q0[k] + A.spl_r[k].div(A.spl_r0[k] + v), q, m);
            s[k] = splat_of<T>(A.spl[k][m]);
        }
    }
}

// The stages on one vector.  Unrolled: every stage word and scalar sits at a constant offset of the argument block (wide
// scalar loads up front, in flight with the vector loads); as a run-time loop over byte arrays the stages were three VMEM
// byte loads each -- a round trip to memory per stage and wave (37.3 us for (A * row + B) * 0.5 at 4096 x 4096).
template <typename T, int ND, int NR, int NS>
__device__ __forceinline__ typename VecTraits<T>::full_t chain_eval(const ChainArgs<T> &A, const typename VecTraits<T>::full_t (&d)[ND > 0 ? ND : 1],
                                                                    const typename VecTraits<T>::full_t (&r)[NR > 0 ? NR : 1],
                                                                    const typename VecTraits<T>::full_t (&s)[NS > 0 ? NS : 1]) {
    typedef typename VecTraits<T>::full_t F;
    // Which operand a stage takes is wave-uniform, so it is a scalar BRANCH to one register copy, not a chain of selects over
    // every operand the variant holds: at 1 KiB per wave the instruction stream is a visible part of the kernel (four
    // v_cndmask per held operand and stage; the empty asm keeps the compiler from turning the branches back into selects).
    auto pick = [&](uint32_t slot, T sc) {
        F x = splat_of<T>(sc);
#pragma unroll
        for (int k = 0; k < ND; ++k)
            if (slot == (uint32_t)k) { x = d[k]; asm volatile("" : "+v"(x)); }
#pragma unroll
        for (int k = 0; k < NR; ++k)
            if (slot == (uint32_t)(kSlotRow + k)) { x = r[k]; asm volatile("" : "+v"(x)); }
#pragma unroll
        for (int k = 0; k < NS; ++k)
            if (slot == (uint32_t)(kSlotSplat + k)) { x = s[k]; asm volatile("" : "+v"(x)); }
        return x;
    };
    F acc = pick(A.head, T{});
#pragma unroll
    for (int k = 0; k < kMaxStages; ++k) {
        if ((uint32_t)k >= A.n_stages) break;
        const uint32_t w = A.stage[k];
        acc = chain_step<T>(w & 0xffu, ((w >> 16) & 1u) != 0, acc, pick((w >> 8) & 0xffu, A.scalar[k]));
    }
    return acc;
}

// One tile of blockDim.x * U vectors per workgroup, no loop over the data.  Full tiles are guard-free: all U x ND streaming
// loads of a lane go out in one block (inside one arm of the read-policy branch), then the small operands' cached loads,
// then the stages and the stores.  The last, partial tile and the n % W tail elements take the guarded path.
template <typename T, int ND, int NR, int NS, int U>
__global__ __launch_bounds__(1024) void chain_kernel(ChainArgs<T> A, T *__restrict__ out, size_t n_vec, int tail, int pol) {
    typedef typename VecTraits<T>::vec_t V;
    typedef typename VecTraits<T>::full_t F;
    constexpr int W = VecTraits<T>::width;
    const size_t tile = (size_t)blockDim.x * U;
    const size_t base = (size_t)blockIdx.x * tile + threadIdx.x;
    if (((size_t)blockIdx.x + 1) * tile <= n_vec) {
        F d[U][ND > 0 ? ND : 1], r[U][NR > 0 ? NR : 1], s[U][NS > 0 ? NS : 1];
        if (pol & kLoadNt) {
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int k = 0; k < ND; ++k) d[u][k] = load_stream_as(T, reinterpret_cast<const V *>(A.dense[k]) + base + (size_t)u * blockDim.x, true);
        } else {
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int k = 0; k < ND; ++k) d[u][k] = load_stream_as(T, reinterpret_cast<const V *>(A.dense[k]) + base + (size_t)u * blockDim.x, false);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) chain_small_loads<T, NR, NS>(A, (uint32_t)(base + (size_t)u * blockDim.x), r[u], s[u]);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const F acc = chain_eval<T, ND, NR, NS>(A, d[u], r[u], s[u]);
            store_stream_if(T, reinterpret_cast<V *>(out) + base + (size_t)u * blockDim.x, acc, pol);
        }
        return;
    }
    for (int u = 0; u < U; ++u) {
        const size_t i = base + (size_t)u * blockDim.x;
        const bool body = i < n_vec;
        if (!body && !(i == n_vec && tail)) continue;
        F d[ND > 0 ? ND : 1], r[NR > 0 ? NR : 1], s[NS > 0 ? NS : 1];
#pragma unroll
        for (int k = 0; k < ND; ++k) {
            if (body) d[k] = load_stream(reinterpret_cast<const V *>(A.dense[k]) + i);
            else
                for (int e = 0; e < W; ++e) d[k][e] = e < tail ? A.dense[k][n_vec * W + e] : T{};  // the lane past the body: element by element
        }
        chain_small_loads<T, NR, NS>(A, (uint32_t)i, r, s);
        const F acc = chain_eval<T, ND, NR, NS>(A, d, r, s);
        if (body) store_stream(reinterpret_cast<V *>(out) + i, acc);
        else
            for (int e = 0; e < tail; ++e) out[n_vec * W + e] = acc[e];
    }
}

// ---- a chain's SUM without its result: `sm::pow(a - b, 2.0f).sum()`, the squared error, in one pass of 8 bytes per element ----
// Dense operands and scalars only.  One vector per lane; each element of the chain's value -- rounded to T stage by stage, as
// the operators round -- is widened (fp64 / wrapping 64-bit, reduce.hip: AccOf) and summed: lanes of a wave by DPP moves, waves
// in index order, one partial per workgroup; reduce.hip's finishing launch folds the partials in its fixed order, so the bits
// are the same on every run.  (Same scheme as the hipRTC expression + sum kernel, jit.hip.)
template <typename T, bool INTEGER = std::is_integral<T>::value> struct ChainAcc { typedef double type; };
template <typename T> struct ChainAcc<T, true> { typedef uint64_t type; };
template <typename T, typename A> __device__ __forceinline__ A chain_widen(T x) {
    if constexpr (std::is_integral<T>::value && std::is_signed<T>::value) return (A)(int64_t)x;
    else return (A)x;
}
template <int CTRL, int ROW_MASK, typename A> __device__ __forceinline__ A chain_dpp(A v) {
    const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)b, CTRL, ROW_MASK, 0xf, false);
    const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(b >> 32), CTRL, ROW_MASK, 0xf, false);
    return __builtin_bit_cast(A, ((unsigned long long)hi << 32) | lo);
}
constexpr int kChainSumBlock = 256, kChainSumU = 4;  // vectors per lane: a quarter of the partials, and a lane's loads all in flight together
template <typename T, int ND>
__global__ __launch_bounds__(kChainSumBlock) void chain_sum_kernel(ChainArgs<T> A, typename ChainAcc<T>::type *__restrict__ partials, size_t n_vec, int tail, int pol) {
    typedef typename VecTraits<T>::vec_t V;
    typedef typename VecTraits<T>::full_t F;
    typedef typename ChainAcc<T>::type Acc;
    constexpr int W = VecTraits<T>::width, U = kChainSumU;
    const size_t base = (size_t)blockIdx.x * (kChainSumBlock * U) + threadIdx.x;
    Acc acc = 0;
    F r[1], s[1];
    if (((size_t)blockIdx.x + 1) * (kChainSumBlock * U) <= n_vec) {  // a whole tile: no guards, every load of the lane goes out first
        F d[U][ND];
        if (pol & kLoadNt) {
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int k = 0; k < ND; ++k) d[u][k] = load_stream_as(T, reinterpret_cast<const V *>(A.dense[k]) + base + (size_t)u * kChainSumBlock, true);
        } else {
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int k = 0; k < ND; ++k) d[u][k] = load_stream_as(T, reinterpret_cast<const V *>(A.dense[k]) + base + (size_t)u * kChainSumBlock, false);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const F v = chain_eval<T, ND, 0, 0>(A, d[u], r, s);
#pragma unroll
            for (int e = 0; e < W; ++e) acc += chain_widen<T, Acc>(v[e]);
        }
    } else {
        for (int u = 0; u < U; ++u) {
            const size_t i = base + (size_t)u * kChainSumBlock;
            F d[ND];
            if (i < n_vec) {
#pragma unroll
                for (int k = 0; k < ND; ++k) d[k] = load_stream(reinterpret_cast<const V *>(A.dense[k]) + i);
                const F v = chain_eval<T, ND, 0, 0>(A, d, r, s);
#pragma unroll
                for (int e = 0; e < W; ++e) acc += chain_widen<T, Acc>(v[e]);
            } else if (i == n_vec && tail) {  // the n % W elements past the last whole vector
#pragma unroll
                for (int k = 0; k < ND; ++k)
                    for (int e = 0; e < W; ++e) d[k][e] = e < tail ? A.dense[k][n_vec * W + e] : T(1);
                const F v = chain_eval<T, ND, 0, 0>(A, d, r, s);
                for (int e = 0; e < tail; ++e) acc += chain_widen<T, Acc>(v[e]);
            }
        }
    }
    acc += chain_dpp<0x111, 0xf>(acc);  // row_shr:1, 2, 4, 8, then row_bcast:15 and :31 -- the wave's total ends up in lane 63
    acc += chain_dpp<0x112, 0xf>(acc);
    acc += chain_dpp<0x114, 0xf>(acc);
    acc += chain_dpp<0x118, 0xf>(acc);
    acc += chain_dpp<0x142, 0xa>(acc);
    acc += chain_dpp<0x143, 0xc>(acc);
    __shared__ Acc lds[kChainSumBlock / 64];
    if ((threadIdx.x & 63) == 63) lds[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        Acc total = lds[0];
        for (int w = 1; w < kChainSumBlock / 64; ++w) total += lds[w];  // index order
        partials[blockIdx.x] = total;
    }
}

// ------------------------------------------------------------------------------------------------ host side
enum LeafKind { kDense, kRow, kSplat, kScalar, kComplex };

struct Leaf {
    LeafKind kind = kComplex;
    const void *ptr = nullptr;
    const int64_t *strides = nullptr;  // against the full result shape (caller's)
    unsigned char scalar[8] = {};
    uint64_t P = 0;        // row: period in elements (a whole number of vectors once `owned` is written)
    uint64_t R = 0, C = 0; // splat
};

struct Problem {
    int dtype, esz, W;
    int ndim;                       // size-1 axes dropped
    int64_t shape[SMHIP_MAX_NDIM];
    int axis[SMHIP_MAX_NDIM];       // the caller's axis each kept axis is
    int64_t dense_strides[SMHIP_MAX_NDIM];
    int64_t full_shape[SMHIP_MAX_NDIM], full_dense[SMHIP_MAX_NDIM];
    int full_ndim;
    size_t n;
};

inline bool same_leaf(const Leaf *a, const Leaf *b) { return a->kind == b->kind && a->ptr == b->ptr && a->P == b->P && a->R == b->R && a->C == b->C; }

constexpr uint64_t kMaxWrittenPeriodBytes = (uint64_t)32 << 20;

// How `strides` (against the result shape) index the flat output.  Fills P / R / C; kRow with `needs_writeout` when the
// period must be written out first (not dense within itself, or not a whole number of vectors).
LeafKind classify(const Problem &pb, const int64_t *strides, uint64_t *P, uint64_t *R, uint64_t *C, bool *needs_writeout) {
    *needs_writeout = false;
    int lo = -1, hi = -1;
    bool dense = true;
    for (int j = 0; j < pb.ndim; ++j) {
        const int64_t s = strides[pb.axis[j]];
        if (s != 0) { if (lo < 0) lo = j; hi = j; }
        if (s != pb.dense_strides[j]) dense = false;
    }
    if (dense) return kDense;
    if (lo < 0) { *R = pb.W; *C = 1; return kSplat; }  // one element for the whole output
    // dense within [lo, hi], counted in its own extents?
    bool block = strides[pb.axis[hi]] == 1;
    for (int j = hi; j > lo && block; --j) block = strides[pb.axis[j - 1]] == strides[pb.axis[j]] * pb.shape[j];
    uint64_t trailing = 1;
    for (int j = hi + 1; j < pb.ndim; ++j) trailing *= (uint64_t)pb.shape[j];
    uint64_t period = 1;
    for (int j = lo; j < pb.ndim; ++j) period *= (uint64_t)pb.shape[j];
    if (block && hi == pb.ndim - 1 && period / pb.W < 0x7fffffffull) {
        *P = period;
        if (period % pb.W) *needs_writeout = true;
        return kRow;
    }
    if (block && trailing < 0x7fffffffull) {
        uint64_t c = 1;
        for (int j = lo; j <= hi; ++j) c *= (uint64_t)pb.shape[j];
        if (c < 0x7fffffffull) { *R = trailing; *C = c; return kSplat; }
    }
    // periodic, but with broadcast axes inside the period: worth writing the period out when it is small against the output
    if (lo > 0 && period * pb.esz <= kMaxWrittenPeriodBytes && period * 4 <= pb.n) {
        *P = period;
        *needs_writeout = true;
        return kRow;
    }
    return kComplex;
}

template <typename T, int U>
int launch_variant_u(int nd, int nr, int ns, const ChainArgs<T> &A, T *out, size_t n_vec, int tail, int pol, int block, hipStream_t s) {
    const size_t tiles = n_vec / ((size_t)block * U) + 1;  // the last workgroup: partial tile + tail elements (maybe empty)
    if (tiles > 0x7fffffffu) return fail(SMHIP_ERR_UNSUPPORTED, "chain: array too large for one launch");
    const dim3 grid((unsigned)tiles), blk(block);
#define SMHIP_CHAIN_CASE(ND, NR, NS) \
    case (ND) * 9 + (NR) * 3 + (NS): hipLaunchKernelGGL((chain_kernel<T, ND, NR, NS, U>), grid, blk, 0, s, A, out, n_vec, tail, pol); break;
#define SMHIP_CHAIN_SMALL(ND) \
    SMHIP_CHAIN_CASE(ND, 0, 1) SMHIP_CHAIN_CASE(ND, 0, 2) SMHIP_CHAIN_CASE(ND, 1, 0) SMHIP_CHAIN_CASE(ND, 1, 1) SMHIP_CHAIN_CASE(ND, 1, 2) \
    SMHIP_CHAIN_CASE(ND, 2, 0) SMHIP_CHAIN_CASE(ND, 2, 1) SMHIP_CHAIN_CASE(ND, 2, 2)
    switch (nd * 9 + nr * 3 + ns) {
        SMHIP_CHAIN_SMALL(0)
        SMHIP_CHAIN_CASE(1, 0, 0) SMHIP_CHAIN_SMALL(1)
        SMHIP_CHAIN_CASE(2, 0, 0) SMHIP_CHAIN_SMALL(2)
        SMHIP_CHAIN_CASE(3, 0, 0) SMHIP_CHAIN_SMALL(3)
        SMHIP_CHAIN_CASE(4, 0, 0) SMHIP_CHAIN_SMALL(4)
        default: return fail(SMHIP_ERR_INVALID, "chain: no kernel for %d dense / %d row / %d splat operands", nd, nr, ns);
    }
#undef SMHIP_CHAIN_SMALL
#undef SMHIP_CHAIN_CASE
    SMHIP_LAUNCH_CHECK("chain");
    return SMHIP_OK;
}

template <typename T>
int launch_variant(int nd, int nr, int ns, const ChainArgs<T> &A, T *out, size_t n_vec, int tail, int pol, hipStream_t s) {
    static const int forced_block = [] { const char *e = getenv("SMHIP_CHAIN_BLOCK"); return e && *e ? atoi(e) : 0; }();  // experiments
    static const int forced_u = [] { const char *e = getenv("SMHIP_CHAIN_U"); return e && *e ? atoi(e) : 0; }();
    // workgroups of 256, one vector per lane: (A * row + B) * 0.5 at 4096 x 4096 29.9 us (84 % of peak on its 12 B/elem) against
    // 31.1 with 1024 threads and 29.6 / 32.7 with two vectors per lane; at 8192 x 8192 121.5 against 137.8 / 122.9 / 140.3 us
    // (tools/chain_fused_rates.py, profiles/r04_chain_shapes.txt)
    // ... and 512 where no operand is a per-row / per-channel value and the output is at most 256 MiB (same box, A/B/A/B,
    // profiles/r04_chain_block.txt: (A * row + B) * 0.5 at 4096^2 82.3 -> 84.2 %, at 8192^2 80.0 -> 83.6, on rotating operands
    // 77.1 -> 78.8; (A + B) * 0.5 rotating 75.9 -> 78.3 -- but splat forms on rotating operands 80 -> 75 and everything at
    // 16384 x 8192 one to five points WORSE)
    const bool clamp_ok = forced_block >= 64 && forced_block <= 1024 && forced_block % 64 == 0;
    const int block = clamp_ok ? forced_block : (ns == 0 && n_vec <= ((size_t)1 << 24) ? 512 : 256);
    (void)nd;
    const int u = forced_u ? forced_u : 1;
    if (u == 2) return launch_variant_u<T, 2>(nd, nr, ns, A, out, n_vec, tail, pol, block, s);
    return launch_variant_u<T, 1>(nd, nr, ns, A, out, n_vec, tail, pol, block, s);
}

// A few items in place: planning a chain allocates nothing (a chain call on a 4 MB array is host-bound: every 100 ns of
// planning is 3 % of it).
template <typename X, int CAP>
struct Few {
    X item[CAP];
    int count = 0;
    void push_back(const X &x) { item[count++] = x; }
    void clear() { count = 0; }
    bool empty() const { return count == 0; }
    size_t size() const { return (size_t)count; }
    X &operator[](size_t i) { return item[i]; }
    const X &operator[](size_t i) const { return item[i]; }
    X *data() { return item; }
    const X *begin() const { return item; }
    const X *end() const { return item + count; }
};
// One fused segment: r = leaves[0]; r = r op[k] leaves[k + 1] ... -> out (dense, pb.n elements).
struct Segment {
    Few<const Leaf *, kMaxStages + 1> leaves;  // leaves[0] = head
    Few<int, kMaxStages> ops, swaps;
};

template <typename T>
int run_segment(const Problem &pb, const Segment &sg, void *out_, hipStream_t s) {
    constexpr int W = VecTraits<T>::width;
    ChainArgs<T> A{};
    int nd = 0, nr = 0, ns = 0;
    const Leaf *rows[kMaxRow] = {}, *spls[kMaxSplat] = {};
    Few<Span, kMaxDense> reads;
    auto slot_of = [&](const Leaf *lf, int stage) -> int {
        switch (lf->kind) {
            case kDense:
                for (int k = 0; k < nd; ++k)
                    if (A.dense[k] == lf->ptr) return k;  // the same array twice (a * a): one stream
                if (nd >= kMaxDense) return -1;
                A.dense[nd] = static_cast<const T *>(lf->ptr);
                reads.push_back({lf->ptr, pb.n * sizeof(T)});
                return nd++;
            case kRow:
                for (int k = 0; k < nr; ++k)
                    if (same_leaf(rows[k], lf)) return kSlotRow + k;
                if (nr >= kMaxRow) return -1;
                rows[nr] = lf;
                A.row[nr] = static_cast<const T *>(lf->ptr);
                return kSlotRow + nr++;
            case kSplat:
                for (int k = 0; k < ns; ++k)
                    if (same_leaf(spls[k], lf)) return kSlotSplat + k;
                if (ns >= kMaxSplat) return -1;
                spls[ns] = lf;
                A.spl[ns] = static_cast<const T *>(lf->ptr);
                return kSlotSplat + ns++;
            case kScalar:
                if (stage >= 0) memcpy(&A.scalar[stage], lf->scalar, sizeof(T));
                return kSlotScalar;
            default:
                return -1;
        }
    };
    const int head_slot = slot_of(sg.leaves[0], -1);
    if (head_slot < 0) return fail(SMHIP_ERR_INVALID, "chain: segment head has no operand slot");
    A.head = (uint32_t)head_slot;
    A.n_stages = (uint32_t)sg.ops.size();
    for (size_t k = 0; k < sg.ops.size(); ++k) {
        const int slot = slot_of(sg.leaves[k + 1], (int)k);
        if (slot < 0) return fail(SMHIP_ERR_INVALID, "chain: a segment with more operands than its kernel variant takes (stage %zu)", k);  // the planner's bug, never the GPU's problem
        A.stage[k] = (uint32_t)sg.ops[k] | (uint32_t)slot << 8 | (uint32_t)(sg.swaps[k] ? 1 : 0) << 16;
    }
    T *out = static_cast<T *>(out_);
    const size_t n_vec = pb.n / W;
    const int tail = (int)(pb.n % W);
    int pol = stream_policy((size_t)nd * pb.n * sizeof(T), pb.n * sizeof(T));
    pol = refine_policy(pol, reads.data(), reads.size(), {out, pb.n * sizeof(T)});
    // pieces: the large-array rule of the contiguous kernels (internal.h: piece_for); with row / splat operands a launch also
    // stays below 2^31 elements, its 32-bit index arithmetic's reach
    size_t piece = piece_for(n_vec, nd + 1 < 3 ? nd + 1 : 3);
    const size_t reach = ((size_t)1 << 31) / W / 2;
    if ((nr || ns) && (piece == 0 || piece > reach) && n_vec > reach) piece = reach;
    if (piece == 0 || piece >= n_vec) piece = n_vec ? n_vec : 1;
    const T *dense0[kMaxDense];
    for (int k = 0; k < kMaxDense; ++k) dense0[k] = A.dense[k];
    for (size_t v0 = 0;; v0 += piece) {
        const bool last = v0 + piece >= n_vec;
        const size_t nv = last ? n_vec - v0 : piece;
        for (int k = 0; k < nd; ++k) A.dense[k] = dense0[k] + v0 * W;
        for (int k = 0; k < nr; ++k) {
            const uint64_t pv = rows[k]->P / W;
            A.row_p[k] = FastDiv((uint32_t)pv);
            A.row_ph[k] = (uint32_t)(v0 % pv);
        }
        for (int k = 0; k < ns; ++k) {
            const bool elem = spls[k]->R % W != 0;
            const uint64_t r = elem ? spls[k]->R : spls[k]->R / W, at = elem ? (uint64_t)v0 * W : (uint64_t)v0;
            A.spl_elem[k] = elem;
            A.spl_r[k] = FastDiv((uint32_t)r);
            A.spl_c[k] = FastDiv((uint32_t)spls[k]->C);
            A.spl_r0[k] = (uint32_t)(at % r);
            A.spl_q0[k] = (uint32_t)((at / r) % spls[k]->C);
            // a wave's 64 vectors are consecutive and start at a multiple of 64: with R a whole number of those, and the piece
            // starting on one, every lane of a wave wants the SAME element -- the index is scalar arithmetic and one scalar load
            static const bool uniform_ok = [] { const char *e = getenv("SMHIP_CHAIN_UNIFORM_SPLAT"); return !(e && *e && atoi(e) == 0); }();
            if (!elem && uniform_ok && r % 64 == 0 && A.spl_r0[k] % 64 == 0) A.spl_elem[k] = 2;
        }
        if (int rc = launch_variant<T>(nd, nr, ns, A, out + v0 * W, nv, last ? tail : 0, pol, s)) return rc;
        if (last) break;
    }
    return SMHIP_OK;
}

int run_segment_dtype(const Problem &pb, const Segment &sg, void *out, hipStream_t s) {
    switch (pb.dtype) {
        case SMHIP_F32: return run_segment<float>(pb, sg, out, s);
        case SMHIP_F64: return run_segment<double>(pb, sg, out, s);
        case SMHIP_I32: return run_segment<int32_t>(pb, sg, out, s);
        case SMHIP_I64: return run_segment<int64_t>(pb, sg, out, s);
    }
    return fail(SMHIP_ERR_INVALID, "chain: bad dtype %d", pb.dtype);
}

// Pooled temporaries of one smhip_chain call; handed back when the call returns (the pool orders their reuse after the
// launches queued here).
struct Temps {
    static constexpr int kCap = 4 * SMHIP_CHAIN_MAX_OPERANDS + 8;  // per operand at most a written-out period, two temporaries of a cut and a scalar slot
    Few<void *, kCap> all;
    ~Temps() { for (void *p : all) smhip_free(p); }
    int take(size_t bytes, void **p) {
        if (all.count == kCap) return fail(SMHIP_ERR_UNSUPPORTED, "chain: too many temporaries");
        if (int rc = smhip_alloc(p, bytes ? bytes : 1)) return rc;
        all.push_back(*p);
        return SMHIP_OK;
    }
};

void make_problem(int dtype, const int64_t *shape, int ndim, Problem *out) {
    Problem &pb = *out;
    pb = Problem{};
    pb.dtype = dtype;
    pb.esz = (int)dtype_size(dtype);
    pb.W = 16 / pb.esz;
    pb.n = 1;
    pb.full_ndim = ndim;
    for (int i = 0; i < ndim; ++i) {
        pb.full_shape[i] = shape[i];
        pb.n *= (size_t)shape[i];
        if (shape[i] != 1) {
            pb.shape[pb.ndim] = shape[i];
            pb.axis[pb.ndim++] = i;
        }
    }
    int64_t acc = 1;
    for (int j = pb.ndim; j-- > 0;) { pb.dense_strides[j] = acc; acc *= pb.shape[j]; }
    acc = 1;
    for (int i = ndim; i-- > 0;) { pb.full_dense[i] = acc; acc *= shape[i]; }
}

// Classifies one array operand (lf.ptr / lf.strides set) and, for a row whose period needs it, writes the period out.
int prepare_leaf(const Problem &pb, Leaf &lf, Temps &temps, hipStream_t s) {
    bool writeout = false;
    lf.kind = classify(pb, lf.strides, &lf.P, &lf.R, &lf.C, &writeout);
    if (lf.kind == kRow && writeout) {
        // the period, repeated to a whole number of vectors: (rep, axes from the first one the operand moves along)
        // gathered through SMHIP_OP_LEFT
        const uint64_t rep = pb.W / std::gcd<uint64_t, uint64_t>(lf.P, (uint64_t)pb.W);
        int64_t sh[SMHIP_MAX_NDIM + 1], st[SMHIP_MAX_NDIM + 1], zero[SMHIP_MAX_NDIM + 1] = {};
        int nd = 0;
        uint64_t cover = 1;
        int first = pb.ndim;
        for (int j = pb.ndim; j-- > 0;) {
            if (cover == lf.P) break;
            cover *= (uint64_t)pb.shape[j];
            first = j;
        }
        if (rep > 1) { sh[nd] = (int64_t)rep; st[nd++] = 0; }
        for (int j = first; j < pb.ndim; ++j) { sh[nd] = pb.shape[j]; st[nd++] = lf.strides[pb.axis[j]]; }
        if (nd > SMHIP_MAX_NDIM) { lf.kind = kComplex; return SMHIP_OK; }
        void *buf;
        if (int rc = temps.take(lf.P * rep * pb.esz, &buf)) return rc;
        if (int rc = launch_broadcast(SMHIP_OP_LEFT, pb.dtype, lf.ptr, st, lf.ptr, zero, sh, nd, buf, s)) return rc;
        lf.ptr = buf;
        lf.P *= rep;
    }
    return SMHIP_OK;
}

}  // namespace

// smhip_fused_expr_bcast: the operands of a run-time compiled expression in the index forms above; an operand that has
// none (a transposed or stepped view) is copied dense first (SMHIP_OP_LEFT through the broadcast kernels).
int launch_expr_bcast(const char *expr, int dtype, const void *const *operands, const int64_t *strides, int n_operands,
                      const void *scalars_host, int n_scalars, const int64_t *shape, int ndim, void *out, hipStream_t s) {
    Problem pb;
    make_problem(dtype, shape, ndim, &pb);
    if (pb.n == 0) return SMHIP_OK;
    Temps temps;
    ExprLeaf forms[8];
    for (int k = 0; k < n_operands; ++k) {
        Leaf lf;
        lf.ptr = operands[k];
        lf.strides = strides + (size_t)k * ndim;
        if (int rc = prepare_leaf(pb, lf, temps, s)) return rc;
        if (lf.kind == kComplex) {
            void *buf;
            if (int rc = temps.take(pb.n * pb.esz, &buf)) return rc;
            const int64_t zero[SMHIP_MAX_NDIM] = {};
            if (int rc = launch_broadcast(SMHIP_OP_LEFT, dtype, lf.ptr, lf.strides, lf.ptr, zero, pb.full_shape, pb.full_ndim, buf, s)) return rc;
            lf.ptr = buf;
            lf.kind = kDense;
        }
        forms[k] = ExprLeaf{lf.kind == kDense ? 0 : lf.kind == kRow ? 1 : 2, lf.ptr, lf.P, lf.R, lf.C};
    }
    bool all_dense = true;
    const void *dense[8] = {};
    for (int k = 0; k < n_operands; ++k) { all_dense = all_dense && forms[k].kind == 0; dense[k] = forms[k].ptr; }
    if (all_dense) return jit_fused_expr(expr, dtype, dense, n_operands, scalars_host, n_scalars, out, pb.n, nullptr, s);  // the flat kernel
    return jit_fused_expr_bcast(expr, dtype, forms, n_operands, scalars_host, n_scalars, out, pb.n, s);
}

namespace {
// The one-pass sum of a chain whose operands are all dense or scalars (chain_sum_kernel); false: not that kind of chain.
template <typename T>
int try_chain_sum(const Problem &pb, int n_operands, const void *const *operands, const int64_t *strides, int ndim, const void *scalars_host,
                  const int *ops, const int *swapped, double *sum_dev, hipStream_t s, bool *done) {
    *done = false;
    constexpr int W = VecTraits<T>::width;
    const int n_stages = n_operands - 1;
    if (n_stages > kMaxStages) return SMHIP_OK;
    ChainArgs<T> A{};
    int nd = 0;
    Few<Span, kMaxDense> reads;
    auto slot_of = [&](int k, int stage) -> int {
        if (!operands[k]) {
            if (stage >= 0) memcpy(&A.scalar[stage], static_cast<const char *>(scalars_host) + (size_t)k * sizeof(T), sizeof(T));
            return kSlotScalar;
        }
        uint64_t P, R, C;
        bool writeout;
        if (classify(pb, strides + (size_t)k * ndim, &P, &R, &C, &writeout) != kDense) return -1;
        for (int j = 0; j < nd; ++j)
            if (A.dense[j] == operands[k]) return j;
        if (nd >= kMaxDense) return -1;
        A.dense[nd] = static_cast<const T *>(operands[k]);
        reads.push_back({operands[k], pb.n * sizeof(T)});
        return nd++;
    };
    const int head = slot_of(0, -1);
    if (head < 0 || head == kSlotScalar) return SMHIP_OK;
    A.head = (uint32_t)head;
    A.n_stages = (uint32_t)n_stages;
    for (int k = 0; k < n_stages; ++k) {
        if (ops[k] == SMHIP_OP_POW) {  // only the square is a stage (launch_chain)
            T e;
            memcpy(&e, static_cast<const char *>(scalars_host) + (size_t)(k + 1) * sizeof(T), sizeof(T));
            if (operands[k + 1] || swapped[k] || !(e == T(2))) return SMHIP_OK;
        }
        const int slot = slot_of(k + 1, k);
        if (slot < 0) return SMHIP_OK;
        A.stage[k] = (uint32_t)ops[k] | (uint32_t)slot << 8 | (uint32_t)(swapped[k] ? 1 : 0) << 16;
    }
    const size_t n_vec = pb.n / W;
    const int tail = (int)(pb.n % W);
    constexpr size_t kTile = (size_t)kChainSumBlock * kChainSumU;  // vectors per workgroup
    const size_t grid = (n_vec + (tail ? 1 : 0) + kTile - 1) / kTile;
    if (grid == 0 || grid > 0x7fffffffu) return SMHIP_OK;
    int pol = stream_policy((size_t)nd * pb.n * sizeof(T), 0);
    pol = refine_policy(pol, reads.data(), reads.size(), {nullptr, 0});
    double *scratch;
    ScratchLease lease;
    if (int rc = lease.take(grid + grid / kReduceFoldSpan + 2, &scratch)) return rc;
    typedef typename ChainAcc<T>::type Acc;
    Acc *partials = reinterpret_cast<Acc *>(scratch);
    // very large arrays in pieces, like every streaming kernel (internal.h: piece_for): a piece is a whole number of workgroups,
    // its partials follow the previous piece's
    size_t piece = piece_for(n_vec, nd < 3 ? nd : 3);
    if (piece == 0 || piece >= n_vec) piece = n_vec ? n_vec : 1;
    piece = (piece + kTile - 1) / kTile * kTile;
    const T *dense0[kMaxDense];
    for (int k = 0; k < kMaxDense; ++k) dense0[k] = A.dense[k];
    for (size_t v0 = 0;; v0 += piece) {
        const bool last = v0 + piece >= n_vec;
        const size_t nv = last ? n_vec - v0 : piece;
        const int tl = last ? tail : 0;
        const size_t blocks = (nv + (tl ? 1 : 0) + kTile - 1) / kTile;
        for (int k = 0; k < nd; ++k) A.dense[k] = dense0[k] + v0 * W;
        Acc *part = partials + v0 / kTile;
        if (blocks) {
            switch (nd) {
                case 1: hipLaunchKernelGGL((chain_sum_kernel<T, 1>), dim3((unsigned)blocks), dim3(kChainSumBlock), 0, s, A, part, nv, tl, pol); break;
                case 2: hipLaunchKernelGGL((chain_sum_kernel<T, 2>), dim3((unsigned)blocks), dim3(kChainSumBlock), 0, s, A, part, nv, tl, pol); break;
                case 3: hipLaunchKernelGGL((chain_sum_kernel<T, 3>), dim3((unsigned)blocks), dim3(kChainSumBlock), 0, s, A, part, nv, tl, pol); break;
                default: hipLaunchKernelGGL((chain_sum_kernel<T, 4>), dim3((unsigned)blocks), dim3(kChainSumBlock), 0, s, A, part, nv, tl, pol); break;
            }
            SMHIP_LAUNCH_CHECK("chain_sum_kernel");
        }
        if (last) break;
    }
    *done = true;
    return reduce_finish(pb.dtype, scratch, grid, sum_dev, s);
}
}  // namespace

// The sum of a chain's value (fp64 / wrapping 64-bit accumulation, as smhip_sum) without writing the value: one pass when every
// operand is dense or a scalar; otherwise the chain into a temporary, then the sum of that.
int launch_chain_sum(int dtype, int n_operands, const void *const *operands, const int64_t *strides, const void *scalars_host,
                     const int *ops, const int *swapped, const int64_t *shape, int ndim, double *sum_dev, hipStream_t s) {
    Problem pb;
    make_problem(dtype, shape, ndim, &pb);
    if (pb.n == 0) return fail(SMHIP_ERR_INVALID, "chain_sum: empty shape");
    bool done = false;
    int rc = SMHIP_OK;
    switch (dtype) {
        case SMHIP_F32: rc = try_chain_sum<float>(pb, n_operands, operands, strides, ndim, scalars_host, ops, swapped, sum_dev, s, &done); break;
        case SMHIP_F64: rc = try_chain_sum<double>(pb, n_operands, operands, strides, ndim, scalars_host, ops, swapped, sum_dev, s, &done); break;
        case SMHIP_I32: rc = try_chain_sum<int32_t>(pb, n_operands, operands, strides, ndim, scalars_host, ops, swapped, sum_dev, s, &done); break;
        default: rc = try_chain_sum<int64_t>(pb, n_operands, operands, strides, ndim, scalars_host, ops, swapped, sum_dev, s, &done); break;
    }
    if (rc || done) return rc;
    Temps temps;
    void *value;
    if (int rc2 = temps.take(pb.n * pb.esz, &value)) return rc2;
    if (int rc2 = launch_chain(dtype, n_operands, operands, strides, scalars_host, ops, swapped, shape, ndim, value, s)) return rc2;
    return launch_sum(dtype, value, pb.n, sum_dev, s);
}

int launch_chain(int dtype, int n_operands, const void *const *operands, const int64_t *strides, const void *scalars_host,
                 const int *ops, const int *swapped, const int64_t *shape, int ndim, void *out, hipStream_t s) {
    Problem pb;
    make_problem(dtype, shape, ndim, &pb);
    if (pb.n == 0) return SMHIP_OK;
    Temps temps;
    Leaf leaves[SMHIP_CHAIN_MAX_OPERANDS];
    for (int k = 0; k < n_operands; ++k) {
        Leaf &lf = leaves[k];
        if (!operands[k]) {
            lf.kind = kScalar;
            memcpy(lf.scalar, static_cast<const char *>(scalars_host) + (size_t)k * pb.esz, pb.esz);
            continue;
        }
        lf.ptr = operands[k];
        lf.strides = strides + (size_t)k * ndim;
        if (int rc = prepare_leaf(pb, lf, temps, s)) return rc;
    }
    // Walk the stages, cutting the chain where a leaf is not fusable or a kernel variant's operand counts run out.
    const Leaf *head = &leaves[0];
    Leaf temp_heads[kMaxStages * 4 + 4];
    int n_temp = 0;
    Segment sg;
    int nd = 0, nr = 0, ns = 0;
    Few<const void *, kMaxDense + 1> dense_seen;
    Few<const Leaf *, kMaxRow + 1> row_seen, spl_seen;
    auto reset_counts = [&] { nd = nr = ns = 0; dense_seen.clear(); row_seen.clear(); spl_seen.clear(); };
    auto fits = [&](const Leaf *lf) {  // would this leaf still find a slot?
        switch (lf->kind) {
            case kDense:
                for (const void *p : dense_seen) if (p == lf->ptr) return true;
                return nd < kMaxDense;
            case kRow:
                for (const Leaf *q : row_seen) if (same_leaf(q, lf)) return true;
                return nr < kMaxRow;
            case kSplat:
                for (const Leaf *q : spl_seen) if (same_leaf(q, lf)) return true;
                return ns < kMaxSplat;
            default: return true;
        }
    };
    auto count = [&](const Leaf *lf) {
        switch (lf->kind) {
            case kDense:
                for (const void *p : dense_seen) if (p == lf->ptr) return;
                dense_seen.push_back(lf->ptr); ++nd; break;
            case kRow:
                for (const Leaf *q : row_seen) if (same_leaf(q, lf)) return;
                row_seen.push_back(lf); ++nr; break;
            case kSplat:
                for (const Leaf *q : spl_seen) if (same_leaf(q, lf)) return;
                spl_seen.push_back(lf); ++ns; break;
            default: break;
        }
    };
    auto new_temp_head = [&](void *buf) {
        Leaf &t = temp_heads[n_temp++];
        t = Leaf{};
        t.kind = kDense;
        t.ptr = buf;
        t.strides = pb.full_dense;
        return &t;
    };
    // flush the pending fused segment into `dst` (or a fresh temporary): the chain continues from a dense head
    auto emit = [&](void *dst, const Leaf **new_head) -> int {
        if (sg.ops.empty()) { *new_head = sg.leaves[0]; return SMHIP_OK; }
        void *buf = dst;
        if (!buf) if (int rc = temps.take(pb.n * pb.esz, &buf)) return rc;
        if (int rc = run_segment_dtype(pb, sg, buf, s)) return rc;
        *new_head = new_temp_head(buf);
        return SMHIP_OK;
    };
    auto start = [&](const Leaf *h) {
        sg = Segment{};
        sg.leaves.push_back(h);
        reset_counts();
        count(h);
    };
    // one operator through the broadcast kernels: x op y -> dst (or a temporary)
    auto eager = [&](int op, const Leaf *x, const Leaf *y, void *dst, const Leaf **result) -> int {
        void *buf = dst;
        if (!buf) if (int rc = temps.take(pb.n * pb.esz, &buf)) return rc;
        const int64_t zero[SMHIP_MAX_NDIM] = {};
        const void *px = x->ptr, *py = y->ptr;
        const int64_t *sx = x->strides, *sy = y->strides;
        void *sbuf = nullptr;
        if (x->kind == kScalar || y->kind == kScalar) {  // a scalar in an operator that has to run alone: one element in device memory
            if (int rc = temps.take(8, &sbuf)) return rc;
            const Leaf *sc = x->kind == kScalar ? x : y;
            SMHIP_TRY(hipMemcpyAsync(sbuf, sc->scalar, pb.esz, hipMemcpyHostToDevice, s));
            if (x->kind == kScalar) { px = sbuf; sx = zero; } else { py = sbuf; sy = zero; }
        }
        if (int rc = launch_broadcast(op, dtype, px, sx, py, sy, pb.full_shape, pb.full_ndim, buf, s)) return rc;
        *result = new_temp_head(buf);
        return SMHIP_OK;
    };
    const int n_stages = n_operands - 1;
    // leaves whose period was written out keep their ORIGINAL pointer / strides for operators that run alone
    Leaf originals[SMHIP_CHAIN_MAX_OPERANDS];
    for (int k = 0; k < n_operands; ++k) {
        originals[k] = leaves[k];
        if (operands[k]) { originals[k].ptr = operands[k]; originals[k].strides = strides + (size_t)k * ndim; }
    }
    auto original_of = [&](const Leaf *lf) -> const Leaf * {
        if (lf >= leaves && lf < leaves + n_operands) return &originals[lf - leaves];
        return lf;
    };
    // r ^ s (sm::pow of the chain's value; s a scalar, checked by the entry point).  s = 2 is a stage like any other -- the one
    // multiplication sm::pow(a, 2) is; s = 1 is no stage at all; any other exponent has an evaluation of its own
    // (contiguous.hip: run_scalar -- exact forms, the half-integer chains, pow_core_u, the table kernels), so the chain is cut
    // there and that evaluation runs on the value so far: the same bits as the operator called by itself.
    auto pow_value = [&](const Leaf *x, double *v) {
        switch (dtype) {
            case SMHIP_F32: { float f; memcpy(&f, x->scalar, 4); *v = f; break; }
            case SMHIP_F64: { double f; memcpy(&f, x->scalar, 8); *v = f; break; }
            case SMHIP_I32: { int32_t f; memcpy(&f, x->scalar, 4); *v = (double)f; break; }
            default: { int64_t f; memcpy(&f, x->scalar, 8); *v = (double)f; break; }
        }
    };
    start(head);
    for (int k = 0; k < n_stages; ++k) {
        const Leaf *x = &leaves[k + 1];
        const bool last = k == n_stages - 1;
        const Leaf *h = sg.leaves[0];
        if (ops[k] == SMHIP_OP_POW) {
            double e = 0;
            pow_value(x, &e);
            const bool head_plain = sg.ops.empty() && h->kind != kDense;  // a broadcast operand with nothing applied yet: nothing to square in place
            if (e == 2.0 && !head_plain && (int)sg.ops.size() < kMaxStages) {
                sg.leaves.push_back(x);
                sg.ops.push_back(SMHIP_OP_POW);
                sg.swaps.push_back(0);
                continue;
            }
            if (e == 1.0 && !head_plain && !last) continue;
            const Leaf *cur;
            if (int rc = emit(nullptr, &cur)) return rc;  // the value so far, dense (a dense head as it is)
            if (cur->kind != kDense) {  // a head that broadcasts: written out through the copy Op first
                const Leaf *res;
                if (int rc = eager(SMHIP_OP_LEFT, original_of(cur), original_of(cur), nullptr, &res)) return rc;
                cur = res;
            }
            void *buf = last ? out : nullptr;
            if (!buf) if (int rc = temps.take(pb.n * pb.esz, &buf)) return rc;
            if (int rc = launch_array_scalar(SMHIP_OP_POW, dtype, cur->ptr, x->scalar, pb.n, buf, s)) return rc;
            if (last) return SMHIP_OK;
            start(new_temp_head(buf));
            continue;
        }
        const bool head_complex = sg.ops.empty() && h->kind == kComplex;
        if (x->kind == kComplex || head_complex) {
            const Leaf *cur;
            if (int rc = emit(nullptr, &cur)) return rc;
            const Leaf *res;
            const Leaf *lx = swapped[k] ? original_of(x) : original_of(cur), *ly = swapped[k] ? original_of(cur) : original_of(x);
            if (int rc = eager(ops[k], lx, ly, last ? out : nullptr, &res)) return rc;
            if (last) return SMHIP_OK;
            start(res);
            continue;
        }
        if ((int)sg.ops.size() == kMaxStages || !fits(x)) {
            const Leaf *cur;
            if (int rc = emit(nullptr, &cur)) return rc;
            start(cur);
            if (!fits(x)) {
                // the head itself holds the slot this operand needs (two different rows, two different columns: `row1 + row2`
                // at the very start of a chain, where there is nothing to flush): this one operator runs alone
                const Leaf *res;
                const Leaf *lx = swapped[k] ? original_of(x) : original_of(cur), *ly = swapped[k] ? original_of(cur) : original_of(x);
                if (int rc = eager(ops[k], lx, ly, last ? out : nullptr, &res)) return rc;
                if (last) return SMHIP_OK;
                start(res);
                continue;
            }
        }
        sg.leaves.push_back(x);
        sg.ops.push_back(ops[k]);
        sg.swaps.push_back(swapped[k] ? 1 : 0);
        count(x);
    }
    const Leaf *done;
    return emit(out, &done);
}

}  // namespace smhip
