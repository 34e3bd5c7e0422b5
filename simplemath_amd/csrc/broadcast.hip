// broadcast.hip -- the strided / broadcast kernels.
//
// Replaces the general N-D loop of element_wise_op<T,Op> (reference
// include/math/calculate.h:16-96): per element, ndim x (div + mod) to unravel
// the linear index, scalar Op::apply, OpenMP over 1024-element chunks;
// `canVectorize` is identically false there (:43-46).
//
// Here the host first normalises the problem (drop size-1 dims, merge dims
// that are jointly dense in both operands) and then picks
//   row kernel     inner strides in {0,1}: the contiguous axis is streamed with
//                  16-byte vectors; the unravel runs once per ROW (mul-hi/shift
//                  "fast division", no / or %), an operand with inner stride 0
//                  is one scalar per row, and an operand with all-zero outer
//                  strides (the (1 x 4096) row of BASELINE config 3) is loaded
//                  once per workgroup and kept in registers across its rows;
//   tile kernel    an operand whose own contiguous axis is NOT the output's inner
//                  axis (transpose() views, permuted 3-D+ views): 64 x 128 patches of
//                  the (p, q) plane -- p = the operand's contiguous axis, q = the
//                  output's inner axis -- are read coalesced along p, turned through
//                  a padded LDS tile, and consumed coalesced along q.  The reference
//                  walks such operands with a div/mod chain and a strided scalar load
//                  per element; a naive GPU gather would touch one 64-byte line per
//                  lane (16x read amplification).
//   LDS kernel     one operand dense in output order, the other small (<= 32 KiB once every stride-0
//                  axis is dropped) but broadcast along the INNER axis or with a tiny inner extent
//                  -- the reference tests' own pattern, (N,224,224,3) op (1,224,1,3): the small
//                  operand is staged whole into LDS once per workgroup, the dense one streams
//                  as 16-byte vectors, and the per-element unravel (fast division) only feeds
//                  an LDS read.
//   gather kernel  whatever is left (tiny extents, irregular strides): W
//                  consecutive outputs per lane so the store is still a
//                  coalesced 16-byte vector; operand loads are per-element
//                  gathers through fast-division unravel.
// Roofline: HBM-bound; algorithmic bytes = sizeof(T) * (|a| + |b| + |out|)
// with each broadcast operand counted once (config 3: 134 234 112 B).
#include <type_traits>

#include "internal.h"
#include "ops.hip.h"

namespace smhip {
namespace {

using namespace dev;

constexpr int kMaxOuter = SMHIP_MAX_NDIM - 1;

struct RowParams {
    int64_t sa[kMaxOuter], sb[kMaxOuter];  // outer strides, elements, innermost-outer first
    FastDiv shape[kMaxOuter];              // outer extents, innermost-outer first
    int n_outer;
    uint32_t rows;    // product of outer extents
    uint32_t inner;   // inner extent in elements
    uint32_t vpr;     // vector slots per row = ceil(inner / W)
    uint32_t grid_x;  // workgroups along the row; the launch is 1-D (grid y is limited to 65 535)
};

// INNER_x: 1 = dense along the inner axis, 0 = broadcast along it.
// CONST_x: operand has all outer strides zero -> identical for every row.
// VEC: 16-byte accesses legal (extent, alignment and outer strides all multiples of W).
template <typename T, typename Op, bool VEC, int INNER_A, int INNER_B, bool CONST_A, bool CONST_B, int TX, int ROWS>
__global__ __launch_bounds__(256) void row_kernel(const T *__restrict__ a, const T *__restrict__ b, T *__restrict__ out,
                                                  RowParams p) {
    constexpr int W = VEC ? VecTraits<T>::width : 1;
    typedef typename VecTraits<T>::vec_t V;
    constexpr int TY = 256 / TX;
    OpCtx<Op> ctx;
    ctx.init();
    const uint32_t tx = threadIdx.x % TX, ty = threadIdx.x / TX;
    const uint32_t bx = blockIdx.x % p.grid_x, by = blockIdx.x / p.grid_x;
    const uint32_t col = bx * TX + tx;  // vector slot within the row
    if (col >= p.vpr) return;
    const size_t col_elem = (size_t)col * W;

    T va[ROWS][W], vb[ROWS][W];
    auto load = [&](const T *base, int64_t off, int inner_mode, T (&dst)[W]) {
        if (inner_mode == 0) {
            const T s = base[off];
#pragma unroll
            for (int k = 0; k < W; ++k) dst[k] = s;
        } else if constexpr (VEC) {
            const V v = *reinterpret_cast<const V *>(base + off + col_elem);
#pragma unroll
            for (int k = 0; k < W; ++k) dst[k] = v[k];
        } else {
            dst[0] = base[off + col_elem];
        }
    };

    T ca[W], cb[W];
    if constexpr (CONST_A) load(a, 0, INNER_A, ca);
    if constexpr (CONST_B) load(b, 0, INNER_B, cb);

    const uint32_t row0 = (by * ROWS) * TY + ty;
    uint32_t rows_here = 0;
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        const uint32_t row = row0 + r * TY;
        if (row >= p.rows) break;
        ++rows_here;
        int64_t offA = 0, offB = 0;
        if constexpr (!CONST_A || !CONST_B) {
            uint32_t rem = row;
            for (int k = 0; k < p.n_outer; ++k) {
                uint32_t q, idx;
                p.shape[k].divmod(rem, q, idx);
                rem = q;
                if constexpr (!CONST_A) offA += (int64_t)idx * p.sa[k];
                if constexpr (!CONST_B) offB += (int64_t)idx * p.sb[k];
            }
        }
        if constexpr (!CONST_A) load(a, offA, INNER_A, va[r]);
        if constexpr (!CONST_B) load(b, offB, INNER_B, vb[r]);
    }
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        if (r >= (int)rows_here) break;
        const uint32_t row = row0 + r * TY;
        T res[W];
        apply_n<Op, T, W>(ctx, CONST_A ? ca : va[r], CONST_B ? cb : vb[r], res);
        T *dst = out + (size_t)row * p.inner + col_elem;
        if constexpr (VEC) {
            V v;
#pragma unroll
            for (int k = 0; k < W; ++k) v[k] = res[k];
            store_stream(reinterpret_cast<V *>(dst), v);
        } else {
            __builtin_nontemporal_store(res[0], dst);
        }
    }
}

struct GatherParams {
    int64_t sa[SMHIP_MAX_NDIM], sb[SMHIP_MAX_NDIM];  // innermost first
    FastDiv shape[SMHIP_MAX_NDIM];                    // innermost first
    int ndim;
    uint32_t n;
};

// OUTVEC: `out` is 16-byte aligned -> each lane stores one vector; otherwise
// one element per lane.
template <typename T, typename Op, bool OUTVEC>
__global__ __launch_bounds__(256) void gather_kernel(const T *__restrict__ a, const T *__restrict__ b, T *__restrict__ out,
                                                     GatherParams p) {
    constexpr int W = OUTVEC ? VecTraits<T>::width : 1;
    typedef typename VecTraits<T>::vec_t V;
    const uint32_t first = (blockIdx.x * 256u + threadIdx.x) * W;
    if (first >= p.n) return;
    T res[W];
#pragma unroll
    for (int k = 0; k < W; ++k) {
        const uint32_t linear = first + k;
        if (linear < p.n) {
            uint32_t rem = linear;
            int64_t offA = 0, offB = 0;
            for (int d = 0; d < p.ndim; ++d) {
                uint32_t q, idx;
                p.shape[d].divmod(rem, q, idx);
                rem = q;
                offA += (int64_t)idx * p.sa[d];
                offB += (int64_t)idx * p.sb[d];
            }
            res[k] = Op::apply(a[offA], b[offB]);
        }
    }
    if constexpr (OUTVEC) {
        if (first + W <= p.n) {
            V v;
#pragma unroll
            for (int k = 0; k < W; ++k) v[k] = res[k];
            store_stream(reinterpret_cast<V *>(out + first), v);
        } else {
            for (int k = 0; k < W && first + k < p.n; ++k) out[first + k] = res[k];
        }
    } else {
        out[first] = res[0];
    }
}

// ------------------------------------------------------------------- LDS kernel
struct LdsParams {
    int64_t sy[SMHIP_MAX_NDIM];       // the small operand's strides, innermost first
    FastDiv shape[SMHIP_MAX_NDIM];    // innermost first
    int ndim;
    uint32_t n, n_vec;                // outputs, and whole vectors among them
    uint32_t y_span;                  // elements of the small operand to stage
};

// x: the operand that is dense in output order (streams as vectors); y: the small one, gathered
// from its LDS copy.  SWAPPED: x is the Op's right operand.
template <typename T, typename Op, bool SWAPPED>
__global__ __launch_bounds__(256) void dense_lds_kernel(const T *__restrict__ x, const T *__restrict__ y, T *__restrict__ out,
                                                        LdsParams p) {
    typedef typename VecTraits<T>::vec_t V;
    constexpr int W = VecTraits<T>::width;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    T *ylds = reinterpret_cast<T *>(lds_raw);
    OpCtx<Op> ctx;
    ctx.init();
    for (uint32_t i = threadIdx.x; i < p.y_span; i += 256) ylds[i] = y[i];
    __syncthreads();
    auto y_at = [&](uint32_t linear) {
        uint32_t rem = linear;
        int64_t off = 0;
        for (int d = 0; d < p.ndim; ++d) {
            uint32_t q, idx;
            p.shape[d].divmod(rem, q, idx);
            rem = q;
            off += (int64_t)idx * p.sy[d];
        }
        return ylds[off];
    };
    const uint32_t stride = gridDim.x * 256u;
    for (uint32_t v = blockIdx.x * 256u + threadIdx.x; v < p.n_vec; v += stride) {
        const V xv = load_stream(reinterpret_cast<const V *>(x) + v);
        T xa[W], ya[W], r[W];
#pragma unroll
        for (int k = 0; k < W; ++k) { xa[k] = xv[k]; ya[k] = y_at(v * W + k); }
        if (SWAPPED) apply_n<Op, T, W>(ctx, ya, xa, r);
        else apply_n<Op, T, W>(ctx, xa, ya, r);
        V rv;
#pragma unroll
        for (int k = 0; k < W; ++k) rv[k] = r[k];
        store_stream(reinterpret_cast<V *>(out) + v, rv);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0)
        for (uint32_t e = p.n_vec * W; e < p.n; ++e) out[e] = SWAPPED ? Op::apply(y_at(e), x[e]) : Op::apply(x[e], y_at(e));
}

// ------------------------------------------------------------------ tile kernel
// Patch shape from tools/sweep_transpose.hip (profiles/r01_sweep_transpose.txt): 64 along p x 128 along q --
// 256-byte segments on the strided (transposed) side, 512-byte segments on the output side, consecutive
// workgroups walking q -- matched the plain add's rate; 64 x 64 was 8 % behind, p-fastest ordering 15-25 %.
constexpr int kTileP = 64, kTileQ = 128;

struct TileParams {
    // plane axes: p (operand-contiguous axis), q (output inner axis)
    uint32_t np, nq;            // extents
    int64_t a_p, a_q, b_p, b_q; // operand strides along p and q (elements)
    int64_t o_p;                // output stride along p (its q stride is 1)
    int mode_a, mode_b;         // 1: turned through LDS (operand contiguous along p); 0: read along q directly
    // remaining axes, innermost first
    int n_rest;
    FastDiv rest[SMHIP_MAX_NDIM - 2];
    int64_t a_r[SMHIP_MAX_NDIM - 2], b_r[SMHIP_MAX_NDIM - 2], o_r[SMHIP_MAX_NDIM - 2];
    uint32_t tiles_p, tiles_q;
};

// One workgroup = one 64 x 128 patch (i along p, j along q) of one slice of the remaining axes.
// VEC: every global access is a 16-byte vector (W elements) -- along p for operands turned through
// LDS, along q for direct operands and the output.  LDS tiles are stored already transposed ([i][j],
// pitch kTileQ + 1 words: the 4-byte scatter of phase 1 and the row reads of phase 2 are at most 2-way
// bank conflicted) and only as many of them exist as there are LDS-mode operands (dynamic LDS), so a
// single transposed operand leaves room for 4 workgroups per CU.  MA / MB are compile-time in the
// vector form; the element form (odd extents, pitches, bases) keeps them as runtime values.
template <typename T, typename Op, bool VEC, int MA, int MB>
__global__ __launch_bounds__(256) void tile_kernel(const T *__restrict__ a, const T *__restrict__ b, T *__restrict__ out,
                                                   TileParams p) {
    constexpr int W = VEC ? VecTraits<T>::width : 1;
    constexpr int VP = kTileP / W, VQ = kTileQ / W;  // vector slots per patch row, along p / along q
    constexpr int PITCH = kTileQ + 1;
    typedef typename VecTraits<T>::vec_t V;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    T *lds = reinterpret_cast<T *>(lds_raw);
    const int mode_a = VEC ? MA : p.mode_a, mode_b = VEC ? MB : p.mode_b;
    T *lds_a = lds, *lds_b = lds + (mode_a == 1 ? kTileP * PITCH : 0);
    OpCtx<Op> ctx;
    ctx.init();
    uint32_t bid = blockIdx.x;
    const uint32_t tq = bid % p.tiles_q; bid /= p.tiles_q;
    const uint32_t tp = bid % p.tiles_p; bid /= p.tiles_p;
    int64_t offA = 0, offB = 0, offO = 0;
    for (int k = 0; k < p.n_rest; ++k) {
        uint32_t qd, idx;
        p.rest[k].divmod(bid, qd, idx);
        bid = qd;
        offA += (int64_t)idx * p.a_r[k];
        offB += (int64_t)idx * p.b_r[k];
        offO += (int64_t)idx * p.o_r[k];
    }
    const uint32_t i0 = tp * kTileP, j0 = tq * kTileQ;
    const bool full = i0 + kTileP <= p.np && j0 + kTileQ <= p.nq;  // workgroup-uniform
    // patch origins
    const T *a0 = a + offA + (int64_t)i0 * p.a_p + (int64_t)j0 * p.a_q;
    const T *b0 = b + offB + (int64_t)i0 * p.b_p + (int64_t)j0 * p.b_q;
    T *o0 = out + offO + (int64_t)i0 * p.o_p + j0;

    // phase 1: LDS-mode operands, coalesced along p (slot ig covers i = ig*W .. +W-1 of row jl)
    auto stage = [&](const T *src0, int64_t s_q, T *tile) {
#pragma unroll
        for (int s = 0; s < kTileQ * VP / 256; ++s) {
            const uint32_t v = threadIdx.x + 256 * s, jl = v / VP, ig = v % VP;
            if (full || (i0 + ig * W < p.np && j0 + jl < p.nq)) {
                const T *g = src0 + ig * W + (int64_t)jl * s_q;
                if constexpr (VEC) {
                    const V val = *reinterpret_cast<const V *>(g);
#pragma unroll
                    for (int k = 0; k < W; ++k) tile[(ig * W + k) * PITCH + jl] = val[k];
                } else {
                    tile[ig * PITCH + jl] = *g;
                }
            }
        }
    };
    if (mode_a == 1) stage(a0, p.a_q, lds_a);
    if (mode_b == 1) stage(b0, p.b_q, lds_b);
    __syncthreads();

    // phase 2: everything coalesced along q (slot jg covers j = jg*W .. +W-1 of row il)
    auto fetch = [&](const T *src0, int64_t s_p, int64_t s_q, int mode, const T *tile, uint32_t il, uint32_t jg, T (&dst)[W]) {
        if (mode == 1) {
#pragma unroll
            for (int k = 0; k < W; ++k) dst[k] = tile[il * PITCH + jg * W + k];
        } else {
            const T *g = src0 + (int64_t)il * s_p + (int64_t)(jg * W) * s_q;
            if (VEC && s_q == 1) {
                const V val = load_stream(reinterpret_cast<const V *>(g));
#pragma unroll
                for (int k = 0; k < W; ++k) dst[k] = val[k];
            } else {
#pragma unroll
                for (int k = 0; k < W; ++k) dst[k] = g[(int64_t)k * s_q];
            }
        }
    };
#pragma unroll
    for (int s = 0; s < kTileP * VQ / 256; ++s) {
        const uint32_t v = threadIdx.x + 256 * s, il = v / VQ, jg = v % VQ;
        if (full || (i0 + il < p.np && j0 + jg * W < p.nq)) {
            T xa[W], xb[W], xr[W];
            fetch(a0, p.a_p, p.a_q, mode_a, lds_a, il, jg, xa);
            fetch(b0, p.b_p, p.b_q, mode_b, lds_b, il, jg, xb);
            apply_n<Op, T, W>(ctx, xa, xb, xr);
            T *dst = o0 + (int64_t)il * p.o_p + jg * W;
            if constexpr (VEC) {
                V val;
#pragma unroll
                for (int k = 0; k < W; ++k) val[k] = xr[k];
                store_stream(reinterpret_cast<V *>(dst), val);
            } else {
                *dst = xr[0];
            }
        }
    }
}

struct Plan {
    int ndim;
    int64_t shape[SMHIP_MAX_NDIM], sa[SMHIP_MAX_NDIM], sb[SMHIP_MAX_NDIM];
    size_t n;
};

// Drop size-1 dims; merge neighbours (i, i+1) when both operands satisfy
// stride[i] == shape[i+1] * stride[i+1] (jointly dense, or jointly broadcast).
Plan normalise(const int64_t *shape, const int64_t *sa, const int64_t *sb, int ndim) {
    Plan p{};
    p.n = 1;
    for (int i = 0; i < ndim; ++i) {
        p.n *= (size_t)shape[i];
        if (shape[i] == 1) continue;
        const int k = p.ndim;
        if (k > 0 && p.sa[k - 1] == shape[i] * sa[i] && p.sb[k - 1] == shape[i] * sb[i]) {
            p.shape[k - 1] *= shape[i];
            p.sa[k - 1] = sa[i];
            p.sb[k - 1] = sb[i];
        } else {
            p.shape[k] = shape[i];
            p.sa[k] = sa[i];
            p.sb[k] = sb[i];
            ++p.ndim;
        }
    }
    if (p.ndim == 0) {  // every dim was 1: a single element
        p.ndim = 1;
        p.shape[0] = 1;
        p.sa[0] = p.sb[0] = 1;
    }
    return p;
}

inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

template <typename T, typename Op, bool VEC, int IA, int IB, bool CA, bool CB>
int launch_row_tx(const T *a, const T *b, T *out, const RowParams &p_in, hipStream_t s) {
    constexpr int ROWS = 4;
    RowParams p = p_in;
    bool too_big = false;
    auto go = [&](auto tx_tag) {
        constexpr int TX = decltype(tx_tag)::value;
        constexpr int TY = 256 / TX;
        const size_t gx = (p.vpr + TX - 1) / TX;
        const size_t gy = ((size_t)p.rows + TY * ROWS - 1) / (TY * ROWS);
        if (gx * gy > 0x7fffffffull) { too_big = true; return; }
        p.grid_x = (uint32_t)gx;
        hipLaunchKernelGGL((row_kernel<T, Op, VEC, IA, IB, CA, CB, TX, ROWS>), dim3((unsigned)(gx * gy)), dim3(256), 0, s, a, b, out, p);
    };
    if (p.vpr > 128) go(std::integral_constant<int, 256>{});
    else if (p.vpr > 64) go(std::integral_constant<int, 128>{});
    else if (p.vpr > 32) go(std::integral_constant<int, 64>{});
    else if (p.vpr > 16) go(std::integral_constant<int, 32>{});
    else go(std::integral_constant<int, 16>{});
    if (too_big) return fail(SMHIP_ERR_UNSUPPORTED, "row kernel: more than 2^31 workgroups");
    SMHIP_LAUNCH_CHECK("row_kernel");
    return SMHIP_OK;
}

template <typename T, typename Op, bool VEC>
int launch_row(const T *a, const T *b, T *out, const RowParams &p, int ia, int ib, bool ca, bool cb, hipStream_t s) {
    // inner (1,1): neither, one or the other operand constant over rows; (1,0)/(0,1): the
    // broadcast side may additionally be row-constant only together with being a scalar,
    // which normalise() has already folded away -- so CONST applies to dense sides only.
#define ROW(IA, IB, CA, CB) return launch_row_tx<T, Op, VEC, IA, IB, CA, CB>(a, b, out, p, s)
    if (ia == 1 && ib == 1) {
        if (cb && !ca) ROW(1, 1, false, true);
        if (ca && !cb) ROW(1, 1, true, false);
        ROW(1, 1, false, false);
    }
    if (ia == 1 && ib == 0) {
        if (ca) ROW(1, 0, true, false);
        ROW(1, 0, false, false);
    }
    if (cb) ROW(0, 1, false, true);
    ROW(0, 1, false, false);
#undef ROW
}

template <typename T, typename Op>
int run_broadcast(const void *a_, const void *b_, void *out_, const Plan &pl, hipStream_t s) {
    constexpr int W = VecTraits<T>::width;
    const T *a = static_cast<const T *>(a_), *b = static_cast<const T *>(b_);
    T *out = static_cast<T *>(out_);
    const int nd = pl.ndim;
    const int64_t ia = pl.sa[nd - 1], ib = pl.sb[nd - 1];
    const int64_t inner = pl.shape[nd - 1];
    const size_t rows = pl.n / (size_t)inner;

    const bool row_ok = (ia == 0 || ia == 1) && (ib == 0 || ib == 1) && (ia | ib) != 0 && inner >= 16 &&
                        rows < 0x7fffffffull && inner < 0x7fffffffll;
    if (row_ok) {
        RowParams p{};
        p.n_outer = nd - 1;
        p.rows = (uint32_t)rows;
        p.inner = (uint32_t)inner;
        // a side that is broadcast along the inner axis is read as scalars: no alignment demand
        bool ca = true, cb = true;
        bool vec = (inner % W == 0) && aligned16(out) && (ia == 0 || aligned16(a)) && (ib == 0 || aligned16(b));
        for (int k = 0; k < nd - 1; ++k) {  // innermost-outer first
            const int src = nd - 2 - k;
            p.shape[k] = FastDiv((uint32_t)pl.shape[src]);
            p.sa[k] = pl.sa[src];
            p.sb[k] = pl.sb[src];
            ca &= pl.sa[src] == 0;
            cb &= pl.sb[src] == 0;
            if (ia == 1 && pl.sa[src] % W) vec = false;
            if (ib == 1 && pl.sb[src] % W) vec = false;
        }
        if (vec) {
            p.vpr = (uint32_t)(inner / W);
            return launch_row<T, Op, true>(a, b, out, p, (int)ia, (int)ib, ca, cb, s);
        }
        p.vpr = (uint32_t)inner;
        return launch_row<T, Op, false>(a, b, out, p, (int)ia, (int)ib, ca, cb, s);
    }

    // One operand dense in output order, the other small: stage the small one in LDS.
    if (pl.n < 0x7fffffffull && aligned16(out)) {
        auto dense_in_output_order = [&](const int64_t *st) {
            int64_t expect = 1;
            for (int d = nd - 1; d >= 0; --d) {
                if (st[d] != expect) return false;
                expect *= pl.shape[d];
            }
            return true;
        };
        auto span_of = [&](const int64_t *st) {
            int64_t last = 0;
            for (int d = 0; d < nd; ++d) last += (pl.shape[d] - 1) * st[d];
            return last + 1;
        };
        constexpr int64_t kLdsBytes = 32 << 10;
        const bool a_dense = dense_in_output_order(pl.sa), b_dense = dense_in_output_order(pl.sb);
        const int64_t span_a = span_of(pl.sa), span_b = span_of(pl.sb);
        int pick = -1;  // 0: a streams, b staged;  1: b streams, a staged
        if (a_dense && !b_dense && aligned16(a) && span_b * (int64_t)sizeof(T) <= kLdsBytes) pick = 0;
        else if (b_dense && !a_dense && aligned16(b) && span_a * (int64_t)sizeof(T) <= kLdsBytes) pick = 1;
        if (pick >= 0) {
            LdsParams lp{};
            lp.ndim = nd;
            lp.n = (uint32_t)pl.n;
            lp.n_vec = (uint32_t)(pl.n / W);
            lp.y_span = (uint32_t)(pick == 0 ? span_b : span_a);
            for (int d = 0; d < nd; ++d) {
                const int src = nd - 1 - d;
                lp.shape[d] = FastDiv((uint32_t)pl.shape[src]);
                lp.sy[d] = pick == 0 ? pl.sb[src] : pl.sa[src];
            }
            const size_t want = ((size_t)lp.n_vec + 255) / 256;
            const size_t cap = (size_t)compute_units() * 8;
            const unsigned grid = (unsigned)(want < cap ? (want ? want : 1) : cap);
            const size_t lds = (size_t)lp.y_span * sizeof(T);
            if (pick == 0) hipLaunchKernelGGL((dense_lds_kernel<T, Op, false>), dim3(grid), dim3(256), lds, s, a, b, out, lp);
            else hipLaunchKernelGGL((dense_lds_kernel<T, Op, true>), dim3(grid), dim3(256), lds, s, b, a, out, lp);
            SMHIP_LAUNCH_CHECK("dense_lds_kernel");
            return SMHIP_OK;
        }
    }

    // An operand that is contiguous along some OTHER axis p (a transposed / permuted view): tile the
    // (p, inner) plane through LDS.  Taken when both plane extents are worth a 64 x 64 patch.
    if (nd >= 2 && inner >= 16) {
        auto contiguous_axis = [&](const int64_t *st) {
            for (int d = nd - 2; d >= 0; --d)
                if (st[d] == 1 && pl.shape[d] >= 16) return d;
            return -1;
        };
        const bool a_strided = ia != 0 && ia != 1, b_strided = ib != 0 && ib != 1;
        int pa = a_strided ? contiguous_axis(pl.sa) : -1, pb = b_strided ? contiguous_axis(pl.sb) : -1;
        const int paxis = pa >= 0 ? pa : pb;
        if (paxis >= 0) {
            TileParams t{};
            t.np = (uint32_t)pl.shape[paxis];
            t.nq = (uint32_t)inner;
            t.a_p = pl.sa[paxis]; t.a_q = ia;
            t.b_p = pl.sb[paxis]; t.b_q = ib;
            t.mode_a = (a_strided && pl.sa[paxis] == 1) ? 1 : 0;
            t.mode_b = (b_strided && pl.sb[paxis] == 1) ? 1 : 0;
            // dense output strides
            int64_t ostride[SMHIP_MAX_NDIM];
            ostride[nd - 1] = 1;
            for (int d = nd - 2; d >= 0; --d) ostride[d] = ostride[d + 1] * pl.shape[d + 1];
            t.o_p = ostride[paxis];
            size_t slices = 1;
            for (int d = nd - 2; d >= 0; --d) {  // innermost remaining axis first
                if (d == paxis) continue;
                t.rest[t.n_rest] = FastDiv((uint32_t)pl.shape[d]);
                t.a_r[t.n_rest] = pl.sa[d];
                t.b_r[t.n_rest] = pl.sb[d];
                t.o_r[t.n_rest] = ostride[d];
                ++t.n_rest;
                slices *= (size_t)pl.shape[d];
            }
            t.tiles_p = (t.np + kTileP - 1) / kTileP;
            t.tiles_q = (t.nq + kTileQ - 1) / kTileQ;
            const size_t blocks = slices * t.tiles_p * t.tiles_q;
            if (blocks < 0x7fffffffull && pl.shape[paxis] < 0x7fffffffll && inner < 0x7fffffffll) {
                // 16-byte accesses need every vector to start on a 16-byte boundary: bases, both plane
                // extents, and every stride that moves a vector's start (all but the unit ones)
                bool vec = aligned16(a) && aligned16(b) && aligned16(out) && t.np % W == 0 && t.nq % W == 0;
                auto mult = [&](int64_t v) { return v % W == 0; };
                if (t.mode_a == 1) vec &= mult(t.a_q); else vec &= mult(t.a_p);
                if (t.mode_b == 1) vec &= mult(t.b_q); else vec &= mult(t.b_p);
                vec &= mult(t.o_p);
                for (int k = 0; k < t.n_rest; ++k) vec &= mult(t.a_r[k]) && mult(t.b_r[k]) && mult(t.o_r[k]);
                const size_t lds_bytes = (size_t)(t.mode_a + t.mode_b) * kTileP * (kTileQ + 1) * sizeof(T);
                const dim3 grid((unsigned)blocks), block(256);
                if (!vec) hipLaunchKernelGGL((tile_kernel<T, Op, false, 0, 0>), grid, block, lds_bytes, s, a, b, out, t);
                else if (t.mode_a == 1 && t.mode_b == 1) hipLaunchKernelGGL((tile_kernel<T, Op, true, 1, 1>), grid, block, lds_bytes, s, a, b, out, t);
                else if (t.mode_a == 1) hipLaunchKernelGGL((tile_kernel<T, Op, true, 1, 0>), grid, block, lds_bytes, s, a, b, out, t);
                else hipLaunchKernelGGL((tile_kernel<T, Op, true, 0, 1>), grid, block, lds_bytes, s, a, b, out, t);
                SMHIP_LAUNCH_CHECK("tile_kernel");
                return SMHIP_OK;
            }
        }
    }

    if (pl.n >= 0x7fffffffull)
        return fail(SMHIP_ERR_UNSUPPORTED, "gather path limited to < 2^31 elements (got %zu)", pl.n);
    GatherParams g{};
    g.ndim = nd;
    g.n = (uint32_t)pl.n;
    for (int d = 0; d < nd; ++d) {
        const int src = nd - 1 - d;
        g.shape[d] = FastDiv((uint32_t)pl.shape[src]);
        g.sa[d] = pl.sa[src];
        g.sb[d] = pl.sb[src];
    }
    if (aligned16(out)) {
        const unsigned grid = (unsigned)(((pl.n + W - 1) / W + 255) / 256);
        hipLaunchKernelGGL((gather_kernel<T, Op, true>), dim3(grid), dim3(256), 0, s, a, b, out, g);
    } else {
        const unsigned grid = (unsigned)((pl.n + 255) / 256);
        hipLaunchKernelGGL((gather_kernel<T, Op, false>), dim3(grid), dim3(256), 0, s, a, b, out, g);
    }
    SMHIP_LAUNCH_CHECK("gather_kernel");
    return SMHIP_OK;
}


// Kernels below index one launch with 32 bits.  A problem of >= 2^31 elements (a 288 GB device holds 2^36
// floats) is cut along its outermost dimension into pieces under that limit -- each still gigabytes, so
// the extra launches cost nothing -- and a piece of one outer index recurses into the next dimension.
constexpr size_t kMaxLaunchElems = 0x7fffffffull;

int launch_plan(int op, int dtype, const void *a, const void *b, void *out, const Plan &pl, hipStream_t s) {
    if (pl.ndim == 1) {
        // calculate.h:10-11's fast path, decided on the normalised problem
        if (pl.sa[0] == 1 && pl.sb[0] == 1) return launch_contiguous(op, dtype, a, b, out, pl.n, s);
        if (pl.sa[0] == 1 && pl.sb[0] == 0) return launch_array_devscalar(op, dtype, a, b, pl.n, out, false, s);
        if (pl.sa[0] == 0 && pl.sb[0] == 1) return launch_array_devscalar(op, dtype, b, a, pl.n, out, true, s);
    }
    if (pl.n >= kMaxLaunchElems) {
        const size_t esz = dtype_size(dtype);
        const size_t slice = pl.n / (size_t)pl.shape[0];  // elements per index of the outermost dimension
        size_t per = slice >= kMaxLaunchElems ? 1 : (kMaxLaunchElems - 1) / slice;
        if (per > 4) per &= ~(size_t)3;  // keep piece starts 16-byte aligned where the strides allow
        int64_t shape[SMHIP_MAX_NDIM];
        for (int d = 0; d < pl.ndim; ++d) shape[d] = pl.shape[d];
        for (size_t i0 = 0; i0 < (size_t)pl.shape[0]; i0 += per) {
            const size_t left = (size_t)pl.shape[0] - i0;
            shape[0] = (int64_t)(left < per ? left : per);
            const Plan sub = normalise(shape, pl.sa, pl.sb, pl.ndim);
            const char *pa = static_cast<const char *>(a) + (int64_t)i0 * pl.sa[0] * (int64_t)esz;
            const char *pb = static_cast<const char *>(b) + (int64_t)i0 * pl.sb[0] * (int64_t)esz;
            char *po = static_cast<char *>(out) + i0 * slice * esz;
            if (int rc = launch_plan(op, dtype, pa, pb, po, sub, s)) return rc;
        }
        return SMHIP_OK;
    }
#define SMHIP_DISPATCH_OP(T)                                                                   \
    switch (op) {                                                                              \
        case SMHIP_OP_ADD: return run_broadcast<T, AddOp<T>>(a, b, out, pl, s);                \
        case SMHIP_OP_SUB: return run_broadcast<T, SubtractOp<T>>(a, b, out, pl, s);           \
        case SMHIP_OP_MUL: return run_broadcast<T, MultiplyOp<T>>(a, b, out, pl, s);           \
        case SMHIP_OP_DIV: return run_broadcast<T, DivideOp<T>>(a, b, out, pl, s);             \
        case SMHIP_OP_POW: return run_broadcast<T, PowOp<T>>(a, b, out, pl, s);                \
        case SMHIP_OP_LEFT: return run_broadcast<T, LeftOp<T>>(a, b, out, pl, s);                \
    }                                                                                          \
    break;
    switch (dtype) {
        case SMHIP_F32: SMHIP_DISPATCH_OP(float)
        case SMHIP_F64: SMHIP_DISPATCH_OP(double)
        case SMHIP_I32: SMHIP_DISPATCH_OP(int32_t)
        case SMHIP_I64: SMHIP_DISPATCH_OP(int64_t)
    }
#undef SMHIP_DISPATCH_OP
    return fail(SMHIP_ERR_INVALID, "elementwise: bad op %d / dtype %d", op, dtype);
}

}  // namespace

int launch_broadcast(int op, int dtype, const void *a, const int64_t *sa, const void *b, const int64_t *sb,
                     const int64_t *shape, int ndim, void *out, hipStream_t s) {
    const Plan pl = normalise(shape, sa, sb, ndim);
    if (pl.n == 0) return SMHIP_OK;
    return launch_plan(op, dtype, a, b, out, pl, s);
}

}  // namespace smhip
