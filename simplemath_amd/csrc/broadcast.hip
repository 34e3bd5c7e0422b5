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
//                  with four 16-byte vectors in flight per lane; one fast-division chain per vector, then
//                  increment-and-carry, feeds the LDS reads.
//   gather kernel  whatever is left (tiny extents, irregular strides): one fast-division chain per lane,
//                  then increment-and-carry; W consecutive outputs per lane with a vector store when the
//                  inner strides are 0/1, one output per lane when an operand is strided along the inner
//                  axis (each load instruction then stays inside a few cache lines).
// Every 16-byte access is only element-aligned (VecTraits, ops.hip.h): bases, pitches and row extents are
// unconstrained, there are no per-element fallbacks for alignment.
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
// Every lane owns one 16-byte slot of W elements of a row; accesses are element-aligned vectors (any base,
// any pitch).  The last slot of a row whose extent is not a multiple of W is handled element by element.
template <typename T, typename Op, int INNER_A, int INNER_B, bool CONST_A, bool CONST_B, int TX, int ROWS>
__global__ __launch_bounds__(256) void row_kernel(const T *__restrict__ a, const T *__restrict__ b, T *__restrict__ out,
                                                  RowParams p) {
    constexpr int W = VecTraits<T>::width;
    typedef typename VecTraits<T>::vec_t V;
    constexpr int TY = 256 / TX;
    OpCtx<Op> ctx;
    ctx.init();
    const uint32_t tx = threadIdx.x % TX, ty = threadIdx.x / TX;
    const uint32_t bx = blockIdx.x % p.grid_x, by = blockIdx.x / p.grid_x;
    const uint32_t col = bx * TX + tx;  // vector slot within the row
    if (col >= p.vpr) return;
    const uint32_t col_elem = col * W;
    const bool whole = col_elem + W <= p.inner;          // false only for a row's ragged last slot
    const int count = whole ? W : (int)(p.inner - col_elem);

    T va[ROWS][W], vb[ROWS][W];
    // streamed: the operand changes from row to row (read once, non-temporal like the contiguous kernels);
    // a row-constant operand is read through the caches
    auto load = [&](const T *base, int64_t off, int inner_mode, bool streamed, T (&dst)[W]) {
        if (inner_mode == 0) {
            const T s = base[off];
#pragma unroll
            for (int k = 0; k < W; ++k) dst[k] = s;
        } else if (whole) {
            const V *src = reinterpret_cast<const V *>(base + off + col_elem);
            V v;
            if (streamed) v = load_stream(src);
            else v = *src;
#pragma unroll
            for (int k = 0; k < W; ++k) dst[k] = v[k];
        } else {
#pragma unroll
            for (int k = 0; k < W; ++k) dst[k] = k < count ? base[off + col_elem + k] : base[off + col_elem];
        }
    };

    T ca[W], cb[W];
    if constexpr (CONST_A) load(a, 0, INNER_A, false, ca);
    if constexpr (CONST_B) load(b, 0, INNER_B, false, cb);

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
            for (int k = 0; k < p.n_outer - 1; ++k) {
                uint32_t q, idx;
                p.shape[k].divmod(rem, q, idx);
                rem = q;
                if constexpr (!CONST_A) offA += (int64_t)idx * p.sa[k];
                if constexpr (!CONST_B) offB += (int64_t)idx * p.sb[k];
            }
            // the outermost axis needs no division: what is left IS its index
            if constexpr (!CONST_A) offA += (int64_t)rem * p.sa[p.n_outer - 1];
            if constexpr (!CONST_B) offB += (int64_t)rem * p.sb[p.n_outer - 1];
        }
        if constexpr (!CONST_A) load(a, offA, INNER_A, INNER_B == 1, va[r]);
        if constexpr (!CONST_B) load(b, offB, INNER_B, INNER_A == 1, vb[r]);
    }
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        if (r >= (int)rows_here) break;
        const uint32_t row = row0 + r * TY;
        T res[W];
        apply_n<Op, T, W>(ctx, CONST_A ? ca : va[r], CONST_B ? cb : vb[r], res);
        T *dst = out + (size_t)row * p.inner + col_elem;
        if (whole) {
            V v;
#pragma unroll
            for (int k = 0; k < W; ++k) v[k] = res[k];
            store_stream(reinterpret_cast<V *>(dst), v);
        } else {
#pragma unroll
            for (int k = 0; k < W; ++k)
                if (k < count) dst[k] = res[k];
        }
    }
}

struct GatherParams {
    int64_t sa[SMHIP_MAX_NDIM], sb[SMHIP_MAX_NDIM];  // innermost first
    FastDiv shape[SMHIP_MAX_NDIM];                    // innermost first
    int ndim;
    uint32_t n;
};

// W consecutive outputs per lane: the store is one (element-aligned) 16-byte vector.  The N-D index of the lane's
// first output comes from one fast-division chain (none for the outermost axis); the other W-1 follow by
// increment-and-carry, which is full-rate integer work instead of W more chains of quarter-rate mul-hi / mul-lo.
template <typename T, typename Op, int W>
__global__ __launch_bounds__(256) void gather_kernel(const T *__restrict__ a, const T *__restrict__ b, T *__restrict__ out,
                                                     GatherParams p) {
    constexpr int D = SMHIP_MAX_NDIM;
    typedef typename VecTraits<T>::vec_t V;
    const uint32_t first = (blockIdx.x * 256u + threadIdx.x) * W;
    if (first >= p.n) return;
    uint32_t idx[D], rem = first;
    int64_t offA = 0, offB = 0;
#pragma unroll
    for (int d = 0; d < D; ++d) {
        idx[d] = 0;
        if (d < p.ndim) {
            if (d == p.ndim - 1) {
                idx[d] = rem;
            } else {
                uint32_t q;
                p.shape[d].divmod(rem, q, idx[d]);
                rem = q;
            }
            offA += (int64_t)idx[d] * p.sa[d];
            offB += (int64_t)idx[d] * p.sb[d];
        }
    }
    T xa[W], xb[W], res[W];
    const int count = first + W <= p.n ? W : (int)(p.n - first);
#pragma unroll
    for (int k = 0; k < W; ++k) {
        if (k < count) {
            xa[k] = a[offA];
            xb[k] = b[offB];
        } else {
            xa[k] = xa[0];
            xb[k] = xb[0];
        }
        bool carry = k + 1 < count;
#pragma unroll
        for (int d = 0; d < D; ++d) {
            if (d < p.ndim && carry) {
                offA += p.sa[d];
                offB += p.sb[d];
                if (++idx[d] == p.shape[d].d && d != p.ndim - 1) {
                    idx[d] = 0;
                    offA -= (int64_t)p.shape[d].d * p.sa[d];
                    offB -= (int64_t)p.shape[d].d * p.sb[d];
                } else {
                    carry = false;
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < W; ++k) res[k] = Op::apply(xa[k], xb[k]);
    if constexpr (W == 1) {
        out[first] = res[0];
    } else if (count == W) {
        V v;
#pragma unroll
        for (int k = 0; k < W; ++k) v[k] = res[k];
        store_stream(reinterpret_cast<V *>(out + first), v);
    } else {
        for (int k = 0; k < count; ++k) out[first + k] = res[k];
    }
}

// ------------------------------------------------------------------- LDS kernel
struct LdsParams {
    uint32_t sy[SMHIP_MAX_NDIM];      // the small operand's strides, innermost first (its span is <= 8192 elements)
    uint32_t rewind[SMHIP_MAX_NDIM];  // extent * stride: what a wrap of that axis takes back off the offset
    FastDiv shape[SMHIP_MAX_NDIM];    // innermost first
    int ndim;
    uint32_t n, n_vec;                // outputs, and whole vectors among them
    uint32_t y_span;                  // elements of the small operand to stage
};

// x: the operand that is dense in output order (streams as vectors); y: the small one, gathered
// from its LDS copy.  SWAPPED: x is the Op's right operand.  Each lane keeps U vectors of x in flight
// (all loads issued before any arithmetic).  The N-D index of a vector's first element comes from one
// fast-division chain (none for the outermost axis); its other W-1 elements follow by increment-and-carry,
// which is full-rate integer work instead of W more chains of quarter-rate mul-hi / mul-lo.
template <typename T, typename Op, bool SWAPPED, int U>
__global__ __launch_bounds__(256) void dense_lds_kernel(const T *__restrict__ x, const T *__restrict__ y, T *__restrict__ out,
                                                        LdsParams p) {
    typedef typename VecTraits<T>::vec_t V;
    constexpr int W = VecTraits<T>::width;
    constexpr int D = SMHIP_MAX_NDIM;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    T *ylds = reinterpret_cast<T *>(lds_raw);
    OpCtx<Op> ctx;
    ctx.init();
    for (uint32_t i = threadIdx.x; i < p.y_span; i += 256) ylds[i] = y[i];
    __syncthreads();
    auto unravel = [&](uint32_t linear, uint32_t (&idx)[D]) {
        uint32_t off = 0, rem = linear;
#pragma unroll
        for (int d = 0; d < D; ++d) {
            idx[d] = 0;
            if (d < p.ndim) {
                if (d == p.ndim - 1) {
                    idx[d] = rem;
                } else {
                    uint32_t q;
                    p.shape[d].divmod(rem, q, idx[d]);
                    rem = q;
                }
                off += idx[d] * p.sy[d];
            }
        }
        return off;
    };
    // the small operand's elements for W consecutive outputs starting at `linear` (all W must exist)
    auto y_vec = [&](uint32_t linear, T (&dst)[W]) {
        uint32_t idx[D];
        uint32_t off = unravel(linear, idx);
        dst[0] = ylds[off];
#pragma unroll
        for (int k = 1; k < W; ++k) {
            bool carry = true;
#pragma unroll
            for (int d = 0; d < D; ++d) {
                if (d < p.ndim && carry) {
                    off += p.sy[d];
                    if (++idx[d] == p.shape[d].d && d != p.ndim - 1) {
                        idx[d] = 0;
                        off -= p.rewind[d];
                    } else {
                        carry = false;
                    }
                }
            }
            dst[k] = ylds[off];
        }
    };
    constexpr uint32_t kChunk = 256u * U;
    for (uint32_t base = blockIdx.x * kChunk; base < p.n_vec; base += gridDim.x * kChunk) {
        V xv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t v = base + u * 256u + threadIdx.x;
            if (v < p.n_vec) xv[u] = load_stream(reinterpret_cast<const V *>(x) + v);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t v = base + u * 256u + threadIdx.x;
            if (v < p.n_vec) {
                T xa[W], ya[W], r[W];
#pragma unroll
                for (int k = 0; k < W; ++k) xa[k] = xv[u][k];
                y_vec(v * W, ya);
                if (SWAPPED) apply_n<Op, T, W>(ctx, ya, xa, r);
                else apply_n<Op, T, W>(ctx, xa, ya, r);
                V rv;
#pragma unroll
                for (int k = 0; k < W; ++k) rv[k] = r[k];
                store_stream(reinterpret_cast<V *>(out) + v, rv);
            }
        }
    }
    if (blockIdx.x == 0 && threadIdx.x < p.n - p.n_vec * W) {
        const uint32_t e = p.n_vec * W + threadIdx.x;
        uint32_t idx[D];
        const T ye = ylds[unravel(e, idx)];
        out[e] = SWAPPED ? Op::apply(ye, x[e]) : Op::apply(x[e], ye);
    }
}

// ------------------------------------------------------------------ tile kernel
// Patch shape from tools/sweep_transpose.hip (profiles/r01_sweep_transpose.txt): 64 along p x 128 along q for
// 4-byte elements -- 256-byte segments on the strided (transposed) side, 512-byte segments on the output side,
// consecutive workgroups walking q -- matched the plain add's rate; 64 x 64 was 8 % behind, p-fastest ordering
// 15-25 %.  8-byte elements take 64 x 64: the same 512-byte output segments and the same 33 KiB of LDS per tile,
// so four workgroups still fit a CU (64 x 128 doubles left room for two: 57 % of peak instead of 80 %).
constexpr int kTileP = 64;
template <typename T> constexpr int tile_q() { return 512 / (int)sizeof(T); }

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

// One workgroup = one 64 x TQ patch (i along p, j along q) of one slice of the remaining axes.
// VEC: every global access is a 16-byte vector (W elements) -- along p for operands turned through
// LDS, along q for direct operands and the output.  LDS tiles are stored already transposed ([i][j]) in
// a bank-conflict-free layout (`at` below; measured: profiles/r01_pmc_lds_tile_kernel.txt).  There is ONE tile: with a single LDS-mode operand it holds that operand; when both
// operands are contiguous along p (a.T op b.T) phase 1 loads both coalesced, applies the Op there and
// stages the RESULT, so phase 2 is a pure transposed write-out.  MA / MB are compile-time in the
// vector form; the element form (odd extents, pitches, bases) keeps them as runtime values.
template <typename T, typename Op, bool VEC, int MA, int MB>
__global__ __launch_bounds__(256) void tile_kernel(const T *__restrict__ a, const T *__restrict__ b, T *__restrict__ out,
                                                   TileParams p) {
    constexpr int W = VEC ? VecTraits<T>::width : 1;
    constexpr int TQ = tile_q<T>();
    constexpr int VP = kTileP / W, VQ = TQ / W;  // vector slots per patch row, along p / along q
    // LDS layout of element (i, j): 4-byte types get a skewed layout (one pad word per 32 columns, two per 32 rows,
    // odd pitch) that makes both the 4-byte scatter of phase 1 and the stride-4 reads of phase 2 hit 32 distinct
    // banks per 32-lane group; 8-byte types keep the plain padded pitch.
    constexpr bool SKEW = sizeof(T) == 4;
    constexpr int PITCH = SKEW ? TQ + 5 : TQ + 1;
    auto at = [](uint32_t i, uint32_t j) -> uint32_t { return SKEW ? i * PITCH + j + (j >> 5) + ((i >> 5) << 1) : i * PITCH + j; };
    typedef typename VecTraits<T>::vec_t V;
    __shared__ T tile[kTileP * PITCH];
    const int mode_a = VEC ? MA : p.mode_a, mode_b = VEC ? MB : p.mode_b;
    const bool both = mode_a == 1 && mode_b == 1;
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
    const uint32_t i0 = tp * kTileP, j0 = tq * TQ;
    const bool full = i0 + kTileP <= p.np && j0 + TQ <= p.nq;  // workgroup-uniform
    // patch origins
    const T *a0 = a + offA + (int64_t)i0 * p.a_p + (int64_t)j0 * p.a_q;
    const T *b0 = b + offB + (int64_t)i0 * p.b_p + (int64_t)j0 * p.b_q;
    T *o0 = out + offO + (int64_t)i0 * p.o_p + j0;

    // phase 1: LDS-mode operands, coalesced along p (slot ig covers i = ig*W .. +W-1 of row jl)
    auto along_p = [&](const T *src0, int64_t s_q, uint32_t jl, uint32_t ig, T (&dst)[W]) {
        const T *g = src0 + ig * W + (int64_t)jl * s_q;
        if constexpr (VEC) {
            // two turned streams and no reuse: nt is worth 8 % there; with one it costs (tools/sweep_transpose.hip)
            const V val = both ? load_stream(reinterpret_cast<const V *>(g)) : *reinterpret_cast<const V *>(g);
#pragma unroll
            for (int k = 0; k < W; ++k) dst[k] = val[k];
        } else {
            dst[0] = *g;
        }
    };
#pragma unroll
    for (int s = 0; s < TQ * VP / 256; ++s) {
        const uint32_t v = threadIdx.x + 256 * s, jl = v / VP, ig = v % VP;
        if (full || (i0 + ig * W < p.np && j0 + jl < p.nq)) {
            T x[W];
            if (both) {
                T xa[W], xb[W];
                along_p(a0, p.a_q, jl, ig, xa);
                along_p(b0, p.b_q, jl, ig, xb);
                apply_n<Op, T, W>(ctx, xa, xb, x);
            } else if (mode_a == 1) {
                along_p(a0, p.a_q, jl, ig, x);
            } else {
                along_p(b0, p.b_q, jl, ig, x);
            }
#pragma unroll
            for (int k = 0; k < W; ++k) tile[at(ig * W + k, jl)] = x[k];
        }
    }
    __syncthreads();

    // phase 2: everything coalesced along q (slot jg covers j = jg*W .. +W-1 of row il)
    auto along_q = [&](const T *src0, int64_t s_p, int64_t s_q, uint32_t il, uint32_t jg, T (&dst)[W]) {
        const T *g = src0 + (int64_t)il * s_p + (int64_t)(jg * W) * s_q;
        if (VEC && s_q == 1) {
            const V val = load_stream(reinterpret_cast<const V *>(g));
#pragma unroll
            for (int k = 0; k < W; ++k) dst[k] = val[k];
        } else {
#pragma unroll
            for (int k = 0; k < W; ++k) dst[k] = g[(int64_t)k * s_q];
        }
    };
#pragma unroll
    for (int s = 0; s < kTileP * VQ / 256; ++s) {
        const uint32_t v = threadIdx.x + 256 * s, il = v / VQ, jg = v % VQ;
        if (full || (i0 + il < p.np && j0 + jg * W < p.nq)) {
            T xt[W], xr[W];
#pragma unroll
            for (int k = 0; k < W; ++k) xt[k] = tile[at(il, jg * W + k)];
            if (both) {
#pragma unroll
                for (int k = 0; k < W; ++k) xr[k] = xt[k];
            } else if (mode_a == 1) {
                T xb[W];
                along_q(b0, p.b_p, p.b_q, il, jg, xb);
                apply_n<Op, T, W>(ctx, xt, xb, xr);
            } else {
                T xa[W];
                along_q(a0, p.a_p, p.a_q, il, jg, xa);
                apply_n<Op, T, W>(ctx, xa, xt, xr);
            }
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

template <typename T, typename Op, int IA, int IB, bool CA, bool CB>
int launch_row_tx(const T *a, const T *b, T *out, const RowParams &p_in, hipStream_t s) {
    RowParams p = p_in;
    bool too_big = false;
    auto go2 = [&](auto tx_tag, auto rows_tag) {
        constexpr int TX = decltype(tx_tag)::value;
        constexpr int ROWS = decltype(rows_tag)::value;
        constexpr int TY = 256 / TX;
        const size_t gx = (p.vpr + TX - 1) / TX;
        const size_t gy = ((size_t)p.rows + TY * ROWS - 1) / (TY * ROWS);
        if (gx * gy > 0x7fffffffull) { too_big = true; return; }
        p.grid_x = (uint32_t)gx;
        hipLaunchKernelGGL((row_kernel<T, Op, IA, IB, CA, CB, TX, ROWS>), dim3((unsigned)(gx * gy)), dim3(256), 0, s, a, b, out, p);
    };
    // Rows per lane (tools/bcast_matrix.py, profiles/r01_bcast_matrix.txt): two when one side is a row-constant or a
    // per-row scalar -- two independent 16-byte loads in flight per lane, 83 % of peak on config 3 against 80 % with
    // one or four; one when both operands stream (three full streams behave like the contiguous kernel: 80 % vs
    // 74-77 %) and for pow, whose arithmetic already overlaps the next lane's loads.
    constexpr bool kThreeStreams = IA == 1 && IB == 1 && !CA && !CB;
    constexpr int kRows = (kThreeStreams || std::is_same<Op, PowOp<T>>::value) ? 1 : 2;
    auto go = [&](auto tx_tag) { go2(tx_tag, std::integral_constant<int, kRows>{}); };
    if (p.vpr > 128) go(std::integral_constant<int, 256>{});
    else if (p.vpr > 64) go(std::integral_constant<int, 128>{});
    else if (p.vpr > 32) go(std::integral_constant<int, 64>{});
    else if (p.vpr > 16) go(std::integral_constant<int, 32>{});
    else go(std::integral_constant<int, 16>{});
    if (too_big) return fail(SMHIP_ERR_UNSUPPORTED, "row kernel: more than 2^31 workgroups");
    SMHIP_LAUNCH_CHECK("row_kernel");
    return SMHIP_OK;
}

template <typename T, typename Op>
int launch_row(const T *a, const T *b, T *out, const RowParams &p, int ia, int ib, bool ca, bool cb, hipStream_t s) {
    // inner (1,1): neither, one or the other operand constant over rows; (1,0)/(0,1): the
    // broadcast side may additionally be row-constant only together with being a scalar,
    // which normalise() has already folded away -- so CONST applies to dense sides only.
#define ROW(IA, IB, CA, CB) return launch_row_tx<T, Op, IA, IB, CA, CB>(a, b, out, p, s)
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
        bool ca = true, cb = true;
        for (int k = 0; k < nd - 1; ++k) {  // innermost-outer first
            const int src = nd - 2 - k;
            p.shape[k] = FastDiv((uint32_t)pl.shape[src]);
            p.sa[k] = pl.sa[src];
            p.sb[k] = pl.sb[src];
            ca &= pl.sa[src] == 0;
            cb &= pl.sb[src] == 0;
        }
        p.vpr = (uint32_t)((inner + W - 1) / W);
        return launch_row<T, Op>(a, b, out, p, (int)ia, (int)ib, ca, cb, s);
    }

    // One operand dense in output order, the other small: stage the small one in LDS.
    if (pl.n < 0x7fffffffull) {
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
        if (a_dense && !b_dense && span_b * (int64_t)sizeof(T) <= kLdsBytes) pick = 0;
        else if (b_dense && !a_dense && span_a * (int64_t)sizeof(T) <= kLdsBytes) pick = 1;
        if (pick >= 0) {
            LdsParams lp{};
            lp.ndim = nd;
            lp.n = (uint32_t)pl.n;
            lp.n_vec = (uint32_t)(pl.n / W);
            lp.y_span = (uint32_t)(pick == 0 ? span_b : span_a);
            for (int d = 0; d < nd; ++d) {
                const int src = nd - 1 - d;
                lp.shape[d] = FastDiv((uint32_t)pl.shape[src]);
                lp.sy[d] = (uint32_t)(pick == 0 ? pl.sb[src] : pl.sa[src]);
                lp.rewind[d] = (uint32_t)pl.shape[src] * lp.sy[d];
            }
            // a small staged operand (<= 4 KiB) is re-staged by every workgroup of a one-shot launch, which lets
            // the hardware dispatcher balance the stream; a larger one is staged once per persistent workgroup
            const size_t lds = (size_t)lp.y_span * sizeof(T);
            constexpr int U = 4;  // vectors in flight per lane: 75-78 % of peak; 1: 55-65 %, 2: 73-80 %, 8: 71-74 % (tools/bcast_matrix.py)
            const size_t want = ((size_t)lp.n_vec + 256 * U - 1) / (256 * U);
            const size_t cap = lds <= 4096 ? want : (size_t)compute_units() * 8;
            size_t blocks = want < cap ? want : cap;
            if (blocks == 0) blocks = 1;  // fewer than W elements: the tail lanes of one workgroup do them
            const unsigned grid = (unsigned)blocks;
            if (pick == 0) hipLaunchKernelGGL((dense_lds_kernel<T, Op, false, U>), dim3(grid), dim3(256), lds, s, a, b, out, lp);
            else hipLaunchKernelGGL((dense_lds_kernel<T, Op, true, U>), dim3(grid), dim3(256), lds, s, b, a, out, lp);
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
            t.tiles_q = (t.nq + tile_q<T>() - 1) / tile_q<T>();
            const size_t blocks = slices * t.tiles_p * t.tiles_q;
            if (blocks < 0x7fffffffull && pl.shape[paxis] < 0x7fffffffll && inner < 0x7fffffffll) {
                // the 16-byte form needs whole vectors along both plane axes; bases and pitches may be anything
                const bool vec = t.np % W == 0 && t.nq % W == 0;
                const dim3 grid((unsigned)blocks), block(256);
                if (!vec) hipLaunchKernelGGL((tile_kernel<T, Op, false, 0, 0>), grid, block, 0, s, a, b, out, t);
                else if (t.mode_a == 1 && t.mode_b == 1) hipLaunchKernelGGL((tile_kernel<T, Op, true, 1, 1>), grid, block, 0, s, a, b, out, t);
                else if (t.mode_a == 1) hipLaunchKernelGGL((tile_kernel<T, Op, true, 1, 0>), grid, block, 0, s, a, b, out, t);
                else hipLaunchKernelGGL((tile_kernel<T, Op, true, 0, 1>), grid, block, 0, s, a, b, out, t);
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
    // An operand strided along the inner axis (a[:, ::2], a channel of an interleaved image): consecutive LANES on
    // consecutive outputs keep each load instruction inside a few cache lines -- 119 us against 160 us with W outputs
    // per lane for A[:, ::2] + B[:, ::2] at 8192 x 4096.  Inner strides 0 / 1 (tiny inner extents) keep the vector store.
    if (ia > 1 || ib > 1) {
        const unsigned grid = (unsigned)((pl.n + 255) / 256);
        hipLaunchKernelGGL((gather_kernel<T, Op, 1>), dim3(grid), dim3(256), 0, s, a, b, out, g);
    } else {
        const unsigned grid = (unsigned)(((pl.n + W - 1) / W + 255) / 256);
        hipLaunchKernelGGL((gather_kernel<T, Op, W>), dim3(grid), dim3(256), 0, s, a, b, out, g);
    }
    SMHIP_LAUNCH_CHECK("gather_kernel");
    return SMHIP_OK;
}


// ------------------------------------------------------------------ strided copy
// dst[sum idx_k * sd_k] = src[sum idx_k * ss_k]: the scatter side of SMArray's element-copy assignment
// (`view = array`, reference SMArray.h:89-97), which the reference runs as a host loop.  Rows are the merged
// inner axis; when both inner strides are 1 a lane moves one (element-aligned) 16-byte vector, ragged
// last slot element-wise; otherwise one element per lane.
struct CopyParams {
    int64_t ss[kMaxOuter], sd[kMaxOuter];  // outer strides, innermost-outer first
    FastDiv shape[kMaxOuter];
    int n_outer;
    int64_t inner_ss, inner_sd;
    uint32_t inner;
    FastDiv slots_per_row;
    uint32_t slots;  // rows * slots_per_row
};

template <typename T, bool VEC>
__global__ __launch_bounds__(256) void strided_copy_kernel(const T *__restrict__ src, T *__restrict__ dst, CopyParams p) {
    constexpr int W = VEC ? VecTraits<T>::width : 1;
    typedef typename VecTraits<T>::vec_t V;
    const uint32_t slot = blockIdx.x * 256u + threadIdx.x;
    if (slot >= p.slots) return;
    uint32_t row, col;
    p.slots_per_row.divmod(slot, row, col);
    int64_t offS = 0, offD = 0;
    uint32_t rem = row;
    for (int k = 0; k < p.n_outer - 1; ++k) {
        uint32_t q, idx;
        p.shape[k].divmod(rem, q, idx);
        rem = q;
        offS += (int64_t)idx * p.ss[k];
        offD += (int64_t)idx * p.sd[k];
    }
    if (p.n_outer > 0) {
        offS += (int64_t)rem * p.ss[p.n_outer - 1];
        offD += (int64_t)rem * p.sd[p.n_outer - 1];
    }
    const uint32_t e0 = col * W;
    if constexpr (VEC) {
        if (e0 + W <= p.inner) {
            *reinterpret_cast<V *>(dst + offD + e0) = load_stream(reinterpret_cast<const V *>(src + offS + e0));
        } else {
            for (uint32_t e = e0; e < p.inner; ++e) dst[offD + e] = src[offS + e];
        }
    } else {
        dst[offD + (int64_t)e0 * p.inner_sd] = src[offS + (int64_t)e0 * p.inner_ss];
    }
}

template <typename T>
int run_copy_strided(const void *src_, void *dst_, const Plan &pl, hipStream_t s) {
    constexpr int W = VecTraits<T>::width;
    const T *src = static_cast<const T *>(src_);
    T *dst = static_cast<T *>(dst_);
    const int nd = pl.ndim;
    CopyParams p{};
    p.n_outer = nd - 1;
    p.inner = (uint32_t)pl.shape[nd - 1];
    p.inner_ss = pl.sa[nd - 1];
    p.inner_sd = pl.sb[nd - 1];
    for (int k = 0; k < nd - 1; ++k) {
        const int from = nd - 2 - k;
        p.shape[k] = FastDiv((uint32_t)pl.shape[from]);
        p.ss[k] = pl.sa[from];
        p.sd[k] = pl.sb[from];
    }
    const bool vec = p.inner_ss == 1 && p.inner_sd == 1;
    const size_t per_row = vec ? ((size_t)p.inner + W - 1) / W : p.inner;
    const size_t slots = pl.n / (size_t)p.inner * per_row;
    p.slots_per_row = FastDiv((uint32_t)per_row);
    p.slots = (uint32_t)slots;
    const unsigned grid = (unsigned)((slots + 255) / 256);
    if (vec) hipLaunchKernelGGL((strided_copy_kernel<T, true>), dim3(grid), dim3(256), 0, s, src, dst, p);
    else hipLaunchKernelGGL((strided_copy_kernel<T, false>), dim3(grid), dim3(256), 0, s, src, dst, p);
    SMHIP_LAUNCH_CHECK("strided_copy_kernel");
    return SMHIP_OK;
}

int copy_plan(int dtype, const void *src, void *dst, const Plan &pl, hipStream_t s) {
    const size_t esz = dtype_size(dtype);
    if (pl.ndim == 1 && pl.sa[0] == 1 && pl.sb[0] == 1) {
        SMHIP_TRY(hipMemcpyAsync(dst, src, pl.n * esz, hipMemcpyDeviceToDevice, s));
        return SMHIP_OK;
    }
    if (pl.n >= 0x7fffffffull) {  // same cut as launch_plan: pieces of < 2^31 elements along the outermost axis
        const size_t slice = pl.n / (size_t)pl.shape[0];
        const size_t per = slice >= 0x7fffffffull ? 1 : 0x7ffffffeull / slice;
        int64_t shape[SMHIP_MAX_NDIM];
        for (int d = 0; d < pl.ndim; ++d) shape[d] = pl.shape[d];
        for (size_t i0 = 0; i0 < (size_t)pl.shape[0]; i0 += per) {
            const size_t left = (size_t)pl.shape[0] - i0;
            shape[0] = (int64_t)(left < per ? left : per);
            const Plan sub = normalise(shape, pl.sa, pl.sb, pl.ndim);
            if (int rc = copy_plan(dtype, static_cast<const char *>(src) + (int64_t)i0 * pl.sa[0] * (int64_t)esz,
                                   static_cast<char *>(dst) + (int64_t)i0 * pl.sb[0] * (int64_t)esz, sub, s))
                return rc;
        }
        return SMHIP_OK;
    }
    switch (dtype) {
        case SMHIP_F32: case SMHIP_I32: return run_copy_strided<int32_t>(src, dst, pl, s);   // a copy only needs the width
        case SMHIP_F64: case SMHIP_I64: return run_copy_strided<int64_t>(src, dst, pl, s);
    }
    return fail(SMHIP_ERR_INVALID, "copy_strided: bad dtype %d", dtype);
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

int launch_copy_strided(int dtype, const void *src, const int64_t *src_strides, void *dst, const int64_t *dst_strides,
                        const int64_t *shape, int ndim, hipStream_t s) {
    const Plan pl = normalise(shape, src_strides, dst_strides, ndim);  // merges axes that are jointly dense in src AND dst
    if (pl.n == 0) return SMHIP_OK;
    return copy_plan(dtype, src, dst, pl, s);
}

int launch_broadcast(int op, int dtype, const void *a, const int64_t *sa, const void *b, const int64_t *sb,
                     const int64_t *shape, int ndim, void *out, hipStream_t s) {
    const Plan pl = normalise(shape, sa, sb, ndim);
    if (pl.n == 0) return SMHIP_OK;
    return launch_plan(op, dtype, a, b, out, pl, s);
}

}  // namespace smhip
