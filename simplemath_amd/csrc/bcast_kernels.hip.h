// bcast_kernels.hip.h -- the bodies of the broadcast kernels (row / gather / LDS / tile) and their parameter blocks.
//
// Device code only, no #includes: this text is compiled twice --
//   * ahead of time by hipcc as part of broadcast.hip, which wraps each body in a __global__ template per built-in Op, and
//   * at run time by hipRTC for user-defined Ops (jit.hip embeds this file as a string and wraps the one body variant a
//     launch needs in an extern "C" kernel), so `x.apply<MyOp>(y)` runs the same kernels as `x + y`.
// It expects, already declared where it is included: uint32_t / int64_t / size_t, SMHIP_MAX_NDIM, VecTraits<T>, FastDiv,
// OpCtx<Op>, apply_n<Op, T, W>(), load_stream / store_stream  (ops.hip.h ahead of time; a short prelude in jit.hip at run time).
// All kernels are launched with 256 threads per workgroup.

constexpr int kMaxOuter = SMHIP_MAX_NDIM - 1;
template <bool B> struct BoolTag { static constexpr bool value = B; };
template <int I> struct IntTag { static constexpr int value = I; };

struct RowParams {
    int64_t sa[kMaxOuter], sb[kMaxOuter];  // outer strides, elements, innermost-outer first
    FastDiv shape[kMaxOuter];              // outer extents, innermost-outer first
    int n_outer;
    uint32_t rows;    // product of outer extents
    uint32_t inner;   // inner extent in elements
    uint32_t vpr;     // vector slots per row = ceil(inner / W)
    uint32_t grid_x;  // workgroups along the row; the launch is 1-D (grid y is limited to 65 535)
    uint32_t nt;      // read the streamed operand(s) non-temporally: set when the launch reads more than the Infinity Cache holds
};

// INNER_x: 1 = dense along the inner axis, 0 = broadcast along it.
// CONST_x: operand has all outer strides zero -> identical for every row.
// Every lane owns one 16-byte slot of W elements of a row; accesses are element-aligned vectors (any base,
// any pitch).  The last slot of a row whose extent is not a multiple of W is handled element by element.
template <typename T, typename Op, int INNER_A, int INNER_B, bool CONST_A, bool CONST_B, int TX, int ROWS>
__device__ __forceinline__ void row_body(const T *__restrict__ a, const T *__restrict__ b, T *__restrict__ out,
                                                  RowParams p) {
    constexpr int W = VecTraits<T>::width;
    typedef typename VecTraits<T>::vec_t V;
    constexpr int TY = 256 / TX;
    // (Staging pow's tables behind the rows' loads, as flat_tile_kernel does with OpCtx's fetch / commit, was measured
    // here and made it slower -- 22.7 -> 24.0 us for the 4096 x 4096 row pow: with this body's control flow the commit
    // waits for every load of both rows, where the first row's arithmetic now starts as soon as its own data is in.)
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
    // `streamed` is a TYPE (BoolTag), not a value: written as `if (streamed) load_stream(p) else *p` the two loads are
    // merged into one plain load while the lambda is optimised on its own -- before inlining could fold the flag --
    // and the non-temporal hint is silently gone.  (That is what round 1's row kernels did, and why their 1R+1W shapes
    // ran at 90 %: at those sizes plain loads are the better policy.  It is now a decision: p.nt, see load_stream_if.)
    auto load = [&](const T *base, int64_t off, int inner_mode, auto streamed, T (&dst)[W]) {
        if (inner_mode == 0) {
            const T s = base[off];
#pragma unroll
            for (int k = 0; k < W; ++k) dst[k] = s;
        } else if (whole) {
            const V *src = reinterpret_cast<const V *>(base + off + col_elem);
            V v;
            if constexpr (decltype(streamed)::value) v = load_stream_if(T, src, p.nt);
            else v = *src;
#pragma unroll
            for (int k = 0; k < W; ++k) dst[k] = v[k];
        } else {
#pragma unroll
            for (int k = 0; k < W; ++k) dst[k] = k < count ? base[off + col_elem + k] : base[off + col_elem];
        }
    };

    T ca[W], cb[W];
    if constexpr (CONST_A) load(a, 0, INNER_A, BoolTag<false>{}, ca);
    if constexpr (CONST_B) load(b, 0, INNER_B, BoolTag<false>{}, cb);

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
        if constexpr (!CONST_A) load(a, offA, INNER_A, BoolTag<INNER_B == 1>{}, va[r]);
        if constexpr (!CONST_B) load(b, offB, INNER_B, BoolTag<INNER_A == 1>{}, vb[r]);
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
            store_stream_if(T, reinterpret_cast<V *>(dst), v, p.nt);
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
__device__ __forceinline__ void gather_body(const T *__restrict__ a, const T *__restrict__ b, T *__restrict__ out,
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

// ------------------------------------------------------------------- inner-strided rows
// An operand that takes every 2nd / 3rd / 4th element of the inner axis (a[:, ::2]; one channel of interleaved data).  The
// gather above reads such an operand with one 4-byte load per lane (W load instructions per 16 bytes of output); here a
// lane owns W consecutive outputs of a row and loads the S consecutive 16-byte vectors that hold its W inputs, keeping
// every S-th element -- S full-width loads per operand, every fetched line requested once, the store one vector.  (Half,
// two thirds or three quarters of every line it fetches is not used whatever the kernel does: the roofline of such a
// view is lines fetched, not bytes used.)  SA / SB: the operand's inner stride -- 0 one value per row, 1 dense, 2-4
// strided.  The outer axes are unravelled once per lane like the row kernel's.
struct StridedParams {
    int64_t sa[kMaxOuter], sb[kMaxOuter];  // outer strides, elements, innermost-outer first
    FastDiv shape[kMaxOuter];
    int n_outer;
    uint32_t inner;   // outputs per row
    FastDiv vpr;      // vector slots per row = ceil(inner / W)
    uint32_t slots;   // rows * vpr
    uint32_t nt;      // non-temporal reads: the launch fetches more lines than the Infinity Cache holds
};

// Stride 2 (with the other side stepping by 2 as well, dense, or one value per row) has a lane mapping of its own.  In the
// general form above lane l loads vectors 2l and 2l+1 of the row: each load INSTRUCTION then touches every line of the
// wave's span and uses half of it.  Here a wave owns 64 W consecutive outputs = 128 input vectors, lane l loads vectors l
// and l + 64 -- two fully contiguous instructions -- keeps the W/2 even elements of each, and stores two half-vectors:
// outputs [l W/2, +W/2) and [32 W + l W/2, +W/2) of the chunk, again contiguous across the lanes.  No shuffles, every line
// requested by one instruction.  (StridedParams: vpr = chunks per row, slots = rows x chunks; a chunk is
// kStrided2Groups x 32 W outputs.)
#ifndef SMHIP_STRIDED2_U
#define SMHIP_STRIDED2_U 1
#endif
constexpr int kStrided2Groups = 2 * SMHIP_STRIDED2_U;  // groups of W/2 outputs per lane; a wave owns 32 W of them per group
template <typename T, typename Op, int SA, int SB>
__device__ __forceinline__ void strided2_row_body(const T *__restrict__ a, const T *__restrict__ b, T *__restrict__ out, StridedParams p) {
    constexpr int W = VecTraits<T>::width, H = W / 2, G = kStrided2Groups;
    typedef typename VecTraits<T>::vec_t V;
    typedef typename VecTraits<T>::half_t HV;
    typedef typename VecTraits<T>::half_full_t HF;
    OpCtx<Op> ctx;
    ctx.init();
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t chunk = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (chunk >= p.slots) return;
    uint32_t row, c;
    p.vpr.divmod(chunk, row, c);
    int64_t offA = 0, offB = 0;
    {
        uint32_t rem = row;
        for (int k = 0; k < p.n_outer - 1; ++k) {
            uint32_t q, idx;
            p.shape[k].divmod(rem, q, idx);
            rem = q;
            offA += (int64_t)idx * p.sa[k];
            offB += (int64_t)idx * p.sb[k];
        }
        if (p.n_outer > 0) {
            offA += (int64_t)rem * p.sa[p.n_outer - 1];
            offB += (int64_t)rem * p.sb[p.n_outer - 1];
        }
    }
    T *orow = out + (size_t)row * p.inner;
    const uint32_t o0 = c * (32u * W * G) + lane * H;  // the lane's first group of H outputs; the others follow 32 W apart
    // strictly inside the row: a strided vector reads one element past its last kept one, which then still belongs to the
    // row (the row's last outputs are done element by element)
    if (o0 + 32u * W * (G - 1) + H < p.inner) {
        T xa[G][H], xb[G][H];
        auto fetch = [&](const T *base, auto stride_tag, auto nt_tag, uint32_t o, T (&dst)[H]) {
            constexpr int S = decltype(stride_tag)::value;
            if constexpr (S == 0) {
                const T v = *base;
#pragma unroll
                for (int k = 0; k < H; ++k) dst[k] = v;
            } else if constexpr (S == 1) {  // the dense side: a half-vector, non-temporal whatever the read policy
                const HF v = __builtin_nontemporal_load(reinterpret_cast<const HV *>(base + o));
#pragma unroll
                for (int k = 0; k < H; ++k) dst[k] = v[k];
            } else {
                const V v = load_stream_as(T, reinterpret_cast<const V *>(base + (int64_t)o * 2), decltype(nt_tag)::value);
#pragma unroll
                for (int k = 0; k < H; ++k) dst[k] = v[2 * k];
            }
        };
        if (p.nt & kLoadNt) {  // ONE branch around all the loads of the lane (see load_stream_as)
#pragma unroll
            for (int g = 0; g < G; ++g) fetch(a + offA, IntTag<SA>{}, BoolTag<true>{}, o0 + 32u * W * g, xa[g]);
#pragma unroll
            for (int g = 0; g < G; ++g) fetch(b + offB, IntTag<SB>{}, BoolTag<true>{}, o0 + 32u * W * g, xb[g]);
        } else {
#pragma unroll
            for (int g = 0; g < G; ++g) fetch(a + offA, IntTag<SA>{}, BoolTag<false>{}, o0 + 32u * W * g, xa[g]);
#pragma unroll
            for (int g = 0; g < G; ++g) fetch(b + offB, IntTag<SB>{}, BoolTag<false>{}, o0 + 32u * W * g, xb[g]);
        }
#pragma unroll
        for (int g = 0; g < G; ++g) {
            T res[H];
            apply_n<Op, T, H>(ctx, xa[g], xb[g], res);
            HF r;
#pragma unroll
            for (int k = 0; k < H; ++k) r[k] = res[k];
            __builtin_nontemporal_store(r, reinterpret_cast<HV *>(orow + o0 + 32u * W * g));
        }
    } else {
        for (int g = 0; g < G; ++g)
            for (uint32_t k = 0; k < (uint32_t)H; ++k) {
                const uint32_t o = o0 + 32u * W * g + k;
                if (o < p.inner) orow[o] = Op::apply(a[offA + (int64_t)o * SA], b[offB + (int64_t)o * SB]);
            }
    }
}

// Stride 4 on four-byte elements (one channel of an RGBA image, a[:, ::4]) the same way: in the general form below lane l
// loads the four vectors 4l .. 4l+3 -- four instructions that each touch every line of the wave's 4 KiB span and keep a
// quarter of it (30 % of algorithmic bytes = 61 % of the lines that must move).  Here a wave owns 256 consecutive outputs
// (the stride-2 chunk), lane l loads input vectors l, l + 64, l + 128, l + 192 -- four contiguous instructions --, keeps
// element 0 of each and stores four single elements, again contiguous across the lanes; the other side, if dense, is read
// the same way.
constexpr int kStrided4Outputs = 4;  // per lane: 64 x 4 = 256 outputs per wave = 32 W kStrided2Groups for four-byte types
template <typename T, typename Op, int SA, int SB>
__device__ __forceinline__ void strided4_row_body(const T *__restrict__ a, const T *__restrict__ b, T *__restrict__ out, StridedParams p) {
    static_assert(sizeof(T) == 4 && 32 * VecTraits<T>::width * kStrided2Groups == 64 * kStrided4Outputs, "the plan's chunk is the stride-2 one");
    constexpr int G = kStrided4Outputs;
    typedef typename VecTraits<T>::vec_t V;
    OpCtx<Op> ctx;
    ctx.init();
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t chunk = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (chunk >= p.slots) return;
    uint32_t row, c;
    p.vpr.divmod(chunk, row, c);
    int64_t offA = 0, offB = 0;
    {
        uint32_t rem = row;
        for (int k = 0; k < p.n_outer - 1; ++k) {
            uint32_t q, idx;
            p.shape[k].divmod(rem, q, idx);
            rem = q;
            offA += (int64_t)idx * p.sa[k];
            offB += (int64_t)idx * p.sb[k];
        }
        if (p.n_outer > 0) {
            offA += (int64_t)rem * p.sa[p.n_outer - 1];
            offB += (int64_t)rem * p.sb[p.n_outer - 1];
        }
    }
    T *orow = out + (size_t)row * p.inner;
    const uint32_t o0 = c * (64u * G) + lane;  // the lane's first output; the others follow 64 apart
    // strictly inside the row: a strided vector reads three elements past its kept one, which then still belong to the row
    if (c * (64u * G) + 64u * G < p.inner) {
        T xa[G], xb[G];
        auto fetch = [&](const T *base, auto stride_tag, auto nt_tag, uint32_t o, T &dst) {
            constexpr int S = decltype(stride_tag)::value;
            if constexpr (S == 0) dst = *base;
            else if constexpr (S == 1) dst = __builtin_nontemporal_load(base + o);  // the dense side: non-temporal whatever the read policy
            else {
                const V v = load_stream_as(T, reinterpret_cast<const V *>(base + (int64_t)o * 4), decltype(nt_tag)::value);
                dst = v[0];
            }
        };
        if (p.nt & kLoadNt) {  // ONE branch around all the loads of the lane (see load_stream_as)
#pragma unroll
            for (int g = 0; g < G; ++g) fetch(a + offA, IntTag<SA>{}, BoolTag<true>{}, o0 + 64u * g, xa[g]);
#pragma unroll
            for (int g = 0; g < G; ++g) fetch(b + offB, IntTag<SB>{}, BoolTag<true>{}, o0 + 64u * g, xb[g]);
        } else {
#pragma unroll
            for (int g = 0; g < G; ++g) fetch(a + offA, IntTag<SA>{}, BoolTag<false>{}, o0 + 64u * g, xa[g]);
#pragma unroll
            for (int g = 0; g < G; ++g) fetch(b + offB, IntTag<SB>{}, BoolTag<false>{}, o0 + 64u * g, xb[g]);
        }
        T res[G];
        apply_n<Op, T, G>(ctx, xa, xb, res);
#pragma unroll
        for (int g = 0; g < G; ++g) __builtin_nontemporal_store(res[g], orow + o0 + 64u * g);
    } else {
        for (int g = 0; g < G; ++g) {
            const uint32_t o = o0 + 64u * g;
            if (o < p.inner) orow[o] = Op::apply(a[offA + (int64_t)o * SA], b[offB + (int64_t)o * SB]);
        }
    }
}

template <typename T, typename Op, int SA, int SB>
__device__ __forceinline__ void strided_row_body(const T *__restrict__ a, const T *__restrict__ b, T *__restrict__ out, StridedParams p) {
    if constexpr ((SA == 2 || SB == 2) && SA <= 2 && SB <= 2) {
        strided2_row_body<T, Op, SA, SB>(a, b, out, p);
        return;
    }
    if constexpr (sizeof(T) == 4 && (SA == 4 || SB == 4) && SA != 2 && SA != 3 && SB != 2 && SB != 3) {
        strided4_row_body<T, Op, SA, SB>(a, b, out, p);
        return;
    }
    constexpr int W = VecTraits<T>::width;
    typedef typename VecTraits<T>::vec_t V;
    OpCtx<Op> ctx;
    ctx.init();
    const uint32_t slot = blockIdx.x * 256u + threadIdx.x;
    if (slot >= p.slots) return;
    uint32_t row, col;
    p.vpr.divmod(slot, row, col);
    int64_t offA = 0, offB = 0;
    {
        uint32_t rem = row;
        for (int k = 0; k < p.n_outer - 1; ++k) {
            uint32_t q, idx;
            p.shape[k].divmod(rem, q, idx);
            rem = q;
            offA += (int64_t)idx * p.sa[k];
            offB += (int64_t)idx * p.sb[k];
        }
        if (p.n_outer > 0) {
            offA += (int64_t)rem * p.sa[p.n_outer - 1];
            offB += (int64_t)rem * p.sb[p.n_outer - 1];
        }
    }
    const uint32_t e0 = col * W;
    T *dst = out + (size_t)row * p.inner + e0;
    // strictly inside the row: the S vectors read up to S - 1 elements past the lane's last input, which then still
    // belongs to the row (the row's own last slot is done element by element)
    if (e0 + W < p.inner) {
        T xa[W], xb[W], res[W];
        auto fetch = [&](const T *base, auto stride_tag, auto nt_tag, T (&dst_regs)[W]) {
            constexpr int S = decltype(stride_tag)::value;
            if constexpr (S == 0) {
                const T v = *base;
#pragma unroll
                for (int k = 0; k < W; ++k) dst_regs[k] = v;
            } else {
                V v[S];
#pragma unroll
                for (int i = 0; i < S; ++i) v[i] = load_stream_as(T, reinterpret_cast<const V *>(base + (int64_t)e0 * S) + i, decltype(nt_tag)::value);
#pragma unroll
                for (int k = 0; k < W; ++k) dst_regs[k] = v[(k * S) / W][(k * S) % W];
            }
        };
        if (p.nt & kLoadNt) {  // ONE branch around all the loads of the lane (see load_stream_as)
            fetch(a + offA, IntTag<SA>{}, BoolTag<true>{}, xa);
            fetch(b + offB, IntTag<SB>{}, BoolTag<true>{}, xb);
        } else {
            fetch(a + offA, IntTag<SA>{}, BoolTag<false>{}, xa);
            fetch(b + offB, IntTag<SB>{}, BoolTag<false>{}, xb);
        }
        apply_n<Op, T, W>(ctx, xa, xb, res);
        V r;
#pragma unroll
        for (int k = 0; k < W; ++k) r[k] = res[k];
        store_stream(reinterpret_cast<V *>(dst), r);
    } else {
        for (uint32_t k = 0; e0 + k < p.inner; ++k)
            dst[k] = Op::apply(a[offA + (int64_t)(e0 + k) * SA], b[offB + (int64_t)(e0 + k) * SB]);
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
    uint32_t nt;                      // read the dense operand non-temporally (it exceeds the Infinity Cache)
};

// x: the operand that is dense in output order (streams as vectors); y: the small one, gathered
// from its LDS copy.  SWAPPED: x is the Op's right operand.  Each lane keeps U vectors of x in flight
// (all loads issued before any arithmetic).  The N-D index of a vector's first element comes from one
// fast-division chain (none for the outermost axis); its other W-1 elements follow by increment-and-carry,
// which is full-rate integer work instead of W more chains of quarter-rate mul-hi / mul-lo.
template <typename T, typename Op, bool SWAPPED, int U>
__device__ __forceinline__ void dense_lds_body(const T *__restrict__ x, const T *__restrict__ y, T *__restrict__ out,
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
            if (v < p.n_vec) xv[u] = load_stream_if(T, reinterpret_cast<const V *>(x) + v, p.nt);
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
                store_stream_if(T, reinterpret_cast<V *>(out) + v, rv, p.nt);
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
#ifndef SMHIP_TILE_BOTH_CHUNK
#define SMHIP_TILE_BOTH_CHUNK 8
#endif
constexpr int kTileP = 64;
// Bytes of one patch row along q: 512 (`QB` default); 1024 for the WIDE patch that launches over arrays beyond the
// Infinity Cache take together with a row-major walk (TileParams::order 0); 128 for the SHORT patch of planes whose q extent
// is a few dozen elements (out (4194304, 32) = a.T + b: a 512-byte patch row is three quarters empty there) -- see
// plan_launch() and DESIGN.md section 3.
constexpr int kTileQBytes = 512, kTileQBytesWide = 1024, kTileQBytesShort = 128;
template <typename T, int QB = kTileQBytes> constexpr int tile_q() { return QB / (int)sizeof(T); }

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
    uint32_t nt;                // streamed reads carry the non-temporal hint (the launch reads more than the Infinity Cache holds)
    uint32_t order;             // walk of the patches: 1 diagonal, 0 row-major (q fastest, p unshifted), 2 row-major in eight runs, one per XCD
    uint32_t total;             // order 2: patches in all (the grid is rounded up to a multiple of eight)
    uint32_t in_place;          // the output overlaps an operand: no patch may compute an element twice (see the pull-back below)
};

// One workgroup = one 64 x TQ patch (i along p, j along q) of one slice of the remaining axes.
// VEC: every global access is a 16-byte vector (W elements) -- along p for operands turned through
// LDS, along q for direct operands and the output.  LDS tiles are stored already transposed ([i][j]) in
// a bank-conflict-free layout (`at` below; measured: profiles/r01_pmc_lds_tile_kernel.txt).  There is ONE tile: with a single LDS-mode operand it holds that operand; when both
// operands are contiguous along p (a.T op b.T) phase 1 loads both coalesced, applies the Op there and
// stages the RESULT, so phase 2 is a pure transposed write-out.  MA / MB are compile-time in the
// vector form; the element form (odd extents, pitches, bases) keeps them as runtime values.
template <typename T, typename Op, bool VEC, int MA, int MB, int QB = kTileQBytes>
__device__ __forceinline__ void tile_body(const T *__restrict__ a, const T *__restrict__ b, T *__restrict__ out,
                                                   TileParams p) {
    constexpr int W = VEC ? VecTraits<T>::width : 1;
    constexpr int TQ = tile_q<T, QB>();
    constexpr int VP = kTileP / W, VQ = TQ / W;  // vector slots per patch row, along p / along q
    // LDS layout of element (i, j): 4-byte types get a skewed layout (one pad word per 32 columns, two per 32 rows,
    // odd pitch) that makes both the 4-byte scatter of phase 1 and the stride-4 reads of phase 2 hit 32 distinct
    // banks per 32-lane group; 8-byte types keep the plain padded pitch.
    constexpr bool SKEW = sizeof(T) == 4;
    constexpr int PITCH = SKEW ? TQ + TQ / 32 + 1 : TQ + 1;  // skewed: column j sits at j + j/32, rows 32.. two words further, and the pitch is odd
    auto at = [](uint32_t i, uint32_t j) -> uint32_t { return SKEW ? i * PITCH + j + (j >> 5) + ((i >> 5) << 1) : i * PITCH + j; };
    typedef typename VecTraits<T>::vec_t V;
    __shared__ T tile[kTileP * PITCH];
    const int mode_a = VEC ? MA : p.mode_a, mode_b = VEC ? MB : p.mode_b;
    const bool both = mode_a == 1 && mode_b == 1;
    uint32_t bid = blockIdx.x;
    if (p.order == 2) {
        // Rows that do not start on 128-byte lines: the patches on either side of a seam share the lines there -- of the direct
        // operand AND of the output, whose halves then leave as two partial writes.  Workgroups are dealt to the eight XCDs in
        // turn, so neighbouring patches of the row-major walk land on DIFFERENT L2s, each of which fetches (and writes back) its
        // own copy of the shared line: at 12287 x 12287 the launch fetches 1.18 x the bytes of 12288 x 12288 and 5.9 % of its write
        // requests are partial (profiles/r04_pmc_tile_odd.txt).  Here XCD x walks the x-th eighth of the patch sequence, so
        // that neighbours along q -- and along p, 48 patches further on -- meet in ONE L2, back to back.
        const uint32_t per = gridDim.x >> 3;
        bid = (bid & 7u) * per + (bid >> 3);
        if (bid >= p.total) return;
    }
    uint32_t tq3 = 0, tp3 = 0;
    if (p.order == 3) {
        // Blocks of BP x BQ patches, dealt to the XCDs in turn: the patches on either side of a seam -- along q (the output's and the
        // direct operand's shared lines) and along p (the turned operand's) -- run back to back on ONE XCD, and the eight XCDs work
        // on neighbouring blocks.  p.total packs BP | BQ << 8; the grid is 8 * BP * BQ * ceil(blocks / 8).
        const uint32_t BP = p.total & 0xffu, BQ = (p.total >> 8) & 0xffu, per = BP * BQ;
        const uint32_t x = bid & 7u, local = bid >> 3;
        const uint32_t blk = (local / per) * 8u + x, in = local % per;
        const uint32_t blocks_q = (p.tiles_q + BQ - 1) / BQ;
        tq3 = (blk % blocks_q) * BQ + in % BQ;
        tp3 = (blk / blocks_q) * BP + in / BQ;  // over tiles_p x slices
        uint32_t rows = p.tiles_p;
        for (int k = 0; k < p.n_rest; ++k) rows *= p.rest[k].d;
        if (tq3 >= p.tiles_q || tp3 >= rows) return;
        bid = tp3 * p.tiles_q + tq3;  // the row-major index the code below takes apart
    }
    OpCtx<Op> ctx;
    ctx.init();
    // Consecutive workgroups walk q (walking p instead was 15-25 % slower with one turned operand, r01, and 35 % slower
    // with two, r02: 139 -> 188 us) -- along a DIAGONAL: workgroup (tp, tq) takes patch ((tp + tq) mod tiles_p, tq), so
    // the workgroups in flight together read different column offsets of the turned operand(s) as well as different rows,
    // instead of 64 of them reading the same 256-byte column of 8192 rows at a power-of-two pitch.  (8192, 8192) f32,
    // tools/tile_modes.py -> profiles/r02_tile_order.txt: a.T + b.T 138.0 -> 130.5 us, and 169.9 -> 129.2 us when the
    // operands' pitch is 8256 elements; a.T + b at that pitch 119.6 -> 113.6 us.  Skewing q instead, walking p along the
    // diagonal, and a skew of three were all slower.  That holds up to 256 MiB per array.  Beyond, the diagonal is what
    // hurts: the workgroups in flight then write 512-byte pieces of a thousand different output rows, and the same kernel
    // walking row-major (order 0) with 1024-byte patch rows is 6-10 points faster (a.T + b at 16384^2: 68 -> 78 %).
    const uint32_t tq = bid % p.tiles_q; bid /= p.tiles_q;
    uint32_t tp = bid % p.tiles_p; bid /= p.tiles_p;
    if (p.order == 1) tp = (tp + tq) % p.tiles_p;
    int64_t offA = 0, offB = 0, offO = 0;
    for (int k = 0; k < p.n_rest; ++k) {
        uint32_t qd, idx;
        p.rest[k].divmod(bid, qd, idx);
        bid = qd;
        offA += (int64_t)idx * p.a_r[k];
        offB += (int64_t)idx * p.b_r[k];
        offO += (int64_t)idx * p.o_r[k];
    }
    uint32_t i0 = tp * kTileP, j0 = tq * TQ;
    if (VEC) {
        // A patch that would hang over the plane's edge is pulled back inside: it overlaps its neighbour, whose elements it
        // computes and stores a second time -- the same values -- and every patch of a large plane is a whole one.  (With the hanging patches on the guarded path below, 8191 x 8191 ran at 67 % where
        // 8192 x 8192 runs at 87 %; before the vector form took ragged extents at all, at 28 %.)
        // (only where the second helping is small change: four patches and more along the axis -- pulled back inside
        // 100 x 100 planes, short patches did 1.6 times the work)
        // Storing "the same values" twice needs the operands to be what they were: an output that overlaps an operand
        // (smhip_elementwise(ADD, a, b.T, out = a)) would feed the second computation its own results, so in-place
        // problems keep their hanging patches on the guarded path.
        if (i0 + kTileP > p.np && p.np >= 4u * kTileP && !p.in_place) i0 = p.np - kTileP;
        if (j0 + TQ > p.nq && p.nq >= 4u * TQ && !p.in_place) j0 = p.nq - TQ;
    }
    const bool full = i0 + kTileP <= p.np && j0 + TQ <= p.nq;  // workgroup-uniform
    // patch origins
    const T *a0 = a + offA + (int64_t)i0 * p.a_p + (int64_t)j0 * p.a_q;
    const T *b0 = b + offB + (int64_t)i0 * p.b_p + (int64_t)j0 * p.b_q;
    T *o0 = out + offO + (int64_t)i0 * p.o_p + j0;

    constexpr int S1 = TQ * VP / 256, S2 = kTileP * VQ / 256;
    // ---- whole patches in the vector form: every load of the lane is issued before anything is used -----------------
    // Written slot by slot (load, use, next slot) each load sat in a basic block of its own with an s_waitcnt vmcnt(0)
    // behind it: up to sixteen round trips to memory in a row per lane.  Here the S1 turned loads (x2 when both operands
    // are turned) and the S2 loads of the direct operand all go out first; then LDS writes, barrier, LDS reads, Op, stores.
    if constexpr (VEC) {
        constexpr bool kBoth = MA == 1 && MB == 1;
        const int64_t direct_q = MA == 1 ? p.b_q : p.a_q;
        // (a direct operand that does not move along q -- one value per row, or one value: `dst = src.T` is such a problem --
        // is splat from one element per slot; through the guarded path below A.T * column ran at 48-50 % at 12288^2 and beyond)
        const bool dsplat = !kBoth && direct_q == 0;
        if (full && (kBoth || direct_q == 1 || direct_q == 0)) {
            const T *d0 = MA == 1 ? b0 : a0;  // the direct operand (unused when both are turned)
            const int64_t d_p = MA == 1 ? p.b_p : p.a_p;
            // Slots whose loads go out together.  One turned operand: all of them (and all of the direct operand's).  Two
            // turned operands: SMHIP_TILE_BOTH_CHUNK slots at a time, by default all eight.  (Under the row-major walk of
            // the patches more loads in flight made the two scattered streams slower -- a.T + b.T at 8192^2: 139 us with
            // one slot at a time, 147 / 153 / 151 with two / four / eight -- which was the column camping the diagonal
            // walk removes; with it one, two and eight slots give 131.6 / 130.1 / 130.5 us, and at a pitch of 8256
            // elements 144.6 / 140.8 / 129.2.)
            constexpr int CH = kBoth ? (SMHIP_TILE_BOTH_CHUNK < S1 ? SMHIP_TILE_BOTH_CHUNK : S1) : S1;
            V va[CH], vb[kBoth ? CH : 1], vd[kBoth ? 1 : S2];
#pragma unroll
            for (int c = 0; c < S1; c += CH) {
                auto issue = [&](auto nt_tag) {
                    constexpr bool NT = decltype(nt_tag)::value;
#pragma unroll
                    for (int s = 0; s < CH; ++s) {
                        const uint32_t v = threadIdx.x + 256 * (c + s), jl = v / VP, ig = v % VP;
                        if constexpr (kBoth) {  // two turned streams and no reuse: the footprint decides the policy
                            va[s] = load_stream_as(T, reinterpret_cast<const V *>(a0 + ig * W + (int64_t)jl * p.a_q), NT);
                            vb[s] = load_stream_as(T, reinterpret_cast<const V *>(b0 + ig * W + (int64_t)jl * p.b_q), NT);
                        } else if constexpr (MA == 1) {  // one turned stream: cached reads win up to 256 MiB per array (tools/sweep_transpose.hip); the wide patch follows the policy
                            va[s] = load_stream_as(T, reinterpret_cast<const V *>(a0 + ig * W + (int64_t)jl * p.a_q), NT && QB == kTileQBytesWide);
                        } else {
                            va[s] = load_stream_as(T, reinterpret_cast<const V *>(b0 + ig * W + (int64_t)jl * p.b_q), NT && QB == kTileQBytesWide);
                        }
                    }
                    if constexpr (!kBoth) {
#pragma unroll
                        for (int s = 0; s < S2; ++s) {
                            const uint32_t v = threadIdx.x + 256 * s, il = v / VQ, jg = v % VQ;
                            if (dsplat) {
                                const T one = d0[(int64_t)il * d_p];
#pragma unroll
                                for (int k = 0; k < W; ++k) vd[s][k] = one;
                            } else {
                                vd[s] = load_stream_as(T, reinterpret_cast<const V *>(d0 + (int64_t)il * d_p + jg * W), NT);
                            }
                        }
                    }
                };
                if (p.nt & kLoadNt) issue(BoolTag<true>{});  // ONE branch around the loads of a chunk (see load_stream_as)
                else issue(BoolTag<false>{});
#pragma unroll
                for (int s = 0; s < CH; ++s) {
                    const uint32_t v = threadIdx.x + 256 * (c + s), jl = v / VP, ig = v % VP;
                    T x[W];
                    if constexpr (kBoth) {
                        T xa[W], xb[W];
#pragma unroll
                        for (int k = 0; k < W; ++k) { xa[k] = va[s][k]; xb[k] = vb[s][k]; }
                        apply_n<Op, T, W>(ctx, xa, xb, x);
                    } else {
#pragma unroll
                        for (int k = 0; k < W; ++k) x[k] = va[s][k];
                    }
#pragma unroll
                    for (int k = 0; k < W; ++k) tile[at(ig * W + k, jl)] = x[k];
                }
            }
            __syncthreads();
#pragma unroll
            for (int s = 0; s < S2; ++s) {
                const uint32_t v = threadIdx.x + 256 * s, il = v / VQ, jg = v % VQ;
                T xt[W], xr[W];
#pragma unroll
                for (int k = 0; k < W; ++k) xt[k] = tile[at(il, jg * W + k)];
                if constexpr (kBoth) {
#pragma unroll
                    for (int k = 0; k < W; ++k) xr[k] = xt[k];
                } else {
                    T xd[W];
#pragma unroll
                    for (int k = 0; k < W; ++k) xd[k] = vd[s][k];
                    if constexpr (MA == 1) apply_n<Op, T, W>(ctx, xt, xd, xr);
                    else apply_n<Op, T, W>(ctx, xd, xt, xr);
                }
                V val;
#pragma unroll
                for (int k = 0; k < W; ++k) val[k] = xr[k];
                store_stream_if(T, reinterpret_cast<V *>(o0 + (int64_t)il * p.o_p + jg * W), val, p.nt);
            }
            return;
        }
    }

    // ---- edge patches, a direct operand that is not dense along q, and the element form: slot by slot, guarded -------
    // phase 1: LDS-mode operands, coalesced along p (slot ig covers i = ig*W .. +W-1 of row jl)
    // (a slot that hangs over the plane's edge -- extents need not be multiples of the vector width -- moves element by
    // element, the elements outside left at zero: only whole patches, i.e. the interior, need whole vectors)
    auto along_p = [&](const T *src0, int64_t s_q, uint32_t jl, uint32_t ig, T (&dst)[W]) {
        const T *g = src0 + ig * W + (int64_t)jl * s_q;
        if constexpr (VEC) {
            if (full || i0 + ig * W + W <= p.np) {
                const V val = *reinterpret_cast<const V *>(g);
#pragma unroll
                for (int k = 0; k < W; ++k) dst[k] = val[k];
            } else {
#pragma unroll
                for (int k = 0; k < W; ++k) dst[k] = i0 + ig * W + k < p.np ? g[k] : T{};
            }
        } else {
            dst[0] = *g;
        }
    };
#pragma unroll
    for (int s = 0; s < S1; ++s) {
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
        const bool whole = full || j0 + jg * W + W <= p.nq;
        if (VEC && s_q == 1 && whole) {
            const V val = load_stream_if(T, reinterpret_cast<const V *>(g), p.nt);
#pragma unroll
            for (int k = 0; k < W; ++k) dst[k] = val[k];
        } else {
#pragma unroll
            for (int k = 0; k < W; ++k) dst[k] = (whole || j0 + jg * W + k < p.nq) ? g[(int64_t)k * s_q] : T{};
        }
    };
#pragma unroll
    for (int s = 0; s < S2; ++s) {
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
                if (full || j0 + jg * W + W <= p.nq) {
                    V val;
#pragma unroll
                    for (int k = 0; k < W; ++k) val[k] = xr[k];
                    store_stream(reinterpret_cast<V *>(dst), val);
                } else {
#pragma unroll
                    for (int k = 0; k < W; ++k)
                        if (j0 + jg * W + k < p.nq) dst[k] = xr[k];
                }
            } else {
                *dst = xr[0];
            }
        }
    }
}

#ifndef SMHIP_TILE_SHIFT_FORM
#define SMHIP_TILE_SHIFT_FORM 0
#endif
#if SMHIP_TILE_SHIFT_FORM
// ------------------------------------------------------------------ tile kernel, output rows off the 128-byte lines
// An inner extent like 12287 puts every output row at its own phase against the 128-byte lines.  tile_body's patches cut every
// row at the same COLUMNS, so each patch row starts and ends inside a line it shares with the neighbouring patch: the output
// leaves as two partial writes per seam and row, the direct operand's seam lines are fetched twice.  Where that costs
// (tools/tile_align_probe.py -> profiles/r04_tile_align_probe.txt, out = A.T + B, f32, 12288^2 with one base pointer moved off
// the lines: everything aligned 298 us; out off 337; A, the turned side, off 330; B off 309; all three 377 = the 370 us of
// 12287^2), the output is the largest term -- and the one a different cut can remove: here row i of patch tq covers the columns
// [tq TQ - m_i, (tq + 1) TQ - m_i) with m_i = the row's phase, (address of out[i][0]) mod 128 bytes in elements -- a parallelogram
// in index space whose every row is a whole number of lines of the output (and of the direct operand when it shares the
// output's pitch and phase).  The turned operand is staged for TQ + G columns (G = one line of elements: the union of the rows'
// windows), phase 2 reads row i of the LDS tile at its own shift.  One more patch column than tile_body needs at most; the
// windows that hang over a row's ends (first and last patch column) move those vectors element by element.  Loads are issued
// unconditionally (addresses clamped into the row), so every load of a lane still goes out ahead of the first use.
// MEASURED AND NOT ADOPTED (profiles/r04_tile_align_probe.txt): at 12287^2 the aligned output wins what the 12.5 % of extra turned
// loads and the extra patch column lose (370 -> 378 us; on rows that ARE on the lines the form costs 7.7 %: 300 -> 323 us) -- the
// kernel is bound by the number of 128-byte line requests, not by DRAM bytes, and this form trades one kind for another.  Built
// with -DSMHIP_TILE_SHIFT_FORM=1 and selected with SMHIP_TILE_SHIFT=1 (2: also for rows on the lines).
// The host guarantees: np >= 4 * kTileP and the output overlaps no operand (row-edge patches are pulled back inside and
// recompute their neighbour's values), the direct operand is dense or constant along q.
constexpr int kTileShiftBytes = 128;
template <typename T, typename Op, int MA, int MB, int QB>
__device__ __forceinline__ void tile_shift_body(const T *__restrict__ a, const T *__restrict__ b, T *__restrict__ out, TileParams p) {
    constexpr int W = VecTraits<T>::width;
    constexpr int TQ = tile_q<T, QB>();
    constexpr int G = kTileShiftBytes / (int)sizeof(T);
    constexpr int TC = TQ + G;  // columns staged
    constexpr int VP = kTileP / W, VQ = TQ / W;
    constexpr bool SKEW = sizeof(T) == 4;
    constexpr int PITCH = (SKEW ? TC + TC / 32 + 1 : TC + 1) | 1;  // tile_body's layout, the pitch kept odd
    auto at = [](uint32_t i, uint32_t j) -> uint32_t { return SKEW ? i * PITCH + j + (j >> 5) + ((i >> 5) << 1) : i * PITCH + j; };
    typedef typename VecTraits<T>::vec_t V;
    constexpr bool kBoth = MA == 1 && MB == 1;
    constexpr int S1 = TC * VP / 256, S2 = kTileP * VQ / 256;
    static_assert((TC * VP) % 256 == 0 && (kTileP * VQ) % 256 == 0, "whole slots");
    __shared__ T tile[kTileP * PITCH];
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
    uint32_t i0 = tp * kTileP;
    if (i0 + kTileP > p.np) i0 = p.np - kTileP;
    const int32_t j0 = (int32_t)(tq * TQ), nq = (int32_t)p.nq;
    const T *a0 = a + offA + (int64_t)i0 * p.a_p;  // column 0 of the patch's rows
    const T *b0 = b + offB + (int64_t)i0 * p.b_p;
    T *o0 = out + offO + (int64_t)i0 * p.o_p;
    // the rows' phases against the lines: row il starts m(il) elements past one
    const uint32_t m0 = (uint32_t)((reinterpret_cast<uintptr_t>(o0) / sizeof(T)) & (G - 1)), opm = (uint32_t)p.o_p & (G - 1);
    const int64_t direct_q = MA == 1 ? p.b_q : p.a_q;
    const bool dsplat = !kBoth && direct_q == 0;
    const T *d0 = MA == 1 ? b0 : a0;
    const int64_t d_p = MA == 1 ? p.b_p : p.a_p;
    V va[S1], vb[kBoth ? S1 : 1], vd[kBoth ? 1 : S2];
    auto issue = [&](auto nt_tag) {
        constexpr bool NT = decltype(nt_tag)::value;
#pragma unroll
        for (int s = 0; s < S1; ++s) {
            const uint32_t v = threadIdx.x + 256 * s, jl = v / VP, ig = v % VP;
            int32_t j = j0 - G + (int32_t)jl;
            j = j < 0 ? 0 : (j >= nq ? nq - 1 : j);
            if constexpr (kBoth) {
                va[s] = load_stream_as(T, reinterpret_cast<const V *>(a0 + ig * W + (int64_t)j * p.a_q), NT);
                vb[s] = load_stream_as(T, reinterpret_cast<const V *>(b0 + ig * W + (int64_t)j * p.b_q), NT);
            } else if constexpr (MA == 1) {
                va[s] = load_stream_as(T, reinterpret_cast<const V *>(a0 + ig * W + (int64_t)j * p.a_q), NT);
            } else {
                va[s] = load_stream_as(T, reinterpret_cast<const V *>(b0 + ig * W + (int64_t)j * p.b_q), NT);
            }
        }
        if constexpr (!kBoth) {
#pragma unroll
            for (int s = 0; s < S2; ++s) {
                const uint32_t v = threadIdx.x + 256 * s, il = v / VQ, jg = v % VQ;
                if (dsplat) {
                    const T one = d0[(int64_t)il * d_p];
#pragma unroll
                    for (int k = 0; k < W; ++k) vd[s][k] = one;
                } else {
                    int32_t j = j0 + (int32_t)(jg * W) - (int32_t)((m0 + il * opm) & (G - 1));
                    j = j < 0 ? 0 : (j > nq - W ? nq - W : j);
                    vd[s] = load_stream_as(T, reinterpret_cast<const V *>(d0 + (int64_t)il * d_p + j), NT);
                }
            }
        }
    };
    if (p.nt & kLoadNt) issue(BoolTag<true>{});
    else issue(BoolTag<false>{});
#pragma unroll
    for (int s = 0; s < S1; ++s) {
        const uint32_t v = threadIdx.x + 256 * s, jl = v / VP, ig = v % VP;
        T x[W];
        if constexpr (kBoth) {
            T xa[W], xb[W];
#pragma unroll
            for (int k = 0; k < W; ++k) { xa[k] = va[s][k]; xb[k] = vb[s][k]; }
            apply_n<Op, T, W>(ctx, xa, xb, x);
        } else {
#pragma unroll
            for (int k = 0; k < W; ++k) x[k] = va[s][k];
        }
#pragma unroll
        for (int k = 0; k < W; ++k) tile[at(ig * W + k, jl)] = x[k];
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < S2; ++s) {
        const uint32_t v = threadIdx.x + 256 * s, il = v / VQ, jg = v % VQ;
        const int32_t jr = (int32_t)(jg * W) - (int32_t)((m0 + il * opm) & (G - 1));  // against j0; the tile's column 0 is j0 - G
        const int32_t j = j0 + jr;
        T xt[W], xr[W];
#pragma unroll
        for (int k = 0; k < W; ++k) xt[k] = tile[at(il, (uint32_t)(jr + G + k))];
        T *orow = o0 + (int64_t)il * p.o_p;
        if ((uint32_t)j <= (uint32_t)(nq - W)) {  // the whole vector inside the row (j < 0 compares as huge)
            if constexpr (kBoth) {
#pragma unroll
                for (int k = 0; k < W; ++k) xr[k] = xt[k];
            } else {
                T xd[W];
#pragma unroll
                for (int k = 0; k < W; ++k) xd[k] = vd[s][k];
                if constexpr (MA == 1) apply_n<Op, T, W>(ctx, xt, xd, xr);
                else apply_n<Op, T, W>(ctx, xd, xt, xr);
            }
            V val;
#pragma unroll
            for (int k = 0; k < W; ++k) val[k] = xr[k];
            store_stream_if(T, reinterpret_cast<V *>(orow + j), val, p.nt);
        } else {
#pragma unroll
            for (int k = 0; k < W; ++k) {
                if ((uint32_t)(j + k) < (uint32_t)nq) {
                    if constexpr (kBoth) {
                        orow[j + k] = xt[k];
                    } else {
                        const T xd = dsplat ? vd[s][0] : d0[(int64_t)il * d_p + (j + k)];
                        orow[j + k] = MA == 1 ? Op::apply(xt[k], xd) : Op::apply(xd, xt[k]);
                    }
                }
            }
        }
    }
}
#endif  // SMHIP_TILE_SHIFT_FORM
