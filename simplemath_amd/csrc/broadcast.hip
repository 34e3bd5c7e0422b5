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
// The four bodies above live in bcast_kernels.hip.h (shared with jit.hip, which compiles them around user-defined Ops);
// plan_launch() below chooses among them.  Four small kernels in this file cover shapes on which the gather is slow:
//   short-rows     rows of 2-15 elements against one value per row ((N,3) / (N,1)), per-row values staged in LDS;
//   repeat         SMHIP_OP_LEFT over (N, r) with strides (1, 0), r < 16: SMArray::repeat(r) of a dense array;
//   deinterleave   SMHIP_OP_LEFT over a view that takes every 2nd / 3rd / 4th element of the inner axis;
//   strided copy   assignment into a view whose rows have a pitch (smhip_copy_strided); destinations that are dense in
//                  some axis order are written through SMHIP_OP_LEFT instead.
// Every 16-byte access is only element-aligned (VecTraits, ops.hip.h): bases, pitches and row extents are
// unconstrained, there are no per-element fallbacks for alignment.
// Roofline: HBM-bound; algorithmic bytes = sizeof(T) * (|a| + |b| + |out|)
// with each broadcast operand counted once (config 3: 134 234 112 B).
#include <type_traits>

#include "bcast_plan.h"

#ifndef SMHIP_HEAVY_ROWS
#define SMHIP_HEAVY_ROWS 2
#endif
#ifndef SMHIP_FLAT_ROWS_MIN_COLS
#define SMHIP_FLAT_ROWS_MIN_COLS 16
#endif

namespace smhip {
namespace {

using namespace dev;

using namespace bk;

// The ahead-of-time kernels: one __global__ template per body.
template <typename T, typename Op, int INNER_A, int INNER_B, bool CONST_A, bool CONST_B, int TX, int ROWS>
__global__ __launch_bounds__(256) void row_kernel(const T *__restrict__ a, const T *__restrict__ b, T *__restrict__ out, RowParams p) {
    row_body<T, Op, INNER_A, INNER_B, CONST_A, CONST_B, TX, ROWS>(a, b, out, p);
}
template <typename T, typename Op, int W>
__global__ __launch_bounds__(256) void gather_kernel(const T *__restrict__ a, const T *__restrict__ b, T *__restrict__ out, GatherParams p) {
    gather_body<T, Op, W>(a, b, out, p);
}
template <typename T, typename Op, int SA, int SB>
__global__ __launch_bounds__(256) void strided_row_kernel(const T *__restrict__ a, const T *__restrict__ b, T *__restrict__ out, StridedParams p) {
    strided_row_body<T, Op, SA, SB>(a, b, out, p);
}
template <typename T, typename Op, bool SWAPPED, int U>
__global__ __launch_bounds__(256) void dense_lds_kernel(const T *__restrict__ x, const T *__restrict__ y, T *__restrict__ out, LdsParams p) {
    dense_lds_body<T, Op, SWAPPED, U>(x, y, out, p);
}
template <typename T, typename Op, bool VEC, int MA, int MB, int QB>
__global__ __launch_bounds__(256) void tile_kernel(const T *__restrict__ a, const T *__restrict__ b, T *__restrict__ out, TileParams p) {
    tile_body<T, Op, VEC, MA, MB, QB>(a, b, out, p);
}
#if SMHIP_TILE_SHIFT_FORM
template <typename T, typename Op, int MA, int MB, int QB>
__global__ __launch_bounds__(256) void tile_shift_kernel(const T *__restrict__ a, const T *__restrict__ b, T *__restrict__ out, TileParams p) {
    tile_shift_body<T, Op, MA, MB, QB>(a, b, out, p);
}
#endif

// ---------------------------------------------------------------------------- choosing a kernel
// plan_launch() turns a normalised problem into a Launch: which body, which compile-time variant of it, the grid and the
// parameter block.  It depends on the element type only through its size, so the built-in Ops (launch_aot below: a switch
// over template instantiations) and user-defined Ops (jit.hip: the same variant compiled by hipRTC on first use) share it.
}  // namespace

namespace bk {

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

int plan_launch(const Plan &pl, int esz, bool heavy, Launch *L) {
    const int W = 16 / esz;
    const int nd = pl.ndim;
    const int64_t ia = pl.sa[nd - 1], ib = pl.sb[nd - 1];
    const int64_t inner = pl.shape[nd - 1];
    const size_t rows = pl.n / (size_t)inner;
    *L = Launch{};

    const bool row_ok = nd >= 2 && (ia == 0 || ia == 1) && (ib == 0 || ib == 1) && (ia | ib) != 0 && inner >= 16 &&
                        rows < 0x7fffffffull && inner < 0x7fffffffll;
    if (row_ok) {
        RowParams &p = L->p.row;
        p = RowParams{};
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
        {   // bytes the launch reads: each operand's distinct elements (a stride-0 axis is read once)
            size_t da = 1, db = 1;
            for (int d = 0; d < nd; ++d) {
                if (pl.sa[d] != 0) da *= (size_t)pl.shape[d];
                if (pl.sb[d] != 0) db *= (size_t)pl.shape[d];
            }
            p.nt = (uint32_t)stream_policy((da + db) * (size_t)esz, pl.n * (size_t)esz);
        }
        L->kind = Launch::kRow;
        L->ia = (int)ia;
        L->ib = (int)ib;
        // inner (1,1): neither, one or the other operand constant over rows; (1,0)/(0,1): the broadcast side may be
        // row-constant only together with being a scalar, which normalise() has folded away -- CONST is for dense sides
        if (ia == 1 && ib == 1) {
            if (cb && !ca) L->cb = true;
            else if (ca && !cb) L->ca = true;
        } else if (ia == 1) {
            L->ca = ca;
        } else {
            L->cb = cb;
        }
        L->tx = p.vpr > 128 ? 256 : p.vpr > 64 ? 128 : p.vpr > 32 ? 64 : p.vpr > 16 ? 32 : 16;
        // Rows per lane (tools/bcast_matrix.py, profiles/r01_bcast_matrix.txt): two when one side is a row-constant or a
        // per-row scalar -- two independent 16-byte loads in flight per lane, 83 % of peak on config 3 against 80 % with
        // one or four; one when both operands stream (three full streams behave like the contiguous kernel: 80 % vs
        // 74-77 %).
        const bool three_streams = ia == 1 && ib == 1 && !L->ca && !L->cb;
        // no streamed read at all (a per-row scalar against a row-constant: an outer product) is a pure write stream: one
        // row per lane, like the fill kernel (83 % of peak against 78 % with two)
        const bool write_only = (ia == 0 && L->cb) || (ib == 0 && L->ca);
        // float / double pow: SMHIP_HEAVY_ROWS rows per lane (round 1: one; two loads in flight per lane win there too)
        L->rows = (three_streams || write_only) ? 1 : (heavy ? SMHIP_HEAVY_ROWS : 2);
        const int ty = 256 / L->tx;
        const size_t gx = (p.vpr + L->tx - 1) / L->tx;
        const size_t gy = ((size_t)p.rows + ty * L->rows - 1) / (ty * L->rows);
        if (gx * gy > 0x7fffffffull) return fail(SMHIP_ERR_UNSUPPORTED, "row kernel: more than 2^31 workgroups");
        p.grid_x = (uint32_t)gx;
        L->grid = (unsigned)(gx * gy);
        return SMHIP_OK;
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
        if (a_dense && !b_dense && span_b * esz <= kLdsBytes) pick = 0;
        else if (b_dense && !a_dense && span_a * esz <= kLdsBytes) pick = 1;
        if (pick >= 0) {
            LdsParams &lp = L->p.lds;
            lp = LdsParams{};
            lp.ndim = nd;
            lp.n = (uint32_t)pl.n;
            lp.n_vec = (uint32_t)(pl.n / W);
            lp.y_span = (uint32_t)(pick == 0 ? span_b : span_a);
            lp.nt = (uint32_t)stream_policy(pl.n * (size_t)esz, pl.n * (size_t)esz);
            for (int d = 0; d < nd; ++d) {
                const int src = nd - 1 - d;
                lp.shape[d] = FastDiv((uint32_t)pl.shape[src]);
                lp.sy[d] = (uint32_t)(pick == 0 ? pl.sb[src] : pl.sa[src]);
                lp.rewind[d] = (uint32_t)pl.shape[src] * lp.sy[d];
            }
            // a small staged operand (<= 4 KiB) is re-staged by every workgroup of a one-shot launch, which lets
            // the hardware dispatcher balance the stream; a larger one is staged once per persistent workgroup
            const size_t lds = (size_t)lp.y_span * esz;
            constexpr int U = kLdsVectorsInFlight;
            const size_t want = ((size_t)lp.n_vec + 256 * U - 1) / (256 * U);
            const size_t cap = lds <= 4096 ? want : (size_t)compute_units() * 8;
            size_t blocks = want < cap ? want : cap;
            if (blocks == 0) blocks = 1;  // fewer than W elements: the tail lanes of one workgroup do them
            L->kind = Launch::kLds;
            L->swapped = pick == 1;
            L->grid = (unsigned)blocks;
            L->lds_bytes = lds;
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
            TileParams &t = L->p.tile;
            t = TileParams{};
            const bool vec = true;  // extents need not be multiples of the vector width: slots over the plane's edge go element by element (until round 3 they put the WHOLE problem on the element form: 8191 x 8191 at 28 % where 8192 x 8192 ran at 86 %)
            // The patch and its walk (tools/tile_shapes.py, profiles/r03_tile_shapes.txt).  Up to the Infinity Cache's size per
            // array: 64 x 512 B patches along a diagonal.  Beyond: 64 x 1024 B patches, row-major, so that the workgroups in
            // flight together write (and read the direct operand in) whole rows of the output -- where both plane extents
            // give that patch and its walk something to work with: skinny planes lose badly ((4194304, 32): 40 -> 15 %,
            // (1048576, 128): 77 -> 38 %, (256, 524288): 71 -> 63 %; from 512 x 512 on the two are level).  A q extent of at
            // most 256 bytes (or of 384 bytes): 64 x 128 B patches ((4194304, 32): 40 -> 78 %, (2097152, 16): 33 -> 82 %).
            // SMHIP_TILE_QB = 128 / 512 / 1024 forces one (tests, sweeps); SMHIP_TILE_WIDE = 0 / 1 is the older spelling of 512 / 1024.
            static const int forced = [] {
                if (const char *e = getenv("SMHIP_TILE_QB"); e && *e) { const int v = atoi(e); return v == kTileQBytesShort || v == kTileQBytesWide ? v : kTileQBytes; }
                if (const char *e = getenv("SMHIP_TILE_WIDE"); e && *e) return atoi(e) != 0 ? kTileQBytesWide : kTileQBytes;
                return 0;
            }();
            const bool roomy = pl.shape[paxis] >= 512 && inner * esz >= 2 * kTileQBytesWide;
            const int qb = !vec ? kTileQBytes
                         : forced ? forced
                         : (inner * esz <= 2 * kTileQBytesShort || (inner * esz < kTileQBytes && (inner * esz) % kTileQBytesShort == 0)) ? kTileQBytesShort  // ... or a whole number of short patch rows under 512 bytes ((B, 96, 96): 58 -> 82-84 %; 100 columns: 58 -> 32 %, so not those)
                         : roomy && pl.n * (size_t)esz > kInfinityCacheBytes ? kTileQBytesWide : kTileQBytes;
            const int tq = qb / esz;  // tile_q<T, QB>()
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
            {
                size_t da = 1, db = 1;
                for (int d = 0; d < nd; ++d) {
                    if (pl.sa[d] != 0) da *= (size_t)pl.shape[d];
                    if (pl.sb[d] != 0) db *= (size_t)pl.shape[d];
                }
                t.nt = (uint32_t)stream_policy((da + db) * (size_t)esz, pl.n * (size_t)esz);
            }
            t.tiles_p = (t.np + kTileP - 1) / kTileP;
            t.tiles_q = (t.nq + tq - 1) / tq;
            const size_t blocks = slices * t.tiles_p * t.tiles_q;
            if (blocks < 0x7fffffffull && pl.shape[paxis] < 0x7fffffffll && inner < 0x7fffffffll) {
                // extents, bases and pitches may be anything (the 16-byte form moves the slots over the edges element by element)
                L->kind = Launch::kTile;
                L->vec = vec;
                L->qb = qb;
                static const int forced_order = [] { const char *e = getenv("SMHIP_TILE_ORDER"); return e && *e ? atoi(e) : -1; }();
                // output rows that do not start on 128-byte lines (an inner extent like 8190): neighbouring patches share the
                // lines at their seams, and the row-major walk runs them back to back (8190 x 8190: 61 -> 77 %, cold 48 -> 61 %)
                const bool ragged_rows = ((size_t)inner * (size_t)esz) % 128 != 0;
                // (order 2 -- the row-major walk in eight runs, one per XCD, so that the patches on either side of a seam meet in one
                // L2 -- was measured for ragged rows and NOT adopted: 8191^2 75.7 -> 77.7 %, but 12287^2 60.3 -> 56.3 % and 16383^2
                // 60 -> 57.6 %; SMHIP_TILE_ORDER=2 selects it; profiles/r04_pmc_tile_odd.txt)
                t.order = forced_order >= 0 ? (uint32_t)forced_order : (qb == kTileQBytesWide || ragged_rows) ? 0 : 1;  // SMHIP_TILE_ORDER: for tools/tile_variants.sh
                L->ma = L->vec ? t.mode_a : 0;
                L->mb = L->vec ? t.mode_b : 0;
                L->grid = (unsigned)blocks;
                if (t.order == 3) {  // blocks of BP x BQ patches dealt to the XCDs in turn (SMHIP_TILE_BLOCK = BP * 16 + BQ in hex digits: 24 = 2 x 4)
                    static const int blk = [] { const char *e = getenv("SMHIP_TILE_BLOCK"); return e && *e ? (int)strtol(e, nullptr, 16) : 0x24; }();
                    const uint32_t BP = (uint32_t)(blk >> 4) & 15u ? (uint32_t)(blk >> 4) & 15u : 1u, BQ = (uint32_t)blk & 15u ? (uint32_t)blk & 15u : 1u;
                    const size_t rows = slices * t.tiles_p, nblocks = ((rows + BP - 1) / BP) * ((t.tiles_q + BQ - 1) / BQ);
                    const size_t g = (nblocks + 7) / 8 * 8 * BP * BQ;
                    if (g < 0x7fffffffull) { t.total = BP | (BQ << 8); L->grid = (unsigned)g; }
                    else t.order = 0;
                }
                if (t.order == 2) {  // eight runs of the row-major walk, one per XCD (bcast_kernels.hip.h: tile_body)
                    t.total = (uint32_t)blocks;
                    L->grid = (unsigned)((blocks + 7) / 8 * 8);
                    if (blocks + 8 >= 0x7fffffffull) { t.order = 0; L->grid = (unsigned)blocks; }
                }
                return SMHIP_OK;
            }
        }
    }

    // An operand that steps over the inner axis by 2-4 elements, the other side dense, one value per row, or stepping
    // by the same amount: rows of wide loads + select (strided_row_body).
    {
        const int64_t big = ia > ib ? ia : ib, small = ia > ib ? ib : ia;
        const size_t vpr = ((size_t)inner + W - 1) / W;
        if (big >= 2 && big <= 4 && (small == big || small == 0 || small == 1) && inner >= 64 && rows * vpr < 0x7fffffffull &&
            inner < 0x7fffffffll / 4) {
            StridedParams &p = L->p.strided;
            p = StridedParams{};
            p.n_outer = nd - 1;
            p.inner = (uint32_t)inner;
            for (int k = 0; k < nd - 1; ++k) {  // innermost-outer first
                const int src = nd - 2 - k;
                p.shape[k] = FastDiv((uint32_t)pl.shape[src]);
                p.sa[k] = pl.sa[src];
                p.sb[k] = pl.sb[src];
            }
            p.vpr = FastDiv((uint32_t)vpr);
            p.slots = (uint32_t)(rows * vpr);
            size_t blocks = (rows * vpr + 255) / 256;
            if (big == 2 || (big == 4 && esz == 4)) {  // strided2_row_body / strided4_row_body: a WAVE owns 64 W consecutive outputs (four-byte types: 256) of a row, four waves per workgroup
                const size_t per_chunk = (size_t)32 * W * kStrided2Groups;
                const size_t cpr = ((size_t)inner + per_chunk - 1) / per_chunk;
                p.vpr = FastDiv((uint32_t)cpr);
                p.slots = (uint32_t)(rows * cpr);
                blocks = (rows * cpr + 3) / 4;
            }
            {   // lines fetched: a strided operand touches `stride` times its elements
                size_t da = 1, db = 1;
                for (int d = 0; d < nd - 1; ++d) {
                    if (pl.sa[d] != 0) da *= (size_t)pl.shape[d];
                    if (pl.sb[d] != 0) db *= (size_t)pl.shape[d];
                }
                da *= ia == 0 ? 1 : (size_t)inner * (size_t)ia;
                db *= ib == 0 ? 1 : (size_t)inner * (size_t)ib;
                p.nt = (uint32_t)stream_reads((da + db) * (size_t)esz);
            }
            L->kind = Launch::kStrided;
            L->ia = (int)ia;
            L->ib = (int)ib;
            L->grid = (unsigned)blocks;
            return SMHIP_OK;
        }
    }

    if (pl.n >= 0x7fffffffull)
        return fail(SMHIP_ERR_UNSUPPORTED, "gather path limited to < 2^31 elements (got %zu)", pl.n);
    GatherParams &g = L->p.gather;
    g = GatherParams{};
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
    L->kind = Launch::kGather;
    L->w = (ia > 1 || ib > 1) ? 1 : W;
    L->grid = (unsigned)(((pl.n + L->w - 1) / L->w + 255) / 256);
    return SMHIP_OK;
}

}  // namespace bk

namespace {

// Built-in Ops: the Launch's variant picks a template instantiation.
template <typename T, typename Op>
int launch_aot(const Launch &L, const void *a_, const void *b_, void *out_, hipStream_t s) {
    constexpr int W = VecTraits<T>::width;
    const T *a = static_cast<const T *>(a_), *b = static_cast<const T *>(b_);
    T *out = static_cast<T *>(out_);
    const dim3 grid(L.grid), block(256);
    switch (L.kind) {
        case Launch::kRow: {
            constexpr int kRows = (std::is_same<Op, PowOp<T>>::value && std::is_floating_point<T>::value) ? SMHIP_HEAVY_ROWS : 2;  // what plan_launch gives non-three-stream forms
            bool launched = false;
            auto go_tx = [&](auto ia_t, auto ib_t, auto ca_t, auto cb_t, auto rows_t) {
                constexpr int IA = decltype(ia_t)::value, IB = decltype(ib_t)::value, ROWS = decltype(rows_t)::value;
                constexpr bool CA = decltype(ca_t)::value, CB = decltype(cb_t)::value;
                if (ROWS != L.rows) return;  // the grid was sized for L.rows: refuse rather than cover the wrong rows
                launched = true;
                switch (L.tx) {
                    case 256: hipLaunchKernelGGL((row_kernel<T, Op, IA, IB, CA, CB, 256, ROWS>), grid, block, 0, s, a, b, out, L.p.row); break;
                    case 128: hipLaunchKernelGGL((row_kernel<T, Op, IA, IB, CA, CB, 128, ROWS>), grid, block, 0, s, a, b, out, L.p.row); break;
                    case 64: hipLaunchKernelGGL((row_kernel<T, Op, IA, IB, CA, CB, 64, ROWS>), grid, block, 0, s, a, b, out, L.p.row); break;
                    case 32: hipLaunchKernelGGL((row_kernel<T, Op, IA, IB, CA, CB, 32, ROWS>), grid, block, 0, s, a, b, out, L.p.row); break;
                    default: hipLaunchKernelGGL((row_kernel<T, Op, IA, IB, CA, CB, 16, ROWS>), grid, block, 0, s, a, b, out, L.p.row); break;
                }
            };
            using I0 = std::integral_constant<int, 0>;
            using I1 = std::integral_constant<int, 1>;
            using RN = std::integral_constant<int, kRows>;
            using F = std::false_type;
            using Tr = std::true_type;
            if (L.ia == 1 && L.ib == 1) {
                if (L.cb) go_tx(I1{}, I1{}, F{}, Tr{}, RN{});
                else if (L.ca) go_tx(I1{}, I1{}, Tr{}, F{}, RN{});
                else go_tx(I1{}, I1{}, F{}, F{}, I1{});  // three streams: one row per lane
            } else if (L.ia == 1) {
                if (L.ca) go_tx(I1{}, I0{}, Tr{}, F{}, I1{});  // write-only: one row per lane
                else go_tx(I1{}, I0{}, F{}, F{}, RN{});
            } else {
                if (L.cb) go_tx(I0{}, I1{}, F{}, Tr{}, I1{});  // write-only: one row per lane
                else go_tx(I0{}, I1{}, F{}, F{}, RN{});
            }
            if (!launched) return fail(SMHIP_ERR_INVALID, "row kernel: rows-per-lane mismatch between plan and launch");
            SMHIP_LAUNCH_CHECK("row_kernel");
            return SMHIP_OK;
        }
        case Launch::kLds:
            if (!L.swapped) hipLaunchKernelGGL((dense_lds_kernel<T, Op, false, kLdsVectorsInFlight>), grid, block, L.lds_bytes, s, a, b, out, L.p.lds);
            else hipLaunchKernelGGL((dense_lds_kernel<T, Op, true, kLdsVectorsInFlight>), grid, block, L.lds_bytes, s, b, a, out, L.p.lds);
            SMHIP_LAUNCH_CHECK("dense_lds_kernel");
            return SMHIP_OK;
        case Launch::kTile:
            if (!L.vec) return fail(SMHIP_ERR_INVALID, "tile kernel: the plan always asks for the 16-byte form");  // (the one-element-per-slot form is no longer built: the 16-byte form takes every extent)
            else {
#if SMHIP_TILE_SHIFT_FORM
                // Output rows off the 128-byte lines, past the Infinity Cache: patches whose rows are cut at LINES, not columns
                // (bcast_kernels.hip.h: tile_shift_body) -- measured and not adopted, see there; built only with -DSMHIP_TILE_SHIFT_FORM=1.
                {
                    const TileParams &t = L.p.tile;
                    static const int shift_mode = [] { const char *e = getenv("SMHIP_TILE_SHIFT"); return e && *e ? atoi(e) : 0; }();  // 0: off (default), 1: rows off the lines, 2: also for rows ON the lines (the form's own cost)
                    const bool shift_on = shift_mode != 0;
                    constexpr uint32_t TQ = kTileQBytesWide / sizeof(T), G = kTileShiftBytes / sizeof(T);
                    const bool both = L.ma == 1 && L.mb == 1;
                    const int64_t direct_q = L.ma == 1 ? t.b_q : t.a_q;
                    if (shift_on && L.qb == kTileQBytesWide && t.order == 0 && !t.in_place && t.np >= 4u * kTileP && t.nq >= 2u * TQ &&
                        (((size_t)t.nq * sizeof(T)) % kTileShiftBytes != 0 || shift_mode == 2) && (both || direct_q == 1 || direct_q == 0)) {
                        TileParams ts = t;
                        ts.tiles_q = (t.nq + G - 1 + TQ - 1) / TQ;
                        const size_t blocks = (size_t)L.grid / ((size_t)t.tiles_p * t.tiles_q) * ts.tiles_p * ts.tiles_q;
                        if (blocks < 0x7fffffffull) {
                            const dim3 sgrid((unsigned)blocks);
                            if (both) hipLaunchKernelGGL((tile_shift_kernel<T, Op, 1, 1, kTileQBytesWide>), sgrid, block, 0, s, a, b, out, ts);
                            else if (L.ma == 1) hipLaunchKernelGGL((tile_shift_kernel<T, Op, 1, 0, kTileQBytesWide>), sgrid, block, 0, s, a, b, out, ts);
                            else hipLaunchKernelGGL((tile_shift_kernel<T, Op, 0, 1, kTileQBytesWide>), sgrid, block, 0, s, a, b, out, ts);
                            SMHIP_LAUNCH_CHECK("tile_shift_kernel");
                            return SMHIP_OK;
                        }
                    }
                }
#endif
                auto go = [&](auto qb_tag) {
                    constexpr int QB = decltype(qb_tag)::value;
                    if (L.ma == 1 && L.mb == 1) hipLaunchKernelGGL((tile_kernel<T, Op, true, 1, 1, QB>), grid, block, 0, s, a, b, out, L.p.tile);
                    else if (L.ma == 1) hipLaunchKernelGGL((tile_kernel<T, Op, true, 1, 0, QB>), grid, block, 0, s, a, b, out, L.p.tile);
                    else hipLaunchKernelGGL((tile_kernel<T, Op, true, 0, 1, QB>), grid, block, 0, s, a, b, out, L.p.tile);
                };
                if (L.qb == kTileQBytesWide) go(std::integral_constant<int, kTileQBytesWide>{});
                else if (L.qb == kTileQBytesShort) go(std::integral_constant<int, kTileQBytesShort>{});
                else go(std::integral_constant<int, kTileQBytes>{});
            }
            SMHIP_LAUNCH_CHECK("tile_kernel");
            return SMHIP_OK;
        case Launch::kStrided: {
            bool launched = false;
            auto go = [&](auto sa_t, auto sb_t) {
                constexpr int SA = decltype(sa_t)::value, SB = decltype(sb_t)::value;
                if (L.ia == SA && L.ib == SB) {
                    hipLaunchKernelGGL((strided_row_kernel<T, Op, SA, SB>), grid, block, 0, s, a, b, out, L.p.strided);
                    launched = true;
                }
            };
            using I0 = std::integral_constant<int, 0>;
            using I1 = std::integral_constant<int, 1>;
            using I2 = std::integral_constant<int, 2>;
            using I3 = std::integral_constant<int, 3>;
            using I4 = std::integral_constant<int, 4>;
            go(I2{}, I2{}); go(I2{}, I1{}); go(I1{}, I2{}); go(I2{}, I0{}); go(I0{}, I2{});
            go(I3{}, I3{}); go(I3{}, I1{}); go(I1{}, I3{}); go(I3{}, I0{}); go(I0{}, I3{});
            go(I4{}, I4{}); go(I4{}, I1{}); go(I1{}, I4{}); go(I4{}, I0{}); go(I0{}, I4{});
            if (!launched) return fail(SMHIP_ERR_INVALID, "strided rows: no kernel for inner strides %d / %d", L.ia, L.ib);
            SMHIP_LAUNCH_CHECK("strided_row_kernel");
            return SMHIP_OK;
        }
        case Launch::kGather:
            if (L.w == 1) hipLaunchKernelGGL((gather_kernel<T, Op, 1>), grid, block, 0, s, a, b, out, L.p.gather);
            else hipLaunchKernelGGL((gather_kernel<T, Op, W>), grid, block, 0, s, a, b, out, L.p.gather);
            SMHIP_LAUNCH_CHECK("gather_kernel");
            return SMHIP_OK;
    }
    return fail(SMHIP_ERR_INVALID, "broadcast: no kernel chosen");
}

// ------------------------------------------------------------------ flat repeat
// out[j] = src[j / r] for a small repeat count r (SMArray::repeat(r) of a dense array, or repeat(r, last axis)): as a
// broadcast problem it is (N, r) with strides (1, 0), whose inner extent is too short for the row kernel, and the generic
// gather spends its time on index bookkeeping (issue-bound, 62 % of peak).  Here a lane owns one 16-byte output vector:
// one fast division finds the first source element, and the W outputs need at most W distinct sources.
template <typename T>
__global__ __launch_bounds__(256) void repeat_kernel(const T *__restrict__ src, T *__restrict__ out, FastDiv r, uint32_t n_vec, uint32_t n) {
    constexpr int W = VecTraits<T>::width;
    typedef typename VecTraits<T>::vec_t V;
    const uint32_t v = blockIdx.x * 256u + threadIdx.x;
    if (v > n_vec) return;
    const uint32_t j0 = v * W;
    uint32_t q, rem;
    r.divmod(j0, q, rem);
    if (v < n_vec) {
        V res;
        T cur = src[q];
#pragma unroll
        for (int k = 0; k < W; ++k) {
            res[k] = cur;
            if (++rem == r.d && k + 1 < W) {  // next output starts the next source element
                rem = 0;
                cur = src[++q];
            }
        }
        store_stream(reinterpret_cast<V *>(out + j0), res);
    } else {
        for (uint32_t j = j0; j < n; ++j) out[j] = src[j / r.d];
    }
}

template <typename T>
int run_repeat(const void *src, void *out, size_t n_src, uint32_t r, hipStream_t s) {
    constexpr int W = VecTraits<T>::width;
    const size_t n = n_src * r;
    const uint32_t n_vec = (uint32_t)(n / W);
    const unsigned grid = (unsigned)(((size_t)n_vec + 1 + 255) / 256);
    hipLaunchKernelGGL(repeat_kernel<T>, dim3(grid), dim3(256), 0, s, static_cast<const T *>(src), static_cast<T *>(out), FastDiv(r), n_vec, (uint32_t)n);
    SMHIP_LAUNCH_CHECK("repeat_kernel");
    return SMHIP_OK;
}

// ------------------------------------------------------------------ dense copy of an inner-strided view
// out[row][i] = src[row * pitch + i * S] for S = 2, 3, 4 (every other column; one channel of interleaved data):
// SMArray::contiguous() of such a view, which sm::expr, the reductions and repeat() call before they run.  The generic
// gather reads it one element per lane (four 4-byte load instructions per 16 bytes of output); here a lane loads the S
// consecutive vectors that contain its W outputs and keeps every S-th element: S full-width loads per 16 bytes of
// output, every fetched line requested once.  The last slot of a row is done element by element (the S vectors would
// read up to S - 1 elements past the last one the view owns).
template <typename T, int S>
__global__ __launch_bounds__(256) void deinterleave_kernel(const T *__restrict__ src, T *__restrict__ out, int64_t pitch, uint32_t inner,
                                                           FastDiv vpr, uint32_t slots, int nt) {
    constexpr int W = VecTraits<T>::width;
    typedef typename VecTraits<T>::vec_t V;
    const uint32_t slot = blockIdx.x * 256u + threadIdx.x;
    if (slot >= slots) return;
    uint32_t row, col;
    vpr.divmod(slot, row, col);
    const uint32_t e0 = col * W;
    const T *s = src + (int64_t)row * pitch + (int64_t)e0 * S;
    T *d = out + (size_t)row * inner + e0;
    if (e0 + W < inner) {  // strictly inside the row: the over-read stays inside it too
        V v[S];
        if (nt & kLoadNt) {  // one branch around the S loads (see load_stream_as)
#pragma unroll
            for (int i = 0; i < S; ++i) v[i] = load_stream_as(T, reinterpret_cast<const V *>(s) + i, true);
        } else {
#pragma unroll
            for (int i = 0; i < S; ++i) v[i] = load_stream_as(T, reinterpret_cast<const V *>(s) + i, false);
        }
        V r;
#pragma unroll
        for (int k = 0; k < W; ++k) r[k] = v[(k * S) / W][(k * S) % W];
        store_stream_if(T, reinterpret_cast<V *>(d), r, nt);
    } else {
        for (uint32_t k = 0; e0 + k < inner; ++k) d[k] = s[(int64_t)k * S];
    }
}

template <typename T>
int run_deinterleave(const void *src, void *out, size_t rows, uint32_t inner, int64_t pitch, int stride, hipStream_t s) {
    constexpr int W = VecTraits<T>::width;
    const size_t per_row = ((size_t)inner + W - 1) / W, slots = rows * per_row;
    const dim3 grid((unsigned)((slots + 255) / 256)), block(256);
    const T *a = static_cast<const T *>(src);
    T *o = static_cast<T *>(out);
    const FastDiv vpr((uint32_t)per_row);
    const int nt = stream_policy(rows * (size_t)inner * (size_t)stride * sizeof(T), rows * (size_t)inner * sizeof(T));  // reads: the lines it fetches
    switch (stride) {
        case 2: hipLaunchKernelGGL((deinterleave_kernel<T, 2>), grid, block, 0, s, a, o, pitch, inner, vpr, (uint32_t)slots, nt); break;
        case 3: hipLaunchKernelGGL((deinterleave_kernel<T, 3>), grid, block, 0, s, a, o, pitch, inner, vpr, (uint32_t)slots, nt); break;
        default: hipLaunchKernelGGL((deinterleave_kernel<T, 4>), grid, block, 0, s, a, o, pitch, inner, vpr, (uint32_t)slots, nt); break;
    }
    SMHIP_LAUNCH_CHECK("deinterleave_kernel");
    return SMHIP_OK;
}

// ------------------------------------------------------------------ short rows against one value per row
// out[i][k] = x[i][k] op y[i] for rows of fewer than 16 elements (per-pixel / per-sample scaling of interleaved data:
// (N, 3) / (N, 1)).  x and out are dense, y is a dense vector.  A lane owns one 16-byte vector of x and finds its row with
// one fast division; the workgroup's slice of y (at most 1024 / r + 2 values) is staged in LDS with coalesced loads --
// the generic gather and a first version of this kernel read y with per-lane scalar loads and sat at 67 % of peak.
template <typename T, typename Op, bool SWAPPED>
__global__ __launch_bounds__(256) void short_rows_kernel(const T *__restrict__ x, const T *__restrict__ y, T *__restrict__ out, FastDiv r,
                                                         uint32_t n_vec, uint32_t n, int nt) {
    constexpr int W = VecTraits<T>::width;
    typedef typename VecTraits<T>::vec_t V;
    __shared__ T ylds[256 * W / 2 + 2];
    OpCtx<Op> ctx;
    ctx.init();
    const uint32_t v0 = blockIdx.x * 256u, v = v0 + threadIdx.x;
    const uint32_t q0 = r.div(v0 * W);
    uint32_t last = (v0 + 256u) * W - 1;  // last element this workgroup can touch (FastDiv wants it < 2^31: clip to n - 1)
    if (last >= n) last = n - 1;
    const uint32_t q_last = r.div(last) + 1;
    V xv;
    if (v < n_vec) xv = load_stream_if(T, reinterpret_cast<const V *>(x + (size_t)v * W), nt);
    for (uint32_t i = threadIdx.x; q0 + i < q_last; i += 256) ylds[i] = y[q0 + i];
    __syncthreads();
    if (v > n_vec) return;
    const uint32_t j0 = v * W;
    uint32_t q, rem;
    r.divmod(j0, q, rem);
    if (v < n_vec) {
        T xa[W], ya[W], res[W];
#pragma unroll
        for (int k = 0; k < W; ++k) {
            xa[k] = xv[k];
            ya[k] = ylds[q - q0];
            if (++rem == r.d) { rem = 0; ++q; }
        }
        if (SWAPPED) apply_n<Op, T, W>(ctx, ya, xa, res);
        else apply_n<Op, T, W>(ctx, xa, ya, res);
        V rv;
#pragma unroll
        for (int k = 0; k < W; ++k) rv[k] = res[k];
        store_stream_if(T, reinterpret_cast<V *>(out + j0), rv, nt);
    } else {
        for (uint32_t j = j0; j < n; ++j) out[j] = SWAPPED ? Op::apply(y[j / r.d], x[j]) : Op::apply(x[j], y[j / r.d]);
    }
}

// Short rows of a VIEW: out (rows, c) dense = a op b with both operands stepping by one along the row and by their own pitch
// from row to row (a column slice A[:, :c] of a wider array, or a dense array: pitch = c).  The row kernel gives a row to a
// few lanes and leaves most of a wave idle on rows of 17 ... 127 elements (44-67 % of peak, tools/short_inner.py); here the
// OUTPUT is walked flat, one 16-byte vector per lane: one fast division finds the vector's row and column, a dense operand
// is one vector load, a pitched one W single loads that follow the row boundaries.
// AK / BK: 0 the operand steps by its pitch from row to row, 1 it is dense (pitch = c), 2 it is ONE value (view op scalar).
template <typename T, typename Op, int AK, int BK>
__global__ __launch_bounds__(256) void pitched_rows_kernel(const T *__restrict__ a, const T *__restrict__ b, T *__restrict__ out, FastDiv c,
                                                           int64_t pa, int64_t pb, uint32_t n_vec, uint32_t n, int nt) {
    constexpr int W = VecTraits<T>::width;
    typedef typename VecTraits<T>::vec_t V;
    OpCtx<Op> ctx;
    ctx.init();
    const uint32_t v = blockIdx.x * 256u + threadIdx.x;
    if (v > n_vec) return;
    const uint32_t e0 = v * W;
    uint32_t row, col;
    c.divmod(e0, row, col);
    const bool whole = c.d % W == 0;  // uniform
    if (v < n_vec) {
        T xa[W], xb[W], res[W];
        auto fetch = [&](const T *base, int64_t pitch, auto kind_tag, T (&dst)[W]) {
            if constexpr (decltype(kind_tag)::value == 2) {
                const T one = *base;
#pragma unroll
                for (int k = 0; k < W; ++k) dst[k] = one;
            } else if constexpr (decltype(kind_tag)::value == 1) {
                const V val = load_stream_if(T, reinterpret_cast<const V *>(base + e0), nt);
#pragma unroll
                for (int k = 0; k < W; ++k) dst[k] = val[k];
            } else if (whole && pitch % W == 0 && (reinterpret_cast<uintptr_t>(base) & 15u) == 0) {  // rows of whole vectors on 16-byte boundaries (misaligned vector loads lose to W single loads here: 79 against 87 %)
                const V val = load_stream_if(T, reinterpret_cast<const V *>(base + (int64_t)row * pitch + col), nt);
#pragma unroll
                for (int k = 0; k < W; ++k) dst[k] = val[k];
            } else {
                uint32_t r = row, cc = col;
#pragma unroll
                for (int k = 0; k < W; ++k) {
                    dst[k] = base[(int64_t)r * pitch + cc];
                    if (++cc == c.d) { cc = 0; ++r; }
                }
            }
        };
        fetch(a, pa, IntTag<AK>{}, xa);
        fetch(b, pb, IntTag<BK>{}, xb);
        apply_n<Op, T, W>(ctx, xa, xb, res);
        V rv;
#pragma unroll
        for (int k = 0; k < W; ++k) rv[k] = res[k];
        store_stream_if(T, reinterpret_cast<V *>(out + e0), rv, nt);
    } else {
        uint32_t r = row, cc = col;
        for (uint32_t e = e0; e < n; ++e) {
            out[e] = Op::apply(AK == 2 ? *a : a[(int64_t)r * pa + cc], BK == 2 ? *b : b[(int64_t)r * pb + cc]);
            if (++cc == c.d) { cc = 0; ++r; }
        }
    }
}

template <typename T, typename Op>
int run_pitched_rows(const void *a, const void *b, void *out, size_t rows, uint32_t c, int64_t pa, int64_t pb, hipStream_t s) {
    constexpr int W = VecTraits<T>::width;
    const size_t n = rows * c;
    const uint32_t n_vec = (uint32_t)(n / W);
    const dim3 grid((unsigned)(((size_t)n_vec + 1 + 255) / 256)), block(256);
    const T *ta = static_cast<const T *>(a), *tb = static_cast<const T *>(b);
    const int nt = stream_policy({Span{a, pa ? (size_t)((rows - 1) * pa + c) * sizeof(T) : sizeof(T)}, Span{b, pb ? (size_t)((rows - 1) * pb + c) * sizeof(T) : sizeof(T)}}, Span{out, n * sizeof(T)});
    // kinds: 2 = one value (pitch 0 from the caller), 1 = dense, 0 = pitched
    const int ak = pa == 0 ? 2 : pa == (int64_t)c ? 1 : 0, bk = pb == 0 ? 2 : pb == (int64_t)c ? 1 : 0;
    T *po = static_cast<T *>(out);
    const FastDiv cd(c);
#define SMHIP_GO(AK, BK) hipLaunchKernelGGL((pitched_rows_kernel<T, Op, AK, BK>), grid, block, 0, s, ta, tb, po, cd, pa, pb, n_vec, (uint32_t)n, nt)
    if (ak == 0 && bk == 0) SMHIP_GO(0, 0);
    else if (ak == 0 && bk == 1) SMHIP_GO(0, 1);
    else if (ak == 1 && bk == 0) SMHIP_GO(1, 0);
    else if (ak == 0 && bk == 2) SMHIP_GO(0, 2);
    else if (ak == 2 && bk == 0) SMHIP_GO(2, 0);
    else return fail(SMHIP_ERR_INVALID, "pitched rows: no view among the operands");
#undef SMHIP_GO
    SMHIP_LAUNCH_CHECK("pitched_rows_kernel");
    return SMHIP_OK;
}

template <typename T, typename Op>
int run_short_rows(const void *x, const void *y, void *out, size_t rows, uint32_t r, bool swapped, hipStream_t s) {
    constexpr int W = VecTraits<T>::width;
    const size_t n = rows * r;
    const uint32_t n_vec = (uint32_t)(n / W);
    const dim3 grid((unsigned)(((size_t)n_vec + 1 + 255) / 256)), block(256);
    const int nt = stream_policy((n + rows) * sizeof(T), n * sizeof(T));
    if (swapped) hipLaunchKernelGGL((short_rows_kernel<T, Op, true>), grid, block, 0, s, static_cast<const T *>(x), static_cast<const T *>(y), static_cast<T *>(out), FastDiv(r), n_vec, (uint32_t)n, nt);
    else hipLaunchKernelGGL((short_rows_kernel<T, Op, false>), grid, block, 0, s, static_cast<const T *>(x), static_cast<const T *>(y), static_cast<T *>(out), FastDiv(r), n_vec, (uint32_t)n, nt);
    SMHIP_LAUNCH_CHECK("short_rows_kernel");
    return SMHIP_OK;
}

// ------------------------------------------------------------------ record kernel (AoS <-> SoA)
// A 2-D plane with one TINY extent k and one long extent n, one operand "turned" (its contiguous axis is the other one):
//   SMALL_P:  out (k, n) = t.T op o  with t stored dense (n, k)  -- records of k elements become k rows   (AoS -> SoA)
//   !SMALL_P: out (n, k) = t.T op o  with t stored (k, n)        -- k rows become records of k elements   (SoA -> AoS)
// The tile kernel wants both plane extents >= 16 and wastes most of a 64-row patch on a small one ((32, 4194304): 61 %,
// (16, 4194304): 47 %); below 16 the gather kernel walked these planes at 18-58 % (tools/tile_shapes.py tiny,
// profiles/r03_tile_skinny.txt).  Here a workgroup takes R records.  The side whose memory is ONE contiguous run of R * k
// elements (t for SMALL_P, out and a dense o otherwise) moves as flat 16-byte vectors whatever k is (3, 5, 12 ...); the
// other side moves as 16-byte vectors along its k row segments of R elements; in between sits an LDS tile [k][R] (the
// tile kernel's column skew), written and read element-wise.  R * k <= 8192 four-byte (4096 eight-byte) elements, R a
// multiple of 64: every lane owns eight vector slots on either side, slots past the chunk re-read its last vector
// (clamped index) so that no branch sits between the loads, and only their LDS writes / stores are guarded.  The last,
// partial chunk goes element-wise.  `o` is dense in the plane or ONE value (copies: SMHIP_OP_LEFT).
struct RecordParams {
    uint32_t k, R, pitch; // elements per record, records per workgroup, words between the k rows of the LDS tile
    uint64_t n;           // records
    FastDiv kdiv, vrdiv;  // / k;  / (R / W): vectors per row segment
    int64_t t_pitch;      // !SMALL_P: distance between the k rows of t (SMALL_P: t is dense, record r at r * k)
    int64_t o_pitch;      // SMALL_P: distance between the k rows of o (!SMALL_P: o is dense, record r at r * k)
    int o_scalar;         // o is one value
    uint32_t nt;
    FastDiv chunks;       // workgroups per plane: blockIdx = plane * chunks + chunk
    int64_t t_batch, o_batch;  // distance between the planes of a batch in t and o (the output's planes are dense)
};
template <typename T> constexpr int record_chunk() { return 32768 / (int)sizeof(T); }  // elements per workgroup at most

template <typename T, typename Op, bool SMALL_P, bool T_IS_A>
__global__ __launch_bounds__(256) void record_kernel(const T *__restrict__ t, const T *__restrict__ o, T *__restrict__ out, RecordParams p) {
    constexpr int W = VecTraits<T>::width, CMAX = record_chunk<T>(), S = CMAX / (W * 256);
    typedef typename VecTraits<T>::vec_t V;
    __shared__ T tile[CMAX + CMAX / 32 + 128 * 16];  // k rows of at most R + R / 32 + 16 words, k <= 128
    OpCtx<Op> ctx;
    ctx.init();
    const uint32_t k = p.k, R = p.R, PITCH = p.pitch;
    auto at = [&](uint32_t small, uint32_t rec) -> uint32_t { return small * PITCH + rec + (rec >> 5); };
    uint32_t plane, chunk;
    p.chunks.divmod(blockIdx.x, plane, chunk);
    t += (int64_t)plane * p.t_batch;
    o += (int64_t)plane * p.o_batch;
    out += (uint64_t)plane * p.n * k;
    const uint64_t j0 = (uint64_t)chunk * R;
    const uint32_t Rc = (uint32_t)(p.n - j0 < R ? p.n - j0 : R);
    const T oval = p.o_scalar ? *o : T{};
    auto apply1 = [&](T xt, T xo) { return T_IS_A ? Op::apply(xt, xo) : Op::apply(xo, xt); };
    // flat side: the chunk's R * k elements in memory order;  row side: k segments of R elements
    const T *tflat = t + j0 * k, *oflat = o + j0 * k;
    T *outflat = out + j0 * k;
    if (Rc == R) {  // a whole chunk (uniform)
        const uint32_t NV = R * k / W;  // vectors on either side (R is a multiple of 64)
        V tv[S], ov[S];
        uint32_t rs[S], cs[S];  // row-side coordinates of slot s: small index, first record of its vector
        auto issue = [&](auto nt_tag) {
            constexpr bool NT = decltype(nt_tag)::value;
#pragma unroll
            for (int s = 0; s < S; ++s) {
                const uint32_t v = threadIdx.x + 256u * s, vc = v < NV ? v : NV - 1;
                uint32_t i, jg;
                p.vrdiv.divmod(vc, i, jg);
                rs[s] = i; cs[s] = jg * W;
                if (SMALL_P) {
                    tv[s] = load_stream_as(T, reinterpret_cast<const V *>(tflat) + vc, NT);
                    if (!p.o_scalar) ov[s] = load_stream_as(T, reinterpret_cast<const V *>(o + (int64_t)i * p.o_pitch + j0 + jg * W), NT);
                } else {
                    tv[s] = load_stream_as(T, reinterpret_cast<const V *>(t + (int64_t)i * p.t_pitch + j0 + jg * W), NT);
                    if (!p.o_scalar) ov[s] = load_stream_as(T, reinterpret_cast<const V *>(oflat) + vc, NT);
                }
            }
        };
        if (p.nt & kLoadNt) issue(BoolTag<true>{});
        else issue(BoolTag<false>{});
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const uint32_t v = threadIdx.x + 256u * s;
            if (v < NV) {
                if (SMALL_P) {  // t arrived in memory order: element e of the chunk is (record e / k, index e % k)
                    uint32_t r, i;
                    p.kdiv.divmod(v * W, r, i);
#pragma unroll
                    for (int kk = 0; kk < W; ++kk) {
                        tile[at(i, r)] = tv[s][kk];
                        if (++i == k) { i = 0; ++r; }
                    }
                } else {
#pragma unroll
                    for (int kk = 0; kk < W; ++kk) tile[at(rs[s], cs[s] + kk)] = tv[s][kk];
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const uint32_t v = threadIdx.x + 256u * s;
            if (v < NV) {
                T xt[W], xo[W], xr[W];
                if (SMALL_P) {
#pragma unroll
                    for (int kk = 0; kk < W; ++kk) xt[kk] = tile[at(rs[s], cs[s] + kk)];
                } else {
                    uint32_t r, i;
                    p.kdiv.divmod(v * W, r, i);
#pragma unroll
                    for (int kk = 0; kk < W; ++kk) {
                        xt[kk] = tile[at(i, r)];
                        if (++i == k) { i = 0; ++r; }
                    }
                }
#pragma unroll
                for (int kk = 0; kk < W; ++kk) xo[kk] = p.o_scalar ? oval : ov[s][kk];
                if (T_IS_A) apply_n<Op, T, W>(ctx, xt, xo, xr);
                else apply_n<Op, T, W>(ctx, xo, xt, xr);
                V val;
#pragma unroll
                for (int kk = 0; kk < W; ++kk) val[kk] = xr[kk];
                if (SMALL_P) store_stream_if(T, reinterpret_cast<V *>(out + (uint64_t)rs[s] * p.n + j0 + cs[s]), val, p.nt);
                else store_stream_if(T, reinterpret_cast<V *>(outflat) + v, val, p.nt);
            }
        }
        return;
    }
    // the last, partial chunk: element by element
    const uint32_t C = Rc * k;
    for (uint32_t e = threadIdx.x; e < C; e += 256) {
        if (SMALL_P) {
            const uint32_t r = e / k, i = e - r * k;
            tile[at(i, r)] = tflat[e];
        } else {
            const uint32_t i = e / Rc, r = e - i * Rc;
            tile[at(i, r)] = t[(int64_t)i * p.t_pitch + j0 + r];
        }
    }
    __syncthreads();
    for (uint32_t e = threadIdx.x; e < C; e += 256) {
        if (SMALL_P) {
            const uint32_t i = e / Rc, r = e - i * Rc;
            const T xo = p.o_scalar ? oval : o[(int64_t)i * p.o_pitch + j0 + r];
            out[(uint64_t)i * p.n + j0 + r] = apply1(tile[at(i, r)], xo);
        } else {
            const uint32_t r = e / k, i = e - r * k;
            const T xo = p.o_scalar ? oval : oflat[e];
            outflat[e] = apply1(tile[at(i, r)], xo);
        }
    }
}

// ------------------------------------------------------------------ small planes, batched
// out (B, P, Q) = t op o with t a dense batch of (Q, P) planes read transposed and P * Q small (4 x 4 ... 64 x 64): the tile
// kernel spends a 64-row patch per plane (16 x 16 planes: 25 % of peak, 4 x 4 ... 24 x 24 and 48 x 48, 96 x 96: 50-59 %,
// tools/small_planes.py).  A chunk of G whole planes is ONE contiguous run in t, in o and in out, so everything global is
// flat 16-byte vectors (eight slots per lane, clamped like the record kernel's) and the transposition happens in LDS: t's
// rows are stored with an odd pitch, and an output vector's four elements are read back one row apart.
struct PlanesParams {
    uint32_t P, Q, pitch;   // out plane (P, Q); t plane (Q, P); LDS words between t's rows (P | 1)
    uint32_t G;             // planes per workgroup
    uint64_t planes;        // B
    FastDiv pq, pdiv, qdiv; // / (P Q);  / P;  / Q
    int o_scalar;
    uint32_t nt;
};
template <typename T, typename Op, bool T_IS_A>
__global__ __launch_bounds__(256) void planes_kernel(const T *__restrict__ t, const T *__restrict__ o, T *__restrict__ out, PlanesParams p) {
    constexpr int W = VecTraits<T>::width, CMAX = record_chunk<T>(), S = CMAX / (W * 256);
    typedef typename VecTraits<T>::vec_t V;
    __shared__ T tile[CMAX + CMAX / 2 + 8];
    OpCtx<Op> ctx;
    ctx.init();
    const uint32_t PQ = p.P * p.Q;
    const uint64_t b0 = (uint64_t)blockIdx.x * p.G;
    const uint32_t Gc = (uint32_t)(p.planes - b0 < p.G ? p.planes - b0 : p.G);
    const T *tc = t + b0 * PQ, *oc = o + b0 * PQ;
    T *outc = out + b0 * PQ;
    const T oval = p.o_scalar ? *o : T{};
    auto lds_of_t = [&](uint32_t e) -> uint32_t {  // element e of the chunk in t's order: plane e / PQ, row j, column i
        uint32_t b, r, j, i;
        p.pq.divmod(e, b, r);
        p.pdiv.divmod(r, j, i);
        return (b * p.Q + j) * p.pitch + i;
    };
    if (Gc == p.G) {  // a whole chunk (uniform): G * P * Q elements, a multiple of the vector width
        const uint32_t NV = p.G * PQ / W;
        V tv[S], ov[S];
        auto issue = [&](auto nt_tag) {
            constexpr bool NT = decltype(nt_tag)::value;
#pragma unroll
            for (int s = 0; s < S; ++s) {
                const uint32_t v = threadIdx.x + 256u * s, vc = v < NV ? v : NV - 1;
                tv[s] = load_stream_as(T, reinterpret_cast<const V *>(tc) + vc, NT);
                if (!p.o_scalar) ov[s] = load_stream_as(T, reinterpret_cast<const V *>(oc) + vc, NT);
            }
        };
        if (p.nt & kLoadNt) issue(BoolTag<true>{});
        else issue(BoolTag<false>{});
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const uint32_t v = threadIdx.x + 256u * s;
            if (v < NV) {
                uint32_t b, r, j, i;
                p.pq.divmod(v * W, b, r);
                p.pdiv.divmod(r, j, i);
#pragma unroll
                for (int kk = 0; kk < W; ++kk) {
                    tile[(b * p.Q + j) * p.pitch + i] = tv[s][kk];
                    if (++i == p.P) { i = 0; if (++j == p.Q) { j = 0; ++b; } }
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const uint32_t v = threadIdx.x + 256u * s;
            if (v < NV) {
                uint32_t b, r, i, j;  // out element: plane b, row i (of P), column j (of Q)  <-  t's row j, column i
                p.pq.divmod(v * W, b, r);
                p.qdiv.divmod(r, i, j);
                T xt[W], xo[W], xr[W];
#pragma unroll
                for (int kk = 0; kk < W; ++kk) {
                    xt[kk] = tile[(b * p.Q + j) * p.pitch + i];
                    xo[kk] = p.o_scalar ? oval : ov[s][kk];
                    if (++j == p.Q) { j = 0; if (++i == p.P) { i = 0; ++b; } }
                }
                if (T_IS_A) apply_n<Op, T, W>(ctx, xt, xo, xr);
                else apply_n<Op, T, W>(ctx, xo, xt, xr);
                V val;
#pragma unroll
                for (int kk = 0; kk < W; ++kk) val[kk] = xr[kk];
                store_stream_if(T, reinterpret_cast<V *>(outc) + v, val, p.nt);
            }
        }
        return;
    }
    // the last, partial chunk: element by element
    const uint32_t C = Gc * PQ;
    for (uint32_t e = threadIdx.x; e < C; e += 256) tile[lds_of_t(e)] = tc[e];
    __syncthreads();
    for (uint32_t e = threadIdx.x; e < C; e += 256) {
        uint32_t b, r, i, j;
        p.pq.divmod(e, b, r);
        p.qdiv.divmod(r, i, j);
        const T xt = tile[(b * p.Q + j) * p.pitch + i], xo = p.o_scalar ? oval : oc[e];
        outc[e] = T_IS_A ? Op::apply(xt, xo) : Op::apply(xo, xt);
    }
}

// Is the normalised problem a dense batch of small planes with one operand read transposed?
inline bool plan_planes(const Plan &pl, int esz, PlanesParams *pp, bool *t_is_a) {
    static const bool off = [] { const char *e = getenv("SMHIP_PLANES_KERNEL"); return e && *e && atoi(e) == 0; }();  // tools: SMHIP_PLANES_KERNEL=0
    if (off || pl.ndim != 3) return false;
    const int64_t B = pl.shape[0], P = pl.shape[1], Q = pl.shape[2], PQ = P * Q;
    const int W = 16 / esz, cmax = 32768 / esz;
    if (P < 2 || Q < 2 || PQ > cmax || B < 64) return false;
    if (P % 64 == 0 && (Q * esz) % 128 == 0) return false;  // whole 64-row patches with line-aligned rows: the tile kernel's best case (64 x 64: 89 % against 78 % here)
    auto turned = [&](const int64_t *st) { return st[0] == PQ && st[1] == 1 && st[2] == P; };
    auto dense = [&](const int64_t *st) { return st[0] == PQ && st[1] == Q && st[2] == 1; };
    auto scalar = [&](const int64_t *st) { return st[0] == 0 && st[1] == 0 && st[2] == 0; };
    const bool ta = turned(pl.sa) && (dense(pl.sb) || scalar(pl.sb)), tb = turned(pl.sb) && (dense(pl.sa) || scalar(pl.sa));
    if (ta == tb) return false;
    *t_is_a = ta;
    PlanesParams r{};
    r.o_scalar = scalar(ta ? pl.sb : pl.sa) ? 1 : 0;
    r.P = (uint32_t)P; r.Q = (uint32_t)Q; r.pitch = (uint32_t)P | 1u;
    r.planes = (uint64_t)B;
    uint32_t G = (uint32_t)(cmax / PQ);
    while (G > 0 && ((uint64_t)G * PQ) % W) --G;                       // whole vectors per chunk
    while (G > 0 && (uint64_t)G * Q * r.pitch > (uint64_t)cmax + cmax / 2) --G;  // the padded tile fits
    while (G > 0 && ((uint64_t)G * PQ) % W) --G;
    if (G < 1 || (r.planes + G - 1) / G >= 0x7fffffffull) return false;
    r.G = G;
    r.pq = FastDiv((uint32_t)PQ); r.pdiv = FastDiv(r.P); r.qdiv = FastDiv(r.Q);
    *pp = r;
    return true;
}

template <typename T, typename Op>
int run_planes(const PlanesParams &pp, bool t_is_a, const void *a, const void *b, void *out, hipStream_t s) {
    const T *t = static_cast<const T *>(t_is_a ? a : b), *o = static_cast<const T *>(t_is_a ? b : a);
    const dim3 grid((unsigned)((pp.planes + pp.G - 1) / pp.G)), block(256);
    if (t_is_a) hipLaunchKernelGGL((planes_kernel<T, Op, true>), grid, block, 0, s, t, o, static_cast<T *>(out), pp);
    else hipLaunchKernelGGL((planes_kernel<T, Op, false>), grid, block, 0, s, t, o, static_cast<T *>(out), pp);
    SMHIP_LAUNCH_CHECK("planes_kernel");
    return SMHIP_OK;
}

// Does the normalised plane fit the record kernel?  Fills the parameters and says which operand is the turned one.
inline bool plan_record(const Plan &pl, int esz, RecordParams *rp, bool *small_p, bool *t_is_a) {
    static const bool off = [] { const char *e = getenv("SMHIP_RECORD_KERNEL"); return e && atoi(e) == 0 && *e; }();  // tools: SMHIP_RECORD_KERNEL=0
    if (off || pl.ndim != 2) return false;
    const int64_t P = pl.shape[0], Q = pl.shape[1];
    const int W = 16 / esz, cmax = 32768 / esz;
    auto turned = [&](const int64_t *st) { return st[0] == 1 && st[1] >= P && P > 1; };              // contiguous along dim 0
    auto direct = [&](const int64_t *st) { return st[1] == 1 && (st[0] >= Q || st[0] == 0); };        // contiguous along dim 1 (or one row for every row)
    auto scalar = [&](const int64_t *st) { return st[0] == 0 && st[1] == 0; };
    const bool ta = turned(pl.sa) && (direct(pl.sb) || scalar(pl.sb)), tb = turned(pl.sb) && (direct(pl.sa) || scalar(pl.sa));
    if (ta == tb) return false;
    const int64_t *st = ta ? pl.sa : pl.sb, *so = ta ? pl.sb : pl.sa;
    *t_is_a = ta;
    RecordParams r{};
    r.o_scalar = scalar(so) ? 1 : 0;
    static const int64_t max_p = [] { const char *e = getenv("SMHIP_RECORD_MAX_P"); return e && *e ? (int64_t)atoi(e) : (int64_t)128; }();
    static const int64_t max_q = [] { const char *e = getenv("SMHIP_RECORD_MAX_Q"); return e && *e ? (int64_t)atoi(e) : (int64_t)128; }();
    if (P <= max_p && Q >= 4096 && st[1] == P) {            // few long rows out of dense records: AoS -> SoA
        *small_p = true;
        r.k = (uint32_t)P; r.n = (uint64_t)Q;
        r.o_pitch = so[0];
    } else if ((Q < 16 || (Q <= max_q && (Q * esz) % 128 != 0)) && P >= 4096 && (r.o_scalar || so[0] == Q)) {  // records out of few long rows: SoA -> AoS (whole short patch rows: the tile kernel)
        *small_p = false;
        r.k = (uint32_t)Q; r.n = (uint64_t)P;
        r.t_pitch = st[1];
    } else {
        return false;
    }
    r.R = (uint32_t)(cmax / (int64_t)r.k) / 64 * 64;
    if (r.R < 64 || r.k > 128) return false;
    // a smaller R that divides the plane's record count spares every plane its element-wise last chunk (a batch of
    // 224 x 224 x 3 images: 18.7 chunks of 2688 records -> 28 whole chunks of 1792)
    for (uint32_t m = r.R / 64; m >= (r.R / 64 + 1) / 2; --m)
        if (r.n % (64ull * m) == 0) { r.R = 64 * m; break; }
    // The flat side scatters a lane's four elements e = 4 l + kk to (record e / k, index e % k).  For k a power of two
    // from 8 up a 32-lane group touches the rows kk, kk + 4, kk + 8 ... at 32 * 4 / k consecutive records each: those
    // rows must start 256 / k banks apart (an odd pitch put rows 4 apart four banks apart: (8, n) 64 %, (16, n) 63 %).
    r.pitch = r.R + r.R / 32 + 1;
    if (r.k >= 8 && (r.k & (r.k - 1)) == 0) {
        const uint32_t want = (64 / r.k) % 16;  // pitch mod 16: 4 * pitch = 256 / k (mod 64)
        while (r.pitch % 16 != want) ++r.pitch;
    }
    r.kdiv = FastDiv(r.k);
    r.vrdiv = FastDiv(r.R / (uint32_t)W);
    *rp = r;
    return true;
}

template <typename T, typename Op>
int run_record(const RecordParams &rp, size_t planes, bool small_p, bool t_is_a, const void *a, const void *b, void *out, hipStream_t s) {
    const T *t = static_cast<const T *>(t_is_a ? a : b), *o = static_cast<const T *>(t_is_a ? b : a);
    const dim3 grid((unsigned)(planes * rp.chunks.d)), block(256);
    T *po = static_cast<T *>(out);
    if (small_p) {
        if (t_is_a) hipLaunchKernelGGL((record_kernel<T, Op, true, true>), grid, block, 0, s, t, o, po, rp);
        else hipLaunchKernelGGL((record_kernel<T, Op, true, false>), grid, block, 0, s, t, o, po, rp);
    } else {
        if (t_is_a) hipLaunchKernelGGL((record_kernel<T, Op, false, true>), grid, block, 0, s, t, o, po, rp);
        else hipLaunchKernelGGL((record_kernel<T, Op, false, false>), grid, block, 0, s, t, o, po, rp);
    }
    SMHIP_LAUNCH_CHECK("record_kernel");
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
    uint32_t nt;     // non-temporal reads: the copy reads more than the Infinity Cache holds
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
#ifdef SMHIP_COPY_PLAIN_STORES
            *reinterpret_cast<V *>(dst + offD + e0) = load_stream_if(T, reinterpret_cast<const V *>(src + offS + e0), p.nt);
#else
            store_stream_if(T, reinterpret_cast<V *>(dst + offD + e0), load_stream_if(T, reinterpret_cast<const V *>(src + offS + e0), p.nt), p.nt);
#endif
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
    // the element form's stores are plain (partial lines want the L2's write combining) and occupy the cache too; the
    // vector form (rows contiguous on both sides) streams like the row kernel: nt / sc1 stores by the launch's footprint
    p.nt = vec ? (uint32_t)stream_policy(pl.n * sizeof(T), pl.n * sizeof(T)) : (uint32_t)stream_reads(2 * pl.n * sizeof(T));
    const unsigned grid = (unsigned)((slots + 255) / 256);
    if (vec) hipLaunchKernelGGL((strided_copy_kernel<T, true>), dim3(grid), dim3(256), 0, s, src, dst, p);
    else hipLaunchKernelGGL((strided_copy_kernel<T, false>), dim3(grid), dim3(256), 0, s, src, dst, p);
    SMHIP_LAUNCH_CHECK("strided_copy_kernel");
    return SMHIP_OK;
}

int copy_plan(int dtype, const void *src, void *dst, const Plan &pl, hipStream_t s) {
    const size_t esz = dtype_size(dtype);
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
    const bool user = op >= SMHIP_OP_USER_BASE;  // a registered expression: the same kernels, compiled by hipRTC (jit.hip)
    if (pl.ndim == 1) {
        // calculate.h:10-11's fast path, decided on the normalised problem
        if (pl.sa[0] == 1 && pl.sb[0] == 1) return user ? jit_contiguous(op, dtype, a, b, out, pl.n, s) : launch_contiguous(op, dtype, a, b, out, pl.n, s);
        if (!user && pl.sa[0] == 1 && pl.sb[0] == 0) return launch_array_devscalar(op, dtype, a, b, pl.n, out, false, s);
        if (!user && pl.sa[0] == 0 && pl.sb[0] == 1) return launch_array_devscalar(op, dtype, b, a, pl.n, out, true, s);
    }
    // experiment switch (tools/bcast_pieces.py): cut broadcast problems above 2^k elements along their outermost dimension too
    static const size_t piece_elems = [] { const char *e = getenv("SMHIP_BCAST_PIECE_LOG2"); return e && atoi(e) > 0 ? (size_t)1 << atoi(e) : (size_t)0; }();
    const size_t max_elems = piece_elems && piece_elems < kMaxLaunchElems && pl.ndim >= 2 && (size_t)pl.shape[0] > 1 ? piece_elems + 1 : kMaxLaunchElems;
    if (pl.n >= max_elems) {
        const size_t kMaxLaunchElems = max_elems;  // shadows the file-scope limit inside this block
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
    // (a dense (n, S) array seen as (S, n) -- records into rows -- is ONE pass through the record kernel below instead of S
    // passes over the array here: dst (3, n) = src (n, 3).T 31.9 -> 87.3 %, tools/copy_zoo.py)
    const bool records_into_rows = pl.ndim == 2 && pl.sa[0] == 1 && pl.sa[1] == pl.shape[0] && pl.shape[1] >= 4096;
    if (op == SMHIP_OP_LEFT && pl.ndim <= 2 && pl.sa[pl.ndim - 1] >= 2 && pl.sa[pl.ndim - 1] <= 4 && pl.shape[pl.ndim - 1] >= 64 &&
        (pl.ndim == 1 || pl.sa[0] != 0) && !records_into_rows) {  // every S-th element of each row, made dense
        const size_t rows = pl.ndim == 2 ? (size_t)pl.shape[0] : 1;
        const uint32_t inner = (uint32_t)pl.shape[pl.ndim - 1];
        const int64_t pitch = pl.ndim == 2 ? pl.sa[0] : 0;
        const int stride = (int)pl.sa[pl.ndim - 1];
        switch (dtype) {
            case SMHIP_F32: case SMHIP_I32: return run_deinterleave<int32_t>(a, out, rows, inner, pitch, stride, s);  // only the width matters
            case SMHIP_F64: case SMHIP_I64: return run_deinterleave<int64_t>(a, out, rows, inner, pitch, stride, s);
        }
    }
    if (op == SMHIP_OP_LEFT && pl.ndim == 2 && pl.sa[0] == 1 && pl.sa[1] == 0 && pl.shape[1] < 16) {  // a flat repeat
        switch (dtype) {
            case SMHIP_F32: case SMHIP_I32: return run_repeat<int32_t>(a, out, (size_t)pl.shape[0], (uint32_t)pl.shape[1], s);  // only the width matters
            case SMHIP_F64: case SMHIP_I64: return run_repeat<int64_t>(a, out, (size_t)pl.shape[0], (uint32_t)pl.shape[1], s);
        }
    }
    // A dense operand against a small one that repeats along the FLAT index -- the reference tests' own pattern,
    // (N,224,224,3) op (1,224,1,3) (tests/add.cpp:59-92), or a per-channel bias (..., 3) op (3,): every leading axis the small
    // operand ignores makes its values periodic in the output's linear index, with period P = the product of the axes from
    // the first one it does not ignore.  The period (made a multiple of the vector width by repeating it) is written out
    // once -- 588 KiB for the pattern above, L2-resident -- and the problem becomes config 3's shape, rows of P against
    // one row: the flat tile kernel instead of the LDS kernel's per-vector index chains (88 vector instructions per 16
    // bytes in round 1).  Writing the period out is a launch of its own (~5 us with its dependency), so only outputs of
    // 128 MiB and more go this way: (256,224,224,3) + (1,224,1,3) 49.5-52 -> 46.5 us; at half that size it is a wash.
    static const size_t periodic_min_bytes = [] {  // SMHIP_PERIODIC_MIN_MIB: experiments (tools/periodic_threshold.py)
        const char *e = getenv("SMHIP_PERIODIC_MIN_MIB");
        return (size_t)(e ? atol(e) : 128) << 20;
    }();
    if (!user && op != SMHIP_OP_LEFT && pl.ndim >= 2 && pl.n * dtype_size(dtype) >= periodic_min_bytes) {
        const size_t esz = dtype_size(dtype);
        const int64_t W = 16 / (int64_t)esz;
        for (int role = 0; role < 2; ++role) {
            const int64_t *sx = role == 0 ? pl.sa : pl.sb, *sy = role == 0 ? pl.sb : pl.sa;
            if (role == 1 && op != SMHIP_OP_ADD && op != SMHIP_OP_MUL) break;  // the small operand on the left: commutative Ops only
            bool dense = true;
            int64_t run = 1;
            for (int d = pl.ndim - 1; d >= 0; --d) {
                dense &= sx[d] == run;
                run *= pl.shape[d];
            }
            if (!dense) continue;  // x is a view
            {   // y is constant along the trailing axes (a per-channel bias in NCHW: (B, C, H, W) + (1, C, 1, 1)): with those
                // axes merged the problem is rows against ONE VALUE PER ROW, config 3's column form -- once y's values are
                // written out as a dense vector of `rows` elements (a launch of a few microseconds).  The row kernel took
                // (64, 256, 56, 56) + (1, 256, 1, 1) at 72.6 %; tools/bcast_zoo.py.
                int j = pl.ndim - 1;
                while (j >= 0 && sy[j] == 0) --j;
                size_t cols = 1, rows = 1;
                for (int d = j + 1; d < pl.ndim; ++d) cols *= (size_t)pl.shape[d];
                for (int d = 0; d <= j; ++d) rows *= (size_t)pl.shape[d];
                bool y_dense = true;  // already a dense vector of rows values: the flat route below takes it as it is
                int64_t run_y = 1;
                for (int d = j; d >= 0; --d) { y_dense &= sy[d] == run_y; run_y *= pl.shape[d]; }
                if (j >= 0 && j < pl.ndim - 1 && !y_dense && cols >= (size_t)SMHIP_FLAT_ROWS_MIN_COLS && cols % (size_t)W == 0 && rows >= 4 &&
                    rows * esz <= ((size_t)2 << 20) && cols < 0x7fffffffull && rows < 0x7fffffffull) {
                    ScratchLease lease;
                    double *tmp8;
                    if (int rc = lease.take((rows * esz + 7) / 8, &tmp8)) return rc;
                    int64_t eshape[SMHIP_MAX_NDIM], esy[SMHIP_MAX_NDIM], zeros[SMHIP_MAX_NDIM];
                    for (int d = 0; d <= j; ++d) { eshape[d] = pl.shape[d]; esy[d] = sy[d]; zeros[d] = 0; }
                    const void *x = role == 0 ? a : b, *y = role == 0 ? b : a;
                    const Plan sub = normalise(eshape, esy, zeros, j + 1);
                    if (int rc = launch_plan(SMHIP_OP_LEFT, dtype, y, y, tmp8, sub, s)) return rc;
                    return launch_flat_rows(op, dtype, x, tmp8, out, rows, cols, false, s);
                }
            }
            int k = 0;
            while (k < pl.ndim && sy[k] == 0) ++k;
            if (k == 0 || k == pl.ndim) continue;  // y ignores no leading axis, or y is a single value
            size_t period = 1;
            for (int d = k; d < pl.ndim; ++d) period *= (size_t)pl.shape[d];
            size_t rep = 1;
            while ((period * rep) % (size_t)W) ++rep;  // at most W copies make the period a whole number of vectors
            const size_t cols = period * rep, rows = pl.n / cols;
            const size_t tail_periods = (pl.n % cols) / period;  // whole periods behind the last whole row (< rep of them): a launch of their own
            if (rows < 4 || cols * esz > ((size_t)2 << 20)) continue;
            if (k == pl.ndim - 1 && rep == 1) continue;  // already rows against one row: the flat route below takes it as it is
            ScratchLease lease;
            double *tmp8;
            if (int rc = lease.take((cols * esz + 7) / 8, &tmp8)) return rc;
            // the period, written out: (rep, shape[k..]) with y's strides (0, sy[k..]) through the LEFT Op (out = a)
            int64_t eshape[SMHIP_MAX_NDIM + 1], esy[SMHIP_MAX_NDIM + 1], zeros[SMHIP_MAX_NDIM + 1];
            int en = 0;
            eshape[en] = (int64_t)rep; esy[en] = 0; zeros[en] = 0; ++en;
            for (int d = k; d < pl.ndim; ++d) { eshape[en] = pl.shape[d]; esy[en] = sy[d]; zeros[en] = 0; ++en; }
            const void *x = role == 0 ? a : b, *y = role == 0 ? b : a;
            const Plan sub = normalise(eshape, esy, zeros, en);
            if (int rc = launch_plan(SMHIP_OP_LEFT, dtype, y, y, tmp8, sub, s)) return rc;
            if (int rc = launch_flat_rows(op, dtype, x, tmp8, out, rows, cols, true, s)) return rc;
            if (tail_periods) {
                // (rows of 31 against one row: four periods make whole vectors, and 2^26 / 31 rows are not a multiple of four --
                // until round 3 that sent the whole problem to the row kernel: 40 % instead of 86 %)
                const size_t done = rows * cols;
                eshape[0] = (int64_t)tail_periods;
                int64_t esx[SMHIP_MAX_NDIM + 1];
                int64_t run_x = 1;
                for (int d = en - 1; d >= 1; --d) { esx[d] = run_x; run_x *= eshape[d]; }
                esx[0] = run_x;
                const char *xt = static_cast<const char *>(x) + done * esz;
                char *ot = static_cast<char *>(out) + done * esz;
                const Plan tailp = role == 0 ? normalise(eshape, esx, esy, en) : normalise(eshape, esy, esx, en);
                return role == 0 ? launch_plan(op, dtype, xt, y, ot, tailp, s) : launch_plan(op, dtype, y, xt, ot, tailp, s);
            }
            return SMHIP_OK;
        }
    }
    const bool heavy = op == SMHIP_OP_POW && (dtype == SMHIP_F32 || dtype == SMHIP_F64);  // as launch_aot's kRows: float / double pow only
    if (!user && op != SMHIP_OP_LEFT && pl.ndim == 2 && pl.sa[0] == pl.shape[1] && pl.sa[1] == 1 &&
        pl.shape[1] % (16 / (int64_t)dtype_size(dtype)) == 0 &&
        ((pl.sb[0] == 0 && pl.sb[1] == 1) || (pl.sb[0] == 1 && pl.sb[1] == 0 && pl.shape[1] >= SMHIP_FLAT_ROWS_MIN_COLS))) {
        // (one COLUMN against rows of fewer than 16 elements stays with the short-rows kernel: 82 % there, 74-78 % here;
        // one ROW of 4 / 8 / 12 elements is faster here than through the LDS kernel: 91 against 88 %)
        // config 3's shape: a dense array against one row / one column takes the flat tile kernel -- output vector i pairs
        // a[i] with b[i mod cols/W] (resp. b[i / (cols/W)]): one fast division per vector instead of the row kernel's
        // per-row index arithmetic.  4096 x 4096 f32: multiply 19.9 -> 19.2 us (84.3 -> 87.4 %), pow 23 -> 20.2 us.
        return launch_flat_rows(op, dtype, a, b, out, (size_t)pl.shape[0], (size_t)pl.shape[1], pl.sb[0] == 0, s);
    }
    if (!user && (op == SMHIP_OP_ADD || op == SMHIP_OP_MUL) && pl.ndim == 2 && pl.sb[0] == pl.shape[1] && pl.sb[1] == 1 &&
        pl.shape[1] % (16 / (int64_t)dtype_size(dtype)) == 0 &&
        ((pl.sa[0] == 0 && pl.sa[1] == 1) || (pl.sa[0] == 1 && pl.sa[1] == 0 && pl.shape[1] >= SMHIP_FLAT_ROWS_MIN_COLS))) {
        // the same shape with the roles exchanged (row + A, column * A): + and * commute bit for bit
        return launch_flat_rows(op, dtype, b, a, out, (size_t)pl.shape[0], (size_t)pl.shape[1], pl.sa[0] == 0, s);
    }
    if (!user && pl.ndim == 3) {  // a dense batch of small planes, one operand read transposed
        PlanesParams pp;
        bool t_is_a;
        if (plan_planes(pl, (int)dtype_size(dtype), &pp, &t_is_a)) {
            const size_t esz = dtype_size(dtype), bytes = pl.n * esz;
            const bool sa_scalar = !(pl.sa[0] || pl.sa[1] || pl.sa[2]), sb_scalar = !(pl.sb[0] || pl.sb[1] || pl.sb[2]);
            pp.nt = (uint32_t)stream_policy({Span{a, sa_scalar ? esz : bytes}, Span{b, sb_scalar ? esz : bytes}}, Span{out, bytes});
#define SMHIP_PLANES(T)                                                                    \
    switch (op) {                                                                          \
        case SMHIP_OP_ADD: return run_planes<T, AddOp<T>>(pp, t_is_a, a, b, out, s);      \
        case SMHIP_OP_SUB: return run_planes<T, SubtractOp<T>>(pp, t_is_a, a, b, out, s); \
        case SMHIP_OP_MUL: return run_planes<T, MultiplyOp<T>>(pp, t_is_a, a, b, out, s); \
        case SMHIP_OP_DIV: return run_planes<T, DivideOp<T>>(pp, t_is_a, a, b, out, s);   \
        case SMHIP_OP_POW: return run_planes<T, PowOp<T>>(pp, t_is_a, a, b, out, s);      \
        case SMHIP_OP_LEFT: return run_planes<T, LeftOp<T>>(pp, t_is_a, a, b, out, s);    \
    }                                                                                      \
    break;
            switch (dtype) {
                case SMHIP_F32: SMHIP_PLANES(float)
                case SMHIP_F64: SMHIP_PLANES(double)
                case SMHIP_I32: SMHIP_PLANES(int32_t)
                case SMHIP_I64: SMHIP_PLANES(int64_t)
            }
#undef SMHIP_PLANES
        }
    }
    if (!user && (pl.ndim == 2 || pl.ndim == 3)) {
        // a plane with one tiny extent and one turned operand, or a batch of such planes (an outermost axis in front of
        // them): the record kernel (AoS <-> SoA)
        const size_t esz = dtype_size(dtype);
        Plan pp = pl;
        int64_t batch = 1, ba = 0, bb = 0;
        if (pl.ndim == 3) {
            batch = pl.shape[0]; ba = pl.sa[0]; bb = pl.sb[0];
            pp.ndim = 2;
            for (int d = 0; d < 2; ++d) { pp.shape[d] = pl.shape[d + 1]; pp.sa[d] = pl.sa[d + 1]; pp.sb[d] = pl.sb[d + 1]; }
            pp.n = (size_t)(pp.shape[0] * pp.shape[1]);
        }
        RecordParams rp;
        bool small_p, t_is_a;
        if (plan_record(pp, (int)esz, &rp, &small_p, &t_is_a)) {
            const uint64_t chunks = (rp.n + rp.R - 1) / rp.R;
            if (chunks * (uint64_t)batch < 0x7fffffffull) {
                rp.chunks = FastDiv((uint32_t)chunks);
                rp.t_batch = t_is_a ? ba : bb;
                rp.o_batch = t_is_a ? bb : ba;
                auto span = [&](const void *ptr, const int64_t *st) {
                    int64_t last = 0;
                    for (int d = 0; d < pl.ndim; ++d) last += (pl.shape[d] - 1) * st[d];
                    return Span{ptr, (size_t)(last + 1) * esz};
                };
                rp.nt = (uint32_t)stream_policy({span(a, pl.sa), span(b, pl.sb)}, Span{out, pl.n * esz});
#define SMHIP_RECORD(T)                                                                                              \
    switch (op) {                                                                                                    \
        case SMHIP_OP_ADD: return run_record<T, AddOp<T>>(rp, (size_t)batch, small_p, t_is_a, a, b, out, s);        \
        case SMHIP_OP_SUB: return run_record<T, SubtractOp<T>>(rp, (size_t)batch, small_p, t_is_a, a, b, out, s);   \
        case SMHIP_OP_MUL: return run_record<T, MultiplyOp<T>>(rp, (size_t)batch, small_p, t_is_a, a, b, out, s);   \
        case SMHIP_OP_DIV: return run_record<T, DivideOp<T>>(rp, (size_t)batch, small_p, t_is_a, a, b, out, s);     \
        case SMHIP_OP_POW: return run_record<T, PowOp<T>>(rp, (size_t)batch, small_p, t_is_a, a, b, out, s);        \
        case SMHIP_OP_LEFT: return run_record<T, LeftOp<T>>(rp, (size_t)batch, small_p, t_is_a, a, b, out, s);      \
    }                                                                                                                \
    break;
                switch (dtype) {
                    case SMHIP_F32: SMHIP_RECORD(float)
                    case SMHIP_F64: SMHIP_RECORD(double)
                    case SMHIP_I32: SMHIP_RECORD(int32_t)
                    case SMHIP_I64: SMHIP_RECORD(int64_t)
                }
#undef SMHIP_RECORD
            }
        }
    }
    Launch L;
    if (int rc = plan_launch(pl, (int)dtype_size(dtype), heavy, &L)) return rc;
    {   // the plan's policy word came from sizes alone; now the operands are known: cold ones are read non-temporally
        // (internal.h: refine_policy), and the launch's touches go on record
        const size_t esz = dtype_size(dtype);
        auto span = [&](const void *p, const int64_t *st) {
            int64_t last = 0;
            for (int d = 0; d < pl.ndim; ++d) last += (pl.shape[d] - 1) * st[d];
            return Span{p, (size_t)(last + 1) * esz};
        };
        uint32_t *word = L.kind == Launch::kRow ? &L.p.row.nt : L.kind == Launch::kLds ? &L.p.lds.nt : L.kind == Launch::kTile ? &L.p.tile.nt
                         : L.kind == Launch::kStrided ? &L.p.strided.nt : nullptr;
        const int refined = refine_policy(word ? (int)*word : 0, {span(a, pl.sa), span(b, pl.sb)}, Span{out, pl.n * esz});
        if (word) *word = (uint32_t)refined;
        if (L.kind == Launch::kTile) {  // an output that overlaps an operand: every element is computed exactly once (bcast_kernels.hip.h: tile_body)
            auto meets = [&](const Span &x) {
                const char *lo = static_cast<const char *>(x.p), *o = static_cast<const char *>(out);
                return lo < o + pl.n * esz && o < lo + x.bytes;
            };
            L.p.tile.in_place = meets(span(a, pl.sa)) || meets(span(b, pl.sb));
        }
    }
    if (user) return jit_launch(op, dtype, L, a, b, out, s);
    // rows of 2..15 elements against one value per row: x dense (r, 1), y a dense vector (1, 0)
    // ... and rows of any length that is not a whole number of vectors (17, 31, 63 elements): the row kernel handled those at
    // 46-66 % (tools/short_inner.py); the flat tile kernel's column form takes the whole-vector lengths
    const bool odd_rows = L.kind == Launch::kRow && pl.ndim == 2 && pl.shape[1] >= 16 && pl.shape[1] <= 1024 &&
                          pl.shape[1] % (16 / (int64_t)dtype_size(dtype)) != 0 && pl.n < 0x7fffffffull;
    if ((odd_rows || (L.kind == Launch::kGather && pl.shape[1] >= 2 && pl.shape[1] < 16)) && pl.ndim == 2 && op != SMHIP_OP_LEFT) {
        const bool y_is_b = pl.sa[0] == pl.shape[1] && pl.sa[1] == 1 && pl.sb[0] == 1 && pl.sb[1] == 0;
        const bool y_is_a = pl.sb[0] == pl.shape[1] && pl.sb[1] == 1 && pl.sa[0] == 1 && pl.sa[1] == 0;
        if (y_is_b || y_is_a) {
            const void *x = y_is_b ? a : b, *y = y_is_b ? b : a;
            const size_t rows = (size_t)pl.shape[0];
            const uint32_t r = (uint32_t)pl.shape[1];
#define SMHIP_SHORT_ROWS(T)                                                                                      \
    switch (op) {                                                                                                \
        case SMHIP_OP_ADD: return run_short_rows<T, AddOp<T>>(x, y, out, rows, r, y_is_a, s);                  \
        case SMHIP_OP_SUB: return run_short_rows<T, SubtractOp<T>>(x, y, out, rows, r, y_is_a, s);             \
        case SMHIP_OP_MUL: return run_short_rows<T, MultiplyOp<T>>(x, y, out, rows, r, y_is_a, s);             \
        case SMHIP_OP_DIV: return run_short_rows<T, DivideOp<T>>(x, y, out, rows, r, y_is_a, s);               \
        case SMHIP_OP_POW: return run_short_rows<T, PowOp<T>>(x, y, out, rows, r, y_is_a, s);                  \
    }                                                                                                            \
    break;
            switch (dtype) {
                case SMHIP_F32: SMHIP_SHORT_ROWS(float)
                case SMHIP_F64: SMHIP_SHORT_ROWS(double)
                case SMHIP_I32: SMHIP_SHORT_ROWS(int32_t)
                case SMHIP_I64: SMHIP_SHORT_ROWS(int64_t)
            }
#undef SMHIP_SHORT_ROWS
        }
    }
    // short rows of a view against a dense partner or another view (both stepping by one along the row)
    static const int64_t pitched_max = [] { const char *e = getenv("SMHIP_PITCHED_ROWS_MAX"); return e && *e ? (int64_t)atoi(e) : (int64_t)4096; }();
    auto rowwise = [&](const int64_t *st) { return st[1] == 1 && st[0] >= pl.shape[1]; };        // steps by one along the row
    auto onevalue = [&](const int64_t *st) { return st[0] == 0 && st[1] == 0; };
    auto pitched = [&](const int64_t *st) { return rowwise(st) && st[0] != pl.shape[1]; };        // ... and is a view
    if ((L.kind == Launch::kRow || L.kind == Launch::kGather) && pl.ndim == 2 && pl.shape[1] >= 2 && pl.shape[1] < pitched_max &&
        pl.n < 0x7fffffffull && (pitched(pl.sa) || pitched(pl.sb)) && (rowwise(pl.sa) || onevalue(pl.sa)) && (rowwise(pl.sb) || onevalue(pl.sb))) {
        const size_t rows = (size_t)pl.shape[0];
        const uint32_t c = (uint32_t)pl.shape[1];
#define SMHIP_PITCHED(T)                                                                                         \
    switch (op) {                                                                                                \
        case SMHIP_OP_ADD: return run_pitched_rows<T, AddOp<T>>(a, b, out, rows, c, pl.sa[0], pl.sb[0], s);     \
        case SMHIP_OP_SUB: return run_pitched_rows<T, SubtractOp<T>>(a, b, out, rows, c, pl.sa[0], pl.sb[0], s); \
        case SMHIP_OP_MUL: return run_pitched_rows<T, MultiplyOp<T>>(a, b, out, rows, c, pl.sa[0], pl.sb[0], s); \
        case SMHIP_OP_DIV: return run_pitched_rows<T, DivideOp<T>>(a, b, out, rows, c, pl.sa[0], pl.sb[0], s);  \
        case SMHIP_OP_POW: return run_pitched_rows<T, PowOp<T>>(a, b, out, rows, c, pl.sa[0], pl.sb[0], s);     \
        case SMHIP_OP_LEFT: return run_pitched_rows<T, LeftOp<T>>(a, b, out, rows, c, pl.sa[0], pl.sb[0], s);   \
    }                                                                                                            \
    break;
        switch (dtype) {
            case SMHIP_F32: SMHIP_PITCHED(float)
            case SMHIP_F64: SMHIP_PITCHED(double)
            case SMHIP_I32: SMHIP_PITCHED(int32_t)
            case SMHIP_I64: SMHIP_PITCHED(int64_t)
        }
#undef SMHIP_PITCHED
    }
#define SMHIP_DISPATCH_OP(T)                                                                   \
    switch (op) {                                                                              \
        case SMHIP_OP_ADD: return launch_aot<T, AddOp<T>>(L, a, b, out, s);                \
        case SMHIP_OP_SUB: return launch_aot<T, SubtractOp<T>>(L, a, b, out, s);           \
        case SMHIP_OP_MUL: return launch_aot<T, MultiplyOp<T>>(L, a, b, out, s);           \
        case SMHIP_OP_DIV: return launch_aot<T, DivideOp<T>>(L, a, b, out, s);             \
        case SMHIP_OP_POW: return launch_aot<T, PowOp<T>>(L, a, b, out, s);                \
        case SMHIP_OP_LEFT: return launch_aot<T, LeftOp<T>>(L, a, b, out, s);                \
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
    // A destination that is dense in SOME axis order (a whole array, or a transposed / permuted view of one): walk the
    // axes in the destination's order and the copy is "dense out = strided view", i.e. SMHIP_OP_LEFT through the
    // broadcast kernels -- contiguous stream, row kernel, or the LDS tile kernel when the source is the turned side
    // (tools/misc_rates.py: 80 % of peak where one element per lane reached 17-21 %; plain dense copies run through
    // the array kernel at 81 % where hipMemcpyAsync device-to-device gave 67 %).
    int order[SMHIP_MAX_NDIM];
    for (int d = 0; d < pl.ndim; ++d) order[d] = d;
    for (int i = 1; i < pl.ndim; ++i)  // insertion sort, destination stride descending
        for (int j = i; j > 0 && pl.sb[order[j]] > pl.sb[order[j - 1]]; --j) { const int t = order[j]; order[j] = order[j - 1]; order[j - 1] = t; }
    bool dst_dense = true;
    int64_t expect = 1;
    for (int k = pl.ndim - 1; k >= 0; --k) {
        if (pl.sb[order[k]] != expect) { dst_dense = false; break; }
        expect *= pl.shape[order[k]];
    }
    if (dst_dense) {
        int64_t pshape[SMHIP_MAX_NDIM], psrc[SMHIP_MAX_NDIM], zeros[SMHIP_MAX_NDIM];
        for (int k = 0; k < pl.ndim; ++k) { pshape[k] = pl.shape[order[k]]; psrc[k] = pl.sa[order[k]]; zeros[k] = 0; }
        return launch_plan(SMHIP_OP_LEFT, dtype, src, src, dst, normalise(pshape, psrc, zeros, pl.ndim), s);
    }
    return copy_plan(dtype, src, dst, pl, s);
}

int launch_broadcast(int op, int dtype, const void *a, const int64_t *sa, const void *b, const int64_t *sb,
                     const int64_t *shape, int ndim, void *out, hipStream_t s) {
    const Plan pl = normalise(shape, sa, sb, ndim);
    if (pl.n == 0) return SMHIP_OK;
    return launch_plan(op, dtype, a, b, out, pl, s);
}

}  // namespace smhip
