// contiguous.hip -- the dense streaming kernels.
//
// Replaces handle_contiguous_arrays<T,Op> (reference include/math/calculate.h:
// 101-134, an 8-wide AVX2 loop on ONE host thread) and array_scalar_op<T,Op>
// (calculate.h:137-169) with HBM-streaming gfx950 kernels.
//
// Roofline: HBM-bound.  Algorithmic traffic 3 * sizeof(T) B/elem for a op b
// (2 reads + 1 write; 12 B/elem for f32), 2 * sizeof(T) for a op scalar.
// Shape of the launch (profiles/r01_sweep_stream_add.txt, N = 2^28 f32 add):
//   - one 16-byte vector per lane (float4 / double2 / int4 / long2): a wave
//     moves 1 KiB per memory instruction, fully coalesced;
//   - NO grid-stride loop: one vector per thread, >= 65k workgroups, so the
//     hardware dispatcher (not a software loop) keeps all 256 CUs / 8 XCDs
//     fed and every wave retires after 2 loads + 1 store.  Persistent
//     grid-stride variants with 2-8 vectors in flight per lane measured
//     5-12 % slower on this 2R+1W stream;
//   - non-temporal loads and stores (each byte is touched once): +8 %.
// Blocks are dealt round-robin over the 8 XCDs, consecutive blocks touch
// consecutive 16 KiB spans, so every XCD streams from all HBM stacks at once;
// there is no reuse for an XCD-affine mapping to exploit.
#include <type_traits>

#include "internal.h"
#include "ops.hip.h"

namespace smhip {
namespace {

using namespace dev;

static_assert(kLoadNt == kPolicyLoadNt && kStoreKeep == kPolicyStoreKeep, "device and host disagree on the stream-policy bits");

constexpr int kBlockBig = 1024;   // n_vec >= kBigThreshold
constexpr int kBlockSmall = 256;
constexpr size_t kBigThreshold = (size_t)1 << 20;

// out[i] = a[i] op b[i], vector body + <= width-1 scalar tail elements done
// by the first thread past the body.
// KEEP_STORES: the write side's policy (ops.hip.h) at compile time -- a run-time branch in front of this kernel's one
// store cost the N = 2^28 add 1.6 % (498 -> 507 us, same box, tools/op_matrix.py); the 1R+1W scalar kernels below do not
// notice theirs.
template <typename T, typename Op, int BLOCK, bool KEEP_STORES>
__global__ __launch_bounds__(BLOCK) void contiguous_vec_kernel(const T *__restrict__ a, const T *__restrict__ b,
                                                               T *__restrict__ out, size_t n_vec, int tail, int nt) {
    typedef typename VecTraits<T>::vec_t V;
    constexpr int W = VecTraits<T>::width;
    OpCtx<Op> ctx;
    ctx.init();
    const size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i < n_vec) {
        const V va = load_stream_if(T, reinterpret_cast<const V *>(a) + i, nt);
        const V vb = load_stream_if(T, reinterpret_cast<const V *>(b) + i, nt);
        store_stream_as(T, reinterpret_cast<V *>(out) + i, (apply_vec<Op, T>(ctx, va, vb)), !KEEP_STORES);
    } else if (i == n_vec) {
        for (int k = 0; k < tail; ++k) out[n_vec * W + k] = Op::apply(a[n_vec * W + k], b[n_vec * W + k]);
    }
}

// out[i] = a[i] op s  (SWAPPED: s op a[i]); s is a kernel argument, i.e. it
// lives in SGPRs -- the `set1` of calculate.h:141-146 costs nothing here.
template <typename T, typename Op, int BLOCK, bool SWAPPED>
__global__ __launch_bounds__(BLOCK) void scalar_vec_kernel(const T *__restrict__ a, T s, T *__restrict__ out,
                                                           size_t n_vec, int tail, int nt) {
    typedef typename VecTraits<T>::vec_t V;
    constexpr int W = VecTraits<T>::width;
    OpCtx<Op> ctx;
    ctx.init();
    const size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i < n_vec) {
        const V va = load_stream_if(T, reinterpret_cast<const V *>(a) + i, nt);
        store_stream_if(T, reinterpret_cast<V *>(out) + i, (apply_vec_scalar<Op, T, SWAPPED>(ctx, va, s)), nt);
    } else if (i == n_vec) {
        for (int k = 0; k < tail; ++k) {
            const T x = a[n_vec * W + k];
            out[n_vec * W + k] = SWAPPED ? Op::apply(s, x) : Op::apply(x, s);
        }
    }
}

// Same, with the scalar fetched from device memory (a fully broadcast array
// operand, e.g. `A + one_element_array`): a wave-uniform load.
template <typename T, typename Op, int BLOCK, bool SWAPPED>
__global__ __launch_bounds__(BLOCK) void devscalar_vec_kernel(const T *__restrict__ a, const T *__restrict__ sp,
                                                              T *__restrict__ out, size_t n_vec, int tail, int nt) {
    typedef typename VecTraits<T>::vec_t V;
    constexpr int W = VecTraits<T>::width;
    OpCtx<Op> ctx;
    ctx.init();
    const T s = *sp;
    const size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i < n_vec) {
        const V va = load_stream_if(T, reinterpret_cast<const V *>(a) + i, nt);
        store_stream_if(T, reinterpret_cast<V *>(out) + i, (apply_vec_scalar<Op, T, SWAPPED>(ctx, va, s)), nt);
    } else if (i == n_vec) {
        for (int k = 0; k < tail; ++k) {
            const T x = a[n_vec * W + k];
            out[n_vec * W + k] = SWAPPED ? Op::apply(s, x) : Op::apply(x, s);
        }
    }
}

// Arithmetic-heavy Ops (float / double pow): with ONE vector per lane the streaming kernels above give each wave one
// load, a long stretch of VALU work, one store -- memory and VALU time add up instead of overlapping (97-99 us for
// config 4).  Round 1 answered with a persistent, software-pipelined kernel (91-94 us); its prefetch never overlapped
// anything within a wave, though: `current = next` at the end of the loop makes the compiler put s_waitcnt vmcnt(0)
// right behind the prefetch.  A two-register-set pipeline that really prefetches (vmcnt(1)/(2)) reaches 87.9 us --
// and the one-shot form below, with TWO vectors per lane, 84.3 us (tools/sweep_pow2.hip, profiles/r02_sweep_pow2.txt).
// KIND 0: a[i] op b[i];  1: a[i] op s;  2: s op a[i].
// The one-shot form for arithmetic-heavy Ops whose arithmetic is light enough to hide: U vectors per lane, all U (x2
// operands) loads issued before the first use, then U evaluations and stores; one tile of BLOCK * U vectors per
// workgroup, no loop.  With round 2's f32 pow core (27 instructions per element instead of 45) this beats every
// persistent shape: 84.3 us for config 4 against 87.9 us for the best two-register-set pipeline and 82.4 us for a plain
// copy of the same bytes (tools/sweep_pow2.hip, profiles/r02_sweep_pow2.txt); one vector per lane is 97-99 us -- the
// second load in flight is what covers the arithmetic.  Full tiles are guard-free (per-vector guards make the compiler
// wait for each load in turn); the last, partial tile and the n % W scalar tail belong to the last workgroup.
#ifndef SMHIP_FLAT_TILE_THREADS
#define SMHIP_FLAT_TILE_THREADS 256
#endif
constexpr int kTileBlock = SMHIP_FLAT_TILE_THREADS;  // (512: tools/build_variant.sh; measured for config 3's cold leg on two queues)
// KEEP_STORES: the write side's policy (ops.hip.h: store_stream_if) as a template parameter -- as a run-time branch in
// front of each store it cut the arithmetic of the tile's vectors apart (config 4: 73.2 -> 79.0 us).
// KIND 0: a op b, 1: a op s, 2: s op a (the heavy Ops only); 3 / 4: a dense (rows x cols) against ONE ROW / ONE COLUMN of
// b -- config 3's shape, for EVERY built-in Op (launch_flat_rows).  cv = cols / W; b's vector for output vector i is b[i mod cv] (read through the caches: every workgroup
// wants the same few KiB), resp. the single element b[i / cv].  Through the row kernel this shape paid ~35 vector
// instructions per wave of index arithmetic on top of pow's 257 and staged the tables before its loads: 23.9 us at
// 4096 x 4096 against 20.2 us here (profiles/r02_pow_shapes.txt; 21.1-21.8 us under the profiler, r02_pmc_sq_pow_shapes.txt).
// BLOCK: 256 threads, except for the bank-private double pow (ops.hip.h: PowBanked), whose 80 KiB of tables want 1024.
template <typename T, typename Op, int KIND, int U, bool KEEP_STORES, int BLOCK = kTileBlock>
__global__ __launch_bounds__(BLOCK) void flat_tile_kernel(const T *__restrict__ a, const T *__restrict__ b, T s,
                                                           T *__restrict__ out, size_t n_vec, int tail, int nt, FastDiv cv) {
    constexpr int kTileBlock = BLOCK;  // shadows the file-scope default inside this kernel
    typedef typename VecTraits<T>::vec_t V;
    constexpr int W = VecTraits<T>::width;
    OpCtx<Op> ctx;
    const V *av = reinterpret_cast<const V *>(a), *bv = reinterpret_cast<const V *>(b);
    V *ov = reinterpret_cast<V *>(out);
    // Which tile is this workgroup's?  Its own index -- except for a FEW LONG rows against one row (KIND 3 with `tail` =
    // rows | log2(group) << 16; the kind has no scalar tail): walking the array flat, row r re-reads the broadcast row long
    // after row r - 1 has pushed it out of the L2 ((8, 2^23) + (1, 2^23): the 32 MiB row came back from the Infinity Cache
    // eight times, 65 % of algorithmic).  There 2^lg consecutive workgroups take neighbouring column tiles of one row, the
    // next 2^lg the same columns of the next row: with lg = 10 (4 MiB of a row per visit) the slice of the broadcast row
    // stays in the L2s across the rows -- 65 -> 81 % at (8, 2^23), 68 -> 79 % at (4, 2^24), 71 -> 89 % at (3, 3 * 2^22);
    // groups of 8 ... 256 workgroups scatter the dense operand's stream over the rows instead and gain little or lose
    // (tools/rows_walk.py, profiles/r03_rows_walk.txt).
    size_t tile = blockIdx.x;
    if (KIND == 3 && tail > 0) {
        const uint32_t tpr = cv.d / (uint32_t)(kTileBlock * U), rows = (uint32_t)tail & 0xffffu, lg = (uint32_t)tail >> 16;  // whole tiles per row, a multiple of the group
        if (blockIdx.x < tpr * rows) {
            const uint32_t x = blockIdx.x & ((1u << lg) - 1u), g = blockIdx.x >> lg, r = g % rows, cg = g / rows;
            tile = (size_t)r * tpr + (cg << lg) + x;
        }
        tail = 0;
    }
    const size_t base = tile * (kTileBlock * U) + threadIdx.x;
    auto eval = [&](const V &xa, const V &xb) {
        if constexpr (KIND == 0 || KIND >= 3) return apply_vec<Op, T>(ctx, xa, xb);
        else return apply_vec_scalar<Op, T, KIND == 2>(ctx, xa, s);
    };
    // the broadcast operand's vector for output vector i (KIND 3 / 4; launches of those kinds stay below 2^32 vectors)
    auto bcast_row = [&](size_t i) {
        uint32_t q, r;
        cv.divmod((uint32_t)i, q, r);
        return bv[r];
    };
    auto bcast_col = [&](size_t i) { return b[cv.div((uint32_t)i)]; };
    auto splat = [](T y) {
        V v;
#pragma unroll
        for (int k = 0; k < W; ++k) v[k] = y;
        return v;
    };
    // The Op's tables (a round trip to the L2, LDS writes and a barrier: OpCtx) are fetched first and committed after a
    // full tile's own loads have gone out, so the two latencies overlap instead of adding up -- this one-shot form
    // would otherwise pay the staging once per 256 x U vectors.
    typename OpCtx<Op>::template Stage<kTileBlock> staged;
    ctx.template fetch<kTileBlock>(staged);
    const bool full = (tile + 1) * (kTileBlock * U) <= n_vec;  // uniform over the workgroup
    V va[U], vb[U];
    T ys[U];  // KIND 4: the rows' exponents, splat only after the tables are committed
    if (full) {
        // the commit sits inside each arm of the read-policy branch, in straight-line code behind the tile's loads, so
        // that the wait in front of its LDS writes is a counted one (vmcnt = the tile's loads still in flight); past a
        // join the compiler falls back to vmcnt(0) and the staging would wait for the tile as well
        if (nt & kLoadNt) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                va[u] = load_stream_as(T, av + base + (size_t)u * kTileBlock, true);
                if constexpr (KIND == 0) vb[u] = load_stream_as(T, bv + base + (size_t)u * kTileBlock, true);
                if constexpr (KIND == 3) vb[u] = bcast_row(base + (size_t)u * kTileBlock);
                if constexpr (KIND == 4) ys[u] = bcast_col(base + (size_t)u * kTileBlock);
            }
            ctx.template commit<kTileBlock>(staged);
        } else {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                va[u] = load_stream_as(T, av + base + (size_t)u * kTileBlock, false);
                if constexpr (KIND == 0) vb[u] = load_stream_as(T, bv + base + (size_t)u * kTileBlock, false);
                if constexpr (KIND == 3) vb[u] = bcast_row(base + (size_t)u * kTileBlock);
                if constexpr (KIND == 4) ys[u] = bcast_col(base + (size_t)u * kTileBlock);
            }
            ctx.template commit<kTileBlock>(staged);
        }
        if constexpr (KIND == 4) {
            // the per-row exponents become visible to the optimiser only here: otherwise it hoists the exponent-only part of
            // pow above the commit's barrier and waits for every load of the tile in front of it (vmcnt(0) instead of a count)
#pragma unroll
            for (int u = 0; u < U; ++u) {
                asm volatile("" : "+v"(ys[u]));
                vb[u] = splat(ys[u]);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) store_stream_as(T, ov + base + (size_t)u * kTileBlock, eval(va[u], (KIND == 0 || KIND >= 3) ? vb[u] : va[u]), !KEEP_STORES);
        return;
    }
    ctx.template commit<kTileBlock>(staged);
    for (int u = 0; u < U; ++u) {
        const size_t i = base + (size_t)u * kTileBlock;
        if (i < n_vec) {
            const V va = load_stream(av + i);
            V vb = va;
            if constexpr (KIND == 0) vb = load_stream(bv + i);
            if constexpr (KIND == 3) vb = bcast_row(i);
            if constexpr (KIND == 4) vb = splat(bcast_col(i));
            store_stream(ov + i, eval(va, vb));
        }
    }
    if (threadIdx.x == 0) {
        for (int k = 0; k < tail; ++k) {
            const T x = a[n_vec * W + k];
            const T y = KIND == 0 ? b[n_vec * W + k] : s;  // KIND 3 / 4 have no tail: cols is a multiple of W
            out[n_vec * W + k] = KIND == 2 ? Op::apply(y, x) : Op::apply(x, y);
        }
    }
}

// Measured and NOT adopted (round 3): -DSMHIP_POW64_BANKED=1 builds it for experiments.  N = 2^26, random
// bases, scalar exponent: round 2's kernel 197.5-198.2 us (68 %; 172 us when every lane reads the same entry); bank-private
// replicas as a one-shot launch 245.9 us (LDS bank conflicts 43 -> 12.7 % of the LDS-active cycles, but 80 KiB staged per
// tile); persistent 213.7 us with two vectors per lane, 210.1 with three -- and 207 us even when every lane reads the same
// entry: what the 1024-thread persistent form costs exceeds what the conflicts cost (profiles/r03_pow64_rate.txt).
#ifndef SMHIP_POW64_BANKED
#define SMHIP_POW64_BANKED 0
#endif
#ifndef SMHIP_POW64_BANKED_U
#define SMHIP_POW64_BANKED_U 2
#endif
#if SMHIP_POW64_BANKED
// double pow with bank-private tables (ops.hip.h: PowBanked), PERSISTENT: two 1024-thread workgroups per CU stage the 80 KiB
// of replicas once and walk the tiles grid-stride.  As a one-shot launch (one tile per workgroup) the staging alone moved
// 874 MB through the L2 for a 1 GiB array and the kernel LOST to round 2's (245.9 against 197.5 us, N = 2^26,
// profiles/r03_pow64_rate.txt); arithmetic-bound as this Op is, eight resident waves per SIMD cover each other's loads
// without a software pipeline.  KIND 0: a op b, 1: a op s, 2: s op a.
template <int KIND, int U, bool KEEP_STORES>
__global__ __launch_bounds__(1024, 8) void pow64_banked_kernel(const double *__restrict__ a, const double *__restrict__ b, double s,
                                                                double *__restrict__ out, size_t n_vec, int tail, int nt) {
    typedef double T;
    typedef VecTraits<T>::vec_t V;
    constexpr int W = VecTraits<T>::width, BLOCK = 1024;
    OpCtx<PowBanked> ctx;
    ctx.init();
    const V *av = reinterpret_cast<const V *>(a), *bv = reinterpret_cast<const V *>(b);
    V *ov = reinterpret_cast<V *>(out);
    auto eval = [&](const V &xa, const V &xb) {
        if constexpr (KIND == 0) return apply_vec<PowBanked, T>(ctx, xa, xb);
        else return apply_vec_scalar<PowBanked, T, KIND == 2>(ctx, xa, s);
    };
    constexpr size_t kTile = (size_t)BLOCK * U;
    const size_t full_tiles = n_vec / kTile;
    for (size_t t = blockIdx.x; t < full_tiles; t += gridDim.x) {  // every wave's trip count is finite: t only grows
        const size_t base = t * kTile + threadIdx.x;
        V va[U], vb[U];
        if (nt & kLoadNt) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                va[u] = load_stream_as(T, av + base + (size_t)u * BLOCK, true);
                if constexpr (KIND == 0) vb[u] = load_stream_as(T, bv + base + (size_t)u * BLOCK, true);
            }
        } else {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                va[u] = load_stream_as(T, av + base + (size_t)u * BLOCK, false);
                if constexpr (KIND == 0) vb[u] = load_stream_as(T, bv + base + (size_t)u * BLOCK, false);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) store_stream_as(T, ov + base + (size_t)u * BLOCK, eval(va[u], KIND == 0 ? vb[u] : va[u]), !KEEP_STORES);
    }
    if (blockIdx.x == full_tiles % gridDim.x) {  // the partial tile and the n % W tail: the workgroup whose turn it would be
        for (size_t i = full_tiles * kTile + threadIdx.x; i < n_vec; i += BLOCK) {
            const V va = load_stream(av + i);
            V vb = va;
            if constexpr (KIND == 0) vb = load_stream(bv + i);
            store_stream(ov + i, eval(va, vb));
        }
        if (threadIdx.x == 0) {
            for (int k = 0; k < tail; ++k) {
                const T x = a[n_vec * W + k];
                const T y = KIND == 0 ? b[n_vec * W + k] : s;
                out[n_vec * W + k] = KIND == 2 ? PowBanked::apply(y, x) : PowBanked::apply(x, y);
            }
        }
    }
}

#endif  // SMHIP_POW64_BANKED

template <typename Op> struct IsHeavy : std::false_type {};
// float / double pow only: integer pow is a short square-and-multiply loop, and the plain launch beats the pipelined one on
// it for every exponent distribution tried (tools/ipow_exp.py: 80 % vs 64 % of peak for exponents < 32, 42 % vs 40 % for
// 20-bit exponents)
template <typename T> struct IsHeavy<PowOp<T>> : std::integral_constant<bool, std::is_floating_point<T>::value> {};
// Vectors per lane of the one-shot tile form, by element type and by whether the second operand is an array (KIND 0) or
// a scalar.  double pow, N = 2^26 with random bases (tools/pow64_rate.py, profiles/r02_pow64_rate.txt): scalar exponent
// 207 us with two, 195 with three, 197 with four; array exponents 249 / 249 / 251.
#ifndef SMHIP_HEAVY_F32_TILE_ARRAY
#define SMHIP_HEAVY_F32_TILE_ARRAY 2
#endif
#ifndef SMHIP_HEAVY_F32_TILE_SCALAR
#define SMHIP_HEAVY_F32_TILE_SCALAR 2
#endif
#ifndef SMHIP_HEAVY_F64_TILE_ARRAY
#define SMHIP_HEAVY_F64_TILE_ARRAY 2
#endif
#ifndef SMHIP_HEAVY_F64_TILE_SCALAR
#define SMHIP_HEAVY_F64_TILE_SCALAR 3
#endif
template <typename T, int KIND> struct HeavyTile { static constexpr int value = KIND == 0 ? SMHIP_HEAVY_F32_TILE_ARRAY : SMHIP_HEAVY_F32_TILE_SCALAR; };
template <int KIND> struct HeavyTile<double, KIND> { static constexpr int value = KIND == 0 ? SMHIP_HEAVY_F64_TILE_ARRAY : SMHIP_HEAVY_F64_TILE_SCALAR; };

// Very large arrays go out as several launches.  The f32 add holds 81-82 % of HBM peak up to N = 2^28 and sagged to 78 % at
// 2^30 / 77 % at 2^31 as ONE launch.  Round 2 blamed address translation (UTCL1 misses per 2 MiB page); round 3 disproved it:
// memory mapped through hipMemCreate / hipMemMap shows 24 x the UTCL1 misses and a UTCL2 that is busy 55 % of the kernel
// instead of 2 % -- and runs 2 % FASTER (tools/sweep_vmm.hip, tools/pmc_vmm.sh -> profiles/r03_pmc_vmm.txt); the distance
// between the three streams does nothing either (tools/sweep_distance.hip).  What matters is the launch's LENGTH: the same
// 12 GiB, same placement, as four launches of 2^28 run at 79.9 %, as sixteen of 2^26 at 80.7 % (one launch: 78.0 %).  A
// launch's workgroups are dealt to the eight XCDs in order and each XCD works through its share at its own pace; over
// hundreds of thousands of workgroups their fronts drift apart and the DRAM pages they share stop being open for each
// other; a kernel boundary lines them up again.  So: pieces of 2^24 vectors (256 MiB per operand) once an operand
// exceeds 1 GiB -- and, found later, even the headline's 1 GiB operands gain 1 % from going out as TWO launches
// (internal.h: piece_for has the rule and its numbers).

// Half-integer exponents whose double-double product chain (sm_pow64.h: pow_halfint_n) is slower than the one-exponent exp(s log a)
inline bool m2_is_slow_chain(double s) {
    return s == -7.5 || s == -6.5 || s == -5.5 || s == -4.5 || s == -3.5 || s == 7.5;
}
// Launches the heavy form of `Op` (KIND 0: a op b, 1: a op s, 2: s op a).
template <typename T, typename Op, int KIND>
void launch_heavy(const T *pa, const T *pb, T value, T *po, size_t n_vec, int tail, hipStream_t s);

#if SMHIP_POW64_BANKED
constexpr int kBankedBlock = 1024;
#endif
template <typename T, typename Op, int KIND>
void launch_heavy_piece(const T *pa, const T *pb, T value, T *po, size_t n_vec, int tail, int nt, hipStream_t s);

template <typename T, typename Op, int KIND>
void launch_heavy(const T *pa, const T *pb, T value, T *po, size_t n_vec, int tail, hipStream_t s) {
    constexpr int W = VecTraits<T>::width;
    const int nt = KIND == 0 ? stream_policy({{pa, n_vec * 16}, {pb, n_vec * 16}}, {po, n_vec * 16}) : stream_policy({{pa, n_vec * 16}}, {po, n_vec * 16});
    if (const size_t piece = piece_for(n_vec, KIND == 0 ? 3 : 2)) {  // large: several launches (internal.h: piece_for)
        for (size_t v0 = 0;; v0 += piece) {
            const bool last = v0 + piece >= n_vec;
            launch_heavy_piece<T, Op, KIND>(pa + v0 * W, KIND == 0 ? pb + v0 * W : pb, value, po + v0 * W, last ? n_vec - v0 : piece, last ? tail : 0, nt, s);
            if (last) return;
        }
    }
    launch_heavy_piece<T, Op, KIND>(pa, pb, value, po, n_vec, tail, nt, s);
}

template <typename T, typename Op, int KIND>
void launch_heavy_piece(const T *pa, const T *pb, T value, T *po, size_t n_vec, int tail, int nt, hipStream_t s) {
#if SMHIP_POW64_BANKED
    if constexpr (std::is_same<Op, PowOp<double>>::value) {
        // double pow reads its tables from bank-private replicas: no lookup of a wave can collide with another (ops.hip.h: PowBanked)
        constexpr int U = SMHIP_POW64_BANKED_U;
        const size_t tiles = n_vec / ((size_t)kBankedBlock * U) + 1;
        const size_t resident = (size_t)compute_units() * 2;  // two 1024-thread workgroups per CU: 80 KiB of LDS and 64 VGPRs each
        const unsigned grid = (unsigned)(tiles < resident ? tiles : resident);
        if (nt & kStoreKeep) hipLaunchKernelGGL((pow64_banked_kernel<KIND, U, true>), dim3(grid), dim3(kBankedBlock), 0, s, pa, pb, value, po, n_vec, tail, nt);
        else hipLaunchKernelGGL((pow64_banked_kernel<KIND, U, false>), dim3(grid), dim3(kBankedBlock), 0, s, pa, pb, value, po, n_vec, tail, nt);
        return;
    }
#endif
    // the lightest table form is served best by two vectors per lane (same box, tools/pow64_rate.py with -DSMHIP_HEAVY_F64_TILE_SCALAR=2 / 3 / 4:
    // LEVEL 2 78.0-78.3 / 77.2-77.7 / 76.5-77.0 %; LEVEL 1 75.1-76.7 / 77.7-78.3 / 77.1-77.5 %; profiles/r04_pow64_rate.txt)
    constexpr int U = std::is_same<Op, PowScalar64<2>>::value ? 2 : HeavyTile<T, KIND>::value;
    const size_t tiles = n_vec / ((size_t)kTileBlock * U) + 1;  // the last workgroup: partial tile + scalar tail (maybe empty)
    if (nt & kStoreKeep) hipLaunchKernelGGL((flat_tile_kernel<T, Op, KIND, U, true>), dim3((unsigned)tiles), dim3(kTileBlock), 0, s, pa, pb, value, po, n_vec, tail, nt, FastDiv(1));
    else hipLaunchKernelGGL((flat_tile_kernel<T, Op, KIND, U, false>), dim3((unsigned)tiles), dim3(kTileBlock), 0, s, pa, pb, value, po, n_vec, tail, nt, FastDiv(1));
}

#ifndef SMHIP_FLAT_ROWS_U
#define SMHIP_FLAT_ROWS_U 2
#endif
// a (rows x cols, dense) op one row / one column of b (KIND 3 / 4 of flat_tile_kernel), every built-in Op
template <typename T, typename Op, int U>
void launch_rows(const T *pa, const T *pb, T *po, size_t n_vec, bool b_is_row, int nt, FastDiv cv, hipStream_t s) {
    const size_t tiles = n_vec / ((size_t)kTileBlock * U) + 1;
    const dim3 grid((unsigned)tiles), block(kTileBlock);
    if (b_is_row) {
        // few long rows (the broadcast row outlives no row in the L2): the kernel walks them column block by column block
        const size_t rows = n_vec / cv.d, row_bytes = (size_t)cv.d * 16;
        static const int lg = [] { const char *e = getenv("SMHIP_ROWS_WALK_LOG2"); const int v = e && *e ? atoi(e) : 10; return v < -1 ? -1 : (v > 15 ? 15 : v); }();  // workgroups per visit of a row: 2^lg (-1: flat walk)
        const int walk = (lg >= 0 && rows >= 2 && rows <= 4096 && row_bytes >= ((size_t)4 << 20) && cv.d % ((uint32_t)(kTileBlock * U) << lg) == 0) ? (int)rows | (lg << 16) : 0;
        if (nt & kStoreKeep) hipLaunchKernelGGL((flat_tile_kernel<T, Op, 3, U, true>), grid, block, 0, s, pa, pb, T{}, po, n_vec, walk, nt, cv);
        else hipLaunchKernelGGL((flat_tile_kernel<T, Op, 3, U, false>), grid, block, 0, s, pa, pb, T{}, po, n_vec, walk, nt, cv);
    } else {
        if (nt & kStoreKeep) hipLaunchKernelGGL((flat_tile_kernel<T, Op, 4, U, true>), grid, block, 0, s, pa, pb, T{}, po, n_vec, 0, nt, cv);
        else hipLaunchKernelGGL((flat_tile_kernel<T, Op, 4, U, false>), grid, block, 0, s, pa, pb, T{}, po, n_vec, 0, nt, cv);
    }
}
template <typename T, typename Op>
int run_heavy_rows(const void *a, const void *b, void *out, size_t rows, size_t cols, bool b_is_row, hipStream_t s) {
    constexpr int W = VecTraits<T>::width;
    // vectors per lane: the array form's tile for the heavy Ops (per-lane exponents); two for the others, like the row
    // kernel's two rows per lane
    constexpr int U = IsHeavy<Op>::value ? HeavyTile<T, 0>::value : SMHIP_FLAT_ROWS_U;
    const size_t n_vec = rows * (cols / W);
    const T *pa = static_cast<const T *>(a), *pb = static_cast<const T *>(b);
    T *po = static_cast<T *>(out);
    const size_t bytes = rows * cols * sizeof(T);
    const int nt = stream_policy({{pa, bytes}}, {po, bytes});
    const FastDiv cv((uint32_t)(cols / W));
    // A dense operand that is read with `nt` is not coming from the Infinity Cache: it is COLD (internal.h: the residency rule)
    // or larger than the cache.  Then ONE vector per lane is the faster shape for the light Ops -- config 3's multiply 23.1 -> 22.5 us (72.6 -> 74.5 %),
    // twice its size 44.2 -> 42.5 us (75.9 -> 79.0 %) -- while replayed and chained operands keep two (19.27 against 19.41 us;
    // tools/cold_rates.py, profiles/r03_rows_u.txt): twice the workgroups retire, and free their slots, half a tile earlier.
    const bool cold = (nt & kLoadNt) != 0;  // cold by the residency rule, or simply larger than the cache
    static const int cold_u = [] { const char *e = getenv("SMHIP_FLAT_ROWS_COLD_U"); return e && *e ? atoi(e) : 1; }();  // experiments
    if (!IsHeavy<Op>::value && U > 1 && cold && cold_u == 1) launch_rows<T, Op, 1>(pa, pb, po, n_vec, b_is_row, nt, cv, s);
    else launch_rows<T, Op, U>(pa, pb, po, n_vec, b_is_row, nt, cv, s);
    SMHIP_LAUNCH_CHECK("heavy rows");
    return SMHIP_OK;
}

inline int grid_for(size_t threads, int block, unsigned *grid) {
    const size_t g = (threads + block - 1) / block;
    if (g > 0x7fffffffu) return fail(SMHIP_ERR_UNSUPPORTED, "array too large for one launch (%zu workgroups)", g);
    *grid = (unsigned)g;
    return SMHIP_OK;
}

template <typename T, typename Op>
int run_contiguous(const void *a, const void *b, void *out, size_t n, hipStream_t s) {
    constexpr int W = VecTraits<T>::width;
    const T *pa = static_cast<const T *>(a), *pb = static_cast<const T *>(b);
    T *po = static_cast<T *>(out);
    unsigned grid;
    const size_t n_vec = n / W;
    const int tail = (int)(n % W);
    const size_t threads = n_vec + (tail ? 1 : 0);
    if constexpr (IsHeavy<Op>::value) {
        if (n_vec / ((size_t)kTileBlock * 2) + 1 > 0x7fffffffu) return fail(SMHIP_ERR_UNSUPPORTED, "array too large for one launch");
        launch_heavy<T, Op, 0>(pa, pb, T{}, po, n_vec, tail, s);
    } else if (const size_t piece = piece_for(n_vec, 3)) {
        const int pol = stream_policy({{pa, n * sizeof(T)}, {pb, n * sizeof(T)}}, {po, n * sizeof(T)});  // above the cache: nt both ways
        for (size_t v0 = 0; v0 < n_vec || (v0 == n_vec && tail); v0 += piece) {
            const bool last = v0 + piece >= n_vec;
            const size_t nv = last ? n_vec - v0 : piece;
            if (int rc = grid_for(nv + (last && tail ? 1 : 0), kBlockBig, &grid)) return rc;
            hipLaunchKernelGGL((contiguous_vec_kernel<T, Op, kBlockBig, false>), dim3(grid), dim3(kBlockBig), 0, s, pa + v0 * W, pb + v0 * W, po + v0 * W, nv,
                               last ? tail : 0, pol & ~kStoreKeep);
            if (last) break;
        }
    } else if (n_vec >= kBigThreshold) {
        if (int rc = grid_for(threads, kBlockBig, &grid)) return rc;
        const int pol = stream_policy({{pa, n * sizeof(T)}, {pb, n * sizeof(T)}}, {po, n * sizeof(T)});
        if (pol & kStoreKeep) hipLaunchKernelGGL((contiguous_vec_kernel<T, Op, kBlockBig, true>), dim3(grid), dim3(kBlockBig), 0, s, pa, pb, po, n_vec, tail, pol);
        else hipLaunchKernelGGL((contiguous_vec_kernel<T, Op, kBlockBig, false>), dim3(grid), dim3(kBlockBig), 0, s, pa, pb, po, n_vec, tail, pol);
    } else {
        if (int rc = grid_for(threads, kBlockSmall, &grid)) return rc;
        // below kBigThreshold vectors (16 MiB per operand: 48 MiB in all at most) a footprint above the keep-store floor
        // is a corner this form does not serve
        hipLaunchKernelGGL((contiguous_vec_kernel<T, Op, kBlockSmall, false>), dim3(grid), dim3(kBlockSmall), 0, s, pa, pb, po, n_vec, tail, stream_policy({{pa, n * sizeof(T)}, {pb, n * sizeof(T)}}, {po, n * sizeof(T)}) & ~kStoreKeep);
    }
    SMHIP_LAUNCH_CHECK("contiguous");
    return SMHIP_OK;
}

// sm::pow(a, s) with an exponent whose power is ONE correctly rounded IEEE operation: x^2 = x*x, x^1 = x, x^-1 = 1/x,
// x^0.5 = sqrt(x) (with pow's own answers for -0 and -inf).  These are exact-rounded -- never worse than the general
// 2^(y log2 x) path's <= 1 ULP -- and stream at the array-scalar rate (81 % of peak instead of 71 %).  Squares are what
// the reference's own float pow benchmark raises to (benchmark/pow.cpp:33-49).
template <typename T> struct PowSquare { static __device__ __forceinline__ T apply(T a, T) { return a * a; } };
template <typename T> struct PowIdentity { static __device__ __forceinline__ T apply(T a, T) { return a * T(1); } };  // quiets a signalling NaN
template <typename T> struct PowReciprocal { static __device__ __forceinline__ T apply(T a, T) { return T(1) / a; } };
template <typename T> struct PowSqrt {
    static __device__ __forceinline__ T apply(T a, T) {
        if (a == T(0)) return T(0);                               // pow(-0, 0.5) = +0 (sqrt keeps the sign)
        if (a == -__builtin_huge_val()) return (T)__builtin_huge_val();  // pow(-inf, 0.5) = +inf (sqrt: NaN)
        if constexpr (sizeof(T) == 4) return __builtin_sqrtf(a);
        else return __builtin_sqrt(a);
    }
};

// One read and one write stream through scalar_vec_kernel: workgroups of 256 at every size (tools/sweep_scalar.hip,
// profiles/r01_sweep_scalar.txt: 81.7 % of peak at N = 2^28 against 78.7 % with 1024, and two or more vectors per lane
// lose 4-10 %); very large arrays in pieces (internal.h: piece_for).
template <typename T, typename Op, bool SWAPPED>
int launch_scalar_stream(const T *pa, T value, T *po, size_t n_vec, int tail, hipStream_t s) {
    constexpr int W = VecTraits<T>::width;
    const size_t n = n_vec * W + (size_t)tail;
    const int pol = stream_policy({{pa, n * sizeof(T)}}, {po, n * sizeof(T)});
    unsigned grid;
    if (const size_t piece = piece_for(n_vec, 2)) {
        for (size_t v0 = 0;; v0 += piece) {
            const bool last = v0 + piece >= n_vec;
            const size_t nv = last ? n_vec - v0 : piece;
            if (int rc = grid_for(nv + (last && tail ? 1 : 0), kBlockSmall, &grid)) return rc;
            hipLaunchKernelGGL((scalar_vec_kernel<T, Op, kBlockSmall, SWAPPED>), dim3(grid), dim3(kBlockSmall), 0, s, pa + v0 * W, value, po + v0 * W, nv,
                               last ? tail : 0, pol);
            if (last) break;
        }
        return SMHIP_OK;
    }
    if (int rc = grid_for(n_vec + (tail ? 1 : 0), kBlockSmall, &grid)) return rc;
    hipLaunchKernelGGL((scalar_vec_kernel<T, Op, kBlockSmall, SWAPPED>), dim3(grid), dim3(kBlockSmall), 0, s, pa, value, po, n_vec, tail, pol);
    return SMHIP_OK;
}

template <typename T, typename Op, bool SWAPPED>
int run_scalar(const void *a, T value, size_t n, void *out, hipStream_t s) {
    constexpr int W = VecTraits<T>::width;
    const T *pa = static_cast<const T *>(a);
    T *po = static_cast<T *>(out);
    const size_t n_vec = n / W;
    const int tail = (int)(n % W);
    if constexpr (IsHeavy<Op>::value && std::is_floating_point<T>::value && !SWAPPED) {
        if (value == T(2) || value == T(1) || value == T(-1) || value == T(0.5)) {
            int rc;
            if (value == T(2)) rc = launch_scalar_stream<T, PowSquare<T>, false>(pa, value, po, n_vec, tail, s);
            else if (value == T(1)) rc = launch_scalar_stream<T, PowIdentity<T>, false>(pa, value, po, n_vec, tail, s);
            else if (value == T(-1)) rc = launch_scalar_stream<T, PowReciprocal<T>, false>(pa, value, po, n_vec, tail, s);
            else rc = launch_scalar_stream<T, PowSqrt<T>, false>(pa, value, po, n_vec, tail, s);
            if (rc) return rc;
            SMHIP_LAUNCH_CHECK("array_scalar pow (exact form)");
            return SMHIP_OK;
        }
        if constexpr (std::is_same<T, double>::value) {
            int m2;
            static const int max_m2 = [] { const char *e = getenv("SMHIP_POW_HALFINT_MAX"); return e && *e ? atoi(e) : 16; }();  // tools/pow64_halfint.py
            // the chains slower than exp(s log a) with a known exponent (pow_core_u below: 171-172 us): -7.5 213.6 us, -6.5 191.5, -5.5 188.0,
            // -4.5 174.9, -3.5 175.1, 7.5 175.5-181 (profiles/r03_pow64_halfint.txt, r04_pow64_rate.txt)
            const bool slow_chain = m2_is_slow_chain(value);
            if (!slow_chain && smpow64::halfint_exponent(value, &m2) && (m2 < 0 ? -m2 : m2) <= max_m2) {
                // sm_pow64.h: pow_halfint_n.  Through the one-shot tile form like the general pow (three vectors per lane; with
                // one vector per lane the chain's latency adds to the load's: 189.7 us, no faster than exp(s log a)), one kernel
                // per exponent (with a run-time exponent the chain is a loop with branches: 174-218 us).  N = 2^26, random
                // bases (tools/pow64_halfint.py, profiles/r03_pow64_halfint.txt): pow(a, 2.5) 193.8 -> 168.2 us (69.3 -> 79.8 %
                // of HBM peak), integers 166.6-169.9 us, a * 2.5 on the same box 165.4 us.
                if (n_vec / ((size_t)kTileBlock * 2) + 1 > 0x7fffffffu) return fail(SMHIP_ERR_UNSUPPORTED, "array too large for one launch");
                const double *none = nullptr;
                switch (m2) {
#define SMHIP_HALFINT_CASE(M) case M: launch_heavy<double, PowHalfInt64<M>, 1>(pa, none, value, po, n_vec, tail, s); break;
                    SMHIP_HALFINT_CASE(-16) SMHIP_HALFINT_CASE(-14) SMHIP_HALFINT_CASE(-12) 
                    SMHIP_HALFINT_CASE(-10) SMHIP_HALFINT_CASE(-8) SMHIP_HALFINT_CASE(-6) SMHIP_HALFINT_CASE(-5)
                    SMHIP_HALFINT_CASE(-4) SMHIP_HALFINT_CASE(-3) SMHIP_HALFINT_CASE(-2) SMHIP_HALFINT_CASE(-1) SMHIP_HALFINT_CASE(1) SMHIP_HALFINT_CASE(2)
                    SMHIP_HALFINT_CASE(3) SMHIP_HALFINT_CASE(4) SMHIP_HALFINT_CASE(5) SMHIP_HALFINT_CASE(6) SMHIP_HALFINT_CASE(7) SMHIP_HALFINT_CASE(8)
                    SMHIP_HALFINT_CASE(9) SMHIP_HALFINT_CASE(10) SMHIP_HALFINT_CASE(11) SMHIP_HALFINT_CASE(12) SMHIP_HALFINT_CASE(13) SMHIP_HALFINT_CASE(14)
                    SMHIP_HALFINT_CASE(16)
#undef SMHIP_HALFINT_CASE
                    default: return fail(SMHIP_ERR_INVALID, "half-integer exponent out of range");
                }
                SMHIP_LAUNCH_CHECK("array_scalar pow (half-integer exponent)");
                return SMHIP_OK;
            }
            // any other exponent up to 1024 in magnitude: the general arithmetic minus what only an UNKNOWN exponent needs
            // (sm_pow64.h: pow_core_u).  N = 2^26, random bases (tools/pow64_rate.py, profiles/r04_pow64_rate.txt).
            static const int max_level = [] { const char *e = getenv("SMHIP_POW_SCALAR_LEVEL"); return e && *e ? atoi(e) : 2; }();
            int level = smpow64::scalar_level(value);
            if (level > max_level) level = max_level;
            if (level > 0) {
                if (n_vec / ((size_t)kTileBlock * 2) + 1 > 0x7fffffffu) return fail(SMHIP_ERR_UNSUPPORTED, "array too large for one launch");
                const double *none = nullptr;
                if (level == 2) launch_heavy<double, PowScalar64<2>, 1>(pa, none, value, po, n_vec, tail, s);
                else launch_heavy<double, PowScalar64<1>, 1>(pa, none, value, po, n_vec, tail, s);
                SMHIP_LAUNCH_CHECK("array_scalar pow (one exponent)");
                return SMHIP_OK;
            }
        }
    }
    constexpr bool kHeavy = IsHeavy<Op>::value;
    if constexpr (kHeavy) {
        if (n_vec / ((size_t)kTileBlock * 2) + 1 > 0x7fffffffu) return fail(SMHIP_ERR_UNSUPPORTED, "array too large for one launch");
        launch_heavy<T, Op, SWAPPED ? 2 : 1>(pa, static_cast<const T *>(nullptr), value, po, n_vec, tail, s);
    } else {
        if (int rc = launch_scalar_stream<T, Op, SWAPPED>(pa, value, po, n_vec, tail, s)) return rc;
    }
    SMHIP_LAUNCH_CHECK("array_scalar");
    return SMHIP_OK;
}

template <typename T, typename Op, bool SWAPPED>
int run_devscalar(const void *a, const void *sp, size_t n, void *out, hipStream_t s) {
    constexpr int W = VecTraits<T>::width;
    const T *pa = static_cast<const T *>(a);
    const T *ps = static_cast<const T *>(sp);
    T *po = static_cast<T *>(out);
    unsigned grid;
    const size_t n_vec = n / W;
    const int tail = (int)(n % W);
    if (int rc = grid_for(n_vec + (tail ? 1 : 0), kBlockSmall, &grid)) return rc;
    hipLaunchKernelGGL((devscalar_vec_kernel<T, Op, kBlockSmall, SWAPPED>), dim3(grid), dim3(kBlockSmall), 0, s, pa, ps, po, n_vec, tail, stream_policy({{pa, n * sizeof(T)}}, {po, n * sizeof(T)}));
    SMHIP_LAUNCH_CHECK("array_devscalar");
    return SMHIP_OK;
}

// op x dtype dispatch: F(T, Op) expands to a call.
#define SMHIP_DISPATCH_OP(T, F)                                  \
    switch (op) {                                                \
        case SMHIP_OP_ADD: return F(T, AddOp<T>);                \
        case SMHIP_OP_SUB: return F(T, SubtractOp<T>);           \
        case SMHIP_OP_MUL: return F(T, MultiplyOp<T>);           \
        case SMHIP_OP_DIV: return F(T, DivideOp<T>);             \
        case SMHIP_OP_POW: return F(T, PowOp<T>);                \
        case SMHIP_OP_LEFT: return F(T, LeftOp<T>);                \
    }                                                            \
    break;
#define SMHIP_DISPATCH(F)                                        \
    switch (dtype) {                                             \
        case SMHIP_F32: SMHIP_DISPATCH_OP(float, F)              \
        case SMHIP_F64: SMHIP_DISPATCH_OP(double, F)             \
        case SMHIP_I32: SMHIP_DISPATCH_OP(int32_t, F)            \
        case SMHIP_I64: SMHIP_DISPATCH_OP(int64_t, F)            \
    }

}  // namespace

int launch_contiguous(int op, int dtype, const void *a, const void *b, void *out, size_t n, hipStream_t s) {
    if (n == 0) return SMHIP_OK;
#define F(T, OP) run_contiguous<T, OP>(a, b, out, n, s)
    SMHIP_DISPATCH(F)
#undef F
    return fail(SMHIP_ERR_INVALID, "contiguous: bad op %d / dtype %d", op, dtype);
}

// a dense (rows x cols) array against one row / one column of b -- config 3's shape -- through the flat tile kernel
// (broadcast.hip routes that shape here for the built-in Ops)
int launch_flat_rows(int op, int dtype, const void *a, const void *b, void *out, size_t rows, size_t cols, bool b_is_row, hipStream_t s) {
    if (rows == 0 || cols == 0) return SMHIP_OK;
#define F(T, OP) run_heavy_rows<T, OP>(a, b, out, rows, cols, b_is_row, s)
    SMHIP_DISPATCH(F)
#undef F
    return fail(SMHIP_ERR_INVALID, "flat rows: bad op %d / dtype %d", op, dtype);
}

int launch_array_scalar(int op, int dtype, const void *a, const void *value_host, size_t n, void *out, hipStream_t s) {
    if (n == 0) return SMHIP_OK;
#define F(T, OP) run_scalar<T, OP, false>(a, *static_cast<const T *>(value_host), n, out, s)
    SMHIP_DISPATCH(F)
#undef F
    return fail(SMHIP_ERR_INVALID, "array_scalar: bad op %d / dtype %d", op, dtype);
}

int launch_array_devscalar(int op, int dtype, const void *a, const void *value_dev, size_t n, void *out, bool swapped,
                           hipStream_t s) {
    if (n == 0) return SMHIP_OK;
    if (swapped) {
#define F(T, OP) run_devscalar<T, OP, true>(a, value_dev, n, out, s)
        SMHIP_DISPATCH(F)
#undef F
    } else {
#define F(T, OP) run_devscalar<T, OP, false>(a, value_dev, n, out, s)
        SMHIP_DISPATCH(F)
#undef F
    }
    return fail(SMHIP_ERR_INVALID, "array_devscalar: bad op %d / dtype %d", op, dtype);
}

}  // namespace smhip
