// reduce.hip -- whole-array reductions.
//
// Replaces dot_product<T> (reference include/math/product.h:8-224, reached
// through SMArray::operator% at include/SMArray.h:213-215): one host thread,
// 8 f32 lane accumulators that stop absorbing addends past 2^24 (SURVEY 0:
// the dot of 2^28 ones returns 2^27).  Adds sum() and the fused a op b + sum
// of BASELINE config 5, which the reference lacks.
//
// Structure: each lane streams 16-byte vectors and accumulates privately
// (fp64 for f32/f64, wrapping 64-bit integers for i32/i64 -- so integer results
// are bit-identical to the reference in any order), then a 64-lane wavefront
// shuffle tree (__shfl_down), then LDS across the workgroup's waves, one partial
// per workgroup, and a second single-workgroup pass over the partials in a
// fixed order: results are deterministic run to run.
// Roofline: HBM-bound; sizeof(T) B/elem (sum), 2*sizeof(T) (dot),
// 3*sizeof(T) (fused op+sum: the sum adds no traffic).
#include <stdlib.h>

#include <map>
#include <mutex>
#include <type_traits>
#include <utility>

#include "internal.h"
#include "ops.hip.h"

namespace smhip {
namespace {

using namespace dev;

constexpr int kBlock = 256;  // sum, dot, the finishing launch (the fused op+sum: SMHIP_FUSED_BLOCK below)
// 16-byte vectors per lane and operand (tools/reduce_rates.py, profiles/r01_reduce_rates.txt).  Read-only streams want
// more loads in flight than the 2R+1W streams do: a plain sum runs at 86 % of peak with two vectors per lane (65-69 %
// with one, 82-84 % with four or eight), a dot at 84 % with two per operand (79 % with one).  The fused op+sum writes as
// well and keeps ONE per operand like the streaming kernels (79.5 %; 75 % with two; profiles/r01_sweep_fused_sum.txt).
#ifndef SMHIP_SUM_VECTORS
#define SMHIP_SUM_VECTORS 2
#endif
#ifndef SMHIP_DOT_VECTORS
#define SMHIP_DOT_VECTORS 2
#endif
#ifndef SMHIP_FUSED_VECTORS
#define SMHIP_FUSED_VECTORS 1
#endif
constexpr int kSumVectors = SMHIP_SUM_VECTORS, kDotVectors = SMHIP_DOT_VECTORS;
constexpr int vec_per_thread(int mode) { return mode == 0 /* kSum */ ? kSumVectors : mode == 1 /* kDot */ ? kDotVectors : SMHIP_FUSED_VECTORS; }

template <typename T, bool INTEGER = std::is_integral<T>::value> struct AccOf { typedef double type; };
template <typename T> struct AccOf<T, true> { typedef uint64_t type; };  // wrapping: exact modulo 2^64, hence modulo 2^(8 sizeof T)

// Wave total through DPP moves (v_mov_b32 row_shr / row_bcast: VALU-speed lane exchange inside the SIMD) instead of
// __shfl_down, which the compiler lowers to ds_bpermute_b32 -- an LDS-crossbar round trip per 32-bit half and stage, with a
// full lgkmcnt wait behind each: 6 stages x 2 halves for the wave, and round 2 ran the same 6 stages AGAIN in wave 0 to add
// four numbers.  Every wave of the fused op+sum kernel carried ~1000 cycles of that behind its last store, holding its slot
// (the kernel ran 2 % behind the plain add: 496-498 us against 486 on one box, tools/sweep_fused2.hip).
// The scan: row_shr 1, 2, 4, 8 leave each row of 16 lanes' running sum in its lane 15; row_bcast:15 adds it to the next
// row (rows 1 and 3), row_bcast:31 adds lane 31 to rows 2 and 3: lane 63 holds the wave's total.  Lanes without a source
// receive `old` = 0, the sum's identity.  The order of the additions is fixed, so the bits are the same on every run
// (they differ from round 2's tree order in the last place, as any reassociation does).
constexpr int kWaveTotalLane = 63;
template <int CTRL, int ROW_MASK> __device__ __forceinline__ double dpp_move(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xf, false);
    return __hiloint2double(hi, lo);
}
template <int CTRL, int ROW_MASK> __device__ __forceinline__ uint64_t dpp_move(uint64_t v) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)v, CTRL, ROW_MASK, 0xf, false);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(v >> 32), CTRL, ROW_MASK, 0xf, false);
    return ((uint64_t)hi << 32) | lo;
}
// the wave's total, valid in lane kWaveTotalLane
template <typename A> __device__ __forceinline__ A wave_reduce(A v) {
    v += dpp_move<0x111, 0xf>(v);  // row_shr:1
    v += dpp_move<0x112, 0xf>(v);  // row_shr:2
    v += dpp_move<0x114, 0xf>(v);  // row_shr:4
    v += dpp_move<0x118, 0xf>(v);  // row_shr:8
    v += dpp_move<0x142, 0xa>(v);  // row_bcast:15 into rows 1 and 3
    v += dpp_move<0x143, 0xc>(v);  // row_bcast:31 into rows 2 and 3
    return v;
}

// Workgroup total in thread 0: the waves' totals meet in LDS and thread 0 adds them in wave order.
template <typename A, int BLOCK> __device__ __forceinline__ A block_reduce(A v) {
    __shared__ A lds[BLOCK / 64];
    v = wave_reduce(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == kWaveTotalLane) lds[wave] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        v = lds[0];
#pragma unroll
        for (int w = 1; w < BLOCK / 64; ++w) v += lds[w];
    }
    return v;
}

template <typename T, typename A> __device__ __forceinline__ A widen(T x) {
    if constexpr (std::is_integral<T>::value && std::is_signed<T>::value) return (A)(int64_t)x;  // sign-extended
    else return (A)x;
}

// acc += x*y.  f32: the product of two 24-bit significands is exact in fp64, so the
// whole dot is an fp64 fma chain (the reference's -mfma build fuses too, but into f32
// lanes).  f64: fused, one rounding per term.  i32/i64: product wraps in T like
// _mm256_mullo_epi32, then accumulates mod 2^64.
// The generic dot_product<T> (product.h:8-20: `T sum = 0; sum += a[i] * b[i]`) for 8- / 16-bit and unsigned types: the product
// is formed in the promoted type and the sum cut back to T every step -- the exact sum of products modulo 2^(8 sizeof T).
// Here: widened operands, 64-bit wrapping product and sum, cut to T once at the end (the same residue).
template <typename T, typename A> __device__ __forceinline__ void add_prod(A &acc, T x, T y) {
    if constexpr (std::is_same<T, int32_t>::value || std::is_same<T, int64_t>::value) acc += widen<T, A>(MultiplyOp<T>::apply(x, y));
    else acc += widen<T, A>(x) * widen<T, A>(y);
}
template <> __device__ __forceinline__ void add_prod<float, double>(double &acc, float x, float y) { acc = __builtin_fma((double)x, (double)y, acc); }
template <> __device__ __forceinline__ void add_prod<double, double>(double &acc, double x, double y) { acc = __builtin_fma(x, y, acc); }

enum Mode { kSum = 0, kDot = 1, kFused = 2 };

// One vector's contribution: MODE kSum: acc += a;  kDot: acc += a*b;  kFused: out = a op b, acc += out.
// KEEP: the write side's policy (ops.hip.h) at compile time -- as a run-time branch in front of the fused kernel's one store
// it cost what it cost the plain add (1.6 %, contiguous.hip): round 2's main kernel ran at 498.9 us where the same loop
// without the branch runs at the add's 493-494 us (tools/sweep_reduce.hip, profiles/r01_sweep_fused_sum.txt).
template <typename T, typename Op, int MODE, bool KEEP, typename A>
__device__ __forceinline__ void consume(const OpCtx<Op> &ctx, A &acc, typename VecTraits<T>::vec_t va, typename VecTraits<T>::vec_t vb,
                                        typename VecTraits<T>::vec_t *out_slot) {
    typedef typename VecTraits<T>::vec_t V;
    constexpr int W = VecTraits<T>::width;
    if constexpr (MODE == kFused) {
        const V r = apply_vec<Op, T>(ctx, va, vb);
        store_stream_as(T, out_slot, r, !KEEP);
#pragma unroll
        for (int k = 0; k < W; ++k) acc += widen<T, A>(r[k]);
    } else if constexpr (MODE == kDot) {
#pragma unroll
        for (int k = 0; k < W; ++k) add_prod<T, A>(acc, va[k], vb[k]);
    } else {
#pragma unroll
        for (int k = 0; k < W; ++k) acc += widen<T, A>(va[k]);
    }
}

// What a reduction hands back.  *out8 receives 8 bytes:
//   floats            the fp64 total;
//   ints, AS_DOUBLE   (double) of the exact 64-bit total         (sum, fused op+sum);
//   ints, !AS_DOUBLE  the total wrapped to T, sign-extended to int64 (dot: per-rank partials add up mod 2^32 / 2^64
//                     exactly like the reference's _mm256_add_epi32 accumulators, in any order).
// *out_native (optional) receives the value narrowed to T (dot's return type).
template <typename T, bool AS_DOUBLE>
__device__ __forceinline__ void write_result(typename AccOf<T>::type acc, void *__restrict__ out8, T *__restrict__ out_native) {
    if constexpr (std::is_floating_point<T>::value) {
        if (out8) *static_cast<double *>(out8) = acc;
        if (out_native) *out_native = (T)acc;
    } else {
        const T wrapped = (T)acc;
        if (out8) {
            if constexpr (AS_DOUBLE) *static_cast<double *>(out8) = (double)(int64_t)acc;
            else *static_cast<int64_t *>(out8) = (int64_t)wrapped;
        }
        if (out_native) *out_native = wrapped;
    }
}

// Arrival counters for finish_kernel: a few zeroed uint32 per (device, stream), made on that pair's first
// reduction and never freed.  Reductions queued on one stream run one after the other and every launch leaves its
// counters at zero, so a stream's buffer is always ready for the next launch; two streams never share one.  (A caller
// stream that is destroyed leaves its 4 KiB behind; a later stream that gets the same handle finds them zeroed.)
int reduce_counters(hipStream_t s, uint32_t **out);
// A finishing launch on (device, stream) failed to enqueue: its counters may not be what the next launch expects (they
// are, if the kernel never ran -- but nothing is assumed): the next reduce_counters() call zeroes them on the stream first.
void reduce_counters_suspect(hipStream_t s);

// What the finishing launch of a multi-workgroup reduction needs (finish_kernel).
template <typename A> struct Finish {
    A *level2;           // one total per group
    uint32_t *counters;  // the arrival counter(s), zero between launches
    uint32_t gsize, groups;
    uint32_t pitch;      // blockIdx.y = plane (complex dot: real, imaginary): plane y's partials, group totals and result
                         // slot lie y * pitch accumulators / y counters / y doubles behind plane 0's
};
constexpr uint32_t kMaxGroups = 1024, kGroupTarget = 1024;

// Each workgroup owns one tile of kBlock * kVecPerThread vectors.  Full tiles (all
// but possibly the last) take a guard-free path: every load of the tile is issued
// before the first use, so kVecPerThread (x2 operands) 16-byte loads are in flight
// per lane.  (Per-vector bounds guards made the compiler wait on each load pair
// in turn -- 2.6 TB/s instead of 6.)
// The fused op+sum's workgroup: 1024 threads since round 3.  With ds_bpermute reductions 256 was the fastest (507 us per step
// against 510 / 514 with 512 / 1024); with the DPP scan and thread 0 adding the sixteen wave totals the order turned round --
// 505.3 (256), 510.8 (512), 501.8 us (1024) per step including the finishing launch, which also has a quarter of the
// partials to add (profiles/r03_reduce_variants.txt).
#ifndef SMHIP_FUSED_BLOCK
#define SMHIP_FUSED_BLOCK 1024
#endif
constexpr int block_of(int mode) { return mode == 2 /* kFused */ ? SMHIP_FUSED_BLOCK : kBlock; }
template <typename T, typename Op, int MODE, bool KEEP, int BLOCK>
__global__ __launch_bounds__(BLOCK) void reduce_kernel(const T *__restrict__ a, const T *__restrict__ b, T *__restrict__ out,
                                                           size_t n_vec, size_t n, typename AccOf<T>::type *__restrict__ partials,
                                                           void *__restrict__ out8, T *__restrict__ out_native, int nt, int single) {
    typedef typename AccOf<T>::type A;
    typedef typename VecTraits<T>::vec_t V;
    constexpr int W = VecTraits<T>::width;
    constexpr int kVecPerThread = vec_per_thread(MODE);
    constexpr size_t kTile = (size_t)BLOCK * kVecPerThread;
    const V *av = reinterpret_cast<const V *>(a), *bv = reinterpret_cast<const V *>(b);
    V *ov = reinterpret_cast<V *>(out);
    const size_t tile0 = (size_t)blockIdx.x * kTile + threadIdx.x;
    OpCtx<Op> ctx;
    ctx.init();
    A acc = A(0);
    if ((size_t)blockIdx.x * kTile + kTile <= n_vec) {
        V va[kVecPerThread], vb[kVecPerThread];
        // a branch per load, as round 2 had it: grouped under one branch this kernel measured 2 us SLOWER (497.5-498.5 against
        // 495.6-496.5 us, tools/sweep_fused2.hip -> profiles/r03_sweep_fused2.txt) -- the opposite of the strided-row kernel's
        // experience (ops.hip.h); what this kernel waits for is its one pair of loads either way
#pragma unroll
        for (int u = 0; u < kVecPerThread; ++u) {
            va[u] = load_stream_if(T, av + tile0 + (size_t)u * BLOCK, nt);
            if constexpr (MODE != kSum) vb[u] = load_stream_if(T, bv + tile0 + (size_t)u * BLOCK, nt);
            else vb[u] = va[u];
        }
#pragma unroll
        for (int u = 0; u < kVecPerThread; ++u) consume<T, Op, MODE, KEEP, A>(ctx, acc, va[u], vb[u], ov + tile0 + (size_t)u * BLOCK);
    } else {
        for (int u = 0; u < kVecPerThread; ++u) {
            const size_t i = tile0 + (size_t)u * BLOCK;
            if (i < n_vec) {
                const V va = load_stream(av + i);
                const V vb = MODE != kSum ? load_stream(bv + i) : va;
                consume<T, Op, MODE, KEEP, A>(ctx, acc, va, vb, ov + i);
            }
        }
        // scalar tail (n % W elements): the last workgroup's first lane
        if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) {
            for (size_t k = n_vec * W; k < n; ++k) {
                if constexpr (MODE == kFused) {
                    const T r = Op::apply(a[k], b[k]);
                    out[k] = r;
                    acc += widen<T, A>(r);
                } else if constexpr (MODE == kDot) add_prod<T, A>(acc, a[k], b[k]);
                else acc += widen<T, A>(a[k]);
            }
        }
    }
    acc = block_reduce<A, BLOCK>(acc);
    if (threadIdx.x == 0) {
        if (single) write_result<T, MODE != kDot>(acc, out8, out_native);  // a small array (one workgroup in all): no second launch
        else partials[blockIdx.x] = acc;
    }
}

// The second (and last) launch of a multi-workgroup reduction.  Workgroup g adds `gsize` partials IN INDEX ORDER into
// level2[g]; the last workgroup to finish -- whichever it is -- adds the level2 values in index order and writes the
// result, so the bits are the same on every run.  Round 1 ran this as two launches (fold, then finalize: 4.7 us each
// behind a 502 us main kernel, profiles/r01_add_sum_kernel_stats.csv).  Doing it inside the main kernel instead was
// measured and dropped twice: with a device-scope release per workgroup (each writes back the whole L2: 11.4 ms), and with
// the fence-free hand-over below per workgroup (131 072 returning atomics on ~130 counters serialise at the memory side:
// the 157 us sum took 1079 us).  The arrival counter resets itself.
template <typename T, bool AS_DOUBLE>
__global__ __launch_bounds__(kBlock) void finish_kernel(const typename AccOf<T>::type *__restrict__ partials, uint32_t count,
                                                        Finish<typename AccOf<T>::type> fin, void *__restrict__ out8,
                                                        T *__restrict__ out_native) {
    typedef typename AccOf<T>::type A;
    partials += (size_t)blockIdx.y * fin.pitch;
    fin.level2 += (size_t)blockIdx.y * fin.pitch;
    fin.counters += blockIdx.y;
    if (out8) out8 = static_cast<double *>(out8) + blockIdx.y;
    const uint32_t first = blockIdx.x * fin.gsize;
    const uint32_t members = first + fin.gsize <= count ? fin.gsize : count - first;
    // a thread's share of the group (<= kGroupTarget / kBlock values), every load issued before the first addition: as a
    // loop with a run-time trip count these were four dependent round trips to the memory side, most of this kernel's 7 us
    constexpr uint32_t kShare = kGroupTarget / kBlock;
    A acc = A(0);
    if (fin.gsize <= kGroupTarget) {
        A v[kShare];
#pragma unroll
        for (uint32_t u = 0; u < kShare; ++u) {
            const uint32_t i = threadIdx.x + u * kBlock;
            v[u] = i < members ? partials[first + i] : A(0);
        }
#pragma unroll
        for (uint32_t u = 0; u < kShare; ++u) acc += v[u];  // the loop's order: the same bits
    } else {
        for (uint32_t i = threadIdx.x; i < members; i += kBlock) acc += partials[first + i];
    }
    acc = block_reduce<A, kBlock>(acc);
    if (gridDim.x == 1) {
        if (threadIdx.x == 0) write_result<T, AS_DOUBLE>(acc, out8, out_native);
        return;
    }
    // The hand-over uses device-scope RELAXED atomics only: they are performed at the memory side (write-through, past
    // the per-XCD L2s), so no fence is needed -- a device-scope release fence writes back the whole L2, which right behind
    // a kernel that streamed a gigabyte of results costs more than the launch it saves.  Order between the total and the
    // ticket comes from waiting for the store's acknowledgement (vmcnt) before the ticket is taken.
    __shared__ int last;
    if (threadIdx.x == 0) {
        __hip_atomic_store(&fin.level2[blockIdx.x], acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // The ticket itself stays RELAXED: an acquire on EVERY workgroup's ticket cost each reduction 1.2 us (504.6 -> 505.8 us
        // per fused add+sum step, tools/sweep_fused2.hip).  The acquire sits behind `if (!last) return;` below instead: only the
        // workgroup that goes on to read the others' totals needs it.
        last = __hip_atomic_fetch_add(fin.counters, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1;
    }
    __syncthreads();
    if (!last) return;
    // The LAST workgroup only: an agent-scope acquire behind its ticket, so that its reads of the other workgroups' totals are
    // ordered by the memory model and not just by how gfx950 performs agent-scope atomics.  One
    // invalidate per reduction -- on every workgroup's ticket it cost 1.2 us; here it is inside the noise of the step
    // (profiles/r04_add_sum_kernel_stats.csv against r03's: finish_kernel 4.5 us before and after).
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    acc = A(0);
    {
        constexpr uint32_t kShare2 = kMaxGroups / kBlock;  // the group totals: again every load in flight before the first use
        A v[kShare2];
#pragma unroll
        for (uint32_t u = 0; u < kShare2; ++u) {
            const uint32_t i = threadIdx.x + u * kBlock;
            v[u] = i < gridDim.x ? __hip_atomic_load(&fin.level2[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : A(0);
        }
#pragma unroll
        for (uint32_t u = 0; u < kShare2; ++u) acc += v[u];
    }
    __syncthreads();  // block_reduce's LDS slots are reused
    acc = block_reduce<A, kBlock>(acc);
    if (threadIdx.x == 0) {
        __hip_atomic_store(fin.counters, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        write_result<T, AS_DOUBLE>(acc, out8, out_native);
    }
}

// ---- the ONE-LAUNCH form for small arrays ---------------------------------------------------------------------------
// Up to kOneLaunchTiles tiles (f32 sum: 2^18 elements) the main kernel's workgroups hand their totals over themselves and
// the last one to arrive adds them in index order and writes the result: no finishing launch.  Fixed tile-to-workgroup
// assignment and fixed addition order: the same bits on every run.  The workgroup that collects takes an agent-scope
// ACQUIRE behind its ticket (ADVICE r02 / VERDICT r03 #10): one cache invalidate per group instead of one per workgroup,
// so that its reads of the others' totals are ordered by the memory model and not only by how gfx950 performs agent-scope
// atomics.  Measured over 2^16 .. 2^26 elements (tools/reduce_mid_rates.py, profiles/r04_reduce_mid_rates.txt; kernels alone,
// us per call, two launches / one launch): f32 sum 2^16 5.9 / 5.1, 2^18 6.3 / 5.5, 2^20 6.3 / 8.1, 2^22 6.3 / 10.5; f32 dot
// 2^18 6.4 / 5.6, 2^22 6.9 / 8.7; complex dot 2^16 7.9 / 6.7, 2^20 9.3 / 10.3 -- the hand-over's tickets cost more than the
// finishing launch as soon as there are more than ~128 workgroups (with all tickets on ONE counter it was worse still:
// 15.2 us for the f32 sum at 2^22), and walking the tiles grid-stride instead of one tile per workgroup loses to the hardware's
// dispatcher exactly as in the streaming kernels (the fused add+sum at 2^22: 8.6 -> 29.5 us; it keeps two launches at every
// size).  So the form is used where it wins: at most 128 tiles.  What remains at mid sizes is one launch's fixed cost plus
// the data: an f32 dot of 2^22 elements (32 MiB) takes 6.9 us = 2.2 us + 32 MiB at 7 TB/s -- 61 % of peak, and at the bound of
// this launch model (VERDICT r03 asked for 70 %: 6.0 us).
constexpr size_t kOneLaunchTiles = 128;
inline bool reduce_one_launch() {
    static const bool on = [] { const char *e = getenv("SMHIP_REDUCE_ONE_LAUNCH"); return !(e && e[0] == '0'); }();  // 0: always two launches (experiments)
    return on;
}
// Tickets are taken in TWO levels: a workgroup's ticket goes to the counter of its group of kTicketGroup consecutive
// workgroups (each counter on a cache line of its own), the group's last arrival adds the group's totals in index order and
// takes a ticket at the final counter; its last arrival adds the group totals.  With all 1024 tickets on ONE counter the
// returning atomics serialise at the memory side (~8 ns each): the hand-over alone took 8 us, more than the finishing
// launch it replaces (f32 sum at 2^22: 15.2 us against 6.9 for two launches).
constexpr uint32_t kTicketGroup = 32, kCounterStride = 16;  // uint32 per counter slot: 64 bytes
template <typename A, int BLOCK, int PLANES>
__device__ __forceinline__ bool hand_over_and_collect(A (&acc)[PLANES], A *__restrict__ totals, uint32_t pitch, uint32_t *__restrict__ counters) {
    // thread 0 holds the workgroup's totals; returns true in ONE workgroup, with acc[] = the sum of all workgroups' totals (thread 0).
    // totals: [PLANES][pitch] workgroup totals, then [PLANES][kTicketGroup] group totals behind them.
    __shared__ int last;
    const uint32_t group = blockIdx.x / kTicketGroup, groups = (gridDim.x + kTicketGroup - 1) / kTicketGroup;
    const uint32_t first = group * kTicketGroup, members = first + kTicketGroup <= gridDim.x ? kTicketGroup : gridDim.x - first;
    A *group_totals = totals + (size_t)PLANES * pitch;
    auto publish = [&](A *where, uint32_t slot, uint32_t row_pitch, uint32_t *counter, uint32_t expected) {
        if (threadIdx.x == 0) {
#pragma unroll
            for (int p = 0; p < PLANES; ++p) __hip_atomic_store(&where[(size_t)p * row_pitch + slot], acc[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the totals have reached the memory side before the ticket is taken
            last = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == expected - 1;
        }
        __syncthreads();
        return last != 0;
    };
    auto collect = [&](const A *where, uint32_t from, uint32_t count, uint32_t row_pitch) {  // count <= kTicketGroup values per plane, in index order
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // this workgroup only: one invalidate per group, not per workgroup
#pragma unroll
        for (int p = 0; p < PLANES; ++p) {
            A v = threadIdx.x < count ? __hip_atomic_load(&where[(size_t)p * row_pitch + from + threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : A(0);
            __syncthreads();  // block_reduce's LDS slots are reused
            acc[p] = block_reduce<A, BLOCK>(v);
        }
    };
    if (!publish(totals, blockIdx.x, pitch, counters + (size_t)(1 + group) * kCounterStride, members)) return false;
    collect(totals, first, members, pitch);
    if (threadIdx.x == 0) __hip_atomic_store(counters + (size_t)(1 + group) * kCounterStride, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (groups == 1) return true;
    __syncthreads();
    if (!publish(group_totals, group, kTicketGroup, counters, groups)) return false;
    collect(group_totals, 0, groups, kTicketGroup);
    if (threadIdx.x == 0) __hip_atomic_store(counters, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return true;
}

template <typename T, typename Op, int MODE, bool KEEP, int BLOCK>
__global__ __launch_bounds__(BLOCK) void reduce_once_kernel(const T *__restrict__ a, const T *__restrict__ b, T *__restrict__ out,
                                                            size_t n_vec, size_t n, typename AccOf<T>::type *__restrict__ totals,
                                                            uint32_t *__restrict__ counter, void *__restrict__ out8, T *__restrict__ out_native, int nt) {
    typedef typename AccOf<T>::type A;
    typedef typename VecTraits<T>::vec_t V;
    constexpr int W = VecTraits<T>::width;
    constexpr int kVecPerThread = vec_per_thread(MODE);
    constexpr size_t kTile = (size_t)BLOCK * kVecPerThread;
    const V *av = reinterpret_cast<const V *>(a), *bv = reinterpret_cast<const V *>(b);
    V *ov = reinterpret_cast<V *>(out);
    OpCtx<Op> ctx;
    ctx.init();
    A acc = A(0);
    const size_t tiles = n_vec / kTile + 1;  // the last one: the partial tile and the n % W tail (maybe empty)
    for (size_t t = blockIdx.x; t < tiles; t += gridDim.x) {  // t only grows: every wave's trip count is finite
        const size_t tile0 = t * kTile + threadIdx.x;
        if (t * kTile + kTile <= n_vec) {
            V va[kVecPerThread], vb[kVecPerThread];
#pragma unroll
            for (int u = 0; u < kVecPerThread; ++u) {
                va[u] = load_stream_if(T, av + tile0 + (size_t)u * BLOCK, nt);
                if constexpr (MODE != kSum) vb[u] = load_stream_if(T, bv + tile0 + (size_t)u * BLOCK, nt);
                else vb[u] = va[u];
            }
#pragma unroll
            for (int u = 0; u < kVecPerThread; ++u) consume<T, Op, MODE, KEEP, A>(ctx, acc, va[u], vb[u], ov + tile0 + (size_t)u * BLOCK);
        } else {
            for (int u = 0; u < kVecPerThread; ++u) {
                const size_t i = tile0 + (size_t)u * BLOCK;
                if (i < n_vec) {
                    const V va = load_stream(av + i);
                    const V vb = MODE != kSum ? load_stream(bv + i) : va;
                    consume<T, Op, MODE, KEEP, A>(ctx, acc, va, vb, ov + i);
                }
            }
            if (threadIdx.x == 0) {
                for (size_t k = n_vec * W; k < n; ++k) {
                    if constexpr (MODE == kFused) {
                        const T r = Op::apply(a[k], b[k]);
                        out[k] = r;
                        acc += widen<T, A>(r);
                    } else if constexpr (MODE == kDot) add_prod<T, A>(acc, a[k], b[k]);
                    else acc += widen<T, A>(a[k]);
                }
            }
        }
    }
    A total[1] = {block_reduce<A, BLOCK>(acc)};
    if (gridDim.x > 1 && !hand_over_and_collect<A, BLOCK, 1>(total, totals, kMaxGroups, counter)) return;
    if (threadIdx.x == 0) write_result<T, MODE != kDot>(total[0], out8, out_native);
}

inline bool aligned16(const void *p) { return p == nullptr || (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// std::complex<double> dot (reference product.h:168-224): sum a[i] * b[i], unconjugated, as the
// reference's scalar tail defines it (its AVX body adds every real/imaginary product twice --
// _mm256_permute_pd(va, 0x0) duplicates lanes, rbuf[0..3] are then all summed -- which is taken as a
// bug, not as semantics).  One complex = one 16-byte vector {re, im}; separate fp64 fma chains for
// the real and imaginary sums, grid-stride, then the same wave / LDS / partials tree.
typedef double dbl2 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(kBlock) void cdot_kernel(const dbl2 *__restrict__ a, const dbl2 *__restrict__ b, size_t n,
                                                      double *__restrict__ partials, size_t blocks, int nt) {
    // one-shot like reduce_kernel: a workgroup owns kBlock * 2 consecutive elements, two per lane and operand in flight;
    // its partial sums go to partials[block] (real) and partials[blocks + block] (imaginary; `blocks` = the arrays' pitch)
    double re = 0.0, im = 0.0;
    auto acc = [&](dbl2 x, dbl2 y) {
        re = __builtin_fma(x[0], y[0], re);
        re = __builtin_fma(-x[1], y[1], re);
        im = __builtin_fma(x[0], y[1], im);
        im = __builtin_fma(x[1], y[0], im);
    };
    const size_t i0 = (size_t)blockIdx.x * (kBlock * 2) + threadIdx.x, i1 = i0 + kBlock;
    if (i1 < n) {
        typedef const VecTraits<double>::vec_t *vp;
        dbl2 x0, y0, x1, y1;
        if (nt & kLoadNt) {  // one branch around the group (ops.hip.h)
            x0 = load_stream_as(double, (vp)(a + i0), true), y0 = load_stream_as(double, (vp)(b + i0), true);
            x1 = load_stream_as(double, (vp)(a + i1), true), y1 = load_stream_as(double, (vp)(b + i1), true);
        } else {
            x0 = load_stream_as(double, (vp)(a + i0), false), y0 = load_stream_as(double, (vp)(b + i0), false);
            x1 = load_stream_as(double, (vp)(a + i1), false), y1 = load_stream_as(double, (vp)(b + i1), false);
        }
        acc(x0, y0);
        acc(x1, y1);
    } else if (i0 < n) {
        acc(load_stream(a + i0), load_stream(b + i0));
    }
    __shared__ double lds[2][kBlock / 64];
    re = wave_reduce(re);
    im = wave_reduce(im);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == kWaveTotalLane) { lds[0][wave] = re; lds[1][wave] = im; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double r = 0.0, m = 0.0;
        for (int w = 0; w < kBlock / 64; ++w) { r += lds[0][w]; m += lds[1][w]; }
        partials[blockIdx.x] = r;
        partials[blocks + blockIdx.x] = m;
    }
}
// The generic dot_product<T> with T = std::complex<float> (product.h:8-20: `sum += a[i] * b[i]` in complex<float>): one 16-byte
// vector = two {re, im} pairs; the products of two floats are exact in fp64, so each part is an fp64 fma chain -- more
// accurate than the reference's sequential float sums, like the f32 dot.  Same one-shot shape and partial layout as cdot_kernel.
__global__ __launch_bounds__(kBlock) void cdot32_kernel(const float *__restrict__ a, const float *__restrict__ b, size_t n,
                                                        double *__restrict__ partials, size_t blocks, int nt) {
    typedef VecTraits<float>::vec_t V;
    const V *av = reinterpret_cast<const V *>(a), *bv = reinterpret_cast<const V *>(b);
    const size_t n_vec = n / 2;  // two complex numbers per vector
    double re = 0.0, im = 0.0;
    auto acc1 = [&](float ar, float ai, float br, float bi) {
        re = __builtin_fma((double)ar, (double)br, re);
        re = __builtin_fma(-(double)ai, (double)bi, re);
        im = __builtin_fma((double)ar, (double)bi, im);
        im = __builtin_fma((double)ai, (double)br, im);
    };
    auto acc = [&](V x, V y) { acc1(x[0], x[1], y[0], y[1]); acc1(x[2], x[3], y[2], y[3]); };
    const size_t i0 = (size_t)blockIdx.x * (kBlock * 2) + threadIdx.x, i1 = i0 + kBlock;
    if (i1 < n_vec) {
        V x0, y0, x1, y1;
        if (nt & kLoadNt) {
            x0 = load_stream_as(float, av + i0, true), y0 = load_stream_as(float, bv + i0, true);
            x1 = load_stream_as(float, av + i1, true), y1 = load_stream_as(float, bv + i1, true);
        } else {
            x0 = load_stream_as(float, av + i0, false), y0 = load_stream_as(float, bv + i0, false);
            x1 = load_stream_as(float, av + i1, false), y1 = load_stream_as(float, bv + i1, false);
        }
        acc(x0, y0);
        acc(x1, y1);
    } else if (i0 < n_vec) {
        acc(load_stream(av + i0), load_stream(bv + i0));
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0 && (n & 1)) acc1(a[2 * (n - 1)], a[2 * (n - 1) + 1], b[2 * (n - 1)], b[2 * (n - 1) + 1]);  // odd n: the last pair
    __shared__ double lds[2][kBlock / 64];
    re = wave_reduce(re);
    im = wave_reduce(im);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == kWaveTotalLane) { lds[0][wave] = re; lds[1][wave] = im; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double r = 0.0, m = 0.0;
        for (int w = 0; w < kBlock / 64; ++w) { r += lds[0][w]; m += lds[1][w]; }
        partials[blockIdx.x] = r;
        partials[blocks + blockIdx.x] = m;
    }
}
// The complex dots in the one-launch form (see hand_over_and_collect): F32 = false: n {re, im} pairs of doubles, one per
// 16-byte vector; F32 = true: pairs of floats, two per vector (+ the odd last pair).
template <bool F32>
__global__ __launch_bounds__(kBlock) void cdot_once_kernel(const void *__restrict__ a_, const void *__restrict__ b_, size_t n, double *__restrict__ totals,
                                                           uint32_t *__restrict__ counter, double *__restrict__ out2, int nt) {
    typedef typename std::conditional<F32, float, double>::type E;
    typedef typename VecTraits<E>::vec_t V;
    typedef typename VecTraits<E>::full_t F;
    const V *av = static_cast<const V *>(a_), *bv = static_cast<const V *>(b_);
    const size_t n_vec = F32 ? n / 2 : n;
    double re = 0.0, im = 0.0;
    auto acc1 = [&](double ar, double ai, double br, double bi) {
        re = __builtin_fma(ar, br, re);
        re = __builtin_fma(-ai, bi, re);
        im = __builtin_fma(ar, bi, im);
        im = __builtin_fma(ai, br, im);
    };
    auto acc = [&](F x, F y) {
        acc1(x[0], x[1], y[0], y[1]);
        if constexpr (F32) acc1(x[2], x[3], y[2], y[3]);
    };
    constexpr size_t kTile = (size_t)kBlock * 2;
    const size_t tiles = n_vec / kTile + 1;
    for (size_t t = blockIdx.x; t < tiles; t += gridDim.x) {  // t only grows: every wave's trip count is finite
        const size_t i0 = t * kTile + threadIdx.x, i1 = i0 + kBlock;
        if (t * kTile + kTile <= n_vec) {
            F x0, y0, x1, y1;
            if (nt & kLoadNt) {
                x0 = load_stream_as(E, av + i0, true), y0 = load_stream_as(E, bv + i0, true);
                x1 = load_stream_as(E, av + i1, true), y1 = load_stream_as(E, bv + i1, true);
            } else {
                x0 = load_stream_as(E, av + i0, false), y0 = load_stream_as(E, bv + i0, false);
                x1 = load_stream_as(E, av + i1, false), y1 = load_stream_as(E, bv + i1, false);
            }
            acc(x0, y0);
            acc(x1, y1);
        } else {
            if (i0 < n_vec) acc(load_stream(av + i0), load_stream(bv + i0));
            if (i1 < n_vec) acc(load_stream(av + i1), load_stream(bv + i1));
            if (F32 && threadIdx.x == 0 && (n & 1)) {  // odd n: the last pair
                const float *a = static_cast<const float *>(a_), *b = static_cast<const float *>(b_);
                acc1(a[2 * (n - 1)], a[2 * (n - 1) + 1], b[2 * (n - 1)], b[2 * (n - 1) + 1]);
            }
        }
    }
    __shared__ double lds[2][kBlock / 64];
    re = wave_reduce(re);
    im = wave_reduce(im);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == kWaveTotalLane) { lds[0][wave] = re; lds[1][wave] = im; }
    __syncthreads();
    double total[2] = {0.0, 0.0};
    if (threadIdx.x == 0)
        for (int w = 0; w < kBlock / 64; ++w) { total[0] += lds[0][w]; total[1] += lds[1][w]; }
    if (gridDim.x > 1 && !hand_over_and_collect<double, kBlock, 2>(total, totals, kMaxGroups, counter)) return;
    if (threadIdx.x == 0) { out2[0] = total[0]; out2[1] = total[1]; }
}

// One launch for a complex dot of at most kOneLaunchTiles tiles; false: the caller goes on with the two-launch form.
template <bool F32>
int try_cdot_once(const void *a, const void *b, size_t n, size_t blocks, size_t bytes, double *out2_dev, hipStream_t s, bool *done) {
    static const size_t max_tiles = [] { const char *e = getenv("SMHIP_REDUCE_ONE_TILES"); return e && *e ? (size_t)atol(e) : kOneLaunchTiles; }();
    *done = false;
    if (blocks > max_tiles) return SMHIP_OK;
    const unsigned grid = (unsigned)(blocks < kMaxGroups ? blocks : kMaxGroups);
    double *scratch = nullptr;
    ScratchLease lease;
    uint32_t *counter = nullptr;
    if (grid > 1) {
        if (int rc = lease.take(2 * (kMaxGroups + kTicketGroup), &scratch)) return rc;
        if (int rc = reduce_counters(s, &counter)) return rc;
    }
    hipLaunchKernelGGL((cdot_once_kernel<F32>), dim3(grid), dim3(kBlock), 0, s, a, b, n, scratch, counter, out2_dev, stream_policy({{a, bytes}, {b, bytes}}, {nullptr, 0}));
    if (hipError_t e = hipGetLastError(); e != hipSuccess) {
        if (grid > 1) reduce_counters_suspect(s);
        return fail(SMHIP_ERR_HIP, "launch complex dot: %s", hipGetErrorString(e));
    }
    *done = true;
    return SMHIP_OK;
}

// Queues finish_kernel over `blocks` partials (blocks >= 1); `partials` has room for the group totals behind them
// (blocks / kGroupTarget + 2 more accumulators are enough).
template <typename T, bool AS_DOUBLE>
int launch_finish(typename AccOf<T>::type *partials, size_t blocks, void *out8, T *out_native, hipStream_t s, uint32_t planes = 1,
                  size_t pitch = 0) {
    typedef typename AccOf<T>::type A;
    uint32_t groups = (uint32_t)((blocks + kGroupTarget - 1) / kGroupTarget);
    if (groups > kMaxGroups) groups = kMaxGroups;
    const uint32_t gsize = (uint32_t)((blocks + groups - 1) / groups);
    groups = (uint32_t)((blocks + gsize - 1) / gsize);
    Finish<A> fin{partials + blocks, nullptr, gsize, groups, (uint32_t)pitch};
    if (groups > 1)
        if (int rc = reduce_counters(s, &fin.counters)) return rc;
    hipLaunchKernelGGL((finish_kernel<T, AS_DOUBLE>), dim3(groups, planes), dim3(kBlock), 0, s, partials, (uint32_t)blocks, fin, out8, out_native);
    if (hipError_t e = hipGetLastError(); e != hipSuccess) {
        if (groups > 1) reduce_counters_suspect(s);
        return fail(SMHIP_ERR_HIP, "launch reduce finish: %s", hipGetErrorString(e));
    }
    return SMHIP_OK;
}

template <typename T, typename Op, int MODE>
int run_reduce(const void *a_, const void *b_, void *out_, size_t n, void *out8, void *out_native, hipStream_t s) {
    typedef typename AccOf<T>::type A;
    constexpr int W = VecTraits<T>::width;
    const T *a = static_cast<const T *>(a_), *b = static_cast<const T *>(b_);
    T *out = static_cast<T *>(out_);
    const size_t n_vec = n / W;
    constexpr int BLOCK = block_of(MODE);
    const size_t tile = (size_t)BLOCK * vec_per_thread(MODE);
    size_t blocks;
    blocks = n_vec / tile + 1;  // the last workgroup takes the partial tile and the n % W tail (maybe empty)
    if (blocks > 0x7fffffffu) return fail(SMHIP_ERR_UNSUPPORTED, "reduction too large (%zu workgroups)", blocks);
    double *scratch = nullptr;
    ScratchLease lease;
    if (blocks > 1)
        if (int rc = lease.take(blocks + blocks / kGroupTarget + 2 > kMaxGroups + kTicketGroup ? blocks + blocks / kGroupTarget + 2 : kMaxGroups + kTicketGroup, &scratch)) return rc;
    A *partials = reinterpret_cast<A *>(scratch);
    static const size_t max_tiles = [] { const char *e = getenv("SMHIP_REDUCE_ONE_TILES"); return e && *e ? (size_t)atol(e) : kOneLaunchTiles; }();
    if (reduce_one_launch() && blocks <= max_tiles && MODE != kFused) {
        const int pol1 = MODE == kSum ? stream_policy({{a, n * sizeof(T)}}, {nullptr, 0})
                                      : stream_policy({{a, n * sizeof(T)}, {b, n * sizeof(T)}}, {MODE == kFused ? out : nullptr, MODE == kFused ? n * sizeof(T) : 0});
        const unsigned grid = (unsigned)(blocks < kMaxGroups ? blocks : kMaxGroups);
        uint32_t *counter = nullptr;
        if (grid > 1)
            if (int rc = reduce_counters(s, &counter)) return rc;
        if (MODE == kFused && (pol1 & kStoreKeep))
            hipLaunchKernelGGL((reduce_once_kernel<T, Op, MODE, MODE == kFused, BLOCK>), dim3(grid), dim3(BLOCK), 0, s, a, b, out, n_vec, n, partials, counter, out8,
                               static_cast<T *>(out_native), pol1);
        else
            hipLaunchKernelGGL((reduce_once_kernel<T, Op, MODE, false, BLOCK>), dim3(grid), dim3(BLOCK), 0, s, a, b, out, n_vec, n, partials, counter, out8,
                               static_cast<T *>(out_native), pol1);
        if (hipError_t e = hipGetLastError(); e != hipSuccess) {
            if (grid > 1) reduce_counters_suspect(s);
            return fail(SMHIP_ERR_HIP, "launch reduce: %s", hipGetErrorString(e));
        }
        return SMHIP_OK;
    }
    const int pol = MODE == kSum ? stream_policy({{a, n * sizeof(T)}}, {nullptr, 0})
                                 : stream_policy({{a, n * sizeof(T)}, {b, n * sizeof(T)}}, {MODE == kFused ? out : nullptr, MODE == kFused ? n * sizeof(T) : 0});
    const bool keep = MODE == kFused && (pol & kStoreKeep);
    // Very large operands go out as several launches, like the streaming kernels (contiguous.hip: a launch of hundreds of
    // thousands of workgroups loses 3-6 % to its own length): pieces of whole tiles, each writing its workgroups' partials
    // behind the previous piece's; ONE finishing launch adds them all.  The partials and their order are the same as for a
    // single launch, so the bits do not depend on the piece size.
    const size_t piece = piece_for(n_vec, MODE == kSum ? 1 : MODE == kDot ? 2 : 3);
    const size_t piece_tiles = piece ? (piece / tile ? piece / tile : 1) : blocks;
    const size_t full_tiles = blocks - 1;  // the last workgroup of the whole array is the partial tile + tail: it rides with the last piece
    for (size_t b0 = 0; b0 < blocks;) {
        const bool last = b0 + piece_tiles >= full_tiles;
        const size_t nb = last ? blocks - b0 : piece_tiles;
        const size_t v0 = b0 * tile;                                     // first vector of this piece
        const size_t nv = last ? n_vec - v0 : nb * tile;                  // its vectors (the last piece: the partial tile too)
        const size_t ne = last ? n - v0 * W : nv * W;                     // its elements (the last piece: the n % W tail too)
        const T *pa = a + v0 * W, *pb = b ? b + v0 * W : b;
        T *po = out ? out + v0 * W : out;
        A *pp = partials ? partials + b0 : partials;
        if (keep)
            hipLaunchKernelGGL((reduce_kernel<T, Op, MODE, MODE == kFused, BLOCK>), dim3((unsigned)nb), dim3(BLOCK), 0, s, pa, pb, po, nv, ne, pp, out8,
                               static_cast<T *>(out_native), pol, blocks == 1 ? 1 : 0);
        else
            hipLaunchKernelGGL((reduce_kernel<T, Op, MODE, false, BLOCK>), dim3((unsigned)nb), dim3(BLOCK), 0, s, pa, pb, po, nv, ne, pp, out8,
                               static_cast<T *>(out_native), pol, blocks == 1 ? 1 : 0);
        b0 += nb;
    }
    SMHIP_LAUNCH_CHECK("reduce");
    if (blocks == 1) return SMHIP_OK;  // the single workgroup wrote the result itself
    return launch_finish<T, MODE != kDot>(partials, blocks, out8, static_cast<T *>(out_native), s);
}

struct CounterBuf { uint32_t *p; bool suspect; };
std::mutex g_counter_mutex;
std::map<std::pair<int, hipStream_t>, CounterBuf> g_counter_bufs;
constexpr size_t kCounterBytes = 4096;  // finish_kernel: two counters (the complex dot's planes) at the front; the one-launch form: 1 + kTicketGroup counters, 64 bytes apart

int reduce_counters(hipStream_t s, uint32_t **out) {
    const std::pair<int, hipStream_t> key(current_device(), s);
    std::lock_guard<std::mutex> lock(g_counter_mutex);
    auto it = g_counter_bufs.find(key);
    if (it == g_counter_bufs.end()) {
        void *p = nullptr;
        SMHIP_TRY(hipMalloc(&p, kCounterBytes));
        SMHIP_TRY(hipMemset(p, 0, kCounterBytes));
        SMHIP_TRY(hipDeviceSynchronize());
        it = g_counter_bufs.emplace(key, CounterBuf{static_cast<uint32_t *>(p), false}).first;
    }
    if (it->second.suspect) {
        SMHIP_TRY(hipMemsetAsync(it->second.p, 0, kCounterBytes, s));
        it->second.suspect = false;
    }
    *out = it->second.p;
    return SMHIP_OK;
}

void reduce_counters_suspect(hipStream_t s) {
    const std::pair<int, hipStream_t> key(current_device(), s);
    std::lock_guard<std::mutex> lock(g_counter_mutex);
    auto it = g_counter_bufs.find(key);
    if (it != g_counter_bufs.end()) it->second.suspect = true;
}

}  // namespace

int launch_sum(int dtype, const void *a, size_t n, double *out_dev, hipStream_t s) {
    switch (dtype) {
        case SMHIP_F32: return run_reduce<float, AddOp<float>, kSum>(a, nullptr, nullptr, n, out_dev, nullptr, s);
        case SMHIP_F64: return run_reduce<double, AddOp<double>, kSum>(a, nullptr, nullptr, n, out_dev, nullptr, s);
        case SMHIP_I32: return run_reduce<int32_t, AddOp<int32_t>, kSum>(a, nullptr, nullptr, n, out_dev, nullptr, s);
        case SMHIP_I64: return run_reduce<int64_t, AddOp<int64_t>, kSum>(a, nullptr, nullptr, n, out_dev, nullptr, s);
    }
    return fail(SMHIP_ERR_INVALID, "sum: bad dtype %d", dtype);
}

int launch_dot(int dtype, const void *a, const void *b, size_t n, double *out8_dev, void *out_native_dev, hipStream_t s) {
    switch (dtype) {
        case SMHIP_F32: return run_reduce<float, AddOp<float>, kDot>(a, b, nullptr, n, out8_dev, out_native_dev, s);
        case SMHIP_F64: return run_reduce<double, AddOp<double>, kDot>(a, b, nullptr, n, out8_dev, out_native_dev, s);
        case SMHIP_I32: return run_reduce<int32_t, AddOp<int32_t>, kDot>(a, b, nullptr, n, out8_dev, out_native_dev, s);
        case SMHIP_I64: return run_reduce<int64_t, AddOp<int64_t>, kDot>(a, b, nullptr, n, out8_dev, out_native_dev, s);
        // the generic dot_product<T>'s other integer element types (product.h:8-20)
        case SMHIP_I8: return run_reduce<int8_t, AddOp<int8_t>, kDot>(a, b, nullptr, n, out8_dev, out_native_dev, s);
        case SMHIP_U8: return run_reduce<uint8_t, AddOp<uint8_t>, kDot>(a, b, nullptr, n, out8_dev, out_native_dev, s);
        case SMHIP_I16: return run_reduce<int16_t, AddOp<int16_t>, kDot>(a, b, nullptr, n, out8_dev, out_native_dev, s);
        case SMHIP_U16: return run_reduce<uint16_t, AddOp<uint16_t>, kDot>(a, b, nullptr, n, out8_dev, out_native_dev, s);
        case SMHIP_U32: return run_reduce<uint32_t, AddOp<uint32_t>, kDot>(a, b, nullptr, n, out8_dev, out_native_dev, s);
        case SMHIP_U64: return run_reduce<uint64_t, AddOp<uint64_t>, kDot>(a, b, nullptr, n, out8_dev, out_native_dev, s);
    }
    return fail(SMHIP_ERR_INVALID, "dot: bad dtype %d", dtype);
}

int launch_cdot(const void *a, const void *b, size_t n, double *out2_dev, hipStream_t s) {
    const size_t blocks = n / (kBlock * 2) + 1;
    if (blocks > 0x7fffffffu) return fail(SMHIP_ERR_UNSUPPORTED, "complex dot too large (%zu workgroups)", blocks);
    if (reduce_one_launch()) {
        bool done;
        if (int rc = try_cdot_once<false>(a, b, n, blocks, n * sizeof(dbl2), out2_dev, s, &done)) return rc;
        if (done) return SMHIP_OK;
    }
    // two partial arrays (real, imaginary), each with room for its group totals behind it
    const size_t span = blocks + blocks / kGroupTarget + 2;
    double *scratch;
    ScratchLease lease;
    if (int rc = lease.take(2 * span, &scratch)) return rc;
    hipLaunchKernelGGL(cdot_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, s, static_cast<const dbl2 *>(a), static_cast<const dbl2 *>(b), n, scratch, span,
                       stream_policy({{a, n * sizeof(dbl2)}, {b, n * sizeof(dbl2)}}, {nullptr, 0}));
    SMHIP_LAUNCH_CHECK("cdot");
    // one finishing launch for both sums (plane 0: real -> out2_dev[0], plane 1: imaginary -> out2_dev[1])
    return launch_finish<double, true>(scratch, blocks, out2_dev, static_cast<double *>(nullptr), s, 2, span);
}

int launch_cdot32(const void *a, const void *b, size_t n, double *out2_dev, hipStream_t s) {
    const size_t blocks = (n / 2) / (kBlock * 2) + 1;
    if (blocks > 0x7fffffffu) return fail(SMHIP_ERR_UNSUPPORTED, "complex dot too large (%zu workgroups)", blocks);
    if (reduce_one_launch()) {
        bool done;
        if (int rc = try_cdot_once<true>(a, b, n, blocks, n * 8, out2_dev, s, &done)) return rc;
        if (done) return SMHIP_OK;
    }
    const size_t span = blocks + blocks / kGroupTarget + 2;
    double *scratch;
    ScratchLease lease;
    if (int rc = lease.take(2 * span, &scratch)) return rc;
    hipLaunchKernelGGL(cdot32_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, s, static_cast<const float *>(a), static_cast<const float *>(b), n, scratch, span,
                       stream_policy({{a, n * 8}, {b, n * 8}}, {nullptr, 0}));
    SMHIP_LAUNCH_CHECK("cdot32");
    return launch_finish<double, true>(scratch, blocks, out2_dev, static_cast<double *>(nullptr), s, 2, span);
}

int launch_contiguous_sum(int op, int dtype, const void *a, const void *b, void *out, size_t n, double *sum_dev, hipStream_t s) {
#define SMHIP_DISPATCH_OP(T)                                                                                          \
    switch (op) {                                                                                                     \
        case SMHIP_OP_ADD: return run_reduce<T, AddOp<T>, kFused>(a, b, out, n, sum_dev, nullptr, s);                 \
        case SMHIP_OP_SUB: return run_reduce<T, SubtractOp<T>, kFused>(a, b, out, n, sum_dev, nullptr, s);            \
        case SMHIP_OP_MUL: return run_reduce<T, MultiplyOp<T>, kFused>(a, b, out, n, sum_dev, nullptr, s);            \
        case SMHIP_OP_DIV: return run_reduce<T, DivideOp<T>, kFused>(a, b, out, n, sum_dev, nullptr, s);              \
        case SMHIP_OP_POW: return run_reduce<T, PowOp<T>, kFused>(a, b, out, n, sum_dev, nullptr, s);                 \
        case SMHIP_OP_LEFT: return run_reduce<T, LeftOp<T>, kFused>(a, b, out, n, sum_dev, nullptr, s);                 \
    }                                                                                                                 \
    break;
    switch (dtype) {
        case SMHIP_F32: SMHIP_DISPATCH_OP(float)
        case SMHIP_F64: SMHIP_DISPATCH_OP(double)
        case SMHIP_I32: SMHIP_DISPATCH_OP(int32_t)
        case SMHIP_I64: SMHIP_DISPATCH_OP(int64_t)
    }
#undef SMHIP_DISPATCH_OP
    return fail(SMHIP_ERR_INVALID, "contiguous_sum: bad op %d / dtype %d", op, dtype);
}

// For reductions whose first pass is compiled at run time (jit.hip: fused expression + sum): `partials` holds one
// accumulator per workgroup of that pass (double for float types, uint64 for integer types, as AccOf<T>), with room
// for blocks / kGroupTarget + 2 more behind them; this queues the finishing launch into *out8 (fp64).
int reduce_finish(int dtype, void *partials, size_t blocks, double *out8, hipStream_t s) {
    if (blocks < 1) return fail(SMHIP_ERR_INVALID, "reduce_finish: no partials");
    switch (dtype) {
        case SMHIP_F32: return launch_finish<float, true>(static_cast<double *>(partials), blocks, out8, static_cast<float *>(nullptr), s);
        case SMHIP_F64: return launch_finish<double, true>(static_cast<double *>(partials), blocks, out8, static_cast<double *>(nullptr), s);
        case SMHIP_I32: return launch_finish<int32_t, true>(static_cast<uint64_t *>(partials), blocks, out8, static_cast<int32_t *>(nullptr), s);
        case SMHIP_I64: return launch_finish<int64_t, true>(static_cast<uint64_t *>(partials), blocks, out8, static_cast<int64_t *>(nullptr), s);
    }
    return fail(SMHIP_ERR_INVALID, "reduce_finish: bad dtype %d", dtype);
}

}  // namespace smhip
