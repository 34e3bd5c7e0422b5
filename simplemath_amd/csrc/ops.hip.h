// ops.hip.h -- device-side Op policies and vector traits for gfx950.
//
// Device mirror of the reference's plugin contract: where the reference has
//   template<class T> struct XOp { static T apply(const T&, const T&);
//                                  template<class R> static R apply_simd(const R&, const R&); };
// (include/math/add.h:5-14 and siblings), the device side has
//   template<class T> struct XOp { static __device__ T apply(T, T); };
// and `apply_vec` below plays apply_simd's role on a 16-byte register group
// (float4 / double2 / int4 / long2 instead of __m256 -- SimdTraits<T>,
// include/math/helpers.h:12-119).  One wave = 64 lanes x 16 B = 1 KiB per
// memory instruction.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "sm_pow.h"
#include "sm_pow64.h"

namespace smhip {
namespace dev {

// ----------------------------------------------------------------- vectors
// vec_t is 16 bytes wide but only ELEMENT-aligned: gfx950 takes a global_load/store_dwordx4 at any 4-byte
// address (the HSA ABI runs the memory pipeline in unaligned mode), and a stream of such accesses that is
// shifted off the 16-byte grid runs within 2 % of an aligned one (tools/sweep_unaligned.hip: 6.1-6.4 vs
// 6.3 TB/s; one element per lane: 5.3).  So views that start mid-row, odd row pitches and sliced operands all
// take the same vector kernels -- there is no per-element fallback for alignment.
// Bits of a launch's stream-policy word (host side: internal.h, stream_policy()).
constexpr int kLoadNt = 1, kStoreKeep = 2;
template <typename T> struct VecTraits;
#define SMHIP_VEC(T, N)                                                       \
    template <> struct VecTraits<T> {                                         \
        typedef T full_t __attribute__((ext_vector_type(N)));                \
        typedef full_t vec_t __attribute__((aligned(sizeof(T))));            \
        typedef T half_full_t __attribute__((ext_vector_type(N / 2)));       \
        typedef half_full_t half_t __attribute__((aligned(sizeof(T))));      \
        static constexpr int width = N;                                       \
        static __device__ __forceinline__ full_t join(half_full_t lo, half_full_t hi) { return __builtin_shufflevector(lo, hi, SMHIP_JOIN_##N); } \
    };
#define SMHIP_JOIN_16 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15
#define SMHIP_JOIN_8 0, 1, 2, 3, 4, 5, 6, 7
#define SMHIP_JOIN_4 0, 1, 2, 3
#define SMHIP_JOIN_2 0, 1
SMHIP_VEC(float, 4)
SMHIP_VEC(int32_t, 4)
SMHIP_VEC(double, 2)
SMHIP_VEC(int64_t, 2)
// the generic dot_product<T>'s other integer element types (reduce.hip): 16 bytes per lane all the same
SMHIP_VEC(int8_t, 16)
SMHIP_VEC(uint8_t, 16)
SMHIP_VEC(int16_t, 8)
SMHIP_VEC(uint16_t, 8)
SMHIP_VEC(uint32_t, 4)
SMHIP_VEC(uint64_t, 2)
#undef SMHIP_VEC

// Streaming accesses carry the `nt` (non-temporal) policy: every byte of the
// contiguous path is touched once, and keeping it out of L2's replacement
// order is worth ~8 % on the 2R+1W stream (profiles/r01_sweep_stream_add.txt).
// (Macros, not function templates: a template parameter would strip vec_t's reduced alignment.)
#ifdef SMHIP_CACHED_STREAMS  // experiment switch (tools/chain_exp.py): plain cached accesses everywhere
#define load_stream(ptr) (*(ptr))
#define store_stream(ptr, ...) (*(ptr) = (__VA_ARGS__))
#else
#define load_stream(ptr) __builtin_nontemporal_load(ptr)
#define store_stream(ptr, ...) __builtin_nontemporal_store((__VA_ARGS__), (ptr))
#endif
// The READ side's policy as a launch-time (wave-uniform) choice.  Whether a read stream wants `nt` depends on its size
// (tools/sweep_scalar2.hip, profiles/r02_sweep_scalar2.txt): operands that together fit the 256 MiB Infinity Cache are
// served faster through plain loads -- 1R+1W at 256 MiB: 92.9 % of HBM peak against 83.5 % with nt, 2R+1W at 2 x 64 MiB:
// 88 % against 79.5 % -- and larger ones faster with nt (1 GiB: 79.3 % plain, 82.0 % nt; 2R+1W at 2 x 256 MiB: 74 %
// against 82 %).  The host passes nt = (bytes the launch reads > kInfinityCacheBytes).
// The plain arm is written as two half-width loads on purpose: as `if (nt) nt_load(p) else *p` the two arms are one load
// with different metadata to the optimiser, which merges them and drops the hint; two halves are a different
// instruction sequence until the load/store vectoriser (which runs after the CFG clean-ups) joins them back into one
// global_load_dwordx4 -- so the ISA has exactly `global_load_dwordx4 ... nt` in one arm and `global_load_dwordx4` in the other.
// load_stream_as(T, ptr, NT): the same two forms with a COMPILE-TIME choice -- for a lane that issues several loads, branch
// once around the whole group (`if (nt) { ...as(.., true) } else { ...as(.., false) }`): with one branch per load the
// compiler parks an s_waitcnt vmcnt(0) at every join and the loads go out one at a time (strided rows: 111 -> 134 us).
#define load_stream_as(T, ptr, NT)                                                                 \
    ({                                                                                             \
        typedef typename ::smhip::dev::VecTraits<T> smhip_tr_;                                     \
        const typename smhip_tr_::vec_t *smhip_p_ = (ptr);                                         \
        typename smhip_tr_::full_t smhip_v_;                                                       \
        if constexpr (NT) {                                                                        \
            smhip_v_ = __builtin_nontemporal_load(smhip_p_);                                       \
        } else {                                                                                   \
            const typename smhip_tr_::half_t *smhip_h_ = reinterpret_cast<const typename smhip_tr_::half_t *>(smhip_p_); \
            const typename smhip_tr_::half_full_t smhip_lo_ = smhip_h_[0], smhip_hi_ = smhip_h_[1]; \
            smhip_v_ = smhip_tr_::join(smhip_lo_, smhip_hi_);                                      \
        }                                                                                          \
        smhip_v_;                                                                                  \
    })
#define load_stream_if(T, ptr, nt)                                                                 \
    ({                                                                                             \
        typedef typename ::smhip::dev::VecTraits<T> smhip_tr_;                                     \
        const typename smhip_tr_::vec_t *smhip_p_ = (ptr);                                         \
        typename smhip_tr_::full_t smhip_v_;                                                       \
        if ((nt) & ::smhip::dev::kLoadNt) {                                                        \
            smhip_v_ = __builtin_nontemporal_load(smhip_p_);                                       \
        } else {                                                                                   \
            const typename smhip_tr_::half_t *smhip_h_ = reinterpret_cast<const typename smhip_tr_::half_t *>(smhip_p_); \
            const typename smhip_tr_::half_full_t smhip_lo_ = smhip_h_[0], smhip_hi_ = smhip_h_[1]; \
            smhip_v_ = smhip_tr_::join(smhip_lo_, smhip_hi_);                                      \
        }                                                                                          \
        smhip_v_;                                                                                  \
    })

// The WRITE side's policy, the other bit of the same launch-time word (internal.h: stream_policy).  Results are written
// non-temporally unless the launch's whole footprint -- reads and writes -- fits the Infinity Cache: then they are stored
// so that the NEXT operator's plain loads find them there.  tools/sweep_chain.hip -> profiles/r02_sweep_chain.txt, 1R+1W
// ping-pong (each launch reads what the previous one wrote), 64 / 128 MiB per array: nt stores 72 / 76 % of HBM peak, plain
// stores 86 / 90 % -- the rate bench.py's setting (every launch re-reads the SAME operands) shows with either.
// Which store: `sc1` (tools/sweep_store_flavours.hip, tools/sweep_store_sc1.hip -> profiles/r02_sweep_store_sc1.txt).
// It keeps the chain rate of a plain store without a plain store's cost in the repeated-operand setting -- 1R+1W at
// 32 MiB: same / chain 109 / 81 % against 79 / 78 % (plain) and 82 / 64 % (nt); 2R+1W at 16 MiB: 121 / 92 % against
// 74 / 74 % and 95 / 71 %.  The compiler has no spelling for it on a 16-byte vector (nontemporal = nt, a system-scope
// atomic store is 8 bytes at most), hence the asm; nothing else reads the destination, so it needs no memory clobber, and
// a wave may end with the store in flight like any other.  The `s_nop 1` behind it is NOT optional: gfx940-class hardware
// needs two wait states between a VMEM store of more than 8 bytes and a VALU instruction that overwrites the VGPRs holding
// its data; the compiler inserts them behind stores it can see and cannot see into an asm.  Without them the next vector's
// multiply overwrote a component the store had not read yet (found by tests/fuzz_policy.py in the column form of the flat
// tile kernel, where two results are computed into the same registers back to back: every fourth element wrong).
#define store_stream_as(T, ptr, value, NT)                                                         \
    do {                                                                                           \
        typedef typename ::smhip::dev::VecTraits<T> smhip_tr_;                                     \
        typename smhip_tr_::vec_t *smhip_q_ = (ptr);                                               \
        const typename smhip_tr_::full_t smhip_w_ = (value);                                       \
        if constexpr (NT) {                                                                        \
            __builtin_nontemporal_store(smhip_w_, smhip_q_);                                       \
        } else {                                                                                   \
            asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(smhip_q_), "v"(smhip_w_));   \
        }                                                                                          \
    } while (0)
#ifdef SMHIP_STORES_ALWAYS_NT  // experiment switch: no branch, no plain arm
#define store_stream_if(T, ptr, value, pol) store_stream_as(T, ptr, value, true)
#else
#define store_stream_if(T, ptr, value, pol)                                                        \
    do {                                                                                           \
        if ((pol) & ::smhip::dev::kStoreKeep) store_stream_as(T, ptr, value, false);              \
        else store_stream_as(T, ptr, value, true);                                                 \
    } while (0)
#endif

// -------------------------------------------------------------- Op policies
// f32/f64: one correctly rounded IEEE operation each (add.h:18-59 etc.);
// kernels are built with -ffp-contract=off, denormals on, and hipcc's default
// correctly-rounded f32 division, so results are bit-identical to the AVX2 path.
// i32/i64 + - * wrap like _mm256_{add,sub,mullo}_epi32 (add.h:64-82,
// subtract.h:65-83, multiply.h:68-86).
template <typename T> struct UInt;
template <> struct UInt<int32_t> { typedef uint32_t type; };
template <> struct UInt<int64_t> { typedef uint64_t type; };

template <typename T> struct AddOp { static __device__ __forceinline__ T apply(T a, T b) { return a + b; } };
template <typename T> struct SubtractOp { static __device__ __forceinline__ T apply(T a, T b) { return a - b; } };
template <typename T> struct MultiplyOp { static __device__ __forceinline__ T apply(T a, T b) { return a * b; } };
template <typename T> struct DivideOp { static __device__ __forceinline__ T apply(T a, T b) { return a / b; } };
template <typename T> struct PowOp;
// out = a: dense copy of a strided / broadcast view (SMHIP_OP_LEFT)
template <typename T> struct LeftOp { static __device__ __forceinline__ T apply(T a, T) { return a; } };

#define SMHIP_INT_OPS(T)                                                                             \
    template <> struct AddOp<T> { static __device__ __forceinline__ T apply(T a, T b) {             \
        typedef UInt<T>::type U; return (T)((U)a + (U)b); } };                                       \
    template <> struct SubtractOp<T> { static __device__ __forceinline__ T apply(T a, T b) {        \
        typedef UInt<T>::type U; return (T)((U)a - (U)b); } };                                       \
    template <> struct MultiplyOp<T> { static __device__ __forceinline__ T apply(T a, T b) {        \
        typedef UInt<T>::type U; return (T)((U)a * (U)b); } };
SMHIP_INT_OPS(int32_t)
SMHIP_INT_OPS(int64_t)
#undef SMHIP_INT_OPS

// division.h:67-70 (C `/`: truncation toward zero).  The reference traps on
// x/0 and INT_MIN/-1; a kernel cannot, so both are defined: x/0 = 0,
// INT_MIN/-1 = INT_MIN (DESIGN.md "defined where the reference traps").
template <> struct DivideOp<int32_t> {
    static __device__ __forceinline__ int32_t apply(int32_t a, int32_t b) {
        if (b == 0) return 0;
        if (b == -1) return (int32_t)(0u - (uint32_t)a);
        return a / b;
    }
};
template <> struct DivideOp<int64_t> {
    static __device__ __forceinline__ int64_t apply(int64_t a, int64_t b) {
        if (b == 0) return 0;
        if (b == -1) return (int64_t)(0ull - (uint64_t)a);
        return a / b;
    }
};

// __sm256_powi_ps, include/math/simd/crafted_pow.h:54-103, one lane: square
// and multiply on |e| with wrapping products, then the sign-of-exponent fix-ups.
// With a wave-uniform exponent (sm::pow(arr, scalar)) the loop trip count is
// uniform, so there is no divergence.
template <typename T> __device__ __forceinline__ T powi(T base, T exponent) {
    typedef typename UInt<T>::type U;
    U e = exponent < 0 ? (U)0 - (U)exponent : (U)exponent;  // :60 abs (INT_MIN stays 2^31)
    U cur = (U)base, pos = 1;
    while (e != 0) {                                          // :65
        if (e & 1) pos *= cur;                                // :67-72
        cur *= cur;                                           // :76
        e >>= 1;                                              // :79
    }
    if (base == 0 && exponent > 0) pos = 0;                   // :81-84
    T neg = 0;                                                // :85
    if (base == 1) neg = 1;                                   // :88-89
    if (base == -1) neg = (exponent & 1) ? -1 : 1;            // :92-95
    return exponent < 0 ? neg : (T)pos;                       // :99-102
}
template <> struct PowOp<int32_t> { static __device__ __forceinline__ int32_t apply(int32_t a, int32_t b) { return powi<int32_t>(a, b); } };
template <> struct PowOp<int64_t> { static __device__ __forceinline__ int64_t apply(int64_t a, int64_t b) { return powi<int64_t>(a, b); } };
// PowOp<float>::apply = std::pow (pow.h:8-10) -> in-register fp64 exp2/log2 chain.
template <> struct PowOp<float> { static __device__ __forceinline__ float apply(float a, float b) { return smpow::powf(a, b); } };
// PowOp<double>::apply = std::pow -> double-double log / table exp (sm_pow64.h); this scalar form reads
// the tables from constant memory, the vector kernels from their LDS copies (OpCtx below).
template <> struct PowOp<double> {
    static __device__ __forceinline__ double apply(double a, double b) { return smpow64::pow(a, b, smpow64::kLogTab, smpow64::kExpTab); }
};

// Per-workgroup state an Op may need.  Kernels create one and call init() at
// their top, before any early exit (init may contain a barrier).  Only
// PowOp<float> has any: its 752-byte log2 breakpoint table, staged from
// constant memory into LDS once per workgroup so the per-lane lookups are
// ds_read_b128s instead of divergent global loads.
template <typename Op> struct OpCtx {
    template <int BLOCK> struct Stage {};
    __device__ __forceinline__ void init() {}
    template <int BLOCK> __device__ __forceinline__ void fetch(Stage<BLOCK> &) const {}
    template <int BLOCK> __device__ __forceinline__ void commit(const Stage<BLOCK> &) {}
};
// init() in two halves for a kernel that has loads of its own to issue: fetch() reads this lane's share of the tables
// into registers (issue it FIRST), commit() writes them to LDS and synchronises.  The kernel's own loads go in
// between: the wait before the LDS writes then covers only the table reads (vmcnt counts in issue order), and both
// round trips are in flight together.  BLOCK = the workgroup size.
template <> struct OpCtx<PowOp<float>> {
    static constexpr int kDoubles = 2 * smpow::kTabN;
    const double *tab;
    template <int BLOCK> struct Stage { double v[(kDoubles + BLOCK - 1) / BLOCK]; };
    __device__ __forceinline__ void init() {
        __shared__ __attribute__((aligned(16))) double lds_tab[kDoubles];
        for (int i = threadIdx.x; i < kDoubles; i += blockDim.x) lds_tab[i] = smpow::kLogTab[i];
        __syncthreads();
        tab = lds_tab;
    }
    template <int BLOCK> __device__ __forceinline__ void fetch(Stage<BLOCK> &st) const {
#pragma unroll
        for (int k = 0; k < (kDoubles + BLOCK - 1) / BLOCK; ++k) {
            const int i = threadIdx.x + k * BLOCK;
            st.v[k] = smpow::kLogTab[i < kDoubles ? i : kDoubles - 1];  // unconditional: no branch between the loads
        }
    }
    template <int BLOCK> __device__ __forceinline__ void commit(const Stage<BLOCK> &st) {
        __shared__ __attribute__((aligned(16))) double lds_tab[kDoubles];
#pragma unroll
        for (int k = 0; k < (kDoubles + BLOCK - 1) / BLOCK; ++k) {
            const int i = threadIdx.x + k * BLOCK;
            if (i < kDoubles) lds_tab[i] = st.v[k];
        }
        __syncthreads();
        tab = lds_tab;
    }
};

template <> struct OpCtx<PowOp<double>> {
    static constexpr int kDoubles = smpow64::kLogTabDoubles + smpow64::kExpTabDoubles;
    const double *logtab, *exptab;
    template <int BLOCK> struct Stage { double v[(kDoubles + BLOCK - 1) / BLOCK]; };
    __device__ __forceinline__ void init() {
        __shared__ __attribute__((aligned(16))) double lds_tab[kDoubles];
        for (int i = threadIdx.x; i < smpow64::kLogTabDoubles; i += blockDim.x) lds_tab[i] = smpow64::kLogTab[i];
        for (int i = threadIdx.x; i < smpow64::kExpTabDoubles; i += blockDim.x) lds_tab[smpow64::kLogTabDoubles + i] = smpow64::kExpTab[i];
        __syncthreads();
        logtab = lds_tab;
        exptab = lds_tab + smpow64::kLogTabDoubles;
    }
    template <int BLOCK> __device__ __forceinline__ void fetch(Stage<BLOCK> &st) const {
#pragma unroll
        for (int k = 0; k < (kDoubles + BLOCK - 1) / BLOCK; ++k) {
            const int i = threadIdx.x + k * BLOCK < kDoubles ? threadIdx.x + k * BLOCK : kDoubles - 1;  // unconditional loads
            const double *src = i < smpow64::kLogTabDoubles ? smpow64::kLogTab + i : smpow64::kExpTab + (i - smpow64::kLogTabDoubles);
            st.v[k] = *src;
        }
    }
    template <int BLOCK> __device__ __forceinline__ void commit(const Stage<BLOCK> &st) {
        __shared__ __attribute__((aligned(16))) double lds_tab[kDoubles];
#pragma unroll
        for (int k = 0; k < (kDoubles + BLOCK - 1) / BLOCK; ++k) {
            const int i = threadIdx.x + k * BLOCK;
            if (i < kDoubles) lds_tab[i] = st.v[k];
        }
        __syncthreads();
        logtab = lds_tab;
        exptab = lds_tab + smpow64::kLogTabDoubles;
    }
};

// PowOp<double> for the flat tile kernel: the same arithmetic reading its tables from BANK-PRIVATE replicas (sm_pow64.h:
// TabBanked), so that a wave's scattered lookups never collide.  Round 2's scalar-exponent kernel spent 43 % of its
// LDS-active cycles in bank conflicts and ran at 68 % of HBM peak on random bases against 78 % when every lane hit the same
// entry.  Sixteen replicas of the five table arrays are 80 KiB of LDS: the kernel runs 1024-thread workgroups (two per
// CU, 64 VGPRs: full occupancy) and each thread stages ten doubles.  Only launched by contiguous.hip's heavy form; every
// other kernel keeps PowOp<double> and its 5 KiB copy.
struct PowBanked {
    static __device__ __forceinline__ double apply(double a, double b) { return PowOp<double>::apply(a, b); }
};
template <> struct OpCtx<PowBanked> {
    static constexpr int kDoubles = smpow64::kBankedDoubles;
    smpow64::TabBanked tab;
    template <int BLOCK> struct Stage { double v[kDoubles / BLOCK]; };
    __device__ __forceinline__ void init() {  // (the tile kernel uses fetch / commit)
        __shared__ __attribute__((aligned(16))) double lds_tab[kDoubles];
        for (int f = threadIdx.x; f < kDoubles; f += blockDim.x) lds_tab[f] = smpow64::table_value((f >> 4) / smpow64::kN, (f >> 4) % smpow64::kN);
        __syncthreads();
        tab.mine = lds_tab + (threadIdx.x & (smpow64::kBankedReplicas - 1));
    }
    // flat slot f = (array * 128 + entry) * 16 + replica: thread t stages slots t, t + BLOCK, ... -- consecutive lanes write
    // consecutive doubles (conflict-free), and the sixteen lanes that share a source value read it as one broadcast
    template <int BLOCK> __device__ __forceinline__ void fetch(Stage<BLOCK> &st) const {
        static_assert(kDoubles % BLOCK == 0, "the replicas are staged in whole rounds");
#pragma unroll
        for (int k = 0; k < kDoubles / BLOCK; ++k) {
            const int f = (int)threadIdx.x + k * BLOCK;
            st.v[k] = smpow64::table_value((f >> 4) / smpow64::kN, (f >> 4) % smpow64::kN);
        }
    }
    template <int BLOCK> __device__ __forceinline__ void commit(const Stage<BLOCK> &st) {
        __shared__ __attribute__((aligned(16))) double lds_tab[kDoubles];
#pragma unroll
        for (int k = 0; k < kDoubles / BLOCK; ++k) lds_tab[(int)threadIdx.x + k * BLOCK] = st.v[k];
        __syncthreads();
        tab.mine = lds_tab + (threadIdx.x & (smpow64::kBankedReplicas - 1));
    }
};

// sm::pow(a, s) on doubles with ONE exponent of moderate magnitude (sm_pow64.h: pow_core_u; LEVEL 1: |s| <= 1024, 2: |s| <= 16):
// the general arithmetic minus what only an unknown exponent needs.  Launched by contiguous.hip's array-scalar form.  The tables
// sit in LDS as five plain arrays (5 KiB; sm_pow64.h: TabSoA).
template <int LEVEL> struct PowScalar64 {
    static __device__ __forceinline__ double apply(double a, double b) { return smpow64::pow_scalar(a, b, LEVEL, smpow64::kLogTab, smpow64::kExpTab); }
};
template <typename Op> struct ScalarLevelOf { static constexpr int value = 0; };
template <int LEVEL> struct ScalarLevelOf<PowScalar64<LEVEL>> { static constexpr int value = LEVEL; };
template <int LEVEL> struct OpCtx<PowScalar64<LEVEL>> {
    static constexpr int kDoubles = smpow64::kSoADoubles;
    smpow64::TabSoA tab;
    template <int BLOCK> struct Stage { double v[(kDoubles + BLOCK - 1) / BLOCK]; };
    static __device__ __forceinline__ double source(int f) { return smpow64::table_value(f / smpow64::kN, f % smpow64::kN); }
    __device__ __forceinline__ void init() {
        __shared__ __attribute__((aligned(16))) double lds_tab[kDoubles];
        for (int f = threadIdx.x; f < kDoubles; f += blockDim.x) lds_tab[f] = source(f);
        __syncthreads();
        tab.base = lds_tab;
    }
    template <int BLOCK> __device__ __forceinline__ void fetch(Stage<BLOCK> &st) const {
#pragma unroll
        for (int k = 0; k < (kDoubles + BLOCK - 1) / BLOCK; ++k) {
            const int f = threadIdx.x + k * BLOCK < kDoubles ? threadIdx.x + k * BLOCK : kDoubles - 1;  // unconditional loads
            st.v[k] = source(f);
        }
    }
    template <int BLOCK> __device__ __forceinline__ void commit(const Stage<BLOCK> &st) {
        __shared__ __attribute__((aligned(16))) double lds_tab[kDoubles];
#pragma unroll
        for (int k = 0; k < (kDoubles + BLOCK - 1) / BLOCK; ++k) {
            const int f = threadIdx.x + k * BLOCK;
            if (f < kDoubles) lds_tab[f] = st.v[k];
        }
        __syncthreads();
        tab.base = lds_tab;
    }
};

// sm::pow(a, s) on doubles with s a multiple of one half, |s| <= 8: a double-double product chain instead of exp(s log a)
// (sm_pow64.h: pow_halfint_n) -- no table, and with the exponent a template parameter no branch either.  M2 = 2 s.  Only
// launched by contiguous.hip's array-scalar form (run_scalar), through the one-shot tile kernel.
template <int M2> struct PowHalfInt64 {
    static __device__ __forceinline__ double apply(double a, double) {
        const double xs[1] = {a};
        double r[1];
        smpow64::pow_halfint_n<1, M2>(xs, M2, r);
        return r[0];
    }
};
template <typename Op> struct HalfIntOf { static constexpr int value = 0; };
template <int M2> struct HalfIntOf<PowHalfInt64<M2>> { static constexpr int value = M2; };

// apply_simd's role: the Op across W independent elements held in registers.
// PowOp<float> evaluates them side by side (one constant per polynomial step,
// no branches); every other Op is one instruction per element.
template <typename Op, typename T, int W>
__device__ __forceinline__ void apply_n(const OpCtx<Op> &ctx, const T (&a)[W], const T (&b)[W], T (&r)[W]) {
    if constexpr (std::is_same<Op, PowOp<float>>::value) {
        smpow::pow_n<W>(a, b, r, ctx.tab);
    } else if constexpr (std::is_same<Op, PowOp<double>>::value) {
        smpow64::pow_n<W>(a, b, r, ctx.logtab, ctx.exptab);
    } else if constexpr (std::is_same<Op, PowBanked>::value) {
        smpow64::pow_n<W, smpow64::TabBanked>(a, b, r, ctx.tab);
    } else if constexpr (HalfIntOf<Op>::value != 0) {
        smpow64::pow_halfint_n<W, HalfIntOf<Op>::value>(a, HalfIntOf<Op>::value, r);
    } else if constexpr (ScalarLevelOf<Op>::value != 0) {
        smpow64::pow_scalar_n<ScalarLevelOf<Op>::value, W>(a, b[0], r, ctx.tab);  // one exponent for the launch (apply_vec_scalar)
    } else {
#pragma unroll
        for (int i = 0; i < W; ++i) r[i] = Op::apply(a[i], b[i]);
    }
}

// ... and across one 16-byte register group.
template <typename Op, typename T>
__device__ __forceinline__ typename VecTraits<T>::vec_t apply_vec(const OpCtx<Op> &ctx, typename VecTraits<T>::vec_t a,
                                                                  typename VecTraits<T>::vec_t b) {
    constexpr int W = VecTraits<T>::width;
    T xa[W], xb[W], xr[W];
#pragma unroll
    for (int i = 0; i < W; ++i) { xa[i] = a[i]; xb[i] = b[i]; }
    apply_n<Op, T, W>(ctx, xa, xb, xr);
    typename VecTraits<T>::vec_t r;
#pragma unroll
    for (int i = 0; i < W; ++i) r[i] = xr[i];
    return r;
}
// SWAPPED = false: a[i] op s;  true: s op a[i]
template <typename Op, typename T, bool SWAPPED>
__device__ __forceinline__ typename VecTraits<T>::vec_t apply_vec_scalar(const OpCtx<Op> &ctx, typename VecTraits<T>::vec_t a, T s) {
    typename VecTraits<T>::vec_t sv;
#pragma unroll
    for (int i = 0; i < VecTraits<T>::width; ++i) sv[i] = s;
    return SWAPPED ? apply_vec<Op, T>(ctx, sv, a) : apply_vec<Op, T>(ctx, a, sv);
}

// ------------------------------------------------------------- fast divmod
// Division of a 31-bit index by a launch-constant divisor as mul-hi + shift
// (replaces the per-element `/` and `%` chain of calculate.h:58-63).
// Valid for 0 <= n < 2^31, 1 <= d < 2^31.
struct FastDiv {
    uint32_t d, mul, shr;
    __host__ __device__ FastDiv() : d(1), mul(0), shr(0) {}
    __host__ explicit FastDiv(uint32_t div) : d(div), mul(0), shr(0) {
        if (div > 1) {
            uint32_t l = 0;
            while ((1ull << l) < div) ++l;  // ceil(log2 d)
            const uint32_t p = 31 + l;
            mul = (uint32_t)(((1ull << p) + div - 1) / div);
            shr = p - 32;
        }
    }
    __host__ __device__ __forceinline__ uint32_t div(uint32_t n) const {
#if defined(__HIP_DEVICE_COMPILE__)
        return d == 1 ? n : (__umulhi(n, mul) >> shr);
#else
        return d == 1 ? n : (uint32_t)(((uint64_t)n * mul) >> 32) >> shr;
#endif
    }
    __host__ __device__ __forceinline__ void divmod(uint32_t n, uint32_t &q, uint32_t &r) const {
        q = div(n);
        r = n - q * d;
    }
};

}  // namespace dev
}  // namespace smhip
