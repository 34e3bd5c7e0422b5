// jit.hip -- user-defined Ops on the device, compiled at run time with hipRTC.
//
// The reference's extension recipe (README.md:86-133) is "write a struct with
// apply() and an x86 apply_simd<__m256> specialisation, then call
// element_wise_op<T, MyOp<T>>".  An x86 intrinsic body means nothing on gfx950,
// and a g++-compiled `apply` cannot run there either -- so the device-side
// plugin contract is one string: the Op's arithmetic as a HIP expression in
// `a` and `b` (e.g. "(a + b) * 2").  smhip_register_op() hands back an op id;
// kernels are compiled around the expression for gfx950 on first use (hiprtc -> code object -> hipModule) and cached:
//     contig   out[i] = a[i] op b[i]          16-byte vectors, one per lane, nt
//     scalar   out[i] = a[i] op s  /  s op a[i]
//     broadcast forms: the SAME row / LDS / tile / gather bodies the built-in Ops use (bcast_kernels.hip.h, embedded
//              as text by the build), in exactly the variant broadcast.hip's plan_launch() picks for the problem; each
//              variant is compiled when a launch first needs it.  `x.apply<MyOp>(y)` therefore runs at the rate of `x * y`
//              on every shape (it was 2.4-3.8x slower through a generic gather: tools/jit_rates.py).
// The built-in Ops never come through here (they are AOT kernels).
#include <hip/hiprtc.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <unistd.h>

#include <condition_variable>
#include <functional>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "bcast_plan.h"
#include "jit_sources.inc"

namespace smhip {
namespace {

struct UserOp {
    std::string expr;
};
std::mutex g_jit_mutex;
std::vector<UserOp *> g_ops;  // id = SMHIP_OP_USER_BASE + index; never freed (ids stay valid)

// One compiled program (source + defines): the gfx950 code object, built once per process (or read from the disk
// cache), and the module it is loaded as on each device that has launched it -- a hipModule_t belongs to the device
// that was current when it was loaded, so modules and functions are kept per device (smhip_set_device /
// smhip_set_devices let one process drive several).
constexpr int kJitMaxDevices = 64;
struct Program {
    int state = 0;  // 0: a thread is building it, 1: ready, 2: failed
    int rc = SMHIP_OK;
    std::string error;
    std::vector<char> code;
    hipModule_t module[kJitMaxDevices] = {};
    std::map<std::pair<int, std::string>, hipFunction_t> functions;  // (device, kernel name)
};
std::condition_variable g_jit_cv;
std::map<std::string, Program *> g_programs;  // never freed: modules stay loaded for the life of the process

const char *kTypeName[4] = {"float", "double", "int", "long long"};

// The kernels' source.  TYPE, WIDTH and EXPR are -D defines.
const char *kSource = R"SRC(
typedef TYPE T;
typedef T V0 __attribute__((ext_vector_type(WIDTH)));
typedef V0 V __attribute__((aligned(sizeof(T))));   // element-aligned 16-byte accesses are legal on gfx950
struct UserOp { static __device__ __forceinline__ T apply(T a, T b) { return (T)(EXPR); } };
// the launch's stream-policy word (csrc/internal.h: stream_policy): bit 0 = loads non-temporal, bit 1 = stores `sc1`.  The
// plain load is two half-width loads on purpose, the keep-store an asm, for the reasons given in csrc/ops.hip.h.
typedef T H0 __attribute__((ext_vector_type(WIDTH / 2)));
typedef H0 H __attribute__((aligned(sizeof(T))));
#if WIDTH == 4
#define SMHIP_JOIN(lo, hi) __builtin_shufflevector(lo, hi, 0, 1, 2, 3)
#else
#define SMHIP_JOIN(lo, hi) __builtin_shufflevector(lo, hi, 0, 1)
#endif
__device__ __forceinline__ V0 smhip_ld(const V* p, int pol) {
    if (pol & 1) return __builtin_nontemporal_load(p);
    const H* h = (const H*)p;
    const H0 lo = h[0], hi = h[1];
    return SMHIP_JOIN(lo, hi);
}
__device__ __forceinline__ void smhip_st(V* p, V0 r, int pol) {
    if (pol & 2) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(r));
    else __builtin_nontemporal_store(r, p);
}

extern "C" __global__ __launch_bounds__(256) void smhip_user_contig(const T* __restrict__ a, const T* __restrict__ b,
                                                                     T* __restrict__ out, unsigned long long n_vec,
                                                                     unsigned long long n, int vec, int pol) {
    const unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    if (vec) {
        if (i < n_vec) {
            const V0 va = smhip_ld((const V*)a + i, pol), vb = smhip_ld((const V*)b + i, pol);
            V0 r;
            for (int k = 0; k < WIDTH; ++k) r[k] = UserOp::apply(va[k], vb[k]);
            smhip_st((V*)out + i, r, pol);
        } else if (i == n_vec) {
            for (unsigned long long k = n_vec * WIDTH; k < n; ++k) out[k] = UserOp::apply(a[k], b[k]);
        }
    } else if (i < n) {
        out[i] = UserOp::apply(a[i], b[i]);
    }
}

extern "C" __global__ __launch_bounds__(256) void smhip_user_scalar(const T* __restrict__ a, T s, T* __restrict__ out,
                                                                     unsigned long long n_vec, unsigned long long n, int swapped, int pol) {
    const unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n_vec) {
        const V0 va = smhip_ld((const V*)a + i, pol);
        V0 r;
        for (int k = 0; k < WIDTH; ++k) r[k] = swapped ? UserOp::apply(s, va[k]) : UserOp::apply(va[k], s);
        smhip_st((V*)out + i, r, pol);
    } else if (i == n_vec) {
        for (unsigned long long k = n_vec * WIDTH; k < n; ++k) out[k] = swapped ? UserOp::apply(s, a[k]) : UserOp::apply(a[k], s);
    }
}

)SRC";

// N-ary fused expressions (smhip_fused_expr): out[i] = EXPR(a0[i], ..., a7[i], s0..s3) over dense operands in ONE pass --
// (k + 1) * sizeof(T) bytes per element instead of 3 * sizeof(T) per operator of the chain it replaces.
const char *kExprSource = R"SRC(
typedef TYPE T;
typedef T V0 __attribute__((ext_vector_type(WIDTH)));
typedef V0 V __attribute__((aligned(sizeof(T))));
// the launch's stream-policy word (csrc/internal.h: stream_policy): bit 0 = loads non-temporal, bit 1 = stores `sc1`.  The
// plain load is two half-width loads on purpose, the keep-store an asm, for the reasons given in csrc/ops.hip.h.
typedef T H0 __attribute__((ext_vector_type(WIDTH / 2)));
typedef H0 H __attribute__((aligned(sizeof(T))));
#if WIDTH == 4
#define SMHIP_JOIN(lo, hi) __builtin_shufflevector(lo, hi, 0, 1, 2, 3)
#else
#define SMHIP_JOIN(lo, hi) __builtin_shufflevector(lo, hi, 0, 1)
#endif
__device__ __forceinline__ V0 smhip_ld(const V* p, int pol) {
    if (pol & 1) return __builtin_nontemporal_load(p);
    const H* h = (const H*)p;
    const H0 lo = h[0], hi = h[1];
    return SMHIP_JOIN(lo, hi);
}
__device__ __forceinline__ void smhip_st(V* p, V0 r, int pol) {
    if (pol & 2) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(r));
    else __builtin_nontemporal_store(r, p);
}
struct Operands { const T* p[8]; };
struct Scalars { T v[4]; };
__device__ __forceinline__ T smhip_eval(T a0, T a1, T a2, T a3, T a4, T a5, T a6, T a7, T s0, T s1, T s2, T s3) { return (T)(EXPR); }
extern "C" __global__ __launch_bounds__(BLOCK) void smhip_user_expr(Operands in, Scalars sc, T* __restrict__ out, unsigned long long n_vec,
                                                                   unsigned long long n, int pol) {
    const unsigned long long i = (unsigned long long)blockIdx.x * BLOCK + threadIdx.x;
    if (i < n_vec) {
        V0 v[8];
        if (pol & 1) {  // one branch around the group of loads (csrc/ops.hip.h: load_stream_as)
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (k < NOPS) v[k] = smhip_ld((const V*)in.p[k] + i, 1);
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (k < NOPS) v[k] = smhip_ld((const V*)in.p[k] + i, 0);
        }
#pragma unroll
        for (int k = NOPS; k < 8; ++k) v[k] = v[0];
        V0 r;
#pragma unroll
        for (int e = 0; e < WIDTH; ++e) r[e] = smhip_eval(v[0][e], v[1][e], v[2][e], v[3][e], v[4][e], v[5][e], v[6][e], v[7][e], sc.v[0], sc.v[1], sc.v[2], sc.v[3]);
        smhip_st((V*)out + i, r, pol);
    } else if (i == n_vec) {
        for (unsigned long long j = n_vec * WIDTH; j < n; ++j) {
            T x[8];
            for (int k = 0; k < 8; ++k) x[k] = in.p[k < NOPS ? k : 0][j];
            out[j] = smhip_eval(x[0], x[1], x[2], x[3], x[4], x[5], x[6], x[7], sc.v[0], sc.v[1], sc.v[2], sc.v[3]);
        }
    }
}

// The same with a sum of the results (SUM_OUT: the elementwise result is stored as well; otherwise reduce only): one
// accumulator per workgroup -- fp64 for float types, wrapping 64-bit for integer types, like the built-in reductions.
typedef ACC A;
__device__ __forceinline__ A smhip_widen(T x) { return WIDEN; }
// lane exchange by DPP moves, as reduce.hip's wave_reduce (the wave's total ends up in lane 63; __shfl_down would be a
// ds_bpermute round trip per 32-bit half and stage)
template <int CTRL, int ROW_MASK> __device__ __forceinline__ A smhip_dpp(A v) {
    const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)b, CTRL, ROW_MASK, 0xf, false);
    const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(b >> 32), CTRL, ROW_MASK, 0xf, false);
    return __builtin_bit_cast(A, ((unsigned long long)hi << 32) | lo);
}
extern "C" __global__ __launch_bounds__(BLOCK) void smhip_user_expr_sum(Operands in, Scalars sc, T* __restrict__ out, int store,
                                                                       unsigned long long n_vec, unsigned long long n,
                                                                       A* __restrict__ partials, int pol) {
    const unsigned long long i = (unsigned long long)blockIdx.x * BLOCK + threadIdx.x;
    A acc = 0;
    if (i < n_vec) {
        V0 v[8];
        if (pol & 1) {
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (k < NOPS) v[k] = smhip_ld((const V*)in.p[k] + i, 1);
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (k < NOPS) v[k] = smhip_ld((const V*)in.p[k] + i, 0);
        }
#pragma unroll
        for (int k = NOPS; k < 8; ++k) v[k] = v[0];
        V0 r;
#pragma unroll
        for (int e = 0; e < WIDTH; ++e) {
            r[e] = smhip_eval(v[0][e], v[1][e], v[2][e], v[3][e], v[4][e], v[5][e], v[6][e], v[7][e], sc.v[0], sc.v[1], sc.v[2], sc.v[3]);
            acc += smhip_widen(r[e]);
        }
        if (store) smhip_st((V*)out + i, r, pol);
    } else if (i == n_vec) {
        for (unsigned long long j = n_vec * WIDTH; j < n; ++j) {
            T x[8];
            for (int k = 0; k < 8; ++k) x[k] = in.p[k < NOPS ? k : 0][j];
            const T r = smhip_eval(x[0], x[1], x[2], x[3], x[4], x[5], x[6], x[7], sc.v[0], sc.v[1], sc.v[2], sc.v[3]);
            if (store) out[j] = r;
            acc += smhip_widen(r);
        }
    }
    acc += smhip_dpp<0x111, 0xf>(acc);  // row_shr:1, 2, 4, 8, then row_bcast:15 and :31
    acc += smhip_dpp<0x112, 0xf>(acc);
    acc += smhip_dpp<0x114, 0xf>(acc);
    acc += smhip_dpp<0x118, 0xf>(acc);
    acc += smhip_dpp<0x142, 0xa>(acc);
    acc += smhip_dpp<0x143, 0xc>(acc);
    __shared__ A lds[BLOCK / 64];
    if ((threadIdx.x & 63) == 63) lds[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        A total = lds[0];
        for (int w = 1; w < BLOCK / 64; ++w) total += lds[w];  // index order: the same bits on every run
        partials[blockIdx.x] = total;
    }
}
)SRC";

// What bcast_kernels.hip.h expects to find declared (ops.hip.h provides it ahead of time): fixed-width types, the
// element-aligned 16-byte vector types, the streaming access macros, FastDiv (same layout as dev::FastDiv: the host fills
// it in), and the Op plumbing -- a user Op has no per-workgroup state and is applied element by element.
const char *kBcastPrelude = R"SRC(
typedef __UINT32_TYPE__ uint32_t;
typedef __INT32_TYPE__ int32_t;
typedef __INT64_TYPE__ int64_t;
typedef __UINT64_TYPE__ uint64_t;
typedef __SIZE_TYPE__ size_t;
typedef __UINTPTR_TYPE__ uintptr_t;
#define SMHIP_MAX_NDIM 6
template <typename T> struct VecTraits;
#define SMHIP_VEC(T, N) template <> struct VecTraits<T> { typedef T full_t __attribute__((ext_vector_type(N))); \
    typedef full_t vec_t __attribute__((aligned(sizeof(T)))); typedef T half_full_t __attribute__((ext_vector_type(N / 2))); \
    typedef half_full_t half_t __attribute__((aligned(sizeof(T)))); static constexpr int width = N; \
    static __device__ __forceinline__ full_t join(half_full_t lo, half_full_t hi) { return __builtin_shufflevector(lo, hi, SMHIP_JOIN_##N); } };
#define SMHIP_JOIN_4 0, 1, 2, 3
#define SMHIP_JOIN_2 0, 1
SMHIP_VEC(float, 4) SMHIP_VEC(int32_t, 4) SMHIP_VEC(double, 2) SMHIP_VEC(int64_t, 2)
constexpr int kLoadNt = 1, kStoreKeep = 2;  // bits of a launch's stream-policy word (ops.hip.h)
#define load_stream(ptr) __builtin_nontemporal_load(ptr)
#define store_stream(ptr, ...) __builtin_nontemporal_store((__VA_ARGS__), (ptr))
#define store_stream_as(T, ptr, value, NT) do { typedef VecTraits<T> smhip_tr_; typename smhip_tr_::vec_t *smhip_q_ = (ptr); const typename smhip_tr_::full_t smhip_w_ = (value); \
    if constexpr (NT) { __builtin_nontemporal_store(smhip_w_, smhip_q_); } else { asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(smhip_q_), "v"(smhip_w_)); } } while (0)
#define store_stream_if(T, ptr, value, pol) do { if ((pol) & kStoreKeep) store_stream_as(T, ptr, value, false); else store_stream_as(T, ptr, value, true); } while (0)
#define load_stream_as(T, ptr, NT) ({ typedef VecTraits<T> smhip_tr_; const typename smhip_tr_::vec_t *smhip_p_ = (ptr); typename smhip_tr_::full_t smhip_v_; \
    if constexpr (NT) { smhip_v_ = __builtin_nontemporal_load(smhip_p_); } else { const typename smhip_tr_::half_t *smhip_h_ = reinterpret_cast<const typename smhip_tr_::half_t *>(smhip_p_); \
    const typename smhip_tr_::half_full_t smhip_lo_ = smhip_h_[0], smhip_hi_ = smhip_h_[1]; smhip_v_ = smhip_tr_::join(smhip_lo_, smhip_hi_); } smhip_v_; })
#define load_stream_if(T, ptr, nt) ({ typedef VecTraits<T> smhip_tr_; const typename smhip_tr_::vec_t *smhip_p_ = (ptr); typename smhip_tr_::full_t smhip_v_; \
    if ((nt) & kLoadNt) { smhip_v_ = __builtin_nontemporal_load(smhip_p_); } else { const typename smhip_tr_::half_t *smhip_h_ = reinterpret_cast<const typename smhip_tr_::half_t *>(smhip_p_); \
    const typename smhip_tr_::half_full_t smhip_lo_ = smhip_h_[0], smhip_hi_ = smhip_h_[1]; smhip_v_ = smhip_tr_::join(smhip_lo_, smhip_hi_); } smhip_v_; })
struct FastDiv {
    uint32_t d, mul, shr;
    __device__ FastDiv() : d(1), mul(0), shr(0) {}
    __device__ __forceinline__ uint32_t div(uint32_t n) const { return d == 1 ? n : (__umulhi(n, mul) >> shr); }
    __device__ __forceinline__ void divmod(uint32_t n, uint32_t &q, uint32_t &r) const { q = div(n); r = n - q * d; }
};
template <typename Op> struct OpCtx {
    template <int BLOCK> struct Stage {};
    __device__ __forceinline__ void init() {}
    template <int BLOCK> __device__ __forceinline__ void fetch(Stage<BLOCK> &) const {}
    template <int BLOCK> __device__ __forceinline__ void commit(const Stage<BLOCK> &) {}
};
template <typename Op, typename T, int W>
__device__ __forceinline__ void apply_n(const OpCtx<Op> &, const T (&a)[W], const T (&b)[W], T (&r)[W]) {
#pragma unroll
    for (int i = 0; i < W; ++i) r[i] = Op::apply(a[i], b[i]);
}
)SRC";

const char *kBcastWrapper = R"SRC(
typedef TYPE T;
struct UserOp { static __device__ __forceinline__ T apply(T a, T b) { return (T)(EXPR); } };
extern "C" __global__ __launch_bounds__(256) void smhip_user_bcast(const T* __restrict__ a, const T* __restrict__ b,
                                                                    T* __restrict__ out, PARAMS p) {
    BODY(a, b, out, p);
}
)SRC";

const char *kStdTypeName[4] = {"float", "double", "int32_t", "int64_t"};

// Code objects are cached on disk, keyed by a hash of (library version, source text, options): a program that registers the
// same Ops on every run pays the ~0.3 s of hipRTC per kernel variant once per machine, not once per process.
// SMHIP_JIT_CACHE=<dir> moves the cache (default $XDG_CACHE_HOME/smhip or ~/.cache/smhip); SMHIP_JIT_CACHE=off disables it.
std::string cache_dir() {
    const char *env = getenv("SMHIP_JIT_CACHE");
    if (env && (!strcmp(env, "off") || !strcmp(env, "0") || !*env)) return "";
    std::string dir;
    if (env) dir = env;
    else if (const char *x = getenv("XDG_CACHE_HOME")) dir = std::string(x) + "/smhip";
    else if (const char *h = getenv("HOME")) dir = std::string(h) + "/.cache/smhip";
    else return "";
    std::string partial;
    for (size_t i = 0; i <= dir.size(); ++i) {  // mkdir -p
        if (i == dir.size() || dir[i] == '/') {
            if (!partial.empty()) mkdir(partial.c_str(), 0755);
        }
        if (i < dir.size()) partial += dir[i];
    }
    struct stat st;
    if (stat(dir.c_str(), &st) != 0 || !S_ISDIR(st.st_mode) || access(dir.c_str(), W_OK) != 0) return "";
    return dir;
}

uint64_t fnv1a(uint64_t h, const std::string &s) {
    for (unsigned char c : s) { h ^= c; h *= 0x100000001b3ULL; }
    return h ^ 0xff;  // separator between fields
}

int hiprtc_build(const std::string &source, const std::vector<std::string> &defines, const char *what, const std::string &expr,
                 std::vector<char> *code_out) {
    static const std::string dir = cache_dir();
    std::string path;
    if (!dir.empty()) {
        int rtc_major = 0, rtc_minor = 0;
        hiprtcVersion(&rtc_major, &rtc_minor);
        char tag[96];
        snprintf(tag, sizeof tag, "%s hiprtc %d.%d -O3 -ffp-contract=off -std=c++17", smhip_version(), rtc_major, rtc_minor);
        uint64_t h = fnv1a(0xcbf29ce484222325ULL, tag);
        h = fnv1a(h, source);
        for (const auto &d : defines) h = fnv1a(h, d);
        char name[64];
        snprintf(name, sizeof name, "/%016llx.hsaco", (unsigned long long)h);
        path = dir + name;
        if (FILE *f = fopen(path.c_str(), "rb")) {
            std::vector<char> code;
            char buf[65536];
            size_t got;
            while ((got = fread(buf, 1, sizeof buf, f)) > 0) code.insert(code.end(), buf, buf + got);
            fclose(f);
            // an ELF code object of plausible size; a truncated or foreign file is rebuilt (and overwritten) below
            if (code.size() > 64 && !memcmp(code.data(), "\x7f" "ELF", 4)) {
                code_out->swap(code);
                return SMHIP_OK;
            }
        }
    }
    hiprtcProgram prog;
    if (hiprtcCreateProgram(&prog, source.c_str(), "smhip_user_op.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS)
        return fail(SMHIP_ERR_HIP, "hiprtcCreateProgram failed");
    std::vector<const char *> opts = {"--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-std=c++17"};
    for (const auto &d : defines) opts.push_back(d.c_str());
    const hiprtcResult rc = hiprtcCompileProgram(prog, (int)opts.size(), opts.data());
    if (rc != HIPRTC_SUCCESS) {
        size_t n = 0;
        hiprtcGetProgramLogSize(prog, &n);
        std::string log(n, '\0');
        if (n) hiprtcGetProgramLog(prog, &log[0]);
        hiprtcDestroyProgram(&prog);
        return fail(SMHIP_ERR_INVALID, "user op \"%s\" does not compile for %s: %.400s", expr.c_str(), what, log.c_str());
    }
    size_t size = 0;
    hiprtcGetCodeSize(prog, &size);
    std::vector<char> code(size);
    hiprtcGetCode(prog, code.data());
    hiprtcDestroyProgram(&prog);
    if (!path.empty()) {  // publish atomically: write a private file, then rename it into place
        char tmp[32];
        snprintf(tmp, sizeof tmp, ".%ld.tmp", (long)getpid());
        const std::string tpath = path + tmp;
        if (FILE *f = fopen(tpath.c_str(), "wb")) {
            const bool ok = fwrite(code.data(), 1, code.size(), f) == code.size();
            fclose(f);
            if (!ok || rename(tpath.c_str(), path.c_str()) != 0) remove(tpath.c_str());
        }
    }
    code_out->swap(code);
    return SMHIP_OK;
}

// The kernel `name` of the program `key` on the calling thread's device.  The first caller of a key builds it with
// g_jit_mutex RELEASED (a 0.3 s hipRTC compile does not stall other threads' launches of other kernels); callers
// that arrive meanwhile for the same key wait for that build instead of repeating it.
int get_function(const std::string &key, const std::function<int(std::vector<char> *)> &build, const char *name, hipFunction_t *fn) {
    const int dev = current_device();
    if (dev < 0 || dev >= kJitMaxDevices) return fail(SMHIP_ERR_INVALID, "jit: device %d", dev);
    std::unique_lock<std::mutex> lock(g_jit_mutex);
    Program *p;
    auto it = g_programs.find(key);
    if (it == g_programs.end()) {
        p = new Program;
        g_programs.emplace(key, p);
        lock.unlock();
        std::vector<char> code;
        const int rc = build(&code);
        lock.lock();
        p->code.swap(code);
        p->rc = rc;
        if (rc) p->error = smhip_last_error();
        p->state = rc ? 2 : 1;
        g_jit_cv.notify_all();
    } else {
        p = it->second;
        g_jit_cv.wait(lock, [&] { return p->state != 0; });
    }
    if (p->state == 2) return fail(p->rc, "%s", p->error.c_str());
    auto f = p->functions.find({dev, name});
    if (f != p->functions.end()) {
        *fn = f->second;
        return SMHIP_OK;
    }
    if (!p->module[dev]) SMHIP_TRY(hipModuleLoadData(&p->module[dev], p->code.data()));
    SMHIP_TRY(hipModuleGetFunction(fn, p->module[dev], name));
    p->functions[{dev, name}] = *fn;
    return SMHIP_OK;
}

int user_expr(int op, std::string *expr) {
    std::lock_guard<std::mutex> lock(g_jit_mutex);
    const int idx = op - SMHIP_OP_USER_BASE;
    if (idx < 0 || idx >= (int)g_ops.size()) return fail(SMHIP_ERR_INVALID, "op %d was never registered", op);
    *expr = g_ops[idx]->expr;
    return SMHIP_OK;
}

// The flat kernels (contig / scalar) of a user Op for one element type.
int flat_function(int op, int dtype, const char *name, hipFunction_t *fn) {
    std::string expr;
    if (int rc = user_expr(op, &expr)) return rc;
    return get_function(std::string("flat|") + kTypeName[dtype] + "|" + expr,
                        [&](std::vector<char> *code) {
                            return hiprtc_build(kSource, {std::string("-DTYPE=") + kTypeName[dtype],
                                                          std::string("-DWIDTH=") + ((dtype == SMHIP_F64 || dtype == SMHIP_I64) ? "2" : "4"),
                                                          "-DEXPR=" + expr},
                                                kTypeName[dtype], expr, code);
                        },
                        name, fn);
}

}  // namespace

int jit_register(const char *expr, int *op_id) {
    std::lock_guard<std::mutex> lock(g_jit_mutex);
    for (size_t i = 0; i < g_ops.size(); ++i)
        if (g_ops[i]->expr == expr) {
            *op_id = SMHIP_OP_USER_BASE + (int)i;
            return SMHIP_OK;
        }
    UserOp *u = new UserOp;
    u->expr = expr;
    g_ops.push_back(u);
    *op_id = SMHIP_OP_USER_BASE + (int)g_ops.size() - 1;
    return SMHIP_OK;
}

int jit_contiguous(int op, int dtype, const void *a, const void *b, void *out, size_t n, hipStream_t s) {
    hipFunction_t fn;
    if (int rc = flat_function(op, dtype, "smhip_user_contig", &fn)) return rc;
    const unsigned long long w = (dtype == SMHIP_F64 || dtype == SMHIP_I64) ? 2 : 4;
    int vec = 1;
    unsigned long long n_vec = n / w, nn = n;
    const size_t threads = vec ? n_vec + 1 : n;
    const size_t grid = (threads + 255) / 256;
    if (grid > 0x7fffffffu) return fail(SMHIP_ERR_UNSUPPORTED, "user op: array too large for one launch");
    int pol = stream_policy({{a, n * dtype_size(dtype)}, {b, n * dtype_size(dtype)}}, {out, n * dtype_size(dtype)});
    const unsigned long long kPiece = piece_for(n_vec, 3);  // large arrays go out in pieces (contiguous.hip, internal.h)
    if (kPiece) {
        const size_t esz = dtype_size(dtype);
        for (unsigned long long v0 = 0;; v0 += kPiece) {
            const bool last = v0 + kPiece >= n_vec;
            const void *pa = static_cast<const char *>(a) + v0 * w * esz, *pb = static_cast<const char *>(b) + v0 * w * esz;
            void *po = static_cast<char *>(out) + v0 * w * esz;
            unsigned long long nv = last ? n_vec - v0 : kPiece, ne = last ? n - v0 * w : kPiece * w;
            void *pargs[] = {&pa, &pb, &po, &nv, &ne, &vec, &pol};
            SMHIP_TRY(hipModuleLaunchKernel(fn, (unsigned)((nv + 1 + 255) / 256), 1, 1, 256, 1, 1, 0, s, pargs, nullptr));
            if (last) return SMHIP_OK;
        }
    }
    void *args[] = {&a, &b, &out, &n_vec, &nn, &vec, &pol};
    SMHIP_TRY(hipModuleLaunchKernel(fn, (unsigned)grid, 1, 1, 256, 1, 1, 0, s, args, nullptr));
    return SMHIP_OK;
}

int jit_array_scalar(int op, int dtype, const void *a, const void *value_host, size_t n, void *out, hipStream_t s) {
    hipFunction_t fn;
    if (int rc = flat_function(op, dtype, "smhip_user_scalar", &fn)) return rc;
    const unsigned long long w = (dtype == SMHIP_F64 || dtype == SMHIP_I64) ? 2 : 4;
    unsigned long long n_vec = n / w, nn = n;
    int swapped = 0;
    const size_t grid = (n_vec + 1 + 255) / 256;
    if (grid > 0x7fffffffu) return fail(SMHIP_ERR_UNSUPPORTED, "user op: array too large for one launch");
    unsigned char scalar[8];
    memcpy(scalar, value_host, dtype_size(dtype));
    int pol = stream_policy({{a, n * dtype_size(dtype)}}, {out, n * dtype_size(dtype)});
    const unsigned long long kPiece = piece_for(n_vec, 2);
    if (kPiece) {
        const size_t esz = dtype_size(dtype);
        for (unsigned long long v0 = 0;; v0 += kPiece) {
            const bool last = v0 + kPiece >= n_vec;
            const void *pa = static_cast<const char *>(a) + v0 * w * esz;
            void *po = static_cast<char *>(out) + v0 * w * esz;
            unsigned long long nv = last ? n_vec - v0 : kPiece, ne = last ? n - v0 * w : kPiece * w;
            void *pargs[] = {&pa, scalar, &po, &nv, &ne, &swapped, &pol};
            SMHIP_TRY(hipModuleLaunchKernel(fn, (unsigned)((nv + 1 + 255) / 256), 1, 1, 256, 1, 1, 0, s, pargs, nullptr));
            if (last) return SMHIP_OK;
        }
    }
    void *args[] = {&a, scalar, &out, &n_vec, &nn, &swapped, &pol};
    SMHIP_TRY(hipModuleLaunchKernel(fn, (unsigned)grid, 1, 1, 256, 1, 1, 0, s, args, nullptr));
    return SMHIP_OK;
}

// One broadcast launch of a user Op: the variant plan_launch() chose, compiled on first use.
int jit_launch(int op, int dtype, const bk::Launch &L, const void *a, const void *b, void *out, hipStream_t s) {
    using bk::Launch;
    char body[160];
    const char *params = "";
    const void *x = a, *y = b;
    switch (L.kind) {
        case Launch::kRow:
            snprintf(body, sizeof body, "row_body<T,UserOp,%d,%d,%s,%s,%d,%d>", L.ia, L.ib, L.ca ? "true" : "false", L.cb ? "true" : "false", L.tx, L.rows);
            params = "RowParams";
            break;
        case Launch::kLds:
            snprintf(body, sizeof body, "dense_lds_body<T,UserOp,%s,%d>", L.swapped ? "true" : "false", bk::kLdsVectorsInFlight);
            params = "LdsParams";
            if (L.swapped) { x = b; y = a; }  // the streamed operand comes first
            break;
        case Launch::kTile:
            snprintf(body, sizeof body, "tile_body<T,UserOp,%s,%d,%d,%d>", L.vec ? "true" : "false", L.ma, L.mb, L.vec ? L.qb : bk::kTileQBytes);
            params = "TileParams";
            break;
        case Launch::kGather:
            snprintf(body, sizeof body, "gather_body<T,UserOp,%d>", L.w);
            params = "GatherParams";
            break;
        case Launch::kStrided:
            snprintf(body, sizeof body, "strided_row_body<T,UserOp,%d,%d>", L.ia, L.ib);
            params = "StridedParams";
            break;
    }
    std::string expr;
    if (int rc = user_expr(op, &expr)) return rc;
    hipFunction_t fn = nullptr;
    if (int rc = get_function(std::string("bcast|") + kStdTypeName[dtype] + "|" + body + "|" + expr,
                              [&](std::vector<char> *code) {
                                  const std::string source = std::string(kBcastPrelude) + kBcastKernelsSrc + kBcastWrapper;
                                  return hiprtc_build(source, {std::string("-DTYPE=") + kStdTypeName[dtype], "-DEXPR=" + expr,
                                                               std::string("-DPARAMS=") + params, std::string("-DBODY=") + body},
                                                      kStdTypeName[dtype], expr, code);
                              },
                              "smhip_user_bcast", &fn))
        return rc;
    void *pblock = const_cast<void *>(static_cast<const void *>(&L.p));
    void *args[] = {&x, &y, &out, pblock};
    SMHIP_TRY(hipModuleLaunchKernel(fn, L.grid, 1, 1, 256, 1, 1, (unsigned)L.lds_bytes, s, args, nullptr));
    return SMHIP_OK;
}

int jit_fused_expr(const char *expr, int dtype, const void *const *operands, int n_operands, const void *scalars_host, int n_scalars,
                   void *out, size_t n, double *sum_dev, hipStream_t s) {
    hipFunction_t fn = nullptr;
    // workgroups of 1024 for three or more input streams of a large array, as the built-in fused kernel has: 3R+1W 77.4 ->
    // 79.5 %, 4R+1W 75.9 -> 78.1 % at 256 MiB per array; the reducing form keeps 256 (read-only streams want it)
    const unsigned long long w_ = (dtype == SMHIP_F64 || dtype == SMHIP_I64) ? 2 : 4;
    const int block = (!sum_dev && n_operands >= 3 && n / w_ >= ((size_t)1 << 20)) ? 1024 : 256;
    {
        char head[48];
        snprintf(head, sizeof head, "expr|%d|%d|%d|", dtype, n_operands, block);
        const bool integer = dtype == SMHIP_I32 || dtype == SMHIP_I64;
        if (int rc = get_function(std::string(head) + expr,
                                  [&](std::vector<char> *code) {
                                      char nops[24], blk[24];
                                      snprintf(nops, sizeof nops, "-DNOPS=%d", n_operands);
                                      snprintf(blk, sizeof blk, "-DBLOCK=%d", block);
                                      return hiprtc_build(kExprSource,
                                                          {std::string("-DTYPE=") + kTypeName[dtype],
                                                           std::string("-DWIDTH=") + ((dtype == SMHIP_F64 || dtype == SMHIP_I64) ? "2" : "4"), nops, blk,
                                                           std::string("-DEXPR=") + expr, integer ? "-DACC=unsigned long long" : "-DACC=double",
                                                           integer ? "-DWIDEN=(A)(long long)x" : "-DWIDEN=(A)x"},
                                                          kTypeName[dtype], expr, code);
                                  },
                                  sum_dev ? "smhip_user_expr_sum" : "smhip_user_expr", &fn))
            return rc;
    }
    struct { const void *p[8]; } in;
    for (int k = 0; k < 8; ++k) in.p[k] = operands[k < n_operands ? k : 0];
    const unsigned long long w = (dtype == SMHIP_F64 || dtype == SMHIP_I64) ? 2 : 4;
    unsigned long long n_vec = n / w, nn = n;
    const size_t grid = (n_vec + 1 + (size_t)block - 1) / (size_t)block;
    if (grid > 0x7fffffffu) return fail(SMHIP_ERR_UNSUPPORTED, "fused expression: array too large for one launch");
    unsigned char sc[32] = {};  // Scalars { T v[4]; }: runtime values, so changing them does not recompile
    if (n_scalars > 0) memcpy(sc, scalars_host, (size_t)n_scalars * dtype_size(dtype));
    int pol = stream_policy((size_t)n_operands * n * dtype_size(dtype), out ? n * dtype_size(dtype) : 0);
    {   // operands nothing has touched recently are read non-temporally (internal.h: refine_policy); one call per operand
        // keeps the initializer lists fixed-size -- the hint is per launch, so any cold majority decides
        int refined = pol;
        for (int k = 0; k < n_operands; ++k)
            refined |= refine_policy(pol, {{operands[k], n * dtype_size(dtype)}}, {k == 0 ? out : nullptr, k == 0 && out ? n * dtype_size(dtype) : 0});
        pol = refined;
    }
    if (!sum_dev) {
        void *args[] = {&in, sc, &out, &n_vec, &nn, &pol};
        SMHIP_TRY(hipModuleLaunchKernel(fn, (unsigned)grid, 1, 1, (unsigned)block, 1, 1, 0, s, args, nullptr));
        return SMHIP_OK;
    }
    // expression + sum: per-workgroup partials, then the built-in reductions' fixed-order fold and final pass
    double *scratch;
    ScratchLease lease;
    if (int rc = lease.take(grid + grid / kReduceFoldSpan + 2, &scratch)) return rc;
    int store = out != nullptr;
    void *args[] = {&in, sc, &out, &store, &n_vec, &nn, &scratch, &pol};
    SMHIP_TRY(hipModuleLaunchKernel(fn, (unsigned)grid, 1, 1, 256, 1, 1, 0, s, args, nullptr));
    return reduce_finish(dtype, scratch, grid, sum_dev, s);
}

// smhip_fused_expr_bcast: the n-ary expression kernel over operands that BROADCAST against the result -- each operand in
// the index form chain.hip classified it into (dense / row / splat; internal.h: ExprLeaf).  The kernel text is generated
// per combination of forms: dense operands load inside the read-policy branch (one block of loads), the small ones through
// the caches behind it.  Same arithmetic as smhip_user_expr: each operation rounds as the separate operators do.
namespace {
const char *kExprBcastHead = R"SRC(
typedef TYPE T;
typedef T V0 __attribute__((ext_vector_type(WIDTH)));
typedef V0 V __attribute__((aligned(sizeof(T))));
typedef T H0 __attribute__((ext_vector_type(WIDTH / 2)));
typedef H0 H __attribute__((aligned(sizeof(T))));
#if WIDTH == 4
#define SMHIP_JOIN(lo, hi) __builtin_shufflevector(lo, hi, 0, 1, 2, 3)
#else
#define SMHIP_JOIN(lo, hi) __builtin_shufflevector(lo, hi, 0, 1)
#endif
__device__ __forceinline__ V0 smhip_ld(const V* p, int nt) {
    if (nt) return __builtin_nontemporal_load(p);
    const H* h = (const H*)p;
    const H0 lo = h[0], hi = h[1];
    return SMHIP_JOIN(lo, hi);
}
__device__ __forceinline__ void smhip_st(V* p, V0 r, int pol) {
    if (pol & 2) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(r));
    else __builtin_nontemporal_store(r, p);
}
struct FastDiv {
    unsigned d, mul, shr;
    __device__ __forceinline__ unsigned div(unsigned n) const { return d == 1 ? n : (__umulhi(n, mul) >> shr); }
    __device__ __forceinline__ unsigned mod(unsigned n) const { return n - div(n) * d; }
};
// row: a = the period in vectors, off = this launch's first vector within it.  splat: a = elements (vectors, when whole)
// per value, b = values before the operand repeats, off / q0 = where this launch starts, elem = 1: one index per element.
struct BOperand { const T* p; FastDiv a, b; unsigned off, q0, elem, pad; };
struct Operands { BOperand o[8]; };
struct Scalars { T v[4]; };
__device__ __forceinline__ T smhip_eval(T a0, T a1, T a2, T a3, T a4, T a5, T a6, T a7, T s0, T s1, T s2, T s3) { return (T)(EXPR); }
__device__ __forceinline__ V0 smhip_row(const BOperand& o, unsigned v) { return ((const V*)o.p)[o.a.mod(o.off + v)]; }
__device__ __forceinline__ V0 smhip_splat(const BOperand& o, unsigned v) {
    V0 r;
    if (o.elem) {
        for (int e = 0; e < WIDTH; ++e) r[e] = o.p[o.b.mod(o.q0 + o.a.div(o.off + v * WIDTH + e))];
    } else {
        const T y = o.p[o.b.mod(o.q0 + o.a.div(o.off + v))];
        for (int e = 0; e < WIDTH; ++e) r[e] = y;
    }
    return r;
}
)SRC";
}  // namespace

int jit_fused_expr_bcast(const char *expr, int dtype, const ExprLeaf *leaves, int n_operands, const void *scalars_host, int n_scalars,
                         void *out, size_t n, hipStream_t s) {
    const unsigned long long w = (dtype == SMHIP_F64 || dtype == SMHIP_I64) ? 2 : 4;
    std::string kinds;
    int n_dense = 0;
    for (int k = 0; k < n_operands; ++k) { kinds += (char)('0' + leaves[k].kind); n_dense += leaves[k].kind == 0; }
    hipFunction_t fn = nullptr;
    if (int rc = get_function(std::string("exprb|") + kTypeName[dtype] + "|" + kinds + "|" + expr,
                              [&](std::vector<char> *code) {
                                  std::string dense_nt, dense_pl, small, tail;
                                  for (int k = 0; k < n_operands; ++k) {
                                      const std::string K = std::to_string(k);
                                      if (leaves[k].kind == 0) {
                                          dense_nt += "            v[" + K + "] = smhip_ld((const V*)in.o[" + K + "].p + i, 1);\n";
                                          dense_pl += "            v[" + K + "] = smhip_ld((const V*)in.o[" + K + "].p + i, 0);\n";
                                          tail += "            x[" + K + "] = in.o[" + K + "].p[j];\n";
                                      } else {
                                          const char *f = leaves[k].kind == 1 ? "smhip_row" : "smhip_splat";
                                          small += std::string("        v[") + K + "] = " + f + "(in.o[" + K + "], (unsigned)i);\n";
                                          if (leaves[k].kind == 1) tail += "            x[" + K + "] = in.o[" + K + "].p[(unsigned long long)in.o[" + K + "].a.mod(in.o[" + K + "].off + (unsigned)n_vec) * WIDTH + (j - n_vec * WIDTH)];\n";
                                          else tail += "            x[" + K + "] = smhip_splat(in.o[" + K + "], (unsigned)n_vec)[j - n_vec * WIDTH];\n";
                                      }
                                  }
                                  for (int k = n_operands; k < 8; ++k) {
                                      small += "        v[" + std::to_string(k) + "] = v[0];\n";
                                      tail += "            x[" + std::to_string(k) + "] = x[0];\n";
                                  }
                                  std::string source = kExprBcastHead;
                                  source += "extern \"C\" __global__ __launch_bounds__(256) void smhip_user_expr_b(Operands in, Scalars sc, T* __restrict__ out,\n"
                                            "        unsigned long long n_vec, unsigned long long n, int pol) {\n"
                                            "    const unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x;\n"
                                            "    if (i < n_vec) {\n        V0 v[8];\n        if (pol & 1) {\n" + dense_nt + "        } else {\n" + dense_pl + "        }\n" + small +
                                            "        V0 r;\n        for (int e = 0; e < WIDTH; ++e) r[e] = smhip_eval(v[0][e], v[1][e], v[2][e], v[3][e], v[4][e], v[5][e], v[6][e], v[7][e], sc.v[0], sc.v[1], sc.v[2], sc.v[3]);\n"
                                            "        smhip_st((V*)out + i, r, pol);\n    } else if (i == n_vec) {\n        for (unsigned long long j = n_vec * WIDTH; j < n; ++j) {\n            T x[8];\n" + tail +
                                            "            out[j] = smhip_eval(x[0], x[1], x[2], x[3], x[4], x[5], x[6], x[7], sc.v[0], sc.v[1], sc.v[2], sc.v[3]);\n        }\n    }\n}\n";
                                  return hiprtc_build(source, {std::string("-DTYPE=") + kTypeName[dtype], std::string("-DWIDTH=") + (w == 2 ? "2" : "4"), std::string("-DEXPR=") + expr},
                                                      kTypeName[dtype], expr, code);
                              },
                              "smhip_user_expr_b", &fn))
        return rc;
    struct FD { unsigned d, mul, shr; };
    auto fastdiv = [](uint64_t div) {
        const dev::FastDiv f((uint32_t)div);
        return FD{f.d, f.mul, f.shr};
    };
    struct BOperand { const void *p; FD a, b; unsigned off, q0, elem, pad; };
    struct { BOperand o[8]; } in;
    memset(&in, 0, sizeof in);
    const size_t esz = dtype_size(dtype);
    unsigned char sc[32] = {};
    if (n_scalars > 0) memcpy(sc, scalars_host, (size_t)n_scalars * esz);
    const size_t n_vec = n / w;
    int pol = stream_policy((size_t)n_dense * n * esz, n * esz);
    for (int k = 0; k < n_operands; ++k)
        if (leaves[k].kind == 0) pol |= refine_policy(pol, {{leaves[k].ptr, n * esz}}, {nullptr, 0});
    // pieces: the large-array rule of the streaming kernels; with row / splat operands a launch also stays within the
    // 32-bit index arithmetic's reach
    size_t piece = piece_for(n_vec, n_dense + 1 < 3 ? n_dense + 1 : 3);
    const size_t reach = ((size_t)1 << 31) / w / 2;
    if (n_dense < n_operands && (piece == 0 || piece > reach) && n_vec > reach) piece = reach;
    if (piece == 0 || piece >= n_vec) piece = n_vec ? n_vec : 1;
    for (size_t v0 = 0;; v0 += piece) {
        const bool last = v0 + piece >= n_vec;
        unsigned long long nv = last ? n_vec - v0 : piece, ne = last ? n - v0 * w : piece * w;
        for (int k = 0; k < 8; ++k) {
            const ExprLeaf &lf = leaves[k < n_operands ? k : 0];
            BOperand &o = in.o[k];
            o = BOperand{};
            o.a = o.b = FD{1, 0, 0};
            if (lf.kind == 0) {
                o.p = static_cast<const char *>(lf.ptr) + v0 * 16;
            } else if (lf.kind == 1) {
                const uint64_t pv = lf.P / w;
                o.p = lf.ptr;
                o.a = fastdiv(pv);
                o.off = (unsigned)(v0 % pv);
            } else {
                const bool elem = lf.R % w != 0;
                const uint64_t r = elem ? lf.R : lf.R / w, at = elem ? (uint64_t)v0 * w : (uint64_t)v0;
                o.p = lf.ptr;
                o.a = fastdiv(r);
                o.b = fastdiv(lf.C);
                o.off = (unsigned)(at % r);
                o.q0 = (unsigned)((at / r) % lf.C);
                o.elem = elem;
            }
        }
        void *po = static_cast<char *>(out) + v0 * 16;
        const size_t grid = (nv + 1 + 255) / 256;
        if (grid > 0x7fffffffu) return fail(SMHIP_ERR_UNSUPPORTED, "fused expression: array too large for one launch");
        void *args[] = {&in, sc, &po, &nv, &ne, &pol};
        SMHIP_TRY(hipModuleLaunchKernel(fn, (unsigned)grid, 1, 1, 256, 1, 1, 0, s, args, nullptr));
        if (last) break;
    }
    return SMHIP_OK;
}

}  // namespace smhip
