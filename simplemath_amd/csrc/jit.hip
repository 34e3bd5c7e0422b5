// jit.hip -- user-defined Ops on the device, compiled at run time with hipRTC.
//
// The reference's extension recipe (README.md:86-133) is "write a struct with
// apply() and an x86 apply_simd<__m256> specialisation, then call
// element_wise_op<T, MyOp<T>>".  An x86 intrinsic body means nothing on gfx950,
// and a g++-compiled `apply` cannot run there either -- so the device-side
// plugin contract is one string: the Op's arithmetic as a HIP expression in
// `a` and `b` (e.g. "(a + b) * 2").  smhip_register_op() hands back an op id;
// the first use per element type compiles three kernels around the expression
// for gfx950 (hiprtc -> code object -> hipModule) and caches them:
//     contig   out[i] = a[i] op b[i]          16-byte vectors, one per lane, nt
//     scalar   out[i] = a[i] op s  /  s op a[i]
//     gather   the general broadcast form (fast-division unravel, vector stores)
// The built-in Ops never come through here (they are AOT kernels).
#include <hip/hiprtc.h>
#include <string.h>

#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "internal.h"
#include "ops.hip.h"

namespace smhip {
namespace {

struct UserOp {
    std::string expr;
    hipModule_t module[4] = {};  // per dtype
    hipFunction_t contig[4] = {}, scalar[4] = {}, gather[4] = {};
};
std::mutex g_jit_mutex;
std::vector<UserOp *> g_ops;  // id = SMHIP_OP_USER_BASE + index; never freed (ids stay valid)

const char *kTypeName[4] = {"float", "double", "int", "long long"};

// The kernels' source.  TYPE, WIDTH and EXPR are -D defines.
const char *kSource = R"SRC(
typedef TYPE T;
typedef T V0 __attribute__((ext_vector_type(WIDTH)));
typedef V0 V __attribute__((aligned(sizeof(T))));   // element-aligned 16-byte accesses are legal on gfx950
struct UserOp { static __device__ __forceinline__ T apply(T a, T b) { return (T)(EXPR); } };

extern "C" __global__ __launch_bounds__(256) void smhip_user_contig(const T* __restrict__ a, const T* __restrict__ b,
                                                                     T* __restrict__ out, unsigned long long n_vec,
                                                                     unsigned long long n, int vec) {
    const unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    if (vec) {
        if (i < n_vec) {
            const V va = __builtin_nontemporal_load((const V*)a + i), vb = __builtin_nontemporal_load((const V*)b + i);
            V r;
            for (int k = 0; k < WIDTH; ++k) r[k] = UserOp::apply(va[k], vb[k]);
            __builtin_nontemporal_store(r, (V*)out + i);
        } else if (i == n_vec) {
            for (unsigned long long k = n_vec * WIDTH; k < n; ++k) out[k] = UserOp::apply(a[k], b[k]);
        }
    } else if (i < n) {
        out[i] = UserOp::apply(a[i], b[i]);
    }
}

extern "C" __global__ __launch_bounds__(256) void smhip_user_scalar(const T* __restrict__ a, T s, T* __restrict__ out,
                                                                     unsigned long long n, int swapped) {
    const unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = swapped ? UserOp::apply(s, a[i]) : UserOp::apply(a[i], s);
}

struct GatherParams {
    long long sa[6], sb[6];      // innermost first
    unsigned d[6], mul[6], shr[6];
    int ndim;
    unsigned n;
};
extern "C" __global__ __launch_bounds__(256) void smhip_user_gather(const T* __restrict__ a, const T* __restrict__ b,
                                                                     T* __restrict__ out, GatherParams p) {
    const unsigned linear = blockIdx.x * 256u + threadIdx.x;
    if (linear >= p.n) return;
    unsigned rem = linear;
    long long offA = 0, offB = 0;
    for (int k = 0; k < p.ndim; ++k) {
        const unsigned q = p.d[k] == 1 ? rem : (__umulhi(rem, p.mul[k]) >> p.shr[k]);
        const unsigned idx = rem - q * p.d[k];
        rem = q;
        offA += (long long)idx * p.sa[k];
        offB += (long long)idx * p.sb[k];
    }
    out[linear] = UserOp::apply(a[offA], b[offB]);
}
)SRC";

struct GatherParamsHost {
    long long sa[6], sb[6];
    unsigned d[6], mul[6], shr[6];
    int ndim;
    unsigned n;
};

int compile(UserOp &op, int dtype) {
    if (op.module[dtype]) return SMHIP_OK;
    hiprtcProgram prog;
    if (hiprtcCreateProgram(&prog, kSource, "smhip_user_op.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS)
        return fail(SMHIP_ERR_HIP, "hiprtcCreateProgram failed");
    const std::string dtype_def = std::string("-DTYPE=") + kTypeName[dtype];
    const std::string width_def = std::string("-DWIDTH=") + ((dtype == SMHIP_F64 || dtype == SMHIP_I64) ? "2" : "4");
    const std::string expr_def = "-DEXPR=" + op.expr;
    const char *opts[] = {"--offload-arch=gfx950", "-O3", "-ffp-contract=off", dtype_def.c_str(), width_def.c_str(), expr_def.c_str()};
    const hiprtcResult rc = hiprtcCompileProgram(prog, 6, opts);
    if (rc != HIPRTC_SUCCESS) {
        size_t n = 0;
        hiprtcGetProgramLogSize(prog, &n);
        std::string log(n, '\0');
        if (n) hiprtcGetProgramLog(prog, &log[0]);
        hiprtcDestroyProgram(&prog);
        return fail(SMHIP_ERR_INVALID, "user op \"%s\" does not compile for %s: %.300s", op.expr.c_str(), kTypeName[dtype], log.c_str());
    }
    size_t size = 0;
    hiprtcGetCodeSize(prog, &size);
    std::vector<char> code(size);
    hiprtcGetCode(prog, code.data());
    hiprtcDestroyProgram(&prog);
    hipModule_t mod;
    SMHIP_TRY(hipModuleLoadData(&mod, code.data()));
    SMHIP_TRY(hipModuleGetFunction(&op.contig[dtype], mod, "smhip_user_contig"));
    SMHIP_TRY(hipModuleGetFunction(&op.scalar[dtype], mod, "smhip_user_scalar"));
    SMHIP_TRY(hipModuleGetFunction(&op.gather[dtype], mod, "smhip_user_gather"));
    op.module[dtype] = mod;
    return SMHIP_OK;
}

int lookup(int op, int dtype, UserOp **out) {
    std::lock_guard<std::mutex> lock(g_jit_mutex);
    const int idx = op - SMHIP_OP_USER_BASE;
    if (idx < 0 || idx >= (int)g_ops.size()) return fail(SMHIP_ERR_INVALID, "op %d was never registered", op);
    *out = g_ops[idx];
    return compile(**out, dtype);
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
    UserOp *u;
    if (int rc = lookup(op, dtype, &u)) return rc;
    const unsigned long long w = (dtype == SMHIP_F64 || dtype == SMHIP_I64) ? 2 : 4;
    int vec = 1;
    unsigned long long n_vec = n / w, nn = n;
    const size_t threads = vec ? n_vec + 1 : n;
    const size_t grid = (threads + 255) / 256;
    if (grid > 0x7fffffffu) return fail(SMHIP_ERR_UNSUPPORTED, "user op: array too large for one launch");
    void *args[] = {&a, &b, &out, &n_vec, &nn, &vec};
    SMHIP_TRY(hipModuleLaunchKernel(u->contig[dtype], (unsigned)grid, 1, 1, 256, 1, 1, 0, s, args, nullptr));
    return SMHIP_OK;
}

int jit_array_scalar(int op, int dtype, const void *a, const void *value_host, size_t n, void *out, hipStream_t s) {
    UserOp *u;
    if (int rc = lookup(op, dtype, &u)) return rc;
    unsigned long long nn = n;
    int swapped = 0;
    const size_t grid = (n + 255) / 256;
    if (grid > 0x7fffffffu) return fail(SMHIP_ERR_UNSUPPORTED, "user op: array too large for one launch");
    unsigned char scalar[8];
    memcpy(scalar, value_host, dtype_size(dtype));
    void *args[] = {&a, scalar, &out, &nn, &swapped};
    SMHIP_TRY(hipModuleLaunchKernel(u->scalar[dtype], (unsigned)grid, 1, 1, 256, 1, 1, 0, s, args, nullptr));
    return SMHIP_OK;
}

int jit_elementwise(int op, int dtype, const void *a, const int64_t *sa, const void *b, const int64_t *sb, const int64_t *shape,
                    int ndim, void *out, hipStream_t s) {
    size_t n = 1;
    bool dense = true;
    int64_t expect = 1;
    for (int i = ndim - 1; i >= 0; --i) {
        n *= (size_t)shape[i];
        if (shape[i] != 1 && (sa[i] != expect || sb[i] != expect)) dense = false;
        expect *= shape[i];
    }
    if (n == 0) return SMHIP_OK;
    if (dense) return jit_contiguous(op, dtype, a, b, out, n, s);
    if (n >= 0x7fffffffull) {  // cut along the first dimension with an extent: pieces of < 2^31 elements
        int d = 0;
        while (shape[d] == 1) ++d;
        const size_t esz = dtype_size(dtype), slice = n / (size_t)shape[d];
        const size_t per = slice >= 0x7fffffffull ? 1 : 0x7ffffffeull / slice;
        int64_t sub[SMHIP_MAX_NDIM];
        for (int i = 0; i < ndim; ++i) sub[i] = shape[i];
        for (size_t i0 = 0; i0 < (size_t)shape[d]; i0 += per) {
            const size_t left = (size_t)shape[d] - i0;
            sub[d] = (int64_t)(left < per ? left : per);
            if (int rc = jit_elementwise(op, dtype, static_cast<const char *>(a) + (int64_t)i0 * sa[d] * (int64_t)esz, sa,
                                         static_cast<const char *>(b) + (int64_t)i0 * sb[d] * (int64_t)esz, sb, sub, ndim,
                                         static_cast<char *>(out) + i0 * slice * esz, s))
                return rc;
        }
        return SMHIP_OK;
    }
    UserOp *u;
    if (int rc = lookup(op, dtype, &u)) return rc;
    GatherParamsHost p{};
    p.ndim = ndim;
    p.n = (unsigned)n;
    for (int k = 0; k < ndim; ++k) {
        const int src = ndim - 1 - k;
        const dev::FastDiv fd((uint32_t)shape[src]);
        p.d[k] = fd.d; p.mul[k] = fd.mul; p.shr[k] = fd.shr;
        p.sa[k] = sa[src]; p.sb[k] = sb[src];
    }
    void *args[] = {&a, &b, &out, &p};
    SMHIP_TRY(hipModuleLaunchKernel(u->gather[dtype], (unsigned)((n + 255) / 256), 1, 1, 256, 1, 1, 0, s, args, nullptr));
    return SMHIP_OK;
}

}  // namespace smhip
