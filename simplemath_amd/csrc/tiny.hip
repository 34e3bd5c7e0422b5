// tiny.hip -- operators on tiny arrays, several to a launch.
//
// The reference's smallest benchmarks apply ONE operator to a 5 x 5 or a 10-element array (benchmark/add.cpp:4-19,
// benchmark/pow.cpp:5-28): 0.2-0.6 us on a CPU core.  A kernel launch costs the host ~2.5-3.5 us whatever the kernel does
// (profiles/r04_small_array_breakdown.txt: empty kernels back to back, any-order or not), so one launch per operator can
// never come near that -- but nothing says an operator must be a launch of its own.  The library's calls are asynchronous
// already: the caller sees a result only through another library call (a read-back, a synchronisation, an operator that
// consumes it, the stream handle).  So an eligible tiny operator is RECORDED -- its descriptor and, for host-built
// operands, their bytes, appended to a per-device block -- and the block goes out as ONE launch (one workgroup per
// operator) when
//   * it is full (30 operators or ~3.9 KiB of descriptors),
//   * a new tiny operator would tie two LONG lists together: independent operators run side by side, one workgroup each; an
//     operator that depends on recorded ones (reads or overwrites what they write, overwrites what they read) is appended to
//     THEIR workgroup's list, which runs in call order with a __syncthreads() between operators -- `c = a + b; d = c * c;
//     e = d - a` is one launch of one workgroup; one that depends on two lists joins them into one (they do not depend on
//     each other, so either order is call order) up to 12 operators; beyond that what is recorded goes out first,
//   * ANY other library call on the device acquires the stream (operators, uploads, read-backs, synchronisation, events,
//     the stream handle, peer copies): runtime.hip's acquire_stream() flushes first, so every observation point sees
//     what call order promises.
// A buffer freed while a recorded operator still refers to it (the benchmark bodies' `auto result = ...` dies at the end
// of each iteration) is handed back to the pool only after the launch: the next result gets another block, and the
// operators stay independent.  Eligible: + - * / of every element type and integer pow (Op::apply -- the very functions
// the vector kernels' scalar tails use, so results are bit-identical to the one-launch path), at most 4096 results,
// operands of any strides (views, broadcasts) or host-built (<= 256 bytes), the output overlapping no operand, the
// library's own queue (not a caller's stream).  SMHIP_TINY_BATCH=0 turns it off.
#include <string.h>

#include <atomic>
#include <mutex>
#include <vector>

#include "internal.h"
#include "ops.hip.h"

namespace smhip {
namespace {

using namespace dev;

constexpr int kTinyMaxOps = 30;  // per launch; every one may start a list of its own
constexpr uint32_t kTinyMaxOut = kTinyMaxResults;  // internal.h: 4096
constexpr int kTinyThreads = 256, kTinyThreadsWide = 1024;  // per workgroup; the wide form when an operator of the launch has more than 1024 results (a 64 x 64 array: four passes)
constexpr int kTinyMaxMerged = 12;  // two lists are run as one only up to this many operators together
constexpr size_t kTinyMaxInline = 256;
constexpr int kTinySmallBytes = 1008, kTinyBigBytes = 3904;  // argument blocks of 1 KiB and ~3.9 KiB (the launch writes the block it is given)
constexpr int kTinyDevices = 64;

struct TinyOp {
    uint8_t op, dtype, ndim, flags;  // flags bit 0: b is `scalar` (array_scalar)
    uint32_t n;
    uint32_t shape[SMHIP_MAX_NDIM];  // innermost first
    uint32_t sa[SMHIP_MAX_NDIM], sb[SMHIP_MAX_NDIM];
    uint64_t a, b, out;
    uint64_t scalar;
    uint16_t a_inl, b_inl;  // byte offset of the operand's bytes from this descriptor's start; 0: `a` / `b` is a device pointer
    uint16_t next;          // 1 + the byte offset (in the block) of the next operator of this workgroup's list; 0: the last one
    uint16_t pad16;
    uint32_t pad[2];
};
static_assert(sizeof(TinyOp) == 128, "descriptor layout");

// The argument block as launched: the FIRST operator of list g sits at bytes + g * sizeof(TinyOp), so a workgroup finds its
// work without reading an offset table first -- the argument block is host memory, every dependent read of it a trip over the
// fabric (~1.3 us, inline.hip), and one operator waited for at once pays for each of them.  The other operators and the
// host-built operands' bytes follow; `next`, `a_inl`, `b_inl` are laid out for this order by flush_locked().
template <int BYTES> struct alignas(16) TinyArgs {
    uint32_t n_lists;
    uint32_t pad[3];
    unsigned char bytes[BYTES];
};
static_assert(offsetof(TinyArgs<kTinySmallBytes>, bytes) == 16 && sizeof(TinyArgs<kTinySmallBytes>) == kTinySmallBytes + 16, "argument block layout");
static_assert(kTinyBigBytes + 1 < 65536 && kTinyMaxOps < 256, "offsets and list numbers fit their fields");
static_assert(sizeof(TinyArgs<kTinyBigBytes>) <= 4096, "one argument block");

template <typename T, typename Op>
__device__ __forceinline__ void tiny_run(const TinyOp *d, const char *base) {
    // the output overlaps no operand (tiny_try_enqueue): the passes' loads need not wait for each other's stores
    const T *__restrict__ a = d->a_inl ? reinterpret_cast<const T *>(base + d->a_inl) : reinterpret_cast<const T *>(d->a);
    const T *__restrict__ b = d->b_inl ? reinterpret_cast<const T *>(base + d->b_inl) : reinterpret_cast<const T *>(d->b);
    T *__restrict__ out = reinterpret_cast<T *>(d->out);
    const bool b_scalar = d->flags & 1;
    T sv;
    {
        const uint64_t bits = d->scalar;
        __builtin_memcpy(&sv, &bits, sizeof(T));
    }
    const int nd = d->ndim;
    const uint32_t n = d->n;
    for (uint32_t e = threadIdx.x; e < n; e += blockDim.x) {
        uint32_t rem = e, oa = 0, ob = 0;
        for (int k = 0; k < nd; ++k) {
            const uint32_t ext = d->shape[k];
            const uint32_t idx = k == nd - 1 ? rem : rem % ext;
            rem /= ext;
            oa += idx * d->sa[k];
            ob += idx * d->sb[k];
        }
        out[e] = Op::apply(a[oa], b_scalar ? sv : b[ob]);
    }
}
constexpr int kTinyFill = 250, kTinyCopy = 251;  // descriptor op codes next to SMHIP_OP_*: out[e] = scalar; out[sum idx * sb] = a[sum idx * sa]
template <typename T>
__device__ __forceinline__ void tiny_move(const TinyOp *d, const char *base) {
    const T *__restrict__ a = d->a_inl ? reinterpret_cast<const T *>(base + d->a_inl) : reinterpret_cast<const T *>(d->a);
    T *__restrict__ out = reinterpret_cast<T *>(d->out);
    T sv;
    {
        const uint64_t bits = d->scalar;
        __builtin_memcpy(&sv, &bits, sizeof(T));
    }
    const bool fill = d->op == kTinyFill;
    const int nd = d->ndim;
    const uint32_t n = d->n;
    for (uint32_t e = threadIdx.x; e < n; e += blockDim.x) {
        uint32_t rem = e, oa = 0, od = 0;
        for (int k = 0; k < nd; ++k) {
            const uint32_t ext = d->shape[k];
            const uint32_t idx = k == nd - 1 ? rem : rem % ext;
            rem /= ext;
            oa += idx * d->sa[k];
            od += idx * d->sb[k];
        }
        out[od] = fill ? sv : a[oa];
    }
}
template <typename T>
__device__ __forceinline__ void tiny_dtype(const TinyOp *d, const char *base) {
    switch (d->op) {
        case kTinyFill:
        case kTinyCopy: tiny_move<T>(d, base); break;
        case SMHIP_OP_ADD: tiny_run<T, AddOp<T>>(d, base); break;
        case SMHIP_OP_SUB: tiny_run<T, SubtractOp<T>>(d, base); break;
        case SMHIP_OP_MUL: tiny_run<T, MultiplyOp<T>>(d, base); break;
        case SMHIP_OP_DIV: tiny_run<T, DivideOp<T>>(d, base); break;
        default:
            if constexpr (std::is_integral<T>::value) tiny_run<T, PowOp<T>>(d, base);
            break;
    }
}
// One workgroup (four waves) per LIST of recorded operators (a list: operators that depend on each other, in call order); the
// descriptors are read from the argument block itself.  __syncthreads() between two operators of a list makes the first one's
// stores visible to every lane of the workgroup before the second one loads (the barrier's release / acquire at workgroup scope).
template <int BYTES>
__global__ __launch_bounds__(kTinyThreadsWide) void tiny_batch_kernel(TinyArgs<BYTES> args) {
    (void)args;  // read through the kernarg pointer: indexing the by-value copy would spill it to scratch (inline.hip)
    const char *ka = (const char *)__builtin_amdgcn_kernarg_segment_ptr();
    uint32_t off = blockIdx.x * (uint32_t)sizeof(TinyOp);
    for (;;) {
        const char *base = ka + offsetof(TinyArgs<BYTES>, bytes) + off;
        const TinyOp *d = reinterpret_cast<const TinyOp *>(base);
        switch (d->dtype) {
            case SMHIP_F32: tiny_dtype<float>(d, base); break;
            case SMHIP_F64: tiny_dtype<double>(d, base); break;
            case SMHIP_I32: tiny_dtype<int32_t>(d, base); break;
            default: tiny_dtype<int64_t>(d, base); break;
        }
        const uint32_t next = d->next;  // the same for every lane
        if (next == 0) break;
        __syncthreads();
        off = next - 1;
    }
}

struct TinyQueue {
    std::recursive_mutex m;
    int count = 0, lists = 0;
    uint32_t widest = 0;  // the largest result count recorded
    size_t used = 0;
    uint16_t head[kTinyMaxOps], tail[kTinyMaxOps], list_len[kTinyMaxOps];  // per list: byte offsets of its first and last operator, operators in it
    uint16_t op_off[kTinyMaxOps], op_len[kTinyMaxOps];  // per operator, in recording order: where its descriptor starts, descriptor + operand bytes
    alignas(16) unsigned char bytes[kTinyBigBytes];
    Span reads[2 * kTinyMaxOps], writes[kTinyMaxOps];
    unsigned char read_list[2 * kTinyMaxOps], write_list[kTinyMaxOps];  // which list the span's operator belongs to
    int n_reads = 0;
    std::vector<void *> deferred;  // freed while recorded operators refer to them
    bool flushing = false;
    unsigned long long launches = 0, operators = 0, appended = 0, merged = 0;  // appended: operators that joined the list of one they depend on; merged: lists joined by one that depends on both
};
TinyQueue g_tiny[kTinyDevices];
std::atomic<int> g_tiny_pending{0};

bool overlap(const Span &x, const Span &y) {
    if (!x.p || !y.p || !x.bytes || !y.bytes) return false;
    const uintptr_t a = reinterpret_cast<uintptr_t>(x.p), b = reinterpret_cast<uintptr_t>(y.p);
    return a < b + y.bytes && b < a + x.bytes;
}

// The recorded block in launch order: list heads first, one descriptor slot each, then everything else (TinyArgs).
template <int BYTES>
void lay_out(const TinyQueue &q, TinyArgs<BYTES> &args) {
    args.n_lists = (uint32_t)q.lists;
    const int n = q.count;
    auto index_of = [&](uint32_t off) { for (int i = 0; i < n; ++i) if (q.op_off[i] == off) return i; return -1; };
    uint16_t at[kTinyMaxOps], payload[kTinyMaxOps];  // new descriptor offsets; where a head's operand bytes go
    bool is_head[kTinyMaxOps] = {};
    for (int l = 0; l < q.lists; ++l) {
        const int i = index_of(q.head[l]);
        at[i] = (uint16_t)(l * sizeof(TinyOp));
        is_head[i] = true;
    }
    size_t tail = (size_t)q.lists * sizeof(TinyOp);
    for (int i = 0; i < n; ++i) {
        if (is_head[i]) { payload[i] = (uint16_t)tail; tail += q.op_len[i] - sizeof(TinyOp); }
        else { at[i] = (uint16_t)tail; payload[i] = (uint16_t)(tail + sizeof(TinyOp)); tail += q.op_len[i]; }
    }
    for (int i = 0; i < n; ++i) {
        TinyOp d;
        memcpy(&d, q.bytes + q.op_off[i], sizeof d);
        if (d.next) d.next = (uint16_t)(at[index_of(d.next - 1u)] + 1);
        const int shift = (int)payload[i] - (int)at[i] - (int)sizeof(TinyOp);  // how far the operand bytes moved against the descriptor
        if (d.a_inl) d.a_inl = (uint16_t)(d.a_inl + shift);
        if (d.b_inl) d.b_inl = (uint16_t)(d.b_inl + shift);
        memcpy(args.bytes + at[i], &d, sizeof d);
        if (q.op_len[i] > sizeof(TinyOp)) memcpy(args.bytes + payload[i], q.bytes + q.op_off[i] + sizeof(TinyOp), q.op_len[i] - sizeof(TinyOp));
    }
}

// Launches what is recorded.  Caller holds q.m.
int flush_locked(TinyQueue &q) {
    if (q.count == 0 || q.flushing) return SMHIP_OK;
    q.flushing = true;  // the launch below acquires the stream, whose hook comes back here
    int rc = SMHIP_OK;
    {
        hipStream_t s;
        OpScope scope;
        rc = scope.begin_barrier(&s);
        if (rc == SMHIP_OK) {
            if (q.used <= (size_t)kTinySmallBytes) {
                TinyArgs<kTinySmallBytes> args;
                lay_out(q, args);
                hipLaunchKernelGGL((tiny_batch_kernel<kTinySmallBytes>), dim3((unsigned)q.lists), dim3(q.widest > 1024 ? kTinyThreadsWide : kTinyThreads), 0, s, args);
            } else {
                TinyArgs<kTinyBigBytes> args;
                lay_out(q, args);
                hipLaunchKernelGGL((tiny_batch_kernel<kTinyBigBytes>), dim3((unsigned)q.lists), dim3(q.widest > 1024 ? kTinyThreadsWide : kTinyThreads), 0, s, args);
            }
            const hipError_t e = hipGetLastError();
            if (e != hipSuccess) rc = fail(SMHIP_ERR_HIP, "tiny operators (%d in one launch): %s", q.count, hipGetErrorString(e));
        }
    }
    ++q.launches;
    q.operators += (unsigned long long)q.count;
    g_tiny_pending.fetch_sub(q.count, std::memory_order_relaxed);
    q.count = 0;
    q.lists = 0;
    q.widest = 0;
    q.used = 0;
    q.n_reads = 0;
    std::vector<void *> dead;
    dead.swap(q.deferred);
    q.flushing = false;
    for (void *p : dead) (void)smhip_free(p);  // ordered behind the launch like any free on the library's stream
    return rc;
}

bool enabled() {
    static const bool on = [] { const char *e = getenv("SMHIP_TINY_BATCH"); return !(e && *e && atoi(e) == 0); }();
    return on;
}

}  // namespace

bool tiny_any_recorded() { return g_tiny_pending.load(std::memory_order_relaxed) != 0; }

int tiny_flush_device(int dev) {
    if (dev < 0 || dev >= kTinyDevices) return SMHIP_OK;
    if (g_tiny_pending.load(std::memory_order_relaxed) == 0) return SMHIP_OK;
    TinyQueue &q = g_tiny[dev];
    std::lock_guard<std::recursive_mutex> lock(q.m);
    return flush_locked(q);
}

bool tiny_defer_free(int dev, void *p, size_t bytes) {
    if (dev < 0 || dev >= kTinyDevices || g_tiny_pending.load(std::memory_order_relaxed) == 0) return false;
    TinyQueue &q = g_tiny[dev];
    std::lock_guard<std::recursive_mutex> lock(q.m);
    if (q.count == 0 || q.flushing) return false;
    const Span blk{p, bytes};
    bool used = false;
    for (int i = 0; i < q.count && !used; ++i) used = overlap(blk, q.writes[i]);
    for (int i = 0; i < q.n_reads && !used; ++i) used = overlap(blk, q.reads[i]);
    if (!used) return false;
    q.deferred.push_back(p);
    return true;
}

void tiny_stats(int dev, unsigned long long *launches, unsigned long long *operators) {
    TinyQueue &q = g_tiny[dev < 0 || dev >= kTinyDevices ? 0 : dev];
    std::lock_guard<std::recursive_mutex> lock(q.m);
    if (launches) *launches = q.launches;
    if (operators) *operators = q.operators;
}

// Records `out = a op b` (b an array, or the value at scalar_host) if it is eligible; *taken says whether it was.
// kind 1: a fill, out[e] = the value at scalar_host (a, b unused).  kind 2: a copy, out[sum idx * sb] = a[sum idx * sa] -- sb are
// the DESTINATION's strides (an assignment into a view, a dense device-to-device copy, an upload whose bytes ride as `a`).
int tiny_try_enqueue(int op, int dtype, const void *a, size_t a_host_bytes, const int64_t *sa, const void *b, size_t b_host_bytes,
                     const int64_t *sb, const int64_t *shape, int ndim, const void *scalar_host, void *out, bool *taken, int kind) {
    *taken = false;
    if (!enabled()) return SMHIP_OK;
    int dev;
    if (!tiny_context(&dev) || dev >= kTinyDevices) return SMHIP_OK;
    const bool integral = dtype == SMHIP_I32 || dtype == SMHIP_I64;
    if (kind == 1) op = kTinyFill;
    else if (kind == 2) op = kTinyCopy;
    else if (!(op == SMHIP_OP_ADD || op == SMHIP_OP_SUB || op == SMHIP_OP_MUL || op == SMHIP_OP_DIV || (op == SMHIP_OP_POW && integral))) return SMHIP_OK;
    if (ndim < 1 || ndim > SMHIP_MAX_NDIM || a_host_bytes > kTinyMaxInline || b_host_bytes > kTinyMaxInline) return SMHIP_OK;
    const size_t esz = dtype_size(dtype);
    uint64_t n = 1, span_a = 0, span_b = 0;
    for (int i = 0; i < ndim; ++i) {
        if (shape[i] <= 0) return SMHIP_OK;
        n *= (uint64_t)shape[i];
        if (n > kTinyMaxOut) return SMHIP_OK;
        if (kind != 1) span_a += (uint64_t)(shape[i] - 1) * (uint64_t)sa[i];
        if (!scalar_host) span_b += (uint64_t)(shape[i] - 1) * (uint64_t)sb[i];
        if (span_a >= (1ull << 31) || span_b >= (1ull << 31)) return SMHIP_OK;
    }
    const Span w{out, kind == 2 ? (size_t)(span_b + 1) * esz : (size_t)n * esz};
    const Span ra{(a_host_bytes || kind == 1) ? nullptr : a, (a_host_bytes || kind == 1) ? 0 : (size_t)(span_a + 1) * esz};
    const Span rb{(b_host_bytes || scalar_host || kind != 0) ? nullptr : b, (b_host_bytes || scalar_host || kind != 0) ? 0 : (size_t)(span_b + 1) * esz};
    if (overlap(w, ra) || overlap(w, rb)) return SMHIP_OK;  // in place: the one-launch path has its own rules for that
    const bool same_inline = a_host_bytes && a == b && a_host_bytes == b_host_bytes;  // `ac + ac` on a host-built array: its bytes go in once
    const size_t need = sizeof(TinyOp) + ((a_host_bytes + 15) & ~(size_t)15) + (same_inline ? 0 : ((b_host_bytes + 15) & ~(size_t)15));

    TinyQueue &q = g_tiny[dev];
    std::lock_guard<std::recursive_mutex> lock(q.m);
    if (q.flushing) return SMHIP_OK;
    // Which recorded operators must this one come after?  None: a list of its own.  Those of ONE list: it joins that list.
    // Those of two lists: what is recorded goes out first.
    int list = -1;
    bool two = false;
    auto after = [&](int l) {
        if (list < 0 || list == l) { list = l; return; }
        // a second list: the two do not depend on each other, so one workgroup may run them one after the other -- `a` and `b`
        // uploaded (two lists), then a + b -- unless that would make a long serial run of what could have run side by side
        if (q.list_len[list] + q.list_len[l] > kTinyMaxMerged) { two = true; return; }
        const int keep = list < l ? list : l, drop = list < l ? l : list;
        const uint16_t link = (uint16_t)(q.head[drop] + 1);
        memcpy(q.bytes + q.tail[keep] + offsetof(TinyOp, next), &link, sizeof link);
        q.tail[keep] = q.tail[drop];
        q.list_len[keep] = (uint16_t)(q.list_len[keep] + q.list_len[drop]);
        const int last = q.lists - 1;  // `drop` takes the last list's number
        for (int i = 0; i < q.count; ++i) q.write_list[i] = q.write_list[i] == drop ? (unsigned char)keep : (q.write_list[i] == last ? (unsigned char)drop : q.write_list[i]);
        for (int i = 0; i < q.n_reads; ++i) q.read_list[i] = q.read_list[i] == drop ? (unsigned char)keep : (q.read_list[i] == last ? (unsigned char)drop : q.read_list[i]);
        if (drop != last) { q.head[drop] = q.head[last]; q.tail[drop] = q.tail[last]; q.list_len[drop] = q.list_len[last]; }
        --q.lists;
        ++q.merged;
        list = keep;
    };
    for (int i = 0; i < q.count && !two; ++i)
        if (overlap(w, q.writes[i]) || overlap(ra, q.writes[i]) || overlap(rb, q.writes[i])) after(q.write_list[i]);  // WAW, RAW
    for (int i = 0; i < q.n_reads && !two; ++i)
        if (overlap(w, q.reads[i])) after(q.read_list[i]);  // WAR
    if (two || q.count == kTinyMaxOps || q.used + need > (size_t)kTinyBigBytes) {
        if (int rc = flush_locked(q)) return rc;
        list = -1;
    }
    TinyOp d{};
    d.op = (uint8_t)op;
    d.dtype = (uint8_t)dtype;
    d.ndim = (uint8_t)ndim;
    d.flags = scalar_host ? 1 : 0;
    d.n = (uint32_t)n;
    for (int k = 0; k < ndim; ++k) {
        const int src = ndim - 1 - k;
        d.shape[k] = (uint32_t)shape[src];
        d.sa[k] = kind == 1 ? 0u : (uint32_t)sa[src];
        d.sb[k] = (scalar_host && kind != 1) ? 0u : (uint32_t)sb[src];
    }
    d.a = reinterpret_cast<uint64_t>(a);
    d.b = reinterpret_cast<uint64_t>(b);
    d.out = reinterpret_cast<uint64_t>(out);
    if (scalar_host) memcpy(&d.scalar, scalar_host, esz);
    unsigned char *at = q.bytes + q.used;
    size_t extra = sizeof(TinyOp);
    if (a_host_bytes) {
        d.a_inl = (uint16_t)extra;
        memcpy(at + extra, a, a_host_bytes);
        extra += (a_host_bytes + 15) & ~(size_t)15;
    }
    if (same_inline) {
        d.b_inl = d.a_inl;
    } else if (b_host_bytes) {
        d.b_inl = (uint16_t)extra;
        memcpy(at + extra, b, b_host_bytes);
        extra += (b_host_bytes + 15) & ~(size_t)15;
    }
    memcpy(at, &d, sizeof d);
    if (list < 0) {
        list = q.lists++;
        q.head[list] = (uint16_t)q.used;
        q.list_len[list] = 0;
    } else {  // behind the last operator of the list it depends on
        const uint16_t link = (uint16_t)(q.used + 1);
        memcpy(q.bytes + q.tail[list] + offsetof(TinyOp, next), &link, sizeof link);
        ++q.appended;
    }
    q.tail[list] = (uint16_t)q.used;
    q.op_off[q.count] = (uint16_t)q.used;
    q.op_len[q.count] = (uint16_t)extra;
    ++q.list_len[list];
    q.writes[q.count] = w;
    q.write_list[q.count] = (unsigned char)list;
    if (ra.p) { q.reads[q.n_reads] = ra; q.read_list[q.n_reads++] = (unsigned char)list; }
    if (rb.p) { q.reads[q.n_reads] = rb; q.read_list[q.n_reads++] = (unsigned char)list; }
    q.used += extra;
    if ((uint32_t)n > q.widest) q.widest = (uint32_t)n;
    ++q.count;
    g_tiny_pending.fetch_add(1, std::memory_order_relaxed);
    *taken = true;
    return SMHIP_OK;
}

}  // namespace smhip
