// sharded.hip -- the hot path over the GPUs of one node.
//
// The reference's only fan-out is `#pragma omp parallel for` over 1024-element chunks of the output
// (include/math/calculate.h:47) and over SIMD iterations of array_scalar_op (:152): shared memory, one socket.  On a
// node of 8 MI355X the same independence is used one level up: the RESULT's outermost dimension is cut into one
// block per GPU, every GPU runs the single-device kernels on its block (operands broadcast along that dimension are
// replicated), and nothing crosses xGMI on the data path.  Only a whole-array reduction has an exchange step: each
// GPU's fp64 partial meets the others in ONE ncclAllReduce of one value -- latency-bound, so neither the 7 x 153 GB/s
// links nor ring-vs-tree matter for it.
//
// Two launch models, same kernels:
//   smhip_set_devices(n)        one process, one host thread walking the devices (ncclCommInitAll, one stream per
//                               device, the all-reduce inside ncclGroupStart/End);
//   smhip_comm_init_rank(...)   one process per GPU (ncclCommInitRank), the all-reduce on the rank's own stream.
// RCCL is dlopen'ed on first use: a single-GPU program never loads it.
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <mutex>

#include "internal.h"

namespace smhip {
namespace {

constexpr int kMaxGroup = 64;

struct Rccl {
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclGetVersion) GetVersion = nullptr;
    decltype(&ncclCommCount) CommCount = nullptr;
    decltype(&ncclCommUserRank) CommUserRank = nullptr;
    decltype(&ncclCommCuDevice) CommCuDevice = nullptr;
    decltype(&ncclCommAbort) CommAbort = nullptr;  // optional: CommDestroy if absent
};

std::mutex g_mutex;  // guards everything below
Rccl g_rccl;
bool g_rccl_loaded = false;

int g_ndev = 0;                        // devices 0..g_ndev-1 form the single-process group (0: none)
ncclComm_t g_comms[kMaxGroup] = {};    // from ncclCommInitAll, by device
void *g_slot[kMaxGroup] = {};          // 16 bytes of pooled device memory per device: the partial / the total

ncclComm_t g_rank_comm = nullptr;      // one-process-per-GPU communicator
int g_nranks = 0, g_rank = -1, g_rank_device = -1;

int load_rccl() {  // caller holds g_mutex
    if (g_rccl_loaded) return SMHIP_OK;
    void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!h) return fail(SMHIP_ERR_UNSUPPORTED, "multi-GPU needs RCCL and librccl.so.1 cannot be loaded: %s", dlerror());
#define SMHIP_SYM(field, name)                                                                         \
    g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(h, name));                            \
    if (!g_rccl.field) return fail(SMHIP_ERR_UNSUPPORTED, "librccl has no symbol %s", name)
    SMHIP_SYM(GetUniqueId, "ncclGetUniqueId");
    SMHIP_SYM(CommInitRank, "ncclCommInitRank");
    SMHIP_SYM(CommInitAll, "ncclCommInitAll");
    SMHIP_SYM(CommDestroy, "ncclCommDestroy");
    SMHIP_SYM(AllReduce, "ncclAllReduce");
    SMHIP_SYM(GroupStart, "ncclGroupStart");
    SMHIP_SYM(GroupEnd, "ncclGroupEnd");
    SMHIP_SYM(GetErrorString, "ncclGetErrorString");
    SMHIP_SYM(GetVersion, "ncclGetVersion");
    SMHIP_SYM(CommCount, "ncclCommCount");
    SMHIP_SYM(CommUserRank, "ncclCommUserRank");
    SMHIP_SYM(CommCuDevice, "ncclCommCuDevice");
#undef SMHIP_SYM
    g_rccl.CommAbort = reinterpret_cast<decltype(g_rccl.CommAbort)>(dlsym(h, "ncclCommAbort"));
    g_rccl_loaded = true;
    return SMHIP_OK;
}

// Failure injection for the test-suite.  Off unless the process has called smhip_enable_test_hooks(1): a stray environment
// variable in a job cannot abort its communicators, and a production call pays one relaxed load.
std::atomic<int> g_test_hooks{0};
// SMHIP_TEST_FAIL_COMM_INIT=1 makes smhip_set_devices behave as if ncclCommInitAll had failed: the error path's
// clean-up (slots back to the pool, no group left behind) is then testable on a one-GPU box.
bool test_hook_fail_comm_init() {
    if (!g_test_hooks.load(std::memory_order_relaxed)) return false;
    const char *e = getenv("SMHIP_TEST_FAIL_COMM_INIT");
    return e && e[0] == '1';
}

// SMHIP_TEST_FAIL_ALLREDUCE=<g>: the group all-reduce fails when it reaches device g (after devices 0..g-1 have queued
// their half) -- the "partly issued collective" path.
bool test_hook_fail_allreduce(int g) {
    if (!g_test_hooks.load(std::memory_order_relaxed)) return false;
    const char *e = getenv("SMHIP_TEST_FAIL_ALLREDUCE");
    return e && e[0] && atoi(e) == g;
}

#define SMHIP_NCCL(expr)                                                                                       \
    do {                                                                                                       \
        ncclResult_t nccl_r_ = (expr);                                                                         \
        if (nccl_r_ != ncclSuccess) return fail(SMHIP_ERR_HIP, "%s: %s", #expr, g_rccl.GetErrorString(nccl_r_)); \
    } while (0)

// Takes the group apart: devices 0..upto-1 (the whole group by default; a half-built one from smhip_set_devices' error
// paths).  abort: a collective was only partly issued -- the communicators are aborted (ncclCommAbort tears the queued
// work down instead of waiting for peers that will never arrive) rather than destroyed, and the streams are not waited for
// first (they may be stuck behind that collective).  Caller holds g_mutex.
int dissolve_group(int upto = -1, bool abort = false) {
    if (upto < 0) upto = g_ndev;
    for (int g = 0; g < upto; ++g) {
        ThreadDeviceScope scope(g);
        hipStream_t s;
        if (!abort && acquire(&s) == SMHIP_OK) (void)hipStreamSynchronize(s);
        if (g_comms[g]) {
            if (abort && g_rccl.CommAbort) (void)g_rccl.CommAbort(g_comms[g]);
            else (void)g_rccl.CommDestroy(g_comms[g]);
        }
        g_comms[g] = nullptr;
    }
    for (int g = 0; g < upto; ++g) {
        ThreadDeviceScope scope(g);
        hipStream_t s;
        if (abort && acquire(&s) == SMHIP_OK) (void)hipStreamSynchronize(s);  // after the abort the streams drain
        if (g_slot[g]) smhip_free(g_slot[g]);
        g_slot[g] = nullptr;
    }
    g_ndev = 0;
    return SMHIP_OK;
}

int group_size(const char *who, int *n) {
    std::lock_guard<std::mutex> lock(g_mutex);
    if (g_ndev <= 0) return fail(SMHIP_ERR_INVALID, "%s: no device group (call smhip_set_devices(n) first)", who);
    *n = g_ndev;
    return SMHIP_OK;
}

ncclDataType_t wire_type(int dtype) {  // integers travel unsigned: sums wrap modulo 2^32 / 2^64, like the reference's lanes
    switch (dtype) {
        case SMHIP_F32: return ncclFloat32;
        case SMHIP_F64: return ncclFloat64;
        case SMHIP_I32: return ncclUint32;
        default: return ncclUint64;
    }
}

// One all-reduce of `count` values per device of the group, in place in slot[g], each on its device's stream.
int group_allreduce(int n, ncclDataType_t type, size_t count) {
    hipStream_t streams[kMaxGroup];
    for (int g = 0; g < n; ++g) {
        ThreadDeviceScope scope(g);
        if (int rc = acquire(&streams[g])) return rc;
    }
    std::lock_guard<std::mutex> lock(g_mutex);
    if (g_ndev != n) return fail(SMHIP_ERR_INVALID, "the device group changed while a sharded reduction was being issued");
    SMHIP_NCCL(g_rccl.GroupStart());
    for (int g = 0; g < n; ++g) {
        const ncclResult_t r = test_hook_fail_allreduce(g) ? ncclInternalError
                                                           : g_rccl.AllReduce(g_slot[g], g_slot[g], count, type, ncclSum, g_comms[g], streams[g]);
        if (r != ncclSuccess) {
            // devices 0..g-1 have their half of the collective queued and would wait for the others for ever: close the
            // group, abort the communicators, leave no group behind (the caller gets the error and may form a new one)
            (void)g_rccl.GroupEnd();
            char why[256];
            snprintf(why, sizeof why, "%s", g_rccl.GetErrorString(r));
            dissolve_group(-1, true);
            return fail(SMHIP_ERR_HIP, "ncclAllReduce on device %d: %s; the device group was dissolved (call smhip_set_devices again)", g, why);
        }
    }
    const ncclResult_t end = g_rccl.GroupEnd();
    if (end != ncclSuccess) {
        char why[256];
        snprintf(why, sizeof why, "%s", g_rccl.GetErrorString(end));
        dissolve_group(-1, true);
        return fail(SMHIP_ERR_HIP, "ncclGroupEnd: %s; the device group was dissolved (call smhip_set_devices again)", why);
    }
    return SMHIP_OK;
}

// Device 0's slot to host memory (waits for device 0's stream only).
int read_slot0(void *dst_host, size_t bytes) {
    ThreadDeviceScope scope(0);
    return smhip_download(dst_host, g_slot[0], bytes);
}

bool tables_ok(int n, const void *const *a, const void *const *b, const void *const *out, const size_t *count) {
    if (!count) return false;
    for (int g = 0; g < n; ++g)
        if (count[g] && ((a && !a[g]) || (b && !b[g]) || (out && !out[g]))) return false;
    return true;
}

}  // namespace
}  // namespace smhip

using namespace smhip;

extern "C" {

int smhip_split_range(int64_t n, int world, int rank, int64_t *start, int64_t *count) {
    if (n < 0 || world < 1 || rank < 0 || rank >= world || !start || !count) return fail(SMHIP_ERR_INVALID, "split_range: bad arguments");
    const int64_t base = n / world, extra = n % world;
    *start = rank * base + (rank < extra ? rank : extra);
    *count = base + (rank < extra ? 1 : 0);
    return SMHIP_OK;
}

int smhip_shard_outer(const int64_t *shape, const int64_t *stride_a, const int64_t *stride_b, int ndim, int world, int rank,
                      int64_t *shard_shape, int64_t *offset_a, int64_t *offset_b, int64_t *offset_out, int *replicated_mask) {
    if (!shape || !stride_a || !stride_b || ndim < 1 || ndim > SMHIP_MAX_NDIM) return fail(SMHIP_ERR_INVALID, "shard_outer: bad arguments");
    int64_t start, count;
    if (int rc = smhip_split_range(shape[0], world, rank, &start, &count)) return rc;
    int64_t inner = 1;
    for (int i = 1; i < ndim; ++i) inner *= shape[i];
    if (shard_shape) {
        shard_shape[0] = count;
        for (int i = 1; i < ndim; ++i) shard_shape[i] = shape[i];
    }
    if (offset_a) *offset_a = start * stride_a[0];
    if (offset_b) *offset_b = start * stride_b[0];
    if (offset_out) *offset_out = start * inner;
    if (replicated_mask) *replicated_mask = ((stride_a[0] == 0 && shape[0] > 1) ? 1 : 0) | ((stride_b[0] == 0 && shape[0] > 1) ? 2 : 0);
    return SMHIP_OK;
}

int smhip_set_devices(int n) {
    if (n < 0 || n > kMaxGroup) return fail(SMHIP_ERR_INVALID, "set_devices: %d", n);
    std::lock_guard<std::mutex> lock(g_mutex);
    if (g_ndev > 0) {
        const int had = g_ndev;
        dissolve_group();
        if (n == 0)  // no group any more: the devices' second library queues may be used again (runtime.hip: two queues per device)
            for (int g = 0; g < had; ++g) dispatch_single_queue(g, false);
    }
    if (n == 0) return SMHIP_OK;
    int have = 0;
    smhip_device_count(&have);
    if (have < 1) return fail(SMHIP_ERR_NO_DEVICE, "set_devices: no HIP device available; libsmhip has no CPU fallback");
    if (n > have) return fail(SMHIP_ERR_INVALID, "set_devices: %d devices asked for, %d present", n, have);
    if (int rc = load_rccl()) return rc;
    int devlist[kMaxGroup];
    for (int g = 0; g < n; ++g) {
        devlist[g] = g;
        ThreadDeviceScope scope(g);
        hipStream_t s;
        int rc = acquire(&s);  // checks the architecture and creates the device's stream
        if (!rc) rc = smhip_alloc(&g_slot[g], 16);
        if (rc) {
            dissolve_group(g, false);  // the slots of devices 0..g-1 go back to the pool; no group is left behind
            return rc;
        }
    }
    ncclResult_t r;
    {
        // ncclCommInitAll leaves the last device current; put this thread back afterwards
        ThreadDeviceScope scope(current_device());
        for (int g = 0; g < n; ++g) g_comms[g] = nullptr;
        r = test_hook_fail_comm_init() ? ncclInternalError : g_rccl.CommInitAll(g_comms, n, devlist);
    }
    if (r != ncclSuccess) {
        char why[256];
        snprintf(why, sizeof why, "%s", g_rccl.GetErrorString(r));
        dissolve_group(n, false);
        return fail(SMHIP_ERR_HIP, "ncclCommInitAll(%d devices): %s", n, why);
    }
    // what RCCL itself says it built: every communicator must count n ranks and sit on its device
    for (int g = 0; g < n; ++g) {
        int count = -1, dev = -1;
        if (g_rccl.CommCount(g_comms[g], &count) != ncclSuccess || g_rccl.CommCuDevice(g_comms[g], &dev) != ncclSuccess || count != n || dev != g) {
            dissolve_group(n, false);
            return fail(SMHIP_ERR_HIP, "ncclCommInitAll: communicator %d reports %d ranks on device %d (expected %d ranks on device %d)", g, count, dev, n, g);
        }
    }
    g_ndev = n;
    return SMHIP_OK;
}

int smhip_get_devices(int *n) {
    if (!n) return fail(SMHIP_ERR_INVALID, "get_devices: null");
    std::lock_guard<std::mutex> lock(g_mutex);
    *n = g_ndev;
    return SMHIP_OK;
}

int smhip_sharded_synchronize(void) {
    int n;
    if (int rc = group_size("sharded_synchronize", &n)) return rc;
    for (int g = 0; g < n; ++g) {
        ThreadDeviceScope scope(g);
        hipStream_t s;
        if (int rc = acquire(&s)) return rc;
        SMHIP_TRY(hipStreamSynchronize(s));
    }
    return SMHIP_OK;
}

int smhip_sharded_contiguous(int op, int dtype, const void *const *a, const void *const *b, void *const *out, const size_t *n) {
    int nd;
    if (int rc = group_size("sharded_contiguous", &nd)) return rc;
    if (!a || !b || !out || !tables_ok(nd, a, b, out, n)) return fail(SMHIP_ERR_INVALID, "sharded_contiguous: null table or entry");
    for (int g = 0; g < nd; ++g) {
        ThreadDeviceScope scope(g);
        if (int rc = smhip_contiguous(op, dtype, a[g], b[g], out[g], n[g])) return rc;
    }
    return SMHIP_OK;
}

int smhip_sharded_array_scalar(int op, int dtype, const void *const *a, const void *value_host, const size_t *n, void *const *out) {
    int nd;
    if (int rc = group_size("sharded_array_scalar", &nd)) return rc;
    if (!a || !out || !value_host || !tables_ok(nd, a, nullptr, out, n)) return fail(SMHIP_ERR_INVALID, "sharded_array_scalar: null table or entry");
    for (int g = 0; g < nd; ++g) {
        ThreadDeviceScope scope(g);
        if (int rc = smhip_array_scalar(op, dtype, a[g], value_host, n[g], out[g])) return rc;
    }
    return SMHIP_OK;
}

int smhip_sharded_elementwise(int op, int dtype, const void *const *a, const int64_t *stride_a, const void *const *b,
                              const int64_t *stride_b, const int64_t *shape, int ndim, void *const *out) {
    int nd;
    if (int rc = group_size("sharded_elementwise", &nd)) return rc;
    if (!a || !b || !out || !stride_a || !stride_b || !shape) return fail(SMHIP_ERR_INVALID, "sharded_elementwise: null table");
    if (ndim < 1 || ndim > SMHIP_MAX_NDIM) return fail(SMHIP_ERR_INVALID, "sharded_elementwise: ndim %d outside 1..%d", ndim, SMHIP_MAX_NDIM);
    for (int g = 0; g < nd; ++g) {
        int64_t local[SMHIP_MAX_NDIM];
        if (int rc = smhip_shard_outer(shape, stride_a, stride_b, ndim, nd, g, local, nullptr, nullptr, nullptr, nullptr)) return rc;
        if (local[0] == 0) continue;  // more devices than rows
        ThreadDeviceScope scope(g);
        if (int rc = smhip_elementwise(op, dtype, a[g], stride_a, b[g], stride_b, local, ndim, out[g])) return rc;
    }
    return SMHIP_OK;
}

int smhip_sharded_contiguous_sum(int op, int dtype, const void *const *a, const void *const *b, void *const *out, const size_t *n,
                                 double *sum_host) {
    int nd;
    if (int rc = group_size("sharded_contiguous_sum", &nd)) return rc;
    if (!a || !b || !out || !sum_host || !tables_ok(nd, a, b, out, n)) return fail(SMHIP_ERR_INVALID, "sharded_contiguous_sum: null table or entry");
    for (int g = 0; g < nd; ++g) {
        ThreadDeviceScope scope(g);
        if (int rc = smhip_contiguous_sum_async(op, dtype, a[g], b[g], out[g], n[g], static_cast<double *>(g_slot[g]))) return rc;
    }
    if (int rc = group_allreduce(nd, ncclFloat64, 1)) return rc;
    return read_slot0(sum_host, sizeof(double));
}

int smhip_sharded_sum(int dtype, const void *const *a, const size_t *n, double *sum_host) {
    int nd;
    if (int rc = group_size("sharded_sum", &nd)) return rc;
    if (!a || !sum_host || !tables_ok(nd, a, nullptr, nullptr, n)) return fail(SMHIP_ERR_INVALID, "sharded_sum: null table or entry");
    for (int g = 0; g < nd; ++g) {
        ThreadDeviceScope scope(g);
        if (int rc = smhip_sum_async(dtype, a[g], n[g], static_cast<double *>(g_slot[g]))) return rc;
    }
    if (int rc = group_allreduce(nd, ncclFloat64, 1)) return rc;
    return read_slot0(sum_host, sizeof(double));
}

int smhip_sharded_dot(int dtype, const void *const *a, const void *const *b, const size_t *n, void *out_host) {
    int nd;
    if (int rc = group_size("sharded_dot", &nd)) return rc;
    if (!valid_dtype(dtype)) return fail(SMHIP_ERR_INVALID, "sharded_dot: bad dtype %d", dtype);
    if (!a || !b || !out_host || !tables_ok(nd, a, b, nullptr, n)) return fail(SMHIP_ERR_INVALID, "sharded_dot: null table or entry");
    for (int g = 0; g < nd; ++g) {
        ThreadDeviceScope scope(g);
        // 8 bytes per device: the fp64 partial (floats) or the wrapped partial sign-extended to int64 (integers)
        if (int rc = smhip_dot_async(dtype, a[g], b[g], n[g], static_cast<double *>(g_slot[g]))) return rc;
    }
    const bool integer = dtype == SMHIP_I32 || dtype == SMHIP_I64;
    if (int rc = group_allreduce(nd, integer ? ncclUint64 : ncclFloat64, 1)) return rc;
    unsigned char raw[8];
    if (int rc = read_slot0(raw, 8)) return rc;
    switch (dtype) {
        case SMHIP_F32: { double d; memcpy(&d, raw, 8); const float f = (float)d; memcpy(out_host, &f, 4); break; }
        case SMHIP_F64: memcpy(out_host, raw, 8); break;
        case SMHIP_I32: memcpy(out_host, raw, 4); break;  // low 32 bits (little endian): the sum modulo 2^32
        default: memcpy(out_host, raw, 8); break;
    }
    return SMHIP_OK;
}

/* ------------------------------------------------------ device to device */

namespace {
bool g_peer_enabled[kMaxGroup][kMaxGroup] = {};  // [accessing device][peer]; guarded by g_mutex

// Lets `device` map `peer`'s memory (so copies between them go over xGMI directly).  A pair that cannot is left alone:
// hipMemcpyPeerAsync then stages through the host by itself.
void enable_peer(int device, int peer) {
    if (device == peer || device >= kMaxGroup || peer >= kMaxGroup) return;
    {
        std::lock_guard<std::mutex> lock(g_mutex);
        if (g_peer_enabled[device][peer]) return;
        g_peer_enabled[device][peer] = true;
    }
    int can = 0;
    if (hipDeviceCanAccessPeer(&can, device, peer) != hipSuccess || !can) { (void)hipGetLastError(); return; }
    ThreadDeviceScope scope(device);
    hipStream_t s;
    if (acquire(&s) != SMHIP_OK) return;
    const hipError_t e = hipDeviceEnablePeerAccess(peer, 0);
    if (e != hipSuccess) (void)hipGetLastError();  // hipErrorPeerAccessAlreadyEnabled included
}
}  // namespace

int smhip_copy_peer(void *dst, int dst_device, const void *src, int src_device, size_t bytes) {
    if (bytes == 0) return SMHIP_OK;
    if (!dst || !src) return fail(SMHIP_ERR_INVALID, "copy_peer: null");
    int have = 0;
    smhip_device_count(&have);
    if (have < 1) return fail(SMHIP_ERR_NO_DEVICE, "copy_peer: no HIP device available; libsmhip has no CPU fallback");
    if (dst_device < 0 || src_device < 0 || dst_device >= have || src_device >= have)
        return fail(SMHIP_ERR_INVALID, "copy_peer: device %d -> %d, %d present", src_device, dst_device, have);
    if (src_device == dst_device) {
        ThreadDeviceScope scope(dst_device);
        return smhip_copy(dst, src, bytes);
    }
    enable_peer(dst_device, src_device);
    enable_peer(src_device, dst_device);
    hipStream_t s_src, s_dst;
    hipEvent_t ready = nullptr, done = nullptr;
    {
        ThreadDeviceScope scope(src_device);
        if (int rc = acquire(&s_src)) return rc;
        ready = pool_event_take(src_device);
        if (!ready) return fail(SMHIP_ERR_HIP, "copy_peer: no event for device %d", src_device);
        const hipError_t e = hipEventRecord(ready, s_src);  // everything queued on the source's stream so far
        if (e != hipSuccess) { pool_event_give(src_device, ready); return fail(SMHIP_ERR_HIP, "copy_peer: hipEventRecord: %s", hipGetErrorString(e)); }
    }
    int rc = SMHIP_OK;
    {
        ThreadDeviceScope scope(dst_device);
        rc = acquire(&s_dst);
        hipError_t e = hipSuccess;
        if (!rc) e = hipStreamWaitEvent(s_dst, ready, 0);
        if (!rc && e == hipSuccess) e = hipMemcpyPeerAsync(dst, dst_device, src, src_device, bytes, s_dst);
        residency_forget(dst_device, dst, bytes);  // what arrives over xGMI is in no cache of this GPU
        if (!rc && e == hipSuccess && !(done = pool_event_take(dst_device))) e = hipErrorOutOfMemory;
        if (!rc && e == hipSuccess) e = hipEventRecord(done, s_dst);
        if (!rc && e != hipSuccess) rc = fail(SMHIP_ERR_HIP, "copy_peer %d -> %d: %s", src_device, dst_device, hipGetErrorString(e));
    }
    if (!rc) {
        // the source's stream continues only after the copy has read it (its buffer may be freed or overwritten next)
        ThreadDeviceScope scope(src_device);
        const hipError_t e = hipStreamWaitEvent(s_src, done, 0);
        if (e != hipSuccess) rc = fail(SMHIP_ERR_HIP, "copy_peer: hipStreamWaitEvent: %s", hipGetErrorString(e));
    }
    pool_event_give(src_device, ready);  // recycled events: a later record simply supersedes this one (the waits are already queued)
    if (done) pool_event_give(dst_device, done);
    return rc;
}

/* -------------------------------------------------- one process per GPU */

int smhip_comm_unique_id(void *id128) {
    if (!id128) return fail(SMHIP_ERR_INVALID, "comm_unique_id: null");
    std::lock_guard<std::mutex> lock(g_mutex);
    if (int rc = load_rccl()) return rc;
    static_assert(sizeof(ncclUniqueId) == 128, "smhip.h promises 128 bytes");
    ncclUniqueId id;
    SMHIP_NCCL(g_rccl.GetUniqueId(&id));
    memcpy(id128, &id, sizeof id);
    return SMHIP_OK;
}

int smhip_comm_init_rank(int nranks, int rank, const void *id128) {
    if (nranks < 1 || rank < 0 || rank >= nranks || !id128) return fail(SMHIP_ERR_INVALID, "comm_init_rank: bad arguments");
    hipStream_t s;
    if (int rc = acquire(&s)) return rc;  // selects this thread's device (hipSetDevice) for the communicator
    std::lock_guard<std::mutex> lock(g_mutex);
    if (int rc = load_rccl()) return rc;
    if (g_rank_comm) {
        (void)g_rccl.CommDestroy(g_rank_comm);
        g_rank_comm = nullptr;
    }
    ncclUniqueId id;
    memcpy(&id, id128, sizeof id);
    SMHIP_NCCL(g_rccl.CommInitRank(&g_rank_comm, nranks, id, rank));
    g_nranks = nranks;
    g_rank = rank;
    g_rank_device = current_device();
    return SMHIP_OK;
}

int smhip_comm_info(int *nranks, int *rank) {
    std::lock_guard<std::mutex> lock(g_mutex);
    int n = 0, r = -1;
    if (g_rank_comm) {  // what RCCL says about the communicator, not what this library was told
        SMHIP_NCCL(g_rccl.CommCount(g_rank_comm, &n));
        SMHIP_NCCL(g_rccl.CommUserRank(g_rank_comm, &r));
    }
    if (nranks) *nranks = n;
    if (rank) *rank = r;
    return SMHIP_OK;
}

int smhip_group_info(int index, int *nranks, int *rank, int *device) {
    std::lock_guard<std::mutex> lock(g_mutex);
    if (g_ndev <= 0) return fail(SMHIP_ERR_INVALID, "group_info: no device group (call smhip_set_devices(n) first)");
    if (index < 0 || index >= g_ndev) return fail(SMHIP_ERR_INVALID, "group_info: index %d outside the group of %d", index, g_ndev);
    int n = 0, r = -1, d = -1;
    SMHIP_NCCL(g_rccl.CommCount(g_comms[index], &n));
    SMHIP_NCCL(g_rccl.CommUserRank(g_comms[index], &r));
    SMHIP_NCCL(g_rccl.CommCuDevice(g_comms[index], &d));
    if (nranks) *nranks = n;
    if (rank) *rank = r;
    if (device) *device = d;
    return SMHIP_OK;
}

int smhip_enable_test_hooks(int on) {
    g_test_hooks.store(on ? 1 : 0, std::memory_order_relaxed);
    return SMHIP_OK;
}

int smhip_rccl_version(int *version) {
    if (!version) return fail(SMHIP_ERR_INVALID, "rccl_version: null");
    std::lock_guard<std::mutex> lock(g_mutex);
    if (int rc = load_rccl()) return rc;
    SMHIP_NCCL(g_rccl.GetVersion(version));
    return SMHIP_OK;
}

int smhip_comm_destroy(void) {
    std::lock_guard<std::mutex> lock(g_mutex);
    if (g_rank_comm) {
        (void)g_rccl.CommDestroy(g_rank_comm);
        g_rank_comm = nullptr;
        g_nranks = 0;
        g_rank = -1;
    }
    return SMHIP_OK;
}

int smhip_allreduce_sum_async(int dtype, void *inout_dev, size_t count) {
    if (!valid_dtype(dtype)) return fail(SMHIP_ERR_INVALID, "allreduce: bad dtype %d", dtype);
    if (count == 0) return SMHIP_OK;
    if (!inout_dev) return fail(SMHIP_ERR_INVALID, "allreduce: null buffer");
    {
        std::lock_guard<std::mutex> lock(g_mutex);
        if (!g_rank_comm) return fail(SMHIP_ERR_INVALID, "allreduce: no communicator (call smhip_comm_init_rank first)");
    }
    hipStream_t s;
    if (int rc = acquire(&s)) return rc;
    std::lock_guard<std::mutex> lock(g_mutex);
    if (!g_rank_comm) return fail(SMHIP_ERR_INVALID, "allreduce: no communicator (call smhip_comm_init_rank first)");
    if (current_device() != g_rank_device)
        return fail(SMHIP_ERR_INVALID, "allreduce: the communicator lives on device %d, this thread is on device %d", g_rank_device, current_device());
    SMHIP_NCCL(g_rccl.AllReduce(inout_dev, inout_dev, count, wire_type(dtype), ncclSum, g_rank_comm, s));
    return SMHIP_OK;
}

}  // extern "C"
