// runtime.hip -- libsmhip.so's C ABI (include/smhip.h): per-thread device and
// stream selection, the pooled device allocator, the host-side shape layer and
// the entry points that validate arguments and hand off to the kernel launchers.
//
// The reference has no runtime at all (header-only loops, `new T[n]` per
// operator, SMArray.h:219); this file is what the process/device boundary costs.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "internal.h"

namespace smhip {
namespace {

constexpr int kMaxDevices = 64;

struct ThreadState {
    int device = -1;              // -1: not chosen yet (device 0 on first use)
    bool checked[kMaxDevices] = {};
    hipStream_t user_stream = nullptr;
    bool use_user_stream = false;
    void *result_host[kMaxDevices] = {};  // a pinned, device-mapped 64-byte slot per device: where the synchronous reductions'
    void *result_dev[kMaxDevices] = {};   // kernels leave their scalar for the host (no device buffer, no copy packet)
    int op_depth = 0;   // > 0: inside an operator's scope (OpScope); allocations made there are that operator's scratch
    int op_queue = 0;   // the library queue the operator in progress runs on
    std::string error;
    // pinned staging ring for small uploads (per thread and device): copy in, enqueue, return
    unsigned char *ring[kMaxDevices] = {};
    size_t ring_head[kMaxDevices] = {};
    hipStream_t ring_stream[kMaxDevices] = {};
    ~ThreadState();  // a thread that ends hands its pinned ring to the next thread
};
thread_local ThreadState tls;

// Pinned rings of threads that have ended, per device, with the stream their last copies were queued on
// (the adopting thread drains it before reusing the memory).
struct SpareRing { unsigned char *ptr; hipStream_t stream; };
std::mutex g_ring_mutex;
std::vector<SpareRing> g_spare_rings[kMaxDevices];

// One library-owned stream per device, shared by all threads that did not
// bring their own.
std::mutex g_mutex;
hipStream_t g_streams[kMaxDevices] = {};
int g_cus[kMaxDevices] = {};
int g_device_count = -1;

// ---------------------------------------------------------------- the pool
// Two tiers, both per device:
//   small (< 1 MiB)  size-class free lists (powers of two from 256 B); each block is its own hipMalloc.
//   large (>= 1 MiB) carved, 2 MiB-aligned and first-fit, out of ARENAS -- single hipMalloc'd slabs of
//                    max(4 x request, 256 MiB) (a request above 4 GiB gets a slab of its own size).
// Why arenas: (1) a freed 1 GiB operand buffer is reused as-is by the next same-sized result, which turns
// the reference's per-operator `new T[n]` + first-touch page faults (SMArray.h:219) into a pointer bump;
// (2) placement.  The 2R+1W stream's rate follows the RELATIVE physical placement of its three buffers:
// three separate hipMallocs land anywhere between 6.27 and 6.66 TB/s from one process to the next, three
// carvings of one slab are at 6.49-6.51 TB/s every time (profiles/r01_placement_notes.txt).
// Stream-ordered reuse.  A block remembers the stream it was handed to; when it is freed, the streams that may still
// be working on its bytes are that one and the freeing thread's current one (smhip_set_stream orders a thread's new
// stream after its old one, so a switch between use and free is covered as well).  The device's library stream is
// remembered as a flag and ordered lazily -- an event recorded on it when the bytes are reused covers everything queued
// before the free; a caller-owned stream may be destroyed by then, so an event is recorded on it at free time.  The
// next owner's stream WAITS on those events (hipStreamWaitEvent: nothing blocks on the host); bytes that come back on the
// stream that freed them need no wait at all.  A stream that cannot be recorded on makes the reuse synchronise the device.
constexpr size_t kLargeMin = (size_t)1 << 20, kLargeAlign = (size_t)2 << 20;
constexpr size_t kArenaFloor = (size_t)256 << 20, kArenaSolo = (size_t)4 << 30;

// An event recorded on a caller-owned stream when bytes it may be using were freed.  Shared by the extents the bytes end
// up in (splits copy it, merges keep the later one per stream); it returns to the event pool with its last holder.
struct Recorded {
    int device;
    hipEvent_t ev;
    ~Recorded();
};
struct Pending { hipStream_t stream; uint64_t seq; std::shared_ptr<Recorded> rec; };
struct Tag {
    bool lib = false;      // the library stream of the device may still be using the bytes
    bool unknown = false;  // so may a stream no event could be recorded on
    std::vector<Pending> pending;  // at most one per caller-owned stream: a later record on a stream covers the earlier ones
    void add(const Pending &p) {
        for (Pending &q : pending)
            if (q.stream == p.stream) {
                if (q.seq < p.seq) q = p;
                return;
            }
        pending.push_back(p);
    }
    void merge(const Tag &o) {
        lib |= o.lib;
        unknown |= o.unknown;
        for (const Pending &p : o.pending) add(p);
    }
};
struct Extent { size_t off, size; Tag tag; };
struct Arena {
    char *base = nullptr;
    size_t size = 0, used = 0;
    int device = 0;
    std::vector<Extent> free;  // sorted by offset, coalesced
};
struct Block { size_t cls; int device; Arena *arena; size_t off; hipStream_t stream; std::thread::id owner; };
std::unordered_map<void *, Block> g_live;                                      // handed out
std::map<std::pair<int, size_t>, std::vector<std::pair<void *, Tag>>> g_free;  // small blocks, cached
std::vector<Arena *> g_arenas;
std::mutex g_event_mutex;
std::vector<hipEvent_t> g_event_pool[kMaxDevices];  // timing-disabled events, recycled
uint64_t g_record_seq = 0;
size_t g_bytes_live = 0, g_bytes_cached = 0;

size_t size_class(size_t bytes) {
    if (bytes <= 256) return 256;
    if (bytes >= kLargeMin) return (bytes + kLargeAlign - 1) / kLargeAlign * kLargeAlign;
    size_t c = 256;
    while (c < bytes) c <<= 1;
    return c;
}

// First fit in this device's arenas.  Caller holds g_mutex.
bool carve(int dev, size_t need, Arena **arena, size_t *off, Tag *tag) {
    for (Arena *a : g_arenas) {
        if (a->device != dev) continue;
        for (size_t i = 0; i < a->free.size(); ++i) {
            Extent &e = a->free[i];
            if (e.size < need) continue;
            *arena = a;
            *off = e.off;
            *tag = e.tag;  // a split leaves both parts with the same obligations
            if (e.size == need) a->free.erase(a->free.begin() + i);
            else { e.off += need; e.size -= need; }
            a->used += need;
            return true;
        }
    }
    return false;
}

// Return [off, off + size) to its arena, merging with neighbours.  Caller holds g_mutex.
void release(Arena *a, size_t off, size_t size, Tag &&tag) {
    size_t i = 0;
    while (i < a->free.size() && a->free[i].off < off) ++i;
    a->free.insert(a->free.begin() + i, Extent{off, size, std::move(tag)});
    auto merge = [&](size_t lo) {  // merge free[lo] and free[lo + 1] if adjacent
        if (lo + 1 >= a->free.size()) return false;
        Extent &x = a->free[lo], &y = a->free[lo + 1];
        if (x.off + x.size != y.off) return false;
        x.size += y.size;
        x.tag.merge(y.tag);
        a->free.erase(a->free.begin() + lo + 1);
        return true;
    };
    merge(i);
    if (i > 0) merge(i - 1);
    a->used -= size;
}

// A recycled (or new) timing-disabled event of device `dev`; nullptr if none can be made.
hipEvent_t take_event(int dev) {
    {
        std::lock_guard<std::mutex> lock(g_event_mutex);
        if (!g_event_pool[dev].empty()) {
            hipEvent_t e = g_event_pool[dev].back();
            g_event_pool[dev].pop_back();
            return e;
        }
    }
    int cur = -1;
    (void)hipGetDevice(&cur);
    if (cur != dev && hipSetDevice(dev) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    hipEvent_t e = nullptr;
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); e = nullptr; }
    if (cur != dev && cur >= 0) (void)hipSetDevice(cur);
    return e;
}
void give_event(int dev, hipEvent_t e) {
    std::lock_guard<std::mutex> lock(g_event_mutex);
    g_event_pool[dev].push_back(e);
}
Recorded::~Recorded() { give_event(device, ev); }

// Marks `tag` with "stream `t` of device `dev` may still be working on these bytes".  `t` must be alive: the library's
// stream, or the calling thread's CURRENT stream (smhip_set_stream's contract) -- a stream handle that may have been
// destroyed is never passed to HIP.
void tag_stream(Tag &tag, int dev, hipStream_t t) {
    if (t == g_streams[dev]) { tag.lib = true; return; }
    hipEvent_t e = take_event(dev);
    if (e && hipEventRecord(e, t) == hipSuccess) {
        uint64_t seq;
        {
            std::lock_guard<std::mutex> lock(g_event_mutex);
            seq = ++g_record_seq;
        }
        tag.add(Pending{t, seq, std::shared_ptr<Recorded>(new Recorded{dev, e})});
        return;
    }
    (void)hipGetLastError();
    if (e) give_event(dev, e);
    tag.unknown = true;
}

// Orders stream `s` (device `dev`) after everything `tag` names.
void queues_join(int dev);
void queues_after_external_wait(int dev);
int order_after(Tag &tag, int dev, hipStream_t s) {
    hipError_t err = hipSuccess;
    if (tag.unknown) {
        err = hipDeviceSynchronize();
    } else {
        bool waited = false;
        for (const Pending &p : tag.pending)
            if (err == hipSuccess && p.stream != s) { err = hipStreamWaitEvent(s, p.rec->ev, 0); waited = true; }
        if (waited && s == g_streams[dev]) queues_after_external_wait(dev);  // the library's second queue orders itself behind this as well
        if (err == hipSuccess && tag.lib && s != g_streams[dev] && g_streams[dev]) {
            queues_join(dev);  // "the library stream" = both of its queues
            hipEvent_t e = take_event(dev);
            if (!e) {
                err = hipStreamSynchronize(g_streams[dev]);
            } else {
                err = hipEventRecord(e, g_streams[dev]);
                if (err == hipSuccess) err = hipStreamWaitEvent(s, e, 0);
                give_event(dev, e);
            }
        }
    }
    tag.pending.clear();
    if (err != hipSuccess) return fail(SMHIP_ERR_HIP, "pool: ordering a reused block: %s", hipGetErrorString(err));
    return SMHIP_OK;
}


// ------------------------------------------------------------------------------------------------ two queues per device
// The reference's threads run whatever chunk is ready (calculate.h:47); a GPU stream runs its kernels strictly one after
// the other, and between two kernels sits a fixed ~2.2 us -- dispatch, the cache write-back / invalidate at the kernel
// boundary, the completion signal (cold operands).  A 500 us kernel hides that, a 20 us kernel shows it as ten
// points.  Independent operators therefore go out on TWO hardware queues per device, so that one's tail overlaps the
// next one's head: the library knows every operand span of an operator, keeps (per device) who last wrote and who last read
// which span on which queue, puts an operator on the queue its dependencies already sit on, and on the other queue -- the
// one the previous independent operator did NOT take -- when it has none.  A dependency that crosses queues becomes an
// event edge (record on the producer's queue, wait on the consumer's): RAW, WAR and WAW alike.  What the tracker cannot see
// falls back to order: operators whose spans are not declared run as BARRIER operators on queue 0 (they wait for everything
// and everything later waits for them); a span that drops out of the tracker's ring raises a floor every later operator
// orders itself behind; the caller's own stream (smhip_set_stream) is outside all this -- operators run there in the
// order they were called, as before -- and a caller that takes the library's stream handle (smhip_get_stream) or drives
// the device group (sharded entry points) turns the second queue off for that device.  SMHIP_QUEUES=1 turns it off everywhere.
constexpr int kUses = 256;
constexpr size_t kOverlapMinBytes = (size_t)1 << 20;   // smaller operators are barrier operators: nothing to overlap, nothing to track
// Larger operators stay on queue 0 (tracked like the others).  Two queues do not just overlap a kernel's tail with the next
// one's head: the hardware runs the two kernels side by side, which halves the fixed cost per launch and doubles the streams
// the memory side sees at once.  Cold operands, % of peak with one / two queues (tools/cold_rates.py, profiles/r04_cold_rates_queues.txt):
// a + b at 16 / 32 / 64 MiB per array 63 / 71 / 78 -> 77 / 79 / 80, at 128 / 256 MiB 80.9 / 82.5 -> 79.7 / 78.8;
// (R, 4096) * (1, 4096) at 16 / 32 / 64 MiB 56 / 68 / 75.2 -> 65 / 74 / 75.8, at 128 MiB 79.2 -> 76.8; a * s gains up to 128 MiB.
// (A second queue of another PRIORITY, high or low, ran like one queue: no overlap at all.)
constexpr size_t kOverlapMaxBytes = (size_t)224 << 20;
struct Dispatch {
    std::recursive_mutex m;
    hipStream_t q1 = nullptr;
    uint64_t seq[2] = {0, 0};      // operators issued per queue
    uint64_t synced[2][2] = {};    // synced[q][p]: queue q is ordered behind queue p's operators up to this one
    uint64_t floor[2] = {0, 0};    // every later operator is ordered behind these (evicted spans, barrier operators)
    // who last wrote / read which span: structure of arrays, so that the overlap scan of an operator (its <= 17 spans against
    // every span on record) is a handful of vector compares
    uintptr_t lo[kUses] = {}, hi[kUses] = {};
    uint64_t wseq[kUses] = {}, rseq[kUses][2] = {};
    unsigned char wq[kUses] = {};
    int used = 0, next = 0;
    int last = 1;                  // the queue the last independent operator took
    bool single = false;           // second queue off (the stream handle was handed out, sharded use, SMHIP_QUEUES=1)
    unsigned long long edges = 0, alternations = 0;  // diagnostics
};
Dispatch g_dispatch[kMaxDevices];
bool queues_enabled() {
    static const bool on = [] { const char *e = getenv("SMHIP_QUEUES"); return !(e && atoi(e) == 1); }();
    return on;
}
hipStream_t queue_stream(int dev, int q) { return q == 0 ? g_streams[dev] : g_dispatch[dev].q1; }

// Orders queue q behind queue p's operators up to `need` (no-op when it already is).  Caller holds d.m.
int queue_wait(Dispatch &d, int dev, int q, int p, uint64_t need) {
    if (q == p || need <= d.synced[q][p] || !queue_stream(dev, p) || !queue_stream(dev, q)) return SMHIP_OK;
    hipEvent_t e = take_event(dev);
    if (!e) {
        SMHIP_TRY(hipStreamSynchronize(queue_stream(dev, p)));
    } else {
        hipError_t err = hipEventRecord(e, queue_stream(dev, p));
        if (err == hipSuccess) err = hipStreamWaitEvent(queue_stream(dev, q), e, 0);
        give_event(dev, e);
        if (err != hipSuccess) return fail(SMHIP_ERR_HIP, "queues: ordering queue %d behind queue %d: %s", q, p, hipGetErrorString(err));
    }
    d.synced[q][p] = d.seq[p];  // the event covers everything queued on p so far
    ++d.edges;
    return SMHIP_OK;
}
// Queue 0 waits for everything on queue 1 and becomes the point everything later is ordered behind: what an operator
// with undeclared spans, a read-back, a timing event or a hand-over to a caller's stream needs.  Caller holds d.m.
int queue_barrier(Dispatch &d, int dev) {
    if (d.q1) {
        if (int rc = queue_wait(d, dev, 0, 1, d.seq[1])) return rc;
    }
    d.floor[0] = ++d.seq[0];
    return SMHIP_OK;
}
// The record of span [lo, hi): the one that names exactly it, or a fresh one (what drops out of the table becomes a floor).
int use_slot(Dispatch &d, uintptr_t lo, uintptr_t hi) {
    for (int i = 0; i < d.used; ++i)
        if (d.lo[i] == lo && d.hi[i] == hi) return i;
    int i;
    if (d.used < kUses) {
        i = d.used++;
    } else {
        i = d.next;
        d.next = (d.next + 1) % kUses;
        if (d.wseq[i] > d.floor[d.wq[i]]) d.floor[d.wq[i]] = d.wseq[i];
        for (int k = 0; k < 2; ++k)
            if (d.rseq[i][k] > d.floor[k]) d.floor[k] = d.rseq[i][k];
    }
    d.lo[i] = lo;
    d.hi[i] = hi;
    d.wseq[i] = d.rseq[i][0] = d.rseq[i][1] = 0;
    d.wq[i] = 0;
    return i;
}
// What an operator that reads `reads` and writes `write` must be ordered behind, per queue -- on top of the floors, which
// every operator is behind and which therefore say nothing about the queue that suits it.
void dependencies(const Dispatch &d, const Span *reads, size_t n_reads, Span write, uint64_t (&need)[2]) {
    need[0] = need[1] = 0;
    const int n = d.used;
    if (write.p && write.bytes) {  // WAW and WAR
        const uintptr_t lo = reinterpret_cast<uintptr_t>(write.p), hi = lo + write.bytes;
        for (int i = 0; i < n; ++i) {
            if (d.lo[i] < hi && lo < d.hi[i]) {
                if (d.wseq[i] > need[d.wq[i]]) need[d.wq[i]] = d.wseq[i];
                if (d.rseq[i][0] > need[0]) need[0] = d.rseq[i][0];
                if (d.rseq[i][1] > need[1]) need[1] = d.rseq[i][1];
            }
        }
    }
    for (size_t r = 0; r < n_reads; ++r) {  // RAW
        if (!reads[r].p || !reads[r].bytes) continue;
        const uintptr_t lo = reinterpret_cast<uintptr_t>(reads[r].p), hi = lo + reads[r].bytes;
        for (int i = 0; i < n; ++i)
            if (d.lo[i] < hi && lo < d.hi[i] && d.wseq[i] > need[d.wq[i]]) need[d.wq[i]] = d.wseq[i];
    }
}
}  // namespace

// An operator's scope: picks its queue, orders it behind what it depends on, and keeps the device's dispatcher locked
// until the operator's launches are queued (another thread's operator cannot slip an event between the bookkeeping and
// the launch it describes).
OpScope::~OpScope() {
    if (locked_) {
        --tls.op_depth;
        static_cast<Dispatch *>(locked_)->m.unlock();
    }
}
int OpScope::begin_barrier(hipStream_t *s) { return begin(nullptr, 0, Span{nullptr, 0}, s, true); }
int OpScope::begin(const Span *reads, size_t n_reads, Span write, hipStream_t *s, bool barrier) {
    if (int rc = acquire_stream(s)) return rc;
    if (tls.use_user_stream) return SMHIP_OK;  // the caller's stream: operators run there in call order
    const int dev = tls.device;
    Dispatch &d = g_dispatch[dev];
    d.m.lock();
    locked_ = &d;
    if (tls.op_depth++ > 0) {  // an entry point called from inside another operator: part of that operator
        *s = queue_stream(dev, tls.op_queue);
        return SMHIP_OK;
    }
    tls.op_queue = 0;
    size_t bytes = write.bytes;
    for (size_t i = 0; i < n_reads; ++i) bytes += reads[i].bytes;
    if (barrier || d.single || !queues_enabled() || bytes < kOverlapMinBytes) return queue_barrier(d, dev);
    if (!d.q1) {
        hipStream_t q = nullptr;
        if (hipStreamCreateWithFlags(&q, hipStreamNonBlocking) != hipSuccess) {
            (void)hipGetLastError();
            d.single = true;
            return queue_barrier(d, dev);
        }
        d.q1 = q;
    }
    uint64_t need[2];
    dependencies(d, reads, n_reads, write, need);
    // the queue that costs no event edge; an operator that depends on nothing unfinished elsewhere takes the queue the
    // previous such operator did not
    const bool edge0 = need[1] > d.synced[0][1], edge1 = need[0] > d.synced[1][0];  // what running on queue 0 / 1 would have to wait for
    int q;
    static const size_t overlap_max = [] { const char *e = getenv("SMHIP_OVERLAP_MAX_MIB"); return e && *e ? (size_t)atol(e) << 20 : kOverlapMaxBytes; }();  // experiments
    if (bytes > overlap_max) q = 0;
    else if (edge0 != edge1) q = edge0 ? 1 : 0;
    else if (!edge0) { q = d.last ^ 1; d.last = q; ++d.alternations; }
    else q = 0;
    if (int rc = queue_wait(d, dev, q, q ^ 1, need[q ^ 1] > d.floor[q ^ 1] ? need[q ^ 1] : d.floor[q ^ 1])) return rc;
    const uint64_t my = ++d.seq[q];
    if (write.p && write.bytes) {
        const int w = use_slot(d, reinterpret_cast<uintptr_t>(write.p), reinterpret_cast<uintptr_t>(write.p) + write.bytes);
        d.wq[w] = (unsigned char)q;
        d.wseq[w] = my;
        d.rseq[w][0] = d.rseq[w][1] = 0;
    }
    for (size_t i = 0; i < n_reads; ++i) {
        if (!reads[i].p || !reads[i].bytes) continue;
        const int r = use_slot(d, reinterpret_cast<uintptr_t>(reads[i].p), reinterpret_cast<uintptr_t>(reads[i].p) + reads[i].bytes);
        d.rseq[r][q] = my;
    }
    tls.op_queue = q;
    *s = queue_stream(dev, q);
    return SMHIP_OK;
}

namespace {
// Bytes an operator in progress allocated for itself (partial sums, written-out periods, the temporaries of a cut chain): a
// write of that operator, ordered behind whoever used those bytes before.
int note_scratch(void *p, size_t bytes) {
    if (tls.op_depth <= 0 || tls.use_user_stream) return SMHIP_OK;
    const int dev = tls.device;
    Dispatch &d = g_dispatch[dev];
    std::lock_guard<std::recursive_mutex> lock(d.m);
    if (d.single || !d.q1) return SMHIP_OK;  // one queue: stream order
    const int q = tls.op_queue;
    uint64_t need[2];
    dependencies(d, nullptr, 0, Span{p, bytes}, need);
    if (int rc = queue_wait(d, dev, q, q ^ 1, need[q ^ 1] > d.floor[q ^ 1] ? need[q ^ 1] : d.floor[q ^ 1])) return rc;
    const int w = use_slot(d, reinterpret_cast<uintptr_t>(p), reinterpret_cast<uintptr_t>(p) + bytes);
    d.wq[w] = (unsigned char)q;
    d.wseq[w] = d.seq[q];
    d.rseq[w][0] = d.rseq[w][1] = 0;
    return SMHIP_OK;
}
// The library's stream handle is about to order something OUTSIDE the dispatcher (a caller's stream, a host wait, a peer
// copy): make queue 0's tail stand for all the library's work on the device.
void queues_join(int dev) {
    Dispatch &d = g_dispatch[dev];
    std::lock_guard<std::recursive_mutex> lock(d.m);
    if (d.q1) (void)queue_wait(d, dev, 0, 1, d.seq[1]);
}
// The host has just waited for queue 0 behind a barrier: everything either queue was given is finished, so nothing on
// record constrains the choice of queue any more.  (Without this an operator whose operands were last written long ago
// on queue 0 would stay on queue 0 for ever: an edge to a finished operator costs nothing on the GPU, but the policy avoids
// edges.)
void queues_all_complete(int dev) {
    Dispatch &d = g_dispatch[dev];
    std::lock_guard<std::recursive_mutex> lock(d.m);
    d.synced[0][1] = d.seq[1];
    d.synced[1][0] = d.seq[0];
}
// Queue 0 was just made to wait for something outside the dispatcher: queue 1's next operator orders itself behind that.
void queues_after_external_wait(int dev) {
    Dispatch &d = g_dispatch[dev];
    std::lock_guard<std::recursive_mutex> lock(d.m);
    d.floor[0] = ++d.seq[0];
}
}  // namespace

void dispatch_single_queue(int dev, bool single) {
    if (dev < 0 || dev >= kMaxDevices) return;
    Dispatch &d = g_dispatch[dev];
    std::lock_guard<std::recursive_mutex> lock(d.m);
    if (single && d.q1) (void)queue_wait(d, dev, 0, 1, d.seq[1]);
    if (single) d.floor[0] = ++d.seq[0];
    d.single = single;
}

int fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    tls.error = buf;
    return code;
}

// A caller outside an operator's scope that wants "the" library stream of the device (the sharded entry points, which walk
// the device group from one host thread): from then on the device runs on one queue.
int acquire(hipStream_t *stream) {
    if (int rc = acquire_stream(stream)) return rc;
    if (!tls.use_user_stream && tls.op_depth == 0) dispatch_single_queue(tls.device, true);
    return SMHIP_OK;
}

namespace {
int acquire_stream_quietly(hipStream_t *stream);
}
int acquire_stream(hipStream_t *stream) {
    if (int rc = acquire_stream_quietly(stream)) return rc;
    // whoever wants the library's stream is about to queue or observe something: the tiny operators recorded so far go first
    // (tiny.hip).  Not from inside an operator -- its own begin() has been here already -- and not on a caller's stream.
    if (tls.op_depth == 0 && !tls.use_user_stream) return tiny_flush_device(tls.device);
    return SMHIP_OK;
}
namespace {
// ... without that: smhip_alloc selects the device and may order a recycled block behind its previous users, but it neither
// queues nor observes results.
int acquire_stream_quietly(hipStream_t *stream) {
    {
        std::lock_guard<std::mutex> lock(g_mutex);
        if (g_device_count < 0) {
            int n = 0;
            hipError_t e = hipGetDeviceCount(&n);
            if (e != hipSuccess || n <= 0) {
                (void)hipGetLastError();
                return fail(SMHIP_ERR_NO_DEVICE,
                            "no HIP device available (%s); libsmhip has no CPU fallback",
                            e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
            }
            g_device_count = n < kMaxDevices ? n : kMaxDevices;
        }
    }
    if (tls.device < 0) tls.device = 0;
    if (tls.device >= g_device_count) return fail(SMHIP_ERR_INVALID, "device %d out of range (%d devices)", tls.device, g_device_count);
    {   // a thread that stays on one device (the usual case) pays a thread-local read here, not a hipSetDevice
        int cur = -1;
        if (hipGetDevice(&cur) != hipSuccess || cur != tls.device) SMHIP_TRY(hipSetDevice(tls.device));
    }
    if (!tls.checked[tls.device]) {
        hipDeviceProp_t prop;
        SMHIP_TRY(hipGetDeviceProperties(&prop, tls.device));
        if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
            return fail(SMHIP_ERR_NO_DEVICE, "device %d is %s; libsmhip carries gfx950 (MI355X) code only", tls.device, prop.gcnArchName);
        g_cus[tls.device] = prop.multiProcessorCount;
        tls.checked[tls.device] = true;
    }
    if (tls.use_user_stream) {
        *stream = tls.user_stream;
        return SMHIP_OK;
    }
    {
        std::lock_guard<std::mutex> lock(g_mutex);
        if (!g_streams[tls.device]) SMHIP_TRY(hipStreamCreateWithFlags(&g_streams[tls.device], hipStreamNonBlocking));
        *stream = g_streams[tls.device];
    }
    return SMHIP_OK;
}
}  // namespace

// Is the calling thread in a state in which a tiny operator may be recorded instead of launched: on the library's queue of a
// device that has been checked, outside any operator?
bool tiny_context(int *dev) {
    const int d = tls.device;
    if (d < 0 || tls.use_user_stream || tls.op_depth != 0 || !tls.checked[d] || !g_streams[d]) return false;
    {   // a caller that holds the library's stream handle (smhip_get_stream) or drives the device group orders its own work by
        // what it has seen enqueued: on such a device every operator is a launch at once, as on one queue
        Dispatch &q = g_dispatch[d];
        std::lock_guard<std::recursive_mutex> lock(q.m);
        if (q.single) return false;
    }
    *dev = d;
    return true;
}

size_t piece_for(size_t n_vec, int streams) {
    static const long forced = [] {  // -1: the built-in rule; 0: never split; k: pieces of 2^k above 2^k
        const char *e = getenv("SMHIP_PIECE_LOG2VEC");
        if (!e) return -1L;
        const long k = atol(e);
        return k <= 0 ? 0L : (k < 12 ? 12L : (k > 40 ? 40L : k));  // 1 << k below: keep it a shift the type can hold
    }();
    if (forced == 0) return 0;
    if (forced > 0) return n_vec > ((size_t)1 << forced) ? (size_t)1 << forced : 0;
    if (streams >= 3) {
        if (n_vec > ((size_t)1 << 26)) return (size_t)1 << 24;
        if (n_vec > ((size_t)1 << 25)) return (size_t)1 << 25;
        return 0;
    }
    return n_vec > ((size_t)1 << 27) ? (size_t)1 << 25 : 0;
}

hipEvent_t pool_event_take(int dev) { return take_event(dev); }
void pool_event_give(int dev, hipEvent_t e) { give_event(dev, e); }

int compute_units() {
    const int d = tls.device < 0 ? 0 : tls.device;
    return g_cus[d] > 0 ? g_cus[d] : 256;
}

int current_device() { return tls.device < 0 ? 0 : tls.device; }

int stream_policy(size_t bytes_read, size_t bytes_written) {
    static const bool stores_always_nt = [] { const char *e = getenv("SMHIP_STORE_POLICY"); return e && strcmp(e, "nt") == 0; }();
    int policy = stream_reads(bytes_read) ? kPolicyLoadNt : 0;
    const size_t footprint = bytes_read + bytes_written;
    if (!stores_always_nt && footprint >= kStoreKeepFloor && footprint <= kInfinityCacheBytes) policy |= kPolicyStoreKeep;
    return policy;
}

// ---- residency tracker (internal.h: stream_policy(reads, write)) ----
namespace {
struct Touch { uintptr_t lo = 0, hi = 0; uint64_t tick = 0; };
struct Tracker {
    std::mutex m;
    uint64_t tick = 0;  // bytes the library's launches have moved on this device
    Touch ring[32];
    int next = 0;
};
Tracker g_track[kMaxDevices];

bool warm(const Tracker &t, const Span &sp) {
    const uintptr_t lo = reinterpret_cast<uintptr_t>(sp.p), hi = lo + sp.bytes;
    for (const Touch &e : t.ring)
        if (e.hi > e.lo && e.lo <= lo && hi <= e.hi && t.tick - e.tick <= kWarmWindow) return true;
    return false;
}
void touch(Tracker &t, const Span &sp) {
    if (!sp.p || sp.bytes < kTrackFloor) return;
    const uintptr_t lo = reinterpret_cast<uintptr_t>(sp.p), hi = lo + sp.bytes;
    for (Touch &e : t.ring)
        if (e.hi > e.lo && e.lo <= lo && hi <= e.hi) {  // inside a span already on record (the same array, or a view of it)
            if (e.lo == lo && e.hi == hi) { e.tick = t.tick; return; }
        }
    for (Touch &e : t.ring)
        if (e.hi > e.lo && lo <= e.lo && e.hi <= hi) e = Touch{};  // a wider touch replaces the records it covers
    t.ring[t.next] = Touch{lo, hi, t.tick};
    t.next = (t.next + 1) % 32;
}
}  // namespace

// Bytes whose contents did not come from a library kernel -- a block handed back to the pool (its next owner's data is
// somebody else's), an upload, a peer copy's destination -- are in no cache: the records that say otherwise go.
void residency_forget(int dev, const void *p, size_t bytes) {
    if (!p || bytes < kTrackFloor || dev < 0 || dev >= kMaxDevices) return;
    Tracker &t = g_track[dev];
    const uintptr_t lo = reinterpret_cast<uintptr_t>(p), hi = lo + bytes;
    std::lock_guard<std::mutex> lock(t.m);
    for (Touch &e : t.ring)
        if (e.hi > e.lo && e.lo < hi && lo < e.hi) e = Touch{};
}

int stream_policy(std::initializer_list<Span> reads, Span write) {
    size_t bytes_read = 0;
    for (const Span &r : reads) bytes_read += r.bytes;
    return refine_policy(stream_policy(bytes_read, write.bytes), reads, write);
}

int refine_policy(int policy, std::initializer_list<Span> reads, Span write) { return refine_policy(policy, reads.begin(), reads.size(), write); }

int refine_policy(int policy, const Span *reads_begin, size_t n_reads, Span write) {
    struct Range { const Span *b, *e; const Span *begin() const { return b; } const Span *end() const { return e; } } reads{reads_begin, reads_begin + n_reads};
    size_t bytes_read = 0;
    for (const Span &r : reads) bytes_read += r.bytes;
    static const bool off = [] { const char *e = getenv("SMHIP_RESIDENCY"); return e && strcmp(e, "off") == 0; }();
    if (off) return policy;
    Tracker &t = g_track[current_device()];
    std::lock_guard<std::mutex> lock(t.m);
    if (!(policy & kPolicyLoadNt)) {
        size_t considered = 0, cold = 0;
        for (const Span &r : reads) {
            if (!r.p || r.bytes < kTrackFloor) continue;  // small operands live in the L2s whatever the hint says
            considered += r.bytes;
            if (!warm(t, r)) cold += r.bytes;
        }
        if (considered && 2 * cold > considered) policy |= kPolicyLoadNt;
    }
    t.tick += bytes_read + write.bytes;
    for (const Span &r : reads) touch(t, r);
    touch(t, write);
    return policy;
}

namespace {
// Orders stream `to` after everything queued on `from` so far (both on device `dev`); best effort, nothing blocks on the host.
void order_stream_after(int dev, hipStream_t from, hipStream_t to) {
    if (!from || !to || from == to) return;
    hipEvent_t e = take_event(dev);
    if (!e) { (void)hipStreamSynchronize(from); return; }
    if (hipEventRecord(e, from) == hipSuccess) (void)hipStreamWaitEvent(to, e, 0);
    (void)hipGetLastError();
    give_event(dev, e);
}
}  // namespace

// The sharded entry points run on the per-device LIBRARY streams.  A caller that brought its own stream (smhip_set_stream:
// torch's current stream, say) has queued the producers of its buffers there, and will queue their consumers there: on
// the caller's own device the library stream is therefore ordered after the caller's stream on entry and the caller's
// stream after the library stream on exit (without this the sharded kernels could read a[g] before the
// caller's kernel had written it).  Buffers on OTHER devices were not produced on this thread's stream; their producers
// are the caller's to order (smhip_sharded_synchronize, or per-device streams of its own).
ThreadDeviceScope::ThreadDeviceScope(int device) : prev_device_(tls.device), prev_use_user_(tls.use_user_stream) {
    tls.device = device;
    tls.use_user_stream = false;
    if (prev_use_user_ && tls.user_stream && device == prev_device_) {
        hipStream_t lib = nullptr;
        if (acquire(&lib) == SMHIP_OK) {
            order_stream_after(device, tls.user_stream, lib);
            bridged_ = lib;
        }
    }
}
ThreadDeviceScope::~ThreadDeviceScope() {
    if (bridged_) order_stream_after(tls.device, static_cast<hipStream_t>(bridged_), tls.user_stream);
    tls.device = prev_device_;
    tls.use_user_stream = prev_use_user_;
    if (prev_device_ >= 0) (void)hipSetDevice(prev_device_);
}

ThreadState::~ThreadState() {
    for (int d = 0; d < kMaxDevices; ++d) {
        if (ring[d]) {
            std::lock_guard<std::mutex> lock(g_ring_mutex);
            g_spare_rings[d].push_back({ring[d], ring_stream[d]});
            ring[d] = nullptr;
        }
    }
}

}  // namespace smhip

using namespace smhip;

// An entry point whose operand spans are not declared: a barrier operator (ordered behind everything, everything later
// behind it) for as long as the function runs.
#define SMHIP_ACQUIRE(stream_var)  \
    hipStream_t stream_var;        \
    OpScope op_scope_;             \
    if (int rc_ = op_scope_.begin_barrier(&stream_var)) return rc_
// An operator with declared spans: reads {ptr, bytes}..., one write.
#define SMHIP_ACQUIRE_OP(stream_var, write_span, ...)                                                         \
    hipStream_t stream_var;                                                                                  \
    OpScope op_scope_;                                                                                       \
    {                                                                                                        \
        const Span reads_[] = {__VA_ARGS__};                                                                 \
        if (int rc_ = op_scope_.begin(reads_, sizeof reads_ / sizeof reads_[0], write_span, &stream_var)) return rc_; \
    }

namespace {
// Bytes from an operand's first element to its last, through `strides` over `shape`.
// The calling thread's result slot on its current device: host pointer and the device's alias of the same 64 bytes.  A
// synchronous reduction's kernel writes its scalar there and the host reads it after the stream has drained -- the
// alternative (a pooled device buffer, hipMemcpyAsync device-to-host, the same wait) queued a copy packet and cost 5-8 us
// per call, most of a mid-size dot.
int result_slot(void **host, void **dev) {
    hipStream_t s;
    if (int rc = acquire_stream(&s)) return rc;
    const int d = tls.device;
    if (!tls.result_host[d]) {
        void *h = nullptr, *p = nullptr;
        SMHIP_TRY(hipHostMalloc(&h, 64, hipHostMallocMapped));
        if (hipHostGetDevicePointer(&p, h, 0) != hipSuccess) {
            (void)hipGetLastError();
            (void)hipHostFree(h);
            return fail(SMHIP_ERR_HIP, "no device alias for the pinned result slot");
        }
        tls.result_host[d] = h;
        tls.result_dev[d] = p;
    }
    *host = tls.result_host[d];
    *dev = tls.result_dev[d];
    return SMHIP_OK;
}
size_t span_bytes(const int64_t *shape, const int64_t *strides, int ndim, size_t esz) {
    size_t last = 0;
    for (int i = 0; i < ndim; ++i)
        if (shape[i] > 0) last += (size_t)(shape[i] - 1) * (size_t)strides[i];
    return (last + 1) * esz;
}
}  // namespace

extern "C" {

const char *smhip_version(void) { return "smhip 0.2 (gfx950)"; }
const char *smhip_last_error(void) { return tls.error.c_str(); }

int smhip_device_count(int *count) {
    if (!count) return fail(SMHIP_ERR_INVALID, "device_count: null");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        n = 0;
    }
    *count = n;
    return SMHIP_OK;
}

int smhip_set_device(int device) {
    if (device < 0 || device >= kMaxDevices) return fail(SMHIP_ERR_INVALID, "set_device: %d", device);
    const int prev = tls.device;
    // leaving a device: what this thread's calls recorded there (tiny.hip) is launched before the thread looks elsewhere
    if (prev >= 0 && prev != device && !tls.use_user_stream && tls.op_depth == 0) (void)tiny_flush_device(prev);
    tls.device = device;
    hipStream_t s;
    if (int rc = acquire_stream(&s)) {
        tls.device = prev;
        return rc;
    }
    return SMHIP_OK;
}

int smhip_get_device(int *device) {
    if (!device) return fail(SMHIP_ERR_INVALID, "get_device: null");
    *device = tls.device < 0 ? 0 : tls.device;
    return SMHIP_OK;
}

int smhip_set_stream(void *hip_stream) {
    hipStream_t next = static_cast<hipStream_t>(hip_stream);
    if (tls.device >= 0) {
        // Order the new stream after the old one, so that buffers this thread used on the old stream and frees (or
        // reuses) under the new one stay stream-ordered (the pool tags a freed block with the CURRENT stream).
        hipStream_t prev = tls.use_user_stream ? tls.user_stream : g_streams[tls.device];
        hipStream_t now = next ? next : g_streams[tls.device];
        if (prev && now && prev != now) {
            if (prev == g_streams[tls.device]) {
                if (!tls.use_user_stream && tls.op_depth == 0) (void)tiny_flush_device(tls.device);  // what was recorded for the library's queue is launched there first
                queues_join(tls.device);  // the library's first queue speaks for both
            }
            hipEvent_t e = take_event(tls.device);
            if (e) {
                if (hipEventRecord(e, prev) == hipSuccess) (void)hipStreamWaitEvent(now, e, 0);
                (void)hipGetLastError();  // a previous stream that is already gone has nothing left to order
                give_event(tls.device, e);
            }
            if (now == g_streams[tls.device]) queues_after_external_wait(tls.device);
        }
    }
    tls.user_stream = next;
    tls.use_user_stream = next != nullptr;
    return SMHIP_OK;
}

int smhip_get_stream(void **hip_stream) {
    if (!hip_stream) return fail(SMHIP_ERR_INVALID, "get_stream: null");
    hipStream_t s;
    if (int rc = acquire(&s)) return rc;  // the handle leaves the library: the device runs on ONE queue from here on (stream order is what the caller sees)
    *hip_stream = s;
    return SMHIP_OK;
}

int smhip_synchronize(void) {
    SMHIP_ACQUIRE(s);  // a barrier: queue 0 is behind everything on queue 1
    SMHIP_TRY(hipStreamSynchronize(s));  // (polling hipStreamQuery instead of this sleeping wait was tried for the reductions' scalars: no faster, profiles/r04_reduce_mid_rates.txt)
    if (!tls.use_user_stream) queues_all_complete(tls.device);
    return SMHIP_OK;
}

/* ---------------------------------------------------------------- memory */

int smhip_alloc(void **dptr, size_t bytes) {
    if (!dptr) return fail(SMHIP_ERR_INVALID, "alloc: null");
    hipStream_t s;  // no operator: the queues are not touched (who uses the bytes, and on which queue, is the operators' business)
    if (int rc = acquire_stream_quietly(&s)) return rc;
    const size_t cls = size_class(bytes);
    const int dev = tls.device;
    void *p = nullptr;
    Tag tag;
    Arena *arena = nullptr;
    size_t off = 0;
    if (cls >= kLargeMin) {
        bool found;
        {
            std::lock_guard<std::mutex> lock(g_mutex);
            found = carve(dev, cls, &arena, &off, &tag);
        }
        if (!found) {
            // a new slab: room for the request and the operands that usually come with it
            size_t want = cls > kArenaSolo ? cls : (4 * cls > kArenaFloor ? 4 * cls : kArenaFloor);
            void *base = nullptr;
            hipError_t e = hipMalloc(&base, want);
            if (e == hipErrorOutOfMemory && want > cls) {
                (void)hipGetLastError();
                want = cls;
                e = hipMalloc(&base, want);
            }
            if (e == hipErrorOutOfMemory) {
                (void)hipGetLastError();
                smhip_pool_trim();
                e = hipMalloc(&base, want);
            }
            if (e != hipSuccess) return fail(SMHIP_ERR_HIP, "hipMalloc(%zu): %s", want, hipGetErrorString(e));
            std::lock_guard<std::mutex> lock(g_mutex);
            Arena *a = new Arena;
            a->base = static_cast<char *>(base);
            a->size = want;
            a->device = dev;
            a->free.push_back(Extent{0, want, Tag{}});
            g_arenas.push_back(a);
            g_bytes_cached += want;
            found = carve(dev, cls, &arena, &off, &tag);
            if (!found) return fail(SMHIP_ERR_HIP, "pool: fresh arena could not satisfy %zu bytes", cls);
        }
        p = arena->base + off;
    } else {
        {
            std::lock_guard<std::mutex> lock(g_mutex);
            auto it = g_free.find({dev, cls});
            if (it != g_free.end() && !it->second.empty()) {
                p = it->second.back().first;
                tag = std::move(it->second.back().second);
                it->second.pop_back();
            }
        }
        if (!p) {
            hipError_t e = hipMalloc(&p, cls);
            if (e == hipErrorOutOfMemory) {
                (void)hipGetLastError();
                smhip_pool_trim();
                e = hipMalloc(&p, cls);
            }
            if (e != hipSuccess) return fail(SMHIP_ERR_HIP, "hipMalloc(%zu): %s", cls, hipGetErrorString(e));
            std::lock_guard<std::mutex> lock(g_mutex);
            g_bytes_cached += cls;
        }
    }
    // stream-ordered reuse: work queued by the previous owner on another stream is ordered before ours
    if (int rc = order_after(tag, dev, s)) {
        std::lock_guard<std::mutex> lock(g_mutex);  // hand the bytes back rather than leak them
        Tag lost;
        lost.unknown = true;
        if (arena) release(arena, off, cls, std::move(lost));
        else g_free[{dev, cls}].push_back({p, std::move(lost)});
        return rc;
    }
    {
        std::lock_guard<std::mutex> lock(g_mutex);
        g_live[p] = Block{cls, dev, arena, off, s, std::this_thread::get_id()};
        g_bytes_live += cls;
        g_bytes_cached -= cls;
    }
    *dptr = p;
    return note_scratch(p, cls);  // inside an operator: its scratch, ordered behind the bytes' previous users on the other queue
}

int smhip_free(void *dptr) {
    if (!dptr) return SMHIP_OK;
    if (tls.op_depth == 0 && tiny_any_recorded()) {  // a recorded tiny operator may still refer to it: then the block goes back to the pool after that launch (tiny.hip)
        // (not from inside an operator: what an operator frees is its own scratch, which no caller ever saw -- and its scope
        // holds the dispatcher, which a flush in progress on another thread is waiting for)
        int dev = -1;
        size_t cls = 0;
        {
            std::lock_guard<std::mutex> lock(g_mutex);
            auto it = g_live.find(dptr);
            if (it != g_live.end()) { dev = it->second.device; cls = it->second.cls; }
        }
        if (dev >= 0 && tiny_defer_free(dev, dptr, cls)) return SMHIP_OK;
    }
    Block b;
    {
        std::lock_guard<std::mutex> lock(g_mutex);
        auto it = g_live.find(dptr);
        if (it == g_live.end()) return fail(SMHIP_ERR_INVALID, "free: %p was not allocated by smhip_alloc", dptr);
        b = it->second;
        g_live.erase(it);
        g_bytes_live -= b.cls;
        g_bytes_cached += b.cls;
    }
    residency_forget(b.device, dptr, b.cls);
    // Who may still be working on it: the stream it was handed to, and the stream this thread's work goes to now.
    //   handed to the library stream                      -> flag, ordered lazily at reuse;
    //   handed to the stream this thread is on            -> an event recorded on it now;
    //   handed to another stream BY THIS THREAD           -> covered: smhip_set_stream ordered every later stream of this
    //                                                        thread after it, and the current one is tagged below;
    //   handed to another thread's own stream             -> that stream may be gone by now and cannot be asked: the
    //                                                        next owner synchronises the device (rare, always safe).
    Tag tag;
    const int here = tls.device < 0 ? 0 : tls.device;
    hipStream_t cur = nullptr;
    if (here == b.device) cur = tls.use_user_stream ? tls.user_stream : g_streams[b.device];
    if (b.stream == g_streams[b.device]) tag.lib = true;
    else if (b.stream == cur) tag_stream(tag, b.device, cur);
    else if (b.owner != std::this_thread::get_id()) tag.unknown = true;
    else if (here != b.device) tag.unknown = true;  // this thread's own stream on ANOTHER device than the one it is on now:
                                                    // which of its streams that was cannot be said from here
    if (cur && cur != b.stream) tag_stream(tag, b.device, cur);
    std::lock_guard<std::mutex> lock(g_mutex);
    if (b.arena) release(b.arena, b.off, b.cls, std::move(tag));
    else g_free[{b.device, b.cls}].push_back({dptr, std::move(tag)});
    return SMHIP_OK;
}

int smhip_pool_trim(void) {
    std::vector<std::pair<int, void *>> victims;
    {
        std::lock_guard<std::mutex> lock(g_mutex);
        for (auto &kv : g_free) {
            for (auto &e : kv.second) {
                victims.push_back({kv.first.first, e.first});
                g_bytes_cached -= kv.first.second;
            }
            kv.second.clear();
        }
        for (size_t i = 0; i < g_arenas.size();) {  // slabs nobody is using any more
            Arena *a = g_arenas[i];
            if (a->used == 0) {
                victims.push_back({a->device, a->base});
                g_bytes_cached -= a->size;
                delete a;
                g_arenas.erase(g_arenas.begin() + i);
            } else {
                ++i;
            }
        }
    }
    int cur = 0;
    (void)hipGetDevice(&cur);
    for (auto &v : victims) {
        (void)hipSetDevice(v.first);
        (void)hipDeviceSynchronize();
        (void)hipFree(v.second);
    }
    (void)hipSetDevice(cur);
    return SMHIP_OK;
}

int smhip_tiny_stats(unsigned long long *launches, unsigned long long *operators) {
    tiny_stats(tls.device < 0 ? 0 : tls.device, launches, operators);
    return SMHIP_OK;
}

int smhip_pool_stats(size_t *bytes_in_use, size_t *bytes_cached) {
    if (tls.device >= 0 && !tls.use_user_stream && tls.op_depth == 0) (void)tiny_flush_device(tls.device);  // blocks freed under a recorded tiny operator return with its launch
    std::lock_guard<std::mutex> lock(g_mutex);
    if (bytes_in_use) *bytes_in_use = g_bytes_live;
    if (bytes_cached) *bytes_cached = g_bytes_cached;
    return SMHIP_OK;
}

int smhip_upload(void *dst, const void *src_host, size_t bytes) {
    if (bytes == 0) return SMHIP_OK;
    if (!dst || !src_host) return fail(SMHIP_ERR_INVALID, "upload: null");
    if (bytes <= 256 && bytes % 4 == 0 && reinterpret_cast<uintptr_t>(dst) % 4 == 0) {  // a few bytes: they ride in a recorded copy (tiny.hip), no copy packet
        const int64_t one = 1, shape1 = (int64_t)(bytes / 4);
        bool taken;
        if (int rc = tiny_try_enqueue(0, SMHIP_I32, src_host, bytes, &one, nullptr, 0, &one, &shape1, 1, nullptr, dst, &taken, 2)) return rc;
        if (taken) return SMHIP_OK;
    }
    SMHIP_ACQUIRE(s);
    residency_forget(tls.device, dst, bytes);
    // Small uploads (the reference's simple_check builds 25-element arrays per iteration) go through a
    // pinned ring: the caller's buffer is free again on return and nothing waits for the GPU.  The
    // stream is drained only when the ring wraps (every kRingBytes of small uploads) or changes stream.
    constexpr size_t kRingBytes = 4u << 20, kSmall = 64u << 10;
    if (bytes <= kSmall) {
        const int d = tls.device;
        if (!tls.ring[d]) {
            {
                std::lock_guard<std::mutex> lock(g_ring_mutex);
                if (!g_spare_rings[d].empty()) {  // adopt the ring of a thread that has ended
                    tls.ring[d] = g_spare_rings[d].back().ptr;
                    tls.ring_stream[d] = g_spare_rings[d].back().stream;
                    tls.ring_head[d] = kRingBytes;  // forces the drain of that stream before the first reuse
                    g_spare_rings[d].pop_back();
                }
            }
            if (!tls.ring[d]) SMHIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&tls.ring[d]), kRingBytes, hipHostMallocDefault));
        }
        const size_t need = (bytes + 255) & ~(size_t)255;
        if (tls.ring_head[d] + need > kRingBytes || tls.ring_stream[d] != s) {
            if (tls.ring_stream[d]) SMHIP_TRY(hipStreamSynchronize(tls.ring_stream[d]));
            tls.ring_head[d] = 0;
            tls.ring_stream[d] = s;
        }
        unsigned char *slot = tls.ring[d] + tls.ring_head[d];
        tls.ring_head[d] += need;
        memcpy(slot, src_host, bytes);
        SMHIP_TRY(hipMemcpyAsync(dst, slot, bytes, hipMemcpyHostToDevice, s));
        return SMHIP_OK;
    }
    SMHIP_TRY(hipMemcpyAsync(dst, src_host, bytes, hipMemcpyHostToDevice, s));
    SMHIP_TRY(hipStreamSynchronize(s));  // pageable source: the caller may reuse it at once
    return SMHIP_OK;
}

int smhip_download(void *dst_host, const void *src, size_t bytes) {
    if (bytes == 0) return SMHIP_OK;
    if (!dst_host || !src) return fail(SMHIP_ERR_INVALID, "download: null");
    SMHIP_ACQUIRE(s);
    SMHIP_TRY(hipMemcpyAsync(dst_host, src, bytes, hipMemcpyDeviceToHost, s));
    SMHIP_TRY(hipStreamSynchronize(s));
    if (!tls.use_user_stream) queues_all_complete(tls.device);
    return SMHIP_OK;
}

int smhip_copy(void *dst, const void *src, size_t bytes) {
    if (bytes == 0) return SMHIP_OK;
    if (!dst || !src) return fail(SMHIP_ERR_INVALID, "copy: null");
    if (bytes <= 4 * (size_t)kTinyMaxResults && bytes % 4 == 0 && (reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(src)) % 4 == 0) {
        const int64_t one = 1, shape1 = (int64_t)(bytes / 4);
        bool taken;
        if (int rc = tiny_try_enqueue(0, SMHIP_I32, src, 0, &one, nullptr, 0, &one, &shape1, 1, nullptr, dst, &taken, 2)) return rc;
        if (taken) return SMHIP_OK;
    }
    SMHIP_ACQUIRE_OP(s, (Span{dst, bytes}), Span{src, bytes});
    // the array kernel streams a copy at 82 % of HBM peak; hipMemcpyAsync device-to-device gave 67 % (tools/misc_rates.py)
    if (bytes % 4 == 0 && (reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(src)) % 4 == 0) {
        const int32_t unused = 0;
        return launch_array_scalar(SMHIP_OP_LEFT, SMHIP_I32, src, &unused, bytes / 4, dst, s);
    }
    SMHIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, s));
    return SMHIP_OK;
}

int smhip_fill(int dtype, void *dst, const void *value_host, size_t n) {
    if (!valid_dtype(dtype)) return fail(SMHIP_ERR_INVALID, "fill: bad dtype %d", dtype);
    if (n == 0) return SMHIP_OK;
    if (!dst || !value_host) return fail(SMHIP_ERR_INVALID, "fill: null");
    if ((int64_t)n <= kTinyMaxResults) {  // tiny: recorded (tiny.hip)
        const int64_t one = 1, shape1 = (int64_t)n;
        bool taken;
        if (int rc = tiny_try_enqueue(0, dtype, nullptr, 0, &one, nullptr, 0, &one, &shape1, 1, value_host, dst, &taken, 1)) return rc;
        if (taken) return SMHIP_OK;
    }
    SMHIP_ACQUIRE_OP(s, (Span{dst, n * dtype_size(dtype)}), Span{nullptr, 0});
    return launch_fill(dtype, dst, value_host, n, s);
}

int smhip_fill_uniform_f32(float *dst, size_t n, uint64_t seed, uint64_t first, float lo, float hi) {
    if (n == 0) return SMHIP_OK;
    if (!dst) return fail(SMHIP_ERR_INVALID, "fill_uniform_f32: null");
    SMHIP_ACQUIRE_OP(s, (Span{dst, n * sizeof(float)}), Span{nullptr, 0});
    return launch_fill_uniform_f32(dst, n, seed, first, lo, hi, s);
}

/* ----------------------------------------------------------- shape layer */

int smhip_broadcast(int nd1, const int64_t *shape1, const int64_t *strides1, int nd2, const int64_t *shape2,
                    const int64_t *strides2, int64_t *result_shape, int64_t *new_strides1, int64_t *new_strides2,
                    int64_t *total_size) {
    if (nd1 < 0 || nd2 < 0 || (nd1 && (!shape1 || !strides1)) || (nd2 && (!shape2 || !strides2)))
        return fail(SMHIP_ERR_INVALID, "broadcast: bad arguments");
    const int nd = nd1 > nd2 ? nd1 : nd2;
    if (nd && (!result_shape || !new_strides1 || !new_strides2)) return fail(SMHIP_ERR_INVALID, "broadcast: null output");
    const int off1 = nd - nd1, off2 = nd - nd2;
    int64_t total = 1;
    for (int i = 0; i < nd; ++i) {
        // right-align; absent leading dims count as extent 1 / stride 0 (SMUtils.h:51-72)
        const int64_t d1 = i < off1 ? 1 : shape1[i - off1], d2 = i < off2 ? 1 : shape2[i - off2];
        int64_t s1 = i < off1 ? 0 : strides1[i - off1], s2 = i < off2 ? 0 : strides2[i - off2];
        if (d1 != d2 && d1 != 1 && d2 != 1)
            return fail(SMHIP_ERR_BROADCAST, "Cannot broadcast shapes: incompatible dimensions");  // SMUtils.h:76-78
        const int64_t d = d1 > d2 ? d1 : d2;
        if (d1 == 1 && d2 > 1) s1 = 0;  // SMUtils.h:83-88
        if (d2 == 1 && d1 > 1) s2 = 0;
        result_shape[i] = d;
        new_strides1[i] = s1;
        new_strides2[i] = s2;
        total *= d;
    }
    if (total_size) *total_size = total;
    return nd;
}

int smhip_is_contiguous(int ndim, const int64_t *shape, const int64_t *strides) {
    int64_t expected = 1;
    for (int i = ndim - 1; i >= 0; --i) {
        if (strides[i] != expected) return 0;
        expected *= shape[i];
    }
    return 1;
}

int smhip_register_op(const char *hip_expression, int *op_id) {
    if (!hip_expression || !*hip_expression || !op_id) return fail(SMHIP_ERR_INVALID, "register_op: null / empty expression");
    return jit_register(hip_expression, op_id);
}

/* -------------------------------------------------------------- hot path */

int smhip_elementwise(int op, int dtype, const void *a, const int64_t *stride_a, const void *b, const int64_t *stride_b,
                      const int64_t *shape, int ndim, void *out) {
    if ((!valid_op(op) && !user_op(op)) || !valid_dtype(dtype)) return fail(SMHIP_ERR_INVALID, "elementwise: bad op %d / dtype %d", op, dtype);
    if (ndim < 1 || ndim > SMHIP_MAX_NDIM)
        return fail(SMHIP_ERR_INVALID, "elementwise: ndim %d outside 1..%d (the reference's MAX_NDIM, helpers.h:4)", ndim, SMHIP_MAX_NDIM);
    if (!stride_a || !stride_b || !shape) return fail(SMHIP_ERR_INVALID, "elementwise: null shape/stride");
    int64_t n = 1;
    for (int i = 0; i < ndim; ++i) {
        if (shape[i] < 0 || stride_a[i] < 0 || stride_b[i] < 0) return fail(SMHIP_ERR_INVALID, "elementwise: negative extent or stride at dim %d", i);
        n *= shape[i];
    }
    if (n == 0) return SMHIP_OK;
    if (!a || !b || !out) return fail(SMHIP_ERR_INVALID, "elementwise: null buffer");
    const size_t esz = dtype_size(dtype);
    if ((int64_t)n <= kTinyMaxResults) {  // tiny: recorded, several operators to a launch (tiny.hip)
        bool taken;
        if (int rc = tiny_try_enqueue(op, dtype, a, 0, stride_a, b, 0, stride_b, shape, ndim, nullptr, out, &taken)) return rc;
        if (taken) return SMHIP_OK;
    }
    SMHIP_ACQUIRE_OP(s, (Span{out, (size_t)n * esz}), Span{a, span_bytes(shape, stride_a, ndim, esz)}, Span{b, span_bytes(shape, stride_b, ndim, esz)});
    return launch_broadcast(op, dtype, a, stride_a, b, stride_b, shape, ndim, out, s);
}

int smhip_elementwise_inline(int op, int dtype, const void *a, size_t a_host_bytes, const int64_t *stride_a, const void *b,
                             size_t b_host_bytes, const int64_t *stride_b, const int64_t *shape, int ndim, void *out) {
    if (!valid_dtype(dtype)) return fail(SMHIP_ERR_INVALID, "elementwise_inline: bad dtype %d", dtype);
    if (!valid_op(op)) return fail(user_op(op) ? SMHIP_ERR_UNSUPPORTED : SMHIP_ERR_INVALID, "elementwise_inline: op %d (built-in Ops only)", op);
    if (ndim < 1 || ndim > SMHIP_MAX_NDIM) return fail(SMHIP_ERR_INVALID, "elementwise_inline: ndim %d outside 1..%d", ndim, SMHIP_MAX_NDIM);
    if (!stride_a || !stride_b || !shape) return fail(SMHIP_ERR_INVALID, "elementwise_inline: null shape/stride");
    if (a_host_bytes > SMHIP_INLINE_MAX_BYTES || b_host_bytes > SMHIP_INLINE_MAX_BYTES)
        return fail(SMHIP_ERR_UNSUPPORTED, "elementwise_inline: an inline operand is limited to %d bytes", SMHIP_INLINE_MAX_BYTES);
    int64_t n = 1, span_a = 0, span_b = 0;
    for (int i = 0; i < ndim; ++i) {
        if (shape[i] < 0 || stride_a[i] < 0 || stride_b[i] < 0) return fail(SMHIP_ERR_INVALID, "elementwise_inline: negative extent or stride at dim %d", i);
        n *= shape[i];
        if (shape[i] > 0) { span_a += (shape[i] - 1) * stride_a[i]; span_b += (shape[i] - 1) * stride_b[i]; }
        if (n > SMHIP_INLINE_MAX_OUTPUTS) return fail(SMHIP_ERR_UNSUPPORTED, "elementwise_inline: more than %d results", SMHIP_INLINE_MAX_OUTPUTS);
    }
    if (n == 0) return SMHIP_OK;
    if (!a || !b || !out) return fail(SMHIP_ERR_INVALID, "elementwise_inline: null buffer");
    const int64_t esz = (int64_t)dtype_size(dtype);
    if ((a_host_bytes && (span_a + 1) * esz > (int64_t)a_host_bytes) || (b_host_bytes && (span_b + 1) * esz > (int64_t)b_host_bytes))
        return fail(SMHIP_ERR_INVALID, "elementwise_inline: the strides reach past the inline operand's bytes");
    if ((int64_t)n <= kTinyMaxResults) {
        bool taken;
        if (int rc = tiny_try_enqueue(op, dtype, a, a_host_bytes, stride_a, b, b_host_bytes, stride_b, shape, ndim, nullptr, out, &taken)) return rc;
        if (taken) return SMHIP_OK;
    }
    SMHIP_ACQUIRE(s);
    return launch_inline(op, dtype, a, a_host_bytes, stride_a, b, b_host_bytes, stride_b, shape, ndim, out, s);
}

int smhip_fused_expr(const char *hip_expression, int dtype, const void *const *operands, int n_operands, const void *scalars_host,
                     int n_scalars, void *out, size_t n) {
    if (!valid_dtype(dtype)) return fail(SMHIP_ERR_INVALID, "fused_expr: bad dtype %d", dtype);
    if (!hip_expression || !*hip_expression) return fail(SMHIP_ERR_INVALID, "fused_expr: empty expression");
    if (n_operands < 1 || n_operands > 8) return fail(SMHIP_ERR_INVALID, "fused_expr: %d operands (1..8)", n_operands);
    if (n_scalars < 0 || n_scalars > 4 || (n_scalars > 0 && !scalars_host)) return fail(SMHIP_ERR_INVALID, "fused_expr: %d scalars (0..4)", n_scalars);
    if (n == 0) return SMHIP_OK;
    if (!operands || !out) return fail(SMHIP_ERR_INVALID, "fused_expr: null buffer");
    for (int k = 0; k < n_operands; ++k)
        if (!operands[k]) return fail(SMHIP_ERR_INVALID, "fused_expr: operand %d is null", k);
    Span reads[8];
    for (int k = 0; k < n_operands; ++k) reads[k] = Span{operands[k], n * dtype_size(dtype)};
    hipStream_t s;
    OpScope op_scope_;
    if (int rc = op_scope_.begin(reads, (size_t)n_operands, Span{out, n * dtype_size(dtype)}, &s)) return rc;
    return jit_fused_expr(hip_expression, dtype, operands, n_operands, scalars_host, n_scalars, out, n, nullptr, s);
}

int smhip_fused_expr_sum_async(const char *hip_expression, int dtype, const void *const *operands, int n_operands,
                               const void *scalars_host, int n_scalars, void *out_or_null, size_t n, double *sum_dev) {
    if (!valid_dtype(dtype)) return fail(SMHIP_ERR_INVALID, "fused_expr_sum: bad dtype %d", dtype);
    if (!hip_expression || !*hip_expression) return fail(SMHIP_ERR_INVALID, "fused_expr_sum: empty expression");
    if (n_operands < 1 || n_operands > 8) return fail(SMHIP_ERR_INVALID, "fused_expr_sum: %d operands (1..8)", n_operands);
    if (n_scalars < 0 || n_scalars > 4 || (n_scalars > 0 && !scalars_host)) return fail(SMHIP_ERR_INVALID, "fused_expr_sum: %d scalars (0..4)", n_scalars);
    if (!sum_dev) return fail(SMHIP_ERR_INVALID, "fused_expr_sum: null result");
    if (n > 0 && !operands) return fail(SMHIP_ERR_INVALID, "fused_expr_sum: null buffer");
    for (int k = 0; n > 0 && k < n_operands; ++k)
        if (!operands[k]) return fail(SMHIP_ERR_INVALID, "fused_expr_sum: operand %d is null", k);
    SMHIP_ACQUIRE(s);
    if (n == 0) {
        SMHIP_TRY(hipMemsetAsync(sum_dev, 0, sizeof(double), s));
        return SMHIP_OK;
    }
    return jit_fused_expr(hip_expression, dtype, operands, n_operands, scalars_host, n_scalars, out_or_null, n, sum_dev, s);
}

int smhip_copy_strided(int dtype, const void *src, const int64_t *src_strides, void *dst, const int64_t *dst_strides,
                       const int64_t *shape, int ndim) {
    if (!valid_dtype(dtype)) return fail(SMHIP_ERR_INVALID, "copy_strided: bad dtype %d", dtype);
    if (ndim < 1 || ndim > SMHIP_MAX_NDIM) return fail(SMHIP_ERR_INVALID, "copy_strided: ndim %d outside 1..%d", ndim, SMHIP_MAX_NDIM);
    if (!src_strides || !dst_strides || !shape) return fail(SMHIP_ERR_INVALID, "copy_strided: null shape/stride");
    int64_t n = 1;
    for (int i = 0; i < ndim; ++i) {
        if (shape[i] < 0 || src_strides[i] < 0 || dst_strides[i] < 0) return fail(SMHIP_ERR_INVALID, "copy_strided: negative extent or stride at dim %d", i);
        if (shape[i] > 1 && dst_strides[i] == 0) return fail(SMHIP_ERR_INVALID, "copy_strided: destination stride 0 at dim %d (elements would overwrite each other)", i);
        n *= shape[i];
    }
    if (n == 0) return SMHIP_OK;
    if (!src || !dst) return fail(SMHIP_ERR_INVALID, "copy_strided: null buffer");
    const size_t esz = dtype_size(dtype);
    if ((int64_t)n <= kTinyMaxResults) {
        bool taken;
        if (int rc = tiny_try_enqueue(0, dtype, src, 0, src_strides, nullptr, 0, dst_strides, shape, ndim, nullptr, dst, &taken, 2)) return rc;
        if (taken) return SMHIP_OK;
    }
    SMHIP_ACQUIRE_OP(s, (Span{dst, span_bytes(shape, dst_strides, ndim, esz)}), Span{src, span_bytes(shape, src_strides, ndim, esz)});
    return launch_copy_strided(dtype, src, src_strides, dst, dst_strides, shape, ndim, s);
}

int smhip_contiguous(int op, int dtype, const void *a, const void *b, void *out, size_t n) {
    if ((!valid_op(op) && !user_op(op)) || !valid_dtype(dtype)) return fail(SMHIP_ERR_INVALID, "contiguous: bad op %d / dtype %d", op, dtype);
    if (n == 0) return SMHIP_OK;
    if (!a || !b || !out) return fail(SMHIP_ERR_INVALID, "contiguous: null buffer");
    const size_t nbytes = n * dtype_size(dtype);
    if ((int64_t)n <= kTinyMaxResults) {
        const int64_t one = 1, shape1 = (int64_t)n;
        bool taken;
        if (int rc = tiny_try_enqueue(op, dtype, a, 0, &one, b, 0, &one, &shape1, 1, nullptr, out, &taken)) return rc;
        if (taken) return SMHIP_OK;
    }
    SMHIP_ACQUIRE_OP(s, (Span{out, nbytes}), Span{a, nbytes}, Span{b, nbytes});
    if (user_op(op)) return jit_contiguous(op, dtype, a, b, out, n, s);
    return launch_contiguous(op, dtype, a, b, out, n, s);
}

int smhip_array_scalar(int op, int dtype, const void *a, const void *value_host, size_t n, void *out) {
    if ((!valid_op(op) && !user_op(op)) || !valid_dtype(dtype)) return fail(SMHIP_ERR_INVALID, "array_scalar: bad op %d / dtype %d", op, dtype);
    if (n == 0) return SMHIP_OK;
    if (!a || !value_host || !out) return fail(SMHIP_ERR_INVALID, "array_scalar: null buffer");
    const size_t nbytes = n * dtype_size(dtype);
    if ((int64_t)n <= kTinyMaxResults) {
        const int64_t one = 1, zero = 0, shape1 = (int64_t)n;
        bool taken;
        if (int rc = tiny_try_enqueue(op, dtype, a, 0, &one, nullptr, 0, &zero, &shape1, 1, value_host, out, &taken)) return rc;
        if (taken) return SMHIP_OK;
    }
    SMHIP_ACQUIRE_OP(s, (Span{out, nbytes}), Span{a, nbytes});
    if (user_op(op)) return jit_array_scalar(op, dtype, a, value_host, n, out, s);
    return launch_array_scalar(op, dtype, a, value_host, n, out, s);
}

int smhip_fused_contiguous(int op1, int op2, int dtype, const void *a, const void *b, const void *c, const void *c_scalar_host,
                           void *out, size_t n) {
    if (!valid_op(op1) || !valid_op(op2) || !valid_dtype(dtype)) return fail(SMHIP_ERR_INVALID, "fused: bad op %d/%d or dtype %d", op1, op2, dtype);
    if (n == 0) return SMHIP_OK;
    if (!a || !b || !out || (!c && !c_scalar_host)) return fail(SMHIP_ERR_INVALID, "fused: null buffer");
    const size_t nbytes = n * dtype_size(dtype);
    SMHIP_ACQUIRE_OP(s, (Span{out, nbytes}), Span{a, nbytes}, Span{b, nbytes}, Span{c, c ? nbytes : 0});
    return launch_fused(op1, op2, dtype, a, b, c, c_scalar_host, out, n, s);
}

int smhip_fused_expr_bcast(const char *hip_expression, int dtype, const void *const *operands, const int64_t *strides, int n_operands,
                           const void *scalars_host, int n_scalars, const int64_t *shape, int ndim, void *out) {
    if (!valid_dtype(dtype)) return fail(SMHIP_ERR_INVALID, "fused_expr_bcast: bad dtype %d", dtype);
    if (!hip_expression || !*hip_expression) return fail(SMHIP_ERR_INVALID, "fused_expr_bcast: empty expression");
    if (n_operands < 1 || n_operands > 8) return fail(SMHIP_ERR_INVALID, "fused_expr_bcast: %d operands outside 1..8", n_operands);
    if (n_scalars < 0 || n_scalars > 4 || (n_scalars && !scalars_host)) return fail(SMHIP_ERR_INVALID, "fused_expr_bcast: %d scalars outside 0..4 (or none given)", n_scalars);
    if (ndim < 1 || ndim > SMHIP_MAX_NDIM) return fail(SMHIP_ERR_INVALID, "fused_expr_bcast: ndim %d outside 1..%d", ndim, SMHIP_MAX_NDIM);
    if (!operands || !strides || !shape) return fail(SMHIP_ERR_INVALID, "fused_expr_bcast: null argument");
    int64_t n = 1;
    for (int i = 0; i < ndim; ++i) {
        if (shape[i] < 0) return fail(SMHIP_ERR_INVALID, "fused_expr_bcast: negative extent at dim %d", i);
        n *= shape[i];
    }
    for (int k = 0; k < n_operands; ++k) {
        if (!operands[k]) return fail(SMHIP_ERR_INVALID, "fused_expr_bcast: operand %d is NULL", k);
        for (int i = 0; i < ndim; ++i)
            if (strides[(size_t)k * ndim + i] < 0) return fail(SMHIP_ERR_INVALID, "fused_expr_bcast: negative stride (operand %d, dim %d)", k, i);
    }
    if (n == 0) return SMHIP_OK;
    if (!out) return fail(SMHIP_ERR_INVALID, "fused_expr_bcast: null output");
    Span reads[8];
    for (int k = 0; k < n_operands; ++k) reads[k] = Span{operands[k], span_bytes(shape, strides + (size_t)k * ndim, ndim, dtype_size(dtype))};
    hipStream_t s;
    OpScope op_scope_;
    if (int rc = op_scope_.begin(reads, (size_t)n_operands, Span{out, (size_t)n * dtype_size(dtype)}, &s)) return rc;
    return launch_expr_bcast(hip_expression, dtype, operands, strides, n_operands, scalars_host, n_scalars, shape, ndim, out, s);
}

namespace {
// What smhip_chain and smhip_chain_sum check alike; *n_out = the result's element count.
int check_chain(const char *who, int dtype, int n_operands, const void *const *operands, const int64_t *strides, const void *scalars_host,
                const int *ops, const int *swapped, const int64_t *shape, int ndim, int64_t *n_out) {
    if (!valid_dtype(dtype)) return fail(SMHIP_ERR_INVALID, "%s: bad dtype %d", who, dtype);
    if (n_operands < 2 || n_operands > SMHIP_CHAIN_MAX_OPERANDS) return fail(SMHIP_ERR_INVALID, "%s: %d operands outside 2..%d", who, n_operands, SMHIP_CHAIN_MAX_OPERANDS);
    if (ndim < 1 || ndim > SMHIP_MAX_NDIM) return fail(SMHIP_ERR_INVALID, "%s: ndim %d outside 1..%d", who, ndim, SMHIP_MAX_NDIM);
    if (!operands || !strides || !ops || !swapped || !shape) return fail(SMHIP_ERR_INVALID, "%s: null argument", who);
    if (!operands[0]) return fail(SMHIP_ERR_INVALID, "%s: the first operand must be an array", who);
    int64_t n = 1;
    for (int i = 0; i < ndim; ++i) {
        if (shape[i] < 0) return fail(SMHIP_ERR_INVALID, "%s: negative extent at dim %d", who, i);
        n *= shape[i];
    }
    for (int k = 0; k < n_operands; ++k) {
        if (!operands[k]) {
            if (!scalars_host) return fail(SMHIP_ERR_INVALID, "%s: operand %d is a scalar but scalars_host is NULL", who, k);
            continue;
        }
        for (int i = 0; i < ndim; ++i)
            if (strides[(size_t)k * ndim + i] < 0) return fail(SMHIP_ERR_INVALID, "%s: negative stride (operand %d, dim %d)", who, k, i);
    }
    for (int k = 0; k + 1 < n_operands; ++k) {
        if (ops[k] == SMHIP_OP_POW) {  // r ^ scalar only: sm::pow(<expression>, s)
            if (operands[k + 1] || swapped[k]) return fail(SMHIP_ERR_UNSUPPORTED, "%s: pow (stage %d) takes the chain's value to a SCALAR power", who, k);
            continue;
        }
        if (ops[k] < SMHIP_OP_ADD || ops[k] > SMHIP_OP_DIV) return fail(SMHIP_ERR_UNSUPPORTED, "%s: op %d (stage %d) is not one of add, sub, mul, div, pow", who, ops[k], k);
    }
    *n_out = n;
    return SMHIP_OK;
}
}  // namespace

int smhip_chain(int dtype, int n_operands, const void *const *operands, const int64_t *strides, const void *scalars_host, const int *ops,
                const int *swapped, const int64_t *shape, int ndim, void *out) {
    int64_t n;
    if (int rc = check_chain("chain", dtype, n_operands, operands, strides, scalars_host, ops, swapped, shape, ndim, &n)) return rc;
    if (n == 0) return SMHIP_OK;
    if (!out) return fail(SMHIP_ERR_INVALID, "chain: null output");
    Span reads[SMHIP_CHAIN_MAX_OPERANDS];
    for (int k = 0; k < n_operands; ++k)
        reads[k] = operands[k] ? Span{operands[k], span_bytes(shape, strides + (size_t)k * ndim, ndim, dtype_size(dtype))} : Span{nullptr, 0};
    hipStream_t s;
    OpScope op_scope_;
    if (int rc = op_scope_.begin(reads, (size_t)n_operands, Span{out, (size_t)n * dtype_size(dtype)}, &s)) return rc;
    return launch_chain(dtype, n_operands, operands, strides, scalars_host, ops, swapped, shape, ndim, out, s);
}

int smhip_chain_sum_async(int dtype, int n_operands, const void *const *operands, const int64_t *strides, const void *scalars_host,
                          const int *ops, const int *swapped, const int64_t *shape, int ndim, double *sum_dev) {
    int64_t n;
    if (int rc = check_chain("chain_sum", dtype, n_operands, operands, strides, scalars_host, ops, swapped, shape, ndim, &n)) return rc;
    if (!sum_dev) return fail(SMHIP_ERR_INVALID, "chain_sum: null result");
    if (n == 0) return fail(SMHIP_ERR_INVALID, "chain_sum: empty shape");
    SMHIP_ACQUIRE(s);  // a reduction: undeclared spans, ordered behind everything (like smhip_sum_async)
    return launch_chain_sum(dtype, n_operands, operands, strides, scalars_host, ops, swapped, shape, ndim, sum_dev, s);
}

int smhip_chain_sum(int dtype, int n_operands, const void *const *operands, const int64_t *strides, const void *scalars_host, const int *ops,
                    const int *swapped, const int64_t *shape, int ndim, double *sum_host) {
    if (!sum_host) return fail(SMHIP_ERR_INVALID, "chain_sum: null result");
    void *h = nullptr, *d = nullptr;
    if (int rc = result_slot(&h, &d)) return rc;
    if (int rc = smhip_chain_sum_async(dtype, n_operands, operands, strides, scalars_host, ops, swapped, shape, ndim, static_cast<double *>(d))) return rc;
    if (int rc = smhip_synchronize()) return rc;
    *sum_host = *static_cast<const volatile double *>(h);
    return SMHIP_OK;
}

int smhip_sum_async(int dtype, const void *a, size_t n, double *out_dev) {
    if (!valid_dtype(dtype)) return fail(SMHIP_ERR_INVALID, "sum: bad dtype %d", dtype);
    if (!out_dev || (n && !a)) return fail(SMHIP_ERR_INVALID, "sum: null buffer");
    SMHIP_ACQUIRE(s);
    return launch_sum(dtype, a, n, out_dev, s);
}

int smhip_dot_async(int dtype, const void *a, const void *b, size_t n, double *out_dev) {
    if (!valid_dot_dtype(dtype)) return fail(SMHIP_ERR_INVALID, "dot: bad dtype %d", dtype);
    if (!out_dev || (n && (!a || !b))) return fail(SMHIP_ERR_INVALID, "dot: null buffer");
    SMHIP_ACQUIRE(s);
    return launch_dot(dtype, a, b, n, out_dev, nullptr, s);
}

int smhip_contiguous_sum_async(int op, int dtype, const void *a, const void *b, void *out, size_t n, double *sum_dev) {
    if (!valid_op(op) || !valid_dtype(dtype)) return fail(SMHIP_ERR_INVALID, "contiguous_sum: bad op %d / dtype %d", op, dtype);
    if (!sum_dev || (n && (!a || !b || !out))) return fail(SMHIP_ERR_INVALID, "contiguous_sum: null buffer");
    SMHIP_ACQUIRE(s);
    return launch_contiguous_sum(op, dtype, a, b, out, n, sum_dev, s);
}

int smhip_sum(int dtype, const void *a, size_t n, double *out_host) {
    if (!out_host) return fail(SMHIP_ERR_INVALID, "sum: null result");
    void *h = nullptr, *d = nullptr;
    if (int rc = result_slot(&h, &d)) return rc;
    if (int rc = smhip_sum_async(dtype, a, n, static_cast<double *>(d))) return rc;
    if (int rc = smhip_synchronize()) return rc;
    *out_host = *static_cast<const volatile double *>(h);
    return SMHIP_OK;
}

int smhip_dot(int dtype, const void *a, const void *b, size_t n, void *out_host) {
    if (!valid_dot_dtype(dtype)) return fail(SMHIP_ERR_INVALID, "dot: bad dtype %d", dtype);
    if (!out_host || (n && (!a || !b))) return fail(SMHIP_ERR_INVALID, "dot: null buffer");
    void *h = nullptr, *d = nullptr;
    if (int rc = result_slot(&h, &d)) return rc;
    {
        hipStream_t s;
        OpScope op_scope_;
        if (int rc = op_scope_.begin_barrier(&s)) return rc;
        if (int rc = launch_dot(dtype, a, b, n, nullptr, d, s)) return rc;
    }
    if (int rc = smhip_synchronize()) return rc;
    memcpy(out_host, h, dtype_size(dtype));
    return SMHIP_OK;
}

int smhip_dot_c64_async(const void *a, const void *b, size_t n, double *out2_dev) {
    if (!out2_dev || (n && (!a || !b))) return fail(SMHIP_ERR_INVALID, "dot_c64: null buffer");
    if ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b)) & 15u) return fail(SMHIP_ERR_INVALID, "dot_c64: operands must be 16-byte aligned");
    SMHIP_ACQUIRE(s);
    return launch_cdot(a, b, n, out2_dev, s);
}

int smhip_dot_c64(const void *a, const void *b, size_t n, double *out2_host) {
    if (!out2_host) return fail(SMHIP_ERR_INVALID, "dot_c64: null buffer");
    void *h = nullptr, *d = nullptr;
    if (int rc = result_slot(&h, &d)) return rc;
    if (int rc = smhip_dot_c64_async(a, b, n, static_cast<double *>(d))) return rc;
    if (int rc = smhip_synchronize()) return rc;
    memcpy(out2_host, h, 16);
    return SMHIP_OK;
}

int smhip_dot_c32_async(const void *a, const void *b, size_t n, double *out2_dev) {
    if (!out2_dev || (n && (!a || !b))) return fail(SMHIP_ERR_INVALID, "dot_c32: null buffer");
    if ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b)) & 7u) return fail(SMHIP_ERR_INVALID, "dot_c32: operands must be 8-byte aligned");
    SMHIP_ACQUIRE(s);
    return launch_cdot32(a, b, n, out2_dev, s);
}

int smhip_dot_c32(const void *a, const void *b, size_t n, float *out2_host) {
    if (!out2_host) return fail(SMHIP_ERR_INVALID, "dot_c32: null buffer");
    void *h = nullptr, *d = nullptr;
    if (int rc = result_slot(&h, &d)) return rc;
    if (int rc = smhip_dot_c32_async(a, b, n, static_cast<double *>(d))) return rc;
    if (int rc = smhip_synchronize()) return rc;
    double both[2];
    memcpy(both, h, 16);
    out2_host[0] = (float)both[0];
    out2_host[1] = (float)both[1];
    return SMHIP_OK;
}

/* ----------------------------------------------------------- diagnostics */

int smhip_policy_probe(const void *a, size_t a_bytes, const void *b, size_t b_bytes, const void *out, size_t out_bytes, int *policy) {
    if (!policy) return fail(SMHIP_ERR_INVALID, "policy_probe: null");
    *policy = stream_policy({{a, a ? a_bytes : 0}, {b, b ? b_bytes : 0}}, {out, out ? out_bytes : 0});
    return SMHIP_OK;
}

int smhip_policy_peek(const void *a, size_t a_bytes, const void *b, size_t b_bytes, const void *out, size_t out_bytes, int *policy) {
    if (!policy) return fail(SMHIP_ERR_INVALID, "policy_peek: null");
    const Span reads[2] = {{a, a ? a_bytes : 0}, {b, b ? b_bytes : 0}};
    int pol = stream_policy(reads[0].bytes + reads[1].bytes, out ? out_bytes : 0);
    if (!(pol & kPolicyLoadNt)) {  // refine_policy's rule without its bookkeeping
        Tracker &t = g_track[current_device()];
        std::lock_guard<std::mutex> lock(t.m);
        size_t considered = 0, cold = 0;
        for (const Span &r : reads) {
            if (!r.p || r.bytes < kTrackFloor) continue;
            considered += r.bytes;
            if (!warm(t, r)) cold += r.bytes;
        }
        static const bool off = [] { const char *e = getenv("SMHIP_RESIDENCY"); return e && strcmp(e, "off") == 0; }();
        if (!off && considered && 2 * cold > considered) pol |= kPolicyLoadNt;
    }
    *policy = pol;
    return SMHIP_OK;
}

int smhip_queue_stats(int *queues, unsigned long long *alternations, unsigned long long *edges) {
    const int dev = tls.device < 0 ? 0 : tls.device;
    Dispatch &d = g_dispatch[dev];
    std::lock_guard<std::recursive_mutex> lock(d.m);
    if (queues) *queues = (d.single || !queues_enabled()) ? 1 : 2;
    if (alternations) *alternations = d.alternations;
    if (edges) *edges = d.edges;
    return SMHIP_OK;
}

int smhip_launch_pieces(size_t bytes_per_operand, int streams, int *pieces) {
    if (!pieces || streams < 1) return fail(SMHIP_ERR_INVALID, "launch_pieces: bad arguments");
    const size_t n_vec = bytes_per_operand / 16, piece = piece_for(n_vec, streams);
    *pieces = piece ? (int)((n_vec + piece - 1) / piece) : 1;
    return SMHIP_OK;
}

/* ---------------------------------------------------------------- timing */

int smhip_event_create(void **event) {
    if (!event) return fail(SMHIP_ERR_INVALID, "event_create: null");
    SMHIP_ACQUIRE(s);
    (void)s;
    hipEvent_t e;
    SMHIP_TRY(hipEventCreate(&e));
    *event = e;
    return SMHIP_OK;
}

int smhip_event_record(void *event) {
    SMHIP_ACQUIRE(s);
    SMHIP_TRY(hipEventRecord(static_cast<hipEvent_t>(event), s));
    return SMHIP_OK;
}

int smhip_event_synchronize(void *event) {
    SMHIP_TRY(hipEventSynchronize(static_cast<hipEvent_t>(event)));
    return SMHIP_OK;
}

int smhip_event_elapsed_ms(void *start, void *stop, float *ms) {
    if (!ms) return fail(SMHIP_ERR_INVALID, "event_elapsed_ms: null");
    SMHIP_TRY(hipEventElapsedTime(ms, static_cast<hipEvent_t>(start), static_cast<hipEvent_t>(stop)));
    return SMHIP_OK;
}

int smhip_event_destroy(void *event) {
    if (event) SMHIP_TRY(hipEventDestroy(static_cast<hipEvent_t>(event)));
    return SMHIP_OK;
}

}  // extern "C"
