// tools/sweep_vmm.hip -- can the allocator buy address-translation reach for very large arrays?  (VERDICT r02 "next" #4)
// The f32 add holds 81-82 % of HBM peak to N = 2^28 and sags to 78.4 % at 2^30 / 77.5 % at 2^31; the counters
// (profiles/r02_pmc_translation.txt) say UTCL1 misses, about one per 2 MiB page and stream.  The pool's arenas are plain
// hipMalloc slabs.  Here the same kernel runs on memory obtained four ways:
//   malloc        hipMalloc (what the arenas do)
//   vmm-min       hipMemAddressReserve + hipMemCreate + hipMemMap, handle size = the MINIMUM granularity, VA 2 MiB aligned
//   vmm-rec       ... one handle per array, recommended granularity, VA aligned to 1 GiB
//   vmm-1g        ... 1 GiB handles (if the granularity divides), VA aligned to 1 GiB
//   ext-*         hipExtMallocWithFlags (default / fine-grained / uncached) for completeness
// N = 2^28, 2^30 (and 2^31 with `big`).  Run it under rocprofv3 --pmc TCP_UTCL1_TRANSLATION_MISS_sum to see the misses.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(1024) void add_k(const f4 *__restrict__ a, const f4 *__restrict__ b, f4 *__restrict__ o) {
    const size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x;
    __builtin_nontemporal_store(__builtin_nontemporal_load(a + i) + __builtin_nontemporal_load(b + i), o + i);
}
__global__ void init_k(float *p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 1.0f + (float)(i % 1000) * 1e-3f;
}

struct Slab {
    void *ptr = nullptr;
    size_t bytes = 0;
    bool vmm = false;
    std::vector<hipMemGenericAllocationHandle_t> handles;
};

static bool vmm_alloc(Slab &s, size_t bytes, size_t handle_bytes, size_t va_align) {
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    s.bytes = bytes;
    s.vmm = true;
    hipError_t e = hipMemAddressReserve(&s.ptr, bytes, va_align, nullptr, 0);
    if (e != hipSuccess) { printf("  hipMemAddressReserve(align %zu): %s\n", va_align, hipGetErrorString(e)); (void)hipGetLastError(); return false; }
    for (size_t off = 0; off < bytes; off += handle_bytes) {
        hipMemGenericAllocationHandle_t h;
        const size_t sz = std::min(handle_bytes, bytes - off);
        e = hipMemCreate(&h, sz, &prop, 0);
        if (e != hipSuccess) { printf("  hipMemCreate(%zu): %s\n", sz, hipGetErrorString(e)); (void)hipGetLastError(); return false; }
        s.handles.push_back(h);
        e = hipMemMap((char *)s.ptr + off, sz, 0, h, 0);
        if (e != hipSuccess) { printf("  hipMemMap: %s\n", hipGetErrorString(e)); (void)hipGetLastError(); return false; }
    }
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    e = hipMemSetAccess(s.ptr, bytes, &acc, 1);
    if (e != hipSuccess) { printf("  hipMemSetAccess: %s\n", hipGetErrorString(e)); (void)hipGetLastError(); return false; }
    return true;
}
// Round 3: two of six runs of this tool under `rocprofv3 --pmc` ended in a GPU memory access fault and a hang, both right
// after the "1 GiB handles" variant at 2^30, i.e. while that mapping was torn down or the next variant -- ONE 12 GiB handle,
// reserved right afterwards and, as the driver hands addresses out, at the SAME virtual address -- was brought up
// (gpurun_out/r03x/pmc_vmm_utcl2.log:17, gpurun_out/r03final/pmc_vmm_utcl1.log:17).  The teardown discarded every return
// code and the pmc mode did not print addresses, so the logs could not say whose address faulted.  Now: every call is
// checked and reported, the device is synchronised before the unmap, a torn-down reservation is NOT handed back until
// the program ends (the next variant cannot get the same virtual address: no stale translation can alias it), and every
// variant prints its address range and handle count in both modes.
static std::vector<std::pair<void *, size_t>> g_retired_va;
static void report(const char *what, hipError_t e) {
    if (e != hipSuccess) { printf("  %s: %s\n", what, hipGetErrorString(e)); (void)hipGetLastError(); }
}
static void slab_free(Slab &s) {
    if (!s.ptr) return;
    report("hipDeviceSynchronize before teardown", hipDeviceSynchronize());
    if (s.vmm) {
        report("hipMemUnmap", hipMemUnmap(s.ptr, s.bytes));
        for (auto h : s.handles) report("hipMemRelease", hipMemRelease(h));
        g_retired_va.push_back({s.ptr, s.bytes});  // the reservation stays until the end: its address is not reused
    } else {
        report("hipFree", hipFree(s.ptr));
    }
    s = Slab();
}
static void retire_all() {
    report("hipDeviceSynchronize at exit", hipDeviceSynchronize());
    for (auto &va : g_retired_va) report("hipMemAddressFree", hipMemAddressFree(va.first, va.second));
    g_retired_va.clear();
}

int main(int argc, char **argv) {
    const bool big = argc > 1 && !strcmp(argv[1], "big");
    const bool pmc = argc > 1 && !strcmp(argv[1], "pmc");  // under rocprofv3 --pmc: exactly 4 launches of add_k per variant, in the order printed
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    size_t gmin = 0, grec = 0;
    CK(hipMemGetAllocationGranularity(&gmin, &prop, hipMemAllocationGranularityMinimum));
    CK(hipMemGetAllocationGranularity(&grec, &prop, hipMemAllocationGranularityRecommended));
    printf("allocation granularity: minimum %zu B, recommended %zu B\n", gmin, grec);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    std::vector<int> lgs = {28, 30};
    if (big) lgs.push_back(31);
    for (int lg : lgs) {
        const size_t n = (size_t)1 << lg, bytes = 3 * n * 4;
        struct How { const char *name; int kind; size_t handle, align; };
        const size_t G1 = (size_t)1 << 30;
        const size_t M2 = (size_t)2 << 20;  // the reported granularity is 4 KiB here: 786 432 handles for 3 GiB -- not tried
        std::vector<How> hows = {{"malloc", 0, 0, 0}, {"vmm 2 MiB handles, 2 MiB VA", 1, std::max(gmin, M2), M2}, {"vmm 64 MiB handles, 1 GiB VA", 1, (size_t)64 << 20, G1},
                                 {"vmm 1 GiB handles, 1 GiB VA", 1, G1, G1}, {"vmm one handle, 1 GiB VA", 1, bytes, G1},
                                 {"ext default", 2, 0, 0}, {"ext fine-grained", 3, 0, 0}, {"ext uncached", 4, 0, 0}};
        for (auto &h : hows) {
            Slab s;
            bool ok = true;
            if (h.kind == 0) { ok = hipMalloc(&s.ptr, bytes) == hipSuccess; s.bytes = bytes; }
            else if (h.kind == 1) ok = vmm_alloc(s, bytes, h.handle, h.align);
            else {
                const unsigned flag = h.kind == 2 ? hipDeviceMallocDefault : h.kind == 3 ? hipDeviceMallocFinegrained : hipDeviceMallocUncached;
                ok = hipExtMallocWithFlags(&s.ptr, bytes, flag) == hipSuccess;
                s.bytes = bytes;
            }
            if (!ok) { printf("2^%d %-30s allocation failed\n", lg, h.name); (void)hipGetLastError(); slab_free(s); continue; }
            float *a = (float *)s.ptr, *b = a + n, *o = b + n;
            init_k<<<4096, 256>>>(a, 2 * n);
            CK(hipDeviceSynchronize());
            const unsigned grid = (unsigned)(n / 4 / 1024);
            if (pmc) {
                for (int i = 0; i < 4; ++i) add_k<<<grid, 1024>>>((const f4 *)a, (const f4 *)b, (f4 *)o);
                CK(hipDeviceSynchronize());
                printf("2^%d %-30s 4 launches   [%p, %p)  %zu handle(s)\n", lg, h.name, s.ptr, (void *)((char *)s.ptr + s.bytes), s.handles.size());
                fflush(stdout);
                slab_free(s);
                continue;
            }
            for (int i = 0; i < 5; ++i) add_k<<<grid, 1024>>>((const f4 *)a, (const f4 *)b, (f4 *)o);
            std::vector<float> ms(5);
            const int reps = lg >= 30 ? 10 : 30;
            for (auto &m : ms) {
                CK(hipEventRecord(e0));
                for (int i = 0; i < reps; ++i) add_k<<<grid, 1024>>>((const f4 *)a, (const f4 *)b, (f4 *)o);
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                CK(hipEventElapsedTime(&m, e0, e1));
                m /= reps;
            }
            std::sort(ms.begin(), ms.end());
            printf("2^%d %-30s ptr %p  %9.1f us  %6.0f GB/s  %5.1f %%\n", lg, h.name, s.ptr, ms[2] * 1e3, bytes / (ms[2] * 1e-3) * 1e-9, bytes / (ms[2] * 1e-3) / 8e12 * 100);
            fflush(stdout);
            slab_free(s);
        }
    }
    retire_all();
    return 0;
}
