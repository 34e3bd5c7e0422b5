// tools/sweep_load_flavours.hip -- the read side's encodings below the Infinity Cache: out = a * s (1R+1W) with the store
// the library's rule picks (sc1 up to 128 MiB per array, nt at 256 MiB), loads written as asm with each cache-policy
// combination, in the `same` and `chain` settings of sweep_chain.hip.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
template <int LD> __device__ __forceinline__ f4 ld(const f4* p) {
    f4 v;
    if constexpr (LD == 0) asm volatile("global_load_dwordx4 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p));
    else if constexpr (LD == 1) asm volatile("global_load_dwordx4 %0, %1, off nt\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p));
    else if constexpr (LD == 2) asm volatile("global_load_dwordx4 %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p));
    else if constexpr (LD == 3) asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p));
    else if constexpr (LD == 4) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p));
    else if constexpr (LD == 5) asm volatile("global_load_dwordx4 %0, %1, off sc1 nt\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p));
    else asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1 nt\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p));
    return v;
}
template <int LD, int ST> __global__ __launch_bounds__(256) void scal(const f4* __restrict__ a, float s, f4* __restrict__ o) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const f4 v = ld<LD>(a + i) * s;
    if constexpr (ST == 0) __builtin_nontemporal_store(v, o + i);
    else asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(o + i), "v"(v));
}
__global__ void init_k(float* p, size_t n) { for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 1.0f + (float)(i % 1000) * 1e-3f; }
int main() {
    const size_t slab_floats = (size_t)1 << 28;
    float* slab; CK(hipMalloc(&slab, slab_floats * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    init_k<<<4096, 256>>>(slab, slab_floats); CK(hipDeviceSynchronize());
    auto timed = [&](auto body, int reps) {
        int seq = 0;
        for (int i = 0; i < 16; ++i) body(seq++);
        std::vector<float> ms(5);
        for (auto& m : ms) { CK(hipEventRecord(e0)); for (int i = 0; i < reps; ++i) body(seq++); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&m, e0, e1)); m /= reps; }
        std::sort(ms.begin(), ms.end());
        return ms[2];
    };
    const char* names[7] = {"plain", "nt", "sc0", "sc1", "sc0 sc1", "sc1 nt", "sc0 sc1 nt"};
    printf("%-6s %-12s %12s %12s\n", "MiB", "load", "same %", "chain %");
    for (size_t mib : {32, 64, 128, 256}) {
        const size_t n = mib << 18, nvec = n / 4;
        float *a = slab, *o = slab + n;
        const bool st_sc1 = mib <= 128;
#define ROW(LD) { auto go = [&](const float* x, float* y) { const unsigned g = (unsigned)(nvec / 256); if (st_sc1) scal<LD, 1><<<g, 256>>>((const f4*)x, 1.0000001f, (f4*)y); else scal<LD, 0><<<g, 256>>>((const f4*)x, 1.0000001f, (f4*)y); }; \
                  const float s_ = timed([&](int) { go(a, o); }, 40); const float c_ = timed([&](int i) { (i & 1) ? go(o, a) : go(a, o); }, 40); \
                  printf("%-6zu %-12s %11.1f%% %11.1f%%\n", mib, names[LD], 8.0 * n / s_ * 1e-6 / 80, 8.0 * n / c_ * 1e-6 / 80); fflush(stdout); }
        ROW(0) ROW(1) ROW(2) ROW(3) ROW(4) ROW(5) ROW(6)
    }
    return 0;
}
