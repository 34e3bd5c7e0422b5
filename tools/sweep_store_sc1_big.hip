// tools/sweep_store_sc1_big.hip -- out = a + b with nt loads (reads above the Infinity Cache) and an output that would fit
// it: nt against sc1 stores when every launch reuses the same three arrays (`same`), and when launches walk K different
// triples (`rotate`: no array is touched again before >= 2 GiB of other traffic).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
template <int ST> __global__ __launch_bounds__(1024) void add(const f4* __restrict__ a, const f4* __restrict__ b, f4* __restrict__ o) {
    const size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x;
    const f4 x = __builtin_nontemporal_load(a + i), y = __builtin_nontemporal_load(b + i);
    if constexpr (ST == 0) __builtin_nontemporal_store(x + y, o + i);
    else asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(o + i), "v"(x + y));
}
__global__ void init_k(float* p, size_t n) { for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 1.0f + (float)(i % 1000) * 1e-3f; }
int main() {
    const size_t slab_floats = (size_t)15 << 26;  // 3.75 GiB
    float* slab; CK(hipMalloc(&slab, slab_floats * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    init_k<<<4096, 256>>>(slab, slab_floats); CK(hipDeviceSynchronize());
    auto timed = [&](auto body, int reps) {
        int seq = 0;
        for (int i = 0; i < 12; ++i) body(seq++);
        std::vector<float> ms(5);
        for (auto& m : ms) { CK(hipEventRecord(e0)); for (int i = 0; i < reps; ++i) body(seq++); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&m, e0, e1)); m /= reps; }
        std::sort(ms.begin(), ms.end());
        return ms[2];
    };
    printf("%-6s %-6s %10s %10s   (K triples in the rotation)\n", "MiB", "store", "same %", "rotate %");
    for (size_t mib : {136, 160, 192, 224, 256, 288, 320}) {
        const size_t n = mib << 18, nvec = n / 4;
        const int K = (int)(slab_floats / (3 * n));
        auto A = [&](int k) { return slab + (size_t)(3 * k) * n; };
        for (int st_ = 0; st_ < 2; ++st_) {
            auto go = [&](int k) { const unsigned g = (unsigned)(nvec / 1024); if (st_ == 0) add<0><<<g, 1024>>>((const f4*)A(k), (const f4*)(A(k) + n), (f4*)(A(k) + 2 * n)); else add<1><<<g, 1024>>>((const f4*)A(k), (const f4*)(A(k) + n), (f4*)(A(k) + 2 * n)); };
            const float s_ = timed([&](int) { go(0); }, 24);
            const float r_ = timed([&](int i) { go(i % K); }, 24);
            printf("%-6zu %-6s %9.1f%% %9.1f%%   K=%d\n", mib, st_ ? "sc1" : "nt", 12.0 * n / s_ * 1e-6 / 80, 12.0 * n / r_ * 1e-6 / 80, K); fflush(stdout);
        }
    }
    return 0;
}
