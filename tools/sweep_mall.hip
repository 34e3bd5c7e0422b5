// tools/sweep_mall.hip -- where does the plain-load advantage of read streams end?  out = a * s over n floats for n between
// 2^24 and 2^28 (read footprint 64 MiB .. 1 GiB): plain vs nt loads, nt stores, one vector per lane, 256-thread workgroups.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
template <int LD> __global__ __launch_bounds__(256) void scal(const f4* __restrict__ a, float s, f4* __restrict__ o) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    f4 v; if constexpr (LD) v = __builtin_nontemporal_load(a + i); else v = a[i];
    __builtin_nontemporal_store(v * s, o + i);
}
template <int LD> __global__ __launch_bounds__(1024) void add(const f4* __restrict__ a, const f4* __restrict__ b, f4* __restrict__ o) {
    const size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x;
    f4 x, y; if constexpr (LD) { x = __builtin_nontemporal_load(a + i); y = __builtin_nontemporal_load(b + i); } else { x = a[i]; y = b[i]; }
    __builtin_nontemporal_store(x + y, o + i);
}
__global__ void init_k(float* p, size_t n) { for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 1.0f + (float)(i % 1000) * 1e-3f; }
int main() {
    float* slab; CK(hipMalloc(&slab, (size_t)3 << 30));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    init_k<<<4096, 256>>>(slab, (size_t)3 << 28); CK(hipDeviceSynchronize());
    auto run = [&](auto launch) {
        for (int i = 0; i < 20; ++i) launch();
        std::vector<float> ms(7);
        for (auto& m : ms) { CK(hipEventRecord(e0)); for (int i = 0; i < 20; ++i) launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&m, e0, e1)); m /= 20; }
        std::sort(ms.begin(), ms.end());
        return ms[3];
    };
    for (size_t mib : {64, 128, 192, 256, 288, 320, 384, 448, 512, 768, 1024}) {
        const size_t n = mib << 18, nvec = n / 4;  // mib MiB of floats
        float *a = slab, *o = slab + n;
        const float p = run([&] { scal<0><<<(unsigned)(nvec / 256), 256>>>((const f4*)a, 2.5f, (f4*)o); });
        const float q = run([&] { scal<1><<<(unsigned)(nvec / 256), 256>>>((const f4*)a, 2.5f, (f4*)o); });
        printf("1R1W read %4zu MiB: plain %.1f%%  nt %.1f%%\n", mib, 8.0 * n / p * 1e-6 / 80, 8.0 * n / q * 1e-6 / 80);
    }
    for (size_t mib : {32, 64, 96, 128, 160, 192, 256, 384, 512}) {
        const size_t n = mib << 18, nvec = n / 4;  // per operand
        float *a = slab, *b = slab + n, *o = slab + 2 * n;
        const float p = run([&] { add<0><<<(unsigned)(nvec / 1024), 1024>>>((const f4*)a, (const f4*)b, (f4*)o); });
        const float q = run([&] { add<1><<<(unsigned)(nvec / 1024), 1024>>>((const f4*)a, (const f4*)b, (f4*)o); });
        printf("2R1W read 2 x %4zu MiB: plain %.1f%%  nt %.1f%%\n", mib, 12.0 * n / p * 1e-6 / 80, 12.0 * n / q * 1e-6 / 80);
    }
    return 0;
}
