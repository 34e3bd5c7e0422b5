// tools/sweep_store_flavours.hip -- which store instruction leaves a result where the next launch's plain loads find it
// without slowing the launch that re-reads the same operands?  out = a * s, one 16-byte vector per lane, plain loads,
// stores written as inline asm with each cache-policy combination gfx950 encodes (sc0, sc1, nt).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
template <int ST> __device__ __forceinline__ void st(f4* p, f4 v) {
    if constexpr (ST == 0) asm volatile("global_store_dwordx4 %0, %1, off" :: "v"(p), "v"(v) : "memory");
    else if constexpr (ST == 1) asm volatile("global_store_dwordx4 %0, %1, off nt" :: "v"(p), "v"(v) : "memory");
    else if constexpr (ST == 2) asm volatile("global_store_dwordx4 %0, %1, off sc0" :: "v"(p), "v"(v) : "memory");
    else if constexpr (ST == 3) asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
    else if constexpr (ST == 4) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(p), "v"(v) : "memory");
    else if constexpr (ST == 5) asm volatile("global_store_dwordx4 %0, %1, off sc0 nt" :: "v"(p), "v"(v) : "memory");
    else if constexpr (ST == 6) asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" :: "v"(p), "v"(v) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" :: "v"(p), "v"(v) : "memory");
}
template <int ST> __global__ __launch_bounds__(256) void scal(const f4* __restrict__ a, float s, f4* __restrict__ o) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    st<ST>(o + i, a[i] * s);
}
__global__ void init_k(float* p, size_t n) { for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 1.0f + (float)(i % 1000) * 1e-3f; }
template <int ST> void launch(const float* a, float* o, size_t nvec) { scal<ST><<<(unsigned)(nvec / 256), 256>>>((const f4*)a, 1.0000001f, (f4*)o); }
int main() {
    const size_t slab_floats = (size_t)1 << 28;  // 1 GiB
    float* slab; CK(hipMalloc(&slab, slab_floats * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    init_k<<<4096, 256>>>(slab, slab_floats); CK(hipDeviceSynchronize());
    auto timed = [&](auto body, int reps) {
        int seq = 0;
        for (int i = 0; i < 24; ++i) body(seq++);
        std::vector<float> ms(5);
        for (auto& m : ms) { CK(hipEventRecord(e0)); for (int i = 0; i < reps; ++i) body(seq++); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&m, e0, e1)); m /= reps; }
        std::sort(ms.begin(), ms.end());
        return ms[2];
    };
    const char* names[8] = {"plain", "nt", "sc0", "sc1", "sc0 sc1", "sc0 nt", "sc1 nt", "sc0 sc1 nt"};
    printf("%-6s %-12s %12s %12s\n", "MiB", "store", "same %", "chain %");
    for (size_t mib : {32, 64, 128}) {
        const size_t n = mib << 18, nvec = n / 4;
        float *a = slab, *o = slab + n;
#define ROW(ST) { const float s_ = timed([&](int) { launch<ST>(a, o, nvec); }, 40); \
                  const float c_ = timed([&](int i) { (i & 1) ? launch<ST>(o, a, nvec) : launch<ST>(a, o, nvec); }, 40); \
                  printf("%-6zu %-12s %11.1f%% %11.1f%%\n", mib, names[ST], 8.0 * n / s_ * 1e-6 / 80, 8.0 * n / c_ * 1e-6 / 80); fflush(stdout); }
        ROW(0) ROW(1) ROW(2) ROW(3) ROW(4) ROW(5) ROW(6) ROW(7)
    }
    return 0;
}
