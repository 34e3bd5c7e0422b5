// tools/sweep_scalar.hip -- out[i] = a[i] * s (1 read + 1 write stream, N = 2^28 and 2^26 f32): vectors per lane x block size.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
// U vectors per lane, block-strided inside the workgroup's span (each wave instruction still covers 1 KiB contiguous)
template <int U, int BLOCK>
__global__ __launch_bounds__(BLOCK) void scal(const f4* __restrict__ a, float s, f4* __restrict__ o, size_t nvec) {
    const size_t base = (size_t)blockIdx.x * BLOCK * U + threadIdx.x;
    f4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) if (base + (size_t)u * BLOCK < nvec) v[u] = __builtin_nontemporal_load(a + base + (size_t)u * BLOCK);
#pragma unroll
    for (int u = 0; u < U; ++u) if (base + (size_t)u * BLOCK < nvec) __builtin_nontemporal_store(v[u] * s, o + base + (size_t)u * BLOCK);
}
__global__ void init_k(float* p, size_t n) { for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 1.0f + (float)(i % 1000) * 1e-3f; }
int main() {
    float* slab; CK(hipMalloc(&slab, (size_t)2 << 30));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int lg : {28, 26, 24}) {
        const size_t n = (size_t)1 << lg, nvec = n / 4;
        float *a = slab, *o = slab + n;
        init_k<<<4096, 256>>>(a, n); CK(hipDeviceSynchronize());
        auto run = [&](const char* name, auto launch) {
            for (int i = 0; i < 20; ++i) launch();
            std::vector<float> ms(7);
            for (auto& m : ms) { CK(hipEventRecord(e0)); for (int i = 0; i < 20; ++i) launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&m, e0, e1)); m /= 20; }
            std::sort(ms.begin(), ms.end());
            printf("n=2^%d %-22s %.4f ms %7.1f GB/s (%.1f%%)\n", lg, name, ms[3], 8.0 * n / ms[3] * 1e-6, 8.0 * n / ms[3] * 1e-6 / 80.0);
        };
#define V(U, B) run("U" #U " block" #B, [&] { scal<U, B><<<(unsigned)((nvec + (size_t)U * B - 1) / ((size_t)U * B)), B>>>((const f4*)a, 2.5f, (f4*)o, nvec); })
        V(1, 256); V(1, 512); V(1, 1024); V(2, 256); V(2, 512); V(2, 1024); V(4, 256); V(4, 512); V(4, 1024); V(8, 256);
    }
    return 0;
}
