// tools/sweep_unaligned.hip -- does a 16-byte vector access that is only element-aligned (4 B) stream as fast as an
// aligned one?  out[i] = a[i] + b[i] over 2^28 floats with a, b, out shifted by 0..3 elements.  Development tool.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
typedef f4 f4u __attribute__((aligned(4)));
__global__ __launch_bounds__(1024) void addu(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ o, size_t nvec) {
    const size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x;
    if (i < nvec) {
        const f4 va = __builtin_nontemporal_load(reinterpret_cast<const f4u*>(a) + i), vb = __builtin_nontemporal_load(reinterpret_cast<const f4u*>(b) + i);
        __builtin_nontemporal_store(va + vb, reinterpret_cast<f4u*>(o) + i);
    }
}
__global__ __launch_bounds__(1024) void add1(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ o, size_t n) {
    const size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x;
    if (i < n) __builtin_nontemporal_store(__builtin_nontemporal_load(a + i) + __builtin_nontemporal_load(b + i), o + i);
}
__global__ void init_k(float* p, size_t n, float v) { for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v + (float)(i % 1000) * 1e-3f; }
int main() {
    const size_t n = 1ull << 28, nvec = n / 4;
    float* slab; CK(hipMalloc(&slab, 3 * (n + 1024) * 4));
    float *A = slab, *B = slab + n + 1024, *O = slab + 2 * (n + 1024);
    init_k<<<4096, 256>>>(A, n + 8, 1.f); init_k<<<4096, 256>>>(B, n + 8, 2.f); CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 200; ++w) addu<<<nvec / 1024, 1024>>>(A, B, O, nvec);
    const int offs[][3] = {{0, 0, 0}, {1, 1, 1}, {1, 0, 0}, {0, 0, 1}, {1, 2, 3}, {2, 2, 0}, {3, 1, 0}};
    for (auto& of : offs) {
        CK(hipEventRecord(e0));
        for (int k = 0; k < 20; ++k) addu<<<nvec / 1024, 1024>>>(A + of[0], B + of[1], O + of[2], nvec);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 20;
        std::vector<float> h(4), ha(4), hb(4);
        const size_t at = n - 4;
        CK(hipMemcpy(h.data(), O + of[2] + at, 16, hipMemcpyDeviceToHost)); CK(hipMemcpy(ha.data(), A + of[0] + at, 16, hipMemcpyDeviceToHost)); CK(hipMemcpy(hb.data(), B + of[1] + at, 16, hipMemcpyDeviceToHost));
        bool ok = true; for (int k = 0; k < 4; ++k) ok &= h[k] == ha[k] + hb[k];
        printf("vector offsets a+%d b+%d out+%d   %.4f ms %7.1f GB/s  %s\n", of[0], of[1], of[2], ms, 12.0 * n / ms * 1e-6, ok ? "ok" : "MISMATCH");
    }
    CK(hipEventRecord(e0));
    for (int k = 0; k < 20; ++k) add1<<<n / 1024, 1024>>>(A + 1, B + 2, O + 3, n);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 20;
    printf("one element per lane (a+1 b+2 out+3) %.4f ms %7.1f GB/s\n", ms, 12.0 * n / ms * 1e-6);
    return 0;
}
