// tools/sweep_anyorder.hip -- what is the fixed cost per launch on an in-order stream, and how much of it is the order?
// (VERDICT r02 "next" #2: "dispatch ramp / drain"?)  Back-to-back launches on ONE stream, in order (every AQL packet carries
// the barrier bit: it starts after its predecessor has completed, caches written back and invalidated in between) against
// the same launches with hipExtAnyOrderLaunch (no barrier bit: independent launches may overlap).  Diagnosis only -- the
// library launches in order: its operators are ordered by the stream, and it cannot see what else a caller queued there.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void empty_k(int *p) { if (p && threadIdx.x == 12345) *p = 1; }
__global__ __launch_bounds__(256) void scal(const f4 *__restrict__ a, float s, f4 *__restrict__ o) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const f4 v = __builtin_nontemporal_load(a + i);
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(o + i), "v"(v * s));
}
__global__ void init_k(float *p, size_t n) { for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 1.0f + (float)(i % 1000) * 1e-3f; }
int main() {
    const size_t slab_bytes = (size_t)6 << 30;
    float *slab;
    CK(hipMalloc(&slab, slab_bytes));
    hipStream_t s;
    CK(hipStreamCreate(&s));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    init_k<<<4096, 256, 0, s>>>(slab, slab_bytes / 4);
    CK(hipStreamSynchronize(s));
    auto timed = [&](auto body, int reps) {
        int seq = 0;
        for (int i = 0; i < 24; ++i) body(seq++);
        std::vector<float> ms(5);
        for (auto &m : ms) {
            CK(hipEventRecord(e0, s));
            for (int i = 0; i < reps; ++i) body(seq++);
            CK(hipEventRecord(e1, s));
            CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&m, e0, e1));
            m /= reps;
        }
        std::sort(ms.begin(), ms.end());
        return ms[2] * 1e3f;
    };
    printf("%-44s %12s %12s\n", "launch", "in order us", "any order us");
    for (unsigned grid : {1u, 256u, 2048u, 8192u}) {
        const float a = timed([&](int) { hipLaunchKernelGGL(empty_k, dim3(grid), dim3(256), 0, s, (int *)nullptr); }, 100);
        const float b = timed([&](int) { hipExtLaunchKernelGGL(empty_k, dim3(grid), dim3(256), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, (int *)nullptr); }, 100);
        printf("empty kernel, %5u workgroups of 256 %8s %12.2f %12.2f\n", grid, "", a, b);
    }
    for (size_t mib : {16, 32, 64, 128}) {
        const size_t n = (mib << 20) / 4, n_vec = n / 4, set_floats = 2 * n;
        const int K = (int)std::min<size_t>(slab_bytes / 4 / set_floats, ((size_t)2560 << 20) / (set_floats * 4) + 1);
        auto body_in = [&](int i) { float *b = slab + (size_t)(i % K) * set_floats; hipLaunchKernelGGL(scal, dim3((unsigned)(n_vec / 256)), dim3(256), 0, s, (const f4 *)b, 1.0000001f, (f4 *)(b + n)); };
        auto body_any = [&](int i) { float *b = slab + (size_t)(i % K) * set_floats; hipExtLaunchKernelGGL(scal, dim3((unsigned)(n_vec / 256)), dim3(256), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, (const f4 *)b, 1.0000001f, (f4 *)(b + n)); };
        const float a = timed(body_in, 40), b = timed(body_any, 40);
        const double bytes = 8.0 * n;
        printf("a*s, %3zu MiB per array, rotating cold operands %12.2f %12.2f    %5.1f %% -> %5.1f %% of 8 TB/s\n", mib, a, b, bytes / a * 1e-6 / 80, bytes / b * 1e-6 / 80);
    }
    return 0;
}
