// tools/sweep_pow2.hip -- round-2 launch-shape sweep for pow(a, 2.5) (BASELINE config 4, N = 2^26, 8 B/elem) around the
// cheaper core of sm_pow.h: one-shot shapes, and persistent shapes that REALLY prefetch (two / three register sets
// filled alternately, so the compiler emits vmcnt(1)/vmcnt(2) instead of the vmcnt(0) of the copy-at-the-end form).
// hipcc -O3 -ffp-contract=off --offload-arch=gfx950 -Isimplemath_amd/csrc -o tools/bin/sweep_pow2 tools/sweep_pow2.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "sm_pow.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ const double* stage_table() {
    __shared__ __attribute__((aligned(16))) double lds_tab[2 * smpow::kTabN];
    for (int i = threadIdx.x; i < 2 * smpow::kTabN; i += blockDim.x) lds_tab[i] = smpow::kLogTab[i];
    __syncthreads();
    return lds_tab;
}
__device__ __forceinline__ f4 pow4(const double* tab, f4 v, float s) {
    float x[4] = {v[0], v[1], v[2], v[3]}, y[4] = {s, s, s, s}, r[4];
    smpow::pow_n<4>(x, y, r, tab);
    return f4{r[0], r[1], r[2], r[3]};
}
template <int U, int BLOCK>
__global__ __launch_bounds__(BLOCK) void oneshot(const f4* __restrict__ a, float s, f4* __restrict__ o, size_t nvec) {
    const double* tab = stage_table();
    const size_t base = (size_t)blockIdx.x * BLOCK * U + threadIdx.x;
    f4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load(a + base + (size_t)u * BLOCK);
#pragma unroll
    for (int u = 0; u < U; ++u) __builtin_nontemporal_store(pow4(tab, v[u], s), o + base + (size_t)u * BLOCK);
}
// two register sets
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void pingpong(const f4* __restrict__ a, float s, f4* __restrict__ o, size_t nvec) {
    const double* tab = stage_table();
    const size_t stride = (size_t)gridDim.x * BLOCK;
    size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= nvec) return;
    f4 a0 = __builtin_nontemporal_load(a + i), a1;
    for (;;) {
        const size_t j = i + stride;
        if (j >= nvec) { __builtin_nontemporal_store(pow4(tab, a0, s), o + i); break; }
        a1 = __builtin_nontemporal_load(a + j);
        __builtin_nontemporal_store(pow4(tab, a0, s), o + i);
        i = j + stride;
        if (i >= nvec) { __builtin_nontemporal_store(pow4(tab, a1, s), o + j); break; }
        a0 = __builtin_nontemporal_load(a + i);
        __builtin_nontemporal_store(pow4(tab, a1, s), o + j);
    }
}
// three register sets: two loads ahead
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void pingpong3(const f4* __restrict__ a, float s, f4* __restrict__ o, size_t nvec) {
    const double* tab = stage_table();
    const size_t stride = (size_t)gridDim.x * BLOCK;
    size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    // nvec is a multiple of 3 * stride in this sweep's calls where it matters; general tails handled by guards
    f4 r0, r1, r2;
    if (i < nvec) r0 = __builtin_nontemporal_load(a + i);
    if (i + stride < nvec) r1 = __builtin_nontemporal_load(a + i + stride);
    for (; i < nvec; i += 3 * stride) {
        if (i + 2 * stride < nvec) r2 = __builtin_nontemporal_load(a + i + 2 * stride);
        __builtin_nontemporal_store(pow4(tab, r0, s), o + i);
        if (i + stride >= nvec) break;
        if (i + 3 * stride < nvec) r0 = __builtin_nontemporal_load(a + i + 3 * stride);
        __builtin_nontemporal_store(pow4(tab, r1, s), o + i + stride);
        if (i + 2 * stride >= nvec) break;
        if (i + 4 * stride < nvec) r1 = __builtin_nontemporal_load(a + i + 4 * stride);
        __builtin_nontemporal_store(pow4(tab, r2, s), o + i + 2 * stride);
    }
}
template <int U, int BLOCK>
__global__ __launch_bounds__(BLOCK) void copy_oneshot(const f4* __restrict__ a, f4* __restrict__ o) {
    const size_t base = (size_t)blockIdx.x * BLOCK * U + threadIdx.x;
    f4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load(a + base + (size_t)u * BLOCK);
#pragma unroll
    for (int u = 0; u < U; ++u) __builtin_nontemporal_store(v[u], o + base + (size_t)u * BLOCK);
}
__global__ void init_k(float* p, size_t n) { for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 0.01f + (float)((i * 2654435761u) & 0xffffff) * (99.99f / 16777216.0f); }

int main() {
    const size_t n = (size_t)1 << 26, nvec = n / 4;
    float *a, *o; CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&o, n * 4));
    init_k<<<4096, 256>>>(a, n); CK(hipDeviceSynchronize());
    const f4* av = (const f4*)a; f4* ov = (f4*)o;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto b2b = [&](const char* name, auto launch) {
        for (int i = 0; i < 20; ++i) launch();
        CK(hipDeviceSynchronize());
        float best = 1e9f, sum = 0;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0));
            for (int i = 0; i < 100; ++i) launch();
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            best = std::min(best, ms / 100); sum += ms / 100;
        }
        printf("%-36s best %.2f us  mean %.2f us  %6.1f GB/s  %.3f of 8 TB/s\n", name, best * 1e3, sum / 3 * 1e3, 8.0 * n / best * 1e-6, 8.0 * n / best * 1e-6 / 8000);
        fflush(stdout);
    };
    // clock ramp
    for (int i = 0; i < 300; ++i) pingpong<512><<<256 * 32, 512>>>(av, 2.5f, ov, nvec);
    CK(hipDeviceSynchronize());
    b2b("COPY oneshot U1 b1024", [&] { copy_oneshot<1, 1024><<<nvec / 1024, 1024>>>(av, ov); });
    b2b("COPY oneshot U1 b256", [&] { copy_oneshot<1, 256><<<nvec / 256, 256>>>(av, ov); });
#define PP(B, M) b2b("pingpong b" #B " x" #M, [&] { pingpong<B><<<256 * M, B>>>(av, 2.5f, ov, nvec); })
#define PP3(B, M) b2b("pingpong3 b" #B " x" #M, [&] { pingpong3<B><<<256 * M, B>>>(av, 2.5f, ov, nvec); })
#define OS(U, B) b2b("oneshot U" #U " b" #B, [&] { oneshot<U, B><<<nvec / (B * U), B>>>(av, 2.5f, ov, nvec); })
    PP(512, 32); PP(512, 16); PP(512, 8); PP(512, 4); PP(512, 64);
    PP(256, 64); PP(256, 32); PP(256, 16); PP(256, 8);
    PP(1024, 16); PP(1024, 8); PP(1024, 4); PP(1024, 2);
    PP(128, 64); PP(128, 32);
    PP3(512, 32); PP3(512, 16); PP3(512, 8); PP3(512, 4); PP3(256, 32); PP3(256, 16); PP3(256, 8); PP3(1024, 8); PP3(1024, 4);
    OS(1, 256); OS(1, 512); OS(1, 1024); OS(2, 256); OS(2, 512); OS(4, 256); OS(4, 128);
    PP(512, 32);
    return 0;
}
