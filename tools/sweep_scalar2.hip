// tools/sweep_scalar2.hip -- 1 read + 1 write streams (out[i] = a[i] * s; N = 2^28, 2^26, 2^24 f32): does the READ want the
// non-temporal hint?  Round 1 swept block sizes with nt on both sides (81.7 %); the row kernel, whose streamed loads had
// silently lost their nt hint, ran the same traffic at 90 %.  LD: 0 plain, 1 nt.  ST: 0 plain, 1 nt.  ROWS2: two vectors
// per lane one row apart (the row kernel's shape) instead of adjacent tiles.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
template <int LD> __device__ __forceinline__ f4 ld(const f4* p) { if constexpr (LD) return __builtin_nontemporal_load(p); else return *p; }
template <int ST> __device__ __forceinline__ void st(f4 v, f4* p) { if constexpr (ST) __builtin_nontemporal_store(v, p); else *p = v; }
template <int U, int BLOCK, int LD, int ST>
__global__ __launch_bounds__(BLOCK) void scal(const f4* __restrict__ a, float s, f4* __restrict__ o, size_t nvec) {
    const size_t base = (size_t)blockIdx.x * BLOCK * U + threadIdx.x;
    f4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = ld<LD>(a + base + (size_t)u * BLOCK);
#pragma unroll
    for (int u = 0; u < U; ++u) st<ST>(v[u] * s, o + base + (size_t)u * BLOCK);
}
// 2R + 1W for comparison (the headline add)
template <int BLOCK, int LD, int ST>
__global__ __launch_bounds__(BLOCK) void add(const f4* __restrict__ a, const f4* __restrict__ b, f4* __restrict__ o) {
    const size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    st<ST>(ld<LD>(a + i) + ld<LD>(b + i), o + i);
}
// 1R reduce-free read-only + 1W write-only calibrations
template <int BLOCK, int ST>
__global__ __launch_bounds__(BLOCK) void fill(f4* __restrict__ o, float s) { st<ST>(f4{s, s, s, s}, o + (size_t)blockIdx.x * BLOCK + threadIdx.x); }
__global__ void init_k(float* p, size_t n) { for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 1.0f + (float)(i % 1000) * 1e-3f; }
int main() {
    float* slab; CK(hipMalloc(&slab, (size_t)3 << 30));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int lg : {28, 26, 24}) {
        const size_t n = (size_t)1 << lg, nvec = n / 4;
        float *a = slab, *o = slab + n, *b = slab + 2 * n;
        init_k<<<4096, 256>>>(a, n); init_k<<<4096, 256>>>(b, n); CK(hipDeviceSynchronize());
        auto run = [&](const char* name, double bytes_per_elem, auto launch) {
            for (int i = 0; i < 20; ++i) launch();
            std::vector<float> ms(7);
            for (auto& m : ms) { CK(hipEventRecord(e0)); for (int i = 0; i < 20; ++i) launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&m, e0, e1)); m /= 20; }
            std::sort(ms.begin(), ms.end());
            printf("n=2^%d %-34s %.4f ms %7.1f GB/s (%.1f%%)\n", lg, name, ms[3], bytes_per_elem * n / ms[3] * 1e-6, bytes_per_elem * n / ms[3] * 1e-6 / 80.0);
            fflush(stdout);
        };
#define V(U, B, LD, ST) run("1R1W U" #U " b" #B " ld" #LD " st" #ST, 8.0, [&] { scal<U, B, LD, ST><<<(unsigned)(nvec / ((size_t)U * B)), B>>>((const f4*)a, 2.5f, (f4*)o, nvec); })
        V(1, 256, 1, 1); V(1, 256, 0, 1); V(1, 256, 0, 0); V(1, 256, 1, 0);
        V(1, 1024, 1, 1); V(1, 1024, 0, 1);
        V(2, 256, 1, 1); V(2, 256, 0, 1); V(2, 512, 0, 1); V(4, 256, 0, 1); V(1, 512, 0, 1); V(1, 128, 0, 1);
#define A(B, LD, ST) run("2R1W b" #B " ld" #LD " st" #ST, 12.0, [&] { add<B, LD, ST><<<(unsigned)(nvec / B), B>>>((const f4*)a, (const f4*)b, (f4*)o); })
        A(1024, 1, 1); A(1024, 0, 1); A(256, 1, 1); A(256, 0, 1);
        run("1W fill b256 st1", 4.0, [&] { fill<256, 1><<<(unsigned)(nvec / 256), 256>>>((f4*)o, 2.5f); });
        run("1W fill b256 st0", 4.0, [&] { fill<256, 0><<<(unsigned)(nvec / 256), 256>>>((f4*)o, 2.5f); });
    }
    return 0;
}
