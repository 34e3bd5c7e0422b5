// tools/sweep_store_sc1.hip -- `sc1` stores against `nt` stores across sizes: out = a * s (1R+1W) from 8 MiB to 1 GiB per
// array with the library's read rule (plain loads up to 256 MiB, nt above), in the `same` and `chain` settings of
// sweep_chain.hip, and out = a + b (2R+1W) at 16 MiB .. 1 GiB per array.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
template <int ST> __device__ __forceinline__ void st(f4* p, f4 v) {
    if constexpr (ST == 0) __builtin_nontemporal_store(v, p);
    else if constexpr (ST == 1) asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(v));
    else *p = v;
}
template <int LD, int ST> __global__ __launch_bounds__(256) void scal(const f4* __restrict__ a, float s, f4* __restrict__ o) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    f4 v; if constexpr (LD) v = __builtin_nontemporal_load(a + i); else v = a[i];
    st<ST>(o + i, v * s);
}
template <int LD, int ST> __global__ __launch_bounds__(1024) void add(const f4* __restrict__ a, const f4* __restrict__ b, f4* __restrict__ o) {
    const size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x;
    f4 x, y; if constexpr (LD) { x = __builtin_nontemporal_load(a + i); y = __builtin_nontemporal_load(b + i); } else { x = a[i]; y = b[i]; }
    st<ST>(o + i, x + y);
}
__global__ void init_k(float* p, size_t n) { for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 1.0f + (float)(i % 1000) * 1e-3f; }
int main() {
    const size_t slab_floats = (size_t)3 << 28;  // 3 GiB
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
    printf("1R+1W   %-6s %-6s %10s %10s\n", "MiB", "store", "same %", "chain %");
    for (size_t mib : {8, 16, 32, 64, 128, 192, 256, 384, 512, 1024}) {
        const size_t n = mib << 18, nvec = n / 4;
        float *a = slab, *o = slab + n;
        const bool ldnt = mib > 256;
        auto go = [&](int st_, const float* x, float* y) {
            const unsigned g = (unsigned)(nvec / 256);
            if (ldnt) { if (st_ == 0) scal<1, 0><<<g, 256>>>((const f4*)x, 1.0000001f, (f4*)y); else if (st_ == 1) scal<1, 1><<<g, 256>>>((const f4*)x, 1.0000001f, (f4*)y); else scal<1, 2><<<g, 256>>>((const f4*)x, 1.0000001f, (f4*)y); }
            else { if (st_ == 0) scal<0, 0><<<g, 256>>>((const f4*)x, 1.0000001f, (f4*)y); else if (st_ == 1) scal<0, 1><<<g, 256>>>((const f4*)x, 1.0000001f, (f4*)y); else scal<0, 2><<<g, 256>>>((const f4*)x, 1.0000001f, (f4*)y); }
        };
        const char* names[3] = {"nt", "sc1", "plain"};
        for (int st_ = 0; st_ < 3; ++st_) {
            const float s_ = timed([&](int) { go(st_, a, o); }, 30);
            const float c_ = timed([&](int i) { (i & 1) ? go(st_, o, a) : go(st_, a, o); }, 30);
            printf("        %-6zu %-6s %9.1f%% %9.1f%%\n", mib, names[st_], 8.0 * n / s_ * 1e-6 / 80, 8.0 * n / c_ * 1e-6 / 80); fflush(stdout);
        }
    }
    printf("2R+1W   %-6s %-6s %10s %10s\n", "MiB", "store", "same %", "chain %");
    for (size_t mib : {16, 32, 64, 128, 256, 512, 1024}) {
        const size_t n = mib << 18, nvec = n / 4;
        float *a = slab, *b = slab + n, *o = slab + 2 * n;
        const bool ldnt = 2 * mib > 256;
        auto go = [&](int st_, const float* x, const float* y, float* z) {
            const unsigned g = (unsigned)(nvec / 1024);
            if (ldnt) { if (st_ == 0) add<1, 0><<<g, 1024>>>((const f4*)x, (const f4*)y, (f4*)z); else if (st_ == 1) add<1, 1><<<g, 1024>>>((const f4*)x, (const f4*)y, (f4*)z); else add<1, 2><<<g, 1024>>>((const f4*)x, (const f4*)y, (f4*)z); }
            else { if (st_ == 0) add<0, 0><<<g, 1024>>>((const f4*)x, (const f4*)y, (f4*)z); else if (st_ == 1) add<0, 1><<<g, 1024>>>((const f4*)x, (const f4*)y, (f4*)z); else add<0, 2><<<g, 1024>>>((const f4*)x, (const f4*)y, (f4*)z); }
        };
        const char* names[3] = {"nt", "sc1", "plain"};
        for (int st_ = 0; st_ < 3; ++st_) {
            const float s_ = timed([&](int) { go(st_, a, b, o); }, 30);
            const float c_ = timed([&](int i) { (i & 1) ? go(st_, o, b, a) : go(st_, a, b, o); }, 30);
            printf("        %-6zu %-6s %9.1f%% %9.1f%%\n", mib, names[st_], 12.0 * n / s_ * 1e-6 / 80, 12.0 * n / c_ * 1e-6 / 80); fflush(stdout);
        }
    }
    return 0;
}
