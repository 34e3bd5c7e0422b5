// tools/sweep_perm.hip -- is the f32 add's rate less placement-sensitive when workgroups walk the arrays in a
// permuted order?  For 8 re-allocations of a, b, c: natural order vs strided permutations vs per-XCD chunks.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));

// MODE 0: natural; 1: tile = (b * mul) mod nb (mul odd, nb power of two); 2: XCD-chunked: tile = (b % 8) * (nb / 8) + b / 8
template <int MODE>
__global__ __launch_bounds__(1024) void add_perm(const f4* __restrict__ a, const f4* __restrict__ b, f4* __restrict__ c, unsigned nb, unsigned mul) {
    unsigned t = blockIdx.x;
    if (MODE == 1) t = (t * mul) & (nb - 1);
    else if (MODE == 2) t = (t & 7u) * (nb >> 3) + (t >> 3);
    const size_t i = (size_t)t * 1024 + threadIdx.x;
    __builtin_nontemporal_store(__builtin_nontemporal_load(a + i) + __builtin_nontemporal_load(b + i), c + i);
}
__global__ void init_k(float* p, size_t n, float v) { for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v + (float)(i & 1023) * 1e-3f; }

int main() {
    const size_t n = (size_t)1 << 28, nvec = n / 4; const unsigned nb = (unsigned)(nvec / 1024);
    struct V { const char* name; int mode; unsigned mul; std::vector<double> gbs; };
    std::vector<V> vs = {{"natural", 0, 1, {}}, {"stride 257", 1, 257, {}}, {"stride 4099", 1, 4099, {}}, {"stride 16385", 1, 16385, {}},
                         {"stride 9 (XCD+1)", 1, 9, {}}, {"stride 2049", 1, 2049, {}}, {"xcd-chunked", 2, 1, {}}, {"stride 33", 1, 33, {}}};
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int trial = 0; trial < 8; ++trial) {
        void* junk1; void* junk2; CK(hipMalloc(&junk1, (size_t)(37 + 61 * trial) << 20)); 
        float *a, *b, *c; CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&junk2, (size_t)(5 + 3 * trial) << 20)); CK(hipMalloc(&b, n * 4)); CK(hipMalloc(&c, n * 4));
        init_k<<<4096, 256>>>(a, n, 1.f); init_k<<<4096, 256>>>(b, n, 2.f); CK(hipDeviceSynchronize());
        printf("trial %d:", trial);
        for (auto& v : vs) {
            auto launch = [&] {
                if (v.mode == 0) add_perm<0><<<nb, 1024>>>((const f4*)a, (const f4*)b, (f4*)c, nb, v.mul);
                else if (v.mode == 1) add_perm<1><<<nb, 1024>>>((const f4*)a, (const f4*)b, (f4*)c, nb, v.mul);
                else add_perm<2><<<nb, 1024>>>((const f4*)a, (const f4*)b, (f4*)c, nb, v.mul);
            };
            for (int i = 0; i < 3; ++i) launch();
            CK(hipEventRecord(e0)); for (int i = 0; i < 20; ++i) launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            const double g = 12.0 * n / (ms / 20) * 1e-6; v.gbs.push_back(g); printf(" %s=%.0f", v.name, g);
        }
        printf("\n");
        CK(hipFree(a)); CK(hipFree(b)); CK(hipFree(c)); CK(hipFree(junk1)); CK(hipFree(junk2));
    }
    for (auto& v : vs) { auto g = v.gbs; std::sort(g.begin(), g.end()); double sum = 0; for (double x : g) sum += x;
        printf("%-18s min %.0f  mean %.0f  max %.0f\n", v.name, g.front(), sum / g.size(), g.back()); }
    return 0;
}
