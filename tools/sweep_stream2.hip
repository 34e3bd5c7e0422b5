// tools/sweep_stream2.hip -- second-level sweep of the contiguous f32 add (N = 2^28): cache-policy bits on the
// loads/stores (inline asm), lane width, block size; every variant timed in interleaved rounds in ONE process
// (cdna_hip_programming.md rule 24) and reported as median / min.  Development tool.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));

// POL: 0 plain, 1 nt, 2 sc1, 3 sc0 sc1, 4 sc0 sc1 nt, 5 sc0
template <int POL> __device__ __forceinline__ void ld_issue(f4& v, const f4* p) {
    if constexpr (POL == 0) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
    else if constexpr (POL == 1) asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(v) : "v"(p) : "memory");
    else if constexpr (POL == 2) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
    else if constexpr (POL == 3) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(v) : "v"(p) : "memory");
    else if constexpr (POL == 4) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1 nt" : "=v"(v) : "v"(p) : "memory");
    else asm volatile("global_load_dwordx4 %0, %1, off sc0" : "=v"(v) : "v"(p) : "memory");
}
template <int POL> __device__ __forceinline__ void st_issue(f4* p, f4 v) {
    if constexpr (POL == 0) asm volatile("global_store_dwordx4 %0, %1, off" :: "v"(p), "v"(v) : "memory");
    else if constexpr (POL == 1) asm volatile("global_store_dwordx4 %0, %1, off nt" :: "v"(p), "v"(v) : "memory");
    else if constexpr (POL == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
    else if constexpr (POL == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(p), "v"(v) : "memory");
    else if constexpr (POL == 4) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" :: "v"(p), "v"(v) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, off sc0" :: "v"(p), "v"(v) : "memory");
}

// one vector per lane, no loop
template <int LP, int SP, int BLOCK>
__global__ __launch_bounds__(BLOCK) void add1(const f4* __restrict__ a, const f4* __restrict__ b, f4* __restrict__ c, size_t nvec) {
    const size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= nvec) return;
    f4 va, vb;
    ld_issue<LP>(va, a + i);
    ld_issue<LP>(vb, b + i);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    st_issue<SP>(c + i, va + vb);
}
// two ADJACENT vectors per lane (32 contiguous bytes), no loop
template <int LP, int SP, int BLOCK>
__global__ __launch_bounds__(BLOCK) void add2adj(const f4* __restrict__ a, const f4* __restrict__ b, f4* __restrict__ c, size_t nvec) {
    const size_t i = ((size_t)blockIdx.x * BLOCK + threadIdx.x) * 2;
    if (i >= nvec) return;
    f4 a0, a1, b0, b1;
    ld_issue<LP>(a0, a + i); ld_issue<LP>(a1, a + i + 1); ld_issue<LP>(b0, b + i); ld_issue<LP>(b1, b + i + 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    st_issue<SP>(c + i, a0 + b0); st_issue<SP>(c + i + 1, a1 + b1);
}
// two vectors per lane, BLOCK apart (each instruction 1 KiB contiguous per wave)
template <int LP, int SP, int BLOCK>
__global__ __launch_bounds__(BLOCK) void add2str(const f4* __restrict__ a, const f4* __restrict__ b, f4* __restrict__ c, size_t nvec) {
    const size_t i = (size_t)blockIdx.x * BLOCK * 2 + threadIdx.x;
    if (i >= nvec) return;
    f4 a0, a1, b0, b1;
    ld_issue<LP>(a0, a + i); ld_issue<LP>(b0, b + i); ld_issue<LP>(a1, a + i + BLOCK); ld_issue<LP>(b1, b + i + BLOCK);
    asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    st_issue<SP>(c + i, a0 + b0);
    asm volatile("s_waitcnt vmcnt(1)" ::: "memory");   // the store counts too
    st_issue<SP>(c + i + BLOCK, a1 + b1);
}
// compiler-scheduled reference (what the library ships): builtin nt
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void add_builtin_nt(const f4* __restrict__ a, const f4* __restrict__ b, f4* __restrict__ c, size_t nvec) {
    const size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i < nvec) __builtin_nontemporal_store(__builtin_nontemporal_load(a + i) + __builtin_nontemporal_load(b + i), c + i);
}

__global__ void init_k(float* p, size_t n, float v) { for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v + (float)(i & 1023) * 1e-3f; }

struct V { std::string name; void (*fn)(const f4*, const f4*, f4*, size_t); int block; int per_thread; std::vector<float> ms; };

int main(int argc, char** argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 7;
    const size_t n = (size_t)1 << 28, nvec = n / 4;
    float *a, *b, *c; CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&b, n * 4)); CK(hipMalloc(&c, n * 4));
    init_k<<<4096, 256>>>(a, n, 1.f); init_k<<<4096, 256>>>(b, n, 2.f); init_k<<<4096, 256>>>(c, n, 0.f); CK(hipDeviceSynchronize());
    std::vector<V> vs;
#define A1(LP, SP, B) vs.push_back({"add1 ld" #LP " st" #SP " b" #B, add1<LP, SP, B>, B, 1, {}})
#define A2A(LP, SP, B) vs.push_back({"add2adj ld" #LP " st" #SP " b" #B, add2adj<LP, SP, B>, B, 2, {}})
#define A2S(LP, SP, B) vs.push_back({"add2str ld" #LP " st" #SP " b" #B, add2str<LP, SP, B>, B, 2, {}})
    vs.push_back({"builtin_nt b1024", add_builtin_nt<1024>, 1024, 1, {}});
    vs.push_back({"builtin_nt b512", add_builtin_nt<512>, 512, 1, {}});
    vs.push_back({"builtin_nt b256", add_builtin_nt<256>, 256, 1, {}});
    A1(0, 0, 1024); A1(1, 1, 1024); A1(1, 0, 1024); A1(0, 1, 1024); A1(2, 2, 1024); A1(3, 3, 1024); A1(4, 4, 1024); A1(5, 5, 1024);
    A1(1, 2, 1024); A1(1, 4, 1024); A1(4, 1, 1024); A1(2, 1, 1024); A1(1, 3, 1024); A1(5, 1, 1024);
    A1(1, 1, 256); A1(1, 1, 512); A1(4, 4, 256);
    A2A(1, 1, 1024); A2A(1, 1, 256); A2A(1, 1, 512); A2S(1, 1, 1024); A2S(1, 1, 256); A2S(1, 1, 512);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int r = 0; r < rounds; ++r)
        for (auto& v : vs) {
            const size_t threads = (nvec + v.per_thread - 1) / v.per_thread;
            const unsigned grid = (unsigned)((threads + v.block - 1) / v.block);
            v.fn<<<grid, v.block>>>((const f4*)a, (const f4*)b, (f4*)c, nvec);  // warm
            CK(hipEventRecord(e0));
            for (int k = 0; k < 5; ++k) v.fn<<<grid, v.block>>>((const f4*)a, (const f4*)b, (f4*)c, nvec);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); v.ms.push_back(ms / 5);
        }
    // correctness spot check of the last variant's output
    std::vector<float> h(1024), ha(1024), hb(1024);
    CK(hipMemcpy(h.data(), c + 12345 * 4, 4096, hipMemcpyDeviceToHost)); CK(hipMemcpy(ha.data(), a + 12345 * 4, 4096, hipMemcpyDeviceToHost)); CK(hipMemcpy(hb.data(), b + 12345 * 4, 4096, hipMemcpyDeviceToHost));
    for (int i = 0; i < 1024; ++i) if (h[i] != ha[i] + hb[i]) { printf("MISMATCH at %d\n", i); break; }
    std::sort(vs.begin(), vs.end(), [](const V& x, const V& y) { auto mx = x.ms, my = y.ms; std::sort(mx.begin(), mx.end()); std::sort(my.begin(), my.end()); return mx[mx.size() / 2] < my[my.size() / 2]; });
    for (auto& v : vs) {
        auto m = v.ms; std::sort(m.begin(), m.end());
        const double med = m[m.size() / 2], mn = m[0];
        printf("%-28s median %.4f ms %7.1f GB/s (%.1f%%)   best %.4f ms %7.1f GB/s\n", v.name.c_str(), med, 12.0 * n / med * 1e-6, 12.0 * n / med * 1e-6 / 80.0, mn, 12.0 * n / mn * 1e-6);
    }
    return 0;
}
