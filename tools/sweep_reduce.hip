// tools/sweep_reduce.hip -- launch-shape sweep for the fused c = a + b, sum(c) kernel (config 5 shard, 2^28 f32).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ double wave_reduce(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}
template <int BLOCK> __device__ __forceinline__ double block_reduce(double v) {
    __shared__ double lds[BLOCK / 64];
    v = wave_reduce(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) lds[wave] = v;
    __syncthreads();
    if (wave == 0) { v = lane < BLOCK / 64 ? lds[lane] : 0.0; v = wave_reduce(v); }
    return v;
}
// ACC: 0 = fp64 per element (cvt + add per float), 1 = f32 pairwise inside the vector then fp64 (1 cvt + 1 add per vector)
template <int BLOCK, int VPT, int ACC>
__global__ __launch_bounds__(BLOCK) void fused(const f4* __restrict__ a, const f4* __restrict__ b, f4* __restrict__ c, double* __restrict__ partials) {
    const size_t t0 = (size_t)blockIdx.x * BLOCK * VPT + threadIdx.x;
    f4 va[VPT], vb[VPT];
#pragma unroll
    for (int u = 0; u < VPT; ++u) { va[u] = __builtin_nontemporal_load(a + t0 + (size_t)u * BLOCK); vb[u] = __builtin_nontemporal_load(b + t0 + (size_t)u * BLOCK); }
    double acc = 0.0;
#pragma unroll
    for (int u = 0; u < VPT; ++u) {
        const f4 r = va[u] + vb[u];
        __builtin_nontemporal_store(r, c + t0 + (size_t)u * BLOCK);
        if (ACC == 0) { acc += (double)r[0]; acc += (double)r[1]; acc += (double)r[2]; acc += (double)r[3]; }
        else acc += ((double)r[0] + (double)r[1]) + ((double)r[2] + (double)r[3]);
    }
    acc = block_reduce<BLOCK>(acc);
    if (threadIdx.x == 0) partials[blockIdx.x] = acc;
}
template <int BLOCK, int VPT>
__global__ __launch_bounds__(BLOCK) void plain_add(const f4* __restrict__ a, const f4* __restrict__ b, f4* __restrict__ c, double*) {
    const size_t t0 = (size_t)blockIdx.x * BLOCK * VPT + threadIdx.x;
#pragma unroll
    for (int u = 0; u < VPT; ++u) __builtin_nontemporal_store(__builtin_nontemporal_load(a + t0 + (size_t)u * BLOCK) + __builtin_nontemporal_load(b + t0 + (size_t)u * BLOCK), c + t0 + (size_t)u * BLOCK);
}
__global__ void init_k(float* p, size_t n, float v) { for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v + (float)(i & 1023) * 1e-3f; }

struct V { std::string name; void (*fn)(const f4*, const f4*, f4*, double*); int block, vpt; std::vector<float> ms; };
int main() {
    const size_t n = (size_t)1 << 28, nvec = n / 4;
    float *a, *b, *c; double* part; CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&b, n * 4)); CK(hipMalloc(&c, n * 4)); CK(hipMalloc(&part, 8 << 20));
    init_k<<<4096, 256>>>(a, n, 1.f); init_k<<<4096, 256>>>(b, n, 2.f); CK(hipDeviceSynchronize());
    std::vector<V> vs;
#define F(B, U, A) vs.push_back({"fused b" #B " vpt" #U " acc" #A, fused<B, U, A>, B, U, {}})
    F(256, 4, 0); F(256, 2, 0); F(256, 1, 0); F(512, 1, 0); F(1024, 1, 0); F(1024, 2, 0); F(512, 2, 0); F(256, 8, 0);
    F(256, 4, 1); F(256, 1, 1); F(1024, 1, 1); F(256, 2, 1);
    vs.push_back({"plain add b1024 vpt1", plain_add<1024, 1>, 1024, 1, {}});
    vs.push_back({"plain add b256 vpt4", plain_add<256, 4>, 256, 4, {}});
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int r = 0; r < 7; ++r)
        for (auto& v : vs) {
            const unsigned grid = (unsigned)(nvec / ((size_t)v.block * v.vpt));
            v.fn<<<grid, v.block>>>((const f4*)a, (const f4*)b, (f4*)c, part);
            CK(hipEventRecord(e0));
            for (int k = 0; k < 5; ++k) v.fn<<<grid, v.block>>>((const f4*)a, (const f4*)b, (f4*)c, part);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); v.ms.push_back(ms / 5);
        }
    std::sort(vs.begin(), vs.end(), [](const V& x, const V& y) { auto mx = x.ms, my = y.ms; std::sort(mx.begin(), mx.end()); std::sort(my.begin(), my.end()); return mx[3] < my[3]; });
    for (auto& v : vs) { auto m = v.ms; std::sort(m.begin(), m.end()); printf("%-26s median %.4f ms %7.1f GB/s (%.1f%%)  best %.4f\n", v.name.c_str(), m[3], 12.0 * n / m[3] * 1e-6, 12.0 * n / m[3] * 1e-6 / 80.0, m[0]); }
    return 0;
}
