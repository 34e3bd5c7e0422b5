// tools/sweep_stream.hip -- launch-shape sweep for the contiguous f32 add
// (BASELINE config 2: N = 2^28, 12 B/elem).  Development tool, not shipped:
// its winners are baked into simplemath_amd/csrc/contiguous.hip.
//
//   hipcc -O3 --offload-arch=gfx950 -o tools/sweep_stream tools/sweep_stream.hip
//   gpurun -- ./tools/sweep_stream [log2N] > gpurun_out/sweep.txt
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));

template <bool NT> __device__ __forceinline__ f4 ld(const f4* p) {
    if constexpr (NT) return __builtin_nontemporal_load(p);
    else return *p;
}
template <bool NT> __device__ __forceinline__ void st(f4* p, f4 v) {
    if constexpr (NT) __builtin_nontemporal_store(v, p);
    else *p = v;
}

// MODE 0: interleaved grid-stride tiles (tile = blockDim*U vectors)
// MODE 1: each block owns one contiguous span of nvec/gridDim vectors
template <int U, bool NTL, bool NTS, int MODE>
__global__ void add_k(const f4* __restrict__ a, const f4* __restrict__ b, f4* __restrict__ c, size_t nvec) {
    const size_t tile = (size_t)blockDim.x * U;
    size_t begin, end, step;
    if constexpr (MODE == 0) { begin = (size_t)blockIdx.x * tile; end = nvec; step = (size_t)gridDim.x * tile; }
    else {
        size_t per = (nvec + gridDim.x - 1) / gridDim.x;
        per = (per + tile - 1) / tile * tile;
        begin = (size_t)blockIdx.x * per; end = begin + per < nvec ? begin + per : nvec; step = tile;
    }
    for (size_t base = begin; base < end; base += step) {
        f4 va[U], vb[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            size_t i = base + (size_t)u * blockDim.x + threadIdx.x;
            if (i < end) { va[u] = ld<NTL>(a + i); vb[u] = ld<NTL>(b + i); }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            size_t i = base + (size_t)u * blockDim.x + threadIdx.x;
            if (i < end) st<NTS>(c + i, va[u] + vb[u]);
        }
    }
}

template <int U, bool NTL, bool NTS>
__global__ void copy_k(const f4* __restrict__ a, f4* __restrict__ c, size_t nvec) {
    const size_t tile = (size_t)blockDim.x * U;
    for (size_t base = (size_t)blockIdx.x * tile; base < nvec; base += (size_t)gridDim.x * tile) {
        f4 va[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { size_t i = base + (size_t)u * blockDim.x + threadIdx.x; if (i < nvec) va[u] = ld<NTL>(a + i); }
#pragma unroll
        for (int u = 0; u < U; ++u) { size_t i = base + (size_t)u * blockDim.x + threadIdx.x; if (i < nvec) st<NTS>(c + i, va[u]); }
    }
}

__global__ void init_k(float* p, size_t n, float v) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v + (float)(i & 1023) * 1e-3f;
}

struct Variant { const char* name; void (*fn)(const f4*, const f4*, f4*, size_t); int U; };

static double time_launch(void (*fn)(const f4*, const f4*, f4*, size_t), int grid, int block, const f4* a, const f4* b, f4* c, size_t nvec, int iters) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(fn, dim3(grid), dim3(block), 0, 0, a, b, c, nvec);
    CK(hipDeviceSynchronize());
    std::vector<float> ms(iters);
    for (int i = 0; i < iters; ++i) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(fn, dim3(grid), dim3(block), 0, 0, a, b, c, nvec);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms[i], e0, e1));
    }
    std::sort(ms.begin(), ms.end());
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    return ms[iters / 2];
}

int main(int argc, char** argv) {
    int lg = argc > 1 ? atoi(argv[1]) : 28;
    size_t n = (size_t)1 << lg, nvec = n / 4;
    float *a, *b, *c;
    CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&b, n * 4)); CK(hipMalloc(&c, n * 4));
    init_k<<<4096, 256>>>(a, n, 1.0f); init_k<<<4096, 256>>>(b, n, 2.0f); init_k<<<4096, 256>>>(c, n, 0.0f);
    CK(hipDeviceSynchronize());
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    printf("# device %s CUs=%d clock=%d MHz memclk=%d MHz bus=%d\n", prop.name, prop.multiProcessorCount, prop.clockRate / 1000, prop.memoryClockRate / 1000, prop.memoryBusWidth);
    const int CUS = prop.multiProcessorCount;
    const double bytes = 12.0 * n;

#define V(U, NTL, NTS, MODE) { "U" #U "_ntl" #NTL "_nts" #NTS "_m" #MODE, add_k<U, NTL, NTS, MODE>, U }
    Variant vs[] = {
        V(1, false, false, 0), V(2, false, false, 0), V(4, false, false, 0), V(8, false, false, 0),
        V(1, true, true, 0), V(2, true, true, 0), V(4, true, true, 0), V(8, true, true, 0),
        V(4, false, true, 0), V(4, true, false, 0), V(2, false, true, 0), V(8, false, true, 0),
        V(2, false, false, 1), V(4, false, false, 1), V(4, true, true, 1), V(4, false, true, 1), V(8, false, true, 1),
    };
    int blocks[] = {256, 512, 1024};
    int mults[] = {1, 2, 4, 8, 16, 32, 0};  // 0 = one tile per block (no loop)
    double best = 0; char bestname[128] = "";
    for (auto& v : vs) for (int blk : blocks) for (int m : mults) {
        size_t tile = (size_t)blk * v.U;
        size_t full = (nvec + tile - 1) / tile;
        size_t grid = m == 0 ? full : std::min<size_t>(full, (size_t)CUS * m);
        if (grid > 0x7fffffff) continue;
        double ms = time_launch(v.fn, (int)grid, blk, (const f4*)a, (const f4*)b, (f4*)c, nvec, 15);
        double gbs = bytes / ms * 1e-6;
        printf("add %-22s block=%4d grid=%9zu (x%2d)  %.4f ms  %8.1f GB/s  %.1f%%\n", v.name, blk, grid, m, ms, gbs, gbs / 80.0);
        if (gbs > best) { best = gbs; snprintf(bestname, sizeof bestname, "%s block=%d mult=%d", v.name, blk, m); }
        fflush(stdout);
    }
    printf("# BEST add: %s %.1f GB/s (%.1f%% of 8 TB/s)\n", bestname, best, best / 80.0);
    // copy baselines (1R + 1W = 8 B/elem)
    {
        struct CV { const char* name; void (*fn)(const f4*, f4*, size_t); int U; };
        CV cvs[] = { {"copy_U4", copy_k<4, false, false>, 4}, {"copy_U4_nt", copy_k<4, true, true>, 4}, {"copy_U8_nts", copy_k<8, false, true>, 8}, {"copy_U2", copy_k<2, false, false>, 2} };
        for (auto& v : cvs) for (int m : {4, 8, 16, 0}) {
            int blk = 256; size_t tile = (size_t)blk * v.U; size_t full = (nvec + tile - 1) / tile;
            size_t grid = m == 0 ? full : std::min<size_t>(full, (size_t)CUS * m);
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(v.fn, dim3(grid), dim3(blk), 0, 0, (const f4*)a, (f4*)c, nvec);
            std::vector<float> ms(15);
            for (int i = 0; i < 15; ++i) { CK(hipEventRecord(e0, 0)); hipLaunchKernelGGL(v.fn, dim3(grid), dim3(blk), 0, 0, (const f4*)a, (f4*)c, nvec); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms[i], e0, e1)); }
            std::sort(ms.begin(), ms.end());
            printf("copy %-12s grid=%9zu (x%2d) %.4f ms %8.1f GB/s\n", v.name, grid, m, ms[7], 8.0 * n / ms[7] * 1e-6);
        }
    }
    // hipMemcpy D2D for comparison
    {
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipMemcpy(c, a, n * 4, hipMemcpyDeviceToDevice));
        CK(hipEventRecord(e0, 0)); for (int i = 0; i < 5; ++i) CK(hipMemcpyAsync(c, a, n * 4, hipMemcpyDeviceToDevice, 0)); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("hipMemcpy D2D %.4f ms %8.1f GB/s\n", ms / 5, 8.0 * n / (ms / 5) * 1e-6);
    }
    return 0;
}
