// tools/sweep_transpose.hip -- tile-shape sweep for out = A.T + B (A, B, out: N x N f32, N = 8192), the
// transposed-view case of broadcast.hip's tile kernel.  Development tool.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));

// out[i][j] = A[j][i] + B[i][j];  i along p (A's contiguous axis), j along q (output inner axis)
// ORDER 0: blockIdx -> tq fastest; 1: tp fastest.  NT: non-temporal loads of A.
template <int TP, int TQ, int ORDER, bool NT>
__global__ __launch_bounds__(256) void tadd(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ out, int N) {
    __shared__ float lds[TP][TQ + 1];  // [i][j]
    const int tiles_p = N / TP, tiles_q = N / TQ;
    int tp, tq;
    if (ORDER == 0) { tq = blockIdx.x % tiles_q; tp = blockIdx.x / tiles_q; } else if (ORDER == 1) { tp = blockIdx.x % tiles_p; tq = blockIdx.x / tiles_p; }
    else {  // ORDER >= 2: panels of ORDER q-patches; inside a panel q fastest, then p 
        const int panel = blockIdx.x / (ORDER * tiles_p), r = blockIdx.x % (ORDER * tiles_p);
        const int w = tiles_q - panel * ORDER < ORDER ? tiles_q - panel * ORDER : ORDER;  // the last panel may be narrower
        tq = panel * ORDER + r % w; tp = r / w;
    }
    const int i0 = tp * TP, j0 = tq * TQ;
    constexpr int VP = TP / 4, VQ = TQ / 4;
#pragma unroll
    for (int s = 0; s < TQ * VP / 256; ++s) {
        const int v = threadIdx.x + 256 * s, jl = v / VP, ig = v % VP;
        const f4* src = reinterpret_cast<const f4*>(A + (size_t)(j0 + jl) * N + i0 + ig * 4);
        const f4 val = NT ? __builtin_nontemporal_load(src) : *src;
#pragma unroll
        for (int k = 0; k < 4; ++k) lds[ig * 4 + k][jl] = val[k];
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < TP * VQ / 256; ++s) {
        const int v = threadIdx.x + 256 * s, il = v / VQ, jg = v % VQ;
        const size_t o = (size_t)(i0 + il) * N + j0 + jg * 4;
        const f4 vb = __builtin_nontemporal_load(reinterpret_cast<const f4*>(B + o));
        f4 r;
#pragma unroll
        for (int k = 0; k < 4; ++k) r[k] = lds[il][jg * 4 + k] + vb[k];
        __builtin_nontemporal_store(r, reinterpret_cast<f4*>(out + o));
    }
}
// both operands transposed: out[i][j] = A[j][i] + B[j][i]; the add happens before the turn, one LDS tile
template <int TP, int TQ, int ORDER, bool NT>
__global__ __launch_bounds__(256) void tadd2(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ out, int N) {
    __shared__ float lds[TP][TQ + 1];  // [i][j]
    const int tiles_p = N / TP, tiles_q = N / TQ;
    int tp, tq;
    if (ORDER == 0) { tq = blockIdx.x % tiles_q; tp = blockIdx.x / tiles_q; } else { tp = blockIdx.x % tiles_p; tq = blockIdx.x / tiles_p; }
    const int i0 = tp * TP, j0 = tq * TQ;
    constexpr int VP = TP / 4, VQ = TQ / 4;
#pragma unroll
    for (int s = 0; s < TQ * VP / 256; ++s) {
        const int v = threadIdx.x + 256 * s, jl = v / VP, ig = v % VP;
        const size_t o = (size_t)(j0 + jl) * N + i0 + ig * 4;
        const f4 va = NT ? __builtin_nontemporal_load(reinterpret_cast<const f4*>(A + o)) : *reinterpret_cast<const f4*>(A + o);
        const f4 vb = NT ? __builtin_nontemporal_load(reinterpret_cast<const f4*>(B + o)) : *reinterpret_cast<const f4*>(B + o);
#pragma unroll
        for (int k = 0; k < 4; ++k) lds[ig * 4 + k][jl] = va[k] + vb[k];
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < TP * VQ / 256; ++s) {
        const int v = threadIdx.x + 256 * s, il = v / VQ, jg = v % VQ;
        const size_t o = (size_t)(i0 + il) * N + j0 + jg * 4;
        f4 r;
#pragma unroll
        for (int k = 0; k < 4; ++k) r[k] = lds[il][jg * 4 + k];
        __builtin_nontemporal_store(r, reinterpret_cast<f4*>(out + o));
    }
}
__global__ void plain(const f4* __restrict__ a, const f4* __restrict__ b, f4* __restrict__ c, size_t nvec) {
    const size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x;
    if (i < nvec) __builtin_nontemporal_store(__builtin_nontemporal_load(a + i) + __builtin_nontemporal_load(b + i), c + i);
}
__global__ void init_k(float* p, size_t n, float v) { for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v + (float)(i % 1000) * 1e-3f; }

struct V { std::string name; void (*fn)(const float*, const float*, float*, int); int tp, tq; std::vector<float> ms; };
int main(int argc, char** argv) {
    const bool both = argc > 1 && std::string(argv[1]) == "both";
    const int N = argc > 2 ? atoi(argv[2]) : 8192; const size_t n = (size_t)N * N;   // N: a multiple of 256
    printf("N = %d\n", N);
    float *A, *B, *O; CK(hipMalloc(&A, n * 4)); CK(hipMalloc(&B, n * 4)); CK(hipMalloc(&O, n * 4));
    init_k<<<4096, 256>>>(A, n, 1.f); init_k<<<4096, 256>>>(B, n, 2.f); CK(hipDeviceSynchronize());
    std::vector<V> vs;
#define T(P, Q, O_, NT) vs.push_back({"tile " #P "x" #Q " order" #O_ " nt" #NT, tadd<P, Q, O_, NT>, P, Q, {}})
    T(64, 64, 0, false); T(64, 64, 1, false); T(64, 64, 0, true); T(64, 64, 1, true);
    T(128, 64, 0, false); T(128, 64, 1, false); T(64, 128, 0, false); T(64, 128, 1, false);
    T(128, 128, 0, false); T(128, 128, 1, false); T(128, 32, 0, false); T(128, 32, 1, false); T(32, 128, 0, false); T(32, 128, 1, false);
    T(256, 32, 1, false); T(32, 32, 0, false); T(128, 64, 1, true); T(128, 128, 1, true); T(256, 64, 1, false); T(256, 64, 0, false);
    T(32, 256, 0, false); T(32, 256, 1, false); T(64, 256, 0, false); T(64, 256, 1, false); T(128, 128, 0, true); T(64, 128, 0, true); T(64, 256, 0, true);
    T(64, 128, 4, false); T(64, 128, 8, false); T(64, 128, 16, false); T(64, 128, 32, false); T(64, 256, 4, false); T(64, 256, 8, false); T(64, 256, 16, false); T(64, 256, 32, false);
    T(64, 128, 16, true); T(64, 256, 16, true); T(32, 256, 8, false); T(32, 256, 16, false); T(128, 128, 8, false); T(128, 128, 16, false);
    if (both) {
        vs.clear();
#define T2(P, Q, O_, NT) vs.push_back({"both-T tile " #P "x" #Q " order" #O_ " nt" #NT, tadd2<P, Q, O_, NT>, P, Q, {}})
        T2(64, 64, 0, false); T2(64, 64, 1, false); T2(64, 128, 0, false); T2(64, 128, 1, false); T2(128, 64, 0, false); T2(128, 64, 1, false);
        T2(128, 128, 0, false); T2(128, 128, 1, false); T2(256, 32, 0, false); T2(256, 32, 1, false); T2(128, 32, 0, false); T2(128, 32, 1, false);
        T2(64, 128, 0, true); T2(128, 64, 0, true); T2(128, 64, 1, true); T2(256, 64, 0, false); T2(256, 64, 1, false); T2(32, 128, 0, false);
    }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int r = 0; r < 5; ++r)
        for (auto& v : vs) {
            const unsigned grid = (N / v.tp) * (N / v.tq);
            v.fn<<<grid, 256>>>(A, B, O, N);
            CK(hipEventRecord(e0));
            for (int k = 0; k < 5; ++k) v.fn<<<grid, 256>>>(A, B, O, N);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); v.ms.push_back(ms / 5);
        }
    // correctness of the last variant run
    if (!both) { std::vector<float> h(8), a(1), b(8); CK(hipMemcpy(h.data(), O + (size_t)77 * N + 1000, 32, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), B + (size_t)77 * N + 1000, 32, hipMemcpyDeviceToHost));
      for (int k = 0; k < 8; ++k) { CK(hipMemcpy(a.data(), A + (size_t)(1000 + k) * N + 77, 4, hipMemcpyDeviceToHost)); if (h[k] != a[0] + b[k]) printf("MISMATCH %d\n", k); } }
    { CK(hipEventRecord(e0)); for (int k = 0; k < 5; ++k) plain<<<n / 4 / 1024, 1024>>>((const f4*)A, (const f4*)B, (f4*)O, n / 4); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); printf("%-30s %.4f ms %7.1f GB/s\n", "plain add (no transpose)", ms / 5, 12.0 * n / (ms / 5) * 1e-6); }
    std::sort(vs.begin(), vs.end(), [](const V& x, const V& y) { auto mx = x.ms, my = y.ms; std::sort(mx.begin(), mx.end()); std::sort(my.begin(), my.end()); return mx[2] < my[2]; });
    for (auto& v : vs) { auto m = v.ms; std::sort(m.begin(), m.end()); printf("%-30s median %.4f ms %7.1f GB/s (%.1f%%)\n", v.name.c_str(), m[2], 12.0 * n / m[2] * 1e-6, 12.0 * n / m[2] * 1e-6 / 80.0); }
    return 0;
}
