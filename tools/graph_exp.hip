// tools/graph_exp.hip -- is a chain of small elementwise launches cheaper as a hipGraph replay?  Development tool.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void add(const f4* a, const f4* b, f4* c, size_t nvec) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < nvec) c[i] = a[i] + b[i];
}
int main() {
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    for (int lg : {10, 16, 20, 22}) {
        const size_t n = (size_t)1 << lg, nvec = n / 4;
        float *a, *b, *t[8];
        CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&b, n * 4)); for (auto& p : t) CK(hipMalloc(&p, n * 4));
        CK(hipMemset(a, 0, n * 4)); CK(hipMemset(b, 0, n * 4));
        const unsigned grid = (unsigned)((nvec + 255) / 256);
        for (int chain : {3, 8}) {
            auto enqueue = [&] {
                const float* x = a;
                for (int k = 0; k < chain; ++k) { add<<<grid, 256, 0, s>>>((const f4*)x, (const f4*)b, (f4*)t[k], nvec); x = t[k]; }
            };
            for (int w = 0; w < 200; ++w) enqueue();
            CK(hipStreamSynchronize(s));
            const int reps = 2000;
            auto t0 = std::chrono::steady_clock::now();
            for (int r = 0; r < reps; ++r) enqueue();
            auto t1 = std::chrono::steady_clock::now();
            CK(hipStreamSynchronize(s));
            auto t2 = std::chrono::steady_clock::now();
            hipGraph_t g; hipGraphExec_t ge;
            CK(hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed)); enqueue(); CK(hipStreamEndCapture(s, &g));
            CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            for (int w = 0; w < 200; ++w) CK(hipGraphLaunch(ge, s));
            CK(hipStreamSynchronize(s));
            auto g0 = std::chrono::steady_clock::now();
            for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(ge, s));
            auto g1 = std::chrono::steady_clock::now();
            CK(hipStreamSynchronize(s));
            auto g2 = std::chrono::steady_clock::now();
            auto us = [](auto x, auto y) { return std::chrono::duration<double, std::micro>(y - x).count(); };
            printf("n=2^%-2d chain=%d  launches: enqueue %.2f us/chain, total %.2f us/chain | graph: enqueue %.2f us, total %.2f us/chain\n", lg, chain,
                   us(t0, t1) / reps, us(t0, t2) / reps, us(g0, g1) / reps, us(g0, g2) / reps);
            CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
        }
        CK(hipFree(a)); CK(hipFree(b)); for (auto& p : t) CK(hipFree(p));
    }
    return 0;
}
