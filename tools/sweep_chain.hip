// tools/sweep_chain.hip -- is the plain-load advantage below 256 MiB anything but the benchmark re-reading the same operand?
// out = a * s (1R+1W, one 16-byte vector per lane, 256-thread workgroups) in three settings:
//   same    : every launch reads the same a and writes the same out (what bench.py and the sweeps before this one do)
//   rotate  : launches walk K different (a, out) pairs, K x footprint >= 2 GiB: nothing a launch reads was touched recently
//   chain   : ping-pong, each launch reads what the previous one wrote (an operator chain on resident arrays)
// for load policy {plain, nt} x store policy {plain, nt}.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
template <int LD, int ST> __global__ __launch_bounds__(256) void scal(const f4* __restrict__ a, float s, f4* __restrict__ o) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    f4 v; if constexpr (LD) v = __builtin_nontemporal_load(a + i); else v = a[i];
    if constexpr (ST) __builtin_nontemporal_store(v * s, o + i); else o[i] = v * s;
}
__global__ void init_k(float* p, size_t n) { for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 1.0f + (float)(i % 1000) * 1e-3f; }
template <int LD, int ST> void launch(const float* a, float* o, size_t nvec) { scal<LD, ST><<<(unsigned)(nvec / 256), 256>>>((const f4*)a, 1.0000001f, (f4*)o); }
int main() {
    const size_t slab_floats = (size_t)5 << 28;  // 5 GiB
    float* slab; CK(hipMalloc(&slab, slab_floats * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    init_k<<<4096, 256>>>(slab, slab_floats); CK(hipDeviceSynchronize());
    auto timed = [&](auto body, int reps) {  // body(i) launches the i-th kernel of a long sequence
        int seq = 0;
        for (int i = 0; i < 24; ++i) body(seq++);
        std::vector<float> ms(5);
        for (auto& m : ms) { CK(hipEventRecord(e0)); for (int i = 0; i < reps; ++i) body(seq++); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&m, e0, e1)); m /= reps; }
        std::sort(ms.begin(), ms.end());
        return ms[2];
    };
    printf("%-8s %-8s %-22s %10s %10s %10s\n", "MiB", "setting", "loads / stores", "us", "GB/s", "% of 8TB/s");
    for (size_t mib : {16, 32, 64, 128, 256}) {
        const size_t n = mib << 18, nvec = n / 4;
        const int K = (int)std::max<size_t>(2, (2048 + mib - 1) / mib / 2);  // pairs: K * 2 * mib >= 2 GiB (fits the 5 GiB slab)
        auto a_of = [&](int k) { return slab + (size_t)(2 * k) * n; };
        auto o_of = [&](int k) { return slab + (size_t)(2 * k + 1) * n; };
        auto report = [&](const char* setting, const char* pol, float ms) { printf("%-8zu %-8s %-22s %10.1f %10.0f %9.1f%%\n", mib, setting, pol, ms * 1000, 8.0 * n / ms * 1e-6, 8.0 * n / ms * 1e-6 / 80); fflush(stdout); };
#define POLICIES(X) X(0, 1, "plain / nt") X(1, 1, "nt / nt") X(0, 0, "plain / plain") X(1, 0, "nt / plain")
#define SAME(LD, ST, NAME) report("same", NAME, timed([&](int) { launch<LD, ST>(a_of(0), o_of(0), nvec); }, 40));
#define ROT(LD, ST, NAME) report("rotate", NAME, timed([&](int i) { launch<LD, ST>(a_of(i % K), o_of(i % K), nvec); }, 40));
#define CHAIN(LD, ST, NAME) report("chain", NAME, timed([&](int i) { (i & 1) ? launch<LD, ST>(o_of(0), a_of(0), nvec) : launch<LD, ST>(a_of(0), o_of(0), nvec); }, 40));
        POLICIES(SAME) POLICIES(ROT) POLICIES(CHAIN)
    }
    return 0;
}
