// tools/sweep_pow.hip -- launch-shape sweep for pow(a, 2.5) (BASELINE config 4, N = 2^26, 8 B/elem).
// Development tool: hipcc -O3 -ffp-contract=off --offload-arch=gfx950 -Isimplemath_amd/csrc -o tools/bin/sweep_pow tools/sweep_pow.hip
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

template <int N> __device__ __forceinline__ void pow_vecs(const double* tab, const f4 (&in)[N], float s, f4 (&out)[N]) {
    float x[4 * N], y[4 * N], r[4 * N];
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
        for (int k = 0; k < 4; ++k) { x[4 * i + k] = in[i][k]; y[4 * i + k] = s; }
    smpow::pow_n<4 * N>(x, y, r, tab);
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
        for (int k = 0; k < 4; ++k) out[i][k] = r[4 * i + k];
}

// one-shot: U vectors per thread, no loop
template <int U, int BLOCK>
__global__ __launch_bounds__(BLOCK) void oneshot(const f4* __restrict__ a, float s, f4* __restrict__ o, size_t nvec) {
    const double* tab = stage_table();
    const size_t base = (size_t)blockIdx.x * BLOCK * U + threadIdx.x;
    f4 v[U], r[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load(a + base + (size_t)u * BLOCK);
    pow_vecs<U>(tab, v, s, r);
#pragma unroll
    for (int u = 0; u < U; ++u) __builtin_nontemporal_store(r[u], o + base + (size_t)u * BLOCK);
}

// persistent grid-stride with one-vector software prefetch
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void prefetch(const f4* __restrict__ a, float s, f4* __restrict__ o, size_t nvec) {
    const double* tab = stage_table();
    const size_t stride = (size_t)gridDim.x * BLOCK;
    size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= nvec) return;
    f4 cur[1] = {__builtin_nontemporal_load(a + i)};
    for (;;) {
        const size_t nx = i + stride;
        f4 nxt[1];
        const bool more = nx < nvec;
        if (more) nxt[0] = __builtin_nontemporal_load(a + nx);
        f4 r[1];
        pow_vecs<1>(tab, cur, s, r);
        __builtin_nontemporal_store(r[0], o + i);
        if (!more) break;
        cur[0] = nxt[0]; i = nx;
    }
}

// prefetch + wave priority raised while the next load is issued (memory instructions go out ahead of other waves' VALU)
template <int BLOCK, int MODE>
__global__ __launch_bounds__(BLOCK) void prefetch_prio(const f4* __restrict__ a, float s, f4* __restrict__ o, size_t nvec) {
    const double* tab = stage_table();
    const size_t stride = (size_t)gridDim.x * BLOCK;
    size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= nvec) return;
    f4 cur[1] = {__builtin_nontemporal_load(a + i)};
    for (;;) {
        const size_t nx = i + stride;
        f4 nxt[1];
        const bool more = nx < nvec;
        if (MODE & 1) __builtin_amdgcn_s_setprio(3);
        if (more) nxt[0] = __builtin_nontemporal_load(a + nx);
        if (MODE & 1) __builtin_amdgcn_s_setprio(0);
        f4 r[1];
        if (MODE & 2) {  // half the arithmetic: how sensitive is the time to VALU work?
            float x[4], y[4], rr[4];
            for (int k = 0; k < 4; ++k) { x[k] = cur[0][k]; y[k] = s; }
            smpow::pow_n<2>(reinterpret_cast<const float(&)[2]>(x[0]), reinterpret_cast<const float(&)[2]>(y[0]), reinterpret_cast<float(&)[2]>(rr[0]), tab);
            r[0][0] = rr[0]; r[0][1] = rr[1]; r[0][2] = x[2] * 2.5f; r[0][3] = x[3] * 2.5f;
        } else {
            pow_vecs<1>(tab, cur, s, r);
        }
        if (MODE & 4) __builtin_amdgcn_s_setprio(3);
        __builtin_nontemporal_store(r[0], o + i);
        if (MODE & 4) __builtin_amdgcn_s_setprio(0);
        if (!more) break;
        cur[0] = nxt[0]; i = nx;
    }
}

// calibration: the same access shapes with the arithmetic removed (1R + 1W copy)
template <int U, int BLOCK>
__global__ __launch_bounds__(BLOCK) void copy_oneshot(const f4* __restrict__ a, f4* __restrict__ o) {
    const size_t base = (size_t)blockIdx.x * BLOCK * U + threadIdx.x;
    f4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load(a + base + (size_t)u * BLOCK);
#pragma unroll
    for (int u = 0; u < U; ++u) __builtin_nontemporal_store(v[u], o + base + (size_t)u * BLOCK);
}
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void copy_stride(const f4* __restrict__ a, f4* __restrict__ o, size_t nvec) {
    const size_t stride = (size_t)gridDim.x * BLOCK;
    for (size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x; i < nvec; i += stride) __builtin_nontemporal_store(__builtin_nontemporal_load(a + i), o + i);
}

// persistent, prefetch TWO vectors ahead
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void prefetch2(const f4* __restrict__ a, float s, f4* __restrict__ o, size_t nvec) {
    const double* tab = stage_table();
    const size_t stride = (size_t)gridDim.x * BLOCK;
    size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= nvec) return;
    f4 c0[1] = {__builtin_nontemporal_load(a + i)};
    f4 c1[1];
    bool have1 = i + stride < nvec;
    if (have1) c1[0] = __builtin_nontemporal_load(a + i + stride);
    for (;;) {
        const size_t n2 = i + 2 * stride;
        const bool have2 = n2 < nvec;
        f4 c2[1];
        if (have2) c2[0] = __builtin_nontemporal_load(a + n2);
        f4 r[1];
        pow_vecs<1>(tab, c0, s, r);
        __builtin_nontemporal_store(r[0], o + i);
        if (!have1) break;
        c0[0] = c1[0]; c1[0] = c2[0]; have1 = have2; i += stride;
    }
}
// persistent, two vectors per iteration (evaluated together: 8 elements side by side), prefetch the next pair
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void prefetch_pair(const f4* __restrict__ a, float s, f4* __restrict__ o, size_t nvec) {
    const double* tab = stage_table();
    const size_t stride = (size_t)gridDim.x * BLOCK * 2;
    size_t i = (size_t)blockIdx.x * BLOCK * 2 + threadIdx.x;   // pair: i, i + BLOCK   (nvec multiple of 2*BLOCK assumed)
    if (i >= nvec) return;
    f4 cur[2] = {__builtin_nontemporal_load(a + i), __builtin_nontemporal_load(a + i + BLOCK)};
    for (;;) {
        const size_t nx = i + stride;
        const bool more = nx < nvec;
        f4 nxt[2];
        if (more) { nxt[0] = __builtin_nontemporal_load(a + nx); nxt[1] = __builtin_nontemporal_load(a + nx + BLOCK); }
        f4 r[2];
        pow_vecs<2>(tab, cur, s, r);
        __builtin_nontemporal_store(r[0], o + i); __builtin_nontemporal_store(r[1], o + i + BLOCK);
        if (!more) break;
        cur[0] = nxt[0]; cur[1] = nxt[1]; i = nx;
    }
}

__global__ void init_k(float* p, size_t n) { for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 0.01f + (float)((i * 2654435761u) & 0xffffff) * (99.99f / 16777216.0f); }

template <typename F> double timeit(F launch) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) launch();
    CK(hipDeviceSynchronize());
    std::vector<float> ms(15);
    for (auto& m : ms) { CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&m, e0, e1)); }
    std::sort(ms.begin(), ms.end());
    return ms[7];
}

int main() {
    const size_t n = (size_t)1 << 26, nvec = n / 4;
    float *a, *o; CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&o, n * 4));
    init_k<<<4096, 256>>>(a, n); CK(hipDeviceSynchronize());
    const f4* av = (const f4*)a; f4* ov = (f4*)o;
    auto rep = [&](const char* name, double ms) { printf("%-34s %.4f ms %8.1f GB/s %7.1f Gelem/s\n", name, ms, 8.0 * n / ms * 1e-6, n / ms * 1e-6); fflush(stdout); };
    rep("COPY oneshot U1 block1024", timeit([&] { copy_oneshot<1, 1024><<<nvec / 1024, 1024>>>(av, ov); }));
    rep("COPY oneshot U1 block256", timeit([&] { copy_oneshot<1, 256><<<nvec / 256, 256>>>(av, ov); }));
    rep("COPY oneshot U2 block256", timeit([&] { copy_oneshot<2, 256><<<nvec / 512, 256>>>(av, ov); }));
    rep("COPY oneshot U4 block256", timeit([&] { copy_oneshot<4, 256><<<nvec / 1024, 256>>>(av, ov); }));
    rep("COPY stride block256 x8", timeit([&] { copy_stride<256><<<256 * 8, 256>>>(av, ov, nvec); }));
    rep("COPY stride block512 x4", timeit([&] { copy_stride<512><<<256 * 4, 512>>>(av, ov, nvec); }));
    rep("COPY stride block512 x8", timeit([&] { copy_stride<512><<<256 * 8, 512>>>(av, ov, nvec); }));
#define ONESHOT(U, B) rep("oneshot U" #U " block" #B, timeit([&] { oneshot<U, B><<<nvec / (B * U), B>>>(av, 2.5f, ov, nvec); }))
    ONESHOT(1, 1024); ONESHOT(1, 512); ONESHOT(1, 256); ONESHOT(1, 128); ONESHOT(1, 64);
    ONESHOT(2, 1024); ONESHOT(2, 512); ONESHOT(2, 256); ONESHOT(2, 128); ONESHOT(2, 64);
    ONESHOT(4, 256); ONESHOT(4, 64);
    for (int mult : {2, 4, 8, 16, 32}) {
        char nm[64];
        snprintf(nm, sizeof nm, "prefetch block256 grid=256x%d", mult); rep(nm, timeit([&] { prefetch<256><<<256 * mult, 256>>>(av, 2.5f, ov, nvec); }));
        snprintf(nm, sizeof nm, "prefetch block512 grid=256x%d", mult); rep(nm, timeit([&] { prefetch<512><<<256 * mult, 512>>>(av, 2.5f, ov, nvec); }));
        snprintf(nm, sizeof nm, "prefetch block64 grid=256x%d", mult * 4); rep(nm, timeit([&] { prefetch<64><<<256 * mult * 4, 64>>>(av, 2.5f, ov, nvec); }));
    }
    // back-to-back launches (what bench.py times): does the isolated-launch time hold?
    {
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        auto b2b = [&](const char* name, auto launch) {
            for (int i = 0; i < 5; ++i) launch();
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            for (int i = 0; i < 100; ++i) launch();
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("B2B x100 %-40s %.4f ms/launch %8.1f GB/s\n", name, ms / 100, 8.0 * n / (ms / 100) * 1e-6);
        };
        b2b("prefetch b512 x32 (nt ld+st)", [&] { prefetch<512><<<256 * 32, 512>>>(av, 2.5f, ov, nvec); });
        b2b("prefetch_prio(load) b512 x32", [&] { prefetch_prio<512, 1><<<256 * 32, 512>>>(av, 2.5f, ov, nvec); });
        b2b("prefetch_prio(load+store) b512 x32", [&] { prefetch_prio<512, 5><<<256 * 32, 512>>>(av, 2.5f, ov, nvec); });
        b2b("prefetch HALF arithmetic b512 x32", [&] { prefetch_prio<512, 2><<<256 * 32, 512>>>(av, 2.5f, ov, nvec); });
        b2b("prefetch b512 x32 again", [&] { prefetch<512><<<256 * 32, 512>>>(av, 2.5f, ov, nvec); });
        b2b("prefetch2 b512 x32", [&] { prefetch2<512><<<256 * 32, 512>>>(av, 2.5f, ov, nvec); });
        b2b("prefetch2 b512 x8", [&] { prefetch2<512><<<256 * 8, 512>>>(av, 2.5f, ov, nvec); });
        b2b("prefetch2 b256 x16", [&] { prefetch2<256><<<256 * 16, 256>>>(av, 2.5f, ov, nvec); });
        b2b("prefetch2 b256 x64", [&] { prefetch2<256><<<256 * 64, 256>>>(av, 2.5f, ov, nvec); });
        b2b("prefetch_pair b512 x16", [&] { prefetch_pair<512><<<256 * 16, 512>>>(av, 2.5f, ov, nvec); });
        b2b("prefetch_pair b256 x16", [&] { prefetch_pair<256><<<256 * 16, 256>>>(av, 2.5f, ov, nvec); });
        b2b("prefetch_pair b256 x32", [&] { prefetch_pair<256><<<256 * 32, 256>>>(av, 2.5f, ov, nvec); });
        b2b("prefetch b1024 x16", [&] { prefetch<1024><<<256 * 16, 1024>>>(av, 2.5f, ov, nvec); });
        b2b("prefetch b512 x64", [&] { prefetch<512><<<256 * 64, 512>>>(av, 2.5f, ov, nvec); });
        b2b("prefetch b512 x8", [&] { prefetch<512><<<256 * 8, 512>>>(av, 2.5f, ov, nvec); });
        b2b("prefetch b512 x4", [&] { prefetch<512><<<256 * 4, 512>>>(av, 2.5f, ov, nvec); });
        b2b("prefetch b256 x16", [&] { prefetch<256><<<256 * 16, 256>>>(av, 2.5f, ov, nvec); });
        b2b("oneshot U4 b256", [&] { oneshot<4, 256><<<nvec / 1024, 256>>>(av, 2.5f, ov, nvec); });
        b2b("oneshot U2 b256", [&] { oneshot<2, 256><<<nvec / 512, 256>>>(av, 2.5f, ov, nvec); });
        b2b("COPY oneshot U1 b1024", [&] { copy_oneshot<1, 1024><<<nvec / 1024, 1024>>>(av, ov); });
        // alternate between two output buffers so a launch never rewrites lines the previous one left dirty
        float* o2; CK(hipMalloc(&o2, n * 4)); f4* ov2 = (f4*)o2; int flip = 0;
        b2b("prefetch b512 x32, alternating outputs", [&] { prefetch<512><<<256 * 32, 512>>>(av, 2.5f, (flip ^= 1) ? ov : ov2, nvec); });
    }
    return 0;
}
