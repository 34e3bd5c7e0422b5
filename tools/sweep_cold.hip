// tools/sweep_cold.hip -- launch shapes for MID-SIZE launches on COLD operands (VERDICT r02 "next" #2).
// Every launch-shape sweep before this one ran at N = 2^28 or re-read the same operands each launch; the realistic first
// call of an operator reads arrays nothing has touched recently.  Here launches ROTATE through K disjoint operand sets,
// K x footprint >= 2.5 GiB (10 x the Infinity Cache), for three stream mixes:
//   mode 0  out = a * s            1R+1W  (scalar_vec_kernel's mix)
//   mode 1  out = a + b            2R+1W  (contiguous_vec_kernel's mix)
//   mode 2  out = a * row[i % cv]  1R+1W + a 16 KiB row read through the caches (flat_tile_kernel KIND 3, config 3)
// x workgroup size {256, 512, 1024} x vectors per lane U {1, 2, 4} (one-shot tiles, all loads before the first use)
// x persistent grids {4, 8 workgroups of 256 per CU, grid-stride, U in flight}
// x load policy {plain, nt} x store policy {plain, nt, sc1, sc0 sc1, nt sc1}.
// Also: empty-kernel dispatch time for the same grids (waves per us the dispatcher sustains), and the same rotate
// sequence dealt over TWO streams (how much of the fixed cost per launch is ramp / drain that an independent
// neighbour could cover).
//   hipcc -O3 --offload-arch=gfx950 tools/sweep_cold.hip -o tools/bin/sweep_cold && tools/bin/sweep_cold [same]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));

enum { LD_PLAIN, LD_NT };
enum { ST_PLAIN, ST_NT, ST_SC1, ST_SC0SC1, ST_NTSC1 };
static const char *kLd[] = {"plain", "nt"};
static const char *kSt[] = {"plain", "nt", "sc1", "sc0sc1", "ntsc1"};

template <int LD> __device__ __forceinline__ f4 ld(const f4 *p) {
    if constexpr (LD == LD_NT) return __builtin_nontemporal_load(p);
    else return *p;
}
template <int ST> __device__ __forceinline__ void st(f4 *p, f4 v) {
    if constexpr (ST == ST_PLAIN) *p = v;
    else if constexpr (ST == ST_NT) __builtin_nontemporal_store(v, p);
    else if constexpr (ST == ST_SC1) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v));
    else if constexpr (ST == ST_SC0SC1) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p), "v"(v));
    else asm volatile("global_store_dwordx4 %0, %1, off sc1 nt\n\ts_nop 1" ::"v"(p), "v"(v));
}

// one-shot tile: workgroup g owns vectors [g * BLOCK * U, (g + 1) * BLOCK * U)
template <int MODE, int BLOCK, int U, int LD, int ST>
__global__ __launch_bounds__(BLOCK) void tile(const f4 *__restrict__ a, const f4 *__restrict__ b, f4 *__restrict__ o, float s, unsigned cvmask) {
    const size_t base = (size_t)blockIdx.x * (BLOCK * U) + threadIdx.x;
    f4 va[U], vb[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        va[u] = ld<LD>(a + base + (size_t)u * BLOCK);
        if constexpr (MODE == 1) vb[u] = ld<LD>(b + base + (size_t)u * BLOCK);
        if constexpr (MODE == 2) vb[u] = b[(unsigned)(base + (size_t)u * BLOCK) & cvmask];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        f4 r;
        if constexpr (MODE == 0) r = va[u] * s;
        if constexpr (MODE == 1) r = va[u] + vb[u];
        if constexpr (MODE == 2) r = va[u] * vb[u];
        st<ST>(o + base + (size_t)u * BLOCK, r);
    }
}

// persistent: gridDim.x workgroups walk the tiles grid-stride
template <int MODE, int BLOCK, int U, int LD, int ST>
__global__ __launch_bounds__(BLOCK) void persist(const f4 *__restrict__ a, const f4 *__restrict__ b, f4 *__restrict__ o, float s, unsigned cvmask, size_t n_vec) {
    for (size_t base = (size_t)blockIdx.x * (BLOCK * U) + threadIdx.x; base < n_vec; base += (size_t)gridDim.x * (BLOCK * U)) {
        f4 va[U], vb[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            va[u] = ld<LD>(a + base + (size_t)u * BLOCK);
            if constexpr (MODE == 1) vb[u] = ld<LD>(b + base + (size_t)u * BLOCK);
            if constexpr (MODE == 2) vb[u] = b[(unsigned)(base + (size_t)u * BLOCK) & cvmask];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            f4 r;
            if constexpr (MODE == 0) r = va[u] * s;
            if constexpr (MODE == 1) r = va[u] + vb[u];
            if constexpr (MODE == 2) r = va[u] * vb[u];
            st<ST>(o + base + (size_t)u * BLOCK, r);
        }
    }
}

template <int BLOCK> __global__ __launch_bounds__(BLOCK) void empty_kernel(int *p) { if (p && threadIdx.x == 12345) *p = 1; }

__global__ void init_k(float *p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 1.0f + (float)(i % 1000) * 1e-3f;
}

struct Variant {
    std::string name;
    int mode;
    std::function<void(const float *, const float *, float *, size_t, hipStream_t)> launch;  // a, b, out, n_vec
};

template <int MODE, int BLOCK, int U, int LD, int ST> void add_tile(std::vector<Variant> &v) {
    char buf[96];
    snprintf(buf, sizeof buf, "tile    wg%-4d U%d ld=%-5s st=%-6s", BLOCK, U, kLd[LD], kSt[ST]);
    v.push_back({buf, MODE, [](const float *a, const float *b, float *o, size_t n_vec, hipStream_t s) {
        hipLaunchKernelGGL((tile<MODE, BLOCK, U, LD, ST>), dim3((unsigned)(n_vec / (BLOCK * U))), dim3(BLOCK), 0, s, (const f4 *)a, (const f4 *)b, (f4 *)o, 1.0000001f, 1023u);
    }});
}
template <int MODE, int PERCU, int U, int LD, int ST> void add_persist(std::vector<Variant> &v) {
    char buf[96];
    snprintf(buf, sizeof buf, "persist %d/CU  U%d ld=%-5s st=%-6s", PERCU, U, kLd[LD], kSt[ST]);
    v.push_back({buf, MODE, [](const float *a, const float *b, float *o, size_t n_vec, hipStream_t s) {
        hipLaunchKernelGGL((persist<MODE, 256, U, LD, ST>), dim3(256 * PERCU), dim3(256), 0, s, (const f4 *)a, (const f4 *)b, (f4 *)o, 1.0000001f, 1023u, n_vec);
    }});
}
template <int MODE, int LD, int ST> void add_shapes(std::vector<Variant> &v) {
    add_tile<MODE, 256, 1, LD, ST>(v); add_tile<MODE, 256, 2, LD, ST>(v); add_tile<MODE, 256, 4, LD, ST>(v);
    add_tile<MODE, 512, 1, LD, ST>(v); add_tile<MODE, 512, 2, LD, ST>(v); add_tile<MODE, 512, 4, LD, ST>(v);
    add_tile<MODE, 1024, 1, LD, ST>(v); add_tile<MODE, 1024, 2, LD, ST>(v); add_tile<MODE, 1024, 4, LD, ST>(v);
    add_persist<MODE, 4, 2, LD, ST>(v); add_persist<MODE, 8, 1, LD, ST>(v); add_persist<MODE, 8, 2, LD, ST>(v); add_persist<MODE, 8, 4, LD, ST>(v);
}
template <int MODE> void add_mode(std::vector<Variant> &v) {
    add_shapes<MODE, LD_PLAIN, ST_NT>(v); add_shapes<MODE, LD_NT, ST_NT>(v);
    add_shapes<MODE, LD_PLAIN, ST_SC1>(v); add_shapes<MODE, LD_NT, ST_SC1>(v);
    add_shapes<MODE, LD_PLAIN, ST_PLAIN>(v); add_shapes<MODE, LD_NT, ST_PLAIN>(v);
    add_shapes<MODE, LD_NT, ST_SC0SC1>(v); add_shapes<MODE, LD_NT, ST_NTSC1>(v);
}

int main(int argc, char **argv) {
    const bool same = argc > 1 && !strcmp(argv[1], "same");
    const bool quick = argc > 1 && !strcmp(argv[1], "quick");
    const bool streams_only = argc > 1 && !strcmp(argv[1], "streams");  // only the last section: the rotate sequence over 1 ... 4 streams
    const size_t slab_bytes = (size_t)6 << 30;
    float *slab;
    CK(hipMalloc(&slab, slab_bytes));
    hipStream_t s0, s1;
    CK(hipStreamCreate(&s0));
    CK(hipStreamCreate(&s1));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    init_k<<<4096, 256, 0, s0>>>(slab, slab_bytes / 4);
    CK(hipStreamSynchronize(s0));
    std::vector<Variant> variants;
    add_mode<0>(variants);
    add_mode<1>(variants);
    add_mode<2>(variants);
    static const char *kMode[] = {"1R+1W a*s", "2R+1W a+b", "row   a*row"};

    auto timed = [&](const std::function<void(int)> &body, int reps) {
        int seq = 0;
        for (int i = 0; i < 24; ++i) body(seq++);
        std::vector<float> ms(5);
        for (auto &m : ms) {
            CK(hipEventRecord(e0, s0));
            for (int i = 0; i < reps; ++i) body(seq++);
            CK(hipEventRecord(e1, s0));
            CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&m, e0, e1));
            m /= reps;
        }
        std::sort(ms.begin(), ms.end());
        return ms[2];
    };

    if (!streams_only) {
    // ---- dispatch alone: empty kernels with the grids the shapes above use
    printf("# empty-kernel dispatch (back to back on one stream), us per launch and waves per us\n");
    for (size_t mib : {16, 64, 128}) {
        const size_t n_vec = (mib << 20) / 16;
        auto line = [&](const char *what, unsigned grid, int block, float ms) {
            printf("dispatch %4zu MiB  %-18s grid %7u x %4d : %7.2f us  %8.0f waves/us\n", mib, what, grid, block, ms * 1e3, grid * (block / 64.0) / (ms * 1e3));
        };
        line("wg256 U1", n_vec / 256, 256, timed([&](int) { empty_kernel<256><<<(unsigned)(n_vec / 256), 256, 0, s0>>>(nullptr); }, 40));
        line("wg256 U2", n_vec / 512, 256, timed([&](int) { empty_kernel<256><<<(unsigned)(n_vec / 512), 256, 0, s0>>>(nullptr); }, 40));
        line("wg256 U4", n_vec / 1024, 256, timed([&](int) { empty_kernel<256><<<(unsigned)(n_vec / 1024), 256, 0, s0>>>(nullptr); }, 40));
        line("wg1024 U1", n_vec / 1024, 1024, timed([&](int) { empty_kernel<1024><<<(unsigned)(n_vec / 1024), 1024, 0, s0>>>(nullptr); }, 40));
        line("wg1024 U4", n_vec / 4096, 1024, timed([&](int) { empty_kernel<1024><<<(unsigned)(n_vec / 4096), 1024, 0, s0>>>(nullptr); }, 40));
        line("persist 8/CU", 2048, 256, timed([&](int) { empty_kernel<256><<<2048, 256, 0, s0>>>(nullptr); }, 40));
    }
    fflush(stdout);

    struct Row { std::string name; double us, pct; };
    for (int mode = 0; mode < 3; ++mode) {
        for (size_t mib : {16, 32, 64, 128}) {
            if (quick && mib != 64) continue;
            const size_t n = (mib << 20) / 4, n_vec = n / 4;
            const int arrays = mode == 1 ? 3 : 2;
            const double bytes = (double)arrays * n * 4 + (mode == 2 ? 16384 : 0);
            // K operand sets; the row (mode 2) is shared: it is 16 KiB
            const size_t set_floats = (size_t)arrays * n;
            int K = (int)std::min<size_t>((slab_bytes / 4 - 4096) / set_floats, std::max<size_t>(2, ((size_t)2560 << 20) / (set_floats * 4) + 1));
            if (same) K = 1;
            float *row = slab + slab_bytes / 4 - 4096;
            std::vector<Row> rows;
            for (auto &v : variants) {
                if (v.mode != mode) continue;
                auto body = [&](int i) {
                    float *base = slab + (size_t)(i % K) * set_floats;
                    const float *a = base, *b = mode == 1 ? base + n : row;
                    float *o = base + (size_t)(arrays - 1) * n;
                    v.launch(a, b, o, n_vec, s0);
                };
                const float ms = timed(body, 40);
                rows.push_back({v.name, ms * 1e3, bytes / (ms * 1e-3) / 8e12 * 100});
            }
            std::vector<Row> sorted = rows;
            std::sort(sorted.begin(), sorted.end(), [](const Row &x, const Row &y) { return x.us < y.us; });
            printf("\n== %s, %zu MiB per array, %s (K = %d sets), %.0f bytes per launch\n", kMode[mode], mib, same ? "SAME operands" : "ROTATING cold operands", K, bytes);
            for (auto &r : rows) printf("%-4zu m%d %-44s %8.2f us %6.1f %%\n", mib, mode, r.name.c_str(), r.us, r.pct);
            printf("-- best 8:\n");
            for (size_t i = 0; i < 8 && i < sorted.size(); ++i) printf("   %-44s %8.2f us %6.1f %%\n", sorted[i].name.c_str(), sorted[i].us, sorted[i].pct);
            fflush(stdout);
        }
    }

    }  // !streams_only
    // ---- the rotate sequence dealt over 1 ... 4 streams: launches on different streams are independent, so one's ramp can
    // cover the other's drain -- and the hardware runs them side by side.  (r03: two streams, diagnosis only; r04: the library
    // does this on two queues; would three or four buy more?)
    hipStream_t ss[4] = {s0, s1, nullptr, nullptr};
    CK(hipStreamCreate(&ss[2]));
    CK(hipStreamCreate(&ss[3]));
    printf("\n# 1 ... 4 streams in turn, rotate, library-like shapes: us per launch (wall over all streams)\n");
    for (int mode = 0; mode < 3; ++mode) {
        for (size_t mib : {16, 32, 64, 128}) {
            const size_t n = (mib << 20) / 4, n_vec = n / 4;
            const int arrays = mode == 1 ? 3 : 2;
            const double bytes = (double)arrays * n * 4 + (mode == 2 ? 16384 : 0);
            const size_t set_floats = (size_t)arrays * n;
            const int K = (int)std::min<size_t>((slab_bytes / 4 - 4096) / set_floats, std::max<size_t>(2, ((size_t)2560 << 20) / (set_floats * 4) + 1));
            float *row = slab + slab_bytes / 4 - 4096;
            for (auto &v : variants) {
                if (v.mode != mode) continue;
                if (v.name.find("ld=nt") == std::string::npos || v.name.find("st=nt ") == std::string::npos) continue;
                if (v.name.find("tile    wg256  U1") == std::string::npos && v.name.find("tile    wg256  U2") == std::string::npos) continue;
                for (int ns = streams_only ? 1 : 2; ns <= (streams_only ? 4 : 2); ++ns) {
                    int seq = 0;
                    auto body = [&](hipStream_t st_) {
                        const int i = seq++;
                        float *base = slab + (size_t)(i % K) * set_floats;
                        v.launch(base, mode == 1 ? base + n : row, base + (size_t)(arrays - 1) * n, n_vec, st_);
                    };
                    for (int i = 0; i < 24; ++i) body(ss[i % ns]);
                    CK(hipDeviceSynchronize());
                    std::vector<float> us;
                    for (int r = 0; r < 5; ++r) {
                        hipEvent_t j[4];
                        CK(hipEventRecord(e0, s0));
                        for (int q = 1; q < ns; ++q) CK(hipStreamWaitEvent(ss[q], e0, 0));
                        for (int i = 0; i < 96; ++i) body(ss[i % ns]);
                        for (int q = 1; q < ns; ++q) {
                            CK(hipEventCreateWithFlags(&j[q], hipEventDisableTiming));
                            CK(hipEventRecord(j[q], ss[q]));
                            CK(hipStreamWaitEvent(s0, j[q], 0));
                        }
                        CK(hipEventRecord(e1, s0));
                        CK(hipEventSynchronize(e1));
                        float ms;
                        CK(hipEventElapsedTime(&ms, e0, e1));
                        us.push_back(ms * 1e3f / 96);
                        for (int q = 1; q < ns; ++q) CK(hipEventDestroy(j[q]));
                    }
                    std::sort(us.begin(), us.end());
                    printf("%dstreams %4zu MiB m%d %-44s %8.2f us %6.1f %%\n", ns, mib, mode, v.name.c_str(), us[2], bytes / (us[2] * 1e-6) / 8e12 * 100);
                }
            }
            fflush(stdout);
        }
    }
    return 0;
}
