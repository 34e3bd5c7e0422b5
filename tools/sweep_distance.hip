// tools/sweep_distance.hip -- the large-array sag (f32 add: 81-82 % of HBM peak at N = 2^28, 76-78 % at 2^30 / 2^31) is not
// address translation after all: memory mapped through hipMemCreate / hipMemMap shows 24 x the UTCL1 misses and a UTCL2 that
// is busy 55 % of the kernel instead of 2 %, and runs 2 % FASTER (tools/sweep_vmm.hip, tools/pmc_vmm.sh).  What else grows
// with N?  The DISTANCE between the three streams (a, b, c sit N * 4 bytes apart) and the length of the launch.  Here:
//   (1) N = 2^28 (the fast size) with a, b, c placed D bytes apart inside one 26 GiB slab, D = 1 GiB ... 8 GiB and odd offsets;
//   (2) N = 2^30 with extra gaps between the arrays;
//   (3) N = 2^30 done as FOUR launches of 2^28 (same bytes, same placement): is it the launch's length / footprint?
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(1024) void add_k(const f4 *__restrict__ a, const f4 *__restrict__ b, f4 *__restrict__ o) {
    const size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x;
    __builtin_nontemporal_store(__builtin_nontemporal_load(a + i) + __builtin_nontemporal_load(b + i), o + i);
}
__global__ void init_k(float *p, size_t n) { for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 1.0f + (float)(i % 1000) * 1e-3f; }
int main() {
    const size_t GiB = (size_t)1 << 30, MiB = (size_t)1 << 20;
    const size_t slab_bytes = 26 * GiB;
    char *slab;
    CK(hipMalloc(&slab, slab_bytes));
    init_k<<<8192, 256>>>((float *)slab, slab_bytes / 4);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto run = [&](size_t n, size_t off_a, size_t off_b, size_t off_c, int pieces, const char *what) {
        const size_t piece = n / pieces;
        const unsigned grid = (unsigned)(piece / 4 / 1024);
        auto go = [&] {
            for (int p = 0; p < pieces; ++p)
                add_k<<<grid, 1024>>>((const f4 *)(slab + off_a) + p * (piece / 4), (const f4 *)(slab + off_b) + p * (piece / 4), (f4 *)(slab + off_c) + p * (piece / 4));
        };
        for (int i = 0; i < 3; ++i) go();
        std::vector<float> ms(5);
        const int reps = n >= ((size_t)1 << 30) ? 8 : 30;
        for (auto &m : ms) {
            CK(hipEventRecord(e0));
            for (int i = 0; i < reps; ++i) go();
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&m, e0, e1));
            m /= reps;
        }
        std::sort(ms.begin(), ms.end());
        printf("%-78s %9.1f us  %5.1f %%\n", what, ms[2] * 1e3, 12.0 * n / (ms[2] * 1e-3) / 8e12 * 100);
        fflush(stdout);
    };
    char buf[160];
    const size_t n28 = (size_t)1 << 28, n30 = (size_t)1 << 30;
    printf("(1) N = 2^28, streams D apart\n");
    for (double d : {1.0, 1.0 + 1.0 / 1024, 1.0 + 1.0 / 64, 1.0625, 1.25, 1.5, 2.0, 3.0, 4.0, 4.0 + 1.0 / 64, 4.25, 5.0, 6.0, 8.0, 8.0 + 1.0 / 64, 12.0}) {
        const size_t D = (size_t)(d * 1024) * MiB;
        snprintf(buf, sizeof buf, "2^28  D = %8.3f GiB", d);
        run(n28, 0, D, 2 * D, 1, buf);
    }
    printf("(2) N = 2^30 (arrays of 4 GiB), gap G between them\n");
    for (double g : {0.0, 1.0 / 1024, 1.0 / 64, 0.25, 1.0, 2.0}) {
        const size_t D = 4 * GiB + (size_t)(g * 1024) * MiB;
        snprintf(buf, sizeof buf, "2^30  gap = %8.3f GiB", g);
        run(n30, 0, D, 2 * D, 1, buf);
    }
    printf("(3) N = 2^30 as several launches\n");
    run(n30, 0, 4 * GiB, 8 * GiB, 1, "2^30  one launch");
    run(n30, 0, 4 * GiB, 8 * GiB, 4, "2^30  four launches of 2^28, same placement");
    run(n30, 0, 4 * GiB, 8 * GiB, 16, "2^30  sixteen launches of 2^26, same placement");
    printf("(4) N = 2^28 at the far end of the slab (physical placement?)\n");
    run(n28, 20 * GiB, 21 * GiB, 22 * GiB, 1, "2^28  at 20 / 21 / 22 GiB");
    run(n28, 0, 1 * GiB, 2 * GiB, 1, "2^28  at 0 / 1 / 2 GiB");
    return 0;
}
