// tools/sweep_load_enc.hip -- every cache-policy encoding of the LOADS of the headline kernel (f32 add, N = 2^28, 2R+1W).
// Rounds 1-2 compared plain and `nt` loads (the two the compiler can spell) and swept all eight encodings for STORES only.
// Here the two loads are inline asm with each of sc0 / sc1 / nt combinations, against nt / sc1 / plain stores.
//   hipcc -O3 --offload-arch=gfx950 tools/sweep_load_enc.hip -o tools/bin/sweep_load_enc
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
#define LOAD_KERNEL(NAME, POL)                                                                                                  \
    template <int BLOCK, int ST> __global__ __launch_bounds__(BLOCK) void NAME(const f4 *__restrict__ a, const f4 *__restrict__ b, f4 *__restrict__ o) { \
        const size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;                                                              \
        f4 va, vb;                                                                                                              \
        asm volatile("global_load_dwordx4 %0, %2, off " POL "\n\tglobal_load_dwordx4 %1, %3, off " POL "\n\ts_waitcnt vmcnt(0)" \
                     : "=&v"(va), "=&v"(vb) : "v"(a + i), "v"(b + i) : "memory");                                                \
        const f4 r = va + vb;                                                                                                   \
        if constexpr (ST == 0) __builtin_nontemporal_store(r, o + i);                                                           \
        else if constexpr (ST == 1) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(o + i), "v"(r));     \
        else o[i] = r;                                                                                                          \
    }
LOAD_KERNEL(k_plain, "")
LOAD_KERNEL(k_nt, "nt")
LOAD_KERNEL(k_sc0, "sc0")
LOAD_KERNEL(k_sc1, "sc1")
LOAD_KERNEL(k_sc0sc1, "sc0 sc1")
LOAD_KERNEL(k_sc0nt, "sc0 nt")
LOAD_KERNEL(k_sc1nt, "sc1 nt")
LOAD_KERNEL(k_sc0sc1nt, "sc0 sc1 nt")
template <int BLOCK> __global__ __launch_bounds__(BLOCK) void k_builtin(const f4 *__restrict__ a, const f4 *__restrict__ b, f4 *__restrict__ o) {
    const size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    __builtin_nontemporal_store(__builtin_nontemporal_load(a + i) + __builtin_nontemporal_load(b + i), o + i);
}
__global__ void init_k(float *p, size_t n, float v) { for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v + (float)(i & 1023) * 1e-3f; }
typedef void (*Fn)(const f4 *, const f4 *, f4 *);
struct Var { std::string name; Fn fn; int block; std::vector<float> ms; };
int main() {
    const size_t n = (size_t)1 << 28, n_vec = n / 4;
    float *a;
    CK(hipMalloc(&a, 3 * n * 4));
    float *b = a + n, *c = b + n;
    init_k<<<4096, 256>>>(a, n, 1.f); init_k<<<4096, 256>>>(b, n, 2.f); CK(hipDeviceSynchronize());
    std::vector<Var> vs;
    vs.push_back({"builtin nt loads, nt store (the library's kernel), wg1024", k_builtin<1024>, 1024, {}});
#define ADD(K, LNAME) \
    vs.push_back({std::string("loads ") + LNAME + ", store nt,    wg1024", K<1024, 0>, 1024, {}}); \
    vs.push_back({std::string("loads ") + LNAME + ", store sc1,   wg1024", K<1024, 1>, 1024, {}}); \
    vs.push_back({std::string("loads ") + LNAME + ", store nt,    wg256", K<256, 0>, 256, {}});
    ADD(k_plain, "plain     ") ADD(k_nt, "nt        ") ADD(k_sc0, "sc0       ") ADD(k_sc1, "sc1       ") ADD(k_sc0sc1, "sc0 sc1   ")
    ADD(k_sc0nt, "sc0 nt    ") ADD(k_sc1nt, "sc1 nt    ") ADD(k_sc0sc1nt, "sc0 sc1 nt")
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int r = 0; r < 7; ++r)
        for (auto &v : vs) {
            const unsigned grid = (unsigned)(n_vec / v.block);
            v.fn<<<grid, v.block>>>((const f4 *)a, (const f4 *)b, (f4 *)c);
            CK(hipEventRecord(e0));
            for (int k = 0; k < 10; ++k) v.fn<<<grid, v.block>>>((const f4 *)a, (const f4 *)b, (f4 *)c);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); v.ms.push_back(ms / 10);
        }
    for (auto &v : vs) { auto m = v.ms; std::sort(m.begin(), m.end()); printf("%-62s median %8.2f us  %5.1f %%   best %8.2f\n", v.name.c_str(), m[3] * 1e3, 12.0 * n / m[3] * 1e-6 / 80.0, m[0] * 1e3); }
    return 0;
}
