// tools/ubench_valu.hip -- issue rates of the VALU ops the pow / reduction kernels lean on.
// Development tool.  hipcc -O3 --offload-arch=gfx950 -o tools/bin/ubench_valu tools/ubench_valu.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int ITERS = 4096;
constexpr int CH = 8;  // independent chains per lane

template <int OP>
__global__ __launch_bounds__(256) void k(double* out, double seed, float fseed) {
    double a[CH]; float f[CH]; int iv[CH];
    for (int c = 0; c < CH; ++c) { a[c] = seed + c + threadIdx.x * 1e-9; f[c] = fseed + c; iv[c] = threadIdx.x + c; }
    const double m = 1.0000001, b = 1e-9;
    for (int i = 0; i < ITERS; ++i) {
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            if constexpr (OP == 0) a[c] = __builtin_fma(a[c], m, b);                 // v_fma_f64
            else if constexpr (OP == 1) a[c] = a[c] + b;                               // v_add_f64
            else if constexpr (OP == 2) a[c] = a[c] * m;                               // v_mul_f64
            else if constexpr (OP == 3) a[c] = __builtin_amdgcn_rcp(a[c]);             // v_rcp_f64
            else if constexpr (OP == 4) { f[c] = (float)a[c]; a[c] = (double)f[c] + b; } // cvt both ways + add
            else if constexpr (OP == 5) f[c] = __builtin_fmaf(f[c], 1.0000001f, 1e-9f); // v_fma_f32
            else if constexpr (OP == 6) a[c] = __builtin_rint(a[c] * m);               // mul + rndne
            else if constexpr (OP == 7) { iv[c] = (int)a[c]; a[c] = (double)iv[c] + b; } // cvt_i32_f64 + cvt_f64_i32 + add
            else if constexpr (OP == 8) iv[c] = iv[c] * 3 + 1;                          // v_mad_u32 / mul_lo
            else if constexpr (OP == 9) iv[c] = (iv[c] ^ (iv[c] >> 3)) + 1;             // shifts / xor / add
        }
    }
    double s = 0;
    for (int c = 0; c < CH; ++c) s += a[c] + f[c] + iv[c];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int OP> void run(const char* name, int ops_per_iter, double* out) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int grid = 256 * 8;  // 8 blocks of 4 waves per CU -> 8 waves per SIMD
    k<OP><<<grid, 256>>>(out, 1.0, 1.0f); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); for (int r = 0; r < 5; ++r) k<OP><<<grid, 256>>>(out, 1.0, 1.0f); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
    double lane_ops = (double)grid * 256 * ITERS * CH * ops_per_iter;
    double wave_instr_per_simd = lane_ops / 64 / 1024;
    printf("%-28s %8.3f ms  %7.2f Tlane-op/s  => %5.2f cycles/wave-instr/SIMD @2.4GHz (%5.2f @ 2.0)\n", name, ms, lane_ops / ms * 1e-9,
           ms * 1e-3 * 2.4e9 / wave_instr_per_simd, ms * 1e-3 * 2.0e9 / wave_instr_per_simd);
}

int main() {
    double* out; CK(hipMalloc(&out, 256 * 8 * 256 * 8));
    run<0>("v_fma_f64", 1, out);
    run<1>("v_add_f64", 1, out);
    run<2>("v_mul_f64", 1, out);
    run<3>("v_rcp_f64", 1, out);
    run<4>("cvt_f32_f64+cvt_f64_f32+add", 3, out);
    run<5>("v_fma_f32", 1, out);
    run<6>("v_mul_f64+v_rndne_f64", 2, out);
    run<7>("cvt_i32_f64+cvt_f64_i32+add", 3, out);
    run<8>("int mad", 1, out);
    run<9>("int shift/xor/add (3 ops)", 3, out);
    return 0;
}
