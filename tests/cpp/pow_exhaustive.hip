// tests/cpp/pow_exhaustive.hip -- pow(x, 2.5f) through libsmhip for EVERY positive finite float (2^31 - 2^23 values, denormals
// included) against x*x*sqrt(x) evaluated in fp64 (relative error < 4 * 2^-53: enough to place the f32 result to within a
// 2^-29 ULP, so "0 ULP" below means correctly rounded except possibly at a near-tie).  Prints the ULP-error histogram.
// Built by simplemath_amd/build.py (hipcc) into simplemath_amd/bin/pow_exhaustive; run by tests/test_gpu_cpp.py.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "smhip.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
#define SK(x) do { if ((x) < 0) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, smhip_last_error()); exit(1); } } while (0)

__global__ void fill_bits(float* p, unsigned first, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = __uint_as_float(first + (unsigned)i);
}
__global__ void check(const float* x, const float* r, size_t n, float y, unsigned long long* hist, unsigned* worst_bits) {
    unsigned long long h0 = 0, h1 = 0, h2 = 0, h3 = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const double xd = (double)x[i];
        const double ref = y == 2.5f ? xd * xd * sqrt(xd) : (y == 1.5f ? xd * sqrt(xd) : sqrt(sqrt(xd)) * xd * xd * xd);  // 2.5, 1.5, 3.25
        const float rf = (float)ref;  // correctly rounded (up to near-ties)
        const int a = (int)__float_as_uint(r[i]), b = (int)__float_as_uint(rf);  // positive floats: bit patterns are ordered
        const int d = a > b ? a - b : b - a;
        if (d == 0) ++h0; else if (d == 1) ++h1; else if (d == 2) ++h2; else { ++h3; atomicMax(worst_bits, __float_as_uint(x[i])); }
    }
    atomicAdd(&hist[0], h0); atomicAdd(&hist[1], h1); atomicAdd(&hist[2], h2); atomicAdd(&hist[3], h3);
}
int main(int argc, char** argv) {
    const float y = argc > 1 ? (float)atof(argv[1]) : 2.5f;
    const unsigned first = 1u, last = 0x7f7fffffu;  // smallest denormal .. largest finite
    const size_t chunk = (size_t)1 << 28;
    void *x, *r; SK(smhip_alloc(&x, chunk * 4)); SK(smhip_alloc(&r, chunk * 4));
    unsigned long long* hist; unsigned* worst; CK(hipMalloc(&hist, 32)); CK(hipMalloc(&worst, 4)); CK(hipMemset(hist, 0, 32)); CK(hipMemset(worst, 0, 4));
    for (size_t lo = first; lo <= last; lo += chunk) {
        const size_t n = (last - lo + 1) < chunk ? (last - lo + 1) : chunk;
        fill_bits<<<4096, 256>>>((float*)x, (unsigned)lo, n);
        CK(hipDeviceSynchronize());
        SK(smhip_array_scalar(SMHIP_OP_POW, SMHIP_F32, x, &y, n, r));
        SK(smhip_synchronize());
        check<<<4096, 256>>>((const float*)x, (const float*)r, n, y, hist, worst);
        CK(hipDeviceSynchronize());
    }
    unsigned long long h[4]; unsigned w; CK(hipMemcpy(h, hist, 32, hipMemcpyDeviceToHost)); CK(hipMemcpy(&w, worst, 4, hipMemcpyDeviceToHost));
    const double tot = (double)(h[0] + h[1] + h[2] + h[3]);
    printf("pow(x, %g) over all %llu positive finite floats: 0 ULP %llu (%.4f %%)  1 ULP %llu (%.4f %%)  2 ULP %llu  >2 ULP %llu%s\n", y,
           h[0] + h[1] + h[2] + h[3], h[0], 100.0 * h[0] / tot, h[1], 100.0 * h[1] / tot, h[2], h[3], h[3] ? "  (worst x bits above)" : "");
    if (h[3]) printf("largest x with > 2 ULP: 0x%08x\n", w);
    return h[3] ? 1 : 0;
}
