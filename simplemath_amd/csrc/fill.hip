// fill.hip -- device-side array creation.
//
// sm::ones / sm::zeros (reference include/UserFunctions.h:18-40) fill on the
// host with std::fill_n; here the array is born in HBM.  Also the synthetic
// input generator of the benchmark (inputs are produced on device
// by a counter-based hash so no multi-GiB host-to-device copy is needed),
// bit-identical to the CPU checker's generator (same hash, same fmaf).
#include "internal.h"
#include "ops.hip.h"

namespace smhip {
namespace {

using namespace dev;

template <typename T>
__global__ __launch_bounds__(256) void fill_vec_kernel(T *__restrict__ dst, T v, size_t n_vec, size_t n, int pol) {
    typedef typename VecTraits<T>::vec_t V;
    constexpr int W = VecTraits<T>::width;
    V vv;
#pragma unroll
    for (int k = 0; k < W; ++k) vv[k] = v;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n_vec) store_stream_if(T, reinterpret_cast<V *>(dst) + i, vv, pol);  // a 40-256 MiB array is born in the Infinity Cache
    else if (i == n_vec)
        for (size_t k = n_vec * W; k < n; ++k) dst[k] = v;
}

__device__ __forceinline__ uint64_t mix64(uint64_t x) {  // splitmix64 finaliser
    x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ULL;
    x ^= x >> 27; x *= 0x94D049BB133111EBULL;
    x ^= x >> 31;
    return x;
}

__global__ __launch_bounds__(256) void uniform_f32_kernel(float *__restrict__ dst, size_t n, uint64_t seed_mul, uint64_t first,
                                                          float lo, float span) {
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        const uint64_t h = mix64(first + i + seed_mul);
        const float u = (float)(h >> 40) * 0x1.0p-24f;
        dst[i] = __builtin_fmaf(u, span, lo);
    }
}

template <typename T>
int run_fill(void *dst, const void *value_host, size_t n, hipStream_t s) {
    constexpr int W = VecTraits<T>::width;
    const T v = *static_cast<const T *>(value_host);
    T *p = static_cast<T *>(dst);
    const size_t n_vec = n / W;
    const size_t g = (n_vec + 1 + 255) / 256;
    if (g > 0x7fffffffu) return fail(SMHIP_ERR_UNSUPPORTED, "fill too large");
    hipLaunchKernelGGL(fill_vec_kernel<T>, dim3((unsigned)g), dim3(256), 0, s, p, v, n_vec, n, stream_policy({}, {p, n * sizeof(T)}));
    SMHIP_LAUNCH_CHECK("fill");
    return SMHIP_OK;
}

}  // namespace

int launch_fill(int dtype, void *dst, const void *value_host, size_t n, hipStream_t s) {
    if (n == 0) return SMHIP_OK;
    switch (dtype) {
        case SMHIP_F32: return run_fill<float>(dst, value_host, n, s);
        case SMHIP_F64: return run_fill<double>(dst, value_host, n, s);
        case SMHIP_I32: return run_fill<int32_t>(dst, value_host, n, s);
        case SMHIP_I64: return run_fill<int64_t>(dst, value_host, n, s);
    }
    return fail(SMHIP_ERR_INVALID, "fill: bad dtype %d", dtype);
}

int launch_fill_uniform_f32(float *dst, size_t n, uint64_t seed, uint64_t first, float lo, float hi, hipStream_t s) {
    if (n == 0) return SMHIP_OK;
    const size_t g = (n + 255) / 256;
    hipLaunchKernelGGL(uniform_f32_kernel, dim3((unsigned)(g < 16384 ? g : 16384)), dim3(256), 0, s, dst, n,
                       seed * 0x9E3779B97F4A7C15ULL, first, lo, hi - lo);
    (void)stream_policy({}, {dst, n * sizeof(float)});  // on record as just written (internal.h: residency)
    SMHIP_LAUNCH_CHECK("fill_uniform_f32");
    return SMHIP_OK;
}

}  // namespace smhip
