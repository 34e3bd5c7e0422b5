// benchmark/pow.cpp on MI355X: BM_SMArrayPow_1D / _2D / _Large, the bodies of the
// reference's benchmark/pow.cpp:5-49, through the drop-in header.  The float
// _Large cases cannot even link in the reference (PowOp<float>::apply_simd is
// undefined, include/math/pow.h:12-13); here they run the in-register pow kernel.
// `BM_SMArrayPow_Config4` adds BASELINE config 4 (pow(a, 2.5), N = 2^26).
#include <sm.h>

#include "minibench.h"

int main() {
    using namespace minibench;
    auto sync = [] { sm::synchronize(); };
    {
        sm::SMArray<int> w = {1, 2, 3, 4};
        warm_device([&] { auto r = sm::pow(w, 2); DoNotOptimize(r); }, sync);
    }
    header();
    {
        sm::SMArray<int> arr1d = {1, 2, 3, 4, 5, 6, 7, 8, 9, 10};  // benchmark/pow.cpp:5-14
        print(run("BM_SMArrayPow_1D", [&] {
            auto result = sm::pow(arr1d, 3);
            DoNotOptimize(result);
            ClobberMemory();
        }, sync));
    }
    {
        sm::SMArray<int> arr2d = {{1, 2, 3}, {4, 5, 6}, {7, 8, 9}};  // benchmark/pow.cpp:19-28
        print(run("BM_SMArrayPow_2D", [&] {
            auto result = sm::pow(arr2d, 2);
            DoNotOptimize(result);
            ClobberMemory();
        }, sync));
    }
    for (int N : {100, 500, 1000}) {  // benchmark/pow.cpp:33-49, ->Arg(100)->Arg(500)->Arg(1000)->Iterations(1000)
        auto arr = sm::empty<float>(N, N);
        for (int i = 0; i < N; ++i)
            for (int j = 0; j < N; ++j) arr(i, j) = i + j + 1;
        auto r = run("BM_SMArrayPow_Large/" + std::to_string(N), [&] {
            constexpr float exponent = 2;
            auto result = sm::pow(arr, exponent);
            DoNotOptimize(result);
            ClobberMemory();
        }, sync, 1000);
        char extra[64];
        std::snprintf(extra, sizeof extra, "%.2f Gelem/s", double(N) * N / r.ns_per_iter);
        print(r, extra);
    }
    {
        const std::size_t n = std::size_t(1) << 26;
        auto arr = sm::ones<float>(n) * 1.7f;
        auto r = run("BM_SMArrayPow_Config4/2^26", [&] {
            auto result = sm::pow(arr, 2.5f);
            DoNotOptimize(result);
            ClobberMemory();
        }, sync);  // time-based iteration count (~0.5 s): a VALU-heavy kernel needs tens of ms before the clocks have ramped
        char extra[96];
        std::snprintf(extra, sizeof extra, "%.1f Gelem/s  %.0f GB/s (8 B/elem)", n / r.ns_per_iter, 8.0 * n / r.ns_per_iter);
        print(r, extra);
    }
    return 0;
}
