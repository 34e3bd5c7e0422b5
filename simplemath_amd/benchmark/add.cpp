// benchmark/add.cpp on MI355X: simple_check and million_check, the bodies of the
// reference's benchmark/add.cpp:4-29, through the drop-in header.  As in the
// reference the timed body is the operator call plus the result's destruction;
// here the result is born in (and returned to) libsmhip's device pool, and the
// clock stops after the stream has drained.  `large_check` adds BASELINE config
// 2's size (N = 2^28) with the achieved HBM rate.
#include <sm.h>

#include <string>
#include <vector>

#include "minibench.h"

int main() {
    using namespace minibench;
    auto sync = [] { sm::synchronize(); };
    {
        sm::SMArray<float> w = {1, 2, 3, 4};
        warm_device([&] { auto r = w + w; DoNotOptimize(r); }, sync);
    }
    header();

    print(run("simple_check", [] {  // benchmark/add.cpp:4-19: build a 5x5 array and add it to itself
        sm::SMArray<float> ac = {{1, 2, 3, 4, 5}, {1, 2, 3, 4, 5}, {1, 2, 3, 4, 5}, {1, 2, 3, 4, 5}, {1, 2, 3, 4, 5}};
        auto result = ac + ac;
        DoNotOptimize(result);
        ClobberMemory();
    }, sync));

    {
        const sm::SMArray<float> one = sm::ones<float>(1'000'000);  // benchmark/add.cpp:21-29
        const sm::SMArray<float> two = sm::ones<float>(1'000'000);
        auto r = run("million_check", [&] {
            auto result = one + two;
            DoNotOptimize(result);
            ClobberMemory();
        }, sync);
        char extra[96];
        std::snprintf(extra, sizeof extra, "%.1f Gelem/s", 1e6 / r.ns_per_iter);
        print(r, extra);
    }
    {
        const std::size_t n = std::size_t(1) << 28;
        const sm::SMArray<float> one = sm::ones<float>(n);
        const sm::SMArray<float> two = sm::ones<float>(n);
        auto r = run("large_check/2^28", [&] {
            auto result = one + two;
            DoNotOptimize(result);
            ClobberMemory();
        }, sync, 50);
        char extra[96];
        std::snprintf(extra, sizeof extra, "%.1f Gelem/s  %.0f GB/s (12 B/elem)", n / r.ns_per_iter, 12.0 * n / r.ns_per_iter);
        print(r, extra);
    }
    {
        // An operator CHAIN on resident arrays -- every operator reads what the previous one wrote (temporaries come from
        // and go back to the device pool): (A * row + B) * 0.5f on 4096 x 4096, 28 bytes per element over the three
        // launches.  SMHIP_STORE_POLICY=nt in the environment shows it without the write-side policy.
        // 2048 x 4096: the five arrays alive at a time (160 MiB) fit the Infinity Cache; 4096 x 4096: they do not (320 MiB).
        for (const std::size_t rows : {std::size_t(2048), std::size_t(4096)}) {
            const std::size_t cols = 4096, n = rows * cols;
            const sm::SMArray<float> A = sm::ones<float>(rows, cols), B = sm::ones<float>(rows, cols), row = sm::ones<float>(1, cols);
            auto r = run("chain_check/" + std::to_string(rows) + "x4096", [&] {
                auto t1 = A * row;
                auto t2 = t1 + B;
                auto t3 = t2 * 0.5f;
                DoNotOptimize(t3);
                ClobberMemory();
            }, sync, 200);
            char extra[112];
            std::snprintf(extra, sizeof extra, "%.1f Gelem/s  %.0f GB/s (28 B/elem over 3 launches)", n / r.ns_per_iter, 28.0 * n / r.ns_per_iter);
            print(r, extra);
            // The same chain written as ONE expression: the temporaries are never written -- one kernel (smhip_chain),
            // 12 bytes per element (A, B and the result; the 16 KiB row comes from the caches).
            const auto before = sm::fusion_stats();
            auto rf = run("chain_expr_check/" + std::to_string(rows) + "x4096", [&] {
                auto t3 = (A * row + B) * 0.5f;
                DoNotOptimize(t3);
                ClobberMemory();
            }, sync, 200);
            const auto after = sm::fusion_stats();
            std::snprintf(extra, sizeof extra, "%.1f Gelem/s  %.0f GB/s = %.1f %% of 8 TB/s (12 B/elem, %s)", n / rf.ns_per_iter, 12.0 * n / rf.ns_per_iter,
                          12.0 * n / rf.ns_per_iter / 80.0, after.fused_stages - before.fused_stages == 3 * (after.chains - before.chains) ? "1 launch" : "NOT fused");
            print(rf, extra);
            // ... and ASSIGNED to an existing array: the expression is evaluated straight into C (no temporary, no copy)
            sm::SMArray<float> Cdst = sm::zeros<float>(rows, cols);
            auto ra = run("chain_assign_check/" + std::to_string(rows) + "x4096", [&] {
                Cdst = (A * row + B) * 0.5f;
                ClobberMemory();
            }, sync, 200);
            std::snprintf(extra, sizeof extra, "%.1f Gelem/s  %.0f GB/s = %.1f %% of 8 TB/s (12 B/elem; with a temporary and a copy: 20)", n / ra.ns_per_iter,
                          12.0 * n / ra.ns_per_iter, 12.0 * n / ra.ns_per_iter / 80.0);
            print(ra, extra);
        }
    }
    {
        // The host-pointer loop template (include/math/calculate.h), i.e. the boundary handing over HOST buffers:
        // upload a and b, run the kernel, download the result -- the PCIe-inclusive rate DESIGN.md quotes.
        const std::size_t n = std::size_t(1) << 24;
        std::vector<float> a(n, 1.0f), b(n, 2.0f), r(n);
        auto res = run("host_pointer_check/2^24", [&] {
            handle_contiguous_arrays<float, AddOp<float>>(a.data(), b.data(), r.data(), n);
            DoNotOptimize(r[n / 2]);
            ClobberMemory();
        }, sync, 10);
        char extra[96];
        std::snprintf(extra, sizeof extra, "%.2f Gelem/s  %.1f GB/s over PCIe (12 B/elem)", n / res.ns_per_iter, 12.0 * n / res.ns_per_iter);
        print(res, extra);
    }
    return 0;
}
