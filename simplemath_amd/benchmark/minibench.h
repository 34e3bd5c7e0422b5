// minibench.h -- the few lines of Google Benchmark this harness needs.
//
// The reference's benchmark/ directory pulls Google Benchmark v1.9.4 from GitHub
// at configure time (cmake/gbenchmark.cmake:5-14); there is no network here, so
// the same benchmark bodies are driven by this loop instead: warm up, pick an
// iteration count that fills ~0.5 s, report wall time per iteration in the
// familiar "name  Time  Iterations" table.  `sync` is called before the clock
// stops so queued GPU work is included.
#pragma once

#include <chrono>
#include <cstdio>
#include <functional>
#include <string>

namespace minibench {

template <typename T>
inline void DoNotOptimize(T const &value) { asm volatile("" : : "r,m"(value) : "memory"); }
inline void ClobberMemory() { asm volatile("" : : : "memory"); }

struct Result { std::string name; double ns_per_iter; long iterations; };

inline Result run(const std::string &name, const std::function<void()> &body, const std::function<void()> &sync,
                  long fixed_iterations = 0, double budget_s = 0.5) {
    using clock = std::chrono::steady_clock;
    body();
    sync();
    long iters = fixed_iterations;
    if (iters == 0) {
        auto t0 = clock::now();
        long probe = 0;
        do { body(); ++probe; } while (std::chrono::duration<double>(clock::now() - t0).count() < 0.05);
        sync();
        const double per = std::chrono::duration<double>(clock::now() - t0).count() / probe;
        iters = static_cast<long>(budget_s / per);
        if (iters < 3) iters = 3;
    }
    auto t0 = clock::now();
    for (long i = 0; i < iters; ++i) body();
    sync();
    const double total = std::chrono::duration<double>(clock::now() - t0).count();
    return {name, total / iters * 1e9, iters};
}

// The first benchmark of a process otherwise pays for the device leaving its idle state: the launch path of a GPU that has
// been idle is ~0.4 us per launch slower for the first second or so (BM_SMArrayPow_1D as the first benchmark 3.24 us, the same
// body after others 2.84 us -- profiles/r04_small_array_breakdown.txt).  Google Benchmark's own iteration search runs a body for
// about that long before the run it reports; this does it once per process.
inline void warm_device(const std::function<void()> &body, const std::function<void()> &sync, double seconds = 1.0) {
    using clock = std::chrono::steady_clock;
    auto t0 = clock::now();
    do {
        for (int i = 0; i < 256; ++i) body();
    } while (std::chrono::duration<double>(clock::now() - t0).count() < seconds);
    sync();
}

inline void header() {
    std::printf("%-34s %15s %12s\n", "Benchmark", "Time", "Iterations");
    std::printf("%s\n", std::string(63, '-').c_str());
}
inline void print(const Result &r, const char *extra = "") {
    std::printf("%-34s %12.0f ns %12ld  %s\n", r.name.c_str(), r.ns_per_iter, r.iterations, extra);
}

}  // namespace minibench
