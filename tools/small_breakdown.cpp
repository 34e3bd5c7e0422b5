// tools/small_breakdown.cpp -- where do simple_check's microseconds go?  (host construction / upload / launch / frees)
#include <sm.h>
#include "../simplemath_amd/benchmark/minibench.h"
int main() {
    using namespace minibench;
    auto sync = [] { sm::synchronize(); };
    header();
    print(run("construct 5x5 from nested lists", [] {
        sm::SMArray<float> ac = {{1, 2, 3, 4, 5}, {1, 2, 3, 4, 5}, {1, 2, 3, 4, 5}, {1, 2, 3, 4, 5}, {1, 2, 3, 4, 5}};
        DoNotOptimize(ac);
    }, sync));
    print(run("construct 25 flat", [] {
        sm::SMArray<float> ac = {1, 2, 3, 4, 5, 1, 2, 3, 4, 5, 1, 2, 3, 4, 5, 1, 2, 3, 4, 5, 1, 2, 3, 4, 5};
        DoNotOptimize(ac);
    }, sync));
    {
        sm::SMArray<float> ac = {{1, 2, 3, 4, 5}, {1, 2, 3, 4, 5}, {1, 2, 3, 4, 5}, {1, 2, 3, 4, 5}, {1, 2, 3, 4, 5}};
        auto warm = ac + ac;
        print(run("ac + ac, ac resident", [&] { auto r = ac + ac; DoNotOptimize(r); }, sync));
        print(run("ac * 2.0f, ac resident", [&] { auto r = ac * 2.0f; DoNotOptimize(r); }, sync));
    }
    print(run("construct flat + first op (upload)", [] {
        sm::SMArray<float> ac = {1, 2, 3, 4, 5, 1, 2, 3, 4, 5, 1, 2, 3, 4, 5, 1, 2, 3, 4, 5, 1, 2, 3, 4, 5};
        auto r = ac * 2.0f;
        DoNotOptimize(r);
    }, sync));
    {
        void *p; smhip_alloc(&p, 100); float h[25] = {};
        print(run("smhip_upload 100 B", [&] { smhip_upload(p, h, 100); }, sync));
        void *q; smhip_alloc(&q, 100);
        print(run("smhip_contiguous n=25", [&] { smhip_contiguous(0, 0, p, p, q, 25); }, sync));
        print(run("smhip_alloc + smhip_free 100 B", [&] { void *t; smhip_alloc(&t, 100); smhip_free(t); }, sync));
        std::int64_t sh[2] = {5, 5}, st[2] = {5, 1};
        print(run("smhip_elementwise 5x5", [&] { smhip_elementwise(0, 0, p, st, p, st, sh, 2, q); }, sync));
    }
    return 0;
}
