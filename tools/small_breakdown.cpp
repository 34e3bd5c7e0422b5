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
    {
        sm::SMArray<int> arr1d = {1, 2, 3, 4, 5, 6, 7, 8, 9, 10};
        auto warm = sm::pow(arr1d, 3);
        auto warm2 = sm::pow(arr1d, 3);  // the second use uploads the array: resident from here on
        print(run("sm::pow(arr1d, 3), resident (BM_SMArrayPow_1D)", [&] { auto r = sm::pow(arr1d, 3); DoNotOptimize(r); }, sync));
        void *p, *q; smhip_alloc(&p, 40); smhip_alloc(&q, 40);
        const int three = 3;
        print(run("smhip_array_scalar pow i32 n=10", [&] { smhip_array_scalar(SMHIP_OP_POW, SMHIP_I32, p, &three, 10, q); }, sync));
        {   // the recorded path by itself (csrc/tiny.hip): independent results, so they batch
            void *outs[64];
            for (auto &o : outs) smhip_alloc(&o, 40);
            unsigned k = 0;
            print(run("smhip_array_scalar pow i32 n=10 into 64 rotating outputs (record + 1/30 launch)", [&] { smhip_array_scalar(SMHIP_OP_POW, SMHIP_I32, p, &three, 10, outs[k++ & 63]); }, sync));
            print(run("smhip_alloc + smhip_array_scalar + smhip_free (fresh result each time)", [&] { void *t; smhip_alloc(&t, 40); smhip_array_scalar(SMHIP_OP_POW, SMHIP_I32, p, &three, 10, t); smhip_free(t); }, sync));
            for (auto &o : outs) smhip_free(o);
        }
        const int one = 1;
        print(run("smhip_array_scalar add i32 n=10", [&] { smhip_array_scalar(SMHIP_OP_ADD, SMHIP_I32, p, &one, 10, q); }, sync));
        print(run("smhip_fill i32 n=10", [&] { smhip_fill(SMHIP_I32, q, &one, 10); }, sync));
        print(run("DeviceGuard (get + compare)", [&] { sm::hip::DeviceGuard g(0); DoNotOptimize(g); }, sync));
        print(run("SMArray::device_empty(10) + destroy (no device buffer)", [&] { auto e = sm::SMArray<int>::device_empty({10}); DoNotOptimize(e); }, sync));
        print(run("device_empty + device_data_mut (pool alloc + free)", [&] { auto e = sm::SMArray<int>::device_empty({10}); DoNotOptimize(e.device_data_mut()); }, sync));
        {
            static auto one = sm::ones<float>(1000000), two = sm::ones<float>(1000000);
            void *o; smhip_alloc(&o, 4000000);
            const float *pa = one.device_data(), *pb = two.device_data();
            print(run("smhip_contiguous n = 1e6 (raw C call, output preallocated)", [&] { smhip_contiguous(0, 0, pa, pb, o, 1000000); }, sync));
            std::int64_t sh[1] = {1000000}, st[1] = {1};
            print(run("smhip_elementwise n = 1e6 (raw C call)", [&] { smhip_elementwise(0, 0, pa, st, pb, st, sh, 1, o); }, sync));
            print(run("smhip_alloc + smhip_contiguous + smhip_free, n = 1e6", [&] { void *t; smhip_alloc(&t, 4000000); smhip_contiguous(0, 0, pa, pb, t, 1000000); smhip_free(t); }, sync));
            print(run("one.apply<AddOp>(two), n = 1e6 (eager C++ path)", [&] { auto r = one.apply<AddOp<float>>(two); DoNotOptimize(r); }, sync));
            print(run("one + two, n = 1e6 (million_check: recorded, run at the `;`)", [&] { auto r = one + two; DoNotOptimize(r); }, sync));
            print(run("(one + two) * 0.5f, n = 1e6 (one chain launch)", [&] { auto r = (one + two) * 0.5f; DoNotOptimize(r); }, sync));
        }
        print(run("deferred a + a, n = 8192 (chain record + single op)", [] {
            static auto big = sm::ones<float>(8192);
            auto r = big + big;
            DoNotOptimize(r);
        }, sync));
    }
    return 0;
}
