// pool_streams.hip -- the pooled allocator's stream-ordered reuse under the cases a per-thread "current stream" tag got
// wrong (smhip.h promises "callable from any thread; device and stream selection are per thread"):
//   1. a block used on thread A's own stream, freed and reallocated by thread B on the library stream;
//   2. one thread switching streams between the last use and the free;
//   3. a caller-owned stream destroyed before the block it was handed to is freed;
//   4. two asynchronous reductions queued from one thread on two streams (their partials must not share a buffer);
//   5. a sharded entry point (library streams) called by a thread that is on its OWN stream: it must see what that stream
//      has queued before it and the stream must see its results.
// The reference has nothing like this (new[]/delete[] per operator, SMArray.h:219,342-346).  Exit code 0 = all held.
#include <hip/hip_runtime.h>
#include <smhip.h>

#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

static int g_failures = 0;
#define CHECK(cond)                                                            \
    do {                                                                       \
        if (!(cond)) {                                                         \
            ++g_failures;                                                      \
            std::printf("FAIL %s:%d  %s   [%s]\n", __FILE__, __LINE__, #cond, smhip_last_error()); \
        }                                                                      \
    } while (0)
#define OK(call) CHECK((call) == SMHIP_OK)

static const size_t N = 64u << 20;  // 256 MiB of f32: one pass takes ~85 us, 60 passes keep a stream busy for ~5 ms

static void busy(float *x, int passes) {  // x += 1, `passes` times, on the calling thread's stream
    const float one = 1.0f;
    for (int i = 0; i < passes; ++i) OK(smhip_array_scalar(SMHIP_OP_ADD, SMHIP_F32, x, &one, N, x));
}

static bool all_equal(const float *dev, float want) {  // checked on the calling thread's stream
    std::vector<float> h(N);
    if (smhip_download(h.data(), dev, N * sizeof(float)) != SMHIP_OK) return false;
    for (size_t i = 0; i < N; i += 4099)
        if (h[i] != want) { std::printf("  element %zu is %g, expected %g\n", i, h[i], want); return false; }
    return h[N - 1] == want;
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    OK(smhip_set_device(0));
    hipStream_t s1, s2;
    if (hipStreamCreateWithFlags(&s1, hipStreamNonBlocking) != hipSuccess || hipStreamCreateWithFlags(&s2, hipStreamNonBlocking) != hipSuccess) {
        std::printf("cannot create streams\n");
        return 2;
    }
    const float zero = 0.0f, seven = 7.0f;

    std::printf("case 1\n");
    {  // 1. used on A's stream, freed + reallocated + written by B on the library stream while A's kernels are still queued
        void *x = nullptr;
        OK(smhip_set_stream(s1));
        OK(smhip_alloc(&x, N * sizeof(float)));
        OK(smhip_fill(SMHIP_F32, x, &zero, N));
        busy(static_cast<float *>(x), 60);
        void *y = nullptr;
        bool ok = false;
        std::thread b([&] {
            OK(smhip_set_device(0));
            OK(smhip_free(x));                        // another thread, library stream
            OK(smhip_alloc(&y, N * sizeof(float)));   // first fit: the same bytes
            OK(smhip_fill(SMHIP_F32, y, &seven, N));
            ok = all_equal(static_cast<float *>(y), 7.0f);
        });
        b.join();
        CHECK(y == x);  // otherwise the case was not exercised
        CHECK(ok);
        OK(smhip_synchronize());  // s1
        OK(smhip_set_stream(nullptr));
        CHECK(all_equal(static_cast<float *>(y), 7.0f));  // A's late kernels did not land on B's data
        OK(smhip_free(y));
    }
    std::printf("case 2\n");
    {  // 2. one thread: use on s1, switch to the library stream, free, reallocate, write
        void *x = nullptr, *y = nullptr;
        OK(smhip_set_stream(s1));
        OK(smhip_alloc(&x, N * sizeof(float)));
        OK(smhip_fill(SMHIP_F32, x, &zero, N));
        busy(static_cast<float *>(x), 60);
        OK(smhip_set_stream(nullptr));
        OK(smhip_free(x));
        OK(smhip_alloc(&y, N * sizeof(float)));
        CHECK(y == x);
        OK(smhip_fill(SMHIP_F32, y, &seven, N));
        CHECK(all_equal(static_cast<float *>(y), 7.0f));
        CHECK(hipStreamSynchronize(s1) == hipSuccess);
        CHECK(all_equal(static_cast<float *>(y), 7.0f));
        // ... and the other way round: allocated and used on the library stream, freed under s2
        busy(static_cast<float *>(y), 60);  // -> 67 on the library stream
        OK(smhip_set_stream(s2));
        OK(smhip_free(y));
        void *z = nullptr;
        OK(smhip_alloc(&z, N * sizeof(float)));
        CHECK(z == y);
        OK(smhip_fill(SMHIP_F32, z, &seven, N));
        CHECK(all_equal(static_cast<float *>(z), 7.0f));
        OK(smhip_set_stream(nullptr));
        OK(smhip_synchronize());
        CHECK(all_equal(static_cast<float *>(z), 7.0f));
        OK(smhip_free(z));
    }
    std::printf("case 3\n");
    {  // 3. the stream a block was handed to is destroyed before the block is freed
        hipStream_t s3;
        CHECK(hipStreamCreateWithFlags(&s3, hipStreamNonBlocking) == hipSuccess);
        void *x = nullptr, *y = nullptr;
        OK(smhip_set_stream(s3));
        OK(smhip_alloc(&x, N * sizeof(float)));
        OK(smhip_fill(SMHIP_F32, x, &zero, N));
        busy(static_cast<float *>(x), 10);
        OK(smhip_synchronize());
        OK(smhip_set_stream(nullptr));
        CHECK(hipStreamDestroy(s3) == hipSuccess);
        std::thread other([&] {
            OK(smhip_set_device(0));
            OK(smhip_free(x));  // its stream is gone AND this is another thread: must neither touch the dead handle, fail, nor leak
        });
        other.join();
        OK(smhip_alloc(&y, N * sizeof(float)));
        OK(smhip_fill(SMHIP_F32, y, &seven, N));
        CHECK(all_equal(static_cast<float *>(y), 7.0f));
        OK(smhip_free(y));
        size_t in_use = 1, cached = 0;
        OK(smhip_pool_stats(&in_use, &cached));
        CHECK(in_use == 0);
    }
    std::printf("case 4\n");
    {  // 4. two async reductions from one thread on two streams
        void *a = nullptr, *b = nullptr, *r = nullptr;
        OK(smhip_alloc(&a, N * sizeof(float)));
        OK(smhip_alloc(&b, N * sizeof(float)));
        OK(smhip_alloc(&r, 2 * sizeof(double)));
        const float two = 2.0f, three = 3.0f;
        OK(smhip_fill(SMHIP_F32, a, &two, N));
        OK(smhip_fill(SMHIP_F32, b, &three, N));
        OK(smhip_synchronize());
        for (int rep = 0; rep < 20; ++rep) {
            OK(smhip_set_stream(s1));
            OK(smhip_sum_async(SMHIP_F32, a, N, static_cast<double *>(r)));
            OK(smhip_set_stream(s2));
            OK(smhip_sum_async(SMHIP_F32, b, N, static_cast<double *>(r) + 1));
            CHECK(hipStreamSynchronize(s1) == hipSuccess);
            CHECK(hipStreamSynchronize(s2) == hipSuccess);
            double h[2] = {0, 0};
            OK(smhip_download(h, r, sizeof h));
            CHECK(h[0] == 2.0 * (double)N && h[1] == 3.0 * (double)N);
        }
        OK(smhip_set_stream(nullptr));
        OK(smhip_free(a));
        OK(smhip_free(b));
        OK(smhip_free(r));
    }
    std::printf("case 5\n");
    {  // 5. producer on the caller's stream -> sharded kernel on the library stream -> consumer on the caller's stream
        OK(smhip_set_stream(nullptr));
        OK(smhip_set_devices(1));
        OK(smhip_set_stream(s1));
        void *x = nullptr, *y = nullptr;
        OK(smhip_alloc(&x, N * sizeof(float)));
        OK(smhip_alloc(&y, N * sizeof(float)));
        OK(smhip_fill(SMHIP_F32, x, &zero, N));
        busy(static_cast<float *>(x), 40);  // ~3 ms of x += 1 queued on s1
        const void *pa[1] = {x};
        void *po[1] = {y};
        const size_t pn[1] = {N};
        const float two = 2.0f;
        OK(smhip_sharded_array_scalar(SMHIP_OP_MUL, SMHIP_F32, pa, &two, pn, po));  // y = 2 x on the device's library stream
        busy(static_cast<float *>(y), 1);                                            // y += 1 back on s1
        CHECK(all_equal(static_cast<float *>(y), 81.0f));                            // (0 + 40) * 2 + 1
        OK(smhip_free(x));
        OK(smhip_free(y));
        OK(smhip_set_stream(nullptr));
        OK(smhip_set_devices(0));
    }
    OK(smhip_synchronize());
    hipStreamDestroy(s1);
    hipStreamDestroy(s2);
    std::printf("pool_streams: %d failures\n", g_failures);
    return g_failures ? 1 : 0;
}
