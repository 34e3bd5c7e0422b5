// pool_streams.hip -- the pooled allocator's stream-ordered reuse under the cases a per-thread "current stream" tag got
// wrong (smhip.h promises "callable from any thread; device and stream selection are per thread"):
//   1. a block used on thread A's own stream, freed and reallocated by thread B on the library stream;
//   2. one thread switching streams between the last use and the free;
//   3. a caller-owned stream destroyed before the block it was handed to is freed;
//   4. two asynchronous reductions queued from one thread on two streams (their partials must not share a buffer);
//   5. a sharded entry point (library streams) called by a thread that is on its OWN stream: it must see what that stream
//      has queued before it and the stream must see its results.
//   6. the library's TWO queues per device (runtime.hip): independent operators alternate between them, and every
//      dependency that crosses them -- read after write, write after read, write after write, a consumer on another host
//      thread, an operator's pooled scratch -- is an event edge.
//   7. tiny operators recorded on the library's queue (csrc/tiny.hip) against a switch to a caller's stream, a consumer on another
//      host thread, and frees of results nobody read.
//   8. four host threads recording tiny operators on one device at once, a fifth launching large ones.
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

static bool all_equal(const float *dev, float want, size_t count = N) {  // checked on the calling thread's stream
    std::vector<float> h(count);
    if (smhip_download(h.data(), dev, count * sizeof(float)) != SMHIP_OK) return false;
    for (size_t i = 0; i < count; i += 4099)
        if (h[i] != want) { std::printf("  element %zu is %g, expected %g\n", i, h[i], want); return false; }
    return h[count - 1] == want;
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
    std::printf("case 6\n");
    {  // 6. two library queues: x and y are produced by two long, independent chains (one per queue); everything below
       // depends on both, so whichever queue it lands on it needs an event edge to the other
        OK(smhip_set_stream(nullptr));
        const size_t M = 8u << 20;  // 32 MiB of f32 per array: operators small enough for the second queue (runtime.hip: kOverlapMaxBytes)
        int queues = 0;
        unsigned long long alt0 = 0, edges0 = 0, alt1 = 0, edges1 = 0;
        OK(smhip_queue_stats(&queues, &alt0, &edges0));
        void *x = nullptr, *y = nullptr, *z = nullptr, *u = nullptr;
        OK(smhip_alloc(&x, M * sizeof(float)));
        OK(smhip_alloc(&y, M * sizeof(float)));
        OK(smhip_alloc(&z, M * sizeof(float)));
        OK(smhip_alloc(&u, M * sizeof(float)));
        const float one = 1.0f, two = 2.0f;
        for (int rep = 0; rep < 3; ++rep) {
            OK(smhip_fill(SMHIP_F32, x, &zero, M));
            OK(smhip_fill(SMHIP_F32, y, &zero, M));
            OK(smhip_fill(SMHIP_F32, u, &seven, M));
            for (int i = 0; i < 30; ++i) {  // interleaved: each chain stays on the queue of its own last writer
                OK(smhip_array_scalar(SMHIP_OP_ADD, SMHIP_F32, x, &one, M, x));
                OK(smhip_array_scalar(SMHIP_OP_ADD, SMHIP_F32, y, &two, M, y));
                OK(smhip_array_scalar(SMHIP_OP_ADD, SMHIP_F32, u, &one, M, u));
            }
            OK(smhip_contiguous(SMHIP_OP_ADD, SMHIP_F32, x, y, z, M));   // RAW on both chains: 30 + 60
            OK(smhip_fill(SMHIP_F32, x, &seven, M));                     // WAR: z's launch still reads x
            OK(smhip_contiguous(SMHIP_OP_MUL, SMHIP_F32, z, y, u, M));   // WAW on u (37 by then), RAW on z and y: 90 * 60
            bool ok_thread = false;
            void *w = nullptr;
            std::thread consumer([&] {  // another host thread consumes u without any host-side wait in between
                OK(smhip_set_device(0));
                OK(smhip_alloc(&w, M * sizeof(float)));
                OK(smhip_array_scalar(SMHIP_OP_SUB, SMHIP_F32, u, &one, M, w));
                ok_thread = all_equal(static_cast<float *>(w), 5399.0f, M);
                OK(smhip_free(w));
            });
            consumer.join();
            CHECK(ok_thread);
            CHECK(all_equal(static_cast<float *>(z), 90.0f, M));
            CHECK(all_equal(static_cast<float *>(x), 7.0f, M));
            CHECK(all_equal(static_cast<float *>(u), 5400.0f, M));
        }
        // a chain that is cut (a transposed operand) allocates its temporary inside the operator: the bytes come from the pool
        // and may have been another queue's scratch a moment ago
        {
            const int64_t shape[2] = {2048, 2048}, dense[2] = {2048, 1}, turned[2] = {1, 2048};
            const size_t n = 2048u * 2048u;
            const void *ops_[3] = {x, y, nullptr};
            int64_t strides[6] = {dense[0], dense[1], turned[0], turned[1], 0, 0};
            const int opc[2] = {SMHIP_OP_ADD, SMHIP_OP_MUL}, swp[2] = {0, 0};
            const float scal[3] = {0, 0, 3.0f};
            OK(smhip_fill(SMHIP_F32, x, &one, M));
            OK(smhip_fill(SMHIP_F32, y, &two, M));
            for (int rep = 0; rep < 8; ++rep) {
                OK(smhip_chain(SMHIP_F32, 3, ops_, strides, scal, opc, swp, shape, 2, rep % 2 ? z : u));  // (x + y.T) * 3
                OK(smhip_array_scalar(SMHIP_OP_ADD, SMHIP_F32, rep % 2 ? u : z, &one, n, rep % 2 ? u : z));  // something independent in between
            }
            std::vector<float> h(n);
            OK(smhip_download(h.data(), z, n * sizeof(float)));
            bool nine = true;
            for (size_t i = 0; i < n; i += 997) nine = nine && h[i] == 9.0f;
            CHECK(nine);
        }
        OK(smhip_queue_stats(&queues, &alt1, &edges1));
        std::printf("  queues %d, alternations %llu, event edges %llu\n", queues, alt1 - alt0, edges1 - edges0);
        if (queues == 2) CHECK(alt1 > alt0 && edges1 > edges0);
        OK(smhip_free(x));
        OK(smhip_free(y));
        OK(smhip_free(z));
        OK(smhip_free(u));
    }
    std::printf("case 7\n");
    {  // 7. operators on tiny arrays are RECORDED on the library's queue and launched together later (csrc/tiny.hip): a thread that
       //    switches to its own stream must find them launched -- its stream is ordered behind the library's at the switch --
       //    and one that consumes a recorded result on another host thread, or frees it, must see call order.
        const size_t T = 25;
        const float one = 1.0f, two = 2.0f;
        void *a = nullptr, *b = nullptr, *c = nullptr, *d = nullptr;
        OK(smhip_alloc(&a, T * sizeof(float)));
        OK(smhip_alloc(&b, T * sizeof(float)));
        OK(smhip_alloc(&c, T * sizeof(float)));
        OK(smhip_alloc(&d, T * sizeof(float)));
        OK(smhip_fill(SMHIP_F32, a, &seven, T));
        OK(smhip_synchronize());  // (the fill of such an array is recorded too)
        unsigned long long l0 = 0, o0 = 0, l1 = 0, o1 = 0;
        OK(smhip_tiny_stats(&l0, &o0));
        OK(smhip_array_scalar(SMHIP_OP_ADD, SMHIP_F32, a, &one, T, b));   // recorded: b = 8
        OK(smhip_array_scalar(SMHIP_OP_MUL, SMHIP_F32, a, &two, T, c));   // recorded beside it: c = 14
        OK(smhip_set_stream(s1));                                         // the switch launches them on the library's queue, s1 waits for it
        OK(smhip_contiguous(SMHIP_OP_ADD, SMHIP_F32, b, c, d, T));        // on s1, at once: d = 22
        CHECK(all_equal(static_cast<float *>(d), 22.0f, T));
        OK(smhip_set_stream(nullptr));
        OK(smhip_tiny_stats(&l1, &o1));
        std::printf("  tiny operators recorded %llu, launches %llu\n", o1 - o0, l1 - l0);
        {
            const char *e = getenv("SMHIP_TINY_BATCH");
            if (!(e && *e && atoi(e) == 0)) CHECK(o1 - o0 == 2 && l1 - l0 == 1);  // (switched off: nothing is recorded, everything else must hold all the same)
        }
        // a recorded result consumed by another host thread, then freed by it while the producer records more
        OK(smhip_array_scalar(SMHIP_OP_SUB, SMHIP_F32, a, &one, T, b));   // recorded: b = 6
        bool ok_thread = false;
        std::thread consumer([&] {
            OK(smhip_set_device(0));
            void *w = nullptr;
            OK(smhip_alloc(&w, T * sizeof(float)));
            OK(smhip_contiguous(SMHIP_OP_MUL, SMHIP_F32, b, b, w, T));    // reads what the other thread recorded: 36
            ok_thread = all_equal(static_cast<float *>(w), 36.0f, T);
            OK(smhip_free(w));
            OK(smhip_free(b));
        });
        consumer.join();
        CHECK(ok_thread);
        for (int i = 0; i < 100; ++i) {  // results that die unread, as in the benchmark bodies: their blocks come back after the launch
            void *r = nullptr;
            OK(smhip_alloc(&r, T * sizeof(float)));
            OK(smhip_array_scalar(SMHIP_OP_ADD, SMHIP_F32, a, &one, T, r));
            OK(smhip_free(r));
        }
        OK(smhip_array_scalar(SMHIP_OP_ADD, SMHIP_F32, a, &two, T, c));
        CHECK(all_equal(static_cast<float *>(c), 9.0f, T));
        OK(smhip_free(a));
        OK(smhip_free(c));
        OK(smhip_free(d));
    }
    std::printf("case 8\n");
    {  // 8. four host threads record tiny operators on the SAME device at once -- each a running sum in its own buffers (dependent:
       //    one list), fresh results that die unread, reads of one shared input, a read-back every few hundred operators -- while
       //    a fifth keeps launching a large operator: the per-device block, its deferred frees and the flushes from every
       //    thread's calls must neither lose nor reorder anything.
        const size_t T = 40;
        const float one = 1.0f;
        void *shared = nullptr, *big = nullptr;
        OK(smhip_alloc(&shared, T * sizeof(float)));
        OK(smhip_alloc(&big, (8u << 20) * sizeof(float)));
        OK(smhip_fill(SMHIP_F32, shared, &one, T));
        OK(smhip_fill(SMHIP_F32, big, &zero, 8u << 20));
        OK(smhip_synchronize());
        std::vector<std::thread> workers;
        std::vector<int> bad(4, 0);
        for (int w = 0; w < 4; ++w) {
            workers.emplace_back([&, w] {
                if (smhip_set_device(0) != SMHIP_OK) { bad[w] = 1; return; }
                void *acc = nullptr, *tmp = nullptr;
                if (smhip_alloc(&acc, T * sizeof(float)) != SMHIP_OK || smhip_alloc(&tmp, T * sizeof(float)) != SMHIP_OK) { bad[w] = 1; return; }
                const float start = (float)w;
                if (smhip_fill(SMHIP_F32, acc, &start, T) != SMHIP_OK) bad[w] = 1;
                float expect = start;
                for (int i = 1; i <= 3000 && !bad[w]; ++i) {
                    if (smhip_contiguous(SMHIP_OP_ADD, SMHIP_F32, acc, shared, tmp, T) != SMHIP_OK) bad[w] = 1;   // tmp = acc + 1
                    if (smhip_array_scalar(SMHIP_OP_MUL, SMHIP_F32, tmp, &one, T, acc) != SMHIP_OK) bad[w] = 1;     // acc = tmp
                    expect += 1.0f;
                    void *r = nullptr;  // a result nobody reads
                    if (smhip_alloc(&r, T * sizeof(float)) != SMHIP_OK || smhip_array_scalar(SMHIP_OP_ADD, SMHIP_F32, shared, &one, T, r) != SMHIP_OK ||
                        smhip_free(r) != SMHIP_OK) bad[w] = 1;
                    if (i % 377 == 0 && !all_equal(static_cast<float *>(acc), expect, T)) bad[w] = 2;
                }
                if (!bad[w] && !all_equal(static_cast<float *>(acc), expect, T)) bad[w] = 3;
                smhip_free(acc);
                smhip_free(tmp);
            });
        }
        workers.emplace_back([&] {
            if (smhip_set_device(0) != SMHIP_OK) return;
            double total = 0;
            for (int i = 0; i < 200; ++i) {
                smhip_array_scalar(SMHIP_OP_ADD, SMHIP_F32, big, &one, 8u << 20, big);
                // a reduction allocates and frees its partial sums INSIDE its operator scope, dispatcher held: a flush of recorded
                // operators on another thread waits for that scope, so the frees in it must not wait for the flush
                if (smhip_sum(SMHIP_F32, big, 8u << 20, &total) != SMHIP_OK || total != (double)(i + 1) * (double)(8u << 20)) { std::printf("  sum %d: %g\n", i, total); break; }
            }
        });
        for (auto &t : workers) t.join();
        for (int w = 0; w < 4; ++w) {
            if (bad[w]) std::printf("  worker %d: failure kind %d\n", w, bad[w]);
            CHECK(bad[w] == 0);
        }
        CHECK(all_equal(static_cast<float *>(big), 200.0f, 8u << 20));
        CHECK(all_equal(static_cast<float *>(shared), 1.0f, T));
        OK(smhip_free(shared));
        OK(smhip_free(big));
    }
    OK(smhip_synchronize());
    hipStreamDestroy(s1);
    hipStreamDestroy(s2);
    std::printf("pool_streams: %d failures\n", g_failures);
    return g_failures ? 1 : 0;
}
