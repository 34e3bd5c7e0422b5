// Operands resident on different GPUs must be refused, not launched on the calling thread's device: needs no
// second GPU and no GPU at all -- the check is host-side and comes before any device call.  Storage::device is what
// sm::Sharded<T>::part(g) hands out for GPU g; here it (and a stand-in device buffer, so that the arrays count as resident)
// is set by hand.  An array that exists only in host memory belongs to no GPU yet and adopts its partner's: two
// host-born arrays stamped with different devices -- one built inside a per-GPU loop, a constant built outside it -- are
// NOT refused.
#include <cstdio>
#include <stdexcept>
#include <string>

#include "sm.h"

template <typename F>
static int expect_mismatch(const char *what, F &&f) {
    try {
        f();
    } catch (const std::runtime_error &e) {
        const std::string msg = e.what();
        if (msg.find("different GPUs (0 and 1)") != std::string::npos || msg.find("different GPUs (1 and 0)") != std::string::npos) return 0;
        std::printf("FAIL %s: threw \"%s\"\n", what, e.what());
        return 1;
    }
    std::printf("FAIL %s: did not throw\n", what);
    return 1;
}

int main() {
    sm::SMArray<float> a = {1, 2, 3, 4}, b = {5, 6, 7, 8}, c = {1, 1, 1, 1};
    sm::SMArray<int> ia = {1, 2, 3}, ib = {4, 5, 6};
    b.data.storage()->device = 1;  // as if b lived on GPU 1
    ib.data.storage()->device = 1;
    int bad = 0;
    {   // host-only operands: the stamp alone is no residence -- the operator goes on (and, on this GPU-less host, fails
        // at its first device call with the library's "no device" error instead of the mismatch)
        try {
            auto r = a + b;
            (void)r;
        } catch (const std::runtime_error &e) {
            if (std::string(e.what()).find("different GPUs") != std::string::npos) { std::printf("FAIL host-only a + b was refused: %s\n", e.what()); ++bad; }
        }
        if (b.data.storage()->device != a.data.storage()->device) { std::printf("FAIL host-only operand did not adopt its partner's device\n"); ++bad; }
        a.data.storage()->device = 0;  // whichever adopted the other's: back to the two-GPU picture
        b.data.storage()->device = 1;
    }
    // now as if every array had a device buffer on its GPU
    for (auto *st : {a.data.storage().get(), b.data.storage().get(), c.data.storage().get()}) st->dev = reinterpret_cast<void *>(0x1000);
    for (auto *st : {ia.data.storage().get(), ib.data.storage().get()}) st->dev = reinterpret_cast<void *>(0x1000);
    bad += expect_mismatch("a + b", [&] { auto r = a + b; });
    bad += expect_mismatch("b * a", [&] { auto r = b * a; });
    bad += expect_mismatch("a % b", [&] { volatile float r = a % b; (void)r; });
    bad += expect_mismatch("ia % ib", [&] { volatile int r = ia % ib; (void)r; });
    bad += expect_mismatch("a(view) = b(view)", [&] { a(SLICE(0, 2)) = b(SLICE(0, 2)); });
    bad += expect_mismatch("fused(a, c, b)", [&] { auto r = sm::fused<AddOp<float>, MultiplyOp<float>>(a, c, b); });
    bad += expect_mismatch("fused(a, b, 2)", [&] { auto r = sm::fused<AddOp<float>, MultiplyOp<float>>(a, b, 2.0f); });
    bad += expect_mismatch("expr(a, c, b)", [&] { auto r = sm::expr("a0 + a1 * a2", a, c, b); });
    bad += expect_mismatch("expr_sum(a, b)", [&] { volatile double r = sm::expr_sum("a0 * a1", a, b); (void)r; });
    for (auto *st : {a.data.storage().get(), b.data.storage().get(), c.data.storage().get()}) st->dev = nullptr;  // the stand-ins were never allocated
    for (auto *st : {ia.data.storage().get(), ib.data.storage().get()}) st->dev = nullptr;
    std::printf(bad ? "%d FAILED\n" : "device_mismatch ok\n", bad);
    return bad ? 1 : 0;
}
