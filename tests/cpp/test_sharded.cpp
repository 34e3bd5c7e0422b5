// test_sharded.cpp -- sm::set_devices / sm::Sharded<T> (include/Sharded.h) against the single-GPU operators.
//
// The reference has no multi-device form (its fan-out is OpenMP, include/math/calculate.h:47,152), so the checker is the
// single-GPU path of this library, which the parity suite pins to the oracle: every sharded result must be bit-identical
// to the unsharded one (elementwise) or equal within the fp64 accumulation bound (reductions; integers exactly).
// Usage: test_sharded [n_devices]   (default: every GPU present).  On a one-GPU box this runs the whole path --
// ncclCommInitAll, ncclGroupStart/End, ncclAllReduce -- with a one-rank communicator.
#include <sm.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

static int g_failures = 0, g_checks = 0;
#define CHECK(cond)                                                                   \
    do {                                                                              \
        ++g_checks;                                                                   \
        if (!(cond)) {                                                                \
            ++g_failures;                                                             \
            if (g_failures <= 20) std::printf("FAIL %s:%d  %s\n", __FILE__, __LINE__, #cond); \
        }                                                                             \
    } while (0)

template <typename T> struct HalfSum {
    static T apply(const T &a, const T &b) { return (a + b) * T(0.5); }
    template <typename R> static R apply_simd(const R &a, const R &b);
};
SM_DEVICE_OP(HalfSum, "(a + b) * (T)0.5")

static std::uint64_t g_state = 0x9E3779B97F4A7C15ull;
static double rnd() {  // splitmix64 -> [0, 1)
    std::uint64_t z = (g_state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (double)(z >> 11) / 9007199254740992.0;
}

template <typename T>
static sm::SMArray<T> random_array(std::vector<std::size_t> shape, double lo, double hi) {
    std::size_t n = 1;
    for (auto d : shape) n *= d;
    T *buf = new T[n ? n : 1];
    for (std::size_t i = 0; i < n; ++i) buf[i] = static_cast<T>(lo + (hi - lo) * rnd());
    return sm::SMArray<T>(buf, std::move(shape));
}

template <typename T>
static bool same_bits(const sm::SMArray<T> &x, const sm::SMArray<T> &y) {
    if (x.shape() != y.shape()) return false;
    return std::memcmp(x.data.read(), y.data.read(), x.totalSize * sizeof(T)) == 0;
}

int main(int argc, char **argv) {
    int present = 0;
    smhip_device_count(&present);
    const int n = argc > 1 ? std::atoi(argv[1]) : present;
    if (n < 1 || n > present) {
        std::printf("test_sharded: %d devices asked for, %d present\n", n, present);
        return 2;
    }
    sm::set_devices(n);
    CHECK(sm::devices() == n);

    {  // config 5 in miniature: ones + ones, global sum
        const std::size_t N = (1u << 22) + 5;
        auto a = sm::Sharded<float>::ones(N), b = sm::Sharded<float>::ones(N);
        auto c = a + b;
        CHECK(sm::sum(c) == 2.0 * (double)N);
        double total = 0;
        auto d = a.apply_sum<AddOp<float>>(b, &total);
        CHECK(total == 2.0 * (double)N);
        std::size_t rows = 0;
        for (int g = 0; g < n; ++g) {
            CHECK(c.part(g).device() == g);
            rows += c.rows(g);
        }
        CHECK(rows == N);
        CHECK(d.gather()(N - 1) == 2.0f);
    }
    {  // config 3's shape: (R x C) * (1 x C), the row replicated on every GPU -- bit-identical to one GPU
        const std::size_t R = 1031, C = 517;
        auto A = random_array<float>({R, C}, -1, 1), r = random_array<float>({1, C}, -1, 1);
        auto want = A * r;
        auto sa = sm::Sharded<float>::scatter(A);
        auto sr = sm::Sharded<float>::replicate(r);
        auto got = (sa * sr).gather();
        CHECK(same_bits(got, want));
        auto got2 = (sr * sa).gather();  // the replicated side first
        CHECK(same_bits(got2, want));
        bool threw = false;
        try {
            auto bad = sa * sm::Sharded<float>::scatter(r);  // a broadcast row that was cut up instead of replicated
        } catch (const std::runtime_error &) { threw = true; }
        CHECK(threw || n == 1);
        // the reference's 4-D test pattern (tests/add.cpp:59-92): (N,224,224,3) + (1,224,1,3)
        auto X = random_array<float>({5, 24, 24, 3}, -4, 4), bias = random_array<float>({1, 24, 1, 3}, -1, 1);
        auto want4 = X + bias;
        auto got4 = (sm::Sharded<float>::scatter(X) + sm::Sharded<float>::replicate(bias)).gather();
        CHECK(same_bits(got4, want4));
    }
    {  // fused add + sum on random data: elementwise bits equal, total within the fp64 bound
        const std::size_t N = 3 * 1000 * 1000 + 7;
        auto a = random_array<float>({N}, 0, 1), b = random_array<float>({N}, 0, 1);
        auto want = a + b;
        const double want_sum = want.sum();
        double total = 0;
        auto got = sm::Sharded<float>::scatter(a).apply_sum<AddOp<float>>(sm::Sharded<float>::scatter(b), &total);
        CHECK(same_bits(got.gather(), want));
        CHECK(std::fabs(total - want_sum) <= 1e-12 * want_sum);
        CHECK(std::fabs(sm::Sharded<float>::scatter(want).sum() - want_sum) <= 1e-12 * want_sum);
    }
    {  // integer dot wraps exactly like the single-GPU (and the reference's) accumulators
        const std::size_t N = 1000003;
        auto a = random_array<int>({N}, -2e9, 2e9), b = random_array<int>({N}, -2e9, 2e9);
        const int want = a % b;
        CHECK((sm::Sharded<int>::scatter(a) % sm::Sharded<int>::scatter(b)) == want);
        auto x = random_array<double>({N}, -1, 1), y = random_array<double>({N}, -1, 1);
        const double wd = x % y, gd = sm::Sharded<double>::scatter(x) % sm::Sharded<double>::scatter(y);
        CHECK(std::fabs(wd - gd) <= 1e-12 * (double)N);
    }
    {  // scalar operand, pow, a user Op: block by block, same bits
        auto a = random_array<float>({257, 129}, 0.01, 100);
        auto sa = sm::Sharded<float>::scatter(a);
        CHECK(same_bits((sa * 3.0f).gather(), a * 3.0f));
        CHECK(same_bits(sm::pow(sa, 2.5f).gather(), sm::pow(a, 2.5f)));
        auto b = random_array<float>({257, 129}, -1, 1);
        CHECK(same_bits(sa.apply<HalfSum<float>>(sm::Sharded<float>::scatter(b)).gather(), a.apply<HalfSum<float>>(b)));
        auto i = random_array<std::int64_t>({9, 11}, -1e15, 1e15);
        CHECK(same_bits((sm::Sharded<std::int64_t>::scatter(i) - sm::Sharded<std::int64_t>::scatter(i) * (std::int64_t)3).gather(),
                        i - i * (std::int64_t)3));
    }
    {  // scatter / replicate / gather move DEVICE-resident arrays device to device (smhip_copy_peer), host-born ones from their mirror
        auto a = random_array<float>({1031, 517}, -1, 1), b = random_array<float>({1031, 517}, -1, 1);
        auto dev = a + b;  // born on the device: no host mirror exists
        CHECK(same_bits(sm::Sharded<float>::scatter(dev).gather(), a + b));
        auto turned = dev.transpose();  // a strided view of a device array
        CHECK(same_bits(sm::Sharded<float>::scatter(turned).gather(), turned.contiguous()));
        auto rep = sm::Sharded<float>::replicate(dev);
        for (int g = 0; g < n; ++g) {
            CHECK(rep.part(g).device() == g);
            CHECK(same_bits(rep.part(g), dev));
        }
        CHECK(same_bits(rep.gather(), dev));
        auto fresh = random_array<float>({64, 33}, -1, 1);  // exists only in host memory, and strided
        auto hv = fresh.transpose();
        auto shv = sm::Sharded<float>::scatter(hv);
        auto fresh2 = random_array<float>({33, 64}, 0, 0);
        CHECK(same_bits(shv.gather(), hv.contiguous()));
        // a result gathered on this GPU feeds the next operator at once (the copies ran on this device's stream)
        auto sum_back = sm::Sharded<float>::scatter(dev).gather() + dev;
        CHECK(same_bits(sum_back, dev + dev));
        (void)fresh2;
    }
    {  // fewer rows than GPUs: trailing GPUs hold empty blocks
        auto a = random_array<double>({1, 77}, -1, 1);
        auto sa = sm::Sharded<double>::scatter(a);
        CHECK(same_bits((sa + sa).gather(), a + a));
        CHECK(std::fabs((sa + sa).sum() - (a + a).sum()) <= 1e-13 * 77);
    }
    sm::synchronize_devices();
    sm::set_devices(0);
    CHECK(sm::devices() == 0);
    std::printf("test_sharded: %d devices, %d checks, %d failures\n", n, g_checks, g_failures);
    return g_failures ? 1 : 0;
}
