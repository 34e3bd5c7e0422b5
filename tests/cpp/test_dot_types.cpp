// test_dot_types.cpp -- SMArray<T>::operator% for every element type the reference's dot_product<T> serves (product.h:8-224):
// the generic template's 8- / 16-bit and unsigned integers, and std::complex<double> -- whose arrays are now as
// device-resident as any other (views, contiguous(), repeat(), assignment into views move on the device as pairs of doubles).
// Expected values are computed here on the host with the reference's own statements (`sum += a[i] * b[i]` in T;
// `result += a[i] * b[i]` in std::complex<double>).
#include <sm.h>

#include <complex>
#include <cstdint>
#include <cstdio>
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

static std::uint64_t g_state = 0x1234567ull;
static std::uint64_t rnd() {
    std::uint64_t z = (g_state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

template <typename T>
static void integer_dot(std::size_t rows, std::size_t cols) {
    const std::size_t n = rows * cols;
    T *pa = new T[n], *pb = new T[n];
    for (std::size_t i = 0; i < n; ++i) { pa[i] = static_cast<T>(rnd()); pb[i] = static_cast<T>(rnd()); }
    std::vector<T> ha(pa, pa + n), hb(pb, pb + n);
    sm::SMArray<T> a(pa, {rows, cols}), b(pb, {rows, cols});
    T want = 0;
    for (std::size_t i = 0; i < n; ++i) want += ha[i] * hb[i];  // product.h:16-19, as written
    CHECK((a % b) == want);
    // through transposed views: the elements pair up differently, (a.T)[j][i] = a[i][j]
    T want_t = 0;
    for (std::size_t j = 0; j < cols; ++j)
        for (std::size_t i = 0; i < rows; ++i) want_t += ha[i * cols + j] * hb[i * cols + j];
    CHECK((a.transpose() % b.transpose()) == want_t);
    CHECK(want_t == want);
    // one row against one row
    T want_r = 0;
    for (std::size_t j = 0; j < cols; ++j) want_r += ha[1 * cols + j] * hb[2 * cols + j];
    CHECK((a(1, SLICE_ALL) % b(2, SLICE_ALL)) == want_r);
}

int main() {
    integer_dot<std::int8_t>(37, 129);
    integer_dot<std::uint8_t>(64, 4099);
    integer_dot<std::int16_t>(37, 129);
    integer_dot<std::uint16_t>(5, 100003);
    integer_dot<std::uint32_t>(37, 129);
    integer_dot<std::uint64_t>(37, 129);
    integer_dot<unsigned long long>(3, 7);
    integer_dot<long>(33, 65);

    typedef std::complex<double> C;
    {
        const std::size_t R = 33, K = 70, n = R * K;
        C *pa = new C[n], *pb = new C[n];
        auto unit = [] { return (double)(rnd() >> 11) / 9007199254740992.0 * 2 - 1; };
        for (std::size_t i = 0; i < n; ++i) { pa[i] = C(unit(), unit()); pb[i] = C(unit(), unit()); }
        std::vector<C> ha(pa, pa + n), hb(pb, pb + n);
        sm::SMArray<C> a(pa, {R, K}), b(pb, {R, K});
        C want(0, 0);
        double scale = 0;
        for (std::size_t i = 0; i < n; ++i) { want += ha[i] * hb[i]; scale += std::abs(ha[i]) * std::abs(hb[i]); }
        const C got = a % b;  // both arrays are uploaded once and stay resident
        CHECK(std::abs(got - want) <= 4.0 * n * 1.1102230246251565e-16 * scale);
        CHECK((a % b) == got);  // the same bits again, now from resident data
        // a strided complex view made dense ON THE DEVICE: transpose().contiguous() is the transposed matrix
        auto at = a.transpose().contiguous();
        bool same = at.shape() == std::vector<std::size_t>{K, R};
        const C *t = at.cdata();
        for (std::size_t i = 0; same && i < R; ++i)
            for (std::size_t j = 0; j < K; ++j) same = same && t[j * R + i] == ha[i * K + j];
        CHECK(same);
        // dot through views = dot of the gathered elements
        C want_col(0, 0);
        for (std::size_t i = 0; i < R; ++i) want_col += ha[i * K + 3] * hb[i * K + 5];
        const C got_col = a(SLICE_ALL, 3) % b(SLICE_ALL, 5);
        CHECK(std::abs(got_col - want_col) <= 4.0 * R * 1.1102230246251565e-16 * R);
        // assignment into a view and repeat(), on the device
        sm::SMArray<C> z = sm::SMArray<C>::device_full({R, K}, C(0, 0));
        z(2, SLICE_ALL) = b(7, SLICE_ALL);
        CHECK(z(2, 11) == hb[7 * K + 11] && z(3, 11) == C(0, 0));
        auto rep = a(0, SLICE_ALL).repeat(3);
        CHECK(rep.totalSize == 3 * K && rep(0) == ha[0] && rep(2) == ha[0] && rep(3) == ha[1] && rep(3 * K - 1) == ha[K - 1]);
        // n = 1: the reference's scalar tail, exactly one product
        sm::SMArray<C> x = {C(1.5, -2.0)}, y = {C(0.25, 4.0)};
        CHECK((x % y) == C(1.5 * 0.25 - (-2.0) * 4.0, 1.5 * 4.0 + (-2.0) * 0.25));
    }
    {  // std::complex<float>: the generic template's instantiation, resident like the double form
        typedef std::complex<float> CF;
        const std::size_t R = 17, K = 300, n = R * K;
        CF *pa = new CF[n], *pb = new CF[n];
        auto unit = [] { return (float)((double)(rnd() >> 11) / 9007199254740992.0 * 2 - 1); };
        for (std::size_t i = 0; i < n; ++i) { pa[i] = CF(unit(), unit()); pb[i] = CF(unit(), unit()); }
        std::vector<CF> ha(pa, pa + n), hb(pb, pb + n);
        sm::SMArray<CF> a(pa, {R, K}), b(pb, {R, K});
        std::complex<double> want(0, 0);
        double scale = 0;
        for (std::size_t i = 0; i < n; ++i) {
            want += std::complex<double>(ha[i]) * std::complex<double>(hb[i]);
            scale += std::abs(std::complex<double>(ha[i])) * std::abs(std::complex<double>(hb[i]));
        }
        const CF got = a % b;
        CHECK(std::abs(std::complex<double>(got) - want) <= 2.4e-7 * (std::abs(want) + 1e-3 * scale));
        CHECK((a.transpose() % b.transpose()) == got || std::abs(std::complex<double>(a.transpose() % b.transpose()) - want) <= 2.4e-7 * (std::abs(want) + 1e-3 * scale));
        sm::SMArray<CF> x = {CF(1.5f, -2.0f)}, y = {CF(0.25f, 4.0f)};
        CHECK((x % y) == CF(1.5f * 0.25f + 2.0f * 4.0f, 1.5f * 4.0f - 2.0f * 0.25f));
    }
    std::printf("test_dot_types: %d checks, %d failures\n", g_checks, g_failures);
    return g_failures ? 1 : 0;
}
