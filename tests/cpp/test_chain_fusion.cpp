// test_chain_fusion.cpp -- operator chains through the drop-in surface: a temporary that feeds the next operator of the same
// full-expression is fused into ONE smhip_chain call (include/SMArray.h "deferred operator chains"), named values are computed
// by the end of their statement, and the values are bit-identical to the eager chain the reference evaluates (one operator
// call and one temporary per step, reference SMArray.h:217-305).  Expected values: the same expression written with a
// named value per step (which never fuses) -- itself pinned by the golden / oracle tests -- and, for small cases, the
// statement evaluated on the host in T.
#include <sm.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <stdexcept>
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

static std::uint64_t g_state = 0x9876543ull;
static std::uint64_t rnd() {
    std::uint64_t z = (g_state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
template <typename T>
static T sample() {
    if constexpr (std::is_integral_v<T>) {
        const T v = static_cast<T>(static_cast<std::int64_t>(rnd() % 2001) - 1000);
        return v == 0 ? T(7) : v;
    } else {
        const double m = 0.25 + 3.75 * static_cast<double>(rnd() >> 11) / 9007199254740992.0;
        return static_cast<T>((rnd() & 1) ? m : -m);
    }
}
template <typename T, typename... D>
static sm::SMArray<T> random_array(D... dims) {
    std::vector<std::size_t> shape{static_cast<std::size_t>(dims)...};
    std::size_t n = 1;
    for (auto d : shape) n *= d;
    T *p = new T[n];
    for (std::size_t i = 0; i < n; ++i) p[i] = sample<T>();
    return sm::SMArray<T>(p, std::move(shape));
}
template <typename T>
static bool same_bits(const sm::SMArray<T> &x, const sm::SMArray<T> &y) {
    if (x.shape() != y.shape()) return false;
    return std::memcmp(x.cdata(), y.cdata(), x.totalSize * sizeof(T)) == 0;
}
struct Delta {
    sm::detail::FusionStats at = sm::fusion_stats();
    unsigned long long chains() const { return sm::fusion_stats().chains - at.chains; }
    unsigned long long stages() const { return sm::fusion_stats().fused_stages - at.fused_stages; }
    unsigned long long singles() const { return sm::fusion_stats().single_ops - at.single_ops; }
};

template <typename T>
static void forms(std::size_t rows, std::size_t cols) {
    const auto A = random_array<T>(rows, cols), B = random_array<T>(rows, cols);
    const auto row = random_array<T>(1, cols), col = random_array<T>(rows, 1), one = random_array<T>(1, 1);
    const T s = std::is_integral_v<T> ? T(3) : T(0.5);
    // tiny host-built operands ride in the plain operator's launch packet and are not deferred: the counts below hold for
    // arrays past that size (SMHIP_INLINE_MAX_OUTPUTS results)
    const bool counted = rows * cols > SMHIP_INLINE_MAX_OUTPUTS;
    {   // the harness's chain_check as ONE expression: one call, three operators inside
        Delta d;
        auto fused = (A * row + B) * s;
        CHECK(!counted || (d.chains() == 1 && d.stages() == 3 && d.singles() == 0));
        auto t1 = A * row;
        auto t2 = t1 + B;
        auto t3 = t2 * s;
        CHECK(!counted || (d.chains() == 1 && d.singles() == 3));  // named values: one operator per statement, nothing deferred past its `;`
        CHECK(same_bits(fused, t3));
    }
    {   // column, then a row on the LEFT of a non-commutative operator (the temporary is the right operand)
        auto fused = row / ((A - col) / B + s);
        auto t1 = A - col;
        auto t2 = t1 / B;
        auto t3 = t2 + s;
        auto t4 = row / t3;
        CHECK(same_bits(fused, t4));
    }
    {   // both operands of the last operator are temporaries: the left one is computed first, the right one's chain continues
        Delta d;
        auto fused = (A + B) * (A - row);
        auto l = A + B;
        auto r = A - row;
        auto want = l * r;
        CHECK(same_bits(fused, want));
        CHECK(!counted || d.chains() == 1);  // one side ran alone, the other side's chain took the last operator in
    }
    {   // a one-element operand, and the result shape growing along the chain: (row + col) is (rows, cols)
        auto fused = (row + col) * A - one;
        auto t1 = row + col;
        auto t2 = t1 * A;
        auto t3 = t2 - one;
        CHECK(same_bits(fused, t3));
        auto grown = (row * s + one) * col;  // (1, cols) all the way until `* col`
        auto g1 = row * s;
        auto g2 = g1 + one;
        auto g3 = g2 * col;
        CHECK(grown.shape() == std::vector<std::size_t>({rows, cols}));
        CHECK(same_bits(grown, g3));
    }
    {   // transposed views cut the chain (they run through the tile kernel) and it continues
        const auto Sq = random_array<T>(cols, cols), Sq2 = random_array<T>(cols, cols);
        auto fused = (Sq.transpose() + Sq2) * s - Sq;
        auto t1 = Sq.transpose() + Sq2;
        auto t2 = t1 * s;
        auto t3 = t2 - Sq;
        CHECK(same_bits(fused, t3));
        auto fused2 = (Sq + Sq2) * Sq.transpose() - s;
        auto u1 = Sq + Sq2;
        auto u2 = u1 * Sq.transpose();
        auto u3 = u2 - s;
        CHECK(same_bits(fused2, u3));
    }
    {   // more operands than one chain records (8): cut and continued
        auto fused = ((((((((A + B) * row - col) + A) * B - row) + col) * s + B) - A) * row + one) * s;
        auto t = A + B;
        auto t2 = t * row;
        auto t3 = t2 - col;
        auto t4 = t3 + A;
        auto t5 = t4 * B;
        auto t6 = t5 - row;
        auto t7 = t6 + col;
        auto t8 = t7 * s;
        auto t9 = t8 + B;
        auto t10 = t9 - A;
        auto t11 = t10 * row;
        auto t12 = t11 + one;
        auto t13 = t12 * s;
        CHECK(same_bits(fused, t13));
    }
}

template <typename T>
static void periodic_4d() {
    // the reference tests' broadcast pattern, ones(32,224,224,3)(0, SLICE_ALL) o (1,224,1,3) (tests/add.cpp:59-92), in a chain
    const auto big = random_array<T>(3, 28, 20, 3);
    const auto small = random_array<T>(1, 28, 1, 3), rgb = random_array<T>(1, 1, 1, 3), per_sample = random_array<T>(3, 1, 1, 1);
    const T s = T(2);
    auto fused = (big + small) * rgb - per_sample * s;
    auto t1 = big + small;
    auto t2 = t1 * rgb;
    auto p = per_sample * s;
    auto t3 = t2 - p;
    CHECK(same_bits(fused, t3));
    auto v = big(0, SLICE_ALL);  // the view the reference tests use: (28, 20, 3) of the first sample
    auto fused_v = (v * small + v) / s;
    auto w1 = v * small;
    auto w2 = w1 + v;
    auto w3 = w2 / s;
    CHECK(fused_v.shape() == std::vector<std::size_t>({1, 28, 20, 3}));
    CHECK(same_bits(fused_v, w3));
}

static void host_values() {
    // small enough to evaluate the statement on the host in T
    sm::SMArray<float> a = {{1, 2, 3}, {4, 5, 6}};
    sm::SMArray<float> r = {{10, 20, 30}};
    sm::SMArray<float> c(new float[2]{0.5f, 0.25f}, {2, 1});  // ({{0.5f}, {0.25f}} would be the 1-D list {0.5f, 0.25f}: one-element braces)
    auto out = ((a * r + a) * c - 1.0f) / 3.0f;
    const float av[2][3] = {{1, 2, 3}, {4, 5, 6}}, rv[3] = {10, 20, 30}, cv[2] = {0.5f, 0.25f};
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 3; ++j) {
            const float t1 = av[i][j] * rv[j], t2 = t1 + av[i][j], t3 = t2 * cv[i], t4 = t3 - 1.0f, want = t4 / 3.0f;
            CHECK(out(i, j) == want);
        }
    sm::SMArray<int> ia = {{7, -7, 100}, {2147483647, -5, 9}};
    sm::SMArray<int> ir = {{2, -3, 7}};
    auto io = (ia / ir + ia) * 3;  // truncation toward zero, wrapping products
    const int iav[2][3] = {{7, -7, 100}, {2147483647, -5, 9}}, irv[3] = {2, -3, 7};
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 3; ++j) {
            const std::uint32_t q = static_cast<std::uint32_t>(iav[i][j] / irv[j]), t = q + static_cast<std::uint32_t>(iav[i][j]);
            CHECK(io(i, j) == static_cast<int>(t * 3u));
        }
}

static void ordering() {
    const std::size_t n = 1 << 16;
    // a host write between two statements: the first statement's value was computed at its `;`
    auto a = sm::ones<float>(n), b = sm::ones<float>(n);
    auto c = (a + b) * 2.0f;       // 4 everywhere
    a.data[0] = 100.0f;            // must not reach c
    CHECK(c(0) == 4.0f && c(1) == 4.0f);
    auto d = (a + b) * 2.0f;       // sees the write
    CHECK(d(0) == 202.0f && d(1) == 4.0f);
    // assignment INTO an operand of the expression on its right-hand side
    auto x = sm::ones<float>(8, 16), y = sm::ones<float>(8, 16);
    x = (x + y) * 3.0f - x;        // 5 everywhere
    CHECK(x(0, 0) == 5.0f && x(7, 15) == 5.0f);
    x(SLICE(0, 2), SLICE_ALL) = (x(SLICE(0, 2), SLICE_ALL) * 2.0f + y(SLICE(0, 2), SLICE_ALL)) * 2.0f;  // 22 in rows 0-1
    CHECK(x(0, 0) == 22.0f && x(1, 15) == 22.0f && x(2, 0) == 5.0f);
    {   // `x = <expression>` is evaluated straight into x: no temporary, no copy -- also in place (x among the operands, as itself)
        const auto before = sm::fusion_stats().direct_assignments;
        auto z = sm::zeros<float>(8, 16);
        z = (x + y) * 2.0f;                                  // rows 0-1: 46, the others 12
        CHECK(sm::fusion_stats().direct_assignments == before + 1);
        CHECK(z(0, 0) == 46.0f && z(5, 3) == 12.0f);
        z = z * 0.5f + z;                                    // in place, z twice
        CHECK(sm::fusion_stats().direct_assignments == before + 2);
        CHECK(z(0, 0) == 69.0f && z(5, 3) == 18.0f);
        z(SLICE(4, 6), SLICE_ALL) = z(SLICE(4, 6), SLICE_ALL) - y(SLICE(4, 6), SLICE_ALL);   // a dense block inside z
        CHECK(sm::fusion_stats().direct_assignments == before + 3);
        CHECK(z(4, 0) == 17.0f && z(5, 15) == 17.0f && z(3, 0) == 18.0f && z(6, 0) == 18.0f);
        // NOT in place: the right-hand side reads z through another view of the same storage (a broadcast row of z, z transposed)
        auto q = sm::ones<float>(16, 16);
        q.data[1] = 3.0f;                                    // q(0, 1) = 3
        q = q + q(0, SLICE_ALL);                             // row 0 of q added to every row: must use the OLD row 0 throughout
        CHECK(q(0, 1) == 6.0f && q(5, 1) == 4.0f && q(5, 0) == 2.0f);
        q = q.transpose() * 2.0f + q;
        CHECK(q(1, 0) == 2.0f * 6.0f + 2.0f && q(0, 1) == 2.0f * 2.0f + 6.0f && q(5, 1) == 2.0f * 2.0f + 4.0f && q(1, 5) == 2.0f * 4.0f + 2.0f);  // q.T(i, j) = q(j, i), all from the OLD q
        CHECK(sm::fusion_stats().direct_assignments == before + 3);
    }
    // a temporary's VIEW is somebody else looking at it: computed, not continued
    auto tv = (x + y).transpose() * 2.0f;
    CHECK(tv.shape() == std::vector<std::size_t>({16, 8}) && tv(0, 0) == 46.0f && tv(0, 2) == 12.0f);
    // a discarded expression computes nothing and breaks nothing
    (a + b) * 5.0f;
    auto e = a + b;
    CHECK(e(1) == 2.0f);
    // an exception in the middle of an expression (shape mismatch) leaves the thread usable
    auto p = sm::ones<float>(4, 4), q = sm::ones<float>(3, 4);
    bool threw = false;
    try {
        auto bad = (p + p) * q;
        (void)bad;
    } catch (const std::runtime_error &) { threw = true; }
    CHECK(threw);
    auto ok = (p + p) * p;
    CHECK(ok(3, 3) == 2.0f);
    // pow, dot and sum look at a pending temporary: it is computed for them
    auto pw = sm::pow((p + p) * p, 2.0f);
    CHECK(pw(0, 0) == 4.0f);
    CHECK(((p + p) * 2.0f) % p == 64.0f);
    CHECK(((p + p) * 2.0f).sum() == 64.0);
    // a function taking const SMArray& gets a computed value
    auto f = [](const sm::SMArray<float> &v) { return v(0, 0) + v(3, 3); };
    CHECK(f((p + p) * 3.0f) == 12.0f);
    // std::move of a named value is a plain operand
    auto m = p + p;
    auto m2 = std::move(m) * 2.0f + p;
    CHECK(m2(1, 1) == 5.0f);
}

// sm::pow of an expression's temporary is one more stage of its chain: ^2 inside the kernel, any other exponent by cutting the chain
// and running pow's own evaluation on the value so far -- the same bits as the operators called one by one, for every element type.
template <typename T>
static void powers() {
    const std::size_t rows = 66, cols = 200;
    const auto A = random_array<T>(rows, cols), B = random_array<T>(rows, cols);
    const auto row = random_array<T>(1, cols);
    {   // the squared difference: one launch
        Delta d;
        auto f = sm::pow(A - B, T(2));
        CHECK(d.chains() == 1 && d.stages() == 2 && d.singles() == 0);
        auto t1 = A - B;
        auto t2 = sm::pow(t1, T(2));
        CHECK(same_bits(f, t2));
    }
    {   // ... with more around it: ((A - row)^2 + B) * 2
        Delta d;
        auto f = (sm::pow(A - row, T(2)) + B) * T(2);
        CHECK(d.chains() == 1 && d.stages() == 4);
        auto t1 = A - row;
        auto t2 = sm::pow(t1, T(2));
        auto t3 = t2 + B;
        auto t4 = t3 * T(2);
        CHECK(same_bits(f, t4));
    }
    {   // another exponent cuts the chain; the values are those of the plain operators
        const T e = std::is_integral_v<T> ? T(3) : T(2.5);
        auto pos = A * A + T(1);  // positive bases
        auto f = sm::pow(pos * T(2), e) - B;
        auto t1 = pos * T(2);
        auto t2 = sm::pow(t1, e);
        auto t3 = t2 - B;
        CHECK(same_bits(f, t3));
        auto g = sm::pow(pos + row, T(1)) * T(3);  // ^1: no stage at all
        auto u1 = pos + row;
        auto u2 = sm::pow(u1, T(1));
        auto u3 = u2 * T(3);
        CHECK(same_bits(g, u3));
    }
    {   // the squared error: the sum in the chain's own pass, the squares never written
        const auto before = sm::fusion_stats().summed_chains;
        const double got = sm::pow(A - B, T(2)).sum(), got2 = sm::sum((A - B) * (A - B));
        CHECK(sm::fusion_stats().summed_chains == before + 2);
        auto t1 = A - B;
        auto t2 = sm::pow(t1, T(2));
        const double want = t2.sum();  // a named value: computed, then summed
        CHECK(sm::fusion_stats().summed_chains == before + 2);
        double scale = 0;
        for (std::size_t i = 0; i < t2.totalSize; ++i) scale += std::fabs(static_cast<double>(t2.cdata()[i]));
        CHECK(std::fabs(got - want) <= 1e-15 * scale && std::fabs(got2 - want) <= 1e-15 * scale);
        // a row operand: the chain into a temporary, then its sum -- still through the one call
        const double gr = ((A - row) * B).sum();
        auto u1 = A - row;
        auto u2 = u1 * B;
        double scale2 = 0;
        for (std::size_t i = 0; i < u2.totalSize; ++i) scale2 += std::fabs(static_cast<double>(u2.cdata()[i]));
        CHECK(std::fabs(gr - u2.sum()) <= 1e-15 * scale2);
        // a by-value parameter bound to the temporary is a NAMED array: it may be read after its sum
        auto both = [](sm::SMArray<T> v) { const double s = v.sum(); return s + static_cast<double>(v(0, 0)); };
        CHECK(std::fabs(both(A - B) - (t1.sum() + static_cast<double>(t1(0, 0)))) <= 1e-15 * scale + 1e-9);
    }
    {   // a named operand is not a temporary: computed by itself, as before
        Delta d;
        auto f = sm::pow(A, T(2));
        CHECK(d.chains() == 0);
        auto g = A * A;
        CHECK(same_bits(f, g));
    }
}

template <typename T>
static void hooks() {
    // sm::fused and sm::expr take operands that broadcast against each other (smhip_chain / smhip_fused_expr_bcast)
    const std::size_t rows = 70, cols = 96;
    const auto A = random_array<T>(rows, cols), B = random_array<T>(rows, cols);
    const auto row = random_array<T>(1, cols), col = random_array<T>(rows, 1);
    const T s = T(3);
    {
        Delta d;
        auto f = sm::fused<MultiplyOp<T>, AddOp<T>>(A, row, B);
        CHECK(d.chains() == 1 && d.stages() == 2);
        auto t1 = A * row;
        auto t2 = t1 + B;
        CHECK(same_bits(f, t2));
        auto g = sm::fused<SubtractOp<T>, DivideOp<T>>(A, col, s);
        auto u1 = A - col;
        auto u2 = u1 / s;
        CHECK(same_bits(g, u2));
        auto h = sm::fused<AddOp<T>, MultiplyOp<T>>(A, B, A);  // equal dense shapes: the two-Op kernel
        auto v1 = A + B;
        auto v2 = v1 * A;
        CHECK(same_bits(h, v2));
    }
    {
        auto e = sm::expr("(a0 * a1 + a2) * s0 - a3", {s}, A, row, B, col);
        auto t1 = A * row;
        auto t2 = t1 + B;
        auto t3 = t2 * s;
        auto t4 = t3 - col;
        CHECK(same_bits(e, t4));
        auto e2 = sm::expr("a0 + a1", row, col);  // (1, cols) and (rows, 1): the result is (rows, cols)
        auto w = row + col;
        CHECK(e2.shape() == std::vector<std::size_t>({rows, cols}) && same_bits(e2, w));
        const auto Sq = random_array<T>(cols, cols);
        auto e3 = sm::expr("a0 - a1 * a2", Sq, Sq.transpose(), row);
        auto x1 = Sq.transpose() * row;
        auto x2 = Sq - x1;
        CHECK(same_bits(e3, x2));
        bool threw = false;
        try { auto bad = sm::expr("a0 + a1", A, random_array<T>(rows + 1, cols)); (void)bad; } catch (const std::runtime_error &) { threw = true; }
        CHECK(threw);
    }
}

static void sizes() {
    // around the vector / workgroup boundaries, 1-D (the tail lane) and element types
    for (std::size_t n : {std::size_t(1), std::size_t(3), std::size_t(5), std::size_t(257), std::size_t(4099), std::size_t(100003), std::size_t(1) << 20}) {
        const auto a = random_array<double>(n), b = random_array<double>(n);
        auto fused = (a - b) * a / 3.0;
        auto t1 = a - b;
        auto t2 = t1 * a;
        auto t3 = t2 / 3.0;
        CHECK(same_bits(fused, t3));
    }
}

#define STEP(call) do { std::printf("%s\n", #call); std::fflush(stdout); call; } while (0)
int main() {
    STEP(forms<float>(67, 128));
    STEP(forms<float>(129, 1000));
    STEP(forms<double>(33, 64));
    STEP(forms<std::int32_t>(67, 128));
    STEP(forms<std::int64_t>(21, 34));
    STEP(forms<float>(5, 3));
    STEP(periodic_4d<float>());
    STEP(periodic_4d<double>());
    STEP(periodic_4d<std::int32_t>());
    STEP(periodic_4d<std::int64_t>());
    STEP(powers<float>());
    STEP(powers<double>());
    STEP(powers<std::int32_t>());
    STEP(powers<std::int64_t>());
    STEP(hooks<float>());
    STEP(hooks<double>());
    STEP(hooks<std::int32_t>());
    STEP(hooks<std::int64_t>());
    STEP(host_values());
    STEP(ordering());
    STEP(sizes());
    const auto st = sm::fusion_stats();
    std::printf("%d checks, %d failures; %llu chains with %llu operators inside, %llu operators alone\n", g_checks, g_failures, st.chains,
                st.fused_stages, st.single_ops);
    return g_failures ? 1 : 0;
}
