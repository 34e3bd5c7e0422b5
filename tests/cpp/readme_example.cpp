// The C++ example of README.md ("Using it from C++"), compiled and run by tests/test_gpu_cpp.py so it cannot rot.
#include <sm.h>
#include <iostream>
template <typename T> struct Hypot2 { static T apply(const T &a, const T &b) { return a * a + b * b; }
                                      template <typename R> static R apply_simd(const R &, const R &); };
SM_DEVICE_OP(Hypot2, "a * a + b * b")
int main() {
    auto a = sm::ones<float>(4096, 4096) * 3.0f;
    sm::SMArray<float> row = {1, 2, 3, 4};
    auto b = a.transpose() + a;
    auto c = sm::pow(b, 2.5f) / 2.0f;
    a(SLICE(0, 2), SLICE_ALL) = c(SLICE(2, 4), SLICE_ALL);
    auto d = a.apply<Hypot2<float>>(c);
    auto row4096 = sm::ones<float>(1, 4096) * 0.5f, mean_col = sm::ones<float>(4096, 1) * 2.0f, std_col = sm::ones<float>(4096, 1) * 4.0f;
    auto g = (a * row4096 + b) * 0.5f;
    (void)g;
    auto n = sm::ones<float>(4096, 4096);
    n = (n - mean_col) / std_col;
    auto mse = sm::pow(n - b, 2.0f).sum();
    (void)mse;
    auto e = sm::fused<AddOp<float>, MultiplyOp<float>>(a, b, c);
    auto f = sm::expr("(a0 + a1) * a2 - 3 * a3", a, b, c, d);
    (void)f;
    double total = sm::sum(e);
    float first = e(0, 0);
    std::cout << row * 2.0f << " " << total << " " << first << "\n";
}
