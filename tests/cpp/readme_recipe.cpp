// readme_recipe.cpp -- the reference's "Extending with Custom Operations" recipe (README.md:86-133), restated step by
// step against the drop-in headers, so that the plugin contract (include/math/add.h:5-14: a struct template with
// apply() and a declared-only apply_simd<REG>(), specialised per register type) cannot silently break again.
//
// Built three ways by the tests:
//   (default)                   steps 1-3 exactly as the README gives them: NO device form.  Must NOT compile, and the
//                               error must say what to add (tests/test_plugin_contract.py, a CPU test: g++ -fsyntax-only).
//   -DSM_ALLOW_HOST_USER_OPS    the same source, unmodified, brought up through its host apply() (explicit opt-in).
//   -DRECIPE_WITH_DEVICE_FORM   the one line a maintainer adds for gfx950; the loop then runs on the MI355X and the
//                               string is checked against apply() on first use.
#include <sm.h>

#include <cstdio>

// ---- step 1 (README.md:90-104): include/math/my_op.h -------------------------------------------------------------
#include "math/helpers.h"

template<typename T>
struct MyOp {
    static T apply(const T& a, const T& b) {
    return (a + b) * 2; // example
}

template<typename SIMD_T>
static SIMD_T apply_simd(const SIMD_T& a, const SIMD_T& b);

};

// ---- step 2 (README.md:108-117): the x86 register specialisation -- parses because helpers.h brings <immintrin.h> ----
template<>
template<>
inline __m256 MyOp<float>::apply_simd<__m256>(const __m256& a, const __m256& b) {
    const __m256 two = _mm256_set1_ps(2.0f);
    const __m256 sum = _mm256_add_ps(a, b);
    return _mm256_mul_ps(sum, two);
}

#ifdef RECIPE_WITH_DEVICE_FORM
SM_DEVICE_OP(MyOp, "(a + b) * 2")  // the gfx950 counterpart of step 2
#endif

// ---- step 3 (README.md:121-132): the operator's body, word for word, as a free function (it is a member there) --------
template <typename T>
sm::SMArray<T> my_operator(const sm::SMArray<T>& self, const sm::SMArray<T>& arr) {
    auto broadcastResult = sm::broadcast(self.shape(), self.strides(), arr.shape(), arr.strides());
    T* result = new T[broadcastResult.totalSize];
    element_wise_op<T, MyOp<T>>(self.data, broadcastResult.newStrides1, arr.data,
    broadcastResult.newStrides2,
    broadcastResult.totalSize, result,
    broadcastResult.resultShape);
    return sm::SMArray<T>(result, std::move(broadcastResult.resultShape));
}

int main() {
    sm::SMArray<float> a = {{1, 2, 3}, {4, 5, 6}}, b = {{10, 20, 30}};
    auto r = my_operator(a, b);  // (2,3) with (1,3)
    const float want[6] = {22, 44, 66, 28, 50, 72};
    int bad = 0;
    for (int i = 0; i < 6; ++i) bad += r.data[i] != want[i];
    sm::SMArray<int> ia = {1, 2, 3}, ib = {4, 5, 6};
    auto ri = my_operator(ia, ib);
    bad += ri.data[2] != 18;
    std::printf("readme_recipe: %s, %d mismatches\n",
#ifdef RECIPE_WITH_DEVICE_FORM
                "device form",
#else
                "host apply (SM_ALLOW_HOST_USER_OPS)",
#endif
                bad);
    return bad ? 1 : 0;
}
