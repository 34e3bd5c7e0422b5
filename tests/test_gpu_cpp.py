"""The drop-in C++ surface on the GPU: the reference's own test cases (restated in
tests/cpp/test_reference_suite.cpp against include/sm.h) and the benchmark harness."""
import subprocess

import pytest

pytestmark = pytest.mark.gpu


def _exe(name):
    from simplemath_amd import build
    build.build_lib()
    return build.build_host_programs()[name]


def test_reference_suite_through_cpp_header():
    r = subprocess.run([_exe("test_reference_suite")], capture_output=True, text=True, timeout=600)
    print(r.stdout[-3000:], r.stderr[-2000:])
    assert r.returncode == 0, r.stdout[-3000:]
    assert " 0 failures" in r.stdout


def test_benchmark_harness_runs():
    r = subprocess.run([_exe("benchmark_add")], capture_output=True, text=True, timeout=600)
    print(r.stdout)
    assert r.returncode == 0, r.stderr
    assert "million_check" in r.stdout and "simple_check" in r.stdout
    r = subprocess.run([_exe("benchmark_pow")], capture_output=True, text=True, timeout=600)
    print(r.stdout)
    assert r.returncode == 0, r.stderr
    assert "BM_SMArrayPow_Large/1000" in r.stdout


def test_readme_example():
    r = subprocess.run([_exe("readme_example")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    row, total, first = r.stdout.strip().rsplit(" ", 2)
    assert row == "[2, 4, 6, 8]"
    c = 6.0 ** 2.5 / 2.0                      # a = 3, b = a.T + a = 6, c = b^2.5 / 2; rows 0-1 of a are then set to c
    assert abs(float(first) - (c + 6.0) * c) < 1e-2 * c
    assert abs(float(total) - ((4096 - 2) * 4096 * 9.0 * c + 2 * 4096 * (c + 6.0) * c)) < 1e-5 * float(total)


def test_pow_exhaustive_over_all_positive_floats():
    """BASELINE config 4's exponent over EVERY positive finite float (2 139 095 039 values, denormals and the overflow /
    underflow ranges included) against x*x*sqrt(x) in fp64: nothing beyond 1 ULP (bar: 4), >= 95 % correctly rounded."""
    import re
    for y, floor in (("2.5", 96.5), ("1.5", 94.5)):
        r = subprocess.run([_exe("pow_exhaustive"), y], capture_output=True, text=True, timeout=600)
        print(r.stdout)
        assert r.returncode == 0, r.stdout + r.stderr
        m = re.search(r"0 ULP \d+ \(([\d.]+) %\)  1 ULP \d+ \([\d.]+ %\)  2 ULP (\d+)  >2 ULP (\d+)", r.stdout)
        assert m and int(m.group(2)) == 0 and int(m.group(3)) == 0 and float(m.group(1)) >= floor, r.stdout


def test_pool_stream_ordering_across_threads_and_stream_switches():
    """tests/cpp/pool_streams.hip: a block used on one thread's stream and freed by another thread, a stream switch
    between use and free, a destroyed stream, two async reductions on two streams."""
    r = subprocess.run([_exe("pool_streams")], capture_output=True, text=True, timeout=600)
    print(r.stdout[-3000:], r.stderr[-2000:])
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


def test_readme_plugin_recipe_on_the_gpu():
    """The reference's plugin recipe (README.md:86-133) plus SM_DEVICE_OP: element_wise_op<T, MyOp<T>> runs on the GPU."""
    r = subprocess.run([_exe("readme_recipe")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "device form" in r.stdout and " 0 mismatches" in r.stdout, r.stdout + r.stderr


def test_dot_for_every_element_type_of_the_reference():
    """tests/cpp/test_dot_types.cpp: operator% for the generic dot_product<T>'s 8- / 16-bit and unsigned integer types (bit
    exact against the statement as written) and for std::complex<double> arrays that stay resident (views, contiguous(),
    repeat(), assignment into views on the device)."""
    r = subprocess.run([_exe("test_dot_types")], capture_output=True, text=True, timeout=600)
    print(r.stdout[-3000:], r.stderr[-2000:])
    assert r.returncode == 0 and " 0 failures" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]


def test_operator_chains_fuse_through_the_cpp_surface():
    """tests/cpp/test_chain_fusion.cpp: `(A * row + B) * s` written as one expression is ONE smhip_chain call, bit-identical to
    the same expression with a named value per step; named values are computed at their `;`; host writes, assignment into
    an operand, views of temporaries, exceptions and discarded expressions keep the eager meaning."""
    r = subprocess.run([_exe("test_chain_fusion")], capture_output=True, text=True, timeout=600)
    print(r.stdout[-3000:], r.stderr[-2000:])
    assert r.returncode == 0 and " 0 failures" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
