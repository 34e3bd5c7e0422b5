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
