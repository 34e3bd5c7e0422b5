"""The float pow algorithm itself (simplemath_amd/csrc/sm_pow.h, the source the gfx950
kernels inline), compiled for the host and swept against the correctly rounded x**y:
8.8 M (x, y) pairs incl. subnormals, values near 1 with huge exponents, the whole
special-case lattice against libm."""
import os
import re
import subprocess


def test_pow_algorithm_on_host():
    from simplemath_amd import build
    exe = build.build_host_programs()["pow_host_check"]
    out = subprocess.run([exe], capture_output=True, text=True, check=True, timeout=300).stdout
    m = re.search(r"max_ulp (\d+) over (\d+)", out)
    assert m and int(m.group(2)) > 8_000_000
    assert int(m.group(1)) <= 1, out          # parity bar is 4 ULP; the fp64 chain achieves 1
    assert "lattice_mismatches 0" in out, out


def test_pow64_algorithm_on_host():
    """sm_pow64.h (double pow: double-double log, table exp) against glibc pow over 3.9 M pairs + the lattice."""
    from simplemath_amd import build
    exe = build.build_host_programs()["pow64_host_check"]
    out = subprocess.run([exe], capture_output=True, text=True, check=True, timeout=300).stdout
    m = re.search(r"max_ulp (\d+) over (\d+)", out)
    assert m and int(m.group(2)) > 3_500_000
    assert int(m.group(1)) <= 1, out
    assert "lattice_mismatches 0" in out, out
    # one exponent of moderate magnitude for a whole array (pow_core_u: the arithmetic trimmed to what a KNOWN exponent needs):
    # <= 1 ULP from glibc, and its true error (against long double powl) stays below 0.65 ULP
    m = re.search(r"scalar_max_ulp (\d+) over (\d+) .* true_err ([0-9.]+) levels (\d+) (\d+) (\d+) lattice_mismatches_scalar (\d+)", out)
    assert m and int(m.group(2)) > 3_500_000, out
    assert int(m.group(1)) <= 1, out
    assert float(m.group(3)) < 0.65, out
    assert int(m.group(4)) == 0 and int(m.group(5)) >= 10 and int(m.group(6)) >= 15, out
    assert int(m.group(7)) == 0, out


def test_pow64_half_integer_exponents_on_host():
    """smpow64::pow_halfint (scalar exponents -8 ... 8 in steps of one half: a double-double product chain, no table)
    against glibc pow over 8.8 M bases from the whole range, negative bases and the zero / infinity / NaN lattice."""
    from simplemath_amd import build
    exe = build.build_host_programs()["pow_halfint_host_check"]
    out = subprocess.run([exe, "300000"], capture_output=True, text=True, check=True, timeout=300).stdout
    m = re.search(r"max_ulp (\d+) over (\d+) identical ([0-9.]+)", out)
    assert m and int(m.group(2)) > 8_000_000
    assert int(m.group(1)) <= 1, out
    assert float(m.group(3)) > 0.999, out
    assert "lattice_mismatches 0" in out, out
