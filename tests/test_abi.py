"""CPU-side checks of the C-ABI library: it loads, exports exactly what
include/smhip.h declares, its host-only shape layer matches the reference's
sm::broadcast, and -- with no GPU in the container -- every compute entry
point fails loudly instead of falling back to the CPU."""
import os
import subprocess

import numpy as np
import pytest

import simplemath_amd as sma
from oracle import oracle as orc
from tests import util
from tests.golden import cases


@pytest.fixture(scope="module")
def lib():
    from simplemath_amd import build
    build.build_lib()
    return sma.load()


def _has_gpu(lib):
    return lib.device_count() > 0


def test_exports_match_header(lib):
    declared = sma.declared_symbols()
    assert len(declared) >= 30
    out = subprocess.run(["nm", "-D", "--defined-only", lib.path], capture_output=True, text=True, check=True).stdout
    exported = {line.split()[-1] for line in out.splitlines() if " T " in line}
    missing = [s for s in declared if s not in exported]
    assert not missing, f"declared in smhip.h but not exported: {missing}"
    extra = sorted(s for s in exported if s.startswith("smhip_") and s not in declared)
    assert not extra, f"exported but undeclared: {extra}"


def test_gfx950_code_object_present(lib):
    with open(lib.path, "rb") as f:
        blob = f.read()
    assert b"amdgcn-amd-amdhsa--gfx950" in blob


def test_version_and_error_strings(lib):
    assert "gfx950" in lib.version()


def test_broadcast_matches_oracle_and_golden(lib, oracle):
    st = util.load_npz("broadcast.npz")
    for c in cases.broadcast_cases():
        _, av, _, bv = cases.broadcast_inputs(c)
        got = lib.broadcast(av.shape, orc.elem_strides(av), bv.shape, orc.elem_strides(bv))
        want = oracle.broadcast(av.shape, orc.elem_strides(av), bv.shape, orc.elem_strides(bv))
        assert got == want
        assert got[0] == list(st[f"{c['id']}/shape"])
        assert got[1] == list(st[f"{c['id']}/strides_a"]) and got[2] == list(st[f"{c['id']}/strides_b"])
    # the reference's only error path: SMUtils.h:76-78
    assert lib.broadcast([2, 3], [3, 1], [4, 3], [3, 1]) is None
    assert "Cannot broadcast shapes" in lib.c.smhip_last_error().decode()
    # rank padding, size-1 -> stride 0
    assert lib.broadcast([4, 1, 3], [3, 3, 1], [5, 1], [1, 1]) == ([4, 5, 3], [3, 0, 1], [0, 1, 0], 60)


def test_is_contiguous(lib, oracle):
    for shape, st in [([2, 3, 4], [12, 4, 1]), ([2, 3, 4], [24, 4, 1]), ([5], [1]), ([5], [2]), ([3, 1], [1, 1]), ([], [])]:
        assert lib.is_contiguous(shape, st) == oracle.is_contiguous(shape, st)


def test_argument_validation_needs_no_gpu(lib):
    # rejected before any device is touched
    with pytest.raises(sma.SmhipError) as e:
        lib.elementwise_raw(7, np.float32, 1, [1], 1, [1], [4], 1)
    assert e.value.code == sma.ERR_INVALID
    with pytest.raises(sma.SmhipError) as e:  # ndim > MAX_NDIM
        lib.elementwise_raw(sma.OP_ADD, np.float32, 1, [1] * 7, 1, [1] * 7, [2] * 7, 1)
    assert e.value.code == sma.ERR_INVALID and "MAX_NDIM" in str(e.value)


def test_no_cpu_fallback(lib):
    """Without a HIP device the product fails loudly (never routes to the oracle)."""
    if _has_gpu(lib):
        pytest.skip("a GPU is present")
    host = np.ones(4, dtype=np.float32)
    for call in (lambda: lib.alloc(1024),
                 lambda: lib.elementwise_raw(sma.OP_ADD, np.float32, 16, [1], 16, [1], [4], 16),
                 lambda: lib.binary_inline(sma.OP_ADD, host, host),   # operands in HOST memory: still no host arithmetic
                 lambda: lib.set_devices(1),
                 lambda: lib.comm_init_rank(1, 0, bytes(128)),
                 lambda: lib.copy_peer(16, 0, 32, 0, 16),
                 lambda: lib.synchronize()):
        with pytest.raises(sma.SmhipError) as e:
            call()
        assert e.value.code == sma.ERR_NO_DEVICE
        assert "no CPU fallback" in str(e.value)


def test_product_does_not_reference_oracle():
    """Nothing under simplemath_amd/ or include/ may import, include or link oracle/."""
    bad = []
    for top in ("simplemath_amd", "include"):
        for dp, _, fs in os.walk(os.path.join(sma.ROOT, top)):
            for f in fs:
                if f.endswith((".py", ".h", ".hip", ".cpp", ".hpp")):
                    text = open(os.path.join(dp, f), errors="ignore").read()
                    if "sm_oracle" in text or "libsmoracle" in text or "from oracle" in text or "import oracle" in text:
                        bad.append(os.path.join(dp, f))
    assert not bad, bad
    out = subprocess.run(["ldd", sma.LIB_PATH], capture_output=True, text=True).stdout
    assert "smoracle" not in out and "smref" not in out
