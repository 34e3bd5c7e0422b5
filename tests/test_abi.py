"""CPU-side checks of the C-ABI library: it loads, exports exactly what
include/smhip.h declares, its host-only shape layer matches the reference's
sm::broadcast, and -- with no GPU in the container -- every compute entry
point fails loudly instead of falling back to the CPU."""
import os
import subprocess
import sys

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


def test_residency_rule_for_the_read_hint(lib):
    """DESIGN.md section 3 'Cold operands': operands the library has not touched within the last 192 MiB of its own traffic are
    read non-temporally at any size; operands that repeat, or were just written, are read through the caches (round 2's
    rule for <= 256 MiB of reads); reads above the Infinity Cache are always non-temporal; results are kept (`sc1`) for
    footprints of 40-256 MiB.  smhip_policy_probe does the host-side arithmetic on pointer values: no GPU involved."""
    MiB = 1 << 20
    NT, KEEP = 1, 2
    base = 0x7000_0000_0000
    a, out = base, base + 64 * MiB
    assert lib.policy_probe(a, 64 * MiB, 0, 0, out, 64 * MiB) == NT | KEEP      # first call: cold operand, footprint 128 MiB
    assert lib.policy_probe(a, 64 * MiB, 0, 0, out, 64 * MiB) == KEEP           # replayed: warm
    assert lib.policy_probe(out, 64 * MiB, 0, 0, a, 64 * MiB) == KEEP           # a chain: reads what the previous launch wrote
    # a view inside a warm array is warm; an array next to it is not
    assert lib.policy_probe(a + 8 * MiB, 16 * MiB, 0, 0, base + 512 * MiB, 16 * MiB) & NT == 0
    assert lib.policy_probe(base + 1024 * MiB, 16 * MiB, 0, 0, base + 1100 * MiB, 16 * MiB) & NT == NT
    # other work passes (> 192 MiB of library traffic): the array has gone cold again
    for k in range(4):
        lib.policy_probe(base + (2048 + 200 * k) * MiB, 64 * MiB, 0, 0, base + (2048 + 200 * k + 100) * MiB, 64 * MiB)
    assert lib.policy_probe(a, 64 * MiB, 0, 0, out, 64 * MiB) == NT | KEEP
    # two read streams: the hint follows the majority of the read bytes
    b = base + 4096 * MiB
    assert lib.policy_probe(a, 64 * MiB, b, 64 * MiB, out, 64 * MiB) & NT == 0   # a warm (just touched), b cold: half -> plain
    assert lib.policy_probe(a, 16 * MiB, base + 5000 * MiB, 64 * MiB, out, 16 * MiB) & NT == NT   # mostly cold
    # above the Infinity Cache reads are non-temporal whatever their history; stores then too
    big = base + 8192 * MiB
    assert lib.policy_probe(big, 300 * MiB, 0, 0, big + 512 * MiB, 300 * MiB) == NT
    assert lib.policy_probe(big, 300 * MiB, 0, 0, big + 512 * MiB, 300 * MiB) == NT
    # small launches: below the keep-store floor, and operands under 2 MiB are not tracked (they live in the L2s)
    assert lib.policy_probe(base + 9000 * MiB, 1 * MiB, 0, 0, base + 9100 * MiB, 1 * MiB) == 0


def test_residency_rule_can_be_switched_off():
    code = ("import simplemath_amd as s; l = s.load(); M = 1 << 20; "
            "print(l.policy_probe(0x700000000000, 64 * M, 0, 0, 0x700010000000, 64 * M))")
    env = dict(os.environ, SMHIP_RESIDENCY="off", PYTHONPATH=sma.ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode == 0 and r.stdout.strip() == "2", r.stdout + r.stderr   # round 2's size-only rule: plain reads, kept stores


def test_policy_peek_does_not_touch(lib):
    """looking at the residency rule must not change it -- smhip_policy_peek answers like smhip_policy_probe
    without recording the spans; only the probe (like a launch) makes an operand warm."""
    MiB = 1 << 20
    NT, KEEP = 1, 2
    base = 0x7100_0000_0000
    a, out = base, base + 64 * MiB
    assert lib.policy_peek(a, 64 * MiB, 0, 0, out, 64 * MiB) == NT | KEEP
    assert lib.policy_peek(a, 64 * MiB, 0, 0, out, 64 * MiB) == NT | KEEP       # still cold: peeking recorded nothing
    assert lib.policy_probe(a, 64 * MiB, 0, 0, out, 64 * MiB) == NT | KEEP      # the probe records ...
    assert lib.policy_peek(a, 64 * MiB, 0, 0, out, 64 * MiB) == KEEP            # ... and now it is warm
    assert lib.policy_peek(out, 64 * MiB, 0, 0, a, 64 * MiB) == KEEP
