"""pytest configuration: the `gpu` marker and shared fixtures.

CPU suite (`-m "not gpu"`): oracle vs the reference's golden vectors, host
logic, C-ABI symbol/export checks.  GPU suite (`-m gpu`): the HIP path, called
through the C-ABI, against the oracle and the committed fixtures.
"""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with gpurun)")


def _stale(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources if os.path.exists(s))


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as orc
    odir = os.path.join(ROOT, "oracle")
    if _stale(orc.ORACLE_SO, [os.path.join(odir, "sm_oracle.c"), os.path.join(odir, "sm_oracle.h")]):
        orc.build(ref=False)
    return orc.Oracle()


@pytest.fixture(scope="session")
def reference():
    """The compiled reference (only where oracle/_ref/libsmref.so exists)."""
    from oracle import oracle as orc
    if not orc.Reference.available() and os.path.isdir("/root/reference/include"):
        orc.build(ref=True)  # the reference compiled from where it lies; outputs only under oracle/_ref/
    if not orc.Reference.available():
        pytest.skip("oracle/_ref/libsmref.so not built (needs /root/reference)")
    return orc.Reference()


@pytest.fixture(scope="session")
def smhip():
    """The product C-ABI (libsmhip.so) on a real GPU; fails loudly if it is missing."""
    import simplemath_amd as sma
    return sma.load()
