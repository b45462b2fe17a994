import os
import sys

import pytest

# the single-cohort command lines (geneDriver / targetDriver / elementDriver) promise not to load PyTorch: every CLI child of the
# tests checks it (scripts/DigDriver.py)
os.environ.setdefault("DIG_CLI_ASSERT_NO_TORCH", "1")
os.environ.setdefault("DIG_NN_TUNE", "0")       # (GEMM tuning of NNTrainer: seconds per process and run-to-run choices; one test switches it on)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def oracle_clib():
    """The C oracle (oracle/libdig_oracle.so), built on demand with gcc.  Test infrastructure."""
    import ctypes
    import subprocess
    so = os.path.join(ROOT, "oracle", "libdig_oracle.so")
    src = os.path.join(ROOT, "oracle", "dig_oracle.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
    return ctypes.CDLL(so)


def rel_close(got, want, rtol=1e-6, floor=1e-250):
    """The tolerance contract: |d|/p <= rtol for p >= floor; below the floor both must be
    below the floor (scipy itself is not self-consistent there); NaN positions must match."""
    import numpy as np
    got, want = np.asarray(got, float), np.asarray(want, float)
    assert got.shape == want.shape
    nan_w = np.isnan(want)
    assert (np.isnan(got) == nan_w).all(), "NaN positions differ"
    ok = ~nan_w
    big = ok & (np.abs(want) >= floor)
    with np.errstate(all="ignore"):
        rel = np.abs(got[big] - want[big]) / np.abs(want[big])
    assert rel.size == 0 or rel.max() <= rtol, "max rel err %g at %d" % (rel.max(), int(np.argmax(rel)))
    small = ok & ~big
    assert (np.abs(got[small]) < floor * 1.0001).all(), "tiny reference value but large result"
    return True
