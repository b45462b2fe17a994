"""CPU-only checks: the C-ABI library loads and exports every symbol include/dig_hip.h declares,
host-side index construction matches the reference goldens, and the product path fails loudly
(no CPU fallback) when no GPU is present."""
import json
import os
import re

import numpy as np
import pytest

from conftest import GOLDEN, ROOT
from digdriver_amd import _lib, engine


def _declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "dig_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(dig_[a-z0-9_]+)\s*\(", hdr)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    declared = _declared_symbols()
    assert len(declared) >= 20
    for sym in declared:
        assert hasattr(lib, sym), "libdig_hip.so does not export %s" % sym
    assert sorted(_lib.EXPORTED_SYMBOLS) == declared, "python binding table and header disagree"
    assert lib.dig_abi_version() == _lib.ABI_VERSION == 6


def test_library_abi_version_matches_header_and_binding():
    import re
    from digdriver_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "dig_hip.h")).read()
    declared = int(re.search(r"#define\s+DIG_ABI_VERSION\s+(\d+)", hdr).group(1))
    assert declared == _lib.ABI_VERSION == _lib.load().dig_abi_version()


def test_product_never_imports_oracle_or_scipy():
    """The product path must not route through the oracle or any CPU implementation."""
    pkg = os.path.join(ROOT, "digdriver_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if not (f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")) or f == "Makefile"):
                continue
            txt = open(os.path.join(dirpath, f)).read()
            assert "oracle" not in txt, "%s mentions the oracle" % f
            if f.endswith(".py"):
                for line in txt.splitlines():
                    assert not re.match(r"\s*(import|from)\s+scipy", line), "%s imports scipy" % f


def test_ideal_overlaps_host_matches_reference_goldens():
    cases = json.load(open(os.path.join(GOLDEN, "overlaps_golden.json")))
    for c in cases:
        w = c["window"]
        starts, ends = c["intervals"]
        hi = (max(ends) // w + 2)
        bin_start = np.arange(0, hi * w, w, dtype=np.int64)
        bin_chrom = np.full(len(bin_start), c["chrom"], np.int32)
        ptr, idx = engine.ideal_overlaps([c["chrom"]], [0, len(starts)], starts, ends, w, bin_chrom, bin_start)
        got = [int(bin_start[i]) for i in idx]
        assert got == [o[1] for o in c["overlaps"]], c
        assert ptr.tolist() == [0, len(got)]


def test_ideal_overlaps_host_batch_and_missing_bin():
    d = np.load(os.path.join(GOLDEN, "accumulate_golden.npz"))
    bs, be = d["block_starts"], d["block_ends"]
    nblk = (bs >= 0).sum(axis=1)
    blk_ptr = np.concatenate([[0], np.cumsum(nblk)])
    ptr, idx = engine.ideal_overlaps(d["elt_chrom"], blk_ptr, bs[bs >= 0], be[be >= 0], int(d["window"]),
                                     d["bin_idx"][:, 0], d["bin_idx"][:, 1])
    ovp = d["elt_overlap_bins"]
    assert np.array_equal(ptr, np.concatenate([[0], np.cumsum((ovp >= 0).sum(axis=1))]))
    assert np.array_equal(idx, ovp[ovp >= 0])
    # a block beyond the bin table is an error, like the reference's KeyError at df.loc (genic_driver_tools.py:265)
    with pytest.raises(_lib.DigHipError, match="not in the bin table"):
        engine.ideal_overlaps([1], [0, 1], [10 ** 9], [10 ** 9 + 5], int(d["window"]), d["bin_idx"][:, 0], d["bin_idx"][:, 1])
    # empty input
    ptr, idx = engine.ideal_overlaps(np.zeros(0, np.int32), [0], [], [], 10000, d["bin_idx"][:, 0], d["bin_idx"][:, 1])
    assert ptr.tolist() == [0] and len(idx) == 0


def test_no_cpu_fallback_without_gpu():
    if _lib.device_count() > 0:
        pytest.skip("a GPU is present")
    from digdriver_amd.sequence_model import nb_model
    with pytest.raises(_lib.DigHipError):
        nb_model.nb_pvalue_greater_midp(np.array([1.0, 2.0]), np.array([2.0, 2.0]), np.array([0.5, 0.5]))
    with pytest.raises(_lib.DigHipError):
        _lib.require_device()


def test_bad_arguments_are_reported_through_dig_last_error():
    lib = _lib.load()
    rc = lib.dig_accumulate_elements_host(None, None, None, None, None, None, None, None, 3, None, None, None, None,
                                          None, None, None, None, None, None, None, 1, 1, 1, 0)
    assert rc == -1
    assert "n_class" in _lib.last_error()
