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
    assert lib.dig_abi_version() == _lib.ABI_VERSION == 12


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


def test_native_result_writer_writes_the_bytes_pandas_writes(tmp_path):
    """mapfile.write_results_tsv (dig_write_tsv_host: std::to_chars digits + the layout rules of Python's float repr) against
    DataFrame.to_csv(sep="\\t") -- what DigDriver.py writes (DigDriver.py:115-118) -- byte for byte: p-values down to the
    subnormals, whole numbers, 1e-4 / 1e16 switch-overs, NaN, infinities, signed zeros, integers, bools; and the fall-back to
    pandas for frames the writer does not cover."""
    import numpy as np
    import pandas as pd
    from digdriver_amd.io import mapfile
    rng = np.random.default_rng(0)
    n = 20_011
    edge = np.array([0.0, -0.0, 5e-324, 2.2250738585072014e-308, 1e-5, 9.999999999999999e-05, 1e-4, 0.00012345, 0.1, 1 / 3, 1.0, 12.0, 123456.0,
                     1e15, 9999999999999998.0, 1e16, 1.2345e16, 1e22, 1.7976931348623157e308, np.inf, -np.inf, np.nan, -1.5e-7, -2.5, 100.0, 1e-300])
    cols = {
        'ELT_SIZE': rng.integers(200, 9000, n), 'FLAG': rng.uniform(size=n) < 0.1, 'MU': rng.gamma(9, 3, n),
        'PVAL': 10.0 ** rng.uniform(-320, 0, n), 'Z': np.where(rng.uniform(size=n) < 0.01, np.nan, rng.normal(size=n)),
        'WHOLE': rng.integers(0, 50, n).astype(float), 'BIG': rng.uniform(1e15, 1e18, n), 'SMALL': rng.uniform(1e-6, 1e-3, n),
        'BITS': rng.integers(0, 2 ** 63 - 1, n, dtype=np.int64).view(np.float64), 'OBS': rng.integers(-5, 50, n).astype(np.int32),
        'EDGE': np.resize(edge, n)}
    df = pd.DataFrame(cols, index=pd.Index(['ELT%06d' % i for i in range(n)], name='ELT'))
    a, b = tmp_path / "native.txt", tmp_path / "pandas.txt"
    df.to_csv(str(b), header=True, index=True, sep="\t")
    for threads in (8, 1, 3):                             # several threads with their own chunks; one thread streaming chunk by chunk
        mapfile.write_results_tsv(df, str(a), threads=threads)
        assert a.read_bytes() == b.read_bytes(), threads
    back = pd.read_csv(str(a), sep="\t", index_col=0, float_precision="round_trip")
    assert np.array_equal(back.PVAL.values, df.PVAL.values) and np.array_equal(back.BITS.values, df.BITS.values, equal_nan=True)
    # an unnamed integer index, no rows, no columns
    for frame in (df.reset_index(drop=True).iloc[:7], df.iloc[:0], df.iloc[:5, :0]):
        mapfile.write_results_tsv(frame, str(a))
        frame.to_csv(str(b), header=True, index=True, sep="\t")
        assert a.read_bytes() == b.read_bytes()
    # not covered: a string column, a label with a tab -> pandas writes them
    for frame in (df.iloc[:5].assign(NAME=list("abcde")), df.iloc[:3].rename(index={'ELT000001': 'a\tb'})):
        mapfile.write_results_tsv(frame, str(a))
        frame.to_csv(str(b), header=True, index=True, sep="\t")
        assert a.read_bytes() == b.read_bytes()
    # indexes the native path must leave to pandas: an object index of integers FOLLOWED by the equal index of
    # floats (Index.equals says "same": the label cache must not hand 1 / 2 / 3 to a frame that prints 1.0 / 2.0 / 3.0), a None /
    # NaN label, dates, a float index, a nullable integer column with a missing value
    small = df.iloc[:3, :4]
    frames = [small.set_axis(pd.Index([1, 2, 3], dtype=object)), small.set_axis(pd.Index([1.0, 2.0, 3.0], dtype=object)),
              small.set_axis(pd.Index(['a', None, 'c'], dtype=object)), small.set_axis(pd.Index(['a', np.nan, 'c'], dtype=object)),
              small.set_axis(pd.to_datetime(['2020-01-01', '2020-01-02', '2020-01-03'])), small.set_axis(pd.Index([0.5, 1.5, 2.5])),
              small.assign(NA_INT=pd.array([1, None, 3], dtype="Int64")), small.set_axis(pd.Index([7, 8, 9], name='ID'))]
    for frame in frames:
        mapfile.write_results_tsv(frame, str(a))
        frame.to_csv(str(b), header=True, index=True, sep="\t")
        assert a.read_bytes() == b.read_bytes(), frame.index
