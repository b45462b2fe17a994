#!/opt/conda/bin/python3.9
"""Writes a training file the way the reference's extractor does (scripts/DataExtractor.py:424-426: x_data float64,
compression='gzip', h5py's automatic chunking; values round(x, 2) * 100, :220) with the GENUINE h5py 3.3 of the image's second
interpreter -- the streaming loader (region_model/data_aux/dataset_generator.py::load_track_matrix) is tested on it.
Too large to commit (tens of MB): tests/test_h5_io.py generates it on the fly and skips where that interpreter is absent.

    /opt/conda/bin/python3.9 tests/golden/make_xdata_fixture.py OUT.h5 N L T [n_fractional]
"""
import sys

sys.dont_write_bytecode = True

import h5py            # noqa: E402
import numpy as np     # noqa: E402

out, N, L, T = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
n_frac = int(sys.argv[5]) if len(sys.argv) > 5 else 0
rng = np.random.default_rng(11)
with h5py.File(out, "w") as f:
    d = f.create_dataset("x_data", shape=(N, L, T), maxshape=(N, L, None), dtype=float, data=None, compression="gzip")
    step = max(1, N // 8)
    for lo in range(0, N, step):
        hi = min(N, lo + step)
        d[lo:hi] = np.round(rng.uniform(0, 1, (hi - lo, L, T)), 2) * 100
    for j in range(n_frac):                                   # values int16 cannot carry, late in the file
        d[N - 1 - j, 3, 5] = 12.5
    f.create_dataset("idx", data=np.stack([np.ones(N, np.int32), np.arange(N, dtype=np.int32) * 10000,
                                           (np.arange(N, dtype=np.int32) + 1) * 10000], axis=1), compression="gzip")
    print("chunks", d.chunks)
