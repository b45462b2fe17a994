"""HDF5 mutation-map I/O without h5py / PyTables in the product: digdriver_amd/io/h5lite.py (container),
pandas_fixed.py (DataFrame.to_hdf "fixed" layout), mapfile.py (the map API the drivers use).

Pinned three ways:
  * tests/golden/*_genuine.h5: files written by h5py 3.3 (HDF5 1.10.6) and by pandas 2.3.3 + PyTables 3.6.1 in the build
    container (generator: tests/golden/make_h5_fixture.py), with their content as JSON next to them;
  * the writer must reproduce the genuine frame groups member for member (names, types, shapes, attribute forms, values);
  * when the image's second interpreter (h5py) and h5dump are present, files written here are read back by them and
    files written by h5py there are read here (skipped LOUDLY otherwise).
"""
import gzip
import json
import os
import shutil
import subprocess
import sys

import numpy as np
import pandas as pd
import pytest

from conftest import ROOT
from digdriver_amd.io import h5lite, mapfile, pandas_fixed

GOLD = os.path.join(ROOT, "tests", "golden")
PY39 = "/opt/conda/bin/python3.9"
H5DUMP = "/opt/conda/bin/h5dump"


@pytest.fixture(scope="module")
def expected():
    return json.load(open(os.path.join(GOLD, "h5_fixture_expected.json")))


@pytest.fixture(scope="module")
def pretrained(tmp_path_factory):
    out = tmp_path_factory.mktemp("h5") / "pretrained_genuine.h5"
    with gzip.open(os.path.join(GOLD, "pretrained_genuine.h5.gz"), "rb") as src, open(out, "wb") as dst:
        shutil.copyfileobj(src, dst)
    return str(out)


def _frame_from_json(e):
    cols = {c: np.asarray(e["data"][c]).astype(object if dt == "object" else dt) for c, dt in zip(e["columns"], e["dtypes"])}
    return pd.DataFrame(cols, index=pd.Index(e["index"], name=e["index_name"]))


def test_reads_map_written_by_h5py_and_pytables(pretrained, expected):
    """The reference's own writers: h5py datasets + gzip + attributes (DigPretrain.py:82-89,156-177) and
    DataFrame.to_hdf frames (:96,207-208,234,266), read the way transfer_tools.load_pretrained_model does (:11-19)."""
    e = expected["pretrained"]
    assert mapfile.read_attrs(pretrained) == e["attrs"]
    idx = mapfile.read_array(pretrained, "idx")
    assert idx.dtype == np.int32 and np.array_equal(idx, np.array(e["idx"]))
    mapp = mapfile.read_array(pretrained, "mappability")
    assert mapp.dtype == np.float32 and np.array_equal(mapp, np.array(e["mappability"], np.float32))
    for key, fe in e["frames"].items():
        df = mapfile.read_frame(pretrained, key)
        pd.testing.assert_frame_equal(df, _frame_from_json(fe), check_exact=True)
        assert [str(t) for t in df.dtypes] == fe["dtypes"], key
    assert mapfile.has_key(pretrained, "genic_model") and not mapfile.has_key(pretrained, "no_such_key")
    with pytest.raises(KeyError):
        mapfile.read_frame(pretrained, "no_such_key")


def test_reads_element_data_written_by_h5py(expected):
    """sequence_tools.py:460-478,639-641: substitution_idx, window_{w}/full_window_si_*, per-element groups + attribute."""
    p = os.path.join(GOLD, "element_data_genuine.h5")
    e = expected["element_data"]
    assert [s.decode() for s in mapfile.read_array(p, "substitution_idx")] == e["substitution_idx"]
    assert np.array_equal(mapfile.read_array(p, "window_10000/full_window_si_values"), np.array(e["full_window_si_values"]))
    assert np.array_equal(mapfile.read_array(p, "window_10000/full_window_si_index"), np.array(e["full_window_si_index"]))
    assert sorted(mapfile.list_keys(p, "window_10000/enhancers")) == sorted(e["elements"])
    for name, ee in e["elements"].items():
        base = "window_10000/enhancers/%s" % name
        assert np.array_equal(mapfile.read_array(p, base + "/L_counts"), np.array(ee["L_counts"]))
        assert np.array_equal(mapfile.read_array(p, base + "/region_counts"), np.array(ee["region_counts"]))
        assert np.array_equal(mapfile.read_attrs(p, base)["overlaps"], np.array(ee["overlaps"]))


def _signature(g):
    d = {"@" + k: type(v).__name__ + ":" + repr(v) for k, v in g.attrs.items()}
    for k, c in g.children.items():
        payload = c.data.load().tolist() if isinstance(c.data, h5lite.VLenObject) else c.data.tolist()
        d[k] = (str(c.dtype), tuple(c.shape), c.bitfield, {a: type(v).__name__ + ":" + repr(v) for a, v in c.attrs.items()}, payload)
    return d


def test_frame_writer_reproduces_pytables_layout(pretrained, expected, tmp_path):
    """Every frame of the genuine file, rebuilt from its JSON content and written by THIS package, must come out as the
    same HDF5 group: member names, datatypes (incl. H5T_STD_B8 booleans, pickled object blocks), shapes, block order,
    attribute names and storage forms (fixed UTF-8 strings, NULL-dataspace empty strings, pickled index names)."""
    out = str(tmp_path / "ours.h5")
    for key, fe in expected["pretrained"]["frames"].items():
        mapfile.write_frame(out, key, _frame_from_json(fe))
    theirs, ours = h5lite.read_tree(pretrained), h5lite.read_tree(out)
    for key in expected["pretrained"]["frames"]:
        assert _signature(ours[key]) == _signature(theirs[key]), key
    assert {k: str(v) for k, v in ours.attrs.items()} == {"CLASS": "GROUP", "VERSION": "1.0", "TITLE": "", "PYTABLES_FORMAT_VERSION": "2.1"}


def test_map_api_round_trip_and_update(tmp_path):
    p = str(tmp_path / "cohort.Pretrained.h5")
    rng = np.random.default_rng(5)
    rp = pd.DataFrame({"CHROM": [1, 1, 2], "START": [0, 10000, 0], "Y_PRED": rng.gamma(9, 3, 3), "FLAG": [True, False, False],
                       "NOTE": ["a", "bé", ""]}, index=["chr1:0-10000", "chr1:10000-20000", "chr2:0-10000"])
    mapfile.write_array(p, "idx", np.arange(9, dtype=np.int32).reshape(3, 3), compression="gzip")
    mapfile.write_attrs(p, cohort_name="Panc-AdenoCA", N_SAMPLES=232, mappability_threshold=0.5)
    mapfile.write_frame(p, "region_params", rp)
    mapfile.write_frame(p, "window_10000/sub/frame", rp.iloc[:2])            # nested key
    mapfile.write_array(p, "names", np.array(["x", "yy"]))
    mapfile.write_attrs(p, N_MUT_TRAIN=np.int64(12345))                      # update keeps everything else
    pd.testing.assert_frame_equal(mapfile.read_frame(p, "region_params"), rp)
    pd.testing.assert_frame_equal(mapfile.read_frame(p, "window_10000/sub/frame"), rp.iloc[:2])
    assert mapfile.read_attrs(p) == {"cohort_name": "Panc-AdenoCA", "N_SAMPLES": 232, "mappability_threshold": 0.5, "N_MUT_TRAIN": 12345}
    assert mapfile.read_array(p, "idx").dtype == np.int32 and mapfile.read_array(p, "names").astype(str).tolist() == ["x", "yy"]
    assert np.array_equal(mapfile.read_array_rows(p, "idx", 1, 3), np.arange(9).reshape(3, 3)[1:3])
    assert mapfile.array_shape(p, "idx") == (3, 3)
    empty = rp.iloc[:0]
    mapfile.write_frame(p, "empty", empty)
    assert len(mapfile.read_frame(p, "empty")) == 0 and list(mapfile.read_frame(p, "empty").columns) == list(rp.columns)


def test_large_groups_and_lazy_rows(tmp_path):
    """Thousands of per-element groups (multi-level group B-tree) and windowed reads of a matrix."""
    p = str(tmp_path / "big.h5")
    root = h5lite.Group()
    x = np.arange(5000 * 7, dtype=np.float64).reshape(5000, 7)
    root.set("x_data", h5lite.Dataset(x))
    for i in range(2600):
        root.set("window_1000/elts/E%05d/L_counts" % i, h5lite.Dataset(np.full(3, i, np.int64), {"overlaps": np.array([[1, i, i + 1]])}))
    h5lite.write_tree(p, root)
    t = h5lite.read_tree(p, lazy=True)
    assert len(t["window_1000/elts"].children) == 2600
    assert t["window_1000/elts/E02599/L_counts"].data.tolist() == [2599] * 3
    assert np.array_equal(t["x_data"].read_rows(4990, 6000), x[4990:]) and t["x_data"].shape == (5000, 7)
    if os.path.exists(H5DUMP):
        r = subprocess.run([H5DUMP, "-H", p], capture_output=True, text=True)
        assert r.returncode == 0 and r.stdout.count("GROUP \"E0") == 2600, r.stderr[-500:]


_H5PY_CHILD = r"""
import json, pickle, sys
import numpy as np, h5py
mode, path = sys.argv[1], sys.argv[2]
if mode == "read":
    out = {}
    with h5py.File(path, "r") as f:
        out["attrs"] = {k: ("" if type(v).__name__ == "Empty" else v.decode() if isinstance(v, bytes) else (v.tolist() if hasattr(v, "tolist") else v))
                        for k, v in f.attrs.items()}
        out["idx"] = f["idx"][:].tolist(); out["idx_dtype"] = str(f["idx"].dtype)
        g = f["region_params"]
        out["axis0"] = [s.decode() for s in g["axis0"][:]]
        out["axis1"] = [s.decode() for s in g["axis1"][:]]
        out["nblocks"] = int(g.attrs["nblocks"]); out["pandas_type"] = g.attrs["pandas_type"].decode() if isinstance(g.attrs["pandas_type"], bytes) else str(g.attrs["pandas_type"])
        out["blocks"] = []
        for b in range(out["nblocks"]):
            v = g["block%d_values" % b]
            items = [s.decode() for s in g["block%d_items" % b][:]]
            vals = pickle.loads(v[0].tobytes()).tolist() if v.dtype.kind == "O" else v[:].tolist()
            out["blocks"].append([items, vals])
        out["n_elts"] = len(f["window_1000/elts"]); out["L7"] = f["window_1000/elts/E0007/L"][:].tolist()
        out["ov7"] = f["window_1000/elts/E0007"].attrs["overlaps"].tolist()
    print(json.dumps(out))
else:
    rng = np.random.default_rng(3)
    with h5py.File(path, "w") as f:
        f.attrs["cohort_name"] = "Läng".encode("utf-8"); f.attrs["label"] = "text"; f.attrs["N"] = 7; f.attrs["v"] = np.arange(3.0)
        f.create_dataset("idx", data=rng.integers(0, 1 << 20, (3000, 3)).astype(np.int32), compression="gzip")
        f.create_dataset("x_data", data=np.round(rng.random((300, 10, 6)), 2) * 100, chunks=(16, 10, 6), compression="gzip", shuffle=True)
        f.create_dataset("flags", data=rng.random(40) < 0.5)
        f.create_dataset("scalar", data=2.5)
        g = f.create_group("a/b")
        g.create_dataset("names", data=np.array([b"chr1", b"chr22"]))
        np.savez(path + ".npz", idx=f["idx"][:], x=f["x_data"][:], flags=f["flags"][:])
"""


@pytest.mark.skipif(not os.path.exists(PY39), reason="LOUD SKIP: %s (the interpreter with h5py) is not in this image -- the "
                    "cross-check of h5lite against the HDF5 library cannot run here" % PY39)
def test_cross_check_with_h5py(tmp_path):
    probe = subprocess.run([PY39, "-c", "import h5py"], capture_output=True)
    if probe.returncode != 0:
        pytest.skip("LOUD SKIP: h5py does not import under %s" % PY39)
    child = tmp_path / "child.py"
    child.write_text(_H5PY_CHILD)
    # (1) written here, read by h5py
    p = str(tmp_path / "ours.Pretrained.h5")
    rp = pd.DataFrame({"CHROM": [1, 2], "Y_PRED": [1.5, 2.5], "FLAG": [True, False], "S": ["+", "-"]}, index=["chr1:0-10", "chr2:0-10"])
    mapfile.write_array(p, "idx", np.array([[1, 0, 10], [2, 0, 10]], np.int32))
    mapfile.write_attrs(p, cohort_name="Kidney-RCC", N_SAMPLES=144, thr=0.5)
    mapfile.write_frame(p, "region_params", rp)
    def put(root):
        for i in range(40):
            root.set("window_1000/elts/E%04d/L" % i, h5lite.Dataset(np.full(4, i, np.int64)))
            root["window_1000/elts/E%04d" % i].attrs["overlaps"] = np.array([[1, 1000 * i, 1000 * i + 1000]])
    h5lite.update(p, put)
    r = subprocess.run([PY39, str(child), "read", p], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    got = json.loads(r.stdout.strip().split("\n")[-1])
    assert got["attrs"]["cohort_name"] == "Kidney-RCC" and got["attrs"]["N_SAMPLES"] == 144 and got["attrs"]["thr"] == 0.5
    assert got["idx"] == [[1, 0, 10], [2, 0, 10]] and got["idx_dtype"] == "int32"
    assert got["axis0"] == ["CHROM", "Y_PRED", "FLAG", "S"] and got["axis1"] == ["chr1:0-10", "chr2:0-10"]
    assert got["pandas_type"] == "frame" and got["nblocks"] == 4
    blocks = {tuple(items): vals for items, vals in got["blocks"]}
    assert blocks[("CHROM",)] == [[1], [2]] and blocks[("Y_PRED",)] == [[1.5], [2.5]] and blocks[("S",)] == [["+"], ["-"]]
    assert [bool(v[0]) for v in blocks[("FLAG",)]] == [True, False]
    assert got["n_elts"] == 40 and got["L7"] == [7, 7, 7, 7] and got["ov7"] == [[1, 7000, 8000]]
    # (2) written by h5py (gzip / shuffle / chunked, vlen and byte-string attributes, booleans as enum), read here
    q = str(tmp_path / "theirs.h5")
    r = subprocess.run([PY39, str(child), "write", q], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    want = np.load(q + ".npz")
    at = mapfile.read_attrs(q)
    assert (at["cohort_name"], at["label"], at["N"]) == ("Läng", "text", 7) and np.array_equal(at["v"], [0.0, 1.0, 2.0])
    assert np.array_equal(mapfile.read_array(q, "idx"), want["idx"]) and np.array_equal(mapfile.read_array(q, "x_data"), want["x"])
    assert np.array_equal(mapfile.read_array_rows(q, "x_data", 30, 77), want["x"][30:77])
    assert np.array_equal(mapfile.read_array(q, "flags"), want["flags"]) and mapfile.read_array(q, "flags").dtype == bool
    assert mapfile.read_array(q, "a/b/names").tolist() == [b"chr1", b"chr22"] and float(mapfile.read_array(q, "scalar")) == 2.5


def test_unsupported_structures_fail_loudly(tmp_path):
    p = tmp_path / "not.h5"
    p.write_bytes(b"this is not an HDF5 file" * 10)
    with pytest.raises(mapfile.MapFileError):
        mapfile.read_array(str(p), "idx")
    with pytest.raises(mapfile.MapFileError):
        mapfile.read_frame(str(tmp_path / "missing.h5"), "region_params")
    g = h5lite.Group()
    g.set("t", h5lite.Dataset(np.zeros(3)))
    h5lite.write_tree(str(tmp_path / "x.h5"), g)
    with pytest.raises(mapfile.MapFileError):
        mapfile.read_frame(str(tmp_path / "x.h5"), "t")          # a dataset, not a frame group


def test_kfold_results_on_reference_fold_files(expected):
    """region_model_tools.kfold_results (reference: region_model_tools.py:64-193) on fold files with the group layout of
    GPTrainer.save_results (gp_trainer.py:206-245), written by h5py; the expected frame is the output of the REFERENCE's
    own kfold_results on the same files (computed by make_h5_fixture.py under the interpreter that has h5py)."""
    from digdriver_amd.region_model import region_model_tools
    got = region_model_tools.kfold_results(os.path.join(GOLD, "kfold_genuine"), "Synthetic-Cohort")
    want = _frame_from_json(expected["kfold_results"])
    want.index.name = "Region"
    pd.testing.assert_frame_equal(got, want, check_exact=False, rtol=1e-15, atol=0)
    assert got.FLAG.dtype == bool and got.CHROM.dtype.kind == "i"
    p = os.path.join(GOLD, "kfold_genuine", "gp_results_fold_0.h5")
    assert set(mapfile.read_attrs(p, "Synthetic-Cohort/held-out/1")) == {"R2", "loss"}


def test_compute_pretrained_matches_reference_method():
    """GPTrainer.compute_pretrained (gp_trainer.py:247-261) of the reference, run with h5py on the committed fold files
    (tests/golden/make_pretrained_golden.py), against the map-file version the single-split route uses."""
    from digdriver_amd.region_model.trainers import gp_trainer
    with open(os.path.join(GOLD, "compute_pretrained_golden.json")) as f:
        cases = json.load(f)
    assert len(cases) == 6
    for key, want in cases.items():
        fn, runs = key.split(":")
        got = gp_trainer.compute_pretrained(os.path.join(GOLD, "kfold_genuine", fn), want["cohort"], int(runs))
        for g, name in zip(got, ("chr_locs", "mapps", "quants", "y_true", "means", "stds")):
            np.testing.assert_array_equal(np.asarray(g, float), np.asarray(want[name], float), err_msg=key + " " + name)


def test_store_pretrained_accumulates_sorted_frame(tmp_path):
    """OutputGenerator.store_pretrained (mutations_main.py:148-172): the held-out windows of every rerun join the label's
    frame, which is written position-sorted into <label>.Pretrained.h5:region_params with the reference's columns; the
    accuracy file holds the squared Pearson r of the unflagged rows."""
    from digdriver_amd.io import mapfile
    from digdriver_amd.region_model import mutations_main as mm
    from digdriver_amd.region_model.predict import r2_score
    args = mm.get_cmd_arguments("-c COH -d none -o %s -gp 2 -re 2" % tmp_path)
    og = mm.OutputGenerator(args, "cpu", str(tmp_path))
    rng = np.random.default_rng(0)
    blocks = []
    for fold in range(2):
        n = 7
        locs = np.stack([rng.integers(1, 4, n), rng.integers(0, 50, n) * 10000, np.zeros(n, int)], 1)
        locs[:, 2] = locs[:, 1] + 10000
        y, m, s = rng.poisson(20, n).astype(float), rng.uniform(5, 40, n), rng.uniform(1, 5, n)
        mp, q = rng.uniform(0.7, 1, n), rng.uniform(0, 1, n)
        og.store_pretrained("COH", locs, mp, q, y, m, s, fold, is_flagged=(fold == 1))
        blocks.append((locs, y, m, s, mp, q, fold))
    df = mapfile.read_frame(str(tmp_path / "COH.Pretrained.h5"), "region_params")
    assert list(df.columns) == mm.OutputGenerator.pretrained_cols and len(df) == 14
    key = df.CHROM.values * 1e9 + df.START.values
    assert (np.diff(key) >= 0).all()                                      # sorted by CHROM, START
    assert set(df.FOLD.unique()) == {0.0, 1.0} and set(df.FLAG[df.FOLD == 1].unique()) == {1.0}
    locs, y, m = blocks[0][0], blocks[0][1], blocks[0][2]
    want_acc = r2_score(y, m)                                              # only fold 0 is unflagged
    assert abs(float(open(tmp_path / "COH_pretrained_accuracy.txt").read()) - want_acc) < 1e-12
    row = df[(df.CHROM == locs[0, 0]) & (df.START == locs[0, 1]) & (df.FOLD == 0)].iloc[0]
    assert row.Y_TRUE == y[0] and row.Y_PRED == m[0] and row.END == locs[0, 2]


def test_single_split_generator_splits(tmp_path):
    """DatasetGenerator (dataset_generator.py:115-193): the held-out set comes off first (random share, chromosome tails
    or a file of windows), train / validation splits are redrawn per call."""
    import types
    from digdriver_amd.io import mapfile
    from digdriver_amd.region_model import mutations_main as mm
    rng = np.random.default_rng(1)
    N = 400
    chrom = np.repeat([1, 2, 3, 4], 100)
    idx = np.stack([chrom, np.tile(np.arange(100), 4) * 10000, np.tile(np.arange(100), 4) * 10000 + 10000], 1)
    data = str(tmp_path / "d.map")
    mapfile.write_array(data, "x_data", rng.integers(0, 100, (N, 4, 3)).astype(np.float32))
    mapfile.write_array(data, "idx", idx)
    mapp = rng.uniform(0.5, 1.0, N)
    mapfile.write_array(data, "mappability", mapp)
    y = rng.poisson(30, N).astype(float)
    mapfile.write_array(data, "COH", y)

    def make(extra):
        return mm.SplitData(mm.get_cmd_arguments("-c COH -d %s -o %s -m 0.7 -cq 0.99 %s" % (data, tmp_path, extra)), "cpu")
    d = make("-hr 0.25 --seed 3")
    kept = set(d.idxs) | set(d.heldout_idxs)
    assert not (set(d.idxs) & set(d.heldout_idxs)) and all(mapp[i] >= 0.7 for i in kept)
    assert len(d.heldout_idxs) == len(kept) - int(0.75 * len(kept))
    tr, va = d.get_datasets()
    tr2, _ = d.get_datasets()
    assert set(tr) | set(va) == set(d.idxs) and len(va) == len(d.idxs) - int(0.8 * len(d.idxs)) and not np.array_equal(tr, tr2)
    c = make("-s chr -hr 0.2")
    for ch in (1, 2, 3, 4):                                                # the tail of every chromosome is held out
        rows = np.sort([i for i in (set(c.idxs) | set(c.heldout_idxs)) if chrom[i] == ch])
        cut = int(0.8 * len(rows))
        assert set(rows[cut:]) <= set(c.heldout_idxs) and set(rows[:cut]) <= set(c.idxs)
    pick = [i for i in sorted(kept)][:5]
    hf = tmp_path / "held.tsv"
    hf.write_text("CHROM\tSTART\tEND\tY_TRUE\tY_PRED\tSTD\tPVAL\tRANK\n" +
                  "\n".join("%d\t%d\t%d\t%s\t0\t0\t0\t0" % (idx[i, 0], idx[i, 1], idx[i, 2], y[i]) for i in pick) + "\n")
    h = make("-u %s" % hf)
    assert list(h.heldout_idxs) == pick and not (set(pick) & set(h.idxs)) and len(h.idxs) == len(kept) - 5


_STREAM_PROBE = r'''
import resource, sys, os
sys.path.insert(0, sys.argv[2])
import numpy as np, torch
from digdriver_amd.region_model.data_aux import dataset_generator as dg
from digdriver_amd.io import mapfile
torch.zeros(1)
base = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
x = dg.load_track_matrix(sys.argv[1], torch.device(sys.argv[3]), slab_bytes=int(sys.argv[4]))
peak = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
rows = mapfile.read_array_rows(sys.argv[1], "x_data", 5, 9)
print("RESULT", str(x.dtype), tuple(x.shape), base, peak, float(x.double().sum()), bool(np.array_equal(x[5:9].cpu().float().numpy(), rows.astype(np.float32))))   # = torch.tensor(rows).float(), mut_dataset.py:79
'''


def _xdata_file(tmp_path, N, L, T, n_frac=0):
    import subprocess
    py39 = "/opt/conda/bin/python3.9"
    if not os.path.exists(py39):
        pytest.skip("the image's second interpreter (h5py 3.3) is not here")
    path = str(tmp_path / ("x%d.h5" % n_frac))
    subprocess.check_call([py39, os.path.join(GOLD, "make_xdata_fixture.py"), path, str(N), str(L), str(T), str(n_frac)])
    return path


def test_streaming_track_matrix_loader_keeps_one_slab_on_the_host(tmp_path):
    """load_track_matrix on a GENUINE h5py file in the reference's layout (x_data float64, gzip, chunked: DataExtractor.py:424-426):
    the matrix arrives as int16, values exact, and the host's peak resident set grows by the int16 destination (on the CPU
    device it lives in host memory) plus a few slabs -- NOT by the float64 matrix.  A file with a fractional value near its
    end becomes float32 without re-reading what was already loaded."""
    import subprocess
    import sys
    N, L, T = 6000, 100, 64                                  # 307 MB as float64, 77 MB as int16
    f64_bytes = N * L * T * 8
    path = _xdata_file(tmp_path, N, L, T)
    slab = 8 << 20
    out = subprocess.run([sys.executable, "-c", _STREAM_PROBE, path, ROOT, "cpu", str(slab)], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("RESULT")][0]
    _, dtype, rest = line.split(" ", 2)
    shape, tail = rest.split(") ", 1)
    base_kb, peak_kb, total, same = tail.split()
    assert dtype == "torch.int16" and shape == "(6000, 100, 64" and same == "True"
    grown = (int(peak_kb) - int(base_kb)) * 1024
    # destination (host memory on the CPU device) + the slab in its forms (decoded chunks, assembled rows, float64 / float32 /
    # int16 tensors) + allocator slack; the float64 matrix itself never exists
    assert grown < N * L * T * 2 + 10 * slab + (32 << 20), grown
    assert grown < 0.7 * f64_bytes
    # the values: compare with h5py's own reading through the lazy reader on a slice (done in the probe) and the total
    assert abs(float(total) - 50.0 * N * L * T) < 0.01 * 50.0 * N * L * T
    # a fractional value in the last rows: float32, everything kept
    path2 = _xdata_file(tmp_path, 300, 100, 16, n_frac=2)
    out2 = subprocess.run([sys.executable, "-c", _STREAM_PROBE, path2, ROOT, "cpu", str(1 << 20)], capture_output=True, text=True, timeout=600)
    assert out2.returncode == 0, out2.stderr[-2000:]
    line2 = [l for l in out2.stdout.splitlines() if l.startswith("RESULT")][0]
    assert "torch.float32 (300, 100, 16)" in line2 and line2.endswith("True")
