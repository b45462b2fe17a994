"""Mutation-map containers: the pretrained "h5 mutation map" of the reference and a portable mirror.

The reference keeps everything in one HDF5 file (DigPretrain.py:82-96,156-177,207-208,234,266):
  datasets  idx (int32 [N,3]), mappability (float32 [N])
  attrs     cohort_name, N_SAMPLES, N_MUT_TOTAL, N_MUT_TRAIN, N_MUT_CDS, N_MUT_<panel>, ...
  frames    region_params, sequence_model_192, sequence_model_64, genic_model, <element keys>
            written with pandas DataFrame.to_hdf (PyTables "fixed" format)
and the per-element context counts in a second HDF5 file (sequence_tools.py:460-478,639-641).

Backends, chosen by the path:
  *.h5 / *.hdf5   HDF5.  Frames are read/written with pandas when PyTables is importable, otherwise read
                  through h5py by decoding the PyTables "fixed" layout (axis0/axis1/block*_items/
                  block*_values).  Needs h5py; raises MapFileError when neither is available.
  anything else   a directory of .npy files ("DIG map mirror"): same keys, same frames, no HDF5
                  dependency.  This is what the tests and the synthetic benchmarks use.
"""
import json
import os

import numpy as np
import pandas as pd


class MapFileError(RuntimeError):
    pass


def _is_h5(path):
    return str(path).endswith((".h5", ".hdf5", ".hdf"))


def _safe(key):
    return key.strip("/").replace("/", "__")


# ---------------------------------------------------------------------------------------------
# directory backend
# ---------------------------------------------------------------------------------------------
def _dir_write_array(path, key, arr):
    os.makedirs(path, exist_ok=True)
    arr = np.asarray(arr)
    if arr.dtype == object:
        arr = arr.astype(str)
    np.save(os.path.join(path, "A." + _safe(key) + ".npy"), arr, allow_pickle=False)


def _dir_read_array(path, key):
    f = os.path.join(path, "A." + _safe(key) + ".npy")
    if not os.path.exists(f):
        raise KeyError("no array %r in %s" % (key, path))
    return np.load(f, allow_pickle=False)


def _dir_write_frame(path, key, df):
    os.makedirs(path, exist_ok=True)
    base = os.path.join(path, "F." + _safe(key))
    meta = {"columns": [str(c) for c in df.columns], "index_name": df.index.name}
    with open(base + ".json", "w") as f:
        json.dump(meta, f)
    arrays = {"__index__": np.asarray(df.index.values)}
    if arrays["__index__"].dtype == object:
        arrays["__index__"] = arrays["__index__"].astype(str)
    for i, c in enumerate(df.columns):
        v = np.asarray(df[c].values)
        if v.dtype == object:
            v = v.astype(str)
        arrays["c%d" % i] = v
    np.savez(base + ".npz", **arrays)


def _dir_read_frame(path, key):
    base = os.path.join(path, "F." + _safe(key))
    if not os.path.exists(base + ".json"):
        raise KeyError("no frame %r in %s" % (key, path))
    meta = json.load(open(base + ".json"))
    z = np.load(base + ".npz", allow_pickle=False)
    data = {c: z["c%d" % i] for i, c in enumerate(meta["columns"])}
    df = pd.DataFrame(data, index=pd.Index(z["__index__"], name=meta["index_name"]))
    return df[meta["columns"]]


def _dir_attrs_path(path):
    return os.path.join(path, "attrs.json")


def _dir_read_attrs(path):
    f = _dir_attrs_path(path)
    return json.load(open(f)) if os.path.exists(f) else {}


def _dir_write_attrs(path, **kw):
    os.makedirs(path, exist_ok=True)
    cur = _dir_read_attrs(path)
    for k, v in kw.items():
        cur[k] = v.item() if isinstance(v, np.generic) else v
    with open(_dir_attrs_path(path), "w") as f:
        json.dump(cur, f)


# ---------------------------------------------------------------------------------------------
# HDF5 backend
# ---------------------------------------------------------------------------------------------
def _h5py():
    try:
        import h5py
        return h5py
    except ImportError as exc:
        raise MapFileError("reading %s needs h5py (or use the directory mirror format)" % "HDF5 maps") from exc


def _decode(a):
    a = np.asarray(a)
    if a.dtype.kind == "S":
        return np.char.decode(a, "utf-8")
    if a.dtype == object:
        return np.array([x.decode("utf-8") if isinstance(x, bytes) else x for x in a])
    return a


def _h5_read_fixed_frame(path, key):
    """Decode a PyTables 'fixed' DataFrame group with h5py: axis0 = columns, axis1 = index,
    block{i}_items = column names of block i, block{i}_values = [n_rows, n_cols_in_block]."""
    h5py = _h5py()
    with h5py.File(path, "r") as h5:
        if key not in h5:
            raise KeyError("no frame %r in %s" % (key, path))
        g = h5[key]
        if "axis0" not in g or "axis1" not in g:
            raise MapFileError("%s:%s is not a PyTables fixed-format frame (table format is not supported)" % (path, key))
        columns = list(_decode(g["axis0"][:]))
        index = _decode(g["axis1"][:])
        nblocks = int(g.attrs.get("nblocks", sum(1 for k in g.keys() if k.endswith("_items"))))
        data = {}
        for b in range(nblocks):
            items = _decode(g["block%d_items" % b][:])
            vals = g["block%d_values" % b]
            if vals.dtype.kind == "O" or vals.shape == () or vals.ndim != 2:
                raise MapFileError("%s:%s block %d holds pickled objects; re-save the frame with numeric/string "
                                   "columns or install PyTables" % (path, key, b))
            vals = _decode(vals[:])
            for j, name in enumerate(items):
                data[name] = vals[:, j]
    return pd.DataFrame(data, index=pd.Index(index))[columns]


def _h5_read_frame(path, key):
    try:
        import tables  # noqa: F401
        return pd.read_hdf(path, key)
    except ImportError:
        return _h5_read_fixed_frame(path, key)


def _h5_write_frame(path, key, df):
    try:
        import tables  # noqa: F401
    except ImportError as exc:
        raise MapFileError("writing DataFrames into HDF5 needs PyTables (pandas.to_hdf); use the directory "
                           "mirror format instead") from exc
    df.to_hdf(path, key=key, mode="a")


# ---------------------------------------------------------------------------------------------
# public API
# ---------------------------------------------------------------------------------------------
def read_frame(path, key):
    return _h5_read_frame(path, key) if _is_h5(path) else _dir_read_frame(path, key)


def write_frame(path, key, df):
    return _h5_write_frame(path, key, df) if _is_h5(path) else _dir_write_frame(path, key, df)


def read_array(path, key):
    if _is_h5(path):
        with _h5py().File(path, "r") as h5:
            return h5[key][:]
    return _dir_read_array(path, key)


def write_array(path, key, arr, **kw):
    if _is_h5(path):
        with _h5py().File(path, "a") as h5:
            if key in h5:
                del h5[key]
            h5.create_dataset(key, data=np.asarray(arr), **kw)
        return
    _dir_write_array(path, key, arr)


def read_attrs(path):
    if _is_h5(path):
        with _h5py().File(path, "r") as h5:
            return {k: (v.item() if isinstance(v, np.generic) else v) for k, v in h5.attrs.items()}
    return _dir_read_attrs(path)


def write_attrs(path, **kw):
    if _is_h5(path):
        with _h5py().File(path, "a") as h5:
            for k, v in kw.items():
                h5.attrs[k] = v
        return
    _dir_write_attrs(path, **kw)


def has_key(path, key):
    if _is_h5(path):
        with _h5py().File(path, "r") as h5:
            return key in h5
    s = _safe(key)
    return os.path.exists(os.path.join(path, "A." + s + ".npy")) or os.path.exists(os.path.join(path, "F." + s + ".json"))
