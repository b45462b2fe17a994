"""Mutation-map containers: the pretrained "h5 mutation map" of the reference and a portable mirror.

The reference keeps everything in one HDF5 file (DigPretrain.py:82-96,156-177,207-208,234,266):
  datasets  idx (int32 [N,3]), mappability (float32 [N])
  attrs     cohort_name, N_SAMPLES, N_MUT_TOTAL, N_MUT_TRAIN, N_MUT_CDS, N_MUT_<panel>, ...
  frames    region_params, sequence_model_192, sequence_model_64, genic_model, <element keys>
            written with pandas DataFrame.to_hdf (PyTables "fixed" format)
and the per-element context counts in a second HDF5 file (sequence_tools.py:460-478,639-641).

Backends, chosen by the path:
  *.h5 / *.hdf5   HDF5, read and written by this package's own implementation: io/h5lite.py (the HDF5 container:
                  what h5py / PyTables write with the library's default format) + io/pandas_fixed.py (the
                  DataFrame.to_hdf "fixed" layout).  No h5py, no PyTables needed; pinned to files written by those
                  libraries (tests/golden/*_genuine.h5) and cross-checked with h5py and h5dump (tests/test_h5_io.py).
  anything else   a directory of .npy files ("DIG map mirror"): same keys, same frames.
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
        cur[k] = v.item() if isinstance(v, np.generic) else (v.tolist() if isinstance(v, np.ndarray) else v)
    with open(_dir_attrs_path(path), "w") as f:
        json.dump(cur, f)


# ---------------------------------------------------------------------------------------------
# HDF5 backend: h5lite (own reader / writer) + pandas_fixed (pandas' "fixed" frame layout); no h5py, no PyTables
# ---------------------------------------------------------------------------------------------
from . import h5lite, pandas_fixed          # noqa: E402

_TREE_CACHE = {}      # path -> (mtime_ns, size, lazy tree): a driver run reads several keys of the same map


def _h5_tree(path):
    path = os.fspath(path)
    if path in _PENDING:
        return _PENDING[path]
    try:
        st = os.stat(path)
    except FileNotFoundError:
        raise MapFileError("no such HDF5 map: %s" % path)
    key = (st.st_mtime_ns, st.st_size)
    hit = _TREE_CACHE.get(path)
    if hit is None or hit[0] != key:
        try:
            hit = (key, h5lite.read_tree(path, lazy=True))
        except h5lite.H5LiteError as exc:
            raise MapFileError("%s: %s" % (path, exc)) from exc
        _TREE_CACHE.clear()
        _TREE_CACHE[path] = hit
    return hit[1]


_PENDING = {}         # path -> tree of an open batch()


class batch:
    """`with mapfile.batch(path): ...` -- every write to the HDF5 map `path` inside the block goes to one in-memory tree
    that is written once at the end (h5lite rewrites the file on every update; a fold-results file takes hundreds of
    small writes).  Reads of the same path inside the block see the pending tree.  No effect on directory maps."""

    def __init__(self, path):
        self.path = os.fspath(path)

    def __enter__(self):
        if _is_h5(self.path) and self.path not in _PENDING:
            _TREE_CACHE.pop(self.path, None)
            _PENDING[self.path] = h5lite.read_tree(self.path) if os.path.exists(self.path) else h5lite.Group()
            self.owner = True
        else:
            self.owner = False
        return self

    def __exit__(self, exc_type, exc, tb):
        if self.owner:
            root = _PENDING.pop(self.path)
            if exc_type is None:
                h5lite.write_tree(self.path, root)
        return False


def _h5_update(path, fn):
    path = os.fspath(path)
    _TREE_CACHE.pop(path, None)
    if path in _PENDING:
        root = _PENDING[path]
        for k, v in (("CLASS", "GROUP"), ("VERSION", "1.0"), ("TITLE", ""), ("PYTABLES_FORMAT_VERSION", "2.1")):
            root.attrs.setdefault(k, h5lite.FixedStr(v))
        fn(root)
        return

    def apply(root):
        # what PyTables puts on the root group of every file it creates (pandas.read_hdf opens maps through it)
        for k, v in (("CLASS", "GROUP"), ("VERSION", "1.0"), ("TITLE", ""), ("PYTABLES_FORMAT_VERSION", "2.1")):
            root.attrs.setdefault(k, h5lite.FixedStr(v))
        fn(root)

    h5lite.update(path, apply)


def _h5_read_frame(path, key, index=True):
    root = _h5_tree(path)
    if key not in root:
        raise KeyError("no frame %r in %s" % (key, path))
    try:
        return pandas_fixed.decode_frame(root[key], with_index=index)
    except pandas_fixed.FrameFormatError as exc:
        raise MapFileError("%s:%s: %s" % (path, key, exc)) from exc


def _h5_write_frame(path, key, df):
    g = pandas_fixed.encode_frame(df)
    _h5_update(path, lambda root: root.set(key, g))


def _attr_value(v):
    if isinstance(v, h5lite.NullString):
        return ""
    if isinstance(v, h5lite.B8):
        return bool(v) if v.value.shape == () else v.value
    if isinstance(v, str):
        return str(v)
    if isinstance(v, (bytes, np.bytes_)):
        try:
            return bytes(v).decode("utf-8")
        except UnicodeDecodeError:
            return bytes(v)
    if isinstance(v, np.generic):
        return v.item()
    return v


# ---------------------------------------------------------------------------------------------
# public API
# ---------------------------------------------------------------------------------------------
def read_frame(path, key, index=True):
    """index=False: the rows without their stored labels (a RangeIndex) -- for callers that only take columns."""
    if _is_h5(path):
        return _h5_read_frame(path, key, index)
    df = _dir_read_frame(path, key)
    return df if index else df.reset_index(drop=True)


def read_columns(path, key, columns):
    """{name: 1-D array} of the named columns of a frame, contiguous, without building the DataFrame (HDF5 maps: straight from
    the stored blocks)."""
    import numpy as np
    if _is_h5(path):
        root = _h5_tree(path)
        if key not in root:
            raise KeyError("no frame %r in %s" % (key, path))
        try:
            got = pandas_fixed.decode_columns(root[key], list(columns))
        except pandas_fixed.FrameFormatError as exc:
            raise MapFileError("%s:%s: %s" % (path, key, exc)) from exc
        return {k: np.ascontiguousarray(v) for k, v in got.items()}
    df = _dir_read_frame(path, key)
    return {k: np.ascontiguousarray(df[k].values) for k in columns}


def write_frame(path, key, df):
    return _h5_write_frame(path, key, df) if _is_h5(path) else _dir_write_frame(path, key, df)


def read_array(path, key):
    if _is_h5(path):
        root = _h5_tree(path)
        if key not in root or not isinstance(root[key], h5lite.Dataset):
            raise KeyError("no array %r in %s" % (key, path))
        return root[key].data
    return _dir_read_array(path, key)


def read_array_rows(path, key, lo, hi):
    """array[lo:hi] along the first dimension; an HDF5 map reads only the chunks / bytes it needs (track matrices)."""
    if _is_h5(path):
        root = _h5_tree(path)
        if key not in root:
            raise KeyError("no array %r in %s" % (key, path))
        return root[key].read_rows(lo, hi)
    f = os.path.join(path, "A." + _safe(key) + ".npy")
    if not os.path.exists(f):
        raise KeyError("no array %r in %s" % (key, path))
    return np.array(np.load(f, mmap_mode="r", allow_pickle=False)[lo:hi])       # only the pages of these rows are read


def array_shape(path, key):
    if _is_h5(path):
        return tuple(_h5_tree(path)[key].shape)
    f = os.path.join(path, "A." + _safe(key) + ".npy")
    if not os.path.exists(f):
        raise KeyError("no array %r in %s" % (key, path))
    return tuple(np.load(f, mmap_mode="r", allow_pickle=False).shape)


def array_dtype(path, key):
    if _is_h5(path):
        return np.dtype(_h5_tree(path)[key].dtype)
    return np.load(os.path.join(path, "A." + _safe(key) + ".npy"), mmap_mode="r", allow_pickle=False).dtype


def write_array(path, key, arr, **kw):
    """(compression keywords of the reference's create_dataset calls are accepted and ignored: datasets are written
    contiguous, which every HDF5 reader handles)"""
    if _is_h5(path):
        a = np.asarray(arr)
        if a.dtype.kind == "U":
            a = np.char.encode(a, "utf-8")
        elif a.dtype == object:
            a = np.array([str(x).encode("utf-8") for x in a.reshape(-1)]).reshape(a.shape)
        _h5_update(path, lambda root: root.set(key, h5lite.Dataset(a)))
        return
    _dir_write_array(path, key, arr)


def read_attrs(path, key=None):
    """Attributes of the root group (or of the object `key`)."""
    if _is_h5(path):
        node = _h5_tree(path) if key is None else _h5_tree(path)[key]
        return {k: _attr_value(v) for k, v in node.attrs.items()
                if not (key is None and k in ("CLASS", "VERSION", "TITLE", "PYTABLES_FORMAT_VERSION"))}
    if key is None:
        return {k: v for k, v in _dir_read_attrs(path).items() if "@" not in k}
    pre = key.strip("/") + "@"
    return {k[len(pre):]: v for k, v in _dir_read_attrs(path).items() if k.startswith(pre)}


def write_attrs(path, _key=None, **kw):
    """Attributes of the root group, or of the group / dataset `_key` (created as a group when missing): the
    reference's `h5.attrs[...] = ...` and `grp.attrs['R2'] = ...`."""
    if _is_h5(path):
        def put(root):
            node = root if _key is None else (root[_key] if _key in root else root.require_group(_key))
            for k, v in kw.items():
                node.attrs[k] = v if isinstance(v, (str, bytes)) else np.asarray(v)[()]
        _h5_update(path, put)
        return
    if _key is not None:
        kw = {_key.strip("/") + "@" + k: v for k, v in kw.items()}
    _dir_write_attrs(path, **kw)


def list_keys(path, key=""):
    """Names of the members of group `key` of an HDF5 map."""
    if not _is_h5(path):
        raise MapFileError("list_keys is for HDF5 maps")
    node = _h5_tree(path)[key] if key else _h5_tree(path)
    return list(node.keys()) if isinstance(node, h5lite.Group) else []


def has_key(path, key):
    if _is_h5(path):
        if os.fspath(path) in _PENDING:
            return key in _PENDING[os.fspath(path)]
        return os.path.exists(path) and key in _h5_tree(path)
    s = _safe(key)
    return os.path.exists(os.path.join(path, "A." + s + ".npy")) or os.path.exists(os.path.join(path, "F." + s + ".json"))


_LABEL_CACHE = {}


def _cached_labels(index):
    """(blob, offsets) of the encoded text of `index` if it is the one kept from the last call, else None.  The same object, or an equal index (DataFrame.assign and
    .copy hand on a NEW Index object: an identity test alone missed every one of the 37 frames of a many-cohort run, and the
    120 091 labels were turned into text 37 times under the interpreter lock -- 1.2 s of the 1.3 s the files took)."""
    cached = _LABEL_CACHE.get("last")                   # (read once: other threads may replace the entry)
    if cached is None:
        return None
    if cached[0] is not index:
        try:
            same = len(cached[0]) == len(index) and cached[0].dtype == index.dtype and cached[0].name == index.name and cached[0].equals(index)
        except Exception:
            same = False
        if not same:
            return None
    return cached[1], cached[2]


def encode_labels(index):
    """(blob, offsets) of the text of a str / integer index, or None if it holds something the native writer must leave to pandas
    (see write_results_tsv)."""
    import numpy as np
    int_index = isinstance(index.dtype, np.dtype) and index.dtype.kind in 'iu'
    if not int_index and index.dtype != object:
        return None
    labels = [str(x) for x in index] if int_index else list(index)
    if not int_index and (not all(type(x) is str for x in labels) or any(ch in s_ for s_ in labels for ch in ('\t', '"', '\n', '\r'))):
        return None
    enc = [s_.encode() for s_ in labels]
    off = np.zeros(len(enc) + 1, np.int64)
    np.cumsum([len(b) for b in enc], out=off[1:])
    return b"".join(enc), off


def write_columns_tsv(path, index_name, labels, columns, threads=8):
    """The text of DataFrame(dict(columns), index).to_csv(path, sep="\t") from the column arrays themselves: `labels` =
    encode_labels(index), `columns` = [(name, 1-D array)] of float64 / integer / bool arrays.  The bulk path of the many-cohort
    driver (no pandas object involved); raises ValueError for anything else -- write_results_tsv is the general entry."""
    import ctypes
    import numpy as np
    from .. import _lib
    if labels is None:
        raise ValueError("row labels the native writer does not cover")
    blob, off = labels
    kinds, cols, names = [], [], [str(index_name) if index_name is not None else '']
    for name, v in columns:
        names.append(str(name))
        if v.dtype == np.float64:
            kinds.append(0); cols.append(np.ascontiguousarray(v))
        elif v.dtype.kind in 'iu' and v.dtype != np.uint64:
            kinds.append(1); cols.append(np.ascontiguousarray(v, np.int64))
        elif v.dtype == np.bool_:
            kinds.append(2); cols.append(np.ascontiguousarray(v, np.uint8))
        else:
            raise ValueError("column %s: dtype %s" % (name, v.dtype))
        if len(v) != len(off) - 1:
            raise ValueError("column %s: %d rows, %d labels" % (name, len(v), len(off) - 1))
    if any(ch in s_ for s_ in names for ch in ('\t', '"', '\n', '\r')):
        raise ValueError("a column name the native writer does not cover")
    ptrs = (ctypes.c_void_p * max(len(cols), 1))(*[c.ctypes.data for c in cols])
    kind_arr = np.asarray(kinds or [0], np.int32)
    _lib.call("dig_write_tsv_host", os.fspath(path).encode(), "\t".join(names).encode(), ctypes.c_char_p(blob) if blob else ctypes.c_char_p(b""),
              _lib.host_ptr(off), len(off) - 1, len(cols), ctypes.cast(ptrs, ctypes.c_void_p), _lib.host_ptr(kind_arr), int(threads))
    return path


def write_results_tsv(frame, path, threads=8):
    """frame.to_csv(path, header=True, index=True, sep="\t") -- the text DigDriver.py writes (DigDriver.py:115-118) -- through
    the native writer dig_write_tsv_host: the same bytes (floats as Python's repr, NaN as the empty field, bools as True / False,
    integers as they are), 2.0 s -> 0.05 s for a 120 091-row result frame.  Frames the writer does not cover (a column that is
    neither float, integer nor bool, labels with tabs / quotes / newlines, a MultiIndex) go through pandas."""
    import ctypes
    import numpy as np
    from .. import _lib
    kinds, cols = [], []
    plain = frame.index.nlevels == 1 and frame.columns.nlevels == 1
    if plain:
        for name in frame.columns:
            v = frame[name].values
            if not isinstance(v, np.ndarray):          # (a nullable / extension column: pandas knows how its NA prints)
                plain = False
            elif v.dtype == np.float64:
                kinds.append(0); cols.append(np.ascontiguousarray(v))
            elif v.dtype.kind == 'f':
                plain = False                          # (float32 prints differently: left to pandas)
            elif v.dtype.kind in 'iu' and v.dtype.itemsize <= 8 and v.dtype != np.uint64:
                kinds.append(1); cols.append(np.ascontiguousarray(v, np.int64))
            elif v.dtype == np.bool_:
                kinds.append(2); cols.append(np.ascontiguousarray(v, np.uint8))
            else:
                plain = False
            if not plain:
                break
    labels, hit = None, None
    if plain:
        # the native path takes integer indexes and object indexes of plain str only: None / NaN labels (pandas writes an empty
        # field), datetimes (pandas drops a zero time of day), floats and mixed objects (1 and 1.0 compare equal but print
        # differently -- the label cache below matches on Index.equals) are pandas' business
        idx = frame.index
        names = [str(idx.name) if idx.name is not None else ''] + [str(c) for c in frame.columns]
        special = ('\t', '"', '\n', '\r')
        int_index = isinstance(idx.dtype, np.dtype) and idx.dtype.kind in 'iu'
        if not int_index and idx.dtype != object:
            plain = False
        if plain:
            hit = _cached_labels(idx)
        if plain and hit is None:
            if int_index:
                labels = [str(x) for x in idx]
            else:
                labels = list(idx)
                if not all(type(x) is str for x in labels) or any(ch in s_ for s_ in labels for ch in special):
                    plain = False
        if any(ch in s_ for s_ in names for ch in special):
            plain = False
    if not plain:
        frame.to_csv(path, header=True, index=True, sep="\t")
        return path
    # (the 37 result frames of a many-cohort run share one index object: its text is encoded once)
    if hit is not None:
        blob, off = hit
    else:
        enc = [s_.encode() for s_ in labels]
        off = np.zeros(len(enc) + 1, np.int64)
        np.cumsum([len(b) for b in enc], out=off[1:])
        blob = b"".join(enc)
        _LABEL_CACHE["last"] = (frame.index, blob, off)
    ptrs = (ctypes.c_void_p * max(len(cols), 1))(*[c.ctypes.data for c in cols])
    kind_arr = np.asarray(kinds or [0], np.int32)
    _lib.call("dig_write_tsv_host", os.fspath(path).encode(), "\t".join(names).encode(), ctypes.c_char_p(blob) if blob else ctypes.c_char_p(b""),
              _lib.host_ptr(off), len(off) - 1, len(cols), ctypes.cast(ptrs, ctypes.c_void_p), _lib.host_ptr(kind_arr), int(threads))
    return path
