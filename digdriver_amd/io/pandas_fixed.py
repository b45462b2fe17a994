"""pandas' HDF5 "fixed" format for DataFrames, encoded / decoded on top of h5lite (no PyTables, no h5py).

What `DataFrame.to_hdf(path, key)` writes -- the reference stores region_params, sequence_model_192 / _64,
genic_model and the element frames this way (DigPretrain.py:96,207-208,234,266) and reads them back with
`pd.read_hdf` (transfer_tools.py:15) -- pinned to a file produced by pandas 2.3.3 + PyTables 3.6.1 in the build
container (tests/golden/pretrained_genuine.h5.gz, generator tests/golden/make_h5_fixture.py):

  /<key>            group; attrs CLASS='GROUP', VERSION='1.0', TITLE='' (PyTables), pandas_type='frame',
                    pandas_version='0.15.2', encoding='UTF-8', errors='strict', ndim=2, nblocks=B,
                    axis0_variety = axis1_variety = block{b}_items_variety = 'regular'
  axis0             column labels; axis1: index labels; block{b}_items: labels of the columns of block b
                    -> 'S<n>' arrays (labels utf-8 encoded, fixed width) with attrs kind='string', name=<pickled index
                    name, protocol 0: b'N.' for None>, or int64 / float64 arrays with kind 'integer' / 'float'
  block{b}_values   one block per dtype (pandas' consolidated order: sorted by dtype name), stored TRANSPOSED:
                    array [n_rows, n_cols_of_block], attr transposed=True; bool as H5T_STD_B8; object blocks (strings)
                    as a one-row VLArray of uint8 holding pickle.dumps(values.T) (attrs CLASS='VLARRAY', PSEUDOATOM='object')
  every array       attrs CLASS='ARRAY', VERSION='2.4', TITLE='', FLAVOR='numpy'
String attributes are fixed-length, NUL-terminated, UTF-8 scalars; the empty string is size 1 with a NULL dataspace;
integers are int64 scalars; booleans H5T_STD_B8.
"""
import pickle

import numpy as np
import pandas as pd

from . import h5lite as H


class FrameFormatError(RuntimeError):
    pass


def _fs(s):
    return H.FixedStr(s)


def _array_attrs(extra=None):
    d = {"CLASS": _fs("ARRAY"), "VERSION": _fs("2.4"), "TITLE": _fs(""), "FLAVOR": _fs("numpy"), "transposed": H.B8(True)}
    d.update(extra or {})
    return d


def _encode_index(index):
    """pandas.io.pytables._convert_index for the index kinds the DIG frames use."""
    name = H.FixedBytes(pickle.dumps(index.name, protocol=0)) if not isinstance(index.name, str) else _fs(index.name)
    if isinstance(index, pd.MultiIndex):
        raise FrameFormatError("MultiIndex frames are not supported by the fixed-format writer")
    vals = np.asarray(index.values)
    if vals.dtype.kind in "iu":
        return H.Dataset(vals.astype(np.int64), _array_attrs({"kind": _fs("integer"), "name": name}))
    if vals.dtype.kind == "f":
        return H.Dataset(vals.astype(np.float64), _array_attrs({"kind": _fs("float"), "name": name}))
    if vals.dtype.kind == "b":
        return H.Dataset(H.B8(vals), _array_attrs({"kind": _fs("bool"), "name": name}))
    enc = [str(v).encode("utf-8") for v in vals.tolist()]
    width = max([len(e) for e in enc] + [1])
    return H.Dataset(np.array(enc, dtype="S%d" % width), _array_attrs({"kind": _fs("string"), "name": name}))


def _dtype_key(dt):
    return "object" if dt.kind in "OUS" or str(dt) in ("string", "category") else dt.name


def _block_layout(df):
    """Column positions per block, as BlockManagerFixed.write sees them (pandas/io/pytables.py): the frame's block
    manager, consolidated if two blocks share a dtype.  Consolidation sorts the blocks by dtype name; a manager that is
    already consolidated keeps its construction order (runs of equal dtype in column order).  Pinned by the genuine
    fixture: region_params is written int64, float64, bool; genic_model (interleaved dtypes) bool, float64, int64, object."""
    try:
        mgr = df._mgr
        if not mgr.is_consolidated():
            mgr = mgr.consolidate()
        return [(_dtype_key(blk.dtype), [int(i) for i in blk.mgr_locs.as_array]) for blk in mgr.blocks]
    except AttributeError:          # pandas without these internals: the same rule from the dtypes
        runs = []
        for pos in range(df.shape[1]):
            key = _dtype_key(df.dtypes.iloc[pos])
            if runs and runs[-1][0] == key:
                runs[-1][1].append(pos)
            else:
                runs.append((key, [pos]))
        if len({k for k, _ in runs}) == len(runs):
            return runs
        merged = {}
        for key, cols in runs:
            merged.setdefault(key, []).extend(cols)
        return [(k, merged[k]) for k in sorted(merged)]


def encode_frame(df):
    """DataFrame -> h5lite.Group in pandas' fixed format."""
    if not isinstance(df, pd.DataFrame):
        raise FrameFormatError("only DataFrames are stored")
    if not df.columns.is_unique:
        raise FrameFormatError("Columns index has to be unique for fixed format")
    blocks = _block_layout(df)                     # [(kind key, column positions)] in the order to_hdf writes them
    g = H.Group({"CLASS": _fs("GROUP"), "VERSION": _fs("1.0"), "TITLE": _fs(""), "pandas_type": _fs("frame"),
                 "pandas_version": _fs("0.15.2"), "encoding": _fs("UTF-8"), "errors": _fs("strict"), "ndim": np.int64(2),
                 "axis0_variety": _fs("regular"), "axis1_variety": _fs("regular"), "nblocks": np.int64(len(blocks))})
    g.children["axis0"] = _encode_index(df.columns)
    g.children["axis1"] = _encode_index(df.index)
    for b, (key, cols) in enumerate(blocks):
        g.attrs["block%d_items_variety" % b] = _fs("regular")
        g.children["block%d_items" % b] = _encode_index(df.columns[cols])
        if key == "object":
            vals = np.empty((len(df), len(cols)), dtype=object)
            for j, pos in enumerate(cols):
                vals[:, j] = df.iloc[:, pos].astype(object).values
            g.children["block%d_values" % b] = H.Dataset(
                H.VLenObject(vals), {"CLASS": _fs("VLARRAY"), "VERSION": _fs("1.4"), "TITLE": _fs(""),
                                     "PSEUDOATOM": _fs("object"), "transposed": H.B8(True)})
        else:
            vals = np.ascontiguousarray(np.stack([df.iloc[:, pos].values for pos in cols], axis=1)) if cols else \
                np.zeros((len(df), 0))
            data = H.B8(vals) if vals.dtype.kind == "b" else vals
            g.children["block%d_values" % b] = H.Dataset(data, _array_attrs())
    return g


def _attr_text(v):
    if isinstance(v, bytes):
        return v.decode("utf-8", "replace")
    return v


def _decode_index(node, encoding="utf-8"):
    vals = node.data
    kind = _attr_text(node.attrs.get("kind"))
    name = node.attrs.get("name")
    if isinstance(name, (bytes, np.bytes_)):
        raw = bytes(name)
        try:
            name = pickle.loads(raw)                 # PyTables pickles non-string attribute values (None -> b'N.')
        except Exception:
            name = raw.decode(encoding, "replace")
    elif isinstance(name, str):
        name = str(name)
    elif name is not None and not isinstance(name, (int, float, tuple)):
        name = None if str(name) == "" else name
    if isinstance(vals, np.ndarray) and vals.dtype.kind == "S":
        vals = np.array([v.decode(encoding) for v in vals.tolist()], dtype=object)
    elif kind == "string" and isinstance(vals, np.ndarray) and vals.dtype == object:
        vals = np.array([v.decode(encoding) if isinstance(v, bytes) else v for v in vals.tolist()], dtype=object)
    return pd.Index(vals, name=name)


def decode_columns(g, wanted):
    """The columns `wanted` of a fixed-format frame group as 1-D numpy arrays (views of the stored blocks: nothing is copied,
    no DataFrame is built) -- for readers that take a few numeric columns of a large frame (a whole-genome region_params
    frame: building the DataFrame and re-ordering its columns held the interpreter lock for 15 ms per map, 37 maps side by
    side spent most of their time waiting for it).  KeyError names a column the frame does not have."""
    if not isinstance(g, H.Group) or "axis0" not in g.children:
        raise FrameFormatError("not a pandas fixed-format frame")
    enc = _attr_text(g.attrs.get("encoding")) or "utf-8"
    n_rows = int(np.shape(g.children["axis1"].data)[0])
    nblocks = int(g.attrs.get("nblocks", sum(1 for k in g.children if k.endswith("_items"))))
    need, out = set(wanted), {}
    for b in range(nblocks):
        items = list(_decode_index(g.children["block%d_items" % b], enc))
        hit = [(j, name) for j, name in enumerate(items) if name in need]
        if not hit:
            continue
        node = g.children["block%d_values" % b]
        vals = node.data
        if isinstance(vals, H.VLenObject):
            vals = vals.load()
        vals = np.asarray(vals)
        transposed = bool(node.attrs.get("transposed", True))
        if vals.ndim == 1:
            vals = vals.reshape(n_rows, -1) if transposed else vals.reshape(-1, n_rows)
        if not transposed:
            vals = vals.T
        for j, name in hit:
            out[name] = vals[:, j]
    missing = [k for k in wanted if k not in out]
    if missing:
        raise KeyError("the frame has no column %r" % missing[0])
    return out


def decode_frame(g, with_index=True):
    """h5lite.Group in pandas' fixed format -> DataFrame.  with_index=False: a RangeIndex instead of the stored row labels
    (a whole-genome region_params frame stores 288 000 label strings nobody reads: two thirds of the time to load it)."""
    if not isinstance(g, H.Group):
        raise FrameFormatError("not a frame group (it is a dataset)")
    ptype = _attr_text(g.attrs.get("pandas_type"))
    if ptype not in ("frame", None) or "axis0" not in g.children or "axis1" not in g.children:
        if ptype in ("frame_table", "series_table") or "table" in g.children:
            raise FrameFormatError("the frame was stored with format='table'; only the fixed format (to_hdf default) is read")
        raise FrameFormatError("not a pandas fixed-format frame (pandas_type=%r)" % (ptype,))
    enc = _attr_text(g.attrs.get("encoding")) or "utf-8"
    columns = _decode_index(g.children["axis0"], enc)
    index = _decode_index(g.children["axis1"], enc) if with_index else pd.RangeIndex(int(np.shape(g.children["axis1"].data)[0]))
    nblocks = int(g.attrs.get("nblocks", sum(1 for k in g.children if k.endswith("_items"))))
    data = {}
    for b in range(nblocks):
        items = _decode_index(g.children["block%d_items" % b], enc)
        node = g.children["block%d_values" % b]
        vals = node.data
        if isinstance(vals, H.VLenObject):
            vals = vals.load()
        vals = np.asarray(vals)
        transposed = bool(node.attrs.get("transposed", True))
        if vals.ndim == 1:
            vals = vals.reshape(len(index), -1) if transposed else vals.reshape(-1, len(index))
        if not transposed:
            vals = vals.T
        if vals.dtype.kind == "S":
            vals = np.array([[v.decode(enc) for v in row] for row in vals.tolist()], dtype=object).reshape(vals.shape)
        for j, name in enumerate(items):
            data[name] = vals[:, j]
    df = pd.DataFrame(data, index=index)
    df = df[list(columns)]
    df.columns = columns
    return df
