"""Bin selection for the region model: mappability / count-quantile filters and k-fold splits.

Mirror of the index logic in DIGDriver/region_model/data_aux/dataset_generator.py (BaseDatasetGenerator
:16-50, KFoldDatasetGenerator :189-273) and of the per-bin gather of mut_dataset.py:76-81.  The reference
re-opens the HDF5 file for every sample inside 16 DataLoader workers; here the whole bin x position x track
matrix is resident in HBM (fp32 or int16: values are round(x, 2) * 100, DataExtractor.py:220) and a batch is
one dig_gather_bins launch that also emits the channels-first layout conv1d consumes.
"""
import re

import numpy as np

from ... import engine


def rank_quantiles(labels):
    """stats.mstats.rankdata(labels) / len(labels) (dataset_generator.py:27): average ranks for ties."""
    labels = np.asarray(labels)
    order = np.argsort(labels, kind="mergesort")
    ranks = np.empty(len(labels), float)
    sorted_l = labels[order]
    # average rank over runs of equal values
    boundaries = np.flatnonzero(np.concatenate([[True], sorted_l[1:] != sorted_l[:-1], [True]]))
    for a, b in zip(boundaries[:-1], boundaries[1:]):
        ranks[order[a:b]] = 0.5 * (a + 1 + b)
    return ranks / len(labels)


def select_bins(mappability, labels, mapp_thresh, count_quantile):
    """dataset_generator.py:28-42: keep bins with mappability >= threshold and count <= quantile(cq).
    Returns (idxs, below_mapp) exactly as the reference defines them."""
    mappability, labels = np.asarray(mappability), np.asarray(labels)
    low_map = mappability < mapp_thresh
    high_count = labels > np.quantile(labels, count_quantile)
    return np.where(~low_map & ~high_count)[0], np.where(low_map | high_count)[0]


def load_track_selection(lines):
    """Track-selection grammar of dataset_generator.py:57-80: one integer or a half-open range 'a:b' per line;
    blank lines and '#' comments are skipped; anything else is an error."""
    tracks = []
    for i, raw in enumerate(lines):
        if raw.startswith(('\n', '#')) or raw == "":
            continue
        tok = raw.rstrip()
        if re.search(r'[^:0-9]', tok):
            raise ValueError('Expected track selection lines to contain only digits and colons. Found: {} in line #{}.'.format(tok, i))
        parts = tok.split(':')
        if len(parts) > 2 or not all(p.isdigit() for p in parts):
            raise ValueError('Expected one integer or one "a:b" pair. Found: {} in line #{}.'.format(tok, i))
        if len(parts) == 1:
            tracks.append(int(parts[0]))
        else:
            a, b = int(parts[0]), int(parts[1])
            if not a < b:
                raise ValueError('Expected x < y in pair x:y. Found: {} in line #{}.'.format(tok, i))
            tracks.extend(range(a, b))
    return tracks


def split_folds(idxs, k, seed=None):
    """dataset_generator.py:208-213: shuffle, then k contiguous folds.  The reference's shuffle is unseeded
    (fold membership is not reproducible there); here the seed is explicit."""
    idxs = np.array(idxs)
    np.random.default_rng(seed).shuffle(idxs)
    return np.array_split(idxs, k)


def load_track_matrix(path, device, key='x_data', slab_bytes=256 << 20, log=None):
    """The bin x position x track matrix of a training file, resident on `device`, filled SLAB BY SLAB.

    The reference stores x_data (N, L, T) as float64, gzip-chunked (DataExtractor.py:424-426; 169 GB for a whole genome at
    T = 735) and re-opens the file for every sample (mut_dataset.py:76-81).  Here the matrix lives in HBM as int16 -- track
    values are around(x, 2) * 100 (DataExtractor.py:220), whose float32 image (what the reference's .float() hands to the
    network) is an integer that int16 carries exactly, 42 GB -- and the host never holds more than one
    slab of rows (`slab_bytes` of the on-disk type): rows [lo, hi) are read through the lazy chunk reader
    (mapfile.read_array_rows), uploaded, checked for exactness and converted ON THE DEVICE.  The first slab that holds a
    value int16 cannot carry switches the whole matrix to float32 (the int16 rows already filled are converted on the
    device; nothing is re-read).  Returns a torch tensor [N, L, T] (int16 or float32; float32 / int16 sources keep their type)."""
    import torch
    from ...io import mapfile
    shape = mapfile.array_shape(path, key)
    assert len(shape) == 3, "x_data must be [bins, positions, tracks]"
    N, L, T = (int(v) for v in shape)
    src = mapfile.array_dtype(path, key)
    rows = max(1, int(slab_bytes) // max(1, L * T * src.itemsize))
    keep = {np.dtype(np.int16): torch.int16, np.dtype(np.float32): torch.float32}.get(src)
    out = torch.empty((N, L, T), dtype=keep or torch.int16, device=device)
    exact = keep is None                      # float64 (or another float) source: try int16 first
    for lo in range(0, N, rows):
        hi = min(N, lo + rows)
        slab = torch.as_tensor(np.ascontiguousarray(mapfile.read_array_rows(path, key, lo, hi))).to(device)
        if keep is not None:
            out[lo:hi] = slab
            continue
        if exact:
            # what the network sees in the reference is torch.tensor(x).float() (mut_dataset.py:79): the float32 image of
            # around(x, 2) * 100, whose float64 form is often one ulp off the integer (0.29 * 100 = 28.999999999999996)
            s32 = slab.to(torch.float32)
            r = torch.round(s32)
            if bool(((r == s32) & (s32.abs() < 32768)).all()):
                out[lo:hi] = r.to(torch.int16)
                continue
            exact = False                      # a non-integer (or too large) value: the matrix becomes float32
            wide = torch.empty((N, L, T), dtype=torch.float32, device=device)
            wide[:lo] = out[:lo].to(torch.float32)
            out = wide
            if log:
                log('x_data holds values int16 cannot carry (first in bins {}-{}): keeping float32'.format(lo, hi))
        out[lo:hi] = slab.to(torch.float32)
    return out


class BinTrackStore:
    """x_data [N, L, T] resident on the GPU (torch tensor, fp32 / fp64 / int16) + batch gather."""

    def __init__(self, x_data, selected_tracks=None, row_offset=0, row_ranges=None):
        """row_offset / row_ranges: this store holds the rows [row_offset, row_offset + len(x_data)) of a track matrix that is
        sharded over ranks by contiguous bin ranges `row_ranges` [(lo, hi)] (predict.predict_sharded sends a bin to the rank
        that holds it); bin rows handed to batch() are global."""
        self.x = x_data
        self.n_bins, self.length, self.n_tracks_total = x_data.shape
        self.tracks = None if selected_tracks is None else np.asarray(selected_tracks)    # None = all tracks, in order
        self.row_offset, self.row_ranges = int(row_offset), row_ranges

    def shape(self, n):
        return (n, self.length, self.n_tracks_total if self.tracks is None else len(self.tracks))

    def batch(self, bin_rows, channels_first=True, out_dtype="f32"):
        """x_data[bin_rows, :, tracks] (mut_dataset.py:76-81) for a batch, optionally as [B, T, L]."""
        import torch
        if self.row_offset:
            bin_rows = (bin_rows if torch.is_tensor(bin_rows) else np.asarray(bin_rows)) - self.row_offset
        tracks = self.tracks
        if tracks is not None and torch.is_tensor(self.x) and self.x.is_cuda:
            # (the track list goes to the device once: an upload per batch cannot be part of a captured training step)
            if getattr(self, "_tracks_dev", None) is None or self._tracks_dev.device != self.x.device:
                self._tracks_dev = torch.as_tensor(np.ascontiguousarray(tracks), dtype=torch.int32, device=self.x.device)
            tracks = self._tracks_dev
        return engine.gather_bins(self.x, bin_rows, tracks, out_dtype=out_dtype, transpose=channels_first)
