"""Observed-count tabulation on the GPU for many cohorts at once: mutations x element blocks -> OBS_SNV,
OBS_SAMPLES, OBS_INDEL as [E, C] int32 tensors, ready for dig_element_stats.

Same integer semantics as mutation_tools.tabulate_mutations_in_element (reference mutation_tools.py:155-230):
interval join -> drop duplicates on (chrom, start, end, ref, alt, sample, element) -> SNV / INDEL counts per
(element, sample) -> drop samples whose total count exceeds max_muts_per_sample -> cap the per-(element, sample)
counts -> per element: number of distinct samples, sum of SNVs, sum of indels.  The join runs in the HIP kernels
dig_overlap_join_{count,fill}; de-duplication and the segmented counts are sort/unique calls on the device
(PyTorch as plumbing).  Results are bit-identical to the host path (tests/test_gpu_tabulate.py).
"""
import numpy as np

from .. import _lib
from . import mutation_tools


class ElementBlocks:
    """bed6 blocks of an element set, sorted by (chrom, start), as device tensors with composite keys."""

    def __init__(self, chrom, start, end, elt_id, n_elements, device):
        import torch
        chrom = np.asarray(chrom, np.int64)
        start = np.asarray(start, np.int64)
        end = np.asarray(end, np.int64)
        end_eff = np.where(end == start, start + 1, end)
        order = np.lexsort((start, chrom))
        chrom, start, end_eff, elt_id = chrom[order], start[order], end_eff[order], np.asarray(elt_id, np.int64)[order]
        runmax = np.empty_like(end_eff)
        for c in np.unique(chrom):
            sel = chrom == c
            runmax[sel] = np.maximum.accumulate(end_eff[sel])
        t = lambda a: torch.as_tensor(np.ascontiguousarray(a), device=device)
        self.start_key, self.runmax_key, self.end = t((chrom << 40) | start), t((chrom << 40) | runmax), t(end_eff)
        self.elt = t(elt_id)
        self.n_blocks, self.n_elements, self.device = len(start), int(n_elements), device

    @classmethod
    def from_bed12(cls, f_bed, device, names=None):
        """bed12 file -> blocks; element ids follow `names` (default: first appearance in the file).
        Chromosome labels are compared as TEXT, as bedtools does (mutation_tools.py:193-200): every label of the bed
        file gets an id (`blocks.chrom_ids`, pass it to encode_mutations); '1' and 'chr1' are different chromosomes and
        sex chromosomes are joined like any other."""
        import pandas as pd
        df = pd.read_csv(f_bed, sep="\t", header=None, low_memory=False, dtype={0: str})
        blocks = mutation_tools._bed12_to_bed6(df)
        labels = blocks.CHROM.astype(str)
        chrom_ids = {lab: i + 1 for i, lab in enumerate(dict.fromkeys(labels.tolist()))}
        if names is None:
            names = list(dict.fromkeys(df[3].tolist()))
        pos = {n: i for i, n in enumerate(names)}
        obj = cls(labels.map(chrom_ids).values, blocks.START.values, blocks.END.values, blocks.ELT.map(pos).values,
                  len(names), device)
        obj.chrom_ids = chrom_ids
        return obj, names


def overlap_pairs(blocks, m_chrom, m_start, m_end):
    """(mutation row, block row) pairs on the device: int32 tensors, mutation-major, blocks ascending."""
    import torch
    dev = blocks.device
    n = m_chrom.numel()
    counts = torch.zeros(n, dtype=torch.int32, device=dev)
    args = [_lib.dev_ptr(blocks.start_key), _lib.dev_ptr(blocks.runmax_key), _lib.dev_ptr(blocks.end), blocks.n_blocks,
            _lib.dev_ptr(m_chrom), _lib.dev_ptr(m_start), _lib.dev_ptr(m_end), n]
    with torch.cuda.device(dev):
        _lib.call("dig_overlap_join_count", *args, _lib.dev_ptr(counts), _lib.stream_ptr())
        incl = torch.cumsum(counts, 0, dtype=torch.int64)
        total = int(incl[-1].item()) if n else 0
        offsets = (incl - counts).contiguous()
        pm = torch.empty(max(total, 1), dtype=torch.int32, device=dev)
        pb = torch.empty(max(total, 1), dtype=torch.int32, device=dev)
        if total:
            _lib.call("dig_overlap_join_fill", *args, _lib.dev_ptr(offsets), _lib.dev_ptr(pm), _lib.dev_ptr(pb),
                      _lib.stream_ptr())
    return pm[:total], pb[:total]


def encode_mutations_host(df_mut, cohort_id=0, chrom_ids=None):
    """Mutation frame (reference column names) -> host arrays.  Strings become dense integer ids: `uid` identifies
    (CHROM, START, END, REF, ALT) and `sample` the sample label, both exactly (no hashing).
    `chrom_ids`: label -> id of the block set the mutations will be joined with (ElementBlocks.from_bed12: labels
    compared as text, rows on other chromosomes cannot hit anything and are dropped here).  Without it the blocks
    carry the integer autosome numbers of a pretrained model (bed12_boundaries, mutation_tools.py:383-414) and the
    mutation labels '1' ... '22' (optionally 'chr'-prefixed) are mapped onto them."""
    import pandas as pd
    if chrom_ids is not None:
        ch = df_mut.CHROM.astype(str).map(chrom_ids)
        keep = ch.notna()
        df_mut, ch = df_mut[keep], ch[keep].astype(np.int64)
    else:
        ch = df_mut.CHROM.astype(str).map(_strip_chr)           # (ONE leading "chr", the rule of all three file routes)
        keep = ch.isin(_AUTOSOME_LABELS) & df_mut.SAMPLE.notna()      # (a missing SAMPLE: the reference's groupby drops such rows)
        df_mut, ch = df_mut[keep], ch[keep].astype(np.int64)
    ref_id = pd.factorize(df_mut.REF.astype(str).values)[0]
    alt_id = pd.factorize(df_mut.ALT.astype(str).values)[0]
    samp, sample_names = pd.factorize(df_mut.SAMPLE.astype(str).values)
    # gene label of the row (get_unique_indels keys on it, mutation_tools.py:111-117); files without the column: one label
    gene = pd.factorize(df_mut.GENE.astype(str).values)[0] if 'GENE' in df_mut.columns else np.zeros(len(df_mut), np.int64)
    return _host_record(ch.values, df_mut.START.values, df_mut.END.values, ref_id, alt_id, samp, list(sample_names), gene,
                        (df_mut.ANNOT == 'INDEL').values, cohort_id)


def _host_record(ch, start, end, ref_id, alt_id, samp, sample_names, gene, indel, cohort_id):
    # exact ids of the distinct (CHROM, START, END, REF, ALT): the two string columns are factorized on their own (few
    # distinct alleles), then the five integer columns are grouped as rows (a MultiIndex factorize of the same columns
    # took 0.5 s per 200 000 mutations, 85 % of the many-cohort driver's wall time)
    ch, start, end = np.asarray(ch, np.int64), np.asarray(start, np.int64), np.asarray(end, np.int64)
    ref_id, alt_id = np.asarray(ref_id, np.int64), np.asarray(alt_id, np.int64)
    n = len(ch)
    if n == 0:
        uid = np.zeros(0, np.int64)
    else:
        span = end - start
        if (ch.min() >= 0 and ch.max() < (1 << 20) and start.min() >= 0 and start.max() < (1 << 40) and span.min() >= 0 and
                span.max() < (1 << 22) and ref_id.max() < (1 << 20) and alt_id.max() < (1 << 20)):
            # two packed 62-bit keys and one two-key sort (np.unique over the five-column rows: 0.6 s per 300 000 mutations)
            k1, k2 = (ch << 40) | start, (span << 40) | (ref_id << 20) | alt_id
            order = np.lexsort((k2, k1))
            new = np.concatenate([[True], (k1[order][1:] != k1[order][:-1]) | (k2[order][1:] != k2[order][:-1])])
            uid = np.empty(n, np.int64)
            uid[order] = np.cumsum(new) - 1
        else:
            uid = np.unique(np.stack([ch, start, end, ref_id, alt_id], axis=1), axis=0, return_inverse=True)[1].reshape(-1)
    c = lambda a: np.ascontiguousarray(a, dtype=np.int64)
    first_row, first_indel = dedup_flags(uid, samp, gene, indel)
    return dict(chrom=c(ch), start=c(start), end=c(end), uid=c(uid), sample=c(samp), indel=c(indel), gene=c(gene),
                first_row=first_row, first_indel=first_indel,
                cohort=np.full(n, int(cohort_id), np.int64), sample_names=list(sample_names))


def dedup_flags(uid, sample, gene, indel):
    """The reference's two de-duplications of a cohort's rows, one after the other (read_mutation_file(drop_duplicates=True,
    unique_indels=True), mutation_tools.py:106-117), as per-row 0 / 1 flags in file order: first_row[i] -- no earlier row has
    the same mutation id and sample (drop_duplicate_mutations keeps it); first_indel[i] -- row i is such a row, is an INDEL,
    and no earlier such row has the same mutation id and GENE label (get_unique_indels keeps it).  (The native parser's
    dig_mutation_file_flags_host gives the same arrays.)"""
    uid, sample, gene = np.asarray(uid, np.int64), np.asarray(sample, np.int64), np.asarray(gene, np.int64)
    indel = np.asarray(indel).astype(bool)
    n = len(uid)
    first_row, first_indel = np.zeros(n, np.int64), np.zeros(n, np.int64)
    if n:
        order = np.lexsort((np.arange(n), sample, uid))
        new = np.concatenate([[True], (uid[order][1:] != uid[order][:-1]) | (sample[order][1:] != sample[order][:-1])])
        first_row[order[new]] = 1
        rows = np.flatnonzero((first_row == 1) & indel)
        if len(rows):
            o2 = np.lexsort((rows, gene[rows], uid[rows]))
            r2 = rows[o2]
            new2 = np.concatenate([[True], (uid[r2][1:] != uid[r2][:-1]) | (gene[r2][1:] != gene[r2][:-1])])
            first_indel[r2[new2]] = 1
    return first_row, first_indel


def _encode_mutation_file_native(path, cohort_id):
    """encode_mutation_file through the library's parser (dig_mutation_file_*_host, include/dig_hip.h): the same arrays, no
    interpreter lock held while a file is parsed -- 37 files side by side: 0.83 s -> 0.1 s on the 256 cores of the GPU box.
    None when the file holds something the parser leaves to the Python path (quotes, ragged rows, non-integer coordinates)."""
    import ctypes
    import os
    from .. import _lib
    h, n, ns, nb = ctypes.c_void_p(), ctypes.c_int64(-1), ctypes.c_int64(0), ctypes.c_int64(0)
    _lib.call("dig_mutation_file_parse_host", os.fsencode(path), ctypes.byref(h), ctypes.byref(n), ctypes.byref(ns), ctypes.byref(nb))
    if n.value < 0:
        return None
    try:
        cols = [np.empty(n.value, np.int64) for _ in range(9)]
        names = ctypes.create_string_buffer(max(1, nb.value))
        _lib.call("dig_mutation_file_fetch_host", h, *[c.ctypes.data for c in cols[:7]], names)
        _lib.call("dig_mutation_file_flags_host", h, cols[7].ctypes.data, cols[8].ctypes.data)
    finally:
        _lib.call("dig_mutation_file_free_host", h)
    sample_names = names.raw[:nb.value].decode().split("\n") if ns.value else []
    ch, start, end, uid, samp, indel, gene, first_row, first_indel = cols
    return dict(chrom=ch, start=start, end=end, uid=uid, sample=samp, indel=indel, gene=gene, first_row=first_row,
                first_indel=first_indel, cohort=np.full(n.value, int(cohort_id), np.int64), sample_names=sample_names)


def encode_mutation_file(path, cohort_id=0, native=True):
    """An annotated mutation file (tab-separated, no header; GENE and ANNOT in columns 6 and 7: the reference's 8-, 10- and
    11-column schemas, mutation_tools.py:56-78) -> the host arrays of encode_mutations_host, without a pandas frame in between.
    Three routes, tried in this order, that give the same arrays on the files all of them take:
      * the library's own parser (`native`; it leaves quotes, ragged rows, non-integer coordinates, other column counts and
        label fields that pandas reads as missing -- '', 'NA', 'nan', 'NULL', ... -- to the next routes);
      * pyarrow's multi-threaded reader + dictionary encoding (0.12 s per 300 000 rows; leaves the same files to pandas);
      * pandas, with the reference's schema for the file's column count (what read_mutation_file does: the fall-back that
        defines the semantics -- missing labels are NaN, a row whose SAMPLE is missing is dropped as the reference's groupby does).
    Chromosome labels '1' ... '22', ONE leading 'chr' allowed, in every route."""
    if native:
        enc = _encode_mutation_file_native(path, cohort_id)
        if enc is not None:
            return enc

    def pandas_route():
        from . import mutation_tools
        try:
            df = mutation_tools.read_mutation_file(path, drop_sex=False, drop_duplicates=False, unique_indels=False)
        except StopIteration:                    # (the reference's next(reader) on an empty file)
            raise ValueError("%s: no rows (an annotated mutation file has 7 or more tab-separated columns, mutation_tools.py:45-104)" % path) from None
        if 'ANNOT' not in df.columns:
            raise ValueError("%s: an annotated mutation file has an ANNOT column (7 or more columns, mutation_tools.py:45-104)" % path)
        return encode_mutations_host(df, cohort_id)

    try:
        import pyarrow as pa
        import pyarrow.csv as pcsv
    except ImportError:
        return pandas_route()
    names = ['CHROM', 'START', 'END', 'REF', 'ALT', 'SAMPLE', 'GENE', 'ANNOT']
    with open(path, 'rb') as f:
        n_cols = f.readline().count(b"\t") + 1
    if n_cols not in (8, 10, 11):
        return pandas_route()
    all_names = names + ['X%d' % i for i in range(n_cols - 8)]
    text = {k: pa.string() for k in ('CHROM', 'REF', 'ALT', 'SAMPLE', 'GENE', 'ANNOT')}
    try:
        tb = pcsv.read_csv(path, read_options=pcsv.ReadOptions(column_names=all_names, use_threads=True, block_size=8 << 20),
                           parse_options=pcsv.ParseOptions(delimiter="\t"),
                           convert_options=pcsv.ConvertOptions(column_types=dict(text, START=pa.int64(), END=pa.int64()), include_columns=names,
                                                               strings_can_be_null=False))
    except pa.ArrowInvalid:                      # a header line, a coordinate that is not an integer: pandas says what it says
        return pandas_route()

    def ids(col):
        d = tb[col].combine_chunks().dictionary_encode()
        return d.indices.to_numpy(zero_copy_only=False).astype(np.int64), d.dictionary.to_pylist()
    ch_idx, ch_labels = ids('CHROM')
    label_to_int = np.array([int(_strip_chr(l)) if _strip_chr(l) in _AUTOSOME_LABELS else -1 for l in ch_labels] or [-1], np.int64)
    ch = label_to_int[ch_idx] if len(ch_idx) else np.zeros(0, np.int64)
    keep = ch > 0
    ref_id, ref_labels = ids('REF')
    alt_id, alt_labels = ids('ALT')
    samp, sample_labels = ids('SAMPLE')
    gene, gene_labels = ids('GENE')
    annot_idx, annot_labels = ids('ANNOT')
    if any(l in _PANDAS_NA for labels in (ref_labels, alt_labels, sample_labels, gene_labels, annot_labels) for l in labels):
        return pandas_route()                    # pandas reads such a label as missing
    indel = (np.array([l == 'INDEL' for l in annot_labels] or [False])[annot_idx]) if len(annot_idx) else np.zeros(0, bool)
    start = tb['START'].combine_chunks().to_numpy(zero_copy_only=False)
    end = tb['END'].combine_chunks().to_numpy(zero_copy_only=False)
    # sample ids in order of first appearance among the kept rows (what pandas.factorize gives encode_mutations_host)
    samp_k = samp[keep]
    first = np.full(len(sample_labels), np.iinfo(np.int64).max, np.int64)
    np.minimum.at(first, samp_k, np.arange(len(samp_k)))
    order = np.argsort(first, kind="stable")
    order = order[first[order] < np.iinfo(np.int64).max]
    remap = np.full(len(sample_labels), -1, np.int64)
    remap[order] = np.arange(len(order))
    return _host_record(ch[keep], start[keep], end[keep], ref_id[keep], alt_id[keep], remap[samp_k], [sample_labels[i] for i in order],
                        gene[keep], indel[keep], cohort_id)


_AUTOSOME_LABELS = frozenset(str(i) for i in range(1, 23))
# what pandas.read_csv reads as a missing value (its default na_values)
_PANDAS_NA = frozenset(["", "#N/A", "#N/A N/A", "#NA", "-1.#IND", "-1.#QNAN", "-NaN", "-nan", "1.#IND", "1.#QNAN", "<NA>", "N/A", "NA",
                        "NULL", "NaN", "None", "n/a", "nan", "null"])


def _strip_chr(label):
    """ONE leading 'chr' off a chromosome label (the rule of the native parser's autosome())."""
    return label[3:] if label.startswith("chr") else label


def to_device(enc, device):
    """Host arrays of encode_mutations_host / encode_mutation_file -> device tensors (int64, as the join and the tabulation take
    them).  The upload is narrow -- ids and coordinates that fit 32 bits travel as int32, the three flags as bytes, the constant
    cohort column not at all -- and widened on the device: 31 instead of 88 bytes per row over PCIe from pageable memory."""
    import torch
    out = {}
    n = len(enc["chrom"])
    for k, v in enc.items():
        if k == "sample_names":
            continue
        v = np.asarray(v)
        if k == "cohort":
            out[k] = torch.full((n,), int(v[0]) if n else 0, dtype=torch.int64, device=device)
        elif k in ("indel", "first_row", "first_indel"):
            out[k] = torch.as_tensor(v.astype(np.uint8), device=device).long()
        elif n and v.dtype.kind in "iu" and int(v.min()) >= -2 ** 31 and int(v.max()) < 2 ** 31:
            out[k] = torch.as_tensor(v.astype(np.int32), device=device).long()
        else:
            out[k] = torch.as_tensor(v, device=device)
    out["sample_names"] = list(enc["sample_names"])
    return out


def encode_mutations(df_mut, device, cohort_id=0, chrom_ids=None):
    """Mutation frame (reference column names) -> device tensors (encode_mutations_host + to_device)."""
    return to_device(encode_mutations_host(df_mut, cohort_id, chrom_ids), device)


def tabulate_cohorts(blocks, cohorts, drop_duplicates=True, max_muts_per_sample=1e9, max_muts_per_elt_per_sample=3e9):
    """`cohorts`: list of encode_mutations() dicts (one per cohort).  Returns (OBS_SNV, OBS_SAMPLES, OBS_INDEL) int32
    [E, C] device tensors and the per-cohort lists of blacklisted sample names."""
    import torch
    dev = blocks.device
    C, E = len(cohorts), blocks.n_elements
    cat = lambda k: torch.cat([c[k] for c in cohorts])
    # sample ids are per cohort: make them globally distinct with an offset
    offs = np.concatenate([[0], np.cumsum([len(c["sample_names"]) for c in cohorts])])
    sample = torch.cat([c["sample"] + int(o) for c, o in zip(cohorts, offs[:-1])])
    uid_off = np.concatenate([[0], np.cumsum([int(c["uid"].max().item()) + 1 if c["uid"].numel() else 0 for c in cohorts])])
    uid = torch.cat([c["uid"] + int(o) for c, o in zip(cohorts, uid_off[:-1])])
    chrom, start, end, indel, cohort = cat("chrom"), cat("start"), cat("end"), cat("indel"), cat("cohort")
    pm, pb = overlap_pairs(blocks, chrom, start, end)
    pm = pm.long()
    elt = blocks.elt[pb.long()]
    rec = torch.stack([cohort[pm], elt, sample[pm], uid[pm], indel[pm]], dim=1)        # [n_pairs, 5]
    if drop_duplicates and rec.shape[0]:
        # duplicates on (chrom, start, end, ref, alt, sample, element): ANNOT is a function of the mutation here
        rec = _unique_rows_keep_flag(rec)
    # per (cohort, element, sample): SNV and INDEL counts
    key3, inv = torch.unique(rec[:, :3], dim=0, return_inverse=True)
    n3 = key3.shape[0]
    snv = torch.zeros(n3, dtype=torch.int64, device=dev).index_add_(0, inv, 1 - rec[:, 4])
    ind = torch.zeros(n3, dtype=torch.int64, device=dev).index_add_(0, inv, rec[:, 4])
    # hypermutator blacklist per (cohort, sample): total over elements of OBS_MUT
    n_samp_tot = int(offs[-1])
    tot = torch.zeros(max(n_samp_tot, 1), dtype=torch.int64, device=dev).index_add_(0, key3[:, 2], snv + ind)
    black = tot > max_muts_per_sample
    keep = ~black[key3[:, 2]]
    key3, snv, ind = key3[keep], snv[keep], ind[keep]
    cap = int(min(max_muts_per_elt_per_sample, 2 ** 62))
    snv, ind = snv.clamp(max=cap), ind.clamp(max=cap)
    flat = key3[:, 1] * C + key3[:, 0]                                                  # [E, C] layout
    z = lambda: torch.zeros(E * C, dtype=torch.int64, device=dev)
    obs_snv = z().index_add_(0, flat, snv).view(E, C).to(torch.int32)
    obs_ind = z().index_add_(0, flat, ind).view(E, C).to(torch.int32)
    obs_smp = z().index_add_(0, flat, torch.ones_like(snv)).view(E, C).to(torch.int32)
    black_np = black.cpu().numpy()
    blacklists = [[n for j, n in enumerate(c["sample_names"]) if black_np[int(o) + j]] for c, o in zip(cohorts, offs[:-1])]
    return obs_snv, obs_smp, obs_ind, blacklists


def _unique_rows_keep_flag(rec):
    """Unique on the first four columns (cohort, element, sample, mutation uid), keeping the indel flag of the
    first occurrence (identical for all occurrences: the flag is a property of the mutation)."""
    import torch
    key, inv = torch.unique(rec[:, :4], dim=0, return_inverse=True)
    flag = torch.zeros(key.shape[0], dtype=rec.dtype, device=rec.device)
    flag.scatter_reduce_(0, inv, rec[:, 4], reduce="amax", include_self=True)
    return torch.cat([key, flag[:, None]], dim=1)
