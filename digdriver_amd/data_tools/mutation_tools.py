"""Mutation-file handling and integer observed-count tabulation.

Host mirror of DIGDriver/data_tools/mutation_tools.py.  Same function names, arguments and output
frames; the reference shells out to ``bedtools intersect`` through pybedtools, here the interval
join is a sort + binary-search in numpy (half-open ``[START, END)`` overlap, exactly bedtools'
default: two intervals overlap when ``a.start < b.end and b.start < a.end``; zero-length features
are treated like bedtools does, as overlapping when ``b.start <= a.start < b.end``... see
``_overlap_pairs``).  All outputs of this module are integer counts and must be bit-exact.
"""
import csv
import gzip

import numpy as np
import pandas as pd

_MUT_COLS = {
    5: ['CHROM', 'POS', 'REF', 'ALT', 'SAMPLE'],
    6: ['CHROM', 'START', 'END', 'REF', 'ALT', 'SAMPLE'],
    7: ['CHROM', 'START', 'END', 'REF', 'ALT', 'SAMPLE', 'ANNOT'],
    8: ['CHROM', 'START', 'END', 'REF', 'ALT', 'SAMPLE', 'GENE', 'ANNOT'],
    9: ['CHROM', 'START', 'END', 'REF', 'ALT', 'SAMPLE', 'ANNOT', 'MUT_TYPE', 'CONTEXT'],
    10: ['CHROM', 'START', 'END', 'REF', 'ALT', 'SAMPLE', 'GENE', 'ANNOT', 'MUT_TYPE', 'CONTEXT'],
    11: ['CHROM', 'START', 'END', 'REF', 'ALT', 'SAMPLE', 'GENE', 'ANNOT', 'MUT_TYPE', 'CONTEXT', 'STRAND'],
}
_INT_COLS = ('POS', 'START', 'END')
_AUTOSOMES = [str(i) for i in range(1, 23)]


_MUT_KEY = ['CHROM', 'START', 'END', 'REF', 'ALT']


def _first_row_width(path):
    """Number of tab-separated fields of the first row of a plain or gzip file."""
    for opener in (open, lambda q: gzip.open(q, 'rt')):
        try:
            with opener(path) as handle:
                return len(next(csv.reader(handle, delimiter='\t', skipinitialspace=True)))
        except UnicodeDecodeError:
            continue
    raise ValueError("cannot read {}".format(path))


def read_mutation_file(path, drop_sex=True, drop_duplicates=False, unique_indels=True):
    """mutation_tools.py:45-104: header-less TSV whose schema follows from its column count (5 ... 11), plain or gzip;
    optionally autosomes only (CHROM becomes int), unique rows, unique indels."""
    names = _MUT_COLS[_first_row_width(path)]
    rows = pd.read_csv(path, sep="\t", names=names, low_memory=False, dtype={c: (int if c in _INT_COLS else str) for c in names})
    if drop_sex:
        autosomal = rows.CHROM.isin(_AUTOSOMES)
        if not autosomal.all():
            print('Restricting to autosomes')
            rows = rows.loc[autosomal]
        rows = rows.assign(CHROM=rows.CHROM.astype(int))
    if drop_duplicates:
        rows = drop_duplicate_mutations(rows)
    return get_unique_indels(rows) if unique_indels else rows


def drop_duplicate_mutations(df_mut):
    """mutation_tools.py:107-109: one row per (mutation, sample)."""
    return df_mut.drop_duplicates(_MUT_KEY + ['SAMPLE'])


def get_unique_indels(df_mut):
    """mutation_tools.py:111-117: an indel seen in several samples is kept once per GENE label; SNV rows come first."""
    indel = df_mut.ANNOT == 'INDEL'
    return pd.concat([df_mut.loc[~indel], df_mut.loc[indel].drop_duplicates(subset=_MUT_KEY + ['GENE'])])


def filter_hypermut_samples(df_mut, max_muts_per_sample, return_blacklist=False):
    """mutation_tools.py:293-304: drop every row of the samples with more than `max_muts_per_sample` rows."""
    load = df_mut.SAMPLE.value_counts()
    hyper = load.index[load > max_muts_per_sample].to_list()
    kept = df_mut.loc[~df_mut.SAMPLE.isin(hyper)]
    return (kept, hyper) if return_blacklist else kept


_ANNOT_TO_OBS = {'Missense': 'OBS_MIS', 'Nonsense': 'OBS_NONS', 'Synonymous': 'OBS_SYN', 'Essential_Splice': 'OBS_SPL',
                 'INDEL': 'OBS_INDEL'}


def mutations_per_gene(df_mut_cds, max_muts_per_gene_per_sample=3e9):
    """mutation_tools.py:329-361: rows per (GENE, SAMPLE, ANNOT), capped per sample, summed per (GENE, ANNOT) -> integer
    frame indexed by GENE with one column per annotation class present, the five tested classes named OBS_* (0 when the
    cohort has none)."""
    per_sample = df_mut_cds.groupby(['GENE', 'SAMPLE', 'ANNOT']).size().clip(upper=max_muts_per_gene_per_sample)
    table = per_sample.groupby(level=['GENE', 'ANNOT']).sum().unstack('ANNOT').fillna(0).astype(int)
    table.columns = table.columns.to_list()
    for annot in _ANNOT_TO_OBS:
        if annot not in table.columns:
            table[annot] = 0
    return table.rename(columns=_ANNOT_TO_OBS)


def bed12_boundaries(f_bed):
    """mutation_tools.py:383-414: bed12 -> frame (CHROM int, ELT, STRAND, BLOCK_STARTS, BLOCK_ENDS), autosomes only.
    As in the reference the 'chr' prefix is stripped only when the FIRST row carries it."""
    names = ['CHROM', 'START', 'END', "ELT", "SCORE", "STRAND", 'thickStart', 'thickEnd', 'rgb', 'blockCount',
             'blockSizes', 'blockStarts']
    df = pd.read_table(f_bed, names=names, low_memory=False)
    df.CHROM = df.CHROM.astype(str)
    df.blockSizes = df.blockSizes.astype(str)
    df.blockStarts = df.blockStarts.astype(str)
    if 'chr' in str(df['CHROM'][0]):
        df['CHROM'] = df['CHROM'].map(lambda x: x.lstrip('chr'))
    df = df[df.CHROM.isin(_AUTOSOMES)].copy()
    df.CHROM = df.CHROM.astype(int)

    def _ints(s):
        return [int(x) for x in s.rstrip(',').split(',')]

    starts = [[x + st for x in _ints(bs)] for bs, st in zip(df.blockStarts, df.START)]
    ends = [[s + z for s, z in zip(ss, _ints(bz))] for ss, bz in zip(starts, df.blockSizes)]
    df['BLOCK_STARTS'] = starts
    df['BLOCK_ENDS'] = ends
    return df[['CHROM', 'ELT', 'STRAND', 'BLOCK_STARTS', 'BLOCK_ENDS']]


# ---------------------------------------------------------------------------------------------
# interval join (replaces `bedtools intersect -wa -wb` on mutations x bed6 blocks)
# ---------------------------------------------------------------------------------------------
def _chrom_key(chrom):
    """Chromosome labels compare as text in bedtools: '1' and 'chr1' are different chromosomes."""
    return np.asarray(chrom).astype(str)


def _overlap_pairs(m_chrom, m_start, m_end, b_chrom, b_start, b_end):
    """All (mutation row, block row) pairs with bedtools-intersect semantics:
    overlap iff  m.start < b.end  and  b.start < m.end  (half-open).  A zero-length mutation
    (start == end, bedtools >= 2.27) overlaps block b iff  b.start <= m.start < b.end ... bedtools treats
    it as the 1-bp feature [start, start+1) for the test; the same is done for zero-length blocks.
    Output order: mutation-major (file order of the mutation file), blocks in bed order."""
    m_chrom, b_chrom = _chrom_key(m_chrom), _chrom_key(b_chrom)
    m_start = np.asarray(m_start, np.int64)
    m_end = np.asarray(m_end, np.int64)
    b_start = np.asarray(b_start, np.int64)
    b_end = np.asarray(b_end, np.int64)
    m_end_eff = np.where(m_end == m_start, m_start + 1, m_end)
    b_end_eff = np.where(b_end == b_start, b_start + 1, b_end)
    mi_all, bi_all = [], []
    for ch in np.unique(b_chrom):
        bsel = np.nonzero(b_chrom == ch)[0]
        msel = np.nonzero(m_chrom == ch)[0]
        if len(bsel) == 0 or len(msel) == 0:
            continue
        order = np.argsort(b_start[bsel], kind="stable")
        bs, be, bidx = b_start[bsel][order], b_end_eff[bsel][order], bsel[order]
        # blocks may overlap/nest: candidates are all blocks with start < m.end; filter by end > m.start.
        # running max of block ends bounds the scan from the left
        run_max = np.maximum.accumulate(be)
        hi = np.searchsorted(bs, m_end_eff[msel], side="left")        # blocks [0, hi) have start < m.end
        lo = np.searchsorted(run_max, m_start[msel], side="right")     # first block whose running max end > m.start
        cnt = np.maximum(hi - lo, 0)
        tot = int(cnt.sum())
        if tot == 0:
            continue
        rep_m = np.repeat(np.arange(len(msel)), cnt)
        offs = np.arange(tot) - np.repeat(np.cumsum(cnt) - cnt, cnt)
        cand = np.repeat(lo, cnt) + offs
        keep = be[cand] > m_start[msel][rep_m]
        mi_all.append(msel[rep_m[keep]])
        bi_all.append(bidx[cand[keep]])
    if not mi_all:
        return np.zeros(0, np.int64), np.zeros(0, np.int64)
    mi = np.concatenate(mi_all)
    bi = np.concatenate(bi_all)
    order = np.lexsort((bi, mi))
    return mi[order], bi[order]


def _bed12_to_bed6(df_bed12):
    """bedtools bed12tobed6: one row per block, keeping the parent's name and strand."""
    cols = ['CHROM', 'START', 'END', 'ELT', 'STRAND']
    if len(df_bed12) and df_bed12[10].dtype == object and df_bed12[11].dtype == object:
        # the usual file -- both lists of every row with, or of every row without, the trailing comma -- by array operations: the
        # lists joined into one text and read as numbers in one call (0.13 -> 0.04 s per 120 000 elements)
        try:
            a, b = df_bed12[10].values.astype(str), df_bed12[11].values.astype(str)
            ta, tb = np.char.endswith(a, ','), np.char.endswith(b, ',')
            if (ta.all() or not ta.any()) and (tb.all() or not tb.any()):
                n = np.char.count(a, ',') + (0 if ta.all() else 1)
                if np.array_equal(n, np.char.count(b, ',') + (0 if tb.all() else 1)) and (n > 0).all():
                    sizes = np.fromstring(('' if ta.all() else ',').join(a.tolist()), dtype=np.int64, sep=',')
                    starts = np.fromstring(('' if tb.all() else ',').join(b.tolist()), dtype=np.int64, sep=',')
                    if len(sizes) == len(starts) == int(n.sum()):
                        rep = np.repeat(np.arange(len(a)), n)
                        s0 = np.asarray(df_bed12[1].values, np.int64)[rep] + starts
                        return pd.DataFrame({'CHROM': df_bed12[0].values[rep], 'START': s0, 'END': s0 + sizes,
                                             'ELT': df_bed12[3].values[rep], 'STRAND': df_bed12[5].values[rep]}, columns=cols)
        except (ValueError, TypeError):
            pass                                   # (anything unusual in the lists: the row-by-row forms below)
    bsz = [str(x).rstrip(',') for x in df_bed12[10]]
    bst = [str(x).rstrip(',') for x in df_bed12[11]]
    n = np.fromiter((x.count(',') + 1 for x in bsz), np.int64, len(bsz))
    if len(bsz) and np.array_equal(n, np.fromiter((x.count(',') + 1 for x in bst), np.int64, len(bst))):
        # all rows at once (a Python loop over the blocks of 120 000 elements was 0.3 s of a command-line run)
        sizes = np.array(",".join(bsz).split(','), dtype=np.int64)
        starts = np.array(",".join(bst).split(','), dtype=np.int64)
        rep = np.repeat(np.arange(len(bsz)), n)
        s0 = np.asarray(df_bed12[1].values, np.int64)[rep] + starts
        return pd.DataFrame({'CHROM': df_bed12[0].values[rep], 'START': s0, 'END': s0 + sizes, 'ELT': df_bed12[3].values[rep],
                             'STRAND': df_bed12[5].values[rep]}, columns=cols)
    rows = []                                      # (rows whose two lists differ in length: block by block, the shorter list counts)
    for chrom, start, name, strand, z_, s_ in zip(df_bed12[0], df_bed12[1], df_bed12[3], df_bed12[5], bsz, bst):
        for s, z in zip([int(x) for x in s_.split(',')], [int(x) for x in z_.split(',')]):
            rows.append((chrom, start + s, start + s + z, name, strand))
    return pd.DataFrame(rows, columns=cols)


def tabulate_muts_per_sample_per_element(f_mut, f_elt_bed, bed12=False, drop_duplicates=False, unique_indels=True):
    """mutation_tools.py:191-230: mutations x element blocks -> per (ELT, SAMPLE) integer counts
    OBS_SNV / OBS_INDEL / OBS_MUT.  The raw file rows are joined (no autosome filter), duplicates on
    (chrom, start, end, ref, alt, sample, elt) dropped when asked (:208), SNV vs INDEL by ANNOT (:211-213)."""
    muts = pd.read_csv(f_mut, sep="\t", header=None, low_memory=False, dtype={0: str})
    bed = pd.read_csv(f_elt_bed, sep="\t", header=None, low_memory=False, dtype={0: str})
    blocks = _bed12_to_bed6(bed) if bed12 else bed.rename(columns={0: 'CHROM', 1: 'START', 2: 'END', 3: 'ELT'})
    mi, bi = _overlap_pairs(muts[0].values, muts[1].values, muts[2].values, blocks.CHROM.values, blocks.START.values,
                            blocks.END.values)
    out_cols = ['ELT', 'SAMPLE', 'OBS_SNV', 'OBS_INDEL', 'OBS_MUT']
    if len(mi) == 0:
        return pd.DataFrame({c: [] for c in out_cols})
    hits = pd.DataFrame({'ELT': blocks.ELT.values[bi], 'SAMPLE': muts[5].values[mi],
                         'KIND': np.where(muts[7].values[mi] == 'INDEL', 'OBS_INDEL', 'OBS_SNV')})
    if drop_duplicates:
        ident = muts.iloc[mi, :5].reset_index(drop=True)
        hits = hits.loc[~pd.concat([ident, hits[['SAMPLE', 'ELT']]], axis=1).duplicated().values]
    # counts per (element, sample): the SNV table outer-joined with the INDEL table on (ELT, SAMPLE) (:219-222) -- the
    # row order of the result is pandas' (key-sorted since pandas 2.2, SNV pairs first before)
    snv = hits.loc[hits.KIND == 'OBS_SNV'].groupby(['ELT', 'SAMPLE']).size().reset_index(name='OBS_SNV')
    ind = hits.loc[hits.KIND == 'OBS_INDEL'].groupby(['ELT', 'SAMPLE']).size().reset_index(name='OBS_INDEL')
    wide = snv.merge(ind, how='outer').fillna({'OBS_SNV': 0.0, 'OBS_INDEL': 0.0})
    wide['OBS_MUT'] = wide.OBS_SNV + wide.OBS_INDEL
    return wide[out_cols].rename_axis(columns=None)


def _summary_by_codes(f_mut, f_elt_bed, bed12, drop_duplicates, max_muts_per_sample, max_muts_per_elt_per_sample):
    """The per-element summary and the blacklist of tabulate_mutations_in_element WITHOUT the per-(element, sample) frame of
    strings in between: the joined rows carry integer codes of element and sample, the pair counts are one np.unique + two
    bincounts, names come back at the end (the group-by / merge chain over 200 000 string pairs was 0.25 s of a one-cohort
    command line).  The same table, dtypes, row order (elements by name) and blacklist as the frame route below
    (tests/test_host_tabulate.py); None when there is nothing to count (the caller's frame route handles the empty shapes)."""
    muts = pd.read_csv(f_mut, sep="\t", header=None, low_memory=False, dtype={0: str}, usecols=[0, 1, 2, 3, 4, 5, 7])
    bed = pd.read_csv(f_elt_bed, sep="\t", header=None, low_memory=False, dtype={0: str})
    blocks = _bed12_to_bed6(bed) if bed12 else bed.rename(columns={0: 'CHROM', 1: 'START', 2: 'END', 3: 'ELT'})
    mi, bi = _overlap_pairs(muts[0].values, muts[1].values, muts[2].values, blocks.CHROM.values, blocks.START.values,
                            blocks.END.values)
    if len(mi) == 0:
        return None
    ecode, enames = pd.factorize(blocks.ELT.values[bi])
    scode, snames = pd.factorize(muts[5].values[mi])
    indel = muts[7].values[mi] == 'INDEL'
    if drop_duplicates:                              # (chrom, start, end, ref, alt, sample, element), first occurrence kept (:208)
        ident = pd.DataFrame({0: muts[0].values[mi], 1: muts[1].values[mi], 2: muts[2].values[mi], 3: muts[3].values[mi],
                              4: muts[4].values[mi], 5: scode, 6: ecode})
        keep = ~ident.duplicated().values
        ecode, scode, indel = ecode[keep], scode[keep], indel[keep]
    ns, ne = len(snames), len(enames)
    pairs, inv = np.unique(ecode.astype(np.int64) * ns + scode, return_inverse=True)
    ind = np.bincount(inv, weights=indel, minlength=len(pairs))
    snv = np.bincount(inv, weights=~indel, minlength=len(pairs))
    pe, ps = pairs // ns, pairs % ns
    load = np.bincount(ps, weights=snv + ind, minlength=ns)
    black = load > max_muts_per_sample
    blacklist = pd.Index(np.sort(np.asarray(snames, dtype=object)[black]), name='SAMPLE')
    ok = ~black[ps]
    pe, snv, ind = pe[ok], np.minimum(snv[ok], max_muts_per_elt_per_sample), np.minimum(ind[ok], max_muts_per_elt_per_sample)
    if len(pe) == 0:
        return None
    n_pairs = np.bincount(pe, minlength=ne)
    present = np.nonzero(n_pairs)[0]
    names = np.asarray(enames)[present]                  # (numeric names stay numbers, as in the group-by's index)
    if names.dtype == object and all(type(x) is str for x in names):
        order = np.argsort(names.astype(str), kind="stable")          # (code-point order = Python's string order = the group-by's)
    else:
        order = np.argsort(names, kind="stable")
    present = present[order]
    summary = pd.DataFrame({'OBS_INDEL': np.bincount(pe, weights=ind, minlength=ne)[present],
                            'OBS_SAMPLES': n_pairs[present].astype(np.int64),
                            'OBS_SNV': np.bincount(pe, weights=snv, minlength=ne)[present]},
                           index=pd.Index(names[order], name='ELT'))
    return summary, blacklist


def tabulate_mutations_in_element(f_mut, f_elt_bed, bed12=False, drop_duplicates=False, all_elements=False,
                                  max_muts_per_sample=1e9, max_muts_per_elt_per_sample=3e9, return_blacklist=False):
    """mutation_tools.py:155-189: hypermutator blacklist, per-(element, sample) cap, then per-element
    OBS_SAMPLES (= number of distinct samples), OBS_SNV, OBS_INDEL."""
    fast = None
    try:
        fast = _summary_by_codes(f_mut, f_elt_bed, bed12, drop_duplicates, max_muts_per_sample, max_muts_per_elt_per_sample)
    except (ValueError, TypeError, KeyError):            # (an unusual file -- fewer columns, mixed types: the frame route decides)
        fast = None
    if fast is not None:
        summary, blacklist = fast
    else:
        per_pair = tabulate_muts_per_sample_per_element(f_mut, f_elt_bed, bed12=bed12, drop_duplicates=drop_duplicates)
        blacklist = []
        if len(per_pair):
            load = per_pair.groupby('SAMPLE').OBS_MUT.sum()
            blacklist = load.index[load > max_muts_per_sample]
            per_pair = per_pair.loc[~per_pair.SAMPLE.isin(blacklist)]
        capped = per_pair.assign(OBS_SNV=per_pair.OBS_SNV.clip(upper=max_muts_per_elt_per_sample),
                                 OBS_INDEL=per_pair.OBS_INDEL.clip(upper=max_muts_per_elt_per_sample))
        if len(capped):
            summary = capped.groupby('ELT').agg(OBS_INDEL=('OBS_INDEL', 'sum'), OBS_SAMPLES=('SAMPLE', 'size'), OBS_SNV=('OBS_SNV', 'sum'))        # ('size' = the reference's len per group, without a Python call per element: 0.5 s of a 120 000-element run)
        else:
            summary = pd.DataFrame({'OBS_SAMPLES': [], 'OBS_SNV': [], 'OBS_INDEL': [], 'ELT': []}).set_index('ELT')
    if all_elements:                                  # every element of the bed, zero counts included (:176-183)
        listed = pd.read_csv(f_elt_bed, sep="\t", header=None).set_index(3).rename_axis('ELT')
        summary = listed.merge(summary, left_index=True, right_index=True, how='left')
        summary[['OBS_SNV', 'OBS_INDEL', 'OBS_SAMPLES']] = summary[['OBS_SNV', 'OBS_INDEL', 'OBS_SAMPLES']].fillna(0)
    table = summary[['OBS_SAMPLES', 'OBS_SNV', 'OBS_INDEL']]
    return (table, blacklist) if return_blacklist else table


def restrict_mutations_by_bed_efficient(f_mut, f_bed, bed12=False, drop_duplicates=False, drop_sex=False,
                                        replace_cols=False):
    """mutation_tools.py:32-43: `bedtools intersect -wa`: one output row per (mutation, overlapped block) pair."""
    import os
    import tempfile
    df_raw = pd.read_csv(f_mut, sep="\t", header=None, low_memory=False, dtype=str)
    df_bed = pd.read_csv(f_bed, sep="\t", header=None, low_memory=False, dtype={0: str})
    blocks = _bed12_to_bed6(df_bed) if bed12 else df_bed.rename(columns={0: 'CHROM', 1: 'START', 2: 'END'})
    mi, _ = _overlap_pairs(df_raw[0].values, df_raw[1].astype(np.int64).values, df_raw[2].astype(np.int64).values,
                           blocks.CHROM.values, blocks.START.values, blocks.END.values)
    fd, tmp = tempfile.mkstemp(suffix=".tsv")
    os.close(fd)
    try:
        df_raw.iloc[mi].to_csv(tmp, sep="\t", header=False, index=False)
        if len(mi) == 0:
            cols = _MUT_COLS[df_raw.shape[1]]
            return pd.DataFrame({c: [] for c in cols})
        return read_mutation_file(tmp, drop_duplicates=drop_duplicates, drop_sex=drop_sex)
    finally:
        os.remove(tmp)


def restrict_mutations_by_bed(df_mut, df_bed, unique=True, remove_X=True, replace_cols=False):
    """mutation_tools.py:8-30 (frame-level `bedtools intersect`, reports the overlapping part of each mutation:
    for 1-bp SNVs inside a block that is the mutation itself)."""
    if remove_X:
        df_mut = df_mut[df_mut.iloc[:, 0] != "X"]
        df_bed = df_bed[df_bed.iloc[:, 0] != "X"]
    mi, bi = _overlap_pairs(df_mut.iloc[:, 0].values, df_mut.iloc[:, 1].values, df_mut.iloc[:, 2].values,
                            df_bed.iloc[:, 0].values, df_bed.iloc[:, 1].values, df_bed.iloc[:, 2].values)
    inter = df_mut.iloc[mi].copy()
    s = np.maximum(df_mut.iloc[mi, 1].values, df_bed.iloc[bi, 1].values)
    e = np.minimum(df_mut.iloc[mi, 2].values, df_bed.iloc[bi, 2].values)
    inter.iloc[:, 1] = s
    inter.iloc[:, 2] = np.maximum(e, s)
    if unique:
        inter = inter.drop_duplicates()
    return inter


# ---------------------------------------------------------------------------------------------
# sites route: mutations that hit listed (position, substitution) sites exactly
# ---------------------------------------------------------------------------------------------
_SITE_KEY = ['CHROM', 'START', 'END', 'REF', 'ALT', 'GENE', 'ANNOT', 'MUT_TYPE', 'CONTEXT']


def tabulate_nonc_mutations_at_sites(f_sites, f_mut, return_sites=False):
    """mutation_tools.py:232-276.  The sites file has the mutation-file layout with the element name in the SAMPLE
    column; a mutation counts for a site when the nine annotation columns agree.  Indels are dropped.  Per element:
    OBS_SAMPLES = distinct samples, OBS_SNV = matching rows."""
    df_sites = read_mutation_file(f_sites)
    df_sites = df_sites.rename({"SAMPLE": "ELT"}, axis=1)
    assert 'GENE' in df_sites.columns and 'ANNOT' in df_sites.columns and 'MUT_TYPE' in df_sites.columns
    if 'STRAND' not in df_sites.columns:
        print("WARNING: strand column not detected in sites file. Defaulting all sites to + strand.")
        df_sites['STRAND'] = "+"
    df_mut = read_mutation_file(f_mut, drop_duplicates=False)
    assert 'GENE' in df_mut.columns and 'ANNOT' in df_mut.columns and 'MUT_TYPE' in df_mut.columns
    if len(df_mut[df_mut.ANNOT == 'INDEL']):
        print('WARNING: INDELS found in mutation file. Dig sites model is only applicable to SNVs. INDELS will be dropped.')
    df_mut = df_mut[df_mut.ANNOT != 'INDEL']
    at_sites = df_mut.merge(df_sites, on=_SITE_KEY, how='inner')
    counts = at_sites.groupby('ELT').agg(OBS_SAMPLES=('SAMPLE', lambda x: len(set(x))), OBS_SNV=('CHROM', len)).reset_index()
    if return_sites:
        sites = df_sites.drop(columns=['GENE', 'ANNOT', 'REF', 'ALT']).rename(columns={'SAMPLE': 'GENE'})
        return counts, sites
    return counts


def tabulate_sites_in_element(f_sites, f_mut):
    """mutation_tools.py:279-283"""
    return tabulate_nonc_mutations_at_sites(f_sites, f_mut).set_index('ELT')[['OBS_SAMPLES', 'OBS_SNV']]
