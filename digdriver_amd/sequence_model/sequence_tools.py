"""Trinucleotide sequence model: substitution indexing, context counting, sequence-model training.

Mirror of the parts of DIGDriver/sequence_model/sequence_tools.py that the hot path needs
(reverse_complement :18-19, mk_context_sequences :30-40, seq_to_context :42-57, type_mutation :59-65,
count_sequence_context :67-80, mk_mutation_context :232-262, mk_trans_idx :282-289,
train_sequence_model + mutation_freq_conditional :321-373).  The live path is the 192-type / 64-context
model (collapse=False everywhere, DigPreprocess.py:47,91); collapse=True (96 types) is kept for
completeness.
"""
import itertools as it

import numpy as np
import pandas as pd

from ..data_tools import mutation_tools

_COMP = str.maketrans('NTCGA', 'NAGCT')
_DNA = 'ACGT'


def reverse_complement(seq):
    return seq[::-1].translate(_COMP)


def mk_context_sequences(n_up=2, n_down=2, collapse=False):
    centre = 'CT' if collapse else _DNA
    keys = [''.join(t) for t in it.product(*([_DNA] * n_up + [centre] + [_DNA] * n_down))]
    return {k: 0 for k in keys}


def seq_to_context(seq, baseix=2, collapse=False):
    if 'N' in seq:
        return ''
    if collapse and seq[baseix] in 'GA':
        return reverse_complement(seq)
    return seq


def type_mutation(REF, ALT, collapse=False):
    if collapse and REF in 'GA':
        REF, ALT = REF.translate(_COMP), ALT.translate(_COMP)
    return "{}>{}".format(REF, ALT)


_CODE = np.full(256, -1, np.int64)
for _i, _c in enumerate(_DNA):
    _CODE[ord(_c)] = _i


def count_sequence_context(seq, n_up=2, n_down=2, nuc_dict=None, collapse=False):
    """Counts of every (n_up + 1 + n_down)-mer centred on each position; windows containing N are skipped.
    Vectorised (base-4 rolling code + bincount) instead of the reference's per-position Python loop."""
    if nuc_dict is None:
        nuc_dict = mk_context_sequences(n_up=n_up, n_down=n_down, collapse=collapse)
    k = n_up + 1 + n_down
    codes = _CODE[np.frombuffer(seq.upper().encode('ascii'), dtype=np.uint8)]
    if len(codes) < k:
        return nuc_dict
    win = np.lib.stride_tricks.sliding_window_view(codes, k)
    ok = (win >= 0).all(axis=1)
    val = (win[ok] * (4 ** np.arange(k - 1, -1, -1))).sum(axis=1)
    cnt = np.bincount(val, minlength=4 ** k)
    for idx in np.flatnonzero(cnt):
        s = ''.join(_DNA[(idx // 4 ** (k - 1 - j)) % 4] for j in range(k))
        s = seq_to_context(s, baseix=n_up, collapse=collapse)
        nuc_dict[s] += int(cnt[idx])
    return nuc_dict


def mk_mutation_context(n_up=1, n_down=1, collapse=False, return_df=False):
    """Rows (MUT_TYPE, CONTEXT) in the reference's order: per reference base (A, C, G, T; C, T only when
    collapsed) the three substitutions are the outer loop and the contexts the inner one."""
    muts = {'A': ['A>T', 'A>C', 'A>G'], 'C': ['C>A', 'C>G', 'C>T'], 'G': ['G>T', 'G>C', 'G>A'], 'T': ['T>A', 'T>G', 'T>C']}
    tups = []
    for ref in ('CT' if collapse else 'ACGT'):
        keys = [''.join(t) for t in it.product(*([_DNA] * n_up + [ref] + [_DNA] * n_down))]
        tups += list(it.product(muts[ref], keys))
    if return_df:
        return pd.DataFrame(tups, columns=['MUT_TYPE', 'CONTEXT'])
    return {t: 0 for t in tups}


def mk_trans_idx(n_up=1, n_down=1, collapse=False):
    """Sorted 'XYZ>XaZ' strings (sequence_tools.py:282-289)."""
    keys = mk_mutation_context(n_up=n_up, n_down=n_down, collapse=collapse)
    return sorted(k[1] + '>' + k[1][:n_up] + k[0][2] + k[1][n_up + 1:] for k in keys)


def mutation_freq_conditional(df_freq, S_gen):
    """sequence_tools.py:356-373: FREQ = COUNT / genome count of the context."""
    df_freq["FREQ"] = df_freq.COUNT.values / np.array([S_gen[c] for c in df_freq.CONTEXT], dtype=float)
    return df_freq


def sequence_model_counts(df_mut_white, n_up=1, n_down=1):
    """The per-cohort sufficient statistic of the sequence model: 192 integer counts of (MUT_TYPE, CONTEXT) in the
    model's row order.  Shards of one cohort add (all-reduce / rank-ordered all-gather sum)."""
    empty = mk_mutation_context(n_up=n_up, n_down=n_down, collapse=False, return_df=True)
    pos = {(m, c): i for i, (m, c) in enumerate(zip(empty.MUT_TYPE, empty.CONTEXT))}
    cnt = np.zeros(len(empty), np.int64)
    keys = pd.Series(list(zip(df_mut_white.MUT_TYPE, df_mut_white.CONTEXT))).value_counts()
    for key, v in keys.items():
        if key in pos:
            cnt[pos[key]] = v
    return empty, cnt


def train_sequence_model(regions, df_mut, genome_counts, n_up=1, n_down=1, key_prefix=None, counts=None):
    """sequence_tools.py:321-354 -> (df_freq_mut [192: MUT_TYPE, CONTEXT, COUNT, FREQ], df_freq_context [64: FREQ]).
    `counts` lets a caller pass pre-reduced 192-vector counts (multi-GPU / multi-shard path)."""
    if counts is None:
        df_bed = pd.DataFrame(regions, columns=['CHROM', 'START', 'END'])
        white = mutation_tools.restrict_mutations_by_bed(df_mut, df_bed, unique=True, remove_X=False)
        white.columns = df_mut.columns
        df_ct, counts = sequence_model_counts(white, n_up=n_up, n_down=n_down)
    else:
        df_ct = mk_mutation_context(n_up=n_up, n_down=n_down, collapse=False, return_df=True)
    df_ct["COUNT"] = np.asarray(counts, dtype=float)
    df_freq_mut = mutation_freq_conditional(df_ct, genome_counts)
    df_freq_context = df_freq_mut.pivot_table('FREQ', index=['CONTEXT'], aggfunc="sum")
    return df_freq_mut, df_freq_context


# ---------------------------------------------------------------------------------------------
# context counting from sequence on the GPU (reference: pysam fetch + Python loop per region)
# ---------------------------------------------------------------------------------------------
_GENOMES = {}


def load_genome(f_fasta):
    """FASTA -> PackedGenome (4 bits per base, cached in memory per path and as <fasta>.dig4.npz on disk)."""
    from ..data_tools.genome import PackedGenome
    if f_fasta not in _GENOMES:
        _GENOMES[f_fasta] = f_fasta if isinstance(f_fasta, PackedGenome) else PackedGenome.from_fasta(f_fasta)
    return _GENOMES[f_fasta]


def _require_trinuc(n_up, n_down, collapse):
    if (n_up, n_down) != (1, 1):
        raise NotImplementedError("the GPU context counter handles trinucleotides (n_up = n_down = 1), the only configuration "
                                  "the driver path uses (onthefly_tools.py:70-71,120)")


def count_contexts_by_regions(f_fasta, chrom_lst, start_lst, end_lst, n_up=2, n_down=2, collapse=False):
    """sequence_tools.py:82-99: frame [regions x 64 contexts] (columns in mk_context_sequences order, index
    "{CHROM}:{START}-{END}") -- one dig_count_contexts2 launch for all regions; collapse=True: the 32 pyrimidine-centred
    contexts.  `f_fasta`: path or PackedGenome."""
    from .. import engine
    _require_trinuc(n_up, n_down, collapse)
    genome = f_fasta if hasattr(f_fasta, "words") else load_genome(f_fasta)
    cnt = engine.count_contexts(genome, list(chrom_lst), np.asarray(start_lst, np.int64), np.asarray(end_lst, np.int64))
    idx = ["{}:{}-{}".format(c, s, e) for c, s, e in zip(chrom_lst, start_lst, end_lst)]
    cnt = cnt.cpu().numpy().astype(np.int64)
    ctx64 = list(mk_context_sequences(1, 1).keys())
    if not collapse:
        return pd.DataFrame(cnt, index=idx, columns=ctx64)
    # collapse=True (the K = 96 model): a window centred on A or G counts as its reverse complement (seq_to_context,
    # sequence_tools.py:42-55): the 32 pyrimidine-centred columns, each the sum of a context and its reverse complement
    pos = {c: i for i, c in enumerate(ctx64)}
    ctx32 = list(mk_context_sequences(1, 1, collapse=True).keys())
    cols = np.array([pos[c] for c in ctx32]), np.array([pos[reverse_complement(c)] for c in ctx32])
    return pd.DataFrame(cnt[:, cols[0]] + cnt[:, cols[1]], index=idx, columns=ctx32)


def nonc_elt_context_count(regions, trans_idx, f_fasta, n_up=1, n_down=1):
    """sequence_tools.py:527-566: `regions` = (chrom, start, end, strand) tuples; '-' / -1 strand regions count the
    reverse-complemented sequence; result [regions x 192] with the sorted substitution keys as columns, every
    substitution column holding the count of its context; index "chr{chrom}:{start}-{end}"."""
    from .. import engine
    _require_trinuc(n_up, n_down, False)
    genome = f_fasta if hasattr(f_fasta, "words") else load_genome(f_fasta)
    chroms = ['chr' + str(r[0]) for r in regions]
    starts = np.array([r[1] for r in regions], np.int64)
    ends = np.array([r[2] for r in regions], np.int64)
    minus = np.array([(r[3] == '-' or r[3] == -1) for r in regions], bool)
    cnt = engine.count_contexts(genome, chroms, starts, ends, minus).cpu().numpy().astype(np.float64)
    keys = sorted(set(trans_idx))
    ctx = list(mk_context_sequences(1, 1).keys())
    pos = {c: i for i, c in enumerate(ctx)}
    cols = np.array([pos[k.split('>')[0]] for k in keys])
    idx = ["{}:{}-{}".format(c, s, e) for c, s, e in zip(chroms, starts, ends)]
    return pd.DataFrame(cnt[:, cols], index=idx, columns=keys)


def precount_region_contexts_parallel(f_nonc_bed, f_fasta, n_procs, window, sub_elts=True, n_up=1, n_down=1):
    """sequence_tools.py:481-525: context counts of every block of a bed12 (sub_elts) or of every bed row, rows with a
    repeated index removed.  n_procs is accepted for compatibility (one launch does all regions)."""
    from ..data_tools import mutation_tools
    trans_idx = mk_trans_idx(n_up=1, n_down=1, collapse=False)
    df = pd.read_csv(f_nonc_bed, sep='\t', header=None, names=None, low_memory=False, dtype={0: str})
    if sub_elts:
        df6 = mutation_tools._bed12_to_bed6(df)
        chrom = df6.CHROM.astype(str)
        if 'chr' in str(chrom.iloc[0]):
            chrom = chrom.map(lambda x: x.lstrip('chr'))
        regions = list(zip(chrom, df6.START, df6.END, df6.STRAND))
    else:
        chrom = df[0].astype(str)
        if 'chr' in str(chrom.iloc[0]):
            chrom = chrom.map(lambda x: x.lstrip('chr'))
        regions = list(zip(chrom, df[1], df[2], df[5]))
    results = nonc_elt_context_count(regions, trans_idx, f_fasta, n_up=n_up, n_down=n_down)
    return results.loc[~results.index.duplicated()]


def count_contexts_in_bed(f_fasta, df_bed, n_up=1, n_down=1, N_proc=1, N_chunk=10, collapse=False):
    """sequence_tools.py:101-137: context counts of every row of a bed-like frame (columns 0-2 = chrom, start, end;
    'chr' is prepended to the chromosome label).  One launch; N_proc / N_chunk accepted for compatibility."""
    chrom_lst = ['chr{}'.format(val) for val in df_bed.iloc[:, 0].values]
    return count_contexts_by_regions(f_fasta, chrom_lst, df_bed.iloc[:, 1].values, df_bed.iloc[:, 2].values, n_up=n_up,
                                     n_down=n_down, collapse=collapse)


def initialize_nonc_data(f_nonc_data, f_genome_counts, window, n_up=1, n_down=1):
    """sequence_tools.py:451-478: start an element-data container from the genome-wide window counts: the sorted
    substitution index and window_{w}/full_window_si_{index,values}."""
    from ..io import mapfile
    key = 'window_{}'.format(window)
    with mapfile.batch(f_nonc_data):
        if not mapfile.has_key(f_nonc_data, 'substitution_idx'):
            mapfile.write_array(f_nonc_data, 'substitution_idx', np.array(mk_trans_idx(n_up=n_up, n_down=n_down, collapse=False)))
        if not (mapfile.has_key(f_nonc_data, key + '/full_window_si_index') and
                mapfile.has_key(f_nonc_data, key + '/full_window_si_values')):
            idx = mapfile.read_array(f_genome_counts, 'idx')
            genome_df = mapfile.read_frame(f_genome_counts, 'all_window_genome_counts')
            assert int(str(genome_df.index[0]).split('-')[-1]) == window      # the counts must be on this window size (:476)
            mapfile.write_array(f_nonc_data, key + '/full_window_si_values', genome_df.values.astype(np.int64))
            mapfile.write_array(f_nonc_data, key + '/full_window_si_index', idx)


def preprocess_nonc(f_nonc_bed, f_nonc_data, f_pretrained, L_contexts, save_key, window):
    """sequence_tools.py:596-641: per-element L counts (sum of the block rows of L_contexts) and geometry under
    window_{w}/{save_key}.  The reference stores one h5 group per element plus its overlapped-window counts; the
    overlaps and region counts are recomputed on the GPU at model time here (dig_ideal_overlaps_host +
    dig_accumulate_elements), so the container holds flat arrays: names, chrom, strand, blk_ptr, blk_start, blk_end, L."""
    from ..data_tools import mutation_tools
    from ..io import mapfile
    df_elts = mutation_tools.bed12_boundaries(f_nonc_bed)
    E = len(df_elts)
    blk_ptr = np.concatenate([[0], np.cumsum([len(b) for b in df_elts.BLOCK_STARTS])]).astype(np.int64)
    blk_start = np.array([s for b in df_elts.BLOCK_STARTS for s in b], np.int64)
    blk_end = np.array([e for b in df_elts.BLOCK_ENDS for e in b], np.int64)
    chrom = df_elts.CHROM.values.astype(np.int32)
    owner = np.repeat(np.arange(E), np.diff(blk_ptr))
    keys = ['chr{}:{}-{}'.format(c, s, e) for c, s, e in zip(chrom[owner], blk_start, blk_end)]
    L = np.zeros((E, 192))
    np.add.at(L, owner, L_contexts.loc[keys].values)
    base = 'window_{}/{}/'.format(window, save_key)
    with mapfile.batch(f_nonc_data):                 # one rewrite of the container instead of one per key
        mapfile.write_array(f_nonc_data, base + 'names', df_elts.ELT.values.astype(str))
        mapfile.write_array(f_nonc_data, base + 'chrom', chrom)
        mapfile.write_array(f_nonc_data, base + 'strand', df_elts.STRAND.astype(str).values)
        mapfile.write_array(f_nonc_data, base + 'blk_ptr', blk_ptr)
        mapfile.write_array(f_nonc_data, base + 'blk_start', blk_start)
        mapfile.write_array(f_nonc_data, base + 'blk_end', blk_end)
        mapfile.write_array(f_nonc_data, base + 'L', np.rint(L).astype(np.int32))


def preprocess_sites(f_sites, f_nonc_data, f_pretrained, save_key, window):
    """sequence_tools.py:643-700: element data for a SITES file (mutation-file layout, element name in the SAMPLE
    column, one row per (position, substitution) site, optional STRAND column).  Per element: L[key] = number of its
    sites with substitution key "XYZ>XaZ" (reverse-complemented for '-' strand elements, rows without context
    skipped); its overlapped windows come from all site intervals.  Stored as the flat arrays preprocess_nonc writes
    (names in sorted order, as the reference's groupby gives them); f_pretrained is accepted for compatibility."""
    from ..data_tools import mutation_tools
    from ..io import mapfile
    keys = sorted(mk_trans_idx(n_up=1, n_down=1, collapse=False))
    pos = {k: i for i, k in enumerate(keys)}
    df = mutation_tools.read_mutation_file(f_sites)
    df = df.drop(columns=['GENE', 'ANNOT', 'REF', 'ALT']).rename(columns={'SAMPLE': 'GENE'})
    df.loc[df.CONTEXT.isna(), 'CONTEXT'] = 'nan'
    if 'STRAND' not in df.columns:
        df['STRAND'] = '.'
    names, chroms, strands, blk_ptr, bs, be, Ls = [], [], [], [0], [], [], []
    for name, group in df.groupby('GENE'):
        strand = list(group['STRAND'])[0]
        minus = strand == "-1" or strand == "-"
        L = np.zeros(192, np.int32)
        for m, c in zip(group['MUT_TYPE'], group['CONTEXT']):
            key = (reverse_complement(c) + '>' + reverse_complement(c[0] + m[2] + c[2])) if minus else (c + '>' + c[0] + m[2] + c[2])
            if 'nan' in key:
                continue
            L[pos[key]] += 1
        names.append(str(name))
        chroms.append(int(list(group['CHROM'])[0]))
        strands.append(str(strand))
        bs.extend(int(x) for x in group['START'])
        be.extend(int(x) for x in group['END'])
        blk_ptr.append(len(bs))
        Ls.append(L)
    base = 'window_{}/{}/'.format(window, save_key)
    with mapfile.batch(f_nonc_data):                 # one rewrite of the container instead of one per key
        mapfile.write_array(f_nonc_data, base + 'names', np.array(names))
        mapfile.write_array(f_nonc_data, base + 'chrom', np.array(chroms, np.int32))
        mapfile.write_array(f_nonc_data, base + 'strand', np.array(strands))
        mapfile.write_array(f_nonc_data, base + 'blk_ptr', np.array(blk_ptr, np.int64))
        mapfile.write_array(f_nonc_data, base + 'blk_start', np.array(bs, np.int64))
        mapfile.write_array(f_nonc_data, base + 'blk_end', np.array(be, np.int64))
        mapfile.write_array(f_nonc_data, base + 'L', np.stack(Ls) if Ls else np.zeros((0, 192), np.int32))
