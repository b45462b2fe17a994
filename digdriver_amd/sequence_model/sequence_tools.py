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
    if (n_up, n_down, bool(collapse)) != (1, 1, False):
        raise NotImplementedError("the GPU context counter handles trinucleotides (n_up = n_down = 1, collapse=False), the "
                                  "only configuration the driver path uses (onthefly_tools.py:70-71,120)")


def count_contexts_by_regions(f_fasta, chrom_lst, start_lst, end_lst, n_up=2, n_down=2, collapse=False):
    """sequence_tools.py:82-99: frame [regions x 64 contexts] (columns in mk_context_sequences order, index
    "{CHROM}:{START}-{END}") -- one dig_count_contexts launch for all regions.  `f_fasta`: path or PackedGenome."""
    from .. import engine
    _require_trinuc(n_up, n_down, collapse)
    genome = f_fasta if hasattr(f_fasta, "words") else load_genome(f_fasta)
    cnt = engine.count_contexts(genome, list(chrom_lst), np.asarray(start_lst, np.int64), np.asarray(end_lst, np.int64))
    idx = ["{}:{}-{}".format(c, s, e) for c, s, e in zip(chrom_lst, start_lst, end_lst)]
    return pd.DataFrame(cnt.cpu().numpy().astype(np.int64), index=idx, columns=list(mk_context_sequences(1, 1).keys()))


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
