"""CPU oracle for the DIGDriver burden-test hot path (TEST INFRASTRUCTURE ONLY).

This module is a plain numpy/scipy restatement of the reference's algorithm for
the hot path named in BASELINE.json.  It exists to CHECK the HIP path.  Only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import it; the product package ``digdriver_amd`` never does.

Pinning: every function below is checked against golden vectors produced by
running the real reference in the build container (``tests/golden/make_golden.py``,
scipy 1.15.3) -- see ``tests/test_oracle_golden.py``.  The third-party arithmetic
the reference calls (scipy.special.betainc, scipy.stats.nbinom.pmf,
scipy.stats.chi2.sf) is called here the same way, so the oracle inherits the
reference's numerics exactly for the NB tests.  GP calibration has no pinned
oracle (gpytorch is absent and unpinned): "parity unpinned" for that row.

All ``file:line`` citations are relative to the reference tree.
"""
import itertools as _it

import numpy as np
import scipy.special
import scipy.stats

# --------------------------------------------------------------------------
# substitution index (sequence_tools.py:232-289) and '-' strand permutation
# (sequence_tools.py:610-614,633-634)
# --------------------------------------------------------------------------
_DNA = "ACGT"
_COMP = {"A": "T", "C": "G", "G": "C", "T": "A", "N": "N"}


def reverse_complement(seq):
    """sequence_tools.py:18-19"""
    return "".join(_COMP[c] for c in reversed(seq))


def context64():
    """The 64 trinucleotide contexts in the order of mk_context_sequences
    (sequence_tools.py:30-40): itertools.product over ACGT^3 == sorted order."""
    return ["".join(t) for t in _it.product(_DNA, _DNA, _DNA)]


def subst_idx192():
    """Sorted 'XYZ>XaZ' strings (mk_trans_idx, sequence_tools.py:282-289)."""
    out = []
    for ctx in context64():
        for alt in _DNA:
            if alt != ctx[1]:
                out.append(ctx + ">" + ctx[0] + alt + ctx[2])
    return sorted(out)


def model_rows192():
    """(MUT_TYPE, CONTEXT) rows in the order of mk_mutation_context(collapse=False)
    (sequence_tools.py:232-262): A-, C-, G-, T-centred blocks; within a block the
    three mutation types are the outer loop."""
    rows = []
    muts = {"A": ["A>T", "A>C", "A>G"], "C": ["C>A", "C>G", "C>T"],
            "G": ["G>T", "G>C", "G>A"], "T": ["T>A", "T>G", "T>C"]}
    for ref in "ACGT":
        keys = ["".join(t) for t in _it.product(_DNA, ref, _DNA)]
        for m in muts[ref]:
            for k in keys:
                rows.append((m, k))
    return rows


def model_rows_to_sorted_perm():
    """Index array `perm` such that d_pr_sorted = FREQ[perm]: the reference builds
    the string 'CONTEXT>C0 ALT C2' for each model row and sorts the index
    (genic_driver_tools.py:321-325)."""
    rows = model_rows192()
    names = [c + ">" + c[0] + m[2] + c[2] for m, c in rows]
    order = sorted(range(192), key=lambda i: names[i])
    assert [names[i] for i in order] == subst_idx192()
    return np.array(order)


def minus_strand_gather192():
    """new[i] = old[g[i]] for a '-' strand element (sequence_tools.py:633-634):
    positions are re-ordered by the sort rank of their reverse-complemented name."""
    s = subst_idx192()
    revc = [reverse_complement(x.split(">")[0]) + ">" + reverse_complement(x.split(">")[1]) for x in s]
    return np.array(sorted(range(192), key=lambda i: revc[i]))


def minus_strand_gather64():
    """The same permutation on the 64 context counts: new64[c] = old64[rho[c]],
    rho = reverse-complement of the context."""
    ctx = context64()
    pos = {c: i for i, c in enumerate(ctx)}
    return np.array([pos[reverse_complement(c)] for c in ctx])


# --------------------------------------------------------------------------
# NB arithmetic (nb_model.py:237-337)
# --------------------------------------------------------------------------
def normal_params_to_gamma(mu, sigma):
    """nb_model.py:237-241"""
    alpha = mu ** 2 / sigma ** 2
    theta = sigma ** 2 / mu
    return alpha, theta


def nb_pvalue_greater_midp(k, alpha, p):
    """nb_model.py:271-278"""
    return 0.5 * scipy.stats.nbinom.pmf(k, alpha, p) + scipy.special.betainc(k + 1, alpha, 1 - p)


def nb_pvalue_greater(k, alpha, p):
    """nb_model.py:243-256 (vectorised over the scalar branches)"""
    k, alpha, p = np.broadcast_arrays(*(np.asarray(v, float) for v in (k, alpha, p)))
    with np.errstate(all="ignore"):
        pval = scipy.special.betainc(k, alpha, 1 - p)
        pmf = scipy.stats.nbinom.pmf(k, alpha, p)
    pval = np.where(pval == 0, pmf, pval)
    return np.where(k == 0, 1.0, pval)


def nb_pvalue_exact(k, alpha, p):
    """nb_model.py:298-314 (mu defaults to alpha*(1-p)/p)"""
    k, alpha, p = np.broadcast_arrays(*(np.asarray(v, float) for v in (k, alpha, p)))
    with np.errstate(all="ignore"):
        mu = alpha * (1 - p) / p
        lower = scipy.special.betainc(alpha, k + 1, p)
        upper = scipy.special.betainc(k, alpha, 1 - p)
        pmf = scipy.stats.nbinom.pmf(k, alpha, p)
    upper = np.where(upper == 0, pmf, upper)
    return np.where(k < mu, lower, upper)


def nb_pvalue_midp(k, alpha, p):
    """nb_model.py:316-337"""
    k, alpha, p = np.broadcast_arrays(*(np.asarray(v, float) for v in (k, alpha, p)))
    with np.errstate(all="ignore"):
        mu = alpha * (1 - p) / p
        pmf = scipy.stats.nbinom.pmf(k, alpha, p)
        low = np.where(k > 0, 0.5 * pmf + scipy.special.betainc(alpha, k, p), 0.5 * pmf)
        up = 0.5 * pmf + scipy.special.betainc(k + 1, alpha, 1 - p)
    return np.where(k < mu, low, up)


def fisher_combine(p1, p2):
    """transfer_tools.py:860-861,1086-1087"""
    with np.errstate(all="ignore"):
        x2 = -2 * (np.log(p1) + np.log(p2))
    return scipy.stats.chi2.sf(x2, df=4)


# --------------------------------------------------------------------------
# element statistics block (transfer_tools.py:272-302,343-344,473-482,594-615,
# 731-747,1086-1087)
# --------------------------------------------------------------------------
def element_stats(mu, sigma, pi_sum, pi_indel, obs_snv, obs_samples, obs_indel, cj, cj_indel,
                  mu_indel=None, sigma_indel=None):
    """All arrays broadcast to a common shape (e.g. [E, C]); cj/cj_indel broadcast
    along the cohort axis.  Returns the seven result columns + ALPHA/THETA."""
    mu_indel = mu if mu_indel is None else mu_indel
    sigma_indel = sigma if sigma_indel is None else sigma_indel
    with np.errstate(all="ignore"):
        alpha, theta = normal_params_to_gamma(mu, sigma)            # load_pretrained_model :17-19
        alpha_i, theta_i = normal_params_to_gamma(mu_indel, sigma_indel)  # :46-48
        theta = theta * cj                                          # transfer_element_model_with_indels :300
        exp_snv = alpha * theta * pi_sum                            # :343-344
        p = 1 / (theta * pi_sum + 1)
        pval_snv = nb_pvalue_greater_midp(obs_snv, alpha, p)        # :473-482
        pval_samp = nb_pvalue_greater_midp(obs_samples, alpha, p)   # :594-615
        theta_i = theta_i * cj_indel                                # :737
        exp_indel = alpha_i * theta_i * pi_indel                    # :738
        pval_indel = nb_pvalue_greater_midp(obs_indel, alpha_i, 1 / (theta_i * pi_indel + 1))  # :741-745
        pval_mut = fisher_combine(pval_snv, pval_indel)             # :1086-1087
    return dict(ALPHA=alpha, THETA=theta, EXP_SNV=exp_snv, PVAL_SNV_BURDEN=pval_snv,
                PVAL_SAMPLE_BURDEN=pval_samp, THETA_INDEL=theta_i, EXP_INDEL=exp_indel,
                PVAL_INDEL_BURDEN=pval_indel, PVAL_MUT_BURDEN=pval_mut)


GENE_CLASSES = ["SYN", "MIS", "NONS", "SPL", "TRUNC", "NONSYN"]


def gene_stats(mu, sigma, pi, obs, n_samp, cj, pi_indel=None, obs_indel=None, t_indel=None,
               mu_indel=None, sigma_indel=None):
    """gene twins: transfer_tools.py:331-340 (EXP_*), :425-454 (PVAL_*_BURDEN),
    :554-583 (PVAL_*_BURDEN_SAMPLE), :709-727 (indel), :860-861 (Fisher on TRUNC+INDEL).
    `pi`, `obs`, `n_samp` are dicts keyed by GENE_CLASSES."""
    out = {}
    with np.errstate(all="ignore"):
        alpha, theta = normal_params_to_gamma(mu, sigma)
        theta = theta * cj
        out["ALPHA"], out["THETA"] = alpha, theta
        for c in GENE_CLASSES:
            out["EXP_" + c] = alpha * theta * pi[c]
            p = 1 / (theta * pi[c] + 1)
            out["PVAL_%s_BURDEN" % c] = nb_pvalue_greater_midp(obs[c], alpha, p)
            out["PVAL_%s_BURDEN_SAMPLE" % c] = nb_pvalue_greater_midp(n_samp[c], alpha, p)
        if pi_indel is not None:
            mu_indel = mu if mu_indel is None else mu_indel
            sigma_indel = sigma if sigma_indel is None else sigma_indel
            alpha_i, theta_i = normal_params_to_gamma(mu_indel, sigma_indel)
            theta_i = theta_i * t_indel
            out["THETA_INDEL"] = theta_i
            out["EXP_INDEL"] = alpha_i * theta_i * pi_indel
            out["PVAL_INDEL_BURDEN"] = nb_pvalue_greater_midp(obs_indel, alpha_i, 1 / (theta_i * pi_indel + 1))
            out["PVAL_MUT_BURDEN"] = fisher_combine(out["PVAL_TRUNC_BURDEN"], out["PVAL_INDEL_BURDEN"])
    return out


# --------------------------------------------------------------------------
# bin overlaps (genic_driver_tools.py:275-283)
# --------------------------------------------------------------------------
def ideal_overlap_starts(starts, ends, window):
    """Sorted, de-duplicated bin START coordinates touched by the blocks.
    low=floor(s/w)*w, high=ceil(e/w)*w, bins [low, high) in steps of w; a block that
    ends exactly on a bin edge does not add the next bin; a zero-length block on an
    edge adds nothing."""
    out = set()
    for s, e in zip(starts, ends):
        low = int(np.floor(s / window)) * window
        high = int(np.ceil(e / window)) * window
        for b in range(low, high, window):
            out.add(b)
    return sorted(out)


def build_overlap_csr(elt_chrom, block_starts, block_ends, bin_index, window):
    """CSR (ov_ptr[E+1], ov_idx[nnz]) of bin row numbers per element, ascending.
    `bin_index` maps (chrom, start) -> row of the bin tables."""
    ptr = [0]
    idx = []
    for c, bs, be in zip(elt_chrom, block_starts, block_ends):
        bs = [s for s in bs if s >= 0]
        be = [e for e in be if e >= 0]
        for b in ideal_overlap_starts(bs, be, window):
            idx.append(bin_index[(int(c), int(b))])
        ptr.append(len(idx))
    return np.array(ptr, np.int64), np.array(idx, np.int32)


# --------------------------------------------------------------------------
# per-element accumulation (genic_driver_tools.py:258-272 region part,
# :361-381 sequence part; genic: :110-158; tiled: :645-667)
# --------------------------------------------------------------------------
def accumulate_elements(bin_mu, bin_std, bin_y, bin_flag, bin_ctx, ov_ptr, ov_idx, L, strand_minus, d_pr,
                        gene_length=None):
    """
    bin_mu, bin_std : float64 [N, C]     Y_PRED, STD of region_params per cohort
    bin_y           : int     [N, C]     Y_TRUE
    bin_flag        : bool/u8 [N, C]     FLAG (the result FLAG is the logical OR over the bins: the
                      reference adds numpy bools, `False + np.True_ + np.True_ == True`)
    bin_ctx         : int     [N, 64]    full_window_si_values
    ov_ptr, ov_idx  : CSR of overlapped bin rows per element
    L               : [E, n_class, 192]  L_counts (n_class = 1 elements, 4 genes)
    strand_minus    : bool [E]
    d_pr            : float64 [C, 192]   FREQ re-indexed by sorted substitution string
    gene_length     : optional int [E]; when given P_INDEL = gene_length / R_SIZE
                      (genic_driver_tools.py:158-159) instead of ELT_SIZE / R_SIZE (:380-381)
    returns dict of MU, SIGMA [E,C] f64; R_OBS, FLAG [E,C] int64; P [E,n_class,C] f64;
            R_SIZE, ELT_SIZE [E] int64; P_INDEL [E] f64
    """
    E = len(ov_ptr) - 1
    C = d_pr.shape[0]
    n_class = L.shape[1]
    g192 = minus_strand_gather192()
    MU = np.zeros((E, C)); VAR = np.zeros((E, C))
    ROBS = np.zeros((E, C), np.int64); FLAG = np.zeros((E, C), np.int64)
    P = np.zeros((E, n_class, C))
    RSIZE = np.zeros(E, np.int64); ESIZE = np.zeros(E, np.int64); PIND = np.zeros(E)
    with np.errstate(all="ignore"):
        for e in range(E):
            bins = ov_idx[ov_ptr[e]:ov_ptr[e + 1]]
            for b in bins:                          # get_region_params_direct :264-268 (sequential sums)
                MU[e] += bin_mu[b]
                VAR[e] += bin_std[b] ** 2
                ROBS[e] += bin_y[b]
                FLAG[e] |= bin_flag[b].astype(np.int64)   # False + np.bool_ is a logical OR (:268)
            rc192 = np.repeat(bin_ctx[bins].sum(axis=0), 3)        # sequence_tools.py:630-631
            if strand_minus[e]:
                rc192 = rc192[g192]                                 # :633-634
            for c in range(C):
                prob_sum = rc192 * d_pr[c]                          # :361
                t_pi = d_pr[c] / prob_sum.sum()                     # :364
                for q in range(n_class):
                    P[e, q, c] = (t_pi * L[e, q]).sum()             # :366
            RSIZE[e] = int(rc192.sum() / 3)                         # :375
            ESIZE[e] = int(np.sum(L[e, 0] if n_class == 1 else L[e]) / 3)  # :380
            num = ESIZE[e] if gene_length is None else gene_length[e]
            PIND[e] = np.float64(num) / np.float64(RSIZE[e])        # :381 / :159
    return dict(MU=MU, SIGMA=np.sqrt(VAR), R_OBS=ROBS, FLAG=FLAG, P=P, R_SIZE=RSIZE, ELT_SIZE=ESIZE, P_INDEL=PIND)


def accumulate_elements_fast(bin_mu, bin_std, bin_y, bin_flag, bin_ctx, ov_ptr, ov_idx, L, strand_minus, d_pr):
    """Vectorised form of accumulate_elements for n_class=1 (CPU-baseline timing at
    bench sizes; same arithmetic up to summation order).  Every element needs >=1 bin."""
    E = len(ov_ptr) - 1
    starts = ov_ptr[:-1]
    MU = np.add.reduceat(bin_mu[ov_idx], starts, axis=0)
    VAR = np.add.reduceat(bin_std[ov_idx] ** 2, starts, axis=0)
    ROBS = np.add.reduceat(bin_y[ov_idx].astype(np.int64), starts, axis=0)
    FLAG = (np.add.reduceat(bin_flag[ov_idx].astype(np.int64), starts, axis=0) > 0).astype(np.int64)
    rc64 = np.add.reduceat(bin_ctx[ov_idx].astype(np.int64), starts, axis=0)
    rho = minus_strand_gather64()
    rc64 = np.where(strand_minus[:, None], rc64[:, rho], rc64)
    d64 = d_pr.reshape(d_pr.shape[0], 64, 3).sum(axis=2)           # [C,64]
    denom = rc64.astype(np.float64) @ d64.T                         # [E,C]
    numer = L[:, 0, :].astype(np.float64) @ d_pr.T                  # [E,C]
    RSIZE = rc64.sum(axis=1)
    ESIZE = (L[:, 0, :].sum(axis=1) / 3).astype(np.int64)
    return dict(MU=MU, SIGMA=np.sqrt(VAR), R_OBS=ROBS, FLAG=FLAG, P=(numer / denom)[:, None, :],
                R_SIZE=RSIZE, ELT_SIZE=ESIZE, P_INDEL=ESIZE / RSIZE)


# --------------------------------------------------------------------------
# per-bin track gather (mut_dataset.py:76-81)
# --------------------------------------------------------------------------
def gather_bins(x_data, bin_rows, tracks):
    """x_data[idx, :, tracks] -> float32 (B, L, T_sel)"""
    return np.asarray(x_data)[np.asarray(bin_rows)][:, :, np.asarray(tracks)].astype(np.float32)


# --------------------------------------------------------------------------
# per-base tiled test (nb_model.py:126-186, arithmetic only)
# --------------------------------------------------------------------------
def tiled_nb_test(pt, k, mu, sigma):
    """pt [n_bins, n_tiles] tile probabilities (already summed over `binsize` positions and
    normalised over the bin), k [n_bins, n_tiles] counts, mu/sigma [n_bins].
    Returns (pval, exp) per tile: nb_model.py:141-178."""
    with np.errstate(all="ignore"):
        alpha, theta = normal_params_to_gamma(mu, sigma)
        p = 1 / (pt * theta[:, None] + 1)
        pval = nb_pvalue_exact(k, alpha[:, None], p)
        exp = pt * mu[:, None]
    return pval, exp


# --------------------------------------------------------------------------
# sequence model training (sequence_tools.py:321-373)
# --------------------------------------------------------------------------
def train_sequence_model(mut_type, context, genome_ctx, genome_counts):
    """Counts of (MUT_TYPE, CONTEXT) over the (already whitelisted, de-duplicated) mutations
    divided by the genome count of the context.  Returns (rows192, COUNT[192], FREQ[192],
    ctx64_sorted, FREQ64[64])."""
    rows = model_rows192()
    pos = {r: i for i, r in enumerate(rows)}
    count = np.zeros(192, np.int64)
    for m, c in zip(mut_type, context):
        j = pos.get((m, c))
        if j is not None:
            count[j] += 1
    g = dict(zip(genome_ctx, genome_counts))
    freq = np.array([count[i] / g[rows[i][1]] for i in range(192)])
    ctx_sorted = sorted(set(c for _, c in rows))
    freq64 = np.array([sum(freq[i] for i in range(192) if rows[i][1] == c) for c in ctx_sorted])
    return rows, count, freq, ctx_sorted, freq64


# --------------------------------------------------------------------------
# cohort scale factors (transfer_tools.py:129-159 genome mode)
# --------------------------------------------------------------------------
def scale_factor_genome(bin_y_pred, bin_flag, n_snv_obs, n_indel_obs):
    """cj = N_SNV_OBS / sum(Y_PRED[~FLAG]); cj_indel = N_IND_OBS / same (transfer_tools.py:148-156)"""
    n_exp = bin_y_pred[~bin_flag.astype(bool)].sum()
    return n_snv_obs / n_exp, n_indel_obs / n_exp


# --------------------------------------------------------------------------
# trinucleotide context counting from sequence (sequence_tools.py:21-29,42-55,65-94,527-566)
# --------------------------------------------------------------------------
def count_contexts_region(chrom_seq, start, end):
    """64 trinucleotide counts of the centre positions of [start, end) on one chromosome string.
    fetch_sequence (:21-29): the region is widened by one base on either side, START == 0 becomes 1, the fetch is
    truncated at the chromosome end and upper-cased; count_sequence_context (:65-80) then counts every window of
    three that holds no 'N' (seq_to_context :48-49).  Any other non-ACGT letter would raise KeyError in the
    reference; it is skipped here like 'N'."""
    if start == 0:
        start = 1
    seq = chrom_seq[start - 1:end + 1].upper()
    code = {"A": 0, "C": 1, "G": 2, "T": 3}
    out = np.zeros(64, np.int64)
    for i in range(1, len(seq) - 1):
        tri = seq[i - 1:i + 2]
        if all(ch in code for ch in tri):
            out[16 * code[tri[0]] + 4 * code[tri[1]] + code[tri[2]]] += 1
    return out


def count_contexts_regions(genome, chroms, starts, ends, minus=None):
    """[R, 64] counts (columns in context64() order).  minus[r] = True counts the reverse-complemented sequence
    (nonc_elt_context_count, :551-553), i.e. out[ctx] = plus[revcomp(ctx)]."""
    rho = minus_strand_gather64()
    out = np.zeros((len(chroms), 64), np.int64)
    for r, (c, s, e) in enumerate(zip(chroms, starts, ends)):
        cnt = count_contexts_region(genome[c], int(s), int(e))
        out[r] = cnt[rho] if (minus is not None and minus[r]) else cnt
    return out


def expand_contexts_192(counts64):
    """64 context counts -> the 192 columns of sorted "XYZ>XaZ" keys (nonc_elt_context_count :558-564): every
    substitution column takes the count of its context."""
    keys = subst_idx192()
    ctx = context64()
    pos = {c: i for i, c in enumerate(ctx)}
    cols = np.array([pos[k.split(">")[0]] for k in keys])
    return np.asarray(counts64)[..., cols], keys


# --------------------------------------------------------------------------
# per-base route, front half (sequence_tools.py:292-317, nb_model.py:126-186; trinucleotide contexts, n_up = n_down = 1)
# --------------------------------------------------------------------------
def base_probabilities_by_region(chrom_seq, s_prob, start, end, n_up=1):
    """sequence_tools.py:292-317 with n_up = n_down (1: trinucleotide contexts, the live pipeline; 2: the functions' default,
    penta-nucleotide), normed=True.  chrom_seq: the chromosome string; s_prob: 4^(2 n_up + 1) probabilities in
    itertools.product('ACGT', repeat=2 n_up + 1) order.  fetch_sequence (:21-29): START == 0 becomes n_up, the fetch is
    widened by n_up bases on either side and truncated at the chromosome end; a position whose window holds a non-ACGT
    letter gets 0.  Returns (probs, positions): probs normalised over the region (np.sum, as the reference)."""
    if start == 0:
        start = n_up
    assert start - n_up >= 0, "pysam refuses a negative fetch start"
    seq = chrom_seq[start - n_up:end + n_up].upper()
    code = {"A": 0, "C": 1, "G": 2, "T": 3}
    probs, poss = [], []
    for i in range(n_up, len(seq) - n_up):
        poss.append(start - n_up + i)
        win = seq[i - n_up:i + n_up + 1]
        if all(ch in code for ch in win):
            k = 0
            for ch in win:
                k = 4 * k + code[ch]
            probs.append(s_prob[k])
        else:
            probs.append(0)
    probs = np.array(probs, dtype=float)
    with np.errstate(all="ignore"):
        probs = probs / np.sum(probs)
    return probs, np.array(poss)


def apply_nb_to_region(chrom_seq, s_prob64, start, end, mu, sigma, mut_starts, binsize, n_up=1):
    """nb_model.py:126-186: tiles of `binsize` consecutive positions (the last one may be shorter); pt = np.sum of the
    tile's normalised probabilities, k = number of mutation rows whose START is one of the tile's positions
    (value_counts of START, :135-136,160-163), p = 1 / (pt theta + 1), nb_pvalue_exact, exp = pt mu, pos = mean position.
    mut_starts: START of every mutation row of this chromosome.  Returns (pvals, pos, obs, exps, pts)."""
    probs, pos_lst = base_probabilities_by_region(chrom_seq, s_prob64, start, end, n_up=n_up)
    mut_starts = np.asarray(mut_starts, np.int64)
    alpha, theta = normal_params_to_gamma(mu, sigma)
    pv, ps, ob, ex, pts = [], [], [], [], []
    for i in range(0, len(pos_lst), binsize):
        with np.errstate(all="ignore"):
            pt = np.sum(probs[i:i + binsize])
            lo, hi = pos_lst[i], pos_lst[min(i + binsize, len(pos_lst)) - 1]
            k = int(((mut_starts >= lo) & (mut_starts <= hi)).sum())
            p = 1 / (pt * theta + 1)
            pv.append(float(nb_pvalue_exact(np.array([float(k)]), np.array([alpha]), np.array([p]))[0]))
        ps.append(float(np.mean(pos_lst[i:i + binsize])))
        ob.append(k)
        ex.append(pt * mu)
        pts.append(pt)
    return np.array(pv), np.array(ps), np.array(ob), np.array(ex), np.array(pts)


# --------------------------------------------------------------------------
# mutation x element-block interval join and integer tabulation
# (mutation_tools.py:155-230; the join itself is `bedtools intersect -wa -wb`, a third-party binary:
#  bedtools 2.30.0 / pybedtools 0.8.1 in conda-recipe/meta.yaml:47,65, absent from this image)
# --------------------------------------------------------------------------
def bed12_blocks(bed_rows):
    """`bedtools bed12tobed6` (mutation_tools.py:196-197): one (chrom, start, end, name, strand) row per block of every
    bed12 row, in file order; blockSizes / blockStarts are comma lists with an optional trailing comma."""
    out = []
    for row in bed_rows:
        chrom, start, name, strand = row[0], int(row[1]), row[3], row[5]
        sizes = [int(x) for x in str(row[10]).split(",") if x != ""]
        rel = [int(x) for x in str(row[11]).split(",") if x != ""]
        for s, z in zip(rel, sizes):
            out.append((chrom, start + s, start + s + z, name, strand))
    return out


def _chrom_label(c):
    """bedtools compares chromosome labels as text: '1' and 'chr1' are different chromosomes (the reference hands both
    files to bedtools as they are, mutation_tools.py:193-200)."""
    return str(c)


def interval_join_pairs(m_chrom, m_start, m_end, b_chrom, b_start, b_end):
    """`bedtools intersect -wa -wb` of mutations (A) with element blocks (B), restated from bedtools' published
    definition of an overlap: same chromosome and at least one shared base of the half-open intervals,
        a.start < b.end  and  b.start < a.end.
    One output pair per (mutation, overlapped block), mutation-major in the order of the mutation file
    (mutation_tools.py:200); blocks may overlap or nest each other and every hit is reported.
    A zero-length feature (start == end) is given the one base [start, start + 1): the reference's annotated mutation
    files hold 1-bp SNV rows and longer indel rows only, and bedtools' own treatment of zero-length records cannot be
    run here -- parity unpinned for that corner, stated in tests/golden/make_golden.py::gen_tabulate as well.
    Plain loop over mutations; returns two int64 arrays (mutation row, block row), blocks ascending per mutation."""
    b_lab = np.array([_chrom_label(c) for c in b_chrom], dtype=object)
    b_start = np.asarray(b_start, np.int64)
    b_end = np.asarray(b_end, np.int64)
    b_end = np.where(b_end == b_start, b_start + 1, b_end)
    by_chrom = {}
    for j, c in enumerate(b_lab):
        by_chrom.setdefault(c, []).append(j)
    by_chrom = {c: np.array(v, np.int64) for c, v in by_chrom.items()}
    mi, bi = [], []
    for i in range(len(m_chrom)):
        rows = by_chrom.get(_chrom_label(m_chrom[i]))
        if rows is None:
            continue
        s, e = int(m_start[i]), int(m_end[i])
        if e == s:
            e = s + 1
        hit = rows[(s < b_end[rows]) & (b_start[rows] < e)]
        mi.extend([i] * len(hit))
        bi.extend(hit.tolist())
    return np.array(mi, np.int64), np.array(bi, np.int64)


def tabulate_elements(mut_rows, block_rows, drop_duplicates=False, max_muts_per_sample=1e9,
                      max_muts_per_elt_per_sample=3e9):
    """tabulate_muts_per_sample_per_element + tabulate_mutations_in_element (mutation_tools.py:191-230, 155-189) on
    in-memory rows, with dictionaries and loops only.
      mut_rows:   mutation-file rows, columns 0..9 = CHROM, START, END, REF, ALT, SAMPLE, GENE, ANNOT, MUT_TYPE, CONTEXT
      block_rows: (chrom, start, end, name, ...) bed6 rows (bed12_blocks for a bed12 file)
    Steps: join (:200); duplicates of (chrom, start, end, ref, alt, sample, element) dropped, first kept (:207-208);
    ANNOT != 'INDEL' counts as SNV (:211-213); counts per (element, sample) (:219-227); samples whose counts summed over
    all elements exceed max_muts_per_sample are removed (:163-166); the per-(element, sample) counts are capped
    (:169-170); per element OBS_SAMPLES = number of remaining (element, sample) rows, OBS_SNV / OBS_INDEL = sums
    (:172-174).  Returns (per_pair, per_element, blacklist):
      per_pair    {(element, sample): [OBS_SNV, OBS_INDEL]}   before blacklist and cap  (= the frame of :191-230)
      per_element {element: (OBS_SAMPLES, OBS_SNV, OBS_INDEL)}  only elements with a remaining row
      blacklist   sorted sample labels."""
    mi, bi = interval_join_pairs([r[0] for r in mut_rows], [r[1] for r in mut_rows], [r[2] for r in mut_rows],
                                 [b[0] for b in block_rows], [b[1] for b in block_rows], [b[2] for b in block_rows])
    seen = set()
    per_pair = {}
    for i, j in zip(mi.tolist(), bi.tolist()):
        m, elt = mut_rows[i], block_rows[j][3]
        if drop_duplicates:
            key = (_chrom_label(m[0]), int(m[1]), int(m[2]), str(m[3]), str(m[4]), str(m[5]), elt)
            if key in seen:
                continue
            seen.add(key)
        cell = per_pair.setdefault((elt, str(m[5])), [0, 0])
        cell[1 if m[7] == "INDEL" else 0] += 1
    load = {}
    for (elt, sample), (snv, ind) in per_pair.items():
        load[sample] = load.get(sample, 0) + snv + ind
    blacklist = sorted(s for s, n in load.items() if n > max_muts_per_sample)
    per_element = {}
    for (elt, sample), (snv, ind) in per_pair.items():
        if sample in blacklist:
            continue
        acc = per_element.setdefault(elt, [0, 0, 0])
        acc[0] += 1
        acc[1] += min(snv, max_muts_per_elt_per_sample)
        acc[2] += min(ind, max_muts_per_elt_per_sample)
    return per_pair, {k: tuple(int(x) for x in v) for k, v in per_element.items()}, blacklist
