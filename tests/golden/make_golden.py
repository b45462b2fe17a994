#!/usr/bin/env python
"""Generate golden input/output vectors by running the REAL reference here.

Run in the build container only (it needs /root/reference):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

What this does
--------------
* Stubs the reference's I/O-only dependencies that are absent from this image
  (pysam, h5py, pybedtools, statsmodels, seaborn, bbi, tables, gpytorch,
  tensorboardX) in ``sys.modules``.  ``h5py.File`` is replaced by a small
  in-memory, dict-backed fake and ``pandas.read_hdf`` by a dict lookup so that
  the reference's *own* loops (``nonc_model``, ``genic_model``,
  ``tiled_nonc_model``) run unmodified on synthetic inputs.
* Imports the reference modules from /root/reference (never copied).
* Calls the reference functions on seeded synthetic inputs and stores
  inputs + outputs as small ``.npz`` / ``.json`` fixtures next to this script.

Nothing from the reference's source text is stored: the fixtures are data only.
The arithmetic behind the p-values is scipy's (scipy.special.betainc,
scipy.stats.nbinom.pmf, scipy.stats.chi2.sf); the scipy version used is
recorded in ``MANIFEST.json``.
"""
import io
import json
import os
import sys
import types

sys.dont_write_bytecode = True

import numpy as np
import pandas as pd
import scipy

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("DIG_REFERENCE", "/root/reference")


# ----------------------------------------------------------------------------
# in-memory stand-ins for the I/O modules (test harness only)
# ----------------------------------------------------------------------------
class _FakeNode:
    """dict-backed h5py group/dataset look-alike: supports node['a/b/c'],
    node[...][:] on datasets, .attrs, .keys(), `in`, context manager, close."""

    def __init__(self, tree, attrs=None):
        self._tree = tree
        self.attrs = attrs if attrs is not None else {}

    def _resolve(self, key):
        node = self._tree
        for part in [p for p in key.split("/") if p]:
            node = node[part]
        return node

    def __getitem__(self, key):
        node = self._resolve(key)
        if isinstance(node, dict):
            return _FakeNode(node, _FakeAttrs(node))
        return _FakeDataset(node)

    def __contains__(self, key):
        try:
            self._resolve(key)
            return True
        except KeyError:
            return False

    def keys(self):
        return [k for k in self._tree.keys() if k != "__attrs__"]

    def create_dataset(self, key, data=None, **kw):
        parts = [p for p in key.split("/") if p]
        node = self._tree
        for part in parts[:-1]:
            node = node.setdefault(part, {})
        if node.get(parts[-1].__str__()) is not None and not isinstance(node.get(parts[-1]), dict):
            raise ValueError("dataset exists: " + key)
        # a group that later receives attributes keeps its datasets under the same dict
        node[parts[-1]] = np.asarray(data)

    def close(self):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


class _FakeAttrs(dict):
    """attrs of a fake group: a view on node['__attrs__'] with h5py's .create()."""

    def __init__(self, node):
        super().__init__(node.get("__attrs__", {}))
        self._node = node

    def create(self, name, value):
        self._node.setdefault("__attrs__", {})[name] = value
        self[name] = value


class _FakeDataset:
    def __init__(self, arr):
        self._arr = np.asarray(arr)

    def __getitem__(self, idx):
        return self._arr[idx]

    @property
    def shape(self):
        return self._arr.shape


_H5_FILES = {}    # path -> nested dict
_HDF_FRAMES = {}  # (path, key) -> DataFrame


def _fake_h5_file(path, mode="r"):
    tree = _H5_FILES[path]
    return _FakeNode(tree, tree.get("__attrs__", {}))


def _fake_read_hdf(path, key=None, **kw):
    return _HDF_FRAMES[(str(path), key)].copy()


def install_stubs():
    for name in ["pysam", "pybedtools", "statsmodels", "statsmodels.stats",
                 "statsmodels.stats.multitest", "seaborn", "bbi", "tables",
                 "gpytorch", "tensorboardX", "pkg_resources"]:
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    h5 = types.ModuleType("h5py")
    h5.File = _fake_h5_file
    sys.modules["h5py"] = h5
    pd.read_hdf = _fake_read_hdf
    sys.path.insert(0, REF)


install_stubs()

from DIGDriver.sequence_model import nb_model as ref_nb            # noqa: E402
from DIGDriver.sequence_model import genic_driver_tools as ref_gdt  # noqa: E402
from DIGDriver.sequence_model import sequence_tools as ref_seq      # noqa: E402
from DIGDriver.driver_model import transfer_tools as ref_tt         # noqa: E402
from DIGDriver.data_tools import mutation_tools as ref_mt           # noqa: E402


def save_npz(name, **arrays):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrays)
    print("wrote", name, {k: np.asarray(v).shape for k, v in arrays.items()})


# ----------------------------------------------------------------------------
# (i) nb_pvalue_greater_midp
# ----------------------------------------------------------------------------
def synth_nb_inputs(rng, n, n_uniform):
    mu = np.exp(rng.uniform(np.log(0.05), np.log(500.0), n))
    sigma = mu * np.exp(rng.uniform(np.log(0.05), np.log(1.5), n))
    pi = np.exp(rng.uniform(np.log(1e-5), np.log(1.0), n))
    cj = 0.7
    alpha, theta = ref_nb.normal_params_to_gamma(mu, sigma)
    theta = theta * cj
    lam = alpha * theta * pi
    k = rng.poisson(np.minimum(lam * np.exp(rng.normal(0, 1, n)), 1e6)).astype(np.float64)
    k[:n_uniform] = rng.integers(0, 3000, n_uniform).astype(np.float64)
    p = 1 / (theta * pi + 1)
    return k, alpha, p, mu, sigma, pi


def gen_nb_midp():
    rng = np.random.default_rng(7)
    k, alpha, p, mu, sigma, pi = synth_nb_inputs(rng, 20000, 2000)
    # edge rows: k=0, Pi=0 (p=1), mu<0 (p>1), sigma=0, non-integer k,
    # huge k, tiny / huge alpha, p very close to 0 and 1
    ek = [0, 0, 1, 5, 0, 3, 2.5, 3000, 1, 10, 100, 0, 7, 50, 500, 20, 1, 1, 40, 2000]
    ea = [2.5, 4.0, 4.0, 4.0, 1.0, np.inf, 3.0, 4.0, 1e-3, 1e4, 1e5, 1e-8, 0.5, 40, 3, 2.0, 1e6, 300.0, 1e3, 50.0]
    ep = [0.3, 1.0, 1.0, 1.0, 1.2, 0.5, 0.5, 0.5, 0.5, 0.999, 0.9995, 0.1, 1e-6, 0.1, 1/7., 1 - 1e-12, 1 - 1e-9, 0.01, 0.9, 0.02]
    k = np.concatenate([k, np.array(ek, float)])
    alpha = np.concatenate([alpha, np.array(ea, float)])
    p = np.concatenate([p, np.array(ep, float)])
    with np.errstate(all="ignore"):
        pval = ref_nb.nb_pvalue_greater_midp(k, alpha, p)
        # survey spot values (8c)
        spot_k = np.array([0, 1, 5, 50, 500], float)
        spot_a = np.array([2.5, 2.5, 10, 40, 3])
        spot_t = np.array([.4, .4, .2, 1, 20])
        spot_pi = np.array([.01, .5, 1, .9, .3])
        spot = ref_nb.nb_pvalue_greater_midp(spot_k, spot_a, 1 / (spot_t * spot_pi + 1))
    save_npz("nb_midp_golden.npz", k=k, alpha=alpha, p=p, pval=pval,
             spot_k=spot_k, spot_alpha=spot_a, spot_theta=spot_t, spot_pi=spot_pi, spot_pval=spot)


# ----------------------------------------------------------------------------
# (ii) scalar siblings: nb_pvalue_exact / _greater / _midp
# ----------------------------------------------------------------------------
def gen_nb_exact():
    rng = np.random.default_rng(11)
    k, alpha, p, *_ = synth_nb_inputs(rng, 4000, 300)
    ek = [0, 10, 4, 3000, 0, 1, 2, 7]
    ea = [4, 4, 4, 4, 0.3, 0.3, 50, 1e3]
    ep = [.5, .5, .5, .5, .9, .01, .5, .99]
    k = np.concatenate([k, np.array(ek, float)])
    alpha = np.concatenate([alpha, np.array(ea, float)])
    p = np.concatenate([p, np.array(ep, float)])
    ex, gr, mp_ = [], [], []
    with np.errstate(all="ignore"):
        for ki, ai, pi_ in zip(k, alpha, p):
            ex.append(ref_nb.nb_pvalue_exact(ki, ai, pi_))
            gr.append(ref_nb.nb_pvalue_greater(ki, ai, pi_))
            mp_.append(ref_nb.nb_pvalue_midp(ki, ai, pi_))
    save_npz("nb_exact_golden.npz", k=k, alpha=alpha, p=p,
             pval_exact=np.array(ex, float), pval_greater=np.array(gr, float),
             pval_midp=np.array(mp_, float))


# ----------------------------------------------------------------------------
# (iii) element / gene statistics block on a frame
# ----------------------------------------------------------------------------
def gen_element_stats():
    rng = np.random.default_rng(3)
    n = 6000
    mu = np.exp(rng.uniform(np.log(0.5), np.log(400.0), n))
    sigma = mu * np.exp(rng.uniform(np.log(0.05), np.log(1.2), n))
    pi_sum = np.exp(rng.uniform(np.log(1e-4), np.log(0.5), n))
    pi_indel = np.exp(rng.uniform(np.log(1e-4), np.log(0.5), n))
    cj, cj_indel = 1.37, 0.061
    alpha, theta = ref_nb.normal_params_to_gamma(mu, sigma)
    lam = alpha * theta * cj * pi_sum
    boost = np.where(rng.uniform(size=n) < 0.02, 6.0, 1.0)
    obs_snv = rng.poisson(lam * boost * np.exp(rng.normal(0, 0.5, n)))
    obs_samples = rng.binomial(obs_snv, 0.9)
    obs_indel = rng.poisson(alpha * theta * cj_indel * pi_indel * boost)
    # a few edge rows: mu<0, sigma=0, Pi=0
    mu[:3] = [-1.0, 5.0, 3.0]
    sigma[:3] = [1.0, 0.0, 1.0]
    pi_sum[2] = 0.0
    df_pre = pd.DataFrame(dict(
        ELT_SIZE=rng.integers(100, 5000, n), FLAG=rng.integers(0, 2, n),
        R_SIZE=rng.integers(9000, 30000, n), R_OBS=rng.integers(0, 500, n), R_INDEL=rng.integers(0, 500, n),
        MU=mu, SIGMA=sigma, MU_INDEL=mu, SIGMA_INDEL=sigma, Pi_SUM=pi_sum, Pi_INDEL=pi_indel),
        index=["elt%d" % i for i in range(n)])
    with np.errstate(all="ignore"):
        a, t = ref_nb.normal_params_to_gamma(df_pre.MU, df_pre.SIGMA)
        df_pre["ALPHA"], df_pre["THETA"] = a, t
        df_pre["ALPHA_INDEL"], df_pre["THETA_INDEL"] = a.copy(), t.copy()
        # ~30% of elements are absent from the tabulation (left join -> NaN -> 0)
        present = rng.uniform(size=n) > 0.3
        df_tab = pd.DataFrame(dict(OBS_SAMPLES=obs_samples, OBS_SNV=obs_snv, OBS_INDEL=obs_indel),
                              index=df_pre.index)[present]
        df = ref_tt.transfer_element_model_with_indels(df_tab, df_pre, cj)
        df = ref_tt.element_expected_muts_nb(df)
        df = ref_tt.element_pvalue_burden_nb(df)
        df = ref_tt.element_pvalue_burden_nb_by_sample(df)
        df = ref_tt.element_pvalue_indel(df, cj_indel)
        x2 = -2 * (np.log(df.PVAL_SNV_BURDEN) + np.log(df.PVAL_INDEL_BURDEN))
        df["PVAL_MUT_BURDEN"] = scipy.stats.chi2.sf(x2, df=4)
    out_cols = ["ALPHA", "THETA", "EXP_SNV", "PVAL_SNV_BURDEN", "PVAL_SAMPLE_BURDEN", "THETA_INDEL",
                "EXP_INDEL", "PVAL_INDEL_BURDEN", "PVAL_MUT_BURDEN", "OBS_SNV", "OBS_SAMPLES", "OBS_INDEL"]
    save_npz("element_stats_golden.npz",
             mu=mu, sigma=sigma, pi_sum=pi_sum, pi_indel=pi_indel,
             cj=np.float64(cj), cj_indel=np.float64(cj_indel), present=present,
             tab_obs_snv=obs_snv, tab_obs_samples=obs_samples, tab_obs_indel=obs_indel,
             columns=np.array(list(df.columns)),
             **{"out_" + c: df[c].values.astype(float) for c in out_cols})

    # Fisher spot values straight from scipy as the reference calls it
    p1 = np.array([1e-3, 0.5, 1.0, 0.0, 1e-300, 3e-17])
    p2 = np.array([0.5, 0.5, 1.0, 0.5, 1e-300, 0.2])
    with np.errstate(all="ignore"):
        fisher = scipy.stats.chi2.sf(-2 * (np.log(p1) + np.log(p2)), df=4)
    save_npz("fisher_golden.npz", p1=p1, p2=p2, out=fisher)


def gen_gene_stats():
    rng = np.random.default_rng(5)
    n = 3000
    genes = ["G%d" % i for i in range(n)]
    genes[7] = "TP53"
    mu = np.exp(rng.uniform(np.log(0.5), np.log(300.0), n))
    sigma = mu * np.exp(rng.uniform(np.log(0.05), np.log(1.2), n))
    P = {c: np.exp(rng.uniform(np.log(1e-4), np.log(0.2), n)) for c in ["SILENT", "MIS", "NONS", "SPLICE", "INDEL"]}
    frame = pd.DataFrame(dict(
        CHROM=rng.integers(1, 23, n).astype(str), GENE=genes, GENE_LENGTH=rng.integers(300, 9000, n),
        R_SIZE=rng.integers(9000, 50000, n), R_OBS=rng.integers(0, 800, n), R_INDEL=rng.integers(0, 800, n),
        MU=mu, SIGMA=sigma, MU_INDEL=mu, SIGMA_INDEL=sigma, FLAG=rng.integers(0, 3, n),
        P_MIS=P["MIS"], P_NONS=P["NONS"], P_SILENT=P["SILENT"], P_SPLICE=P["SPLICE"],
        P_TRUNC=P["NONS"] + P["SPLICE"], P_INDEL=P["INDEL"]))
    _HDF_FRAMES[("mem://genes.h5", "genic_model")] = frame
    df_pre = ref_tt.load_pretrained_model("mem://genes.h5")
    # synthetic CDS mutation table
    annots = np.array(["Synonymous", "Missense", "Nonsense", "Essential_Splice", "INDEL"])
    rows = []
    m = 40000
    gi = rng.integers(0, n, m)
    # hot genes
    gi[:1500] = rng.integers(0, 20, 1500)
    for j in range(m):
        rows.append((str(frame.CHROM.iloc[gi[j]]), int(rng.integers(1, 10 ** 6)), 0, "A", "T",
                     "S%d" % rng.integers(0, 300), genes[gi[j]], annots[rng.choice(5, p=[.25, .5, .05, .03, .17])],
                     "A>T", "CAG"))
    df_mut = pd.DataFrame(rows, columns=["CHROM", "START", "END", "REF", "ALT", "SAMPLE", "GENE", "ANNOT",
                                         "MUT_TYPE", "CONTEXT"])
    df_mut["END"] = df_mut.START + 1
    df_mut["CHROM"] = df_mut.CHROM.astype(int)
    df_mut_f = ref_mt.filter_hypermut_samples(df_mut, 170)
    df_cnt = ref_mt.mutations_per_gene(df_mut_f, max_muts_per_gene_per_sample=3)
    exp_mut = (df_pre[df_pre.index != 'TP53'].MU * df_pre[df_pre.index != 'TP53'].Pi_SYN).sum()
    cj = len(df_mut_f[(df_mut_f.GENE != 'TP53') & (df_mut_f.ANNOT == 'Synonymous')]) / exp_mut
    with np.errstate(all="ignore"):
        df = ref_tt.transfer_gene_model(df_mut_f, df_cnt, df_pre, cj)
        df = ref_tt.gene_expected_muts_nb(df)
        df = ref_tt.gene_pvalue_burden_nb(df)
        df = ref_tt.gene_pvalue_burden_nb_by_sample(df)
        # gene_pvalue_indel reads a packaged gene panel via pkg_resources; restate its two
        # lines with an explicit null set (the reference formula is transfer_tools.py:709-727)
        null = ~df.index.isin(["G1", "G2", "G3"])
        t_indel = df[null].OBS_INDEL.sum() / (df[null].Pi_INDEL * df[null].ALPHA_INDEL * df[null].THETA_INDEL).sum()
        df['THETA_INDEL'] = df.THETA_INDEL * t_indel
        df['EXP_INDEL'] = df.ALPHA_INDEL * df.THETA_INDEL * df.Pi_INDEL
        df['PVAL_INDEL_BURDEN'] = ref_nb.nb_pvalue_greater_midp(df.OBS_INDEL, df.ALPHA_INDEL,
                                                               1 / (df.THETA_INDEL * df.Pi_INDEL + 1))
        x2 = -2 * (np.log(df.PVAL_TRUNC_BURDEN) + np.log(df.PVAL_INDEL_BURDEN))
        df['PVAL_MUT_BURDEN'] = scipy.stats.chi2.sf(x2, df=4)
    buf = io.StringIO()
    df_mut.to_csv(buf, sep="\t", header=False, index=False)
    with open(os.path.join(HERE, "gene_mutations.tsv"), "w") as f:
        f.write(buf.getvalue())
    num_cols = [c for c in df.columns if c != "CHROM"]
    save_npz("gene_stats_golden.npz", genes=np.array(genes),
             frame_cols=np.array([c for c in frame.columns if c not in ("GENE", "CHROM")]),
             frame_vals=frame[[c for c in frame.columns if c not in ("GENE", "CHROM")]].values.astype(float),
             frame_chrom=frame.CHROM.values.astype(str),
             max_muts_per_sample=np.int64(170), max_muts_per_gene_per_sample=np.int64(3),
             cj=np.float64(cj), t_indel=np.float64(t_indel), null_excluded=np.array(["G1", "G2", "G3"]),
             cnt_index=np.array(df_cnt.index).astype(str), cnt_cols=np.array(df_cnt.columns).astype(str), cnt_vals=df_cnt.values.astype(np.int64),
             out_index=np.array(df.index).astype(str), out_cols=np.array(num_cols).astype(str), out_vals=df[num_cols].values.astype(float))


# ----------------------------------------------------------------------------
# (iv) get_ideal_overlaps
# ----------------------------------------------------------------------------
def gen_overlaps():
    rng = np.random.default_rng(9)
    cases = []
    fixed = [
        (1, [[100], [900]], 10000),
        (1, [[10000], [20000]], 10000),           # both ends on bin edges
        (2, [[9999], [10001]], 10000),
        (3, [[0], [10000]], 10000),
        (4, [[5, 25000, 99990], [500, 31000, 100010]], 10000),
        (5, [[19990, 20010], [20000, 20020]], 10000),  # block ending exactly on an edge
        (6, [[12345], [12345]], 10000),                # zero-length block
        (7, [[20000], [20000]], 10000),                # zero-length block on an edge -> no bin
        (8, [[150], [950]], 100),
    ]
    for chrom, iv, w in fixed:
        cases.append((chrom, iv, w))
    for _ in range(60):
        nb = int(rng.integers(1, 5))
        starts = np.sort(rng.integers(0, 2_000_000, nb))
        ends = starts + rng.integers(1, 30000, nb)
        cases.append((int(rng.integers(1, 23)), [starts.tolist(), ends.tolist()], int(rng.choice([10000, 1000, 50]))))
    out = []
    for chrom, iv, w in cases:
        res = ref_gdt.get_ideal_overlaps(chrom, np.array(iv), w)
        out.append(dict(chrom=chrom, intervals=iv, window=w,
                        overlaps=sorted([[int(a), int(b), int(c)] for a, b, c in res])))
    with open(os.path.join(HERE, "overlaps_golden.json"), "w") as f:
        json.dump(out, f)
    print("wrote overlaps_golden.json", len(out))


# ----------------------------------------------------------------------------
# (vii) substitution index + reverse-complement permutation
# ----------------------------------------------------------------------------
def gen_subst_index():
    trans_idx = ref_seq.mk_trans_idx(n_up=1, n_down=1, collapse=False)
    ctx64 = list(ref_seq.mk_context_sequences(n_up=1, n_down=1, collapse=False).keys())
    df_empty = ref_seq.mk_mutation_context(n_up=1, n_down=1, collapse=False, return_df=True)
    # the permutation applied to a 192-vector for '-' strand elements (sequence_tools.py:610-634)
    subst_idx = sorted(trans_idx)
    revc = [ref_seq.reverse_complement(s.split('>')[0]) + '>' + ref_seq.reverse_complement(s.split('>')[-1])
            for s in subst_idx]
    revc_dic = dict(zip(subst_idx, revc))
    probe = np.arange(192)
    permuted = [r[1] for r in sorted(enumerate(probe), key=lambda k: revc_dic[subst_idx[k[0]]])]
    # 96-type (collapse=True) listing kept for the optional K=96 mode
    trans96 = ref_seq.mk_trans_idx(n_up=1, n_down=1, collapse=True)
    with open(os.path.join(HERE, "subst_index.json"), "w") as f:
        json.dump(dict(subst_idx=subst_idx, context64=ctx64, revc=revc,
                       minus_strand_gather=[int(x) for x in permuted],
                       model_rows=[[a, b] for a, b in zip(df_empty.MUT_TYPE, df_empty.CONTEXT)],
                       subst_idx_96=trans96), f)
    print("wrote subst_index.json")
    return subst_idx, ctx64, df_empty


# ----------------------------------------------------------------------------
# (v) accumulation: the reference's own nonc_model / tiled_nonc_model / genic_model
#     executed over in-memory h5 stand-ins
# ----------------------------------------------------------------------------
def synth_region_params(rng, chroms, bins_per_chrom, window):
    rows = []
    for c in chroms:
        for b in range(bins_per_chrom):
            rows.append((c, b * window, (b + 1) * window))
    idx = np.array(rows, dtype=np.int64)
    n = len(idx)
    y_pred = rng.gamma(9.0, 3.0, n)
    std = rng.gamma(4.0, 1.0, n)
    y_true = rng.poisson(y_pred)
    flag = rng.uniform(size=n) < 0.1
    df = pd.DataFrame(dict(CHROM=idx[:, 0], START=idx[:, 1], END=idx[:, 2], Y_TRUE=y_true,
                           Y_PRED=y_pred, STD=std, MAPP=rng.uniform(0.4, 1, n), QUANT=rng.uniform(0, 1, n),
                           FLAG=flag))
    df.index = ['chr{}:{}-{}'.format(*r) for r in idx]
    return idx, df


def gen_accumulate(subst_idx, ctx64, df_empty):
    rng = np.random.default_rng(21)
    window = 10000
    chroms = [1, 2, 21]
    idx, df_reg = synth_region_params(rng, chroms, 60, window)
    n = len(idx)
    # 64-context counts per bin ~ Multinomial(window, Dirichlet(1))
    ctx_p = rng.dirichlet(np.ones(64))
    bin_ctx = rng.multinomial(window, ctx_p, size=n).astype(np.int64)
    # sequence_model_192 frame in the reference's row order (mk_mutation_context)
    freq = rng.dirichlet(np.ones(192)) * 1e-6 * 192
    df_seq = df_empty.copy()
    df_seq["COUNT"] = rng.integers(0, 1000, 192)
    df_seq["FREQ"] = freq
    f_pre, f_dat = "mem://pretrained.h5", "mem://element_data.h5"
    _HDF_FRAMES[(f_pre, "region_params")] = df_reg
    _HDF_FRAMES[(f_pre, "sequence_model_192")] = df_seq
    idx_dict = {tuple(int(v) for v in r): i for i, r in enumerate(idx)}

    revc = [ref_seq.reverse_complement(s.split('>')[0]) + '>' + ref_seq.reverse_complement(s.split('>')[-1])
            for s in subst_idx]
    revc_dic = dict(zip(subst_idx, revc))

    # elements: 1-3 blocks, 200-3000 bp, +/- strand; L = per-context counts repeated x3
    E = 400
    save_key = "elts"
    tree = {"window_%d" % window: {save_key: {}, "full_window_si_values": bin_ctx, "full_window_si_index": idx}}
    elt_names, strands, blocks_s, blocks_e, chrom_of, L_all, rc_all, ov_all = [], [], [], [], [], [], [], []
    for e in range(E):
        chrom = int(rng.choice(chroms))
        nb = int(rng.integers(1, 4))
        base = int(rng.integers(0, 58 * window))
        starts, ends = [], []
        pos = base
        for _ in range(nb):
            s = pos + int(rng.integers(0, 4000))
            ln = int(rng.integers(200, 3000))
            starts.append(s)
            ends.append(s + ln)
            pos = s + ln
        if e % 37 == 0:   # block ending exactly on a bin edge
            ends[-1] = (ends[-1] // window + 1) * window
        strand = "+" if rng.uniform() < 0.5 else "-"
        iv = np.vstack((starts, ends))
        overlaps = ref_gdt.get_ideal_overlaps(chrom, iv, window)
        # region_counts exactly as sequence_tools.preprocess_nonc builds them (:630-634)
        region_counts = np.array([np.repeat(bin_ctx[idx_dict[r], :], 3) for r in overlaps]).sum(axis=0)
        if strand == '-':
            region_counts = [r[1] for r in sorted(enumerate(region_counts), key=lambda k: revc_dic[subst_idx[k[0]]])]
        region_counts = np.asarray(region_counts)
        tot = int(sum(e_ - s_ for s_, e_ in zip(starts, ends)))
        L64 = rng.multinomial(tot, ctx_p)
        L = np.repeat(L64, 3).astype(np.float64)
        name = "elt_%04d" % e
        tree["window_%d" % window][save_key][name] = {
            "L_counts": L, "region_counts": region_counts,
            "__attrs__": {"overlaps": np.array(overlaps)}}
        elt_names.append(name); strands.append(strand); chrom_of.append(chrom)
        blocks_s.append(starts); blocks_e.append(ends)
        L_all.append(L); rc_all.append(region_counts)
        ov_all.append(sorted([idx_dict[tuple(int(v) for v in r)] for r in overlaps]))
    _H5_FILES[f_dat] = tree

    df_out = ref_gdt.nonc_model(elt_names, f_pre, f_dat, save_key, False)

    # tiled route: elements are "chr{c}:{s}-{e}" tiles inside one bin; L table via read_hdf
    tiles, tile_L = [], []
    for t in range(150):
        chrom = int(rng.choice(chroms))
        b = int(rng.integers(0, 60))
        s = b * window + int(rng.integers(0, window - 50))
        tiles.append("chr{}:{}-{}".format(chrom, s, s + 50))
        tile_L.append(np.repeat(rng.multinomial(50, ctx_p), 3).astype(np.float64))
    tile_key = "tiles"
    _HDF_FRAMES[(f_dat, "{}/L_counts".format(tile_key))] = pd.DataFrame(np.array(tile_L), index=tiles, columns=subst_idx)
    df_tiled = ref_gdt.tiled_nonc_model(pd.Index(tiles), f_pre, f_dat, tile_key)

    cols = ['ELT_SIZE', 'FLAG', 'R_SIZE', 'R_OBS', 'R_INDEL', 'MU', 'SIGMA', 'MU_INDEL', 'SIGMA_INDEL', 'P_SUM', 'P_INDEL']
    max_ov = max(len(o) for o in ov_all)
    ov_pad = -np.ones((E, max_ov), dtype=np.int64)
    for i, o in enumerate(ov_all):
        ov_pad[i, :len(o)] = o
    maxb = 3
    bs = -np.ones((E, maxb), np.int64); be = -np.ones((E, maxb), np.int64)
    for i in range(E):
        bs[i, :len(blocks_s[i])] = blocks_s[i]; be[i, :len(blocks_e[i])] = blocks_e[i]
    save_npz("accumulate_golden.npz",
             window=np.int64(window), bin_idx=idx, bin_y_pred=df_reg.Y_PRED.values, bin_std=df_reg.STD.values,
             bin_y_true=df_reg.Y_TRUE.values.astype(np.int64), bin_flag=df_reg.FLAG.values,
             bin_ctx=bin_ctx, seq_mut_type=df_seq.MUT_TYPE.values.astype(str), seq_context=df_seq.CONTEXT.values.astype(str),
             seq_freq=freq, elt_names=np.array(elt_names), elt_chrom=np.array(chrom_of), elt_strand=np.array(strands),
             block_starts=bs, block_ends=be, elt_L=np.array(L_all), elt_region_counts=np.array(rc_all),
             elt_overlap_bins=ov_pad, out_cols=np.array(cols), out_vals=df_out[cols].values.astype(float),
             tile_names=np.array(tiles), tile_L=np.array(tile_L), tile_out_names=df_tiled.ELT.values.astype(str),
             tile_out_vals=df_tiled[cols].values.astype(float))

    # ---- genic_model (4 mutation classes) through the same in-memory stand-ins ----
    G = 120
    f_gen = "mem://genic.h5"
    gtree = {"substitution_idx": np.array([s.encode() for s in subst_idx]), "chr": {}, "cds_intervals": {},
             "L_data": {}}
    counts_rows, gnames = [], []
    g_int, g_L, g_chr, g_ov = [], [], [], []
    for g in range(G):
        chrom = int(rng.choice(chroms))
        nb = int(rng.integers(1, 6))
        base = int(rng.integers(0, 55 * window))
        starts, ends, pos = [], [], base
        for _ in range(nb):
            s = pos + int(rng.integers(50, 9000)); ln = int(rng.integers(60, 900))
            starts.append(s); ends.append(s + ln); pos = s + ln
        iv = np.vstack((starts, ends))
        overlaps = ref_gdt.get_ideal_overlaps(str(chrom), iv, window)
        rc = np.array([np.repeat(bin_ctx[idx_dict[(int(a), int(b), int(c))], :], 3) for a, b, c in overlaps]).sum(axis=0)
        Ld = np.zeros((4, 192))
        tot = sum(e_ - s_ + 1 for s_, e_ in zip(starts, ends))
        split = rng.multinomial(3 * tot, [0.22, 0.68, 0.04, 0.06])
        for c in range(4):
            Ld[c] = rng.multinomial(split[c], np.repeat(ctx_p, 3) / 3.0)
        name = "GENE%03d" % g
        gnames.append(name)
        gtree["chr"][name] = np.array([str(chrom).encode()])
        gtree["cds_intervals"][name] = iv
        gtree["L_data"][name] = Ld
        counts_rows.append(rc)
        g_int.append(iv); g_L.append(Ld); g_chr.append(chrom)
        g_ov.append(sorted(idx_dict[(int(a), int(b), int(c))] for a, b, c in overlaps))
    gtree["chr"]["GENEX"] = np.array([b"X"])   # sex-chromosome gene must be skipped (genic_driver_tools.py:90-92)
    gtree["cds_intervals"]["GENEX"] = np.array([[100], [400]])
    gtree["L_data"]["GENEX"] = np.zeros((4, 192))
    _H5_FILES[f_gen] = gtree
    _HDF_FRAMES[(f_gen, "window_10kb/counts")] = pd.DataFrame(np.array(counts_rows + [np.zeros(192)]),
                                                              index=gnames + ["GENEX"], columns=subst_idx)
    df_gen = ref_gdt.genic_model(gnames + ["GENEX"], f_pre, f_gen, "window_10kb/counts", False)
    gcols = ['GENE_LENGTH', 'R_SIZE', 'R_OBS', 'R_INDEL', 'MU', 'SIGMA', 'MU_INDEL', 'SIGMA_INDEL', 'FLAG',
             'P_MIS', 'P_NONS', 'P_SILENT', 'P_SPLICE', 'P_TRUNC', 'P_INDEL']
    maxb = max(iv.shape[1] for iv in g_int)
    gs = -np.ones((G, maxb), np.int64); ge = -np.ones((G, maxb), np.int64)
    for i, iv in enumerate(g_int):
        gs[i, :iv.shape[1]] = iv[0]; ge[i, :iv.shape[1]] = iv[1]
    max_ov = max(len(o) for o in g_ov)
    gov = -np.ones((G, max_ov), np.int64)
    for i, o in enumerate(g_ov):
        gov[i, :len(o)] = o
    save_npz("genic_golden.npz", gene_names=np.array(gnames), gene_chrom=np.array(g_chr), cds_starts=gs, cds_ends=ge,
             L_data=np.array(g_L), region_counts=np.array(counts_rows), gene_overlap_bins=gov,
             out_genes=df_gen.GENE.values.astype(str), out_chrom=df_gen.CHROM.values.astype(str),
             out_cols=np.array(gcols), out_vals=df_gen[gcols].values.astype(float))


# ----------------------------------------------------------------------------
# (viii) integer mutation tabulation helpers
# ----------------------------------------------------------------------------
def gen_accumulate_zero_den(subst_idx, df_empty):
    """Zero denominators (round 5): the reference's own nonc_model where prob_sum.sum() == 0 (genic_driver_tools.py:361-364:
    t_pi = d_pr / 0).  With bins that hold no countable context at all the reference does not get that far (R_size = 0:
    ZeroDivisionError at :381), so the reachable case is the one in which the cohort's FREQ table is exactly zero at every
    context the element's bins hold: t_pi is 0 / 0 = NaN at those substitutions and P_SUM is NaN whatever L is -- rows of L with and
    without a zero.  Three FREQ tables: all positive (finite everywhere); zero at the six substitutions of the two contexts the
    first five bins consist of (NaN for the elements inside them); zero at one other substitution only (finite everywhere)."""
    rng = np.random.default_rng(77)
    window = 10000
    chroms = [1]
    idx, df_reg = synth_region_params(rng, chroms, 14, window)
    n = len(idx)
    ctx_p = rng.dirichlet(np.ones(64))
    bin_ctx = rng.multinomial(window, ctx_p, size=n).astype(np.int64)
    only = [3, 10]                                         # the first five bins consist of two contexts only
    bin_ctx[:5] = 0
    bin_ctx[:5, only] = rng.integers(1000, 5000, (5, 2))
    idx_dict = {tuple(int(v) for v in r): i for i, r in enumerate(idx)}
    revc = [ref_seq.reverse_complement(s.split('>')[0]) + '>' + ref_seq.reverse_complement(s.split('>')[-1]) for s in subst_idx]
    revc_dic = dict(zip(subst_idx, revc))
    f_pre, f_dat = "mem://pretrained_zero.h5", "mem://element_data_zero.h5"
    _HDF_FRAMES[(f_pre, "region_params")] = df_reg
    save_key = "elts"
    tree = {"window_%d" % window: {save_key: {}, "full_window_si_values": bin_ctx, "full_window_si_index": idx}}
    E = 36
    names, strands, L_all, ov_all = [], [], [], []
    for e in range(E):
        zero_bins = e % 3 != 2                             # two elements in three lie in the empty bins
        b0 = int(rng.integers(0, 4)) if zero_bins else int(rng.integers(6, 12))
        s0 = b0 * window + int(rng.integers(0, 4000))
        ln = int(rng.integers(300, 9000 if e % 4 == 0 else 3000))      # some reach into a second bin
        if zero_bins and (s0 + ln) // window > 4:
            ln = 5 * window - s0 - 1
        iv = np.array([[s0], [s0 + ln]])
        overlaps = ref_gdt.get_ideal_overlaps(1, iv, window)
        region_counts = np.array([np.repeat(bin_ctx[idx_dict[r], :], 3) for r in overlaps]).sum(axis=0)
        strand = "+" if e % 2 == 0 else "-"
        if strand == '-':
            region_counts = np.asarray([r[1] for r in sorted(enumerate(region_counts), key=lambda k: revc_dic[subst_idx[k[0]]])])
        L64 = rng.multinomial(ln, ctx_p)
        if e % 6 < 3:
            L64 = L64 + 1                                   # no zero anywhere in L
        else:
            L64[int(rng.integers(0, 64))] = 0               # a zero in L
        L = np.repeat(L64, 3).astype(np.float64)
        name = "z_%03d" % e
        tree["window_%d" % window][save_key][name] = {"L_counts": L, "region_counts": region_counts,
                                                      "__attrs__": {"overlaps": np.array(overlaps)}}
        names.append(name); strands.append(strand); L_all.append(L)
        ov_all.append(sorted([idx_dict[tuple(int(v) for v in r)] for r in overlaps]))
    _H5_FILES[f_dat] = tree
    freqs = []
    f0 = rng.dirichlet(np.ones(192)) * 1e-6 * 192
    ctx_names = sorted(set(df_empty.CONTEXT.values))       # context index 16 b0 + 4 b1 + b2 <-> sorted trinucleotides
    f1 = f0.copy()
    for c in only:
        f1[df_empty.CONTEXT.values == ctx_names[c]] = 0.0
    f2 = f0.copy(); f2[17] = 0.0
    outs = []
    for fr in (f0, f1, f2):
        df_seq = df_empty.copy()
        df_seq["COUNT"] = 1
        df_seq["FREQ"] = fr
        _HDF_FRAMES[(f_pre, "sequence_model_192")] = df_seq
        with np.errstate(all="ignore"):
            df_out = ref_gdt.nonc_model(names, f_pre, f_dat, save_key, False)
        outs.append(df_out["P_SUM"].values.astype(float))
        freqs.append(fr)
    max_ov = max(len(o) for o in ov_all)
    ov_pad = -np.ones((E, max_ov), dtype=np.int64)
    for i, o in enumerate(ov_all):
        ov_pad[i, :len(o)] = o
    save_npz("accumulate_zero_golden.npz", bin_y_pred=df_reg.Y_PRED.values, bin_std=df_reg.STD.values,
             bin_y_true=df_reg.Y_TRUE.values.astype(np.int64), bin_flag=df_reg.FLAG.values, bin_ctx=bin_ctx,
             seq_mut_type=df_empty.MUT_TYPE.values.astype(str), seq_context=df_empty.CONTEXT.values.astype(str),
             seq_freq=np.array(freqs), elt_strand=np.array(strands), elt_L=np.array(L_all), elt_overlap_bins=ov_pad,
             p_sum=np.array(outs))


def gen_mutation_tools():
    rng = np.random.default_rng(13)
    m = 5000
    genes = ["G%d" % i for i in range(40)]
    annots = np.array(["Synonymous", "Missense", "Nonsense", "Essential_Splice", "INDEL", "Noncoding"])
    df = pd.DataFrame(dict(
        CHROM=rng.integers(1, 23, m), START=rng.integers(1, 5000, m), REF=rng.choice(list("ACGT"), m),
        ALT=rng.choice(list("ACGT"), m), SAMPLE=["S%d" % s for s in rng.integers(0, 60, m)],
        GENE=rng.choice(genes, m), ANNOT=rng.choice(annots, m, p=[.2, .4, .05, .05, .2, .1]),
        MUT_TYPE="A>T", CONTEXT="CAG"))
    df.insert(2, "END", df.START + 1)
    df = pd.concat([df, df.iloc[:300]])  # exact duplicates
    path = os.path.join(HERE, "mutations_small.tsv")
    df.to_csv(path, sep="\t", header=False, index=False)
    rd = ref_mt.read_mutation_file(path, drop_sex=True, drop_duplicates=True, unique_indels=True)
    rd2 = ref_mt.read_mutation_file(path, drop_sex=True, drop_duplicates=False, unique_indels=True)
    filt, black = ref_mt.filter_hypermut_samples(rd2, 95, return_blacklist=True)
    cnt = ref_mt.mutations_per_gene(rd2[rd2.GENE != '.'], max_muts_per_gene_per_sample=2)
    # bed12
    bed = []
    for i in range(25):
        chrom = "chr%d" % rng.integers(1, 23) if i % 2 else str(rng.integers(1, 23))
        nb = int(rng.integers(1, 4)); start = int(rng.integers(0, 10 ** 6))
        sizes = rng.integers(50, 500, nb); rel = np.concatenate([[0], np.cumsum(sizes + rng.integers(10, 300, nb))[:-1]])
        bed.append([chrom, start, start + int(rel[-1] + sizes[-1]), "E%d" % i, 0, "+-"[i % 2], start, start, ".", nb,
                    ",".join(map(str, sizes)) + ("," if i % 3 == 0 else ""), ",".join(map(str, rel)) + ("," if i % 3 == 0 else "")])
    bed.append(["chrX", 5, 100, "EX", 0, "+", 5, 5, ".", 1, "95,", "0,"])
    # bed12_boundaries lstrips 'chr' only if the FIRST row has it -> keep first row 'chr'-prefixed
    bed[0][0] = "chr3"
    for r in bed:
        if not str(r[0]).startswith("chr"):
            r[0] = "chr" + str(r[0])
    bed_path = os.path.join(HERE, "elements_small.bed")
    pd.DataFrame(bed).to_csv(bed_path, sep="\t", header=False, index=False)
    bb = ref_mt.bed12_boundaries(bed_path)
    with open(os.path.join(HERE, "mutation_tools_golden.json"), "w") as f:
        json.dump(dict(
            n_read_dedup=int(len(rd)), n_read_nodedup=int(len(rd2)),
            read_dedup_rows=rd.astype(str).values.tolist()[:50],
            read_dedup_checksum=int(pd.util.hash_pandas_object(rd.reset_index(drop=True).astype(str), index=False).sum() % (2 ** 61)),
            blacklist=sorted(black), n_filtered=int(len(filt)),
            cnt_index=list(cnt.index), cnt_cols=list(cnt.columns), cnt_vals=cnt.values.astype(int).tolist(),
            bed12=dict(CHROM=[int(c) for c in bb.CHROM], ELT=list(bb.ELT), STRAND=list(bb.STRAND),
                       BLOCK_STARTS=[list(map(int, x)) for x in bb.BLOCK_STARTS],
                       BLOCK_ENDS=[list(map(int, x)) for x in bb.BLOCK_ENDS])), f)
    print("wrote mutation_tools_golden.json")


# ----------------------------------------------------------------------------
# (ix) sequence model training (192 / 64 frequency tables)
# ----------------------------------------------------------------------------
def gen_sequence_model(df_empty):
    rng = np.random.default_rng(17)
    # whitelisting by bed needs bedtools; the golden uses an identity whitelist (all mutations inside)
    ref_mt.restrict_mutations_by_bed = lambda df_mut, df_bed, unique=True, remove_X=True, replace_cols=False: \
        (df_mut.drop_duplicates() if unique else df_mut).copy()
    m = 30000
    rows = df_empty.sample(m, replace=True, random_state=1).reset_index(drop=True)
    df_mut = pd.DataFrame(dict(CHROM=rng.integers(1, 23, m), START=rng.integers(0, 10 ** 7, m)))
    df_mut["END"] = df_mut.START + 1
    df_mut["REF"] = [c[1] for c in rows.CONTEXT]
    df_mut["ALT"] = [t[2] for t in rows.MUT_TYPE]
    df_mut["SAMPLE"] = ["S%d" % s for s in rng.integers(0, 50, m)]
    df_mut["ANNOT"] = "Noncoding"
    df_mut["MUT_TYPE"] = rows.MUT_TYPE.values
    df_mut["CONTEXT"] = rows.CONTEXT.values
    ctx = sorted(set(df_empty.CONTEXT))
    S_genome = pd.Series(rng.integers(10 ** 6, 10 ** 8, 64), index=ctx)
    regions = np.array([[1, 0, 10000]])
    df192, df64 = ref_seq.train_sequence_model(regions, df_mut, S_genome)
    save_npz("sequence_model_golden.npz", mut_type=df_mut.MUT_TYPE.values.astype(str),
             context=df_mut.CONTEXT.values.astype(str),
             dedup_key=pd.util.hash_pandas_object(df_mut, index=False).values,
             genome_ctx=np.array(ctx), genome_counts=S_genome.values.astype(np.int64),
             out_mut_type=df192.MUT_TYPE.values.astype(str), out_context=df192.CONTEXT.values.astype(str),
             out_count=df192.COUNT.values.astype(np.int64), out_freq=df192.FREQ.values,
             out64_context=np.array(df64.index).astype(str), out64_freq=df64.FREQ.values)


# ----------------------------------------------------------------------------
# (vi) CNN forward (seeded weights regenerated from the seed; only data stored)
# ----------------------------------------------------------------------------
def gen_cnn():
    import importlib.util
    import torch
    spec = importlib.util.spec_from_file_location(
        "ref_cnn", os.path.join(REF, "DIGDriver/region_model/nets/cnn_predictors.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    T, L, C, B = 32, 100, 3, 6
    torch.manual_seed(0)
    net = mod.SimpleMultiTaskResNet((B, L, T), C)
    # non-trivial BatchNorm statistics, drawn in module order from a dedicated generator
    g = torch.Generator().manual_seed(1)
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm1d):
            m.running_mean.copy_(torch.randn(m.num_features, generator=g) * 0.1)
            m.running_var.copy_(torch.rand(m.num_features, generator=g) + 0.5)
            m.weight.data.copy_(torch.rand(m.num_features, generator=g) + 0.5)
            m.bias.data.copy_(torch.randn(m.num_features, generator=g) * 0.1)
    net.eval()
    gx = torch.Generator().manual_seed(2)
    x = torch.round(torch.rand(B, L, T, generator=gx), decimals=2) * 100
    with torch.no_grad():
        outs, feats, att = net(x)
    assert att is None
    save_npz("cnn_forward_golden.npz", x=x.numpy(), shape=np.array([B, L, T, C]),
             outputs=np.stack([o.numpy() for o in outs]), features=np.stack([f.numpy() for f in feats]),
             n_params=np.int64(sum(p.numel() for p in net.parameters())),
             first_conv_w_sum=np.float64(net.conv11.weight.double().sum().item()))


def gen_cnn_735():
    """The reference network at the whole-genome track count (T = 735, 5 heads): seeded weights and BatchNorm statistics
    (regenerated from the seeds by the test), 4 bins of x_data-like values (stored as uint8: round(u, 2) * 100), the
    reference module's own eval-mode outputs and 16-d features."""
    import importlib.util
    import torch
    spec = importlib.util.spec_from_file_location(
        "ref_cnn735", os.path.join(REF, "DIGDriver/region_model/nets/cnn_predictors.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    T, L, C, B = 735, 100, 5, 4
    torch.manual_seed(3)
    net = mod.SimpleMultiTaskResNet((B, L, T), C)
    g = torch.Generator().manual_seed(4)
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm1d):
            m.running_mean.copy_(torch.randn(m.num_features, generator=g) * 0.1)
            m.running_var.copy_(torch.rand(m.num_features, generator=g) + 0.5)
            m.weight.data.copy_(torch.rand(m.num_features, generator=g) + 0.5)
            m.bias.data.copy_(torch.randn(m.num_features, generator=g) * 0.1)
    net.eval()
    gx = torch.Generator().manual_seed(5)
    x = torch.round(torch.rand(B, L, T, generator=gx), decimals=2) * 100
    x = torch.round(x)                                            # exact integers 0..100, as the float32 image of x_data is
    with torch.no_grad():
        outs, feats, att = net(x)
    save_npz("cnn_forward_735_golden.npz", x=x.numpy().astype(np.uint8), shape=np.array([B, L, T, C]),
             outputs=np.stack([o.numpy() for o in outs]), features=np.stack([f.numpy() for f in feats]),
             first_conv_w_sum=np.float64(net.conv11.weight.double().sum().item()))


# ----------------------------------------------------------------------------
# (vi-b) one CNN training epoch + evaluation through the reference's NNTrainer (nn_trainer.py:17-141)
# ----------------------------------------------------------------------------
def gen_nn_training():
    """A seeded SimpleMultiTaskResNet (weights regenerated from the seed by the test), ten training and four validation
    bins, batch size 4 (3 batches: 4, 4, 2), one NNTrainer.train epoch (Adam 1e-3, summed per-task MSE, train-mode
    BatchNorm, features captured DURING the epoch) followed by NNTrainer.test.  The DataLoader's shuffle order is
    recovered from the labels the trainer returns (they are distinct) and stored, so that the other side can visit
    the bins in the same order."""
    import importlib.util
    import torch

    def load(name, rel):
        spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        return mod
    cnn = load("ref_cnn_t", "DIGDriver/region_model/nets/cnn_predictors.py")
    ref_tr = load("ref_nn_trainer", "DIGDriver/region_model/trainers/nn_trainer.py")
    T, L, C, n_train, n_val, bs = 6, 100, 2, 10, 4, 4
    gx = torch.Generator().manual_seed(6)
    x = torch.round(torch.rand(n_train + n_val, L, T, generator=gx), decimals=2) * 100
    labels = torch.rand(C, n_train + n_val, generator=gx) * 10

    class DS(torch.utils.data.Dataset):
        def __init__(self, rows):
            self.rows = rows

        def __len__(self):
            return len(self.rows)

        def __getitem__(self, i):
            r = self.rows[i]
            return x[r], [labels[c][r] for c in range(C)]

    torch.manual_seed(5)
    net = cnn.SimpleMultiTaskResNet((n_train, L, T), C)
    w0 = np.float64(net.conv11.weight.double().sum().item())
    opt = torch.optim.Adam(net.parameters(), lr=1e-3, amsgrad=False)
    trainer = ref_tr.NNTrainer(net, opt, torch.nn.MSELoss(), bs, ["a", "b"], DS(list(range(n_train))),
                               DS(list(range(n_train, n_train + n_val))), torch.device("cpu"))
    torch.manual_seed(7)                       # the shuffle of the epoch
    # cnn_predictors.py:148,153,158 add the residual IN PLACE to a ReLU output that ReLU's backward has saved: torch 2.x
    # refuses the backward pass ("modified by an inplace operation").  allow_mutation_on_saved_tensors keeps a copy of
    # the saved tensor as it was, i.e. the gradient of the network as written; nothing of the reference is changed.
    with torch.autograd.graph.allow_mutation_on_saved_tensors():
        tl, ta, feats, preds, true = trainer.train(1, 0, print_interval=100)   # (:76 divides by int(batches * interval / 100))
    lab0 = labels[0].numpy()
    order = np.array([int(np.argmin(np.abs(lab0 - v))) for v in true[0]])
    assert sorted(order.tolist()) == list(range(n_train))
    vl, va, vfeats, vpreds, vtrue, _ = trainer.test(1, 0)
    save_npz("nn_training_golden.npz", x=x.numpy(), labels=labels.numpy(), shape=np.array([T, L, C, n_train, n_val, bs]),
             first_conv_w_sum_before=w0, order=order, train_losses=tl, train_accs=ta,
             train_features=np.stack([np.stack(f) for f in feats]), train_preds=np.array(preds),
             val_losses=np.array([float(v) for v in vl]), val_accs=va, val_features=np.stack(vfeats), val_preds=np.array(vpreds),
             first_conv_w_sum_after=np.float64(net.conv11.weight.double().sum().item()),
             fc_w_sum_after=np.float64(sum(p.double().sum().item() for n_, p in net.named_parameters() if n_.startswith("fc") or "fc" in n_)),
             bn_running_mean_sum=np.float64(sum(m.running_mean.double().sum().item() for m in net.modules()
                                                if isinstance(m, torch.nn.BatchNorm1d))))


class _FakeFasta:
    """pysam.FastaFile look-alike over an in-memory genome: fetch(chrom, start, end) returns the stored text
    (case preserved), truncated at the end of the chromosome like pysam does."""
    genomes = {}

    def __init__(self, path):
        self._g = _FakeFasta.genomes[path]

    def fetch(self, chrom, start, end):
        assert start >= 0
        return self._g[chrom][start:end]


def gen_contexts():
    """Trinucleotide context counting (sequence_tools.py:21-29,65-94,527-566) on a small random genome with N runs
    and soft-masked stretches: 64-column counts per region and the 192-column strand-aware element counts."""
    rng = np.random.default_rng(64)
    genome = {}
    for chrom, n in (("chr1", 3000), ("chr2", 2111), ("chr3", 517)):
        seq = rng.choice(list("ACGT"), n)
        for _ in range(4):                                    # N runs
            a = int(rng.integers(0, n - 40))
            seq[a:a + int(rng.integers(1, 40))] = "N"
        for _ in range(3):                                    # soft-masked (lower-case) stretches
            a = int(rng.integers(0, n - 100))
            b = a + int(rng.integers(1, 100))
            seq[a:b] = np.char.lower(seq[a:b])
        genome[chrom] = "".join(seq)
    _FakeFasta.genomes["mem://genome"] = genome
    sys.modules["pysam"].FastaFile = _FakeFasta
    regions = [("chr1", 0, 100), ("chr1", 1, 100), ("chr1", 100, 100), ("chr1", 2900, 3000), ("chr1", 2950, 3100),
               ("chr2", 5, 6), ("chr2", 0, 2111), ("chr3", 0, 517), ("chr3", 500, 600)]
    for _ in range(40):
        chrom = ["chr1", "chr2", "chr3"][int(rng.integers(0, 3))]
        n = len(genome[chrom])
        a = int(rng.integers(0, n - 2))
        regions.append((chrom, a, a + int(rng.integers(1, 700))))
    chroms, starts, ends = zip(*regions)
    df64 = ref_seq.count_contexts_by_regions("mem://genome", list(chroms), list(starts), list(ends), n_up=1, n_down=1)
    strands = [["+", "-", -1, "+"][int(rng.integers(0, 4))] for _ in regions]
    trans_idx = ref_seq.mk_trans_idx(n_up=1, n_down=1, collapse=False)
    regs = [(c[3:], s, e, st) for (c, s, e), st in zip(regions, strands)]
    df192 = ref_seq.nonc_elt_context_count(regs, trans_idx, "mem://genome")
    out = dict(genome=genome, regions=[[c, int(s), int(e)] for c, s, e in regions],
               strands=[str(st) for st in strands], columns64=list(df64.columns), index64=list(df64.index),
               counts64=df64.values.astype(int).tolist(), columns192=list(df192.columns), index192=list(df192.index),
               counts192=df192.values.astype(int).tolist())
    with open(os.path.join(HERE, "contexts_golden.json"), "w") as f:
        json.dump(out, f)
    print("wrote contexts_golden.json", df64.shape, df192.shape)


def gen_run_element_expectation():
    """run_element_region_model in its DEFAULT mode (scale_by_expectation=True, transfer_tools.py:969-1096): the
    reference function itself, with the three things this image cannot provide handed in as fixtures -- the packaged CGC
    panel (pkg_resources.resource_stream -> a synthetic panel), the pretrained frames (pd.read_hdf -> in-memory frames) and
    the bedtools tabulation (tabulate_mutations_in_element -> a fixed table + sample blacklist).  Pins the synonymous
    scale factor on blacklist-filtered, de-duplicated rows and the uniform indel factor whose CGC exclusion of the
    mutation frame is a no-op (:1014)."""
    import io as _io
    rng = np.random.default_rng(969)
    g = np.load(os.path.join(HERE, "gene_stats_golden.npz"), allow_pickle=False)
    genes = [str(x) for x in g["genes"]]
    frame = pd.DataFrame(g["frame_vals"], columns=[str(c) for c in g["frame_cols"]])
    frame.insert(0, "GENE", genes)
    frame.insert(0, "CHROM", g["frame_chrom"])
    for c in ("GENE_LENGTH", "R_SIZE", "R_OBS", "R_INDEL", "FLAG"):
        frame[c] = frame[c].astype(int)
    n = 400
    mu = np.exp(rng.uniform(np.log(0.5), np.log(300.0), n))
    sigma = mu * np.exp(rng.uniform(np.log(0.05), np.log(1.2), n))
    elts = pd.DataFrame(dict(ELT=["elt%d" % i for i in range(n)], ELT_SIZE=rng.integers(100, 5000, n), FLAG=rng.integers(0, 2, n).astype(bool),
                             R_SIZE=rng.integers(9000, 30000, n), R_OBS=rng.integers(0, 500, n), R_INDEL=rng.integers(0, 500, n),
                             MU=mu, SIGMA=sigma, MU_INDEL=mu, SIGMA_INDEL=sigma,
                             P_SUM=np.exp(rng.uniform(np.log(1e-4), np.log(0.5), n)), P_INDEL=np.exp(rng.uniform(np.log(1e-4), np.log(0.5), n))))
    _HDF_FRAMES[("mem://run.h5", "genic_model")] = frame
    _HDF_FRAMES[("mem://run.h5", "myelts")] = elts
    panel = ["G%d" % i for i in range(0, 600, 3)] + ["TP53"]
    sys.modules["pkg_resources"].resource_stream = lambda pkg, name: _io.BytesIO(("\n".join(panel) + "\n").encode())
    ref_tt.pkg_resources = sys.modules["pkg_resources"]
    present = rng.uniform(size=n) > 0.25
    tab = pd.DataFrame(dict(OBS_SAMPLES=rng.poisson(3, n), OBS_SNV=rng.poisson(4, n), OBS_INDEL=rng.poisson(0.7, n)),
                       index=pd.Index(elts.ELT.values, name="ELT"))[present].astype(float)
    blacklist = ["S7", "S123", "S250"]
    real_tab = ref_mt.tabulate_mutations_in_element
    ref_mt.tabulate_mutations_in_element = lambda *a, **k: (tab.copy(), list(blacklist))
    try:
        with np.errstate(all="ignore"):
            out = ref_tt.run_element_region_model(os.path.join(HERE, "gene_mutations.tsv"), "unused.bed", "mem://run.h5", "myelts",
                                                  scale_by_expectation=True)
    finally:
        ref_mt.tabulate_mutations_in_element = real_tab
    num = [c for c in out.columns]
    save_npz("run_element_expectation_golden.npz", elt_cols=np.array([c for c in elts.columns if c != "ELT"]),
             elt_vals=elts[[c for c in elts.columns if c != "ELT"]].values.astype(float), elt_names=np.array(elts.ELT.values).astype(str),
             panel=np.array(panel), tab_index=np.array(tab.index).astype(str), tab_vals=tab[["OBS_SAMPLES", "OBS_SNV", "OBS_INDEL"]].values,
             blacklist=np.array(blacklist), out_index=np.array(out.index).astype(str), out_cols=np.array(num).astype(str),
             out_vals=out[num].values.astype(float))
    print("wrote run_element_expectation_golden.npz", out.shape)


def gen_run_target():
    """run_target_model (transfer_tools.py:876-967): the reference function itself on tests/golden/gene_mutations.tsv, once
    per scale rule (mutations in the panel, samples in the panel, manual).  The packaged panel file
    (pkg_resources.resource_stream) and the pretrained map (pd.read_hdf + h5py attrs) are handed in as fixtures."""
    import io as _io
    g = np.load(os.path.join(HERE, "gene_stats_golden.npz"), allow_pickle=False)
    genes = [str(x) for x in g["genes"]]
    frame = pd.DataFrame(g["frame_vals"], columns=[str(c) for c in g["frame_cols"]])
    frame.insert(0, "GENE", genes)
    frame.insert(0, "CHROM", g["frame_chrom"])
    for c in ("GENE_LENGTH", "R_SIZE", "R_OBS", "R_INDEL", "FLAG"):
        frame[c] = frame[c].astype(int)
    _HDF_FRAMES[("mem://target.h5", "genic_model")] = frame
    attrs = {"N_MUT_PANELX": 1234, "N_MUT_SAMPLE_PANELX": 987, "N_SAMPLE_PANELX": 211}
    _H5_FILES["mem://target.h5"] = {"__attrs__": attrs}
    panel = ["G%d" % i for i in range(0, 400, 2)] + ["TP53", "NOT_IN_MODEL"]
    sys.modules["pkg_resources"].resource_stream = lambda pkg, name: _io.BytesIO(("\n".join(panel) + "\n").encode())
    ref_tt.pkg_resources = sys.modules["pkg_resources"]
    outs = {}
    for tag, kw in (("mut", {}), ("sample", dict(scale_by_sample=True)), ("manual", dict(scale_factor=0.37)),
                    ("capped", dict(max_muts_per_sample=170, max_muts_per_gene_per_sample=3, drop_synonymous=False))):
        with np.errstate(all="ignore"):
            out = ref_tt.run_target_model(os.path.join(HERE, "gene_mutations.tsv"), "mem://target.h5", panel="PANELX", **kw)
        num = [c for c in out.columns if c != "CHROM"]
        outs[tag + "_index"] = np.array(out.index).astype(str)
        outs[tag + "_cols"] = np.array(num).astype(str)
        outs[tag + "_vals"] = out[num].values.astype(float)
    save_npz("run_target_golden.npz", panel=np.array(panel), attr_names=np.array(list(attrs)), attr_vals=np.array(list(attrs.values())),
             **outs)


class _FakeTabix:
    """pysam.TabixFile look-alike over in-memory mutation rows (chrom, start, end, ref, alt, id): fetch(chrom, start,
    end) yields the tab-joined rows overlapping [start, end), as tabix does for 0-based half-open bed intervals."""
    tables = {}

    def __init__(self, path):
        self._rows = _FakeTabix.tables[path]

    def fetch(self, chrom, start, end):
        for r in self._rows:
            if r[0] == chrom and r[1] < end and r[2] > start:
                yield "\t".join(str(x) for x in r)


def gen_tiled():
    """The per-base / tiled route (nb_model.py:126-234 -> sequence_tools.py:292-317, nb_model.py:298-314): the
    reference's own nb_model on a small genome (N runs, soft-masked stretches, a bin at position 0 and one cut off by
    the chromosome end), trinucleotide contexts (n_up = n_down = 1), tiles of 1 and of 50 positions, two cohorts."""
    rng = np.random.default_rng(126)
    genome = {}
    for chrom, n in (("chr1", 2600), ("chr2", 1437)):
        seq = rng.choice(list("ACGT"), n)
        for _ in range(3):
            a = int(rng.integers(0, n - 60))
            seq[a:a + int(rng.integers(1, 60))] = "N"
        a = int(rng.integers(0, n - 200))
        seq[a:a + 150] = np.char.lower(seq[a:a + 150])
        genome[chrom] = "".join(seq)
    _FakeFasta.genomes["mem://tiled"] = genome
    sys.modules["pysam"].FastaFile = _FakeFasta
    sys.modules["pysam"].TabixFile = _FakeTabix
    window = 500
    idx = [(1, s, s + window) for s in range(0, 2600, window)] + [(2, s, s + window) for s in range(0, 1437, window)]
    ctx = ["".join(t) for t in __import__("itertools").product("ACGT", repeat=3)]
    out = dict(genome=genome, idx=[list(map(int, r)) for r in idx], window=window, contexts=ctx, cohorts=[])
    for c in range(2):
        d_pr = dict(zip(ctx, (rng.dirichlet(np.ones(64)) * 1e-2).tolist()))
        rows = []
        for chrom, n in (("1", 2600), ("2", 1437)):
            for _ in range(90 + 40 * c):
                p = int(rng.integers(0, n))
                rows.append((chrom, p, p + 1, "A", "T", "S%d" % rng.integers(0, 6)))
            for _ in range(6):                                  # positions hit more than once, and an indel-like longer row
                p = int(rng.integers(0, n - 5))
                rows += [(chrom, p, p + 1, "C", "G", "S1"), (chrom, p, p + 1, "C", "G", "S2"), (chrom, p, p + 4, "CAGT", "C", "S3")]
        rows.sort()
        _FakeTabix.tables["mem://muts%d" % c] = rows
        mu = rng.gamma(9.0, 3.0, len(idx))
        sigma = rng.gamma(4.0, 1.0, len(idx))
        coh = dict(d_pr=d_pr, rows=[list(r) for r in rows], mu=mu.tolist(), sigma=sigma.tolist(), runs={})
        for binsize in (1, 50):
            df = ref_nb.nb_model(d_pr, idx, mu, sigma, "mem://muts%d" % c, "mem://tiled", n_up=1, n_down=1, binsize=binsize)
            coh["runs"][str(binsize)] = {k: df[k].astype(float).tolist() for k in ["CHROM", "POS", "OBS", "EXP", "PVAL", "Pi"]}
            coh["runs"][str(binsize)]["REGION_first_last"] = [df["REGION"].iloc[0], df["REGION"].iloc[-1]]
        out["cohorts"].append(coh)
    import gzip
    with gzip.GzipFile(os.path.join(HERE, "tiled_golden.json.gz"), "wb", mtime=0) as f:
        f.write(json.dumps(out).encode())
    print("wrote tiled_golden.json.gz", {b: len(out["cohorts"][0]["runs"][b]["PVAL"]) for b in ("1", "50")})


def gen_tiled_penta():
    """The per-base route in the reference's DEFAULT mode -- penta-nucleotide contexts, n_up = n_down = 2, the default
    arguments of nb_model / apply_nb_to_region / base_probabilities_by_region (nb_model.py:126,188; sequence_tools.py:292):
    the reference's own nb_model, called without n_up / n_down, on a small genome (N runs, a soft-masked stretch, a bin at
    position 0, one cut off by the chromosome end), tiles of 50 (the default) and of 1 position, two cohorts."""
    import gzip
    import itertools
    rng = np.random.default_rng(188)
    genome = {}
    for chrom, n in (("chr1", 2100), ("chr2", 1237)):
        seq = rng.choice(list("ACGT"), n)
        for _ in range(3):
            a = int(rng.integers(0, n - 60))
            seq[a:a + int(rng.integers(1, 40))] = "N"
        a = int(rng.integers(0, n - 200))
        seq[a:a + 120] = np.char.lower(seq[a:a + 120])
        genome[chrom] = "".join(seq)
    _FakeFasta.genomes["mem://penta"] = genome
    sys.modules["pysam"].FastaFile = _FakeFasta
    sys.modules["pysam"].TabixFile = _FakeTabix
    window = 500
    idx = [(1, s, s + window) for s in range(0, 2100, window)] + [(2, s, s + window) for s in range(0, 1237, window)]
    ctx = ["".join(t) for t in itertools.product("ACGT", repeat=5)]
    out = dict(genome=genome, idx=[list(map(int, r)) for r in idx], window=window, cohorts=[])
    for c in range(2):
        d_pr = dict(zip(ctx, (rng.dirichlet(np.ones(1024)) * 1e-2).tolist()))
        rows = []
        for chrom, n in (("1", 2100), ("2", 1237)):
            for _ in range(80 + 30 * c):
                p_ = int(rng.integers(0, n))
                rows.append((chrom, p_, p_ + 1, "A", "T", "S%d" % rng.integers(0, 6)))
            for _ in range(5):
                p_ = int(rng.integers(0, n - 5))
                rows += [(chrom, p_, p_ + 1, "C", "G", "S1"), (chrom, p_, p_ + 1, "C", "G", "S2")]
        rows.sort()
        _FakeTabix.tables["mem://pmuts%d" % c] = rows
        mu = rng.gamma(9.0, 3.0, len(idx))
        sigma = rng.gamma(4.0, 1.0, len(idx))
        coh = dict(d_pr=[d_pr[k] for k in ctx], rows=[list(r) for r in rows], mu=mu.tolist(), sigma=sigma.tolist(), runs={})
        df = ref_nb.nb_model(d_pr, idx, mu, sigma, "mem://pmuts%d" % c, "mem://penta")          # every default: penta, binsize 50
        df1 = ref_nb.nb_model(d_pr, idx, mu, sigma, "mem://pmuts%d" % c, "mem://penta", binsize=1)
        for tag, frame in (("50", df), ("1", df1)):
            coh["runs"][tag] = {k: frame[k].astype(float).tolist() for k in ["CHROM", "POS", "OBS", "EXP", "PVAL", "Pi"]}
            coh["runs"][tag]["REGION_first_last"] = [frame["REGION"].iloc[0], frame["REGION"].iloc[-1]]
        out["cohorts"].append(coh)
    with gzip.GzipFile(os.path.join(HERE, "tiled_penta_golden.json.gz"), "wb", mtime=0) as f:
        f.write(json.dumps(out).encode())
    print("wrote tiled_penta_golden.json.gz", {b: len(out["cohorts"][0]["runs"][b]["PVAL"]) for b in ("1", "50")})


def gen_sites():
    """The sites route (mutation_tools.py:232-283, sequence_tools.py:643-700): the reference's own preprocess_sites and
    tabulate_sites_in_element on a small synthetic sites file."""
    import tempfile
    rng = np.random.default_rng(77)
    window = 1000
    subst = sorted(ref_seq.mk_trans_idx(n_up=1, n_down=1, collapse=False))
    muts192 = ref_seq.mk_mutation_context(n_up=1, n_down=1, collapse=False, return_df=True) \
        if "return_df" in ref_seq.mk_mutation_context.__code__.co_varnames else None
    # sequence model frame (MUT_TYPE, CONTEXT) in the reference's own row order
    keys = list(ref_seq.mk_mutation_context(n_up=1, n_down=1, collapse=False).keys())
    df_seq = pd.DataFrame({"MUT_TYPE": [k[0] for k in keys], "CONTEXT": [k[1] for k in keys], "FREQ": rng.uniform(size=len(keys))})
    _HDF_FRAMES[("mem://pre_sites", "sequence_model_192")] = df_seq
    nb = {1: 30, 2: 20}
    idx = np.array([(c, b * window, (b + 1) * window) for c, n in nb.items() for b in range(n)])
    si = rng.integers(0, 50, (len(idx), 64))
    _H5_FILES["mem://sites_data"] = {"window_%d" % window: {"full_window_si_index": idx, "full_window_si_values": si}}
    rows = []
    elts = [("siteA", 1, "+"), ("siteB", 1, "-"), ("siteC", 2, "+"), ("siteD", 2, "-"), ("siteE", 1, "+")]
    for name, chrom, strand in elts:
        for _ in range(int(rng.integers(2, 9))):
            p = int(rng.integers(0, nb[chrom] * window - 1))
            k = keys[int(rng.integers(0, len(keys)))]
            ref_, alt_ = k[0][0], k[0][2]
            rows.append((chrom, p, p + 1, ref_, alt_, name, "G_" + name, "Noncoding", k[0], k[1], strand))
    rows.append((1, 999, 1000, "A", "T", "siteE", "G_siteE", "Noncoding", "A>T", "nan", "+"))     # no context
    f_sites = os.path.join(tempfile.mkdtemp(), "sites.tsv")
    pd.DataFrame(rows).to_csv(f_sites, sep="\t", header=False, index=False)
    ref_seq.preprocess_sites(f_sites, "mem://sites_data", "mem://pre_sites", "mysites", window)
    grp = _H5_FILES["mem://sites_data"]["window_%d" % window]["mysites"]
    out = {"window": window, "sites_rows": [list(map(str, r)) for r in rows], "bin_idx": idx.tolist(), "bin_ctx": si.tolist(),
           "seq_mut_type": df_seq.MUT_TYPE.tolist(), "seq_context": df_seq.CONTEXT.tolist(), "subst_sorted": subst, "elements": {}}
    for name in sorted(k for k in grp.keys() if k != "__attrs__"):
        node = grp[name]
        out["elements"][name] = {"L_counts": np.asarray(node["L_counts"]).astype(int).tolist(),
                                 "region_counts": np.asarray(node["region_counts"]).astype(int).tolist(),
                                 "overlaps": [list(map(int, o)) for o in node["__attrs__"]["overlaps"]]}
    # mutations: some exactly at the sites (same nine columns), some not
    mrows = []
    for r in rows[:-1]:
        for _ in range(int(rng.integers(0, 4))):
            mrows.append((r[0], r[1], r[2], r[3], r[4], "S%d" % rng.integers(0, 6), r[6], r[7], r[8], r[9]))
    mrows.append((1, 5, 6, "A", "T", "S1", "X", "Noncoding", "A>T", "CAG"))
    mrows.append((1, 7, 9, "AG", "A", "S2", "X", "INDEL", "DEL", "."))
    f_mut = os.path.join(os.path.dirname(f_sites), "m.tsv")
    pd.DataFrame(mrows).to_csv(f_mut, sep="\t", header=False, index=False)
    tab = ref_mt.tabulate_sites_in_element(f_sites, f_mut)
    out["mut_rows"] = [list(map(str, r)) for r in mrows]
    out["tab_index"] = [str(i) for i in tab.index]
    out["tab_obs_samples"] = tab.OBS_SAMPLES.astype(int).tolist()
    out["tab_obs_snv"] = tab.OBS_SNV.astype(int).tolist()
    with open(os.path.join(HERE, "sites_golden.json"), "w") as f:
        json.dump(out, f)
    print("wrote sites_golden.json", len(out["elements"]), len(tab))


# ----------------------------------------------------------------------------
# (xx) mutation x element tabulation (mutation_tools.py:155-230)
# ----------------------------------------------------------------------------
class _StandInBedTool:
    """Stand-in for pybedtools.BedTool -- pybedtools and the bedtools binary are absent from this image.  It holds the
    rows of a tab-separated file and offers exactly what tabulate_muts_per_sample_per_element calls.  THE ONLY PART OF
    THE ROUTE THAT IS NOT THE REFERENCE'S OWN CODE IS `intersect`: a brute-force restatement of `bedtools intersect
    -wa -wb` (same chromosome label as text, half-open overlap a.start < b.end and b.start < a.end, one output row =
    the A fields followed by the B fields, A in file order).  Everything after the join -- the duplicate drop, the
    SNV / INDEL split, the (element, sample) counts, the blacklist, the caps, the per-element summary -- is executed by
    the reference's functions unmodified.  Inputs hold no zero-length records (bedtools' handling of those cannot be
    checked here)."""

    def __init__(self, src):
        if isinstance(src, str):
            with open(src) as f:
                self.rows = [ln.rstrip("\n").split("\t") for ln in f if ln.strip()]
        else:
            self.rows = [list(r) for r in src]

    def bed12tobed6(self):
        out = []
        for r in self.rows:
            start = int(r[1])
            sizes = [int(x) for x in r[10].split(",") if x != ""]
            rel = [int(x) for x in r[11].split(",") if x != ""]
            for s, z in zip(rel, sizes):
                out.append([r[0], str(start + s), str(start + s + z), r[3], r[4], r[5]])
        return _StandInBedTool(out)

    def intersect(self, other, wa=False, wb=False, **kw):
        assert wa and wb and not kw
        out = []
        for a in self.rows:
            a_s, a_e = int(a[1]), int(a[2])
            assert a_e > a_s
            for b in other.rows:
                b_s, b_e = int(b[1]), int(b[2])
                assert b_e > b_s
                if a[0] == b[0] and a_s < b_e and b_s < a_e:
                    out.append(a + b)
        return _StandInBedTool(out)

    def __len__(self):
        return len(self.rows)

    def field_count(self):
        return len(self.rows[0])

    def to_dataframe(self, **kw):
        # pybedtools: pandas.read_csv(self.fn, sep="\t", **kw)
        text = "".join("\t".join(r) + "\n" for r in self.rows)
        return pd.read_csv(io.StringIO(text), sep="\t", **kw)


def gen_tabulate():
    """tabulate_muts_per_sample_per_element and tabulate_mutations_in_element run by the reference itself on a seeded
    mutation file and a bed12 file (blocks of different elements overlap and nest; indels span two blocks of one
    element; mutations annotated twice; an X chromosome), with `pybedtools.BedTool` = _StandInBedTool above."""
    import gzip
    import tempfile
    import warnings
    warnings.simplefilter("ignore")
    rng = np.random.default_rng(191)
    sys.modules["pybedtools"].BedTool = _StandInBedTool
    ref_mt.pybedtools = sys.modules["pybedtools"]
    chroms = ["1", "2", "3", "4", "X"]
    bed, spans = [], []
    for i in range(160):
        c = chroms[int(rng.integers(0, len(chroms)))] if i % 10 else "X"
        nb = int(rng.integers(1, 5))
        if i % 7 == 3 and spans:                                   # nested inside / overlapping an earlier element
            c, s0, e0 = spans[int(rng.integers(0, len(spans)))]
            start = int(rng.integers(s0, max(s0 + 1, e0 - 50)))
        else:
            start = int(rng.integers(1000, 60000))
        sizes = rng.integers(20, 400, nb)
        gaps = rng.integers(1, 120, nb)
        rel = np.concatenate([[0], np.cumsum(sizes + gaps)[:-1]])
        end = start + int(rel[-1] + sizes[-1])
        spans.append((c, start, end))
        trail = "," if i % 3 == 0 else ""
        bed.append([c, start, end, "E%03d" % i, 0, "+-"[i % 2], start, start, ".", nb,
                    ",".join(map(str, sizes)) + trail, ",".join(map(str, rel)) + trail])
    blocks = [(r[0], r[1] + int(x), r[1] + int(x) + int(z)) for r in bed
              for x, z in zip(str(r[11]).rstrip(",").split(","), str(r[10]).rstrip(",").split(","))]
    rows = []
    for _ in range(3500):
        c, s, e = blocks[int(rng.integers(0, len(blocks)))]
        p = int(rng.integers(max(s - 40, 0), e + 40))
        smp = "S%02d" % int(min(rng.geometric(0.12), 30))            # a few heavy samples: the blacklist has work to do
        u = rng.uniform()
        if u < 0.12:                                                 # indels, some long enough to span a gap between blocks
            ln = int(rng.integers(2, 140))
            rows.append([c, p, p + ln, "A" * min(ln, 5), "A", smp, "G%d" % rng.integers(0, 9), "INDEL", "DEL", "."])
        else:
            rows.append([c, p, p + 1, "ACGT"[int(rng.integers(0, 4))], "ACGT"[int(rng.integers(0, 4))], smp,
                         "G%d" % rng.integers(0, 9), ["Noncoding", "Missense", "Synonymous"][int(rng.integers(0, 3))], "A>T", "CAG"])
    for i in rng.integers(0, len(rows), 250):                        # same mutation under a second gene annotation
        r = list(rows[int(i)]); r[6] = "H" + r[6]; rows.append(r)
    for i in rng.integers(0, len(rows), 60):                         # exact duplicate rows
        rows.append(list(rows[int(i)]))
    rows.append(["7", 5, 6, "A", "T", "S01", ".", "Noncoding", "A>T", "CAG"])      # chromosome without elements
    tmp = tempfile.mkdtemp()
    f_mut, f_bed = os.path.join(tmp, "m.tsv"), os.path.join(tmp, "e.bed")
    pd.DataFrame(rows).to_csv(f_mut, sep="\t", header=False, index=False)
    pd.DataFrame(bed).to_csv(f_bed, sep="\t", header=False, index=False)
    out = {"note": "outputs of the reference's tabulate_* functions; only pybedtools' intersect was replaced by a "
                   "brute-force half-open overlap join (see _StandInBedTool in make_golden.py)",
           "mut_rows": [list(map(str, r)) for r in rows], "bed_rows": [list(map(str, r)) for r in bed], "cases": []}
    for dd in (False, True):
        cnt = ref_mt.tabulate_muts_per_sample_per_element(f_mut, f_bed, bed12=True, drop_duplicates=dd)
        out["per_pair_dedup" if dd else "per_pair"] = dict(
            ELT=cnt.ELT.tolist(), SAMPLE=cnt.SAMPLE.tolist(), OBS_SNV=cnt.OBS_SNV.astype(int).tolist(),
            OBS_INDEL=cnt.OBS_INDEL.astype(int).tolist(), OBS_MUT=cnt.OBS_MUT.astype(int).tolist())
        for caps in ((1e9, 3e9), (60, 2), (25, 1)):
            for all_elements in (False, True):
                tab, black = ref_mt.tabulate_mutations_in_element(
                    f_mut, f_bed, bed12=True, drop_duplicates=dd, all_elements=all_elements,
                    max_muts_per_sample=caps[0], max_muts_per_elt_per_sample=caps[1], return_blacklist=True)
                out["cases"].append(dict(
                    drop_duplicates=dd, max_muts_per_sample=caps[0], max_muts_per_elt_per_sample=caps[1],
                    all_elements=all_elements, index=[str(i) for i in tab.index], columns=list(tab.columns),
                    OBS_SAMPLES=tab.OBS_SAMPLES.astype(int).tolist(), OBS_SNV=tab.OBS_SNV.astype(int).tolist(),
                    OBS_INDEL=tab.OBS_INDEL.astype(int).tolist(), blacklist=sorted(str(b) for b in black)))
    # the empty intersection (mutation_tools.py:201-202)
    f_none = os.path.join(tmp, "none.tsv")
    pd.DataFrame(rows[-1:]).to_csv(f_none, sep="\t", header=False, index=False)
    empty = ref_mt.tabulate_mutations_in_element(f_none, f_bed, bed12=True)
    out["empty_columns"], out["empty_len"] = list(empty.columns), int(len(empty))
    with gzip.GzipFile(os.path.join(HERE, "tabulate_golden.json.gz"), "wb", mtime=0) as f:
        f.write(json.dumps(out).encode())
    print("wrote tabulate_golden.json.gz", len(rows), "mutations", len(bed), "elements",
          len(out["per_pair"]["ELT"]), "pairs", [len(c["blacklist"]) for c in out["cases"]])


# ----------------------------------------------------------------------------
# (xxi) pyrimidine-collapsed contexts (collapse=True: K = 96 substitution types)
# ----------------------------------------------------------------------------
def gen_collapse():
    """The collapse=True branch of the per-base route and of the context counter (sequence_tools.py:31-55,65-99,232-262,
    292-317; nb_model.py:126-234): no live caller of the reference passes it, the north star names it ("96-trinucleotide-
    context").  The reference's own count_contexts_by_regions(collapse=True), mk_mutation_context / mk_trans_idx
    (collapse=True) and nb_model(..., n_up=1, n_down=1, collapse=True) and (penta, collapse=True) on a small genome."""
    import gzip
    import itertools
    rng = np.random.default_rng(96)
    genome = {}
    for chrom, n in (("chr1", 1500), ("chr2", 777)):
        seq = rng.choice(list("ACGT"), n)
        a = int(rng.integers(0, n - 60))
        seq[a:a + 25] = "N"
        genome[chrom] = "".join(seq)
    _FakeFasta.genomes["mem://collapse"] = genome
    sys.modules["pysam"].FastaFile = _FakeFasta
    sys.modules["pysam"].TabixFile = _FakeTabix
    window = 500
    idx = [(1, s, s + window) for s in range(0, 1500, window)] + [(2, s, s + window) for s in range(0, 777, window)]
    out = dict(genome=genome, idx=[list(map(int, r)) for r in idx])
    out["mutation_context_96"] = [list(k) for k in ref_seq.mk_mutation_context(n_up=1, n_down=1, collapse=True).keys()]
    out["trans_idx_96"] = ref_seq.mk_trans_idx(n_up=1, n_down=1, collapse=True)
    regs = [("chr1", 0, 700), ("chr1", 650, 1500), ("chr2", 3, 5), ("chr2", 100, 900)]
    cc = ref_seq.count_contexts_by_regions("mem://collapse", [r[0] for r in regs], [r[1] for r in regs], [r[2] for r in regs],
                                           n_up=1, n_down=1, collapse=True)
    out["count_regions"] = [list(r) for r in regs]
    out["count_columns"] = list(cc.columns)
    out["count_index"] = list(cc.index)
    out["count_values"] = cc.values.astype(int).tolist()
    rows = []
    for chrom, n in (("1", 1500), ("2", 777)):
        for _ in range(90):
            p_ = int(rng.integers(0, n))
            rows.append((chrom, p_, p_ + 1, "A", "T", "S%d" % rng.integers(0, 6)))
    rows.sort()
    _FakeTabix.tables["mem://cmuts"] = rows
    out["rows"] = [list(r) for r in rows]
    mu = rng.gamma(9.0, 3.0, len(idx))
    sigma = rng.gamma(4.0, 1.0, len(idx))
    out["mu"], out["sigma"], out["runs"] = mu.tolist(), sigma.tolist(), {}
    for n_up in (1, 2):
        keys = list(ref_seq.mk_context_sequences(n_up=n_up, n_down=n_up, collapse=True).keys())
        d_pr = dict(zip(keys, (rng.dirichlet(np.ones(len(keys))) * 1e-2).tolist()))
        df = ref_nb.nb_model(d_pr, idx, mu, sigma, "mem://cmuts", "mem://collapse", n_up=n_up, n_down=n_up, binsize=25, collapse=True)
        out["runs"][str(n_up)] = dict(keys=keys, d_pr=[d_pr[k] for k in keys],
                                      **{k: df[k].astype(float).tolist() for k in ["CHROM", "POS", "OBS", "EXP", "PVAL", "Pi"]})
    with gzip.GzipFile(os.path.join(HERE, "collapse_golden.json.gz"), "wb", mtime=0) as f:
        f.write(json.dumps(out).encode())
    print("wrote collapse_golden.json.gz", len(out["trans_idx_96"]), {k: len(v["PVAL"]) for k, v in out["runs"].items()})


def main():
    if "--only-collapse" in sys.argv:
        gen_collapse()
        return
    if "--only-zero-den" in sys.argv:
        subst_idx, ctx64, df_empty = gen_subst_index()
        gen_accumulate_zero_den(subst_idx, df_empty)
        return
    if "--only-tabulate" in sys.argv:
        gen_tabulate()
        return
    if "--only-contexts" in sys.argv:
        gen_contexts()
        return
    if "--only-sites" in sys.argv:
        gen_sites()
        return
    if "--only-tiled-penta" in sys.argv:
        gen_tiled_penta()
        return
    if "--only-tiled" in sys.argv:
        gen_tiled()
        return
    if "--only-run-element" in sys.argv:
        gen_run_element_expectation()
        return
    if "--only-cnn-735" in sys.argv:
        gen_cnn_735()
        return
    if "--only-run-target" in sys.argv:
        gen_run_target()
        return
    if "--only-nn-training" in sys.argv:
        gen_nn_training()
        return
    gen_nb_midp()
    gen_nb_exact()
    gen_element_stats()
    gen_gene_stats()
    gen_overlaps()
    subst_idx, ctx64, df_empty = gen_subst_index()
    gen_accumulate(subst_idx, ctx64, df_empty)
    gen_accumulate_zero_den(subst_idx, df_empty)
    gen_mutation_tools()
    gen_sequence_model(df_empty)
    gen_cnn()
    gen_cnn_735()
    gen_nn_training()
    gen_contexts()
    gen_sites()
    gen_tiled()
    gen_tiled_penta()
    gen_run_element_expectation()
    gen_run_target()
    gen_tabulate()
    gen_collapse()
    import torch
    with open(os.path.join(HERE, "MANIFEST.json"), "w") as f:
        json.dump(dict(generator="tests/golden/make_golden.py", reference=REF,
                       numpy=np.__version__, scipy=scipy.__version__, pandas=pd.__version__,
                       torch=torch.__version__, python=sys.version.split()[0]), f, indent=1)


if __name__ == "__main__":
    main()
