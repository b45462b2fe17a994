"""Negative-binomial burden tests -- host mirror of DIGDriver/sequence_model/nb_model.py.

Same names, argument meaning and NaN behaviour as the reference functions; the
arithmetic runs in the HIP kernels of libdig_hip.so (dig_nb.hip / dig_math.hpp).
Inputs may be scalars, numpy arrays, pandas Series (host path: staged through
the *_host entry points) or torch CUDA tensors (device path: no copies, enqueued
on torch's current stream).  There is no CPU fallback.
"""
import numpy as np

from .. import _lib


def _is_cuda_tensor(x):
    return type(x).__module__.startswith("torch") and getattr(x, "is_cuda", False)


def _series_like(*xs):
    for x in xs:
        if type(x).__name__ == "Series":
            return x
    return None


def _vec3(name, k, alpha, p, device=0):
    if any(_is_cuda_tensor(x) for x in (k, alpha, p)):
        import torch
        dev = next(x.device for x in (k, alpha, p) if _is_cuda_tensor(x))
        k, alpha, p = torch.broadcast_tensors(*[torch.as_tensor(x, dtype=torch.float64, device=dev) for x in (k, alpha, p)])
        k, alpha, p = k.contiguous(), alpha.contiguous(), p.contiguous()
        out = torch.empty_like(k)
        with torch.cuda.device(dev):
            _lib.call(name, _lib.dev_ptr(k), _lib.dev_ptr(alpha), _lib.dev_ptr(p), _lib.dev_ptr(out), k.numel(),
                      _lib.stream_ptr())
        return out
    ser = _series_like(k, alpha, p)
    kb, ab, pb = np.broadcast_arrays(np.asarray(k, dtype=np.float64), np.asarray(alpha, dtype=np.float64),
                                     np.asarray(p, dtype=np.float64))
    shape = kb.shape
    kc, ac, pc = (_lib.as_host(v, np.float64).ravel() for v in (kb, ab, pb))
    out = np.empty(kc.shape, np.float64)
    _lib.call(name + "_host", _lib.host_ptr(kc), _lib.host_ptr(ac), _lib.host_ptr(pc), _lib.host_ptr(out), kc.size,
              device)
    out = out.reshape(shape)
    if ser is not None:
        import pandas as pd
        return pd.Series(out, index=ser.index)
    return out if shape else float(out)


def normal_params_to_gamma(mu, sigma, device=0):
    """nb_model.py:237-241 -- alpha = mu**2 / sigma**2, theta = sigma**2 / mu."""
    if _is_cuda_tensor(mu) or _is_cuda_tensor(sigma):
        import torch
        dev = mu.device if _is_cuda_tensor(mu) else sigma.device
        mu, sigma = torch.broadcast_tensors(*[torch.as_tensor(x, dtype=torch.float64, device=dev) for x in (mu, sigma)])
        mu, sigma = mu.contiguous(), sigma.contiguous()
        alpha, theta = torch.empty_like(mu), torch.empty_like(mu)
        with torch.cuda.device(dev):
            _lib.call("dig_normal_params_to_gamma", _lib.dev_ptr(mu), _lib.dev_ptr(sigma), _lib.dev_ptr(alpha),
                      _lib.dev_ptr(theta), mu.numel(), _lib.stream_ptr())
        return alpha, theta
    ser = _series_like(mu, sigma)
    mb, sb = np.broadcast_arrays(np.asarray(mu, dtype=np.float64), np.asarray(sigma, dtype=np.float64))
    shape = mb.shape
    mc, sc = _lib.as_host(mb, np.float64).ravel(), _lib.as_host(sb, np.float64).ravel()
    alpha, theta = np.empty(mc.shape), np.empty(mc.shape)
    _lib.call("dig_normal_params_to_gamma_host", _lib.host_ptr(mc), _lib.host_ptr(sc), _lib.host_ptr(alpha),
              _lib.host_ptr(theta), mc.size, device)
    alpha, theta = alpha.reshape(shape), theta.reshape(shape)
    if ser is not None:
        import pandas as pd
        return pd.Series(alpha, index=ser.index), pd.Series(theta, index=ser.index)
    if not shape:
        return float(alpha), float(theta)
    return alpha, theta


def nb_pvalue_greater_midp(k, alpha, p, device=0):
    """UPPER TAIL NB p-value with mid-p correction (nb_model.py:271-278)."""
    return _vec3("dig_nb_midp_upper", k, alpha, p, device)


def nb_pvalue_greater(k, alpha, p, device=0):
    """UPPER TAIL NB p-value (nb_model.py:243-256); vectorised over the reference's scalar form."""
    return _vec3("dig_nb_greater", k, alpha, p, device)


def nb_pvalue_exact(k, alpha, p, mu=None, device=0):
    """Upper or lower tail depending on k vs the expectation (nb_model.py:298-314).
    `mu` other than None/0 is not supported (no live caller passes it)."""
    if mu:
        raise NotImplementedError("explicit mu is not supported; the reference's callers never pass it")
    return _vec3("dig_nb_exact", k, alpha, p, device)


def nb_pvalue_midp(k, alpha, p, mu=None, device=0):
    """Two-sided-by-side mid-p variant (nb_model.py:316-337)."""
    if mu:
        raise NotImplementedError("explicit mu is not supported; the reference's callers never pass it")
    return _vec3("dig_nb_midp_twosided", k, alpha, p, device)


def fisher_combine(p1, p2, device=0):
    """chi2.sf(-2 (ln p1 + ln p2), df=4) (transfer_tools.py:860-861,1086-1087)."""
    if _is_cuda_tensor(p1) or _is_cuda_tensor(p2):
        import torch
        dev = p1.device if _is_cuda_tensor(p1) else p2.device
        p1, p2 = torch.broadcast_tensors(*[torch.as_tensor(x, dtype=torch.float64, device=dev) for x in (p1, p2)])
        p1, p2 = p1.contiguous(), p2.contiguous()
        out = torch.empty_like(p1)
        with torch.cuda.device(dev):
            _lib.call("dig_fisher", _lib.dev_ptr(p1), _lib.dev_ptr(p2), _lib.dev_ptr(out), p1.numel(), _lib.stream_ptr())
        return out
    ser = _series_like(p1, p2)
    a, b = np.broadcast_arrays(np.asarray(p1, dtype=np.float64), np.asarray(p2, dtype=np.float64))
    shape = a.shape
    ac, bc = _lib.as_host(a, np.float64).ravel(), _lib.as_host(b, np.float64).ravel()
    out = np.empty(ac.shape)
    _lib.call("dig_fisher_host", _lib.host_ptr(ac), _lib.host_ptr(bc), _lib.host_ptr(out), ac.size, device)
    out = out.reshape(shape)
    if ser is not None:
        import pandas as pd
        return pd.Series(out, index=ser.index)
    return out if shape else float(out)


# ---------------------------------------------------------------------------------------------
# per-base / tiled route (nb_model.py:126-234, 340-342)
# ---------------------------------------------------------------------------------------------
def _s_prob_table(d_pr, n_up=1, collapse=False):
    """S_prob (dict / Series keyed by the (2 n_up + 1)-mer) -> 4^(2 n_up + 1) values in context index order: 64 for the
    trinucleotide models of the live pipeline, 1 024 for the penta-nucleotide default of the reference's signatures.
    collapse=True (pyrimidine-collapsed contexts: K = 96 substitution types, 32 / 512 contexts): the table holds the C- and
    T-centred windows only; a window centred on A or G is looked up as its reverse complement (seq_to_context,
    sequence_tools.py:42-55) -- expanded here into the full table, so that the kernels look every window up directly."""
    import itertools
    from . import sequence_tools
    keys = ["".join(t) for t in itertools.product("ACGT", repeat=2 * n_up + 1)]
    if collapse:
        keys = [k if k[n_up] in "CT" else sequence_tools.reverse_complement(k) for k in keys]
    try:
        return np.array([float(d_pr[k]) for k in keys])
    except KeyError as exc:
        raise KeyError("S_prob has no entry for context %s (n_up = n_down = %d%s needs all %s%d-mers)" % (
            exc, n_up, ", collapse=True" if collapse else "", "C- and T-centred " if collapse else "", 2 * n_up + 1)) from exc


def _mutation_rows(f_mut):
    """The rows a tabix fetch hands to tabix_to_dataframe (nb_model.py:13-33): CHROM, START, END of a bed-like mutation
    file (plain or gzip; 6-9 columns)."""
    import pandas as pd
    df = pd.read_csv(f_mut, sep="\t", header=None, usecols=[0, 1, 2], names=["CHROM", "START", "END"], dtype={0: str},
                     comment="#", low_memory=False)
    df["CHROM"] = df.CHROM.str.replace("chr", "", regex=False)
    return df


def nb_model(d_pr, idx, mu_lst, sigma_lst, f_tabix, f_fasta, n_up=2, n_down=2, binsize=50, collapse=False, device=0):
    """nb_model.py:188-234: the tiled NB test over the bins `idx` [(chrom, start, end)] of one cohort; returns the
    reference's frame (CHROM, POS, OBS, EXP, PVAL, Pi, MU, SIGMA, REGION; numeric columns as float, as its np.hstack makes
    them).  `f_tabix`: the cohort's bed-like mutation file (the reference reads it through tabix; here it is joined on the
    GPU in one pass); `f_fasta`: the genome (data_tools.genome.PackedGenome or a FASTA path).  All bins in three launches
    (engine.tiled_nb_model).  n_up = n_down = 2 (penta-nucleotide contexts, the reference's default) or 1 (the trinucleotide
    models every live part of the pipeline trains); collapse=True: S_prob keyed by the pyrimidine-centred contexts (the
    96-substitution model), windows centred on a purine looked up as their reverse complement."""
    import pandas as pd
    from .. import engine
    from ..data_tools import genome as genome_mod
    if n_up != n_down or n_up not in (1, 2):
        raise NotImplementedError("the tile kernels take n_up = n_down = 1 or 2")
    g = f_fasta if isinstance(f_fasta, genome_mod.PackedGenome) else genome_mod.PackedGenome.from_fasta(f_fasta)
    idx = np.asarray(idx)
    chroms = [str(c) for c in idx[:, 0]]
    starts, ends = idx[:, 1].astype(np.int64), idx[:, 2].astype(np.int64)
    muts = f_tabix if isinstance(f_tabix, pd.DataFrame) else _mutation_rows(f_tabix)
    known = set(n.replace("chr", "") for n in g.names)
    muts = muts[muts.CHROM.astype(str).str.replace("chr", "", regex=False).isin(known)]
    res = engine.tiled_nb_model(g, chroms, starts, ends, _s_prob_table(d_pr, n_up, collapse)[None, :], np.asarray(mu_lst, float)[None, :],
                                np.asarray(sigma_lst, float)[None, :], muts.CHROM.astype(str).values, muts.START.values,
                                muts.END.values, np.zeros(len(muts), np.int32), binsize=binsize, device=device)
    host = {k: v.cpu().numpy() for k, v in res.items()}
    first, nval = host["first_pos"], host["n_valid"].astype(np.int64)
    n_pos = np.minimum(ends, np.array([g.lengths[i] for i in g.chrom_index(chroms)]) - n_up) - first
    cols = ["CHROM", "POS", "OBS", "EXP", "PVAL", "Pi", "MU", "SIGMA", "REGION"]
    if len(idx) == 0 or nval.sum() == 0:
        return pd.DataFrame(columns=cols)
    # all regions at once (the reference appends one block per region): region of every tile, tile number inside it
    reg = np.repeat(np.arange(len(idx)), nval)
    t = np.arange(nval.sum()) - np.repeat(np.cumsum(nval) - nval, nval)
    lo = first[reg] + t * binsize
    hi = np.minimum(lo + binsize, (first + n_pos)[reg]) - 1
    take = lambda a: a[0][reg, t]
    labels = np.array(["{}:{}-{}".format(c, s_, e) for c, s_, e in idx], dtype=object)
    return pd.DataFrame({
        "CHROM": idx[reg, 0].astype(float), "POS": (lo + hi) / 2.0 if binsize > 1 else lo.astype(float),
        "OBS": take(host["k"]).astype(float), "EXP": take(host["exp"]), "PVAL": take(host["pval"]), "Pi": take(host["pt"]),
        "MU": np.asarray(mu_lst, float)[reg], "SIGMA": np.asarray(sigma_lst, float)[reg], "REGION": labels[reg]})[cols]


def bh_ragged(p, row_ptr, n_global=None, rank0=None, carry=None, want_q=True, want_row_min=False, sorted_out=False):
    """dig_bh_qvalues_ragged (csrc/dig_sort.hip): Benjamini-Hochberg q-values of ragged rows of one float64 device tensor -- the
    library's own batched radix sort (63-bit keys, 32-bit payload, one kernel per pass), the Benjamini-Hochberg pass and the
    way back to every p-value's place in one launch sequence.  row_ptr: host offsets (rows + 1).  n_global / rank0 / carry (host
    arrays or None): the rows are ranges of longer lists (parallel.ShardedTiles' sample sort).  Returns (q or None, row_min or None)."""
    import torch
    from .. import _lib
    p = p.contiguous()
    assert p.dtype == torch.float64 and p.is_cuda
    rp = np.ascontiguousarray(row_ptr, dtype=np.int64)
    rows = rp.size - 1
    h = lambda a, dt: None if a is None else np.ascontiguousarray(a, dtype=dt)
    ng, r0, ca = h(n_global, np.float64), h(rank0, np.int64), h(carry, np.float64)
    q = torch.empty_like(p) if want_q else None
    rmin = torch.empty(max(rows, 1), dtype=torch.float64, device=p.device) if want_row_min else None
    wsb = int(_lib.load().dig_bh_ragged_workspace(_lib.host_ptr(rp), rows))
    ws = torch.empty(wsb, dtype=torch.uint8, device=p.device)
    with torch.cuda.device(p.device):
        _lib.call("dig_bh_qvalues_ragged", _lib.dev_ptr(p), _lib.host_ptr(rp), rows, _lib.host_ptr(ng), _lib.host_ptr(r0), _lib.host_ptr(ca),
                  _lib.dev_ptr(q) if q is not None else None, _lib.dev_ptr(rmin) if rmin is not None else None, 1 if sorted_out else 0,
                  _lib.dev_ptr(ws), wsb, _lib.stream_ptr())
    return q, (rmin[:rows] if rmin is not None else None)


def get_q_vals_rows(p_rows):
    """get_q_vals for every row of a [rows, n] device tensor at once (the cohorts of the per-base route): one call of
    dig_bh_qvalues_ragged -- the library's radix sort of all rows, the Benjamini-Hochberg pass and the scatter behind it."""
    import torch
    p = p_rows.to(torch.float64).contiguous()
    rows, n = p.shape
    if n == 0 or rows == 0:
        return p.clone()
    q, _ = bh_ragged(p.reshape(-1), np.arange(rows + 1, dtype=np.int64) * n)
    return q.reshape(rows, n)


def get_q_vals(pvals_lst):
    """nb_model.py:340-342: Benjamini-Hochberg q-values, statsmodels.stats.multitest.fdrcorrection(pvals)[1] (method
    'indep'): q_(i) = min_{j >= i} p_(j) n / j in ascending order of p, capped at 1.  NaNs propagate the way the sort
    places them (last)."""
    if type(pvals_lst).__module__.startswith("torch") and pvals_lst.is_cuda:
        # device form for whole-genome tile sets (57.6 M p-values per cohort): the same IEEE operations in the same order
        # (p / (rank / n), reverse running minimum, cap), so the same bits as the host form and as statsmodels
        return get_q_vals_rows(pvals_lst.reshape(1, -1)).reshape(pvals_lst.shape)
    p = np.asarray(pvals_lst, dtype=np.float64)
    n = p.size
    if n == 0:
        return p.copy()
    order = np.argsort(p, kind="stable")
    ps = p[order]
    q = ps / (np.arange(1, n + 1) / float(n))          # statsmodels' own operation order (p / ecdf): bit-identical
    q = np.minimum.accumulate(q[::-1])[::-1]
    q = np.minimum(q, 1.0)
    out = np.empty_like(q)
    out[order] = q
    return out
