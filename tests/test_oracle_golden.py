"""Pin the oracle (oracle/dig_oracle.py and oracle/dig_oracle.c) to the golden vectors that
tests/golden/make_golden.py produced by running the REAL reference in the build container.
CPU only."""
import ctypes
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, rel_close
from oracle import dig_oracle as O

P = ctypes.POINTER(ctypes.c_double)


def _c3(lib, name, k, a, p):
    k, a, p = (np.ascontiguousarray(v, np.float64) for v in (k, a, p))
    out = np.empty_like(k)
    getattr(lib, name)(k.ctypes.data_as(P), a.ctypes.data_as(P), p.ctypes.data_as(P), out.ctypes.data_as(P),
                       ctypes.c_int64(k.size))
    return out


def test_subst_index_and_strand_permutation():
    g = json.load(open(os.path.join(GOLDEN, "subst_index.json")))
    assert O.subst_idx192() == g["subst_idx"]
    assert O.context64() == g["context64"]
    assert [list(r) for r in O.model_rows192()] == g["model_rows"]
    assert O.minus_strand_gather192().tolist() == g["minus_strand_gather"]
    # 192-permutation == 64-context reverse-complement permutation applied to triples
    g64 = O.minus_strand_gather64()
    v64 = np.arange(64) * 7 + 3
    assert (np.repeat(v64, 3)[O.minus_strand_gather192()] == np.repeat(v64[g64], 3)).all()
    assert (g64[g64] == np.arange(64)).all()   # involution


def test_py_oracle_nb_midp_is_reference_arithmetic():
    d = np.load(os.path.join(GOLDEN, "nb_midp_golden.npz"))
    with np.errstate(all="ignore"):
        got = O.nb_pvalue_greater_midp(d["k"], d["alpha"], d["p"])
    # same scipy calls as the reference -> identical including NaNs and underflow zeros
    assert np.array_equal(got, d["pval"], equal_nan=True)
    spot = O.nb_pvalue_greater_midp(d["spot_k"], d["spot_alpha"], 1 / (d["spot_theta"] * d["spot_pi"] + 1))
    np.testing.assert_allclose(spot, [5.04965209e-01, 2.33991408e-01, 4.81846261e-02, 5.40037089e-02, 8.22104015e-31],
                               rtol=2e-9)


def test_py_oracle_scalar_siblings():
    d = np.load(os.path.join(GOLDEN, "nb_exact_golden.npz"))
    assert np.array_equal(O.nb_pvalue_exact(d["k"], d["alpha"], d["p"]), d["pval_exact"], equal_nan=True)
    assert np.array_equal(O.nb_pvalue_greater(d["k"], d["alpha"], d["p"]), d["pval_greater"], equal_nan=True)
    assert np.array_equal(O.nb_pvalue_midp(d["k"], d["alpha"], d["p"]), d["pval_midp"], equal_nan=True)
    # SURVEY 8c spot values
    assert O.nb_pvalue_exact(0, 4, .5) == 0.0625
    assert O.nb_pvalue_exact(10, 4, .5) == pytest.approx(0.046142578125, rel=1e-12)
    assert O.nb_pvalue_exact(4, 4, .5) == pytest.approx(0.5, rel=1e-12)
    assert O.nb_pvalue_exact(3000, 4, .5) == 0.0


def test_c_oracle_nb_family(oracle_clib):
    d = np.load(os.path.join(GOLDEN, "nb_midp_golden.npz"))
    rel_close(_c3(oracle_clib, "dig_oracle_nb_midp_upper_v", d["k"], d["alpha"], d["p"]), d["pval"])
    e = np.load(os.path.join(GOLDEN, "nb_exact_golden.npz"))
    rel_close(_c3(oracle_clib, "dig_oracle_nb_exact_v", e["k"], e["alpha"], e["p"]), e["pval_exact"])
    rel_close(_c3(oracle_clib, "dig_oracle_nb_greater_v", e["k"], e["alpha"], e["p"]), e["pval_greater"])
    rel_close(_c3(oracle_clib, "dig_oracle_nb_midp_twosided_v", e["k"], e["alpha"], e["p"]), e["pval_midp"])


def test_fisher_oracles(oracle_clib):
    d = np.load(os.path.join(GOLDEN, "fisher_golden.npz"))
    assert np.array_equal(O.fisher_combine(d["p1"], d["p2"]), d["out"], equal_nan=True)
    assert O.fisher_combine(1e-3, 0.5) == pytest.approx(0.004300451229771043, rel=1e-12)
    p1, p2 = np.ascontiguousarray(d["p1"]), np.ascontiguousarray(d["p2"])
    out = np.empty_like(p1)
    oracle_clib.dig_oracle_fisher_v(p1.ctypes.data_as(P), p2.ctypes.data_as(P), out.ctypes.data_as(P),
                                    ctypes.c_int64(p1.size))
    rel_close(out, d["out"], rtol=1e-12)


def _element_inputs(d):
    n = len(d["mu"])
    pres = d["present"]
    obs = {k: np.where(pres, d["tab_" + k], 0).astype(np.int32) for k in ("obs_snv", "obs_samples", "obs_indel")}
    return n, obs


def test_py_oracle_element_stats():
    d = np.load(os.path.join(GOLDEN, "element_stats_golden.npz"))
    n, obs = _element_inputs(d)
    assert np.array_equal(obs["obs_snv"], d["out_OBS_SNV"].astype(np.int32))      # left join + NaN->0, bit exact
    assert np.array_equal(obs["obs_samples"], d["out_OBS_SAMPLES"].astype(np.int32))
    assert np.array_equal(obs["obs_indel"], d["out_OBS_INDEL"].astype(np.int32))
    r = O.element_stats(d["mu"], d["sigma"], d["pi_sum"], d["pi_indel"], obs["obs_snv"], obs["obs_samples"],
                        obs["obs_indel"], float(d["cj"]), float(d["cj_indel"]))
    for name in ["ALPHA", "THETA", "EXP_SNV", "PVAL_SNV_BURDEN", "PVAL_SAMPLE_BURDEN", "THETA_INDEL", "EXP_INDEL",
                 "PVAL_INDEL_BURDEN", "PVAL_MUT_BURDEN"]:
        assert np.array_equal(r[name], d["out_" + name], equal_nan=True), name


def test_c_oracle_element_stats(oracle_clib):
    d = np.load(os.path.join(GOLDEN, "element_stats_golden.npz"))
    n, obs = _element_inputs(d)
    mu, sg, ps, pi = (np.ascontiguousarray(d[k], np.float64) for k in ("mu", "sigma", "pi_sum", "pi_indel"))
    cj, cji = np.array([float(d["cj"])]), np.array([float(d["cj_indel"])])
    out = np.empty((7, n))
    I = ctypes.POINTER(ctypes.c_int32)
    oracle_clib.dig_oracle_element_stats(mu.ctypes.data_as(P), sg.ctypes.data_as(P), ps.ctypes.data_as(P),
                                         pi.ctypes.data_as(P), ctypes.c_int(0), obs["obs_snv"].ctypes.data_as(I),
                                         obs["obs_samples"].ctypes.data_as(I), obs["obs_indel"].ctypes.data_as(I),
                                         cj.ctypes.data_as(P), cji.ctypes.data_as(P), out.ctypes.data_as(P),
                                         ctypes.c_int64(n), ctypes.c_int64(1))
    for i, name in enumerate(["EXP_SNV", "PVAL_SNV_BURDEN", "PVAL_SAMPLE_BURDEN", "THETA_INDEL", "EXP_INDEL",
                              "PVAL_INDEL_BURDEN", "PVAL_MUT_BURDEN"]):
        rel_close(out[i], d["out_" + name])


def test_overlaps_golden():
    cases = json.load(open(os.path.join(GOLDEN, "overlaps_golden.json")))
    for c in cases:
        got = O.ideal_overlap_starts(c["intervals"][0], c["intervals"][1], c["window"])
        want = [o[1] for o in c["overlaps"]]
        assert got == want, c
        assert all(o[2] == o[1] + c["window"] for o in c["overlaps"])


def _acc_inputs(d):
    N = len(d["bin_idx"])
    E = len(d["elt_names"])
    ovp = d["elt_overlap_bins"]
    ptr = np.concatenate([[0], np.cumsum((ovp >= 0).sum(axis=1))]).astype(np.int64)
    idx = ovp[ovp >= 0].astype(np.int32)
    perm = O.model_rows_to_sorted_perm()
    d_pr = d["seq_freq"][perm][None, :]
    return dict(bin_mu=d["bin_y_pred"][:, None], bin_std=d["bin_std"][:, None], bin_y=d["bin_y_true"][:, None],
                bin_flag=d["bin_flag"][:, None], bin_ctx=d["bin_ctx"], ov_ptr=ptr, ov_idx=idx,
                L=d["elt_L"][:, None, :], strand_minus=(d["elt_strand"] == "-"), d_pr=d_pr), N, E


def _check_acc(r, d, cols, vals, n_class_col="P_SUM"):
    col = {c: i for i, c in enumerate(cols)}
    np.testing.assert_allclose(r["MU"][:, 0], vals[:, col["MU"]], rtol=1e-13)
    np.testing.assert_allclose(r["SIGMA"][:, 0], vals[:, col["SIGMA"]], rtol=1e-13)
    assert np.array_equal(r["R_OBS"][:, 0], vals[:, col["R_OBS"]].astype(np.int64))
    assert np.array_equal(r["FLAG"][:, 0], vals[:, col["FLAG"]].astype(np.int64))
    assert np.array_equal(r["R_SIZE"], vals[:, col["R_SIZE"]].astype(np.int64))
    np.testing.assert_allclose(r["P_INDEL"], vals[:, col["P_INDEL"]], rtol=1e-15)


def test_py_oracle_accumulate_elements():
    d = np.load(os.path.join(GOLDEN, "accumulate_golden.npz"))
    a, N, E = _acc_inputs(d)
    # the stored overlap rows reproduce get_ideal_overlaps through our own CSR builder
    bin_index = {(int(c), int(s)): i for i, (c, s, _) in enumerate(d["bin_idx"])}
    ptr, idx = O.build_overlap_csr(d["elt_chrom"], d["block_starts"], d["block_ends"], bin_index, int(d["window"]))
    assert np.array_equal(ptr, a["ov_ptr"]) and np.array_equal(idx, a["ov_idx"])
    # region_counts as stored by the reference == our repeat/permute of summed 64-context rows
    g192 = O.minus_strand_gather192()
    for e in range(0, E, 7):
        rc = np.repeat(d["bin_ctx"][idx[ptr[e]:ptr[e + 1]]].sum(axis=0), 3)
        if a["strand_minus"][e]:
            rc = rc[g192]
        assert np.array_equal(rc, d["elt_region_counts"][e])
    r = O.accumulate_elements(**a)
    cols = list(d["out_cols"])
    vals = d["out_vals"]
    _check_acc(r, d, cols, vals)
    col = {c: i for i, c in enumerate(cols)}
    np.testing.assert_allclose(r["P"][:, 0, 0], vals[:, col["P_SUM"]], rtol=1e-13)
    assert np.array_equal(r["ELT_SIZE"], vals[:, col["ELT_SIZE"]].astype(np.int64))
    # vectorised form agrees
    f = O.accumulate_elements_fast(a["bin_mu"], a["bin_std"], a["bin_y"], a["bin_flag"], a["bin_ctx"], a["ov_ptr"],
                                   a["ov_idx"], a["L"], a["strand_minus"], a["d_pr"])
    np.testing.assert_allclose(f["P"], r["P"], rtol=1e-12)
    np.testing.assert_allclose(f["MU"], r["MU"], rtol=1e-13)
    assert np.array_equal(f["R_SIZE"], r["R_SIZE"]) and np.array_equal(f["FLAG"], r["FLAG"])


def test_py_oracle_zero_denominators_follow_the_reference():
    """accumulate_zero_golden.npz (make_golden.py::gen_accumulate_zero_den): the reference's own nonc_model where
    sum(region_counts * d_pr) == 0 because the cohort's FREQ table is exactly zero at every context the element's bins hold --
    t_pi = d_pr / 0 holds 0 / 0 = NaN there, so P_SUM is NaN whatever L is (genic_driver_tools.py:361-366); a zero in FREQ that
    leaves the denominator positive changes nothing."""
    d = np.load(os.path.join(GOLDEN, "accumulate_zero_golden.npz"))
    ovp = d["elt_overlap_bins"]
    ptr = np.concatenate([[0], np.cumsum((ovp >= 0).sum(axis=1))]).astype(np.int64)
    idx = ovp[ovp >= 0].astype(np.int32)
    perm = O.model_rows_to_sorted_perm()
    C = d["seq_freq"].shape[0]
    rep = lambda v: np.repeat(v[:, None], C, axis=1)
    r = O.accumulate_elements(rep(d["bin_y_pred"]), rep(d["bin_std"]), rep(d["bin_y_true"]), rep(d["bin_flag"]), d["bin_ctx"], ptr, idx,
                              d["elt_L"][:, None, :], d["elt_strand"] == "-", d["seq_freq"][:, perm])
    want = d["p_sum"]                                     # [3 tables, E]
    got = r["P"][:, 0, :].T
    assert np.isnan(want[1]).sum() >= 8 and np.isfinite(want[0]).all() and np.isfinite(want[2]).all()
    assert np.array_equal(np.isnan(got), np.isnan(want))
    ok = ~np.isnan(want)
    np.testing.assert_allclose(got[ok], want[ok], rtol=1e-13)


def test_py_oracle_tiled_model():
    d = np.load(os.path.join(GOLDEN, "accumulate_golden.npz"))
    window = int(d["window"])
    bin_index = {(int(c), int(s)): i for i, (c, s, _) in enumerate(d["bin_idx"])}
    rows = []
    for name in d["tile_names"]:
        chrom = int(name.split(":")[0][3:])
        start = int(name.split(":")[1].split("-")[0])
        rows.append(bin_index[(chrom, start // window * window)])     # genic_driver_tools.py:634-641
    T = len(rows)
    perm = O.model_rows_to_sorted_perm()
    r = O.accumulate_elements(d["bin_y_pred"][:, None], d["bin_std"][:, None], d["bin_y_true"][:, None],
                              d["bin_flag"][:, None], d["bin_ctx"], np.arange(T + 1, dtype=np.int64),
                              np.array(rows, np.int32), d["tile_L"][:, None, :], np.zeros(T, bool),
                              d["seq_freq"][perm][None, :])
    cols = list(d["out_cols"])
    vals = d["tile_out_vals"]
    _check_acc(r, d, cols, vals)
    col = {c: i for i, c in enumerate(cols)}
    np.testing.assert_allclose(r["P"][:, 0, 0], vals[:, col["P_SUM"]], rtol=1e-13)
    # _index_transform naming (genic_driver_tools.py:721-725)
    assert d["tile_out_names"][0].startswith("region_")


def test_py_oracle_genic_model():
    d = np.load(os.path.join(GOLDEN, "genic_golden.npz"))
    a = np.load(os.path.join(GOLDEN, "accumulate_golden.npz"))
    G = len(d["gene_names"])
    assert list(d["out_genes"]) == list(d["gene_names"])      # GENEX (chrX) was skipped by the reference
    ovp = d["gene_overlap_bins"]
    ptr = np.concatenate([[0], np.cumsum((ovp >= 0).sum(axis=1))]).astype(np.int64)
    idx = ovp[ovp >= 0].astype(np.int32)
    perm = O.model_rows_to_sorted_perm()
    glen = ((d["cds_ends"] - d["cds_starts"] + 1) * (d["cds_starts"] >= 0)).sum(axis=1)
    r = O.accumulate_elements(a["bin_y_pred"][:, None], a["bin_std"][:, None], a["bin_y_true"][:, None],
                              a["bin_flag"][:, None], a["bin_ctx"], ptr, idx, d["L_data"], np.zeros(G, bool),
                              a["seq_freq"][perm][None, :], gene_length=glen)
    col = {c: i for i, c in enumerate(d["out_cols"])}
    v = d["out_vals"]
    np.testing.assert_allclose(r["MU"][:, 0], v[:, col["MU"]], rtol=1e-13)
    np.testing.assert_allclose(r["SIGMA"][:, 0], v[:, col["SIGMA"]], rtol=1e-13)
    assert np.array_equal(r["R_SIZE"], v[:, col["R_SIZE"]].astype(np.int64))
    assert np.array_equal(glen, v[:, col["GENE_LENGTH"]].astype(np.int64))
    for q, name in enumerate(["P_SILENT", "P_MIS", "P_NONS", "P_SPLICE"]):
        np.testing.assert_allclose(r["P"][:, q, 0], v[:, col[name]], rtol=1e-12)
    np.testing.assert_allclose(r["P"][:, 2, 0] + r["P"][:, 3, 0], v[:, col["P_TRUNC"]], rtol=1e-12)
    np.testing.assert_allclose(r["P_INDEL"], v[:, col["P_INDEL"]], rtol=1e-15)


def test_c_oracle_accumulate(oracle_clib):
    d = np.load(os.path.join(GOLDEN, "accumulate_golden.npz"))
    a, N, E = _acc_inputs(d)
    r = O.accumulate_elements(**a)
    c = run_c_accumulate(oracle_clib, a, 1)
    np.testing.assert_allclose(c["MU"], r["MU"], rtol=1e-14)
    np.testing.assert_allclose(c["SIGMA"], r["SIGMA"], rtol=1e-14)
    np.testing.assert_allclose(c["P"], r["P"], rtol=1e-12)
    assert np.array_equal(c["R_OBS"], r["R_OBS"]) and np.array_equal(c["FLAG"], r["FLAG"])
    assert np.array_equal(c["R_SIZE"], r["R_SIZE"]) and np.array_equal(c["ELT_SIZE"], r["ELT_SIZE"])
    np.testing.assert_allclose(c["P_INDEL"], r["P_INDEL"], rtol=1e-15)


def run_c_accumulate(lib, a, n_class):
    f8 = lambda v: np.ascontiguousarray(v, np.float64)
    bin_mu, bin_std = f8(a["bin_mu"]), f8(a["bin_std"])
    N, C = bin_mu.shape
    bin_y = np.ascontiguousarray(a["bin_y"], np.int32)
    bin_flag = np.ascontiguousarray(a["bin_flag"], np.uint8)
    bin_ctx = np.ascontiguousarray(a["bin_ctx"], np.int32)
    ptr, idx = np.ascontiguousarray(a["ov_ptr"], np.int64), np.ascontiguousarray(a["ov_idx"], np.int32)
    L = np.ascontiguousarray(a["L"], np.int32)
    E = L.shape[0]
    sm = np.ascontiguousarray(a["strand_minus"], np.uint8)
    d_pr = f8(a["d_pr"])
    rho = np.ascontiguousarray(O.minus_strand_gather64(), np.int32)
    o = dict(MU=np.empty((E, C)), SIGMA=np.empty((E, C)), R_OBS=np.empty((E, C), np.int32),
             FLAG=np.empty((E, C), np.int32), P=np.empty((E, n_class, C)), R_SIZE=np.empty(E, np.int32),
             ELT_SIZE=np.empty(E, np.int32), P_INDEL=np.empty(E))
    vp = lambda x: x.ctypes.data_as(ctypes.c_void_p)
    lib.dig_oracle_accumulate_elements.argtypes = [ctypes.c_void_p] * 8 + [ctypes.c_int] + [ctypes.c_void_p] * 11 + \
        [ctypes.c_int64, ctypes.c_int64]
    lib.dig_oracle_accumulate_elements(vp(bin_mu), vp(bin_std), vp(bin_y), vp(bin_flag), vp(bin_ctx), vp(ptr), vp(idx),
                                       vp(L), n_class, vp(sm), vp(d_pr), vp(rho), vp(o["MU"]), vp(o["SIGMA"]),
                                       vp(o["R_OBS"]), vp(o["FLAG"]), vp(o["P"]), vp(o["R_SIZE"]), vp(o["ELT_SIZE"]),
                                       vp(o["P_INDEL"]), E, C)
    return o


def test_py_oracle_sequence_model():
    d = np.load(os.path.join(GOLDEN, "sequence_model_golden.npz"))
    # de-duplicate exactly as restrict_mutations_by_bed(unique=True) does (whole-row duplicates)
    _, first = np.unique(d["dedup_key"], return_index=True)
    rows, count, freq, ctx64, freq64 = O.train_sequence_model(d["mut_type"][first], d["context"][first],
                                                              d["genome_ctx"], d["genome_counts"])
    assert [r[0] for r in rows] == list(d["out_mut_type"]) and [r[1] for r in rows] == list(d["out_context"])
    assert np.array_equal(count, d["out_count"])
    assert np.array_equal(freq, d["out_freq"])
    assert ctx64 == list(d["out64_context"])
    np.testing.assert_allclose(freq64, d["out64_freq"], rtol=1e-15)


def test_context_counting_oracle_matches_reference():
    """oracle.count_contexts_* against the reference's count_contexts_by_regions / nonc_elt_context_count run on a
    small random genome with N runs, lower-case stretches, START == 0 and regions past the chromosome end."""
    import json
    from oracle import dig_oracle as O
    g = json.load(open(os.path.join(GOLDEN, "contexts_golden.json")))
    assert g["columns64"] == O.context64()
    chroms, starts, ends = zip(*g["regions"])
    got = O.count_contexts_regions(g["genome"], chroms, starts, ends)
    assert np.array_equal(got, np.array(g["counts64"]))
    assert g["index64"] == ["%s:%d-%d" % r for r in zip(chroms, starts, ends)]
    minus = [s in ("-", "-1") for s in g["strands"]]
    got192, keys = O.expand_contexts_192(O.count_contexts_regions(g["genome"], chroms, starts, ends, minus))
    assert keys == g["columns192"]
    assert np.array_equal(got192, np.array(g["counts192"]))


def test_per_base_front_half_oracle_vs_reference_nb_model():
    """oracle.base_probabilities_by_region / apply_nb_to_region against the reference's own nb_model output
    (tests/golden/tiled_golden.json.gz: nb_model.py:188-234 run with in-memory fasta / tabix stand-ins)."""
    import gzip
    import json
    g = json.loads(gzip.open(os.path.join(GOLDEN, "tiled_golden.json.gz")).read())
    ctx = O.context64()
    assert g["contexts"] == ctx
    for coh in g["cohorts"]:
        s64 = np.array([coh["d_pr"][c] for c in ctx])
        for binsize in (1, 50):
            run = coh["runs"][str(binsize)]
            pv, ps, ob, ex, pt = [], [], [], [], []
            for (chrom, s, e), mu, sigma in zip(g["idx"], coh["mu"], coh["sigma"]):
                starts = [r[1] for r in coh["rows"] if r[0] == str(chrom)]
                a, b, c_, d, f = O.apply_nb_to_region(g["genome"]["chr%d" % chrom], s64, s, e, mu, sigma, starts, binsize)
                pv.append(a); ps.append(b); ob.append(c_); ex.append(d); pt.append(f)
            pv, ps, ob, ex, pt = (np.concatenate(v) for v in (pv, ps, ob, ex, pt))
            assert len(pv) == len(run["PVAL"])
            assert np.array_equal(ob, np.array(run["OBS"]).astype(int)) and np.array_equal(ps, np.array(run["POS"]))
            np.testing.assert_allclose(pt, run["Pi"], rtol=1e-14, atol=0)
            np.testing.assert_allclose(ex, run["EXP"], rtol=1e-14, atol=0)
            rel_close(pv, np.array(run["PVAL"]), rtol=1e-12)


def _tabulate_golden():
    import gzip
    with gzip.open(os.path.join(GOLDEN, "tabulate_golden.json.gz"), "rt") as f:
        return json.load(f)


def test_oracle_tabulation_is_the_reference_tabulation():
    """oracle.interval_join_pairs + tabulate_elements against the reference's own tabulate_muts_per_sample_per_element /
    tabulate_mutations_in_element (mutation_tools.py:155-230; only pybedtools' intersect was stood in for when the
    golden was made): the (element, sample) count frame and every per-element summary, bit-exact."""
    g = _tabulate_golden()
    muts = [[r[0], int(r[1]), int(r[2])] + r[3:] for r in g["mut_rows"]]
    blocks = O.bed12_blocks(g["bed_rows"])
    bed_names = [r[3] for r in g["bed_rows"]]
    for dd in (False, True):
        per_pair, _, _ = O.tabulate_elements(muts, blocks, drop_duplicates=dd)
        want = g["per_pair_dedup" if dd else "per_pair"]
        want_map = {(e, s): (a, b) for e, s, a, b in zip(want["ELT"], want["SAMPLE"], want["OBS_SNV"], want["OBS_INDEL"])}
        assert len(want_map) == len(want["ELT"])
        assert {k: tuple(v) for k, v in per_pair.items()} == want_map
        assert [a + b for a, b in want_map.values()] == want["OBS_MUT"]
    assert sum(c["drop_duplicates"] for c in g["cases"]) == 6
    for case in g["cases"]:
        _, per_elt, black = O.tabulate_elements(muts, blocks, drop_duplicates=case["drop_duplicates"],
                                                max_muts_per_sample=case["max_muts_per_sample"],
                                                max_muts_per_elt_per_sample=case["max_muts_per_elt_per_sample"])
        assert black == case["blacklist"]
        got = {e: per_elt.get(e, (0, 0, 0)) for e in bed_names} if case["all_elements"] else per_elt
        want = dict(zip(case["index"], zip(case["OBS_SAMPLES"], case["OBS_SNV"], case["OBS_INDEL"])))
        assert got == want
    # the cases are not trivial: duplicates change counts, the caps bite, samples are blacklisted
    a, b = g["per_pair"], g["per_pair_dedup"]
    assert sum(a["OBS_MUT"]) > sum(b["OBS_MUT"]) and max(b["OBS_SNV"]) > 2 and max(b["OBS_INDEL"]) >= 1
    assert any(c["blacklist"] for c in g["cases"])


def test_oracle_interval_join_edges():
    """Half-open on both sides, text chromosome labels, nested blocks, the 1-bp convention for zero-length records."""
    mi, bi = O.interval_join_pairs(["1"] * 4, [99, 100, 199, 200], [100, 101, 200, 201], ["1"], [100], [200])
    assert mi.tolist() == [1, 2] and bi.tolist() == [0, 0]
    mi, bi = O.interval_join_pairs(["1", "chr1", "2"], [10, 10, 10], [11, 11, 11], ["1", "1", "chr1"], [0, 5, 0], [50, 20, 50])
    assert list(zip(mi.tolist(), bi.tolist())) == [(0, 0), (0, 1), (1, 2)]
    mi, bi = O.interval_join_pairs(["1", "1"], [7, 20], [7, 20], ["1", "1"], [7, 3], [7, 20])
    assert list(zip(mi.tolist(), bi.tolist())) == [(0, 0), (0, 1)]
