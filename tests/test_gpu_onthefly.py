"""quickDriver (DIG_onthefly) end to end on a synthetic genome, against an expectation assembled from the oracle the
way the reference's per-element loop does it (onthefly_tools.py:109-164): sequence -> window / block context counts
-> per-element sums -> NB mid-p tests -> Fisher.  Integers bit-exact, p-values within the tolerance contract."""
import os
import subprocess
import sys

import numpy as np
import pandas as pd
import pytest

from conftest import ROOT, rel_close

pytestmark = pytest.mark.gpu


def _make_case(tmp_path, rng):
    from digdriver_amd.io import mapfile
    from oracle import dig_oracle as O
    window, nbins = 1000, {"1": 40, "2": 25}
    seqs = {}
    for c, n in nbins.items():
        s = rng.choice(np.frombuffer(b"ACGTacgtN", np.uint8), n * window, p=[.24, .24, .24, .24, .005, .005, .005, .005, .02])
        seqs["chr" + c] = s.tobytes().decode()
    fa = tmp_path / "genome.fa"
    with open(fa, "w") as f:
        for name, s in seqs.items():
            f.write(">%s some description\n" % name)
            for i in range(0, len(s), 60):
                f.write(s[i:i + 60] + "\n")
    # region_params on the 1-kb grid + a sequence model
    rows = []
    for c, n in nbins.items():
        for b in range(n):
            mu = float(rng.gamma(9.0, 3.0))
            rows.append((int(c), b * window, (b + 1) * window, int(rng.poisson(mu)), mu, float(rng.gamma(4.0, 1.0)), 1.0, 0.5,
                         bool(rng.uniform() < 0.15)))
    rp = pd.DataFrame(rows, columns=["CHROM", "START", "END", "Y_TRUE", "Y_PRED", "STD", "MAPP", "QUANT", "FLAG"])
    rp.index = ["chr{}:{}-{}".format(c, s, e) for c, s, e in zip(rp.CHROM, rp.START, rp.END)]
    seq_rows = O.model_rows192()
    freq = rng.dirichlet(np.ones(192)) * 1e-3
    sm = pd.DataFrame({"MUT_TYPE": [m for m, _ in seq_rows], "CONTEXT": [c for _, c in seq_rows],
                       "COUNT": rng.integers(1, 1000, 192), "FREQ": freq})
    pre = str(tmp_path / "cohort.map")
    mapfile.write_frame(pre, "region_params", rp)
    mapfile.write_frame(pre, "sequence_model_192", sm)
    mapfile.write_array(pre, "idx", rp[["CHROM", "START", "END"]].values.astype(np.int32))
    # elements: 1-3 blocks, both strands, one crossing a window edge exactly, one at the chromosome start
    elts, lines = [], []
    specs = [("1", 0, [(0, 300)], "+"), ("1", 2000, [(0, 1000)], "-"), ("2", 5100, [(0, 200), (700, 150), (1900, 400)], "-")]
    for i in range(25):
        c = "12"[int(rng.integers(0, 2))]
        st = int(rng.integers(0, nbins[c] * window - 4000))
        nb = int(rng.integers(1, 4))
        blocks, off = [], 0
        for _ in range(nb):
            z = int(rng.integers(50, 900))
            blocks.append((off, z))
            off += z + int(rng.integers(10, 600))
        specs.append((c, st, blocks, "+-"[int(rng.integers(0, 2))]))
    for i, (c, st, blocks, strand) in enumerate(specs):
        name = "E%02d" % i
        end = st + blocks[-1][0] + blocks[-1][1]
        lines.append("%s\t%d\t%d\t%s\t0\t%s\t%d\t%d\t.\t%d\t%s,\t%s,\n" % (
            c, st, end, name, strand, st, st, len(blocks), ",".join(str(z) for _, z in blocks), ",".join(str(o) for o, _ in blocks)))
        elts.append((name, c, strand, [(st + o, st + o + z) for o, z in blocks]))
    bed = tmp_path / "elts.bed"
    bed.write_text("".join(lines))
    muts = []
    for name, c, strand, blocks in elts:
        for _ in range(int(rng.poisson(4))):
            s, e = blocks[int(rng.integers(0, len(blocks)))]
            p = int(rng.integers(s, e))
            muts.append((c, p, p + 1, "A", "T", "S%d" % rng.integers(0, 12), ".", "Noncoding", "A>T", "CAG"))
        if rng.uniform() < 0.4:
            p = blocks[0][0]
            muts.append((c, p, p + 3, "AGG", "A", "S%d" % rng.integers(0, 12), ".", "INDEL", "DEL", "."))
    mut = tmp_path / "cohort.tsv"
    pd.DataFrame(muts).to_csv(mut, sep="\t", header=False, index=False)
    return dict(pre=pre, fa=str(fa), bed=str(bed), mut=str(mut), seqs=seqs, rp=rp, sm=sm, elts=elts, window=window)


def _expected(case, cj, cj_indel, tab):
    """The reference's per-element loop restated with oracle functions."""
    from oracle import dig_oracle as O
    rp, window, seqs = case["rp"], case["window"], case["seqs"]
    names = [m + "|" + c for m, c in zip(case["sm"].MUT_TYPE, case["sm"].CONTEXT)]
    subst = [c + ">" + c[0] + m[2] + c[2] for m, c in zip(case["sm"].MUT_TYPE, case["sm"].CONTEXT)]
    order = np.argsort(np.array(subst), kind="stable")
    d_pr = case["sm"].FREQ.values[order]
    keys = sorted(subst)
    rho192 = O.minus_strand_gather192()
    out = {}
    for name, c, strand, blocks in case["elts"]:
        starts = sorted({x for s, e in blocks for x in range(s // window * window, -(-e // window) * window, window)})
        rc = np.zeros(64, np.int64)
        mu = var = 0.0
        robs = 0
        for x in starts:
            rc += O.count_contexts_region(seqs["chr" + c], x, x + window)
            row = rp.loc["chr{}:{}-{}".format(c, x, x + window)]
            mu += row.Y_PRED
            var += row.STD ** 2
            robs += row.Y_TRUE
        region_counts = np.repeat(rc, 3).astype(float)
        if strand == "-":
            region_counts = region_counts[rho192]
        L = np.zeros(192)
        for s, e in blocks:
            cnt = O.count_contexts_region(seqs["chr" + c], s, e)
            if strand == "-":
                cnt = cnt[O.minus_strand_gather64()]
            L += np.repeat(cnt, 3)
        p_mut = ((d_pr / (region_counts * d_pr).sum()) * L).sum()
        sigma = np.sqrt(var)
        alpha, theta = mu ** 2 / sigma ** 2, sigma ** 2 / mu
        r_size, e_size = int(region_counts.sum() / 3), int(L.sum() / 3)
        out[name] = dict(MU=mu, SIGMA=sigma, R_OBS=robs, R_SIZE=r_size, ELT_SIZE=e_size, Pi_SUM=p_mut, Pi_INDEL=e_size / r_size,
                         ALPHA=alpha, THETA=theta * cj, THETA_INDEL=theta * cj_indel * cj_indel)
    df = pd.DataFrame(out).T
    df = tab.merge(df, left_index=True, right_index=True)
    a = df.ALPHA.values.astype(float)
    p = 1.0 / (df.THETA.values.astype(float) * df.Pi_SUM.values.astype(float) + 1.0)
    df["EXP_SNV"] = a * df.THETA.values.astype(float) * df.Pi_SUM.values.astype(float)
    df["PVAL_SNV_BURDEN"] = O.nb_pvalue_greater_midp(df.OBS_SNV.values.astype(float), a, p)
    df["PVAL_SAMPLE_BURDEN"] = O.nb_pvalue_greater_midp(df.OBS_SAMPLES.values.astype(float), a, p)
    pi = 1.0 / (df.THETA_INDEL.values.astype(float) * df.Pi_INDEL.values.astype(float) + 1.0)
    df["EXP_INDEL"] = a * df.THETA_INDEL.values.astype(float) * df.Pi_INDEL.values.astype(float)
    df["PVAL_INDEL_BURDEN"] = O.nb_pvalue_greater_midp(df.OBS_INDEL.values.astype(float), a, pi)
    df["PVAL_MUT_BURDEN"] = O.fisher_combine(df.PVAL_SNV_BURDEN.values, df.PVAL_INDEL_BURDEN.values)
    return df


def test_quickdriver_matches_oracle_loop(tmp_path):
    from digdriver_amd.data_tools import mutation_tools
    from digdriver_amd.driver_model import onthefly_tools
    rng = np.random.default_rng(17)
    case = _make_case(tmp_path, rng)
    cj, cj_indel = 0.004, 0.0007
    got = onthefly_tools.DIG_onthefly(case["pre"], case["mut"], case["fa"], f_elts_bed=case["bed"], scale_factor=cj,
                                      scale_factor_indel=cj_indel, scale_by_expectation=False)
    tab = mutation_tools.tabulate_mutations_in_element(case["mut"], case["bed"], bed12=True, drop_duplicates=True,
                                                       all_elements=True)
    want = _expected(case, cj, cj_indel, tab)
    got = got.loc[want.index]
    for col in ("OBS_SNV", "OBS_SAMPLES", "OBS_INDEL", "R_OBS", "R_SIZE", "ELT_SIZE"):
        assert np.array_equal(got[col].values.astype(np.int64), want[col].values.astype(np.int64)), col
    for col in ("MU", "SIGMA", "Pi_SUM", "Pi_INDEL", "ALPHA", "THETA", "THETA_INDEL", "EXP_SNV", "EXP_INDEL",
                "PVAL_SNV_BURDEN", "PVAL_SAMPLE_BURDEN", "PVAL_INDEL_BURDEN", "PVAL_MUT_BURDEN"):
        rel_close(got[col].values.astype(float), want[col].values.astype(float), rtol=1e-6)
    assert (got.OBS_SNV.values > 0).any() and (got.OBS_INDEL.values > 0).any()
    # region string form: one element named UserELT, '+' strand
    one = onthefly_tools.DIG_onthefly(case["pre"], case["mut"], case["fa"], region_str="chr1:2000-3000", scale_factor=cj,
                                      scale_factor_indel=cj_indel, scale_by_expectation=False)
    assert list(one.index) == ["UserELT"] and int(one.ELT_SIZE.iloc[0]) <= 1000
    # the command line writes the same table
    out = tmp_path / "out"
    subprocess.check_call([sys.executable, os.path.join(ROOT, "scripts", "DigDriver.py"), "quickDriver", case["mut"],
                           case["pre"], case["fa"], "--f_elts_bed", case["bed"], "--scale-factor-manual", str(cj),
                           "--scale-factor-indel-manual", str(cj_indel), "--outdir", str(out), "--outpfx", "q"],
                          env=dict(os.environ, PYTHONPATH=ROOT))
    res = pd.read_csv(out / "q.results.txt", sep="\t", index_col=0)
    rel_close(res.loc[want.index].PVAL_MUT_BURDEN.values, want.PVAL_MUT_BURDEN.values.astype(float), rtol=1e-6)


def test_preprocess_pretrain_driver_chain_equals_quickdriver(tmp_path):
    """DigPreprocess (window counts -> element data) -> DigPretrain elementModel -> DigDriver elementDriver gives the
    same table as quickDriver on the same inputs: both routes compute the element parameters from the same sequence,
    one through stored counts, one on the fly.  (The indel columns differ by the reference's double scaling in the
    on-the-fly route, so strict_reference=False is used for the comparison.)"""
    from digdriver_amd.driver_model import onthefly_tools
    from digdriver_amd.io import mapfile
    rng = np.random.default_rng(23)
    case = _make_case(tmp_path, rng)
    env = dict(os.environ, PYTHONPATH=ROOT)
    run = lambda *a: subprocess.check_call([sys.executable] + [str(x) for x in a], env=env)
    # windows bed = the region_params grid
    wbed = tmp_path / "windows.bed"
    case["rp"][["CHROM", "START", "END"]].to_csv(wbed, sep="\t", header=False, index=False)
    gc, ed = str(tmp_path / "genome_counts.map"), str(tmp_path / "element_data.map")
    pp = os.path.join(ROOT, "scripts", "DigPreprocess.py")
    run(pp, "countGenomeContext", case["fa"], gc, "--bed", wbed)
    run(pp, "initialize_f_data", ed, gc)
    run(pp, "preprocess_element_model", ed, case["pre"], case["fa"], "myelts", "--f-bed", case["bed"], "--window", case["window"])
    run(os.path.join(ROOT, "scripts", "DigPretrain.py"), "elementModel", case["pre"], ed, "myelts")
    out = tmp_path / "o"
    cj, cji = 0.004, 0.0007
    run(os.path.join(ROOT, "scripts", "DigDriver.py"), "elementDriver", case["mut"], case["pre"], "myelts", "--f-bed", case["bed"],
        "--scale-factor-manual", cj, "--scale-factor-indel-manual", cji, "--outdir", out, "--outpfx", "e")
    res = pd.read_csv(out / "e.results.txt", sep="\t", index_col=0)
    quick = onthefly_tools.DIG_onthefly(case["pre"], case["mut"], case["fa"], f_elts_bed=case["bed"], scale_factor=cj,
                                        scale_factor_indel=cji, scale_by_expectation=False, strict_reference=False)
    common = [i for i in res.index if i in quick.index]
    assert len(common) == len(res) > 5
    for col in ("OBS_SNV", "OBS_SAMPLES", "OBS_INDEL", "R_OBS", "R_SIZE", "ELT_SIZE"):
        assert np.array_equal(res.loc[common, col].values.astype(np.int64), quick.loc[common, col].values.astype(np.int64)), col
    for col in ("MU", "SIGMA", "Pi_SUM", "Pi_INDEL", "EXP_SNV", "PVAL_SNV_BURDEN", "PVAL_SAMPLE_BURDEN", "PVAL_INDEL_BURDEN",
                "PVAL_MUT_BURDEN"):
        rel_close(res.loc[common, col].values.astype(float), quick.loc[common, col].values.astype(float), rtol=1e-9)
    assert int(mapfile.read_frame(gc, "genome_counts").COUNT.sum()) > 0


def test_sites_route_matches_reference_preprocessing(tmp_path):
    """preprocess_sites -> elementModel -> elementDriver --f-sites.  The per-element L counts, overlapped windows and
    region counts are the reference's own (tests/golden/sites_golden.json, made by running its preprocess_sites): the
    model columns computed here from the site file must equal the ones implied by those stored vectors."""
    import json
    from conftest import GOLDEN
    from digdriver_amd.io import mapfile
    from digdriver_amd.sequence_model import genic_driver_tools, sequence_tools
    from oracle import dig_oracle as O
    g = json.load(open(os.path.join(GOLDEN, "sites_golden.json")))
    window = g["window"]
    rng = np.random.default_rng(5)
    idx = np.array(g["bin_idx"])
    n = len(idx)
    rp = pd.DataFrame({"CHROM": idx[:, 0], "START": idx[:, 1], "END": idx[:, 2], "Y_TRUE": rng.poisson(20, n),
                       "Y_PRED": rng.gamma(9.0, 3.0, n), "STD": rng.gamma(4.0, 1.0, n), "FLAG": rng.uniform(size=n) < 0.1},
                      index=["chr{}:{}-{}".format(*r) for r in idx])
    sm = pd.DataFrame({"MUT_TYPE": g["seq_mut_type"], "CONTEXT": g["seq_context"], "FREQ": rng.dirichlet(np.ones(192)) * 1e-3})
    pre, dat = str(tmp_path / "pre.map"), str(tmp_path / "dat.map")
    mapfile.write_frame(pre, "region_params", rp)
    mapfile.write_frame(pre, "sequence_model_192", sm)
    mapfile.write_array(pre, "idx", idx.astype(np.int32))
    mapfile.write_array(dat, "window_%d/full_window_si_index" % window, idx)
    mapfile.write_array(dat, "window_%d/full_window_si_values" % window, np.array(g["bin_ctx"]))
    fs, fm = tmp_path / "sites.tsv", tmp_path / "m.tsv"
    pd.DataFrame(g["sites_rows"]).to_csv(fs, sep="\t", header=False, index=False)
    pd.DataFrame(g["mut_rows"]).to_csv(fm, sep="\t", header=False, index=False)
    env = dict(os.environ, PYTHONPATH=ROOT)
    subprocess.check_call([sys.executable, os.path.join(ROOT, "scripts", "DigPreprocess.py"), "preprocess_element_model", dat, pre,
                           "unused.fa", "mysites", "--f-sites", str(fs), "--window", str(window)], env=env)
    names = list(mapfile.read_array(dat, "window_%d/mysites/names" % window).astype(str))
    assert names == sorted(g["elements"])
    L = mapfile.read_array(dat, "window_%d/mysites/L" % window)
    for i, nme in enumerate(names):
        assert L[i].tolist() == g["elements"][nme]["L_counts"], nme
    frame = genic_driver_tools.nonc_model_parallel(pre, dat, "mysites", 1).set_index("ELT")
    subst = [c + ">" + c[0] + m[2] + c[2] for m, c in zip(sm.MUT_TYPE, sm.CONTEXT)]
    d_pr = sm.FREQ.values[np.argsort(np.array(subst), kind="stable")]
    for nme in names:
        e = g["elements"][nme]
        rc, Lc = np.array(e["region_counts"], float), np.array(e["L_counts"], float)
        want_p = ((d_pr / (rc * d_pr).sum()) * Lc).sum()
        assert abs(frame.loc[nme, "P_SUM"] - want_p) <= 1e-12 * abs(want_p), nme
        assert int(frame.loc[nme, "R_SIZE"]) == int(rc.sum() / 3) and int(frame.loc[nme, "ELT_SIZE"]) == int(Lc.sum() / 3)
        rows = ["chr{}:{}-{}".format(*o) for o in e["overlaps"]]
        assert abs(frame.loc[nme, "MU"] - rp.loc[rows, "Y_PRED"].sum()) < 1e-9
    mapfile.write_frame(pre, "mysites", frame.reset_index())
    out = tmp_path / "o"
    subprocess.check_call([sys.executable, os.path.join(ROOT, "scripts", "DigDriver.py"), "elementDriver", str(fm), pre, "mysites",
                           "--f-sites", str(fs), "--scale-factor-manual", "0.01", "--scale-factor-indel-manual", "0.001",
                           "--outdir", str(out), "--outpfx", "s"], env=env)
    res = pd.read_csv(out / "s.results.txt", sep="\t", index_col=0)
    assert [str(i) for i in res.index] == names or set(map(str, res.index)) == set(names)
    tab = dict(zip(g["tab_index"], g["tab_obs_snv"]))
    for nme in names:
        assert int(res.loc[nme, "OBS_SNV"]) == tab.get(nme, 0)
    a = res.ALPHA.values
    p = 1.0 / (res.THETA.values * res.Pi_SUM.values + 1.0)
    rel_close(res.PVAL_SNV_BURDEN.values, O.nb_pvalue_greater_midp(res.OBS_SNV.values.astype(float), a, p), rtol=1e-6)


def test_cli_chain_on_hdf5_maps(tmp_path):
    """The drop-in claim on the reference's file format: the whole chain -- DigPreprocess countGenomeContext /
    initialize_f_data / preprocess_element_model -> DigPretrain elementModel -> DigDriver elementDriver -- on `.h5` maps
    (pretrained map, genome counts and element data all HDF5, written and read by io/h5lite.py + io/pandas_fixed.py) gives
    the table the directory-mirror run gives; and DigPretrain elementModel on an element container in the REFERENCE's
    own layout (one group per element with L_counts / region_counts / attrs['overlaps'], sequence_tools.py:639-641)
    gives the same element frame."""
    from digdriver_amd.io import h5lite, mapfile
    from digdriver_amd.sequence_model import genic_driver_tools
    rng = np.random.default_rng(23)
    case = _make_case(tmp_path, rng)
    env = dict(os.environ, PYTHONPATH=ROOT)
    run = lambda *a: subprocess.check_call([sys.executable] + [str(x) for x in a], env=env)
    wbed = tmp_path / "windows.bed"
    case["rp"][["CHROM", "START", "END"]].to_csv(wbed, sep="\t", header=False, index=False)
    pre_h5 = str(tmp_path / "cohort.Pretrained.h5")
    for key in ("region_params", "sequence_model_192"):
        mapfile.write_frame(pre_h5, key, mapfile.read_frame(case["pre"], key))
    mapfile.write_array(pre_h5, "idx", mapfile.read_array(case["pre"], "idx"), compression="gzip")
    mapfile.write_attrs(pre_h5, cohort_name="synthetic")
    results = {}
    for tag, pre, ext in (("dir", case["pre"], ".map"), ("h5", pre_h5, ".h5")):
        gc, ed = str(tmp_path / ("genome_counts" + ext)), str(tmp_path / ("element_data" + ext))
        pp = os.path.join(ROOT, "scripts", "DigPreprocess.py")
        run(pp, "countGenomeContext", case["fa"], gc, "--bed", wbed)
        run(pp, "initialize_f_data", ed, gc)
        run(pp, "preprocess_element_model", ed, pre, case["fa"], "myelts", "--f-bed", case["bed"], "--window", case["window"])
        run(os.path.join(ROOT, "scripts", "DigPretrain.py"), "elementModel", pre, ed, "myelts")
        out = tmp_path / ("o_" + tag)
        run(os.path.join(ROOT, "scripts", "DigDriver.py"), "elementDriver", case["mut"], pre, "myelts", "--f-bed", case["bed"],
            "--scale-factor-manual", 0.004, "--scale-factor-indel-manual", 0.0007, "--outdir", out, "--outpfx", "e")
        results[tag] = (pd.read_csv(out / "e.results.txt", sep="\t", index_col=0), ed)
    a, b = results["dir"][0], results["h5"][0]
    assert len(a) > 5 and list(a.index) == list(b.index) and list(a.columns) == list(b.columns)
    pd.testing.assert_frame_equal(a, b, check_exact=True)
    assert (a.PVAL_MUT_BURDEN.values > 0).all() and mapfile.has_key(pre_h5, "myelts")
    # the element frame stored in the HDF5 map is a genuine pandas "fixed" group
    g = h5lite.read_tree(pre_h5)["myelts"]
    assert str(g.attrs["pandas_type"]) == "frame" and "axis0" in g.children and "block0_values" in g.children
    # ---- the reference's per-element container for the same elements ----
    ed_h5 = results["h5"][1]
    w = case["window"]
    names = mapfile.read_array(ed_h5, "window_%d/myelts/names" % w).astype(str)
    L = mapfile.read_array(ed_h5, "window_%d/myelts/L" % w)
    strand = mapfile.read_array(ed_h5, "window_%d/myelts/strand" % w).astype(str)
    chrom = mapfile.read_array(ed_h5, "window_%d/myelts/chrom" % w)
    ptr = mapfile.read_array(ed_h5, "window_%d/myelts/blk_ptr" % w)
    bs, be = mapfile.read_array(ed_h5, "window_%d/myelts/blk_start" % w), mapfile.read_array(ed_h5, "window_%d/myelts/blk_end" % w)
    si_index = mapfile.read_array(ed_h5, "window_%d/full_window_si_index" % w)
    si_values = mapfile.read_array(ed_h5, "window_%d/full_window_si_values" % w)
    row = {(int(c), int(s)): i for i, (c, s, _) in enumerate(si_index)}
    order = genic_driver_tools._minus_strand_order()
    ref_ed = str(tmp_path / "element_data_reference_layout.h5")
    root = h5lite.Group()
    root.set("window_%d/full_window_si_index" % w, h5lite.Dataset(np.asarray(si_index)))
    root.set("window_%d/full_window_si_values" % w, h5lite.Dataset(np.asarray(si_values)))
    for j, nme in enumerate(names):
        bins = sorted({(int(chrom[j]), int(x)) for s, e in zip(bs[ptr[j]:ptr[j + 1]], be[ptr[j]:ptr[j + 1]])
                       for x in range((int(s) // w) * w, -(-int(e) // w) * w, w)})          # get_ideal_overlaps (:275-283)
        rc = np.repeat(np.sum([si_values[row[b]] for b in bins], axis=0), 3)
        if strand[j] in ("-", "-1"):
            rc = rc[order]
        grp = "window_%d/refelts/%s" % (w, nme)
        root.set(grp + "/L_counts", h5lite.Dataset(np.asarray(L[j]).reshape(-1).astype(float)))
        root.set(grp + "/region_counts", h5lite.Dataset(rc))
        root[grp].attrs["overlaps"] = np.array([[c, s, s + w] for c, s in bins])
    h5lite.write_tree(ref_ed, root)
    want = genic_driver_tools.nonc_model_parallel(pre_h5, ed_h5, "myelts", 1).set_index("ELT")
    got = genic_driver_tools.nonc_model_parallel(pre_h5, ref_ed, "refelts", 1).set_index("ELT").loc[want.index]
    pd.testing.assert_frame_equal(got, want, check_exact=True)
