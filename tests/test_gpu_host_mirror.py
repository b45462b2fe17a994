"""GPU tests of the host-side mirror of the reference's function-level API (transfer_tools,
genic_driver_tools) against goldens produced by the reference's own functions."""
import os

import numpy as np
import pandas as pd
import pytest

from conftest import GOLDEN, rel_close

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def _gpu():
    from digdriver_amd import _lib
    _lib.require_device()


def _build_maps(tmp_path):
    """Mirror containers equivalent to the in-memory stand-ins make_golden.py fed to the reference."""
    from digdriver_amd.io import mapfile
    d = np.load(os.path.join(GOLDEN, "accumulate_golden.npz"))
    pre = str(tmp_path / "pretrained.map")
    dat = str(tmp_path / "element_data.map")
    idx = d["bin_idx"]
    df_reg = pd.DataFrame(dict(CHROM=idx[:, 0], START=idx[:, 1], END=idx[:, 2], Y_TRUE=d["bin_y_true"],
                               Y_PRED=d["bin_y_pred"], STD=d["bin_std"], FLAG=d["bin_flag"]),
                          index=['chr{}:{}-{}'.format(*r) for r in idx])
    # shuffle rows: the loader must sort the grid itself
    df_reg = df_reg.sample(frac=1.0, random_state=0)
    mapfile.write_frame(pre, "region_params", df_reg)
    mapfile.write_frame(pre, "sequence_model_192", pd.DataFrame(dict(MUT_TYPE=d["seq_mut_type"], CONTEXT=d["seq_context"],
                                                                      FREQ=d["seq_freq"])))
    mapfile.write_array(pre, "idx", idx.astype(np.int32))
    w = int(d["window"])
    mapfile.write_array(dat, "window_%d/full_window_si_values" % w, d["bin_ctx"])
    mapfile.write_array(dat, "window_%d/full_window_si_index" % w, idx)
    bs, be = d["block_starts"], d["block_ends"]
    nblk = (bs >= 0).sum(axis=1)
    base = "window_%d/elts/" % w
    mapfile.write_array(dat, base + "names", d["elt_names"])
    mapfile.write_array(dat, base + "chrom", d["elt_chrom"])
    mapfile.write_array(dat, base + "strand", d["elt_strand"])
    mapfile.write_array(dat, base + "blk_ptr", np.concatenate([[0], np.cumsum(nblk)]))
    mapfile.write_array(dat, base + "blk_start", bs[bs >= 0])
    mapfile.write_array(dat, base + "blk_end", be[be >= 0])
    mapfile.write_array(dat, base + "L", d["elt_L"].astype(np.int32))
    mapfile.write_frame(dat, "tiles/L_counts", pd.DataFrame(d["tile_L"].astype(np.int32), index=d["tile_names"]))
    g = np.load(os.path.join(GOLDEN, "genic_golden.npz"))
    gs, ge = g["cds_starts"], g["cds_ends"]
    gn = (gs >= 0).sum(axis=1)
    base = "window_%d/genes/" % w
    names = np.concatenate([g["gene_names"], ["GENEX"]])
    mapfile.write_array(dat, base + "names", names)
    mapfile.write_array(dat, base + "chrom", np.concatenate([g["gene_chrom"], [1]]))
    mapfile.write_array(dat, base + "chrom_str", np.concatenate([g["gene_chrom"].astype(str), ["X"]]))
    mapfile.write_array(dat, base + "strand", np.array(["+"] * len(names)))
    mapfile.write_array(dat, base + "blk_ptr", np.concatenate([[0], np.cumsum(np.concatenate([gn, [1]]))]))
    mapfile.write_array(dat, base + "blk_start", np.concatenate([gs[gs >= 0], [100]]))
    mapfile.write_array(dat, base + "blk_end", np.concatenate([ge[ge >= 0], [400]]))
    mapfile.write_array(dat, base + "L", np.concatenate([g["L_data"], np.zeros((1, 4, 192))]).astype(np.int32))
    return pre, dat, d, g


def _cmp_frame(df, cols, vals, skip=()):
    for i, c in enumerate(cols):
        if c in skip:
            continue
        got = df[c].values.astype(float)
        if c in ("ELT_SIZE", "FLAG", "R_SIZE", "R_OBS", "R_INDEL", "GENE_LENGTH"):
            assert np.array_equal(got, vals[:, i]), c
        else:
            rel_close(got, vals[:, i], 1e-11)


def test_nonc_tiled_genic_models_match_reference_loops(_gpu, tmp_path):
    from digdriver_amd.sequence_model import genic_driver_tools as gdt
    pre, dat, d, g = _build_maps(tmp_path)
    df = gdt.nonc_model(list(d["elt_names"]), pre, dat, "elts", False)
    assert list(df.columns) == ['ELT', 'ELT_SIZE', 'FLAG', 'R_SIZE', 'R_OBS', 'R_INDEL', 'MU', 'SIGMA', 'MU_INDEL',
                                'SIGMA_INDEL', 'P_SUM', 'P_INDEL']      # genic_driver_tools.py:404-417
    assert list(df.ELT) == list(d["elt_names"])
    _cmp_frame(df, list(d["out_cols"]), d["out_vals"])
    # subset + order is respected, parallel wrapper == all elements
    sub = list(d["elt_names"][[5, 3, 77]])
    assert list(gdt.nonc_model(sub, pre, dat, "elts", False).ELT) == sub
    allf = gdt.nonc_model_parallel(pre, dat, "elts", 4)
    assert allf.equals(df)
    # two cohorts in one launch == two single launches
    two = gdt.nonc_model(list(d["elt_names"]), [pre, pre], dat, "elts", False)
    assert len(two) == 2 and two[0].equals(df) and two[1].equals(df)

    tl = gdt.tiled_nonc_model(list(d["tile_names"]), pre, dat, "tiles")
    assert list(tl.ELT) == list(d["tile_out_names"])
    _cmp_frame(tl, list(d["out_cols"]), d["tile_out_vals"])

    gm = gdt.genic_model(list(g["gene_names"]) + ["GENEX"], pre, dat, "window_10kb/counts", False)
    assert list(gm.GENE) == list(g["out_genes"])                # the X gene is skipped
    assert list(gm.CHROM) == list(g["out_chrom"])
    _cmp_frame(gm, list(g["out_cols"]), g["out_vals"])

    ov = gdt.get_ideal_overlaps(4, np.array([[5, 25000, 99990], [500, 31000, 100010]]), 10000)
    assert ov == [(4, 0, 10000), (4, 20000, 30000), (4, 30000, 40000), (4, 90000, 100000), (4, 100000, 110000)]


def test_run_gene_model_matches_reference(_gpu, tmp_path):
    from digdriver_amd.driver_model import transfer_tools as tt
    from digdriver_amd.io import mapfile
    g = np.load(os.path.join(GOLDEN, "gene_stats_golden.npz"))
    frame = pd.DataFrame(g["frame_vals"], columns=list(g["frame_cols"]))
    for c in ("GENE_LENGTH", "R_SIZE", "R_OBS", "R_INDEL", "FLAG"):
        frame[c] = frame[c].astype(np.int64)
    frame.insert(0, "GENE", g["genes"])
    frame.insert(0, "CHROM", g["frame_chrom"])
    path = str(tmp_path / "genes.map")
    mapfile.write_frame(path, "genic_model", frame)
    frames = []
    for fused in (False, True):               # the reference's column-by-column sequence; the one-launch form (dig_gene_stats)
        df = tt.run_gene_model(os.path.join(GOLDEN, "gene_mutations.tsv"), path,
                               max_muts_per_sample=int(g["max_muts_per_sample"]),
                               max_muts_per_gene_per_sample=int(g["max_muts_per_gene_per_sample"]),
                               all_cosmic=list(g["null_excluded"]), fused=fused)
        assert list(df.index) == list(g["out_index"])
        cols = list(g["out_cols"])
        assert [c for c in df.columns if c != "CHROM"] == cols          # same columns, same order as the reference
        vals = g["out_vals"]
        for i, c in enumerate(cols):
            got = df[c].values.astype(float)
            if c.startswith(("OBS_", "N_SAMP_")) or c in ("GENE_LENGTH", "R_SIZE", "R_OBS", "R_INDEL", "FLAG"):
                assert np.array_equal(got, vals[:, i]), c                # integer columns: bit-exact
            else:
                rel_close(got, vals[:, i], 1e-6)
        frames.append(df)
    for c in frames[0].columns:                # the two routes run the same device functions: same bits
        a, b = frames[0][c].values, frames[1][c].values
        assert np.array_equal(a, b, equal_nan=True) if a.dtype.kind == "f" else (a == b).all(), c


def test_run_element_region_model_end_to_end(_gpu, tmp_path):
    """DigDriver.py elementDriver path on a synthetic cohort: TSV + bed12 + map -> results frame; fused and
    column-by-column routes agree bit for bit and match the oracle on the tabulated counts."""
    from bench import make_workload
    from digdriver_amd.driver_model import transfer_tools as tt
    from digdriver_amd.io import mapfile
    from oracle import dig_oracle as O
    rng = np.random.default_rng(4)
    E = 300
    names = ["elt%03d" % i for i in range(E)]
    starts = np.sort(rng.integers(1000, 900000, E))
    sizes = rng.integers(200, 1500, E)
    chrom = rng.integers(1, 5, E)
    bed = tmp_path / "e.bed"
    with open(bed, "w") as f:
        for n, c, s, z in zip(names, chrom, starts, sizes):
            f.write("%d\t%d\t%d\t%s\t0\t+\t%d\t%d\t.\t1\t%d,\t0,\n" % (c, s, s + z, n, s, s, z))
    mu = rng.gamma(9.0, 3.0, E)
    sigma = rng.gamma(4.0, 1.0, E)
    pi = sizes / 10000.0
    frame = pd.DataFrame(dict(ELT=names, ELT_SIZE=sizes, FLAG=rng.integers(0, 2, E).astype(bool), R_SIZE=10000,
                              R_OBS=rng.poisson(mu), R_INDEL=rng.poisson(mu), MU=mu, SIGMA=sigma, MU_INDEL=mu,
                              SIGMA_INDEL=sigma, P_SUM=pi, P_INDEL=pi))
    path = str(tmp_path / "cohort.map")
    mapfile.write_frame(path, "my_elts", frame)
    rows = []
    for n, c, s, z, m in zip(names, chrom, starts, sizes, mu * pi * 1.3):
        for _ in range(rng.poisson(m)):
            p = int(s + rng.integers(0, z))
            rows.append((str(c), p, p + 1, "A", "T", "S%d" % rng.integers(0, 40), ".", "Noncoding", "A>T", "CAG"))
        for _ in range(rng.poisson(m * 0.1)):
            p = int(s + rng.integers(0, z))
            rows.append((str(c), p, p + 3, "AGG", "A", "S%d" % rng.integers(0, 40), ".", "INDEL", "DEL", "."))
    mut = tmp_path / "m.tsv"
    pd.DataFrame(rows).to_csv(mut, sep="\t", header=False, index=False)
    a = tt.run_element_region_model(str(mut), str(bed), path, "my_elts", scale_factor=1.3, scale_factor_indel=0.13,
                                    scale_by_expectation=False)
    b = tt.run_element_region_model(str(mut), str(bed), path, "my_elts", scale_factor=1.3, scale_factor_indel=0.13,
                                    scale_by_expectation=False, fused=True)
    assert list(a.columns) == list(b.columns)
    for c in a.columns:
        assert np.array_equal(a[c].values, b[c].values, equal_nan=True), c
    # THETA_INDEL is updated in place; the appended columns and their order are the reference's
    assert list(a.columns)[-9:] == ['OBS_SAMPLES', 'OBS_SNV', 'OBS_INDEL', 'EXP_SNV', 'PVAL_SNV_BURDEN',
                                   'PVAL_SAMPLE_BURDEN', 'EXP_INDEL', 'PVAL_INDEL_BURDEN', 'PVAL_MUT_BURDEN']
    want = O.element_stats(mu, sigma, pi, pi, a.OBS_SNV.values, a.OBS_SAMPLES.values, a.OBS_INDEL.values, 1.3, 0.13)
    for c in ('EXP_SNV', 'PVAL_SNV_BURDEN', 'PVAL_SAMPLE_BURDEN', 'THETA_INDEL', 'EXP_INDEL', 'PVAL_INDEL_BURDEN',
              'PVAL_MUT_BURDEN'):
        rel_close(a[c].values, want[c], 1e-6)
    assert a.OBS_SNV.sum() > 0 and a.OBS_INDEL.sum() > 0 and (a.OBS_SAMPLES <= a.OBS_SNV + a.OBS_INDEL).all()
    # the reference's key guard (transfer_tools.py:1020-1021): PCAWG_cds scaling needs the PCAWG_cds model
    with pytest.raises(AssertionError):
        tt.run_element_region_model(str(mut), str(bed), path, "my_elts", scale_by_expectation=False, scale_type="PCAWG_cds")


def test_run_element_region_model_default_mode_matches_reference(tmp_path, monkeypatch, capsys):
    """The route `DigDriver.py elementDriver` takes when no scale option is given (scale_by_expectation=True,
    transfer_tools.py:989-1017): synonymous scale factor on blacklist-filtered de-duplicated rows, uniform indel factor with
    the reference's no-op CGC exclusion of the mutation frame, then the statistics block.  Golden: the reference function
    itself (tests/golden/make_golden.py::gen_run_element_expectation) with the panel, the frames and the bedtools
    tabulation handed in as fixtures; the same fixtures are used here (panel through --panel-dir's mechanism)."""
    from conftest import GOLDEN
    from digdriver_amd.data_tools import mutation_tools
    from digdriver_amd.driver_model import transfer_tools
    from digdriver_amd.io import mapfile
    g = np.load(os.path.join(GOLDEN, "run_element_expectation_golden.npz"), allow_pickle=False)
    gg = np.load(os.path.join(GOLDEN, "gene_stats_golden.npz"), allow_pickle=False)
    genes = pd.DataFrame(gg["frame_vals"], columns=[str(c) for c in gg["frame_cols"]])
    genes.insert(0, "GENE", [str(x) for x in gg["genes"]])
    genes.insert(0, "CHROM", gg["frame_chrom"])
    elts = pd.DataFrame(g["elt_vals"], columns=[str(c) for c in g["elt_cols"]])
    elts.insert(0, "ELT", [str(x) for x in g["elt_names"]])
    elts["FLAG"] = elts.FLAG.astype(bool)
    for ext in (".map", ".h5"):                    # the directory mirror and the reference's HDF5 container
        pre = str(tmp_path / ("pre" + ext))
        mapfile.write_frame(pre, "genic_model", genes)
        mapfile.write_frame(pre, "myelts", elts)
        panel_dir = tmp_path / "panels"
        panel_dir.mkdir(exist_ok=True)
        (panel_dir / "genes_CGC_ALL.txt").write_text("\n".join(str(x) for x in g["panel"]) + "\n")
        tab = pd.DataFrame(g["tab_vals"], columns=["OBS_SAMPLES", "OBS_SNV", "OBS_INDEL"], index=pd.Index([str(x) for x in g["tab_index"]], name="ELT"))
        monkeypatch.setattr(mutation_tools, "tabulate_mutations_in_element", lambda *a, **k: (tab.copy(), [str(x) for x in g["blacklist"]]))
        monkeypatch.setattr(transfer_tools, "_PANEL_DIRS", [str(panel_dir)])
        want = pd.DataFrame(g["out_vals"], columns=[str(c) for c in g["out_cols"]], index=[str(x) for x in g["out_index"]])
        for fused in (False, True):
            got = transfer_tools.run_element_region_model(os.path.join(GOLDEN, "gene_mutations.tsv"), "unused.bed", pre, "myelts",
                                                          scale_by_expectation=True, fused=fused)
            assert list(got.columns) == list(want.columns) and list(got.index) == list(want.index)
            for col in want.columns:
                tol = 1e-6 if col.startswith("PVAL") else 1e-12
                rel_close(got[col].values.astype(float), want[col].values, rtol=tol)
    out = capsys.readouterr().out
    assert "scaling by expected number of mutations" in out and "INDEL scale factor is: 1.80183481391" in out
    # a missing panel names the file and where it was looked for
    monkeypatch.setattr(transfer_tools, "_PANEL_DIRS", [str(tmp_path / "nowhere")])
    monkeypatch.setenv("DIG_DATA_DIR", str(tmp_path / "nowhere2"))
    with pytest.raises(FileNotFoundError) as err:
        transfer_tools.gene_panel("CGC_ALL")
    assert "genes_CGC_ALL.txt" in str(err.value) and "nowhere" in str(err.value) and "--panel-dir" in str(err.value)


def test_run_target_model_matches_reference(_gpu, tmp_path, monkeypatch, capsys):
    """`DigDriver.py targetDriver` (transfer_tools.py:876-967): panel genes, the three scale rules -- mutations inside the
    panel (CohortRun.panel_scale, the default), samples inside the panel, manual -- and the capped / keep-synonymous form.
    Golden: the reference function itself (tests/golden/make_golden.py::gen_run_target) on tests/golden/gene_mutations.tsv."""
    from digdriver_amd.driver_model import transfer_tools as tt
    from digdriver_amd.io import mapfile
    g = np.load(os.path.join(GOLDEN, "run_target_golden.npz"), allow_pickle=False)
    gg = np.load(os.path.join(GOLDEN, "gene_stats_golden.npz"), allow_pickle=False)
    genes = pd.DataFrame(gg["frame_vals"], columns=[str(c) for c in gg["frame_cols"]])
    for c in ("GENE_LENGTH", "R_SIZE", "R_OBS", "R_INDEL", "FLAG"):
        genes[c] = genes[c].astype(np.int64)
    genes.insert(0, "GENE", [str(x) for x in gg["genes"]])
    genes.insert(0, "CHROM", gg["frame_chrom"])
    panel_dir = tmp_path / "panels"
    panel_dir.mkdir()
    (panel_dir / "genes_PANELX.txt").write_text("\n".join(str(x) for x in g["panel"]) + "\n")
    (panel_dir / "genes_MSK_230.txt").write_text("\n".join(str(x) for x in g["panel"]) + "\n")
    monkeypatch.setattr(tt, "_PANEL_DIRS", [str(panel_dir)])
    attrs = {str(k): int(v) for k, v in zip(g["attr_names"], g["attr_vals"])}
    mut = os.path.join(GOLDEN, "gene_mutations.tsv")
    for ext in (".map", ".h5"):
        pre = str(tmp_path / ("target" + ext))
        mapfile.write_frame(pre, "genic_model", genes)
        mapfile.write_attrs(pre, N_SAMPLE_MSK_230=attrs["N_SAMPLE_PANELX"], **attrs)
        for tag, kw in (("mut", {}), ("sample", dict(scale_by_sample=True)), ("manual", dict(scale_factor=0.37)),
                        ("capped", dict(max_muts_per_sample=170, max_muts_per_gene_per_sample=3, drop_synonymous=False))):
            got = tt.run_target_model(mut, pre, panel="PANELX", **kw)
            cols = [str(c) for c in g[tag + "_cols"]]
            assert list(got.index) == [str(x) for x in g[tag + "_index"]], tag
            assert [c for c in got.columns if c != "CHROM"] == cols, tag
            vals = g[tag + "_vals"]
            for i, c in enumerate(cols):
                col = got[c].values.astype(float)
                if c.startswith(("OBS_", "N_SAMP_")) or c in ("GENE_LENGTH", "R_SIZE", "R_OBS", "R_INDEL", "FLAG"):
                    assert np.array_equal(col, vals[:, i]), (tag, c)
                else:
                    rel_close(col, vals[:, i], 1e-6)
        # the sites route's 'MSK_230' rule goes through the same counting (transfer_tools.py:1131-1149): samples inside the
        # panel over the pretrained cohort's
        run = tt.CohortRun(mut, pre)
        by_sample = run.panel_scale('MSK_230', tt.gene_panel('MSK_230'), (), by_sample=True)
        assert by_sample == 300 / attrs["N_SAMPLE_PANELX"]
    out = capsys.readouterr().out
    assert "Scaling factor is: 1.4218009478672986" in out and "300 211" in out       # the reference's own log lines
