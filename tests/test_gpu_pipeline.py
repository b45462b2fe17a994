"""GPU tests of the region-model inference pipeline (HBM-resident track matrix -> dig_gather_bins -> CNN on
PyTorch-ROCm) and of the command-line shims end to end."""
import os
import subprocess
import sys

import numpy as np
import pandas as pd
import pytest

from conftest import GOLDEN, ROOT

pytestmark = pytest.mark.gpu


def test_gather_plus_cnn_predict_matches_cpu_reference():
    import torch
    from digdriver_amd import _lib
    from digdriver_amd.region_model.data_aux.dataset_generator import BinTrackStore
    from digdriver_amd.region_model.predict import predict
    from test_region_and_sequence_models import _golden_net
    _lib.require_device()
    dev = torch.device("cuda:0")
    net, d = _golden_net()
    rng = np.random.default_rng(0)
    N, L, T = 300, 100, 40
    x = (np.round(rng.uniform(0, 1, (N, L, T)), 2) * 100).astype(np.float32)
    tracks = np.arange(4, 36)                       # the golden net has 32 input tracks
    rows = rng.permutation(N)[:170]
    labels = [rng.poisson(20, N).astype(float) for _ in range(3)]
    with torch.no_grad():
        want_o, want_f, _ = net(torch.tensor(x[rows][:, :, tracks]))
    store = BinTrackStore(torch.as_tensor(x, device=dev), tracks)
    preds, feats, r2 = predict(net.to(dev), store, rows, labels=labels, batch_size=64)
    np.testing.assert_allclose(preds, torch.stack(want_o).numpy(), rtol=2e-3, atol=1e-4)
    np.testing.assert_allclose(feats, torch.stack(want_f).numpy(), rtol=2e-3, atol=1e-4)
    assert preds.shape == (3, 170) and feats.shape == (3, 170, 16) and r2.shape == (3,)
    # int16 storage of the same matrix (values are integers <= 100) gives identical results
    store16 = BinTrackStore(torch.as_tensor(x.astype(np.int16), device=dev), tracks)
    assert torch.equal(store16.batch(rows), store.batch(rows))             # the gathered batch is bit-identical
    preds16, _, _ = predict(net, store16, rows, batch_size=64)
    np.testing.assert_allclose(preds16, preds, rtol=1e-4, atol=1e-5)       # MIOpen may pick another conv algorithm
    # reduced-precision inference (bf16 weights and activations; an option, the parity path is fp32): same shapes, close
    # values, and the caller's model is left in fp32
    pb, fb, _ = predict(net, store16, rows, batch_size=64, dtype=torch.bfloat16)
    assert pb.shape == preds.shape and fb.shape == feats.shape and np.isfinite(pb).all()
    assert np.abs(pb - preds).max() <= 0.05 * np.abs(preds).max() + 0.05
    assert next(net.parameters()).dtype == torch.float32


def test_cli_pretrain_then_driver(tmp_path):
    from test_gpu_host_mirror import _build_maps
    from digdriver_amd.io import mapfile
    pre, dat, d, g = _build_maps(tmp_path)
    env = dict(os.environ, PYTHONPATH=ROOT)
    # DigPretrain.py elementModel: writes the frame under save_key into the map
    subprocess.check_call([sys.executable, os.path.join(ROOT, "scripts", "DigPretrain.py"), "elementModel", pre, dat, "elts"],
                          env=env)
    frame = mapfile.read_frame(pre, "elts")
    col = {c: i for i, c in enumerate(d["out_cols"])}
    np.testing.assert_allclose(frame.P_SUM.values, d["out_vals"][:, col["P_SUM"]], rtol=1e-11)
    assert list(frame.ELT) == list(d["elt_names"])
    # a bed12 + mutation file for those elements, then DigDriver.py elementDriver
    bs, be = d["block_starts"], d["block_ends"]
    bed = tmp_path / "elts.bed"
    rng = np.random.default_rng(3)
    rows = []
    with open(bed, "w") as f:
        for name, c, st, s_, e_ in zip(d["elt_names"], d["elt_chrom"], d["elt_strand"], bs, be):
            s_, e_ = s_[s_ >= 0], e_[e_ >= 0]
            f.write("%d\t%d\t%d\t%s\t0\t%s\t%d\t%d\t.\t%d\t%s\t%s\n" % (
                c, s_[0], e_[-1], name, st, s_[0], s_[0], len(s_), ",".join(map(str, e_ - s_)) + ",",
                ",".join(map(str, s_ - s_[0])) + ","))
            for _ in range(rng.poisson(3)):
                b = rng.integers(0, len(s_))
                p = int(rng.integers(s_[b], e_[b]))
                rows.append((str(c), p, p + 1, "A", "T", "S%d" % rng.integers(0, 30), ".", "Noncoding", "A>T", "CAG"))
            if rng.uniform() < 0.3:
                p = int(s_[0])
                rows.append((str(c), p, p + 2, "AG", "A", "S%d" % rng.integers(0, 30), ".", "INDEL", "DEL", "."))
    mut = tmp_path / "cohort.tsv"
    pd.DataFrame(rows).to_csv(mut, sep="\t", header=False, index=False)
    out = tmp_path / "out"
    subprocess.check_call([sys.executable, os.path.join(ROOT, "scripts", "DigDriver.py"), "elementDriver", str(mut), pre,
                           "elts", "--f-bed", str(bed), "--scale-factor-manual", "0.002", "--scale-factor-indel-manual",
                           "0.0002", "--outdir", str(out), "--outpfx", "run1"], env=env)
    res = pd.read_csv(out / "run1.results.txt", sep="\t", index_col=0)
    assert res.index.name == "ELT" and len(res) == len(d["elt_names"])
    for c in ("OBS_SAMPLES", "OBS_SNV", "OBS_INDEL"):
        assert res[c].dtype.kind == "i"                                   # integer columns in the TSV
    assert {"EXP_SNV", "PVAL_SNV_BURDEN", "PVAL_SAMPLE_BURDEN", "EXP_INDEL", "PVAL_INDEL_BURDEN", "PVAL_MUT_BURDEN"} <= set(res.columns)
    from digdriver_amd.driver_model import transfer_tools as tt
    same = tt.run_element_region_model(str(mut), str(bed), pre, "elts", scale_factor=0.002, scale_factor_indel=0.0002,
                                       scale_by_expectation=False, fused=True)
    np.testing.assert_allclose(res.PVAL_MUT_BURDEN.values, same.PVAL_MUT_BURDEN.values, rtol=1e-12)
    assert res.PVAL_SNV_BURDEN.between(0, 1).all()
    # argument coupling of the reference CLI (DigDriver.py:68,78-80)
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "DigDriver.py"), "elementDriver", str(mut), pre, "elts",
                          "--f-bed", str(bed), "--scale-factor-manual", "1.0", "--outdir", str(out), "--outpfx", "x"],
                         env=env, capture_output=True, text=True)
    assert bad.returncode != 0 and "both" in (bad.stderr + bad.stdout)


def test_scale_factors_kernel_rank_ordered_sum():
    """dig_scale_factors: shards added in rank order, then the two divisions of transfer_tools.py:153-154 (bit-exact
    against the same sequence of IEEE operations in numpy)."""
    import torch
    from digdriver_amd import engine
    rng = np.random.default_rng(11)
    for world, C in ((1, 37), (3, 5), (8, 300)):
        parts = rng.gamma(3.0, 1e5, (world, 3, C))
        cj, cji = engine.scale_factors_from_parts(torch.as_tensor(parts, device="cuda:0"))
        tot = parts[0].copy()
        for r in range(1, world):
            tot += parts[r]
        assert np.array_equal(cj.cpu().numpy(), tot[1] / tot[0])
        assert np.array_equal(cji.cpu().numpy(), tot[2] / tot[0])


def test_kfold_training_writes_results_regionmodel_reads(tmp_path):
    """f4: CNN training + SGPR calibration end to end on a tiny synthetic track matrix whose labels are a smooth
    function of the tracks: the loss falls, every fold writes its results, and `kfold_results` assembles a
    region_params frame whose predictions correlate with the labels."""
    import torch
    from digdriver_amd.io import mapfile
    from digdriver_amd.region_model import kfold_mutations_main as kf, region_model_tools
    rng = np.random.default_rng(4)
    N, L, T = 900, 100, 6
    base = rng.uniform(0, 1, (N, 1, T))
    x = np.round(np.clip(base + 0.15 * rng.normal(size=(N, L, T)), 0, 1), 2) * 100
    y = np.rint(40 * base[:, 0, 0] + 25 * base[:, 0, 1] ** 2 + rng.normal(0, 1.0, N) + 5).clip(0)
    data = str(tmp_path / "train.map")
    mapfile.write_array(data, "x_data", x.astype(np.float32))
    mapfile.write_array(data, "idx", np.stack([np.ones(N, int), np.arange(N) * 10000, (np.arange(N) + 1) * 10000], 1))
    mapfile.write_array(data, "mappability", rng.uniform(0.3, 1.0, N))
    mapfile.write_array(data, "COHORT_A", y)
    args = kf.get_cmd_arguments("-c COHORT_A -d %s -o %s -k 2 -e 4 -b 64 -gp 2 -nd 100 -nt 30 -gd 0.5 -u --seed 1" % (data, tmp_path))
    out_dir = kf.main(args)
    assert os.path.isdir(out_dir)
    df = region_model_tools.kfold_results(out_dir, "COHORT_A")
    assert len(df) == N and {"Y_TRUE", "Y_PRED", "STD", "FLAG"} <= set(df.columns)
    ok = ~df.FLAG.values.astype(bool)
    r = np.corrcoef(df.Y_TRUE.values[ok], df.Y_PRED.values[ok])[0, 1]
    assert np.isfinite(df.Y_PRED.values).all() and (df.STD.values > 0).all()
    assert r > 0.5, r


def test_single_split_training_writes_pretrained_map(tmp_path):
    """a6 / f4, the mutations_main route: two reruns of CNN training + GP calibration on a tiny synthetic track matrix;
    every rerun writes gp_results_run{r}.h5 in GPTrainer.save_results' layout, the held-out windows accumulate in
    <label>.Pretrained.h5:region_params (a real HDF5 frame), the summaries are written, and the calibrated
    predictions correlate with the labels."""
    import glob
    from digdriver_amd.io import mapfile
    from digdriver_amd.region_model import mutations_main as mm
    rng = np.random.default_rng(6)
    N, L, T = 900, 100, 6
    base = rng.uniform(0, 1, (N, 1, T))
    x = np.round(np.clip(base + 0.15 * rng.normal(size=(N, L, T)), 0, 1), 2) * 100
    y = np.rint(40 * base[:, 0, 0] + 25 * base[:, 0, 1] ** 2 + rng.normal(0, 1.0, N) + 5).clip(0)
    data = str(tmp_path / "train.map")
    mapfile.write_array(data, "x_data", x.astype(np.float32))
    mapfile.write_array(data, "idx", np.stack([1 + np.arange(N) // 300, (np.arange(N) % 300) * 10000, (np.arange(N) % 300 + 1) * 10000], 1))
    mapfile.write_array(data, "mappability", rng.uniform(0.3, 1.0, N))
    mapfile.write_array(data, "COHORT_A", y)
    args = mm.get_cmd_arguments("-c COHORT_A -d %s -o %s -m 0.5 -cq 0.999 -e 4 -b 64 -gp 2 -re 2 -nd 100 -nt 30 -gd 0.5 -sm -st --seed 2"
                                % (data, tmp_path))
    out_dir = mm.main(args)
    for r in (0, 1):
        f = os.path.join(out_dir, "gp_results_run%d.h5" % r)
        keys = set(mapfile.list_keys(f, "COHORT_A/held-out"))
        assert {"nn_features", "y_true", "chr_locs", "mappability", "quantiles", "0", "1"} <= keys
        assert {"mean", "std", "params"} <= set(mapfile.list_keys(f, "COHORT_A/val/0"))
        assert 0.0 <= float(mapfile.read_attrs(f, "COHORT_A/held-out/1")["R2"]) <= 1.0
        assert os.path.exists(os.path.join(out_dir, "best_model_%d.pt" % r)) and os.path.exists(os.path.join(out_dir, "preds_%d.h5" % r))
    df = mapfile.read_frame(os.path.join(out_dir, "COHORT_A.Pretrained.h5"), "region_params")
    n_ho = len(np.load(os.path.join(out_dir, "ho_indices_0.npy")))
    assert list(df.columns) == mm.OutputGenerator.pretrained_cols and len(df) == 2 * n_ho       # the held-out set, once per rerun
    assert set(df.FOLD.unique()) == {0.0, 1.0} and (df.FLAG == 0).all() and (df.STD > 0).all()
    assert np.corrcoef(df.Y_TRUE.values, df.Y_PRED.values)[0, 1] > 0.5
    acc = float(open(os.path.join(out_dir, "COHORT_A_pretrained_accuracy.txt")).read())
    assert 0.25 < acc <= 1.0
    summ = open(os.path.join(out_dir, "COHORT_A_gp_runs_summary.csv")).read().splitlines()
    assert summ[0] == "fold,gp_run,nn_acc,val_acc,test_acc" and len(summ) == 5
    assert len(glob.glob(os.path.join(out_dir, "run_*"))) == 2                                  # run_params.txt, run_accuracies.csv


def test_scale_factors_local_equals_two_step_form():
    """dig_scale_factors_local == dig_scale_suffstats + dig_scale_factors(world = 1), bit for bit."""
    import torch
    from digdriver_amd import engine
    rng = np.random.default_rng(12)
    for N, C in ((5000, 37), (777, 3), (64, 300)):
        mu = torch.as_tensor(rng.gamma(9.0, 3.0, (N, C)), device="cuda:0")
        fl = torch.as_tensor((rng.uniform(size=(N, C)) < 0.1).astype(np.uint8), device="cuda:0")
        ns = torch.as_tensor(rng.gamma(3.0, 1e5, C), device="cuda:0")
        ni = torch.as_tensor(rng.gamma(3.0, 1e4, C), device="cuda:0")
        cj, cji, tot = engine.scale_factors_local(mu, fl, ns, ni)
        tot2 = engine.scale_suffstats(mu, fl)
        cj2, cji2 = engine.scale_factors_from_parts(torch.stack([tot2, ns, ni]).unsqueeze(0).contiguous())
        assert torch.equal(tot, tot2) and torch.equal(cj, cj2) and torch.equal(cji, cji2)


def test_plan_ring_two_batches_in_flight():
    """engine.PlanRing: two batches (different elements, observations and scale factors) alternating on two streams give,
    pass for pass, the bits of the same plans run one after the other on one stream."""
    import torch
    sys.path.insert(0, ROOT)
    from bench import make_workload
    torch_dev = torch.device("cuda:0")
    from digdriver_amd import engine
    plans, cjs = [], []
    for seed, E in ((21, 3000), (22, 4100)):
        w = make_workload(n_bins=5000, n_elements=E, n_cohorts=37, seed=seed)
        td = {k: torch.as_tensor(v, device=torch_dev) for k, v in w.items() if isinstance(v, np.ndarray)}
        plans.append(engine.PipelinePlan(td["bin_mu"], td["bin_std"], td["bin_y"], td["bin_flag"], td["bin_ctx"], td["ov_ptr"],
                                         td["ov_idx"], td["L"], td["strand_minus"], td["d_pr"], td["obs_snv"], td["obs_samples"],
                                         td["obs_indel"]))
        cjs.append((td["cj"], td["cj_indel"]))
    want = []
    for plan, (cj, cji) in zip(plans, cjs):            # one after the other, current stream
        acc, st = plan.run(cj, cji)
        torch.cuda.synchronize()
        want.append(({k: v.clone() for k, v in acc.items()}, st.clone()))
        st.fill_(-5.0)
        acc["P"].fill_(-5.0)
    ring = engine.PlanRing(plans)
    for k in range(6):
        ring.run(k, *cjs[k % 2])
    ring.synchronize()
    for plan, (acc_w, st_w) in zip(plans, want):
        assert torch.equal(torch.nan_to_num(plan.stats, nan=-7.0), torch.nan_to_num(st_w, nan=-7.0))
        for name in acc_w:
            assert torch.equal(plan.acc[name], acc_w[name]), name
