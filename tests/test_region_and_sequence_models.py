"""CPU tests of the region-model and sequence-model host code: CNN mirror against the reference's seeded
forward (golden), BN folding, bin filters, GP invariants (no golden exists: gpytorch is absent/unpinned),
k-fold assembly, sequence-model training against the reference golden."""
import json
import os

import numpy as np
import pandas as pd
import pytest
import torch

from conftest import GOLDEN
from digdriver_amd.io import mapfile
from digdriver_amd.region_model import region_model_tools
from digdriver_amd.region_model.data_aux import dataset_generator as dg
from digdriver_amd.region_model.nets.cnn_predictors import SimpleMultiTaskResNet, flops_per_bin
from digdriver_amd.region_model.predict import r2_score
from digdriver_amd.region_model.trainers.gp_trainer import GPTrainer, SparseGP, run_gp
from digdriver_amd.sequence_model import sequence_tools as st


def _golden_net():
    d = np.load(os.path.join(GOLDEN, "cnn_forward_golden.npz"))
    B, L, T, C = [int(v) for v in d["shape"]]
    torch.manual_seed(0)
    net = SimpleMultiTaskResNet((B, L, T), C)
    g = torch.Generator().manual_seed(1)
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm1d):
            m.running_mean.copy_(torch.randn(m.num_features, generator=g) * 0.1)
            m.running_var.copy_(torch.rand(m.num_features, generator=g) + 0.5)
            m.weight.data.copy_(torch.rand(m.num_features, generator=g) + 0.5)
            m.bias.data.copy_(torch.randn(m.num_features, generator=g) * 0.1)
    return net.eval(), d


def _golden_net_735():
    """The T = 735 / 5-head golden of tests/golden/make_golden.py::gen_cnn_735 (the reference module's own outputs)."""
    d = np.load(os.path.join(GOLDEN, "cnn_forward_735_golden.npz"))
    B, L, T, C = [int(v) for v in d["shape"]]
    torch.manual_seed(3)
    net = SimpleMultiTaskResNet((B, L, T), C)
    g = torch.Generator().manual_seed(4)
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm1d):
            m.running_mean.copy_(torch.randn(m.num_features, generator=g) * 0.1)
            m.running_var.copy_(torch.rand(m.num_features, generator=g) + 0.5)
            m.weight.data.copy_(torch.rand(m.num_features, generator=g) + 0.5)
            m.bias.data.copy_(torch.randn(m.num_features, generator=g) * 0.1)
    return net.eval(), d


def test_cnn_matches_reference_forward_at_735_tracks():
    net, d = _golden_net_735()
    assert net.conv11.weight.double().sum().item() == float(d["first_conv_w_sum"])   # same init stream as the reference
    x = torch.tensor(d["x"].astype(np.float32))
    with torch.no_grad():
        outs, feats, _ = net(x)
        o2, f2, _ = net.fold_batchnorm().forward_gemm(x)
    for got_o, got_f in ((outs, feats), (o2, f2)):
        np.testing.assert_allclose(torch.stack(got_o).numpy(), d["outputs"], rtol=1e-4, atol=1e-5 * np.abs(d["outputs"]).max())
        np.testing.assert_allclose(torch.stack(got_f).numpy(), d["features"], rtol=1e-4, atol=1e-5 * np.abs(d["features"]).max())


def test_cnn_matches_reference_forward():
    net, d = _golden_net()
    assert sum(p.numel() for p in net.parameters()) == int(d["n_params"])
    assert net.conv11.weight.double().sum().item() == float(d["first_conv_w_sum"])   # same init stream as the reference
    with torch.no_grad():
        outs, feats, att = net(torch.tensor(d["x"]))
    assert att is None and len(outs) == 3 and feats[0].shape == (6, 16)
    np.testing.assert_allclose(torch.stack(outs).numpy(), d["outputs"], rtol=2e-5, atol=1e-6)    # fp32, other op order
    np.testing.assert_allclose(torch.stack(feats).numpy(), d["features"], rtol=2e-5, atol=1e-6)
    folded = net.fold_batchnorm()
    with torch.no_grad():
        o2, f2, _ = folded.forward_channels_first(torch.tensor(d["x"]).transpose(1, 2).contiguous())
    np.testing.assert_allclose(torch.stack(o2).numpy(), d["outputs"], rtol=1e-4, atol=1e-5)
    assert flops_per_bin(735, 1) == 2 * 223277840 and flops_per_bin(735, 37) - flops_per_bin(735, 1) == 2 * 36 * (13312 * 128 + 128 * 16 + 16)


def test_bin_filters_and_track_grammar():
    rng = np.random.default_rng(0)
    labels = rng.poisson(20, 1000).astype(float)
    mapp = rng.uniform(0, 1, 1000)
    idxs, below = dg.select_bins(mapp, labels, 0.5, 0.99)
    assert set(idxs) | set(below) == set(range(1000)) and not (set(idxs) & set(below))
    assert (mapp[idxs] >= 0.5).all() and (labels[idxs] <= np.quantile(labels, 0.99)).all()
    from scipy import stats
    np.testing.assert_allclose(dg.rank_quantiles(labels), stats.mstats.rankdata(labels) / len(labels))
    assert dg.load_track_selection(["# c\n", "3\n", "5:8\n", "\n"]) == [3, 5, 6, 7]
    with pytest.raises(ValueError):
        dg.load_track_selection(["4:2\n"])
    with pytest.raises(ValueError):
        dg.load_track_selection(["a\n"])
    folds = dg.split_folds(idxs, 5, seed=3)
    assert len(folds) == 5 and sorted(np.concatenate(folds).tolist()) == sorted(idxs.tolist())
    assert all(np.array_equal(a, b) for a, b in zip(folds, dg.split_folds(idxs, 5, seed=3)))


def test_sparse_gp_invariants():
    """No golden vectors exist for the GP (gpytorch absent and unpinned): check the SGPR restatement through
    invariants -- with m = n inducing points it is the exact GP; the bound is <= the exact log marginal
    likelihood; predictive variance is in [0, prior]; de-standardisation identity."""
    rng = np.random.default_rng(1)
    n, d = 120, 3
    X = rng.normal(size=(n, d))
    y = np.sin(X[:, 0]) + 0.3 * X[:, 1] + rng.normal(0, 0.1, n)
    Xt, yt = torch.tensor(X), torch.tensor(y)
    full = SparseGP(Xt, yt, n_inducing=n)
    with torch.no_grad():
        K = full.kernel(Xt, Xt) + full.noise * torch.eye(n, dtype=torch.float64)
        Lc = torch.linalg.cholesky(K)
        r = yt - full.mean_const
        exact = -0.5 * n * np.log(2 * np.pi) - torch.log(torch.diagonal(Lc)).sum() - 0.5 * (r @ torch.cholesky_solve(r[:, None], Lc)[:, 0])
        assert abs(-full.neg_bound_per_point().item() * n - exact.item()) < 1e-3 * abs(exact.item())
        sparse = SparseGP(Xt, yt, n_inducing=15)
        assert -sparse.neg_bound_per_point().item() * n <= exact.item() + 1e-9
        mean, std = sparse.predict(torch.tensor(rng.normal(size=(50, d))))
        assert (std >= 0).all() and (std ** 2 <= sparse.outputscale + 1e-9).all()
        # exact-GP predictive mean from the m = n model
        Xs = torch.tensor(rng.normal(size=(20, d)))
        mean_full, _ = full.predict(Xs)
        want = full.mean_const + full.kernel(Xs, Xt) @ torch.cholesky_solve(r[:, None], Lc)[:, 0]
        np.testing.assert_allclose(mean_full.numpy(), want.numpy(), rtol=1e-4, atol=1e-5)
    # trainer: standardisation / zero-feature drop / de-standardisation
    X2 = np.concatenate([X, np.zeros((n, 1))], axis=1)
    y2 = 40 + 7 * y
    tr = GPTrainer(torch.device("cpu"), (X2[:80], y2[:80]), (X2[80:100], y2[80:100]), (X2[100:], y2[100:]),
                   n_iter=25, n_inducing=30)
    assert tr.idx_feat.tolist() == [0, 1, 2]
    val, hld = tr.run()
    assert set(hld) == {'gp_mean', 'gp_std', 'r2', 'loss', 'params'} and hld['params'].shape == (3,)
    assert hld['r2'] > 0.8 and abs(hld['gp_mean'].mean() - y2[100:].mean()) < 3 and (hld['gp_std'] > 0).all()
    res, means, stds = run_gp(torch.device("cpu"), (X2[:80], y2[:80]), (X2[80:100], y2[80:100]), (X2[100:], y2[100:]),
                              n_runs=2, n_iter=10, n_inducing=20)
    assert len(res) == 2 and means.shape == (20,) and np.allclose(means, np.mean([r['gp_mean'] for r in res], axis=0))
    assert r2_score([1, 1, 1], [1, 2, 3]) == 0.0


def test_kfold_results_assembly(tmp_path):
    rng = np.random.default_rng(2)
    locs = np.array([[c, s * 10000, (s + 1) * 10000] for c in (1, 2) for s in range(30)])
    perm = rng.permutation(60)
    sup, sub = perm[:50], perm[50:]
    folds = np.array_split(sup, 5)
    truth = {}
    for k, f in enumerate(folds):
        p = str(tmp_path / ("gp_results_fold_%d.map" % k))
        mapfile.write_array(p, "COH/held-out/chr_locs", locs[f])
        mapfile.write_array(p, "COH/held-out/mappability", rng.uniform(0.5, 1, len(f)))
        mapfile.write_array(p, "COH/held-out/quantiles", rng.uniform(0, 1, len(f)))
        mapfile.write_array(p, "COH/held-out/y_true", rng.poisson(20, len(f)))
        runs = [rng.gamma(9, 3, len(f)) for _ in range(3)]
        for r, m in enumerate(runs):
            mapfile.write_array(p, "COH/held-out/%d/mean" % r, m)
            mapfile.write_array(p, "COH/held-out/%d/std" % r, m * 0.1)
        for i, b in enumerate(f):
            truth[tuple(locs[b])] = (np.mean([m[i] for m in runs]), False)
        q = str(tmp_path / ("sub_mapp_results_fold_%d.map" % k))
        mapfile.write_array(q, "COH/held-out/chr_locs", locs[sub])
        mapfile.write_array(q, "COH/held-out/mappability", np.full(len(sub), 0.2))
        mapfile.write_array(q, "COH/held-out/quantiles", np.full(len(sub), 0.5))
        mapfile.write_array(q, "COH/held-out/y_true", np.arange(len(sub)))
        mapfile.write_array(q, "COH/held-out/0/mean", np.full(len(sub), float(k)))
        mapfile.write_array(q, "COH/held-out/0/std", np.full(len(sub), 1.0 + k))
    df = region_model_tools.kfold_results(tmp_path, "COH")
    assert list(df.columns) == ['CHROM', 'START', 'END', 'Y_TRUE', 'Y_PRED', 'STD', 'MAPP', 'QUANT', 'FLAG']
    assert len(df) == 60 and df.index[0] == "chr1:0-10000" and df.CHROM.is_monotonic_increasing
    assert df.FLAG.sum() == 10 and np.allclose(df[df.FLAG].Y_PRED, 2.0) and np.allclose(df[df.FLAG].STD, 3.0)
    for (c, s, e), (m, fl) in truth.items():
        assert df.loc["chr%d:%d-%d" % (c, s, e), "Y_PRED"] == pytest.approx(m)
    # a bin present in two folds must trip the duplicate check (region_model_tools.py:186-189)
    p = str(tmp_path / "gp_results_fold_0.map")
    mapfile.write_array(p, "COH/held-out/chr_locs", np.vstack([locs[folds[0]][:-1], locs[folds[1]][:1]]))
    with pytest.raises(AssertionError):
        region_model_tools.kfold_results(tmp_path, "COH")


def test_sequence_model_matches_reference():
    g = json.load(open(os.path.join(GOLDEN, "subst_index.json")))
    assert st.mk_trans_idx(1, 1, False) == g["subst_idx"]
    assert st.mk_trans_idx(1, 1, True) == g["subst_idx_96"]
    assert list(st.mk_context_sequences(1, 1, False)) == g["context64"]
    e = st.mk_mutation_context(1, 1, False, return_df=True)
    assert [[a, b] for a, b in zip(e.MUT_TYPE, e.CONTEXT)] == g["model_rows"]
    d = np.load(os.path.join(GOLDEN, "sequence_model_golden.npz"))
    _, first = np.unique(d["dedup_key"], return_index=True)
    white = pd.DataFrame({"MUT_TYPE": d["mut_type"][first], "CONTEXT": d["context"][first]})
    _, cnt = st.sequence_model_counts(white)
    assert np.array_equal(cnt, d["out_count"])
    S = dict(zip(d["genome_ctx"], d["genome_counts"]))
    f192, f64 = st.train_sequence_model(None, None, S, counts=cnt)
    assert list(f192.MUT_TYPE) == list(d["out_mut_type"]) and list(f192.CONTEXT) == list(d["out_context"])
    assert np.array_equal(f192.FREQ.values, d["out_freq"])
    assert list(f64.index) == list(d["out64_context"])
    np.testing.assert_allclose(f64.FREQ.values, d["out64_freq"], rtol=1e-15)
    # shards add: counts of two halves sum to the whole (the all-reduce target)
    h = len(white) // 2
    assert np.array_equal(st.sequence_model_counts(white[:h])[1] + st.sequence_model_counts(white[h:])[1], cnt)
    # context counting
    seq = "ACGTNACGTACCA"
    got = st.count_sequence_context(seq, 1, 1)
    want = st.mk_context_sequences(1, 1)
    for i in range(1, len(seq) - 1):
        s = seq[i - 1:i + 2]
        if 'N' not in s:
            want[s] += 1
    assert got == want
    assert st.reverse_complement("AACGN") == "NCGTT"


def test_element_data_needs_matching_window_size(tmp_path):
    """sequence_tools.py:476: the genome-wide window counts an element-data container is started from must be on the
    requested window size."""
    import pandas as pd
    from digdriver_amd.io import mapfile
    from digdriver_amd.sequence_model import sequence_tools
    gc, ed = str(tmp_path / "genome_counts"), str(tmp_path / "element_data")
    idx = np.array([[1, 0, 10000], [1, 10000, 20000]], np.int32)
    frame = pd.DataFrame(np.ones((2, 64), np.int64), index=["chr1:0-10000", "chr1:10000-20000"],
                         columns=sequence_tools.mk_context_idx() if hasattr(sequence_tools, "mk_context_idx") else None)
    mapfile.write_array(gc, "idx", idx)
    mapfile.write_frame(gc, "all_window_genome_counts", frame)
    with pytest.raises(AssertionError):
        sequence_tools.initialize_nonc_data(ed, gc, 5000)
    sequence_tools.initialize_nonc_data(ed, gc, 10000)
    assert mapfile.has_key(ed, "window_10000/full_window_si_values") and mapfile.has_key(ed, "substitution_idx")


def test_sgpr_long_dimension_products_match_plain_matmuls():
    """The split products the SGPR uses when the training set is long (gp_trainer._outer_wide, _WideMatmul, _CrossTerm,
    _lower_solve) against plain matmuls / triangular solves, values and gradients."""
    import torch
    from digdriver_amd.region_model.trainers import gp_trainer as G
    torch.manual_seed(0)
    m, n, d = 12, 9001, 5                                  # n above the thresholds and not a multiple of the chunk count
    X = torch.randn(m, n, dtype=torch.float64, requires_grad=True)
    Y = torch.randn(m, n, dtype=torch.float64, requires_grad=True)
    a = G._outer_wide(X, Y)
    b = X @ Y.T
    assert torch.allclose(a, b, rtol=1e-12, atol=1e-10)
    ga = torch.autograd.grad(a.square().sum(), (X, Y))
    gb = torch.autograd.grad(b.square().sum(), (X, Y))
    assert all(torch.allclose(u, v, rtol=1e-10, atol=1e-8) for u, v in zip(ga, gb))
    S = torch.randn(m, m, dtype=torch.float64, requires_grad=True)
    w1, w2 = G._WideMatmul.apply(S, X), S @ X
    assert torch.equal(w1, w2)
    g1 = torch.autograd.grad((w1 * Y.detach()).sum(), (S, X))
    g2 = torch.autograd.grad((w2 * Y.detach()).sum(), (S, X))
    assert all(torch.allclose(u, v, rtol=1e-10, atol=1e-8) for u, v in zip(g1, g2))
    Z = torch.randn(m, d, dtype=torch.float64, requires_grad=True)
    T = torch.randn(n, d, dtype=torch.float64)
    c1, c2 = G._CrossTerm.apply(Z, T), Z @ T.T
    assert torch.equal(c1, c2)
    assert torch.allclose(torch.autograd.grad(c1.sin().sum(), Z)[0], torch.autograd.grad(c2.sin().sum(), Z)[0], rtol=1e-10, atol=1e-8)
    L = torch.linalg.cholesky(torch.eye(m, dtype=torch.float64) * 3 + 0.1 * (S.detach() @ S.detach().T))
    assert torch.allclose(G._lower_solve(L, X.detach()), torch.linalg.solve_triangular(L, X.detach(), upper=False), rtol=1e-10, atol=1e-10)


def test_nn_trainer_epoch_matches_reference_trainer():
    """One NNTrainer.train epoch + NNTrainer.test against the reference's own trainer (nn_trainer.py:40-141) run on the
    same seeded network, bins and visiting order (tests/golden/make_golden.py::gen_nn_training): summed per-task MSE,
    Adam(1e-3), train-mode BatchNorm, a last batch of two, features captured during the epoch, then eval-mode scores."""
    from torch import nn, optim
    from digdriver_amd.region_model.trainers.nn_trainer import NNTrainer
    d = np.load(os.path.join(GOLDEN, "nn_training_golden.npz"))
    T, L, C, n_train, n_val, bs = [int(v) for v in d["shape"]]
    x = torch.tensor(d["x"])

    class Store:                                   # BinTrackStore's interface on the CPU
        def batch(self, rows, channels_first=True, out_dtype="f32"):
            b = x[torch.as_tensor(np.asarray(rows), dtype=torch.long)]
            return b.transpose(1, 2).contiguous() if channels_first else b

    class Order:                                   # the reference DataLoader's shuffle, as recorded
        def permutation(self, rows):
            assert sorted(np.asarray(rows).tolist()) == list(range(n_train))
            return d["order"].copy()

    torch.manual_seed(5)
    net = SimpleMultiTaskResNet((n_train, L, T), C)
    assert net.conv11.weight.double().sum().item() == float(d["first_conv_w_sum_before"])
    tr = NNTrainer(net, optim.Adam(net.parameters(), lr=1e-3, amsgrad=False), nn.MSELoss(), bs, ["a", "b"], Store(),
                   np.arange(n_train), np.arange(n_train, n_train + n_val), list(d["labels"]), torch.device("cpu"))
    tr.rng = Order()
    losses, accs, feats, preds, true = tr.train(1)
    np.testing.assert_array_equal(tr.last_train_rows, d["order"])
    np.testing.assert_allclose(np.stack(true), d["labels"][:, d["order"]], rtol=0, atol=0)
    np.testing.assert_allclose(losses, d["train_losses"], rtol=2e-4)
    np.testing.assert_allclose(accs, d["train_accs"], rtol=2e-3, atol=1e-5)
    np.testing.assert_allclose(np.stack(preds), d["train_preds"], rtol=2e-3, atol=2e-4)
    np.testing.assert_allclose(np.stack(feats), d["train_features"], rtol=2e-3, atol=2e-3)   # (16 ReLU outputs per bin: some sit at the kink)
    assert abs(net.conv11.weight.double().sum().item() - float(d["first_conv_w_sum_after"])) < 2e-3
    bn_sum = sum(m.running_mean.double().sum().item() for m in net.modules() if isinstance(m, torch.nn.BatchNorm1d))
    assert abs(bn_sum - float(d["bn_running_mean_sum"])) < 1e-2 * max(1.0, abs(float(d["bn_running_mean_sum"])))
    vl, va, vfeats, vpreds, vtrue, _ = tr.test(1)
    np.testing.assert_allclose(vl, d["val_losses"], rtol=5e-3)
    np.testing.assert_allclose(np.stack(vpreds), d["val_preds"], rtol=5e-3, atol=5e-3)
    np.testing.assert_allclose(np.stack(vfeats), d["val_features"], rtol=5e-3, atol=5e-3)


def test_collapsed_contexts_k96_tables_match_reference():
    """The K = 96 vocabulary (collapse=True: sequence_tools.py:31-55,232-289): mk_mutation_context / mk_trans_idx /
    mk_context_sequences of this package against the reference's own (tests/golden/collapse_golden.json.gz), and the table
    expansion nb_model uses for collapsed S_prob (a purine-centred window reads its reverse complement's entry)."""
    import gzip
    import itertools
    import json
    from digdriver_amd.sequence_model import nb_model, sequence_tools as st
    g = json.loads(gzip.open(os.path.join(GOLDEN, "collapse_golden.json.gz")).read())
    assert [list(k) for k in st.mk_mutation_context(1, 1, collapse=True).keys()] == g["mutation_context_96"]
    assert st.mk_trans_idx(1, 1, collapse=True) == g["trans_idx_96"] and len(g["trans_idx_96"]) == 96
    assert list(st.mk_context_sequences(1, 1, collapse=True).keys()) == g["count_columns"]
    for n_up in (1, 2):
        run = g["runs"][str(n_up)]
        assert list(st.mk_context_sequences(n_up, n_up, collapse=True).keys()) == run["keys"]
        d_pr = dict(zip(run["keys"], run["d_pr"]))
        tab = nb_model._s_prob_table(d_pr, n_up, collapse=True)
        keys = ["".join(t) for t in itertools.product("ACGT", repeat=2 * n_up + 1)]
        for k, v in zip(keys, tab):
            assert v == d_pr[st.seq_to_context(k, baseix=n_up, collapse=True)]
    with pytest.raises(KeyError):
        nb_model._s_prob_table({"ACA": 1.0}, 1, collapse=True)
