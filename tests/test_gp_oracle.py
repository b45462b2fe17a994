"""GP calibration.  gpytorch is absent and unpinned, so parity with the reference's GP stays
"unpinned"; what is pinned here: the SGPR oracle is self-consistent (two independent numpy forms of Titsias' bound),
GPTrainer.standardize is sklearn's StandardScaler, and -- on the GPU -- SparseGP's bound, its gradients and its
predictive mean / standard deviation equal the oracle's at fixed hyper-parameters, at the reference's sizes too."""
import numpy as np
import pytest

from oracle import sgpr_oracle as S


def _problem(n, m, d, seed):
    rng = np.random.default_rng(seed)
    X = rng.normal(size=(n, d))
    w = rng.normal(size=d)
    y = np.sin(X @ w) + 0.3 * rng.normal(size=n)
    return X, y, X[:m].copy() + 0.05 * rng.normal(size=(m, d))


def test_oracle_forms_agree():
    X, y, Z = _problem(300, 30, 5, 0)
    for ls, os_, s2, c in ((1.3, 0.8, 0.2, 0.1), (0.7, 2.0, 0.05, -0.3)):
        a = S.bound_dense(X, y, Z, ls, os_, s2, c)
        b = S.bound_woodbury(X, y, Z, ls, os_, s2, c)
        assert abs(a - b) <= 1e-9 * abs(a), (a, b)
    # m = n inducing points at the data: the bound is the exact GP log marginal likelihood (jitter aside)
    Xs, ys, _ = _problem(60, 5, 3, 1)
    K = S.rbf(Xs, Xs, 1.1, 0.9) + 0.3 * np.eye(60)
    exact = -0.5 * (60 * np.log(2 * np.pi) + np.linalg.slogdet(K)[1] + ys @ np.linalg.solve(K, ys))
    assert abs(S.bound_woodbury(Xs, ys, Xs, 1.1, 0.9, 0.3, 0.0) - exact) < 1e-4
    mu, sd = S.predict(Xs, ys, Xs, Xs[:7], 1.1, 0.9, 0.3, 0.0)
    Ks = S.rbf(Xs[:7], Xs, 1.1, 0.9)
    np.testing.assert_allclose(mu, Ks @ np.linalg.solve(K, ys), atol=1e-4)


def test_standardize_is_sklearn_standard_scaler():
    """gp_trainer.py:107-120: StandardScaler().fit(train) applied to every set; y by the training mean and (population) std."""
    from sklearn.preprocessing import StandardScaler
    from digdriver_amd.region_model.trainers.gp_trainer import GPTrainer
    rng = np.random.default_rng(2)
    X = rng.normal(size=(200, 16)) * rng.uniform(0.1, 5, 16)
    X[:, 3] = 0.0                                   # dead feature (zero variance: left unscaled)
    Y = rng.poisson(30, 200).astype(float)
    Xt, Yt, scaler, ym, ys = GPTrainer.standardize(X, Y)
    sk = StandardScaler().fit(X)
    np.testing.assert_allclose(Xt, sk.transform(X), rtol=1e-13, atol=1e-13)
    assert ym == Y.mean() and ys == Y.std()
    X2 = rng.normal(size=(50, 16))
    np.testing.assert_allclose(GPTrainer.standardize(X2, Y[:50], scaler, ym, ys)[0], sk.transform(X2), rtol=1e-13, atol=1e-13)


def _model(X, y, Z, ls, os_, s2, c, dev):
    import math
    import torch
    from digdriver_amd.region_model.trainers.gp_trainer import SparseGP
    t = lambda a: torch.as_tensor(a, dtype=torch.float64, device=dev)
    gp = SparseGP(t(X), t(y), n_inducing=len(Z))
    inv = lambda v: math.log(math.expm1(v))
    with torch.no_grad():
        gp.inducing_points.copy_(t(Z))
        gp.raw_lengthscale.fill_(inv(ls))
        gp.raw_outputscale.fill_(inv(os_))
        gp.raw_noise.fill_(inv(s2 - 1e-4))
        gp.mean_const.fill_(c)
    return gp


@pytest.mark.gpu
def test_sparse_gp_bound_gradient_and_prediction_vs_oracle():
    import torch
    dev = torch.device("cuda:0")
    X, y, Z = _problem(2000, 50, 6, 3)
    ls, os_, s2, c = 1.4, 0.9, 0.25, 0.05
    gp = _model(X, y, Z, ls, os_, s2, c, dev)
    loss = gp.neg_bound_per_point()
    want = -S.bound_woodbury(X, y, Z, ls, os_, s2, c) / len(y)
    assert abs(loss.item() - want) <= 1e-10 * abs(want)
    # gradients: autograd against central differences of the oracle, through the softplus parametrisation
    loss.backward()
    sp = lambda r: np.log1p(np.exp(r))
    raw = {"ls": gp.raw_lengthscale.item(), "os": gp.raw_outputscale.item(), "s2": gp.raw_noise.item()}

    def f(ls_r, os_r, s2_r, c_, Z_):
        return -S.bound_woodbury(X, y, Z_, sp(ls_r), sp(os_r), sp(s2_r) + 1e-4, c_, jitter_abs=1e-6 * os_) / len(y)

    h = 1e-5
    fd = {"ls": (f(raw["ls"] + h, raw["os"], raw["s2"], c, Z) - f(raw["ls"] - h, raw["os"], raw["s2"], c, Z)) / (2 * h),
          "os": (f(raw["ls"], raw["os"] + h, raw["s2"], c, Z) - f(raw["ls"], raw["os"] - h, raw["s2"], c, Z)) / (2 * h),
          "s2": (f(raw["ls"], raw["os"], raw["s2"] + h, c, Z) - f(raw["ls"], raw["os"], raw["s2"] - h, c, Z)) / (2 * h),
          "c": (f(raw["ls"], raw["os"], raw["s2"], c + h, Z) - f(raw["ls"], raw["os"], raw["s2"], c - h, Z)) / (2 * h)}
    got = {"ls": gp.raw_lengthscale.grad.item(), "os": gp.raw_outputscale.grad.item(), "s2": gp.raw_noise.grad.item(),
           "c": gp.mean_const.grad.item()}
    for k in fd:
        assert abs(got[k] - fd[k]) <= 1e-6 * max(abs(fd[k]), 1e-3), (k, got[k], fd[k])
    for (i, j) in ((0, 0), (7, 3), (49, 5)):                     # a few inducing-point coordinates
        Zp, Zm = Z.copy(), Z.copy()
        Zp[i, j] += h
        Zm[i, j] -= h
        g_fd = (f(raw["ls"], raw["os"], raw["s2"], c, Zp) - f(raw["ls"], raw["os"], raw["s2"], c, Zm)) / (2 * h)
        assert abs(gp.inducing_points.grad[i, j].item() - g_fd) <= 1e-6 * max(abs(g_fd), 1e-4), (i, j)
    Xs = np.random.default_rng(9).normal(size=(500, 6))
    mu, sd = gp.predict(torch.as_tensor(Xs, dtype=torch.float64, device=dev))
    mu_w, sd_w = S.predict(X, y, Z, Xs, ls, os_, s2, c)
    np.testing.assert_allclose(mu.cpu().numpy(), mu_w, rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(sd.cpu().numpy(), sd_w, rtol=1e-7, atol=1e-10)


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_sparse_gp_at_reference_size_vs_oracle():
    """n = 150 000 training rows (the cap of gp_trainer.py:55), 16 features, m = 400 inducing points (the k-fold
    default, kfold_mutations_main.py:73)."""
    import torch
    dev = torch.device("cuda:0")
    X, y, Z = _problem(150_000, 400, 16, 5)
    X, Z = X * 0.35, Z * 0.35
    ls, os_, s2, c = 1.2, 0.7, 0.3, 0.02
    gp = _model(X, y, Z, ls, os_, s2, c, dev)
    with torch.no_grad():
        got = gp.neg_bound_per_point().item()
    want = -S.bound_woodbury(X, y, Z, ls, os_, s2, c) / len(y)
    assert abs(got - want) <= 1e-9 * abs(want), (got, want)
    Xs = np.random.default_rng(10).normal(size=(3000, 16)) * 0.35
    mu, sd = gp.predict(torch.as_tensor(Xs, dtype=torch.float64, device=dev))
    mu_w, sd_w = S.predict(X, y, Z, Xs, ls, os_, s2, c)
    np.testing.assert_allclose(mu.cpu().numpy(), mu_w, rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(sd.cpu().numpy(), sd_w, rtol=1e-6, atol=1e-9)


@pytest.mark.gpu
def test_rbf_cross_kernels_vs_numpy_and_autograd():
    """dig_rbf_cross / dig_rbf_backward (csrc/dig_gp.hip) through the autograd function the SGPR uses: the values against
    numpy's direct RBF, the gradients against torch autograd of the elementwise formulation (every feature count the
    dispatch covers at its ends, a ragged n, rows past one 16-row group)."""
    import torch
    from digdriver_amd.region_model.trainers import gp_trainer as G
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(11)
    for m, n, d in ((37, 9001, 16), (5, 8192, 1), (130, 8200, 32)):
        Z = torch.tensor(rng.normal(size=(m, d)), device=dev, requires_grad=True)
        X = torch.tensor(rng.normal(size=(n, d)), device=dev)
        ls = torch.tensor(0.9 + 0.2 * d ** 0.5, device=dev, dtype=torch.float64, requires_grad=True)
        osc = torch.tensor(1.7, device=dev, dtype=torch.float64, requires_grad=True)
        K = G._RbfCross.apply(Z, X, ls, osc)
        zn, xn = Z.detach().cpu().numpy(), X.cpu().numpy()
        d2 = ((zn[:, None, :] - xn[None, :, :]) ** 2).sum(-1)
        np.testing.assert_allclose(K.detach().cpu().numpy(), 1.7 * np.exp(-0.5 * d2 / float(ls.detach()) ** 2), rtol=1e-13, atol=1e-300)
        g = torch.tensor(rng.normal(size=(m, n)), device=dev)
        (K * g).sum().backward()
        got = [Z.grad.clone(), ls.grad.clone(), osc.grad.clone()]
        for t in (Z, ls, osc):
            t.grad = None
        ref = osc * torch.exp(-0.5 * ((Z[:, None, :] - X[None, :, :]) ** 2).sum(-1) / ls ** 2)
        (ref * g).sum().backward()
        for a, b in zip(got, (Z.grad, ls.grad, osc.grad)):
            np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=1e-10, atol=1e-12)


def test_gp_retry_ladder_follows_the_reference(monkeypatch, capsys):
    """run_gp's retry ladder (mutations_main.py:177-195): every inducing-point count gets gp_reruns attempts; an attempt fails on
    a RuntimeError of the fit (:185-188) or when the GP's VALIDATION R^2 is more than gp_delta below the CNN's (:189-191); after
    gp_reruns failures the count drops by 100 (:194); when it reaches 0 the fold fails (the caller retrains the CNN).  The fit
    itself is replaced by a script of outcomes: the sequence of inducing-point counts tried is what is checked."""
    import numpy as np
    from digdriver_amd.region_model.trainers import gp_trainer
    script, tried = [], []

    class Scripted:
        def __init__(self, device, train_tup, val_tup, heldout_tup=None, n_iter=50, n_inducing=500, seed=None, **kw):
            tried.append(n_inducing)

        def run(self):
            what = script.pop(0)
            if what == "error":
                raise RuntimeError("cholesky: not positive definite")
            r2 = {"low": 0.40, "ok": 0.48, "good": 0.9}[what]
            res = lambda: {"gp_mean": np.full(4, r2), "gp_std": np.ones(4), "r2": r2, "loss": 0.0, "params": np.zeros(3)}
            return res(), res()

    monkeypatch.setattr(gp_trainer, "GPTrainer", Scripted)
    tup = (np.zeros((4, 2)), np.zeros(4))
    # run 0: three failures at 400 (error, low R^2, error), two at 300, then R^2 0.48 >= 0.5 - 0.03 passes; run 1: first try
    script[:] = ["error", "low", "error", "low", "error", "ok", "good"]
    results, means, stds = gp_trainer.run_gp("cpu", tup, tup, tup, n_runs=2, n_inducing=400, gp_reruns=3, gp_delta=0.03, nn_r2=0.5)
    assert tried == [400, 400, 400, 300, 300, 300, 400] and not script
    assert [r["r2"] for r in results] == [0.48, 0.9] and np.allclose(means, (0.48 + 0.9) / 2)
    assert results[0]["val"]["r2"] == 0.48                              # the validation results of the same fit ride along
    # every count down to 100 fails three times: the fold fails (kfold_mutations_main.py:228 retrains the CNN)
    tried.clear()
    script[:] = ["low"] * 12
    with pytest.raises(AssertionError):
        gp_trainer.run_gp("cpu", tup, tup, tup, n_runs=1, n_inducing=400, gp_reruns=3, gp_delta=0.03, nn_r2=0.5)
    assert tried == [400] * 3 + [300] * 3 + [200] * 3 + [100] * 3
    assert "failed to reach minimal accuracy of 0.4700" in capsys.readouterr().out
