"""Sparse Gaussian-process calibration of the CNN's 16-d features: mean AND standard deviation per bin.

Mirror of GPTrainer / SparseGP (DIGDriver/region_model/trainers/gp_trainer.py:28-204).  The reference builds
the model from gpytorch (ExactGP + InducingPointKernel(ScaleKernel(RBFKernel)) + GaussianLikelihood) --
a third-party dependency that is neither vendored nor version-pinned, so there are no golden vectors for
this row ("parity unpinned").  What is restated here is the published algorithm those classes implement:
Titsias' SGPR collapsed bound with a constant mean, one shared RBF lengthscale, an output scale, Gaussian
noise (softplus-parametrised, noise >= 1e-4) and m inducing points initialised to the first m training
rows and optimised together with the hyper-parameters by Adam(lr = 0.8) for n_iter steps on
-bound / n; prediction returns the latent mean and standard deviation, de-standardised as
mean * y_std + y_mean and std * y_std (gp_trainer.py:190-192).

Everything runs in torch on the GPU (float64: the O(n m^2) work is tiny next to the CNN).
"""
import math

import numpy as np
import torch

from ..predict import r2_score

_JITTER = 1e-6


def _softplus_inv(v):
    return math.log(math.expm1(v))


def _lower_solve(L, B, scale=None):
    """L^-1 B for a lower-triangular [m, m] factor and a wide [m, n] right-hand side.  With n in the hundreds of thousands
    (the 150 000-row training cap of gp_trainer.py:55) rocBLAS' trsm cannot allocate its workspace
    (HIPBLAS_STATUS_ALLOC_FAILED); the m x m inverse is formed by one small triangular solve instead and applied as a
    GEMM, which is also what the MFMA units are good at.  m <= a few thousand and the factor carries jitter, so the
    explicit inverse is harmless in FP64."""
    if B.shape[-1] <= 4096:
        out = torch.linalg.solve_triangular(L, B, upper=False)
        return out if scale is None else out * scale
    Linv = torch.linalg.solve_triangular(L, torch.eye(L.shape[0], dtype=L.dtype, device=L.device), upper=False)
    return _WideMatmul.apply(Linv if scale is None else Linv * scale, B)


def _outer_wide(X, Y, chunks=64):
    """X @ Y.T for two [m, n] matrices with n >> m.  The product has only (m / 128)^2 output tiles, so a plain GEMM keeps a
    dozen of the 256 CUs busy (measured: 48 GFLOP in 24 ms); splitting the long dimension into batched products and
    summing them fills the chip."""
    m, n = X.shape
    if n < 8192:
        return X @ Y.T
    if n % chunks:                                        # a chunk count that divides n: padding copies both operands
        for cand in range(96, 31, -1):
            if n % cand == 0:
                chunks = cand
                break
    pad = (-n) % chunks
    if pad:
        X = torch.nn.functional.pad(X, (0, pad))
        Y = torch.nn.functional.pad(Y, (0, pad))
    Xc = X.view(m, chunks, -1).transpose(0, 1)           # [chunks, m, n / chunks]
    Yc = Y.view(Y.shape[0], chunks, -1).transpose(0, 1)
    return torch.bmm(Xc, Yc.transpose(1, 2)).sum(0)


class _WideMatmul(torch.autograd.Function):
    """S @ K for a small [m, m] S and a wide [m, n] K: the gradient with respect to S is a long-dimension product too."""

    @staticmethod
    def forward(ctx, S, K):
        ctx.save_for_backward(S, K)
        return S @ K

    @staticmethod
    def backward(ctx, g):
        S, K = ctx.saved_tensors
        return _outer_wide(g, K), S.T @ g


class _CrossTerm(torch.autograd.Function):
    """a @ b.T for a small [m, d] a (the inducing points) and a long [n, d] b (the training rows): the gradient with
    respect to a is g[m, n] @ b -- 150 000 deep with a 400 x 16 result, one workgroup's worth of output for a plain GEMM."""

    @staticmethod
    def forward(ctx, a, b):
        ctx.save_for_backward(a, b)
        return a @ b.T

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        da = _outer_wide(g, b.T.contiguous()) if ctx.needs_input_grad[0] else None
        db = g.T @ a if ctx.needs_input_grad[1] else None
        return da, db


class _GramAndProject(torch.autograd.Function):
    """(A A^T, A r) for the wide [m, n] factor A and the residual r [n].  As separate torch operators the backward costs a
    second long-dimension GEMM (the two operands of A A^T are the same tensor), three copies and two additions over the
    m x n matrix; by symmetry it is ONE product and one rank-1 update in place:  dA = (G + G^T) A + g (x) r."""

    @staticmethod
    def forward(ctx, A, r):
        ctx.save_for_backward(A, r)
        return _outer_wide(A, A), A @ r

    @staticmethod
    def backward(ctx, gB, gv):
        A, r = ctx.saved_tensors
        dA = (gB + gB.T) @ A
        dA.addr_(gv, r)
        dr = A.T @ gv if ctx.needs_input_grad[1] else None
        return dA, dr


class _RbfCross(torch.autograd.Function):
    """outputscale * exp(-|a_i - b_j|^2 / (2 lengthscale^2)) for the [m, d] inducing points a and a long [n, d] b that needs
    no gradient (the training rows): ONE pass writing the m x n matrix straight from the points (dig_rbf_cross); the backward is one pass too (dig_rbf_backward: W = g o K with the
    sums the scalar gradients need) plus the wide product W b.  From torch's elementwise operators the same took about
    ten passes over 0.5 GB each way -- half of a fit's time (rocprofv3: 474 launches per Adam step)."""

    @staticmethod
    def forward(ctx, a, b, lengthscale, outputscale):
        from ... import _lib
        ls, os_ = float(lengthscale), float(outputscale)
        a, b = a.contiguous(), b.contiguous()
        K = torch.empty((a.shape[0], b.shape[0]), dtype=a.dtype, device=a.device)
        with torch.cuda.device(a.device):
            _lib.call("dig_rbf_cross", _lib.dev_ptr(a), _lib.dev_ptr(b), a.shape[0], b.shape[0], a.shape[1], ls, os_,
                      _lib.dev_ptr(K), _lib.stream_ptr())
        ctx.save_for_backward(a, b, K)
        ctx.scalars = (ls, os_)
        return K

    @staticmethod
    def backward(ctx, g):
        from ... import _lib
        a, b, K = ctx.saved_tensors
        ls, os_ = ctx.scalars
        m, n = K.shape
        g = g.contiguous()
        W = torch.empty_like(K)
        chunks = (n + 1023) // 1024
        part = torch.empty((m, chunks, 2), dtype=K.dtype, device=K.device)
        with torch.cuda.device(K.device):
            _lib.call("dig_rbf_backward", _lib.dev_ptr(g), _lib.dev_ptr(K), m, n, ls, os_, _lib.dev_ptr(W), _lib.dev_ptr(part),
                      _lib.stream_ptr())
        rows = part[:, :, 0].sum(1)                                        # rowsum(W)
        da = (_outer_wide(W, b.T.contiguous()) - rows[:, None] * a) / (ls * ls) if ctx.needs_input_grad[0] else None
        d_ls = part[:, :, 1].sum() / ls ** 3 if ctx.needs_input_grad[2] else None
        d_os = rows.sum() / os_ if ctx.needs_input_grad[3] else None
        return da, None, d_ls, d_os


class SparseGP(torch.nn.Module):
    """SGPR with learnable inducing locations."""

    def __init__(self, train_x, train_y, n_inducing=2000):
        super().__init__()
        self.train_x, self.train_y = train_x, train_y
        self.raw_lengthscale = torch.nn.Parameter(torch.zeros((), dtype=train_x.dtype, device=train_x.device))
        self.raw_outputscale = torch.nn.Parameter(torch.zeros((), dtype=train_x.dtype, device=train_x.device))
        self.raw_noise = torch.nn.Parameter(torch.zeros((), dtype=train_x.dtype, device=train_x.device))
        self.mean_const = torch.nn.Parameter(torch.zeros((), dtype=train_x.dtype, device=train_x.device))
        self.inducing_points = torch.nn.Parameter(train_x[:n_inducing, :].clone())   # gp_trainer.py:39

    lengthscale = property(lambda self: torch.nn.functional.softplus(self.raw_lengthscale))
    outputscale = property(lambda self: torch.nn.functional.softplus(self.raw_outputscale))
    noise = property(lambda self: torch.nn.functional.softplus(self.raw_noise) + 1e-4)

    def kernel(self, a, b):
        if a.is_cuda and a.dtype == torch.float64 and b.shape[0] >= 8192 and not b.requires_grad and a.shape[1] <= 32:
            return _RbfCross.apply(a, b, self.lengthscale, self.outputscale)
        d2 = (a * a).sum(-1, keepdim=True) - 2.0 * _CrossTerm.apply(a, b) + (b * b).sum(-1)[None, :]
        return self.outputscale * torch.exp(-0.5 * d2.clamp_min(0.0) / self.lengthscale ** 2)

    def _factor(self):
        Z, X = self.inducing_points, self.train_x
        m = Z.shape[0]
        Kmm = self.kernel(Z, Z) + _JITTER * self.outputscale.detach() * torch.eye(m, dtype=Z.dtype, device=Z.device)
        L = torch.linalg.cholesky(Kmm)
        sig = torch.sqrt(self.noise)
        A = _lower_solve(L, self.kernel(Z, X), scale=1.0 / sig)        # [m, n]; the 1 / sigma rides on the m x m factor
        r = self.train_y - self.mean_const
        AAt, Ar = _GramAndProject.apply(A, r)
        self._trace_aat = torch.diagonal(AAt).sum()                    # = (A * A).sum() without another pass over m x n
        B = torch.eye(m, dtype=Z.dtype, device=Z.device) + AAt
        LB = torch.linalg.cholesky(B)
        c = torch.linalg.solve_triangular(LB, Ar[:, None], upper=False)[:, 0] / sig
        return L, A, LB, r, c

    def neg_bound_per_point(self):
        """-(collapsed SGPR bound) / n  == the reference's  -mll(model(X), y)  in train mode."""
        L, A, LB, r, c = self._factor()
        n = self.train_x.shape[0]
        s2 = self.noise
        bound = (-0.5 * n * math.log(2 * math.pi) - torch.log(torch.diagonal(LB)).sum() - 0.5 * n * torch.log(s2)
                 - 0.5 * (r @ r) / s2 + 0.5 * (c @ c)
                 - 0.5 * n * self.outputscale / s2 + 0.5 * self._trace_aat)
        return -bound / n

    @torch.no_grad()
    def predict(self, x, chunk=65536):
        L, A, LB, r, c = self._factor()
        means, stds = [], []
        for s in range(0, x.shape[0], chunk):
            xs = x[s:s + chunk]
            t1 = _lower_solve(L, self.kernel(self.inducing_points, xs))
            t2 = _lower_solve(LB, t1)
            means.append(self.mean_const + t2.T @ c)
            var = self.outputscale - (t1 * t1).sum(0) + (t2 * t2).sum(0)
            stds.append(torch.sqrt(var.clamp_min(0.0)))
        return torch.cat(means), torch.cat(stds)


    def predictor_state(self):
        """Everything predict() needs, without the training rows: the inducing points, the two m x m factors, the weight
        vector and three scalars (a few MB at m = 400) -- what a bin-sharded run broadcasts from the rank that fitted the
        model ("fit on rank 0 and broadcast")."""
        with torch.no_grad():
            L, A, LB, r, c = self._factor()
            scal = torch.stack([self.lengthscale, self.outputscale, self.noise, self.mean_const]).detach()
            # row-major copies: the factorisation routines hand back column-major factors, and a predictor that went through
            # broadcast_state (a flat buffer) is row-major -- with one layout both predict with the same kernels and bits
            return dict(Z=self.inducing_points.detach().clone().contiguous(), L=L.detach().contiguous(),
                        LB=LB.detach().contiguous(), c=c.detach().contiguous(), scalars=scal.contiguous())

    @staticmethod
    @torch.no_grad()
    def predict_from_state(state, x, chunk=65536):
        """predict() from a predictor_state(): latent mean and standard deviation at x [n, d] (standardised features)."""
        Z, L, LB, c = state["Z"], state["L"], state["LB"], state["c"]
        ls, os_, _, mean_const = state["scalars"].unbind(0)
        means, stds = [], []
        for s in range(0, x.shape[0], chunk):
            xs = x[s:s + chunk]
            d2 = (Z * Z).sum(-1, keepdim=True) - 2.0 * (Z @ xs.T) + (xs * xs).sum(-1)[None, :]
            K = os_ * torch.exp(-0.5 * d2.clamp_min(0.0) / ls ** 2)
            t1 = torch.linalg.solve_triangular(L, K, upper=False)
            t2 = torch.linalg.solve_triangular(LB, t1, upper=False)
            means.append(mean_const + t2.T @ c)
            stds.append(torch.sqrt((os_ - (t1 * t1).sum(0) + (t2 * t2).sum(0)).clamp_min(0.0)))
        if not means:
            return x.new_zeros(0), x.new_zeros(0)
        return torch.cat(means), torch.cat(stds)


class GPTrainer:
    samp_bound = int(1.5e5)      # gp_trainer.py:55: training-set cap

    def __init__(self, device, train_tup, val_tup, heldout_tup=None, n_iter=50, n_inducing=500, seed=None,
                 dtype=torch.float64, stats=None):
        """stats: (feature means, feature stds, y mean, y std) when they were formed elsewhere -- a bin-sharded run gets them
        from all ranks' rows by an all-reduce (parallel.standardisation_stats); default: from train_tup (gp_trainer.py:107-120)."""
        self.device, self.n_iter, self.n_inducing, self.dtype = device, n_iter, n_inducing, dtype
        self.org_train_x, self.org_train_y = train_tup[0], train_tup[1]
        self.org_val_x, self.org_val_y = val_tup[0], val_tup[1]
        self.train_meta, self.val_meta = train_tup[2:], val_tup[2:]
        if stats is not None:
            self.train_x, self.train_y, self.scaler, self.y_mean, self.y_std = self.standardize(
                train_tup[0], train_tup[1], (np.asarray(stats[0], float), np.asarray(stats[1], float)), float(stats[2]), float(stats[3]))
        else:
            self.train_x, self.train_y, self.scaler, self.y_mean, self.y_std = self.standardize(train_tup[0], train_tup[1])
        self.val_x, self.val_y, _, _, _ = self.standardize(val_tup[0], val_tup[1], self.scaler, self.y_mean, self.y_std)
        self.idx_feat = np.where(np.abs(self.train_x).mean(axis=0) > 0)[0]       # gp_trainer.py:79
        n = self.train_x.shape[0]
        if n > self.samp_bound:                                                  # :81-86
            pick = np.random.default_rng(seed).choice(n, size=self.samp_bound, replace=False)
            self.train_x, self.train_y = self.train_x[pick], self.train_y[pick]
            print('Reduced train set size from {} to {}, to stay within memory limits'.format(n, self.samp_bound))
        self.train_x = self.train_x[:, self.idx_feat]
        self.val_x = self.val_x[:, self.idx_feat]
        print('After zero features reduction feature vectors are now of size: {}'.format(self.train_x.shape[1]))
        self.held_x = self.held_y = None
        if heldout_tup is not None:
            self.org_ho_x, self.org_ho_y, self.ho_meta = heldout_tup[0], heldout_tup[1], heldout_tup[2:]
            self.held_x, self.held_y, _, _, _ = self.standardize(heldout_tup[0], heldout_tup[1], self.scaler,
                                                                 self.y_mean, self.y_std)
            self.held_x = self.held_x[:, self.idx_feat]

    @staticmethod
    def standardize(X, Y, scaler=None, y_mean=None, y_std=None):
        """gp_trainer.py:107-120 (sklearn StandardScaler: population std, zero-variance columns left unscaled)."""
        X, Y = np.asarray(X, float), np.asarray(Y, float)
        if scaler is None:
            mean, std = X.mean(axis=0), X.std(axis=0)
            std = np.where(std == 0, 1.0, std)
            scaler = (mean, std)
        if not y_mean:
            y_mean, y_std = Y.mean(), Y.std()
        return (X - scaler[0]) / scaler[1], (Y - y_mean) / y_std, scaler, y_mean, y_std

    def _t(self, a):
        return torch.as_tensor(np.ascontiguousarray(a), dtype=self.dtype, device=self.device)

    def train_model(self):
        model = SparseGP(self._t(self.train_x), self._t(self.train_y), n_inducing=self.n_inducing)
        opt = torch.optim.Adam(model.parameters(), lr=0.8)                       # gp_trainer.py:129
        for _ in range(self.n_iter):
            opt.zero_grad()
            loss = model.neg_bound_per_point()
            loss.backward()
            opt.step()
        return model

    def predict(self, model, x, y):
        mean, std = model.predict(self._t(x))
        y_t = self._t(y)
        var = std ** 2 + model.noise.detach()
        nll = 0.5 * (torch.log(2 * math.pi * var) + (y_t - mean) ** 2 / var).mean()
        return mean.cpu().numpy(), std.cpu().numpy(), float(nll)

    @staticmethod
    def get_results_dict(mean, std, r2, loss, params):
        return {'gp_mean': mean, 'gp_std': std, 'r2': r2, 'loss': loss, 'params': params}

    def run(self):
        """gp_trainer.py:173-204 -> (val_res, hld_res) dicts with keys gp_mean, gp_std, r2, loss, params."""
        model = self.model = self.train_model()
        params = np.array([model.lengthscale.item(), model.outputscale.item(), model.noise.item()])
        v_mean, v_std, v_loss = self.predict(model, self.val_x, self.val_y)
        v_r2 = r2_score(self.val_y, v_mean)
        print('Validation set R2: {}'.format(v_r2))
        val_res = self.get_results_dict(v_mean * self.y_std + self.y_mean, v_std * self.y_std, v_r2, v_loss, params)
        if self.held_x is None:
            return val_res, None
        h_mean, h_std, h_loss = self.predict(model, self.held_x, self.held_y)
        h_r2 = r2_score(self.held_y, h_mean)
        print('Held-out set R2: {}'.format(h_r2))
        return val_res, self.get_results_dict(h_mean * self.y_std + self.y_mean, h_std * self.y_std, h_r2, h_loss, params)


def run_gp(device, train_tup, val_tup, heldout_tup, n_runs=5, n_iter=50, n_inducing=400, gp_reruns=3, gp_delta=0.03,
           nn_r2=None, seed=0):
    """Retry ladder of OutputGenerator.run_gp / run_gp_iteration (mutations_main.py:174-247): `n_runs` GP fits;
    each fit is retried up to `gp_reruns` times on a numerical failure or when the GP's R^2 on the VALIDATION set
    (r2_score(val labels, gp_mean), :183) falls more than `gp_delta` below the CNN's (:189); after exhausting the retries
    the number of inducing points drops by 100 (:194).
    Returns (list of held-out result dicts, mean over runs of gp_mean, mean over runs of gp_std)."""
    results = []
    for run in range(n_runs):
        m, attempt, done = n_inducing, 0, None
        while done is None and m > 0:
            try:
                tr = GPTrainer(device, train_tup, val_tup, heldout_tup, n_iter=n_iter, n_inducing=m,
                               seed=seed + 1000 * run + attempt)
                val, hld = tr.run()
                if nn_r2 is not None and val['r2'] - nn_r2 < -gp_delta:
                    raise RuntimeError("GP run R2=%.4f, failed to reach minimal accuracy of %.4f" % (val['r2'], nn_r2 - gp_delta))
                hld = dict(hld, val=val)                                          # the validation results of the same fit
                done = hld
            except (RuntimeError, torch.linalg.LinAlgError) as exc:
                print('GP attempt failed: {}'.format(exc))
                attempt += 1
                if attempt >= gp_reruns:
                    attempt, m = 0, m - 100
        if done is None:
            raise AssertionError("GP failed for every inducing-point count")      # kfold_mutations_main.py:228
        results.append(done)
    means = np.mean([r['gp_mean'] for r in results], axis=0)                      # gp_trainer.py:247-261
    stds = np.mean([r['gp_std'] for r in results], axis=0)
    return results, means, stds


def run_gp_sharded(device, train_tup, val_tup, heldout_tup, group=None, src=0, n_runs=5, n_iter=50, n_inducing=400, gp_reruns=3,
                   gp_delta=0.03, nn_r2=None, seed=0, dtype=torch.float64):
    """run_gp for a bin-sharded run: the three tuples hold THIS RANK's rows (features [n_r, d], labels [n_r]).
      1. standardisation statistics of features and labels over all ranks' training rows: two rank-ordered sums
         (parallel.standardisation_stats -- the "GP/standardisation" collective: n, sum y, sum y^2, sum x, sum x^2);
      2. the training rows are all-gathered (<= 150 000 x 16 values) and rank `src` fits the SGPR;
      3. the fitted predictor (inducing points, two m x m factors, a vector, four scalars) is broadcast;
      4. every rank predicts mean / std of ITS validation and held-out rows; the R^2 of the retry ladder
         (mutations_main.py:177-195) is formed from the all-gathered predictions, and `src` takes the retry decision.
    Returns (results, means, stds) as run_gp does, for the rank's own held-out rows."""
    import torch.distributed as dist
    from ... import parallel
    on = parallel.collectives_on(group)
    rank = dist.get_rank(group) if on else 0
    tx, ty = np.asarray(train_tup[0], float), np.asarray(train_tup[1], float)
    comm = parallel.comm_device(device, group) if on else torch.device("cpu")
    mean, std, y_mean, y_std, _ = parallel.standardisation_stats(tx, ty, group, comm=comm)
    gather = lambda a: parallel.all_gather_rows(torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float64, device=comm), group).cpu().numpy()
    all_tx, all_ty = gather(tx), gather(ty)
    vy_all, hy_all = gather(np.asarray(val_tup[1], float)), gather(np.asarray(heldout_tup[1], float))
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=dtype, device=device)
    results = []
    for run in range(n_runs):
        m, attempt, done = n_inducing, 0, None
        while done is None and m > 0:
            state, idx_feat, failed = None, None, False
            if rank == src:
                try:
                    tr = GPTrainer(device, (all_tx, all_ty), (np.asarray(val_tup[0], float), np.asarray(val_tup[1], float)), None,
                                   n_iter=n_iter, n_inducing=m, seed=seed + 1000 * run + attempt, dtype=dtype,
                                   stats=(mean, std, y_mean, y_std))
                    model = tr.train_model()
                    state, idx_feat = model.predictor_state(), tr.idx_feat
                except (RuntimeError, torch.linalg.LinAlgError) as exc:
                    print('GP attempt failed: {}'.format(exc))
                    failed = True
            failed = parallel.broadcast_flag(failed, comm, src, group)
            if not failed:
                state, idx_feat = parallel.broadcast_state(state, idx_feat, device, comm, src, group, dtype)
                out = {}
                for name, tup in (('val', val_tup), ('held', heldout_tup)):
                    x = (np.asarray(tup[0], float) - mean) / std
                    mu_, sd_ = SparseGP.predict_from_state(state, t(x[:, idx_feat]))
                    out[name] = (mu_.cpu().numpy() * y_std + y_mean, sd_.cpu().numpy() * y_std)
                params = state["scalars"][:3].cpu().numpy()
                h_r2 = r2_score(hy_all, gather(out['held'][0]))
                v_r2 = r2_score(vy_all, gather(out['val'][0]))
                if nn_r2 is not None and v_r2 - nn_r2 < -gp_delta:
                    print('GP attempt failed: GP run R2=%.4f, failed to reach minimal accuracy of %.4f' % (v_r2, nn_r2 - gp_delta))
                    failed = True
            if failed:
                attempt += 1
                if attempt >= gp_reruns:
                    attempt, m = 0, m - 100
                continue
            done = dict(GPTrainer.get_results_dict(out['held'][0], out['held'][1], h_r2, float('nan'), params),
                        val=GPTrainer.get_results_dict(out['val'][0], out['val'][1], v_r2, float('nan'), params))
        if done is None:
            raise AssertionError("GP failed for every inducing-point count")
        results.append(done)
    means = np.mean([r['gp_mean'] for r in results], axis=0)
    stds = np.mean([r['gp_std'] for r in results], axis=0)
    return results, means, stds


def compute_pretrained(path, label, runs_num, held_key='held-out'):
    """GPTrainer.compute_pretrained (gp_trainer.py:247-261) on a results container: the held-out set of `label` with the
    mean over the first `runs_num` GP runs of the predicted mean and standard deviation.
    Returns (chr_locs, mappability, quantiles, y_true, means, stds)."""
    from ...io import mapfile
    base = '{}/{}/'.format(label, held_key)
    keys = set(mapfile.list_keys(path, base.rstrip('/')))
    assert 'chr_locs' in keys, 'Cannot compute pretrained model with no saved held-out set. Existing fields are: {}'.format(sorted(keys))
    get = lambda k: np.asarray(mapfile.read_array(path, base + k))
    means = np.mean([get('{}/mean'.format(i)) for i in range(int(runs_num))], axis=0)
    stds = np.mean([get('{}/std'.format(i)) for i in range(int(runs_num))], axis=0)
    return get('chr_locs'), get('mappability'), get('quantiles'), get('y_true'), means, stds
