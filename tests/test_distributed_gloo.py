"""world_size-2 tests of the N > 1 path on CPU (gloo): shard plan (ownership + halo), rank-ordered exchange of the
per-cohort sufficient statistics, result gather.  The HIP kernels themselves are covered by the -m gpu tests; this
file checks that the sharded inputs they would receive reproduce the single-process problem exactly."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from bench import make_workload
from digdriver_amd import parallel


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_plan_shards_covers_problem_without_remote_bins():
    w = make_workload(n_bins=2000, n_elements=1500, n_cohorts=2, seed=5)
    for world in (1, 2, 8):
        plans = parallel.plan_shards(w["ov_ptr"], w["ov_idx"], 2000, world)
        owned = np.concatenate([p["elements"] for p in plans])
        assert sorted(owned.tolist()) == list(range(1500))                    # every element exactly once
        for p, (lo, hi) in zip(plans, parallel.bin_ranges(2000, world)):
            rows = p["bin_rows"]
            assert np.all(np.diff(rows) > 0) and p["n_halo"] == len(rows) - (hi - lo)
            for j, e in enumerate(p["elements"]):
                glob = w["ov_idx"][w["ov_ptr"][e]:w["ov_ptr"][e + 1]]
                loc = p["ov_idx"][p["ov_ptr"][j]:p["ov_ptr"][j + 1]]
                assert np.array_equal(rows[loc], glob)                         # local CSR points at the same bins
                assert lo <= glob[0] < hi                                      # owner holds the first bin
        if world > 1:
            assert sum(p["n_halo"] for p in plans) < 0.05 * 2000               # halo is a few boundary bins


def test_shard_generator_equals_sliced_global_problem():
    """bench.make_shard_workload(rank, world) -- what a rank of the N > 1 bench builds, from the global bin tables and the
    element blocks that hold its elements only -- is exactly parallel.shard_inputs of the whole problem, for every rank of
    2- and 8-rank plans (the element blocks are drawn whole from their own seeded streams); plan_shards(only_rank=r) is
    entry r of the full list."""
    import bench
    n_bins, E, C = 6400, 20_011, 2                          # three element blocks, shard boundaries inside blocks
    w = make_workload(n_bins, E, C, seed=5)
    for world in (1, 2, 8):
        plans = parallel.plan_shards(w["ov_ptr"], w["ov_idx"], n_bins, world)
        for r in range(world):
            one = parallel.plan_shards(w["ov_ptr"], w["ov_idx"], n_bins, world, only_rank=r)
            assert len(one) == 1 and all(np.array_equal(np.asarray(one[0][k]), np.asarray(plans[r][k])) for k in plans[r])
            want = parallel.shard_inputs(w, plans[r], world)
            got, plan = bench.make_shard_workload(r, world, n_bins, E, C, seed=5)
            assert set(want) <= set(got)
            for k in want:
                assert np.array_equal(np.asarray(want[k]), np.asarray(got[k])), (world, r, k)
            for k in plans[r]:
                assert np.array_equal(np.asarray(plans[r][k]), np.asarray(plan[k])), (world, r, k)


def _worker(rank, world, port, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        w = make_workload(n_bins=1200, n_elements=900, n_cohorts=3, seed=9)   # identical on every rank
        plans = parallel.plan_shards(w["ov_ptr"], w["ov_idx"], 1200, world)
        lo, hi = parallel.bin_ranges(1200, world)[rank]
        # per-shard sufficient statistics over the OWNED bins only (the halo belongs to a neighbour)
        unflag = (w["bin_flag"][lo:hi] == 0)
        local_exp = torch.tensor((w["bin_mu"][lo:hi] * unflag).sum(axis=0))
        share = (hi - lo) / 1200.0
        local_snv = torch.tensor(w["n_snv_obs"] * share)
        local_ind = torch.tensor(w["n_ind_obs"] * share)
        cj, cji = parallel.scale_factors(local_exp, local_snv, local_ind)
        # bit-identical on all ranks
        both = [torch.empty_like(cj) for _ in range(world)]
        dist.all_gather(both, cj)
        assert all(torch.equal(both[0], b) for b in both)
        # equals the rank-ordered sum computed by hand from the global arrays
        want = torch.zeros(3, dtype=torch.float64)
        for r, (a, b) in enumerate(parallel.bin_ranges(1200, world)):
            want = want + torch.tensor((w["bin_mu"][a:b] * (w["bin_flag"][a:b] == 0)).sum(axis=0)) if r else \
                torch.tensor((w["bin_mu"][a:b] * (w["bin_flag"][a:b] == 0)).sum(axis=0))
        n_tot = sum(torch.tensor(w["n_snv_obs"] * ((b - a) / 1200.0)) for a, b in parallel.bin_ranges(1200, world))
        assert torch.equal(cj, n_tot / want)
        np.testing.assert_allclose(cj.numpy(), w["cj"], rtol=1e-3)
        # sequence-model style integer counts: exact under the rank-ordered sum
        cnt = torch.arange(192, dtype=torch.int64) * (rank + 1)
        tot = parallel.rank_ordered_sum(cnt)
        assert torch.equal(tot, torch.arange(192, dtype=torch.int64) * sum(range(1, world + 1)))
        # sharded "results" gathered to rank 0 reproduce the global element order
        mine = torch.tensor(plans[rank]["elements"], dtype=torch.float64)[:, None] * torch.ones(1, 4, dtype=torch.float64)
        allrows = parallel.gather_to_rank0(mine)
        if rank == 0:
            ids = allrows[:, 0].long().numpy()
            assert sorted(ids.tolist()) == list(range(900))
            np.save(os.path.join(tmp, "ok.npy"), ids)
        else:
            assert allrows is None
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_exchange_gloo(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert os.path.exists(tmp_path / "ok.npy")


def _grad_worker(rank, world, port, tmp):
    """Data-parallel CNN step semantics: each rank's mean-loss gradient on its half of a batch, averaged with the flat
    all-reduce, equals the single-process gradient of the mean loss over the whole batch."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(0)
        net = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.ReLU(), torch.nn.Linear(5, 1)).double()
        x, y = torch.randn(8, 6, dtype=torch.float64), torch.randn(8, 1, dtype=torch.float64)
        full = torch.nn.functional.mse_loss(net(x), y)
        want = torch.autograd.grad(full, list(net.parameters()))
        net.zero_grad()
        torch.nn.functional.mse_loss(net(x[rank::world]), y[rank::world]).backward()
        parallel.average_gradients(list(net.parameters()))
        for p, w_ in zip(net.parameters(), want):
            assert torch.allclose(p.grad, w_, rtol=1e-12, atol=1e-14)
        if rank == 0:
            np.save(os.path.join(tmp, "grad_ok.npy"), np.ones(1))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_gradient_average_gloo(tmp_path):
    port = _free_port()
    mp.spawn(_grad_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert os.path.exists(tmp_path / "grad_ok.npy")


def _visit_worker(rank, world, port, tmp):
    """What the data-parallel CNN epoch hands to the GP: per-row outputs of batch[rank::world] slices, re-assembled in the
    global visiting order on every rank; BatchNorm running statistics taken from rank 0; decisions taken on rank 0."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n_rows, bs = 37, 8                                    # ragged last batch, odd batch size per rank
        order = np.random.default_rng(4).permutation(1000)[:n_rows]
        mine = np.concatenate([order[j:j + bs][rank::world] for j in range(0, n_rows, bs)])
        assert np.array_equal(order[parallel.strided_positions(n_rows, bs, rank, world)], mine)
        feats = torch.tensor(mine, dtype=torch.float32)[:, None] * torch.arange(1, 17, dtype=torch.float32)[None, :]
        whole = parallel.gather_visiting_order(feats, n_rows, bs)
        assert whole.shape == (n_rows, 16)
        assert torch.equal(whole[:, 0], torch.tensor(order, dtype=torch.float32))          # every row, global order
        assert torch.equal(whole[:, 15], torch.tensor(order, dtype=torch.float32) * 16)
        bn = torch.nn.BatchNorm1d(4)
        bn.running_mean.fill_(float(rank + 1))
        parallel.broadcast_module_buffers(bn, 0)
        assert torch.equal(bn.running_mean, torch.ones(4))
        assert parallel.broadcast_flag(rank == 0, torch.device("cpu")) is True             # rank 0 decides
        assert parallel.broadcast_flag(rank != 0, torch.device("cpu")) is False
        if rank == 0:
            np.save(os.path.join(tmp, "visit_ok.npy"), np.ones(1))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_feature_gather_gloo(tmp_path):
    port = _free_port()
    mp.spawn(_visit_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert os.path.exists(tmp_path / "visit_ok.npy")


def _chunk_worker(rank, world, port, tmp):
    """Bin-sharded scale factors through canonical chunks: every rank sums the chunks it owns (one fixed order per chunk),
    the chunk sums are all-gathered and added first to last -> the SAME bits as the single-process evaluation."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        w = make_workload(n_bins=2003, n_elements=700, n_cohorts=3, seed=13)
        edges = parallel.canonical_chunks(2003)

        def chunk_sum(rows_lo, rows_hi):          # the fixed per-chunk order of this test: plain row order
            acc = torch.zeros(3, dtype=torch.float64)
            for r in range(rows_lo, rows_hi):
                acc = acc + torch.tensor(w["bin_mu"][r] * (w["bin_flag"][r] == 0))
            return acc

        plans = parallel.plan_shards(w["ov_ptr"], w["ov_idx"], 2003, world)
        sh = parallel.shard_inputs(w, plans[rank], world)
        n_own = parallel.N_CHUNKS // world
        cr = sh["chunk_rows"]
        rows = plans[rank]["bin_rows"]
        part = torch.stack([chunk_sum(int(rows[cr[j]]), int(rows[cr[j + 1] - 1]) + 1) if cr[j + 1] > cr[j] else torch.zeros(3, dtype=torch.float64)
                            for j in range(n_own)] + [torch.tensor(sh["n_snv_obs"]), torch.tensor(sh["n_ind_obs"])])
        cj, cji = parallel.chunked_scale_factors_reference(part, n_own)
        # single-process evaluation of the same definition
        e = torch.zeros(3, dtype=torch.float64)
        for j in range(parallel.N_CHUNKS):
            e = e + chunk_sum(int(edges[j]), int(edges[j + 1]))
        assert torch.equal(cj, torch.tensor(w["n_snv_obs"]) / e) and torch.equal(cji, torch.tensor(w["n_ind_obs"]) / e)
        if rank == 0:
            np.save(os.path.join(tmp, "chunk_ok.npy"), np.ones(1))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world", [2, 4])
def test_chunked_scale_factors_gloo(tmp_path, world):
    """(world 4: every rank owns 16 of the 64 canonical chunks; the scale factors still have the single-process bits)"""
    port = _free_port()
    mp.spawn(_chunk_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    assert os.path.exists(tmp_path / "chunk_ok.npy")


def _tiles_worker(rank, world, port, tmp):
    """Per-base route sharded by bins (parallel.ShardedTiles) over a real 2-process group, CPU: what a rank would hand to
    the tile kernels -- its slab of the packed genome, shifted bin coordinates, its mutations -- evaluated with the ORACLE
    (the checker) equals the oracle on the whole genome for the rank's bins; then the one exchange of the route, the
    Benjamini-Hochberg pass over all ranks' p-values (all_gather_rows), against the single-process q-values."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from test_gpu_tiles import _tile_problem
        from digdriver_amd.sequence_model import nb_model
        from oracle import dig_oracle as O
        seqs, genome, chroms, starts, ends, S, mu, sg, mc, ms, me, co = _tile_problem(C=2)
        C, R, B = 2, len(chroms), 50
        sh = parallel.ShardedTiles(genome, chroms, starts, ends, S, mu, sg, mc, ms, me, co, B, None, rank, world)
        assert (sh.lo, sh.hi) == parallel.shard_rows(R, rank, world)

        def decode(g, c):
            w = g.words
            nib = np.stack([(w >> np.uint32(4 * k)) & np.uint32(15) for k in range(8)], 1).reshape(-1)[8:]
            o, n = int(g.offsets[c]), int(g.lengths[c])
            return "".join("ACGTN"[x] for x in nib[o:o + n])

        slab_seq = {n: decode(sh.genome, i) for i, n in enumerate(sh.genome.names)}
        T = sh.n_tiles
        assert T == 20                                           # 1000 positions per bin / 50
        pval = np.full((C, sh.hi - sh.lo, T), np.nan)
        nval = np.zeros(sh.hi - sh.lo, np.int32)
        smc, sms, sme, sco = sh.mut
        for j in range(sh.hi - sh.lo):
            name = sh.chroms[j]
            for c in range(C):
                here = (smc == name) & (sco == c)
                got = O.apply_nb_to_region(slab_seq[name], S[c], int(sh.starts[j]), int(sh.ends[j]), sh.mu[c, j], sh.sigma[c, j],
                                           sms[here], B)
                there = (mc == name) & (co == c)
                want = O.apply_nb_to_region(seqs[name].upper(), S[c], int(starts[sh.lo + j]), int(ends[sh.lo + j]), mu[c, sh.lo + j],
                                            sg[c, sh.lo + j], ms[there], B)
                for a, b in zip(got, want):
                    if a is got[1]:
                        assert np.array_equal(a + sh.reg_shift[j], b)            # positions: shifted back
                    else:
                        assert np.array_equal(a, b, equal_nan=True)              # p-values, counts, expectations, pt: same bits
                pval[c, j, :len(got[0])] = got[0]
                nval[j] = len(got[0])
        sh.result = dict(pval=torch.tensor(pval), n_valid=torch.tensor(nval))
        for fill in (None, 0.5):          # as computed (the all-N bin's NaN p-values make every q NaN, as in statsmodels); NaN-free
            if fill is not None:
                sh.result["pval"] = torch.nan_to_num(sh.result["pval"], nan=fill)
            for c in range(C):
                q = sh.q_values(c)                                    # the collective
                mine, mask = sh.valid_pvalues(c)
                everything = parallel.all_gather_rows(mine)
                want = nb_model.get_q_vals(everything.numpy())
                before = int(parallel.all_gather_rows(torch.tensor([mine.numel()]))[:rank].sum())
                assert np.array_equal(q[mask].numpy(), want[before:before + mine.numel()], equal_nan=True)
                assert fill is None or np.isfinite(q[mask].numpy()).all()
                assert torch.isnan(q[~mask]).all()
            q_all = sh.q_values_all()                                 # every cohort through ONE sample sort
            assert np.array_equal(q_all.numpy(), np.stack([sh.q_values(c).numpy() for c in range(C)]), equal_nan=True)
        if rank == 0:
            np.save(os.path.join(tmp, "tiles_ok.npy"), np.ones(1))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("world", [2, 4])
def test_sharded_tiles_gloo(tmp_path, world):
    port = _free_port()
    mp.spawn(_tiles_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    assert os.path.exists(tmp_path / "tiles_ok.npy")


def _sample_sort_worker(rank, world, port, tmp):
    """parallel.sample_sort_q_values over a real process group: ranks with very different numbers of p-values (one with none),
    heavy ties (also across the splitters), zeros and ones, a cohort with a NaN (every q of that cohort NaN, as statsmodels), a cohort
    whose values all sit on one rank's side of every splitter -- against the single-process Benjamini-Hochberg pass over the
    concatenation of all ranks' values, bit for bit."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from digdriver_amd.sequence_model import nb_model
        C = 5
        sizes = [0, 4001, 37, 12345][:world] if world > 2 else [2500, 0]
        rng_all = [np.random.default_rng(100 + r) for r in range(world)]
        parts = []
        for r in range(world):
            g, m = rng_all[r], sizes[r]
            a = np.empty((C, m))
            a[0] = g.random(m) ** 3
            a[1] = g.choice(np.array([0.0, 1e-9, 0.25, 0.5, 1.0]), m)                       # five values: every splitter is a tie
            a[2] = np.minimum(1.0, g.exponential(0.3, m))
            a[3] = g.random(m) * 1e-3 + r                                                 # rank r's values lie in [r, r + 0.001): already apart
            a[4] = g.random(m)
            if m and r == max(k for k in range(world) if sizes[k]):
                a[4, m // 2] = np.nan
            parts.append(a)
        whole = np.concatenate(parts, axis=1)
        want = np.stack([nb_model.get_q_vals(whole[c]) for c in range(C)])
        before = int(sum(sizes[:rank]))
        got = parallel.sample_sort_q_values(torch.from_numpy(parts[rank].copy()), None, samples=8).numpy()
        assert got.shape == parts[rank].shape
        assert np.array_equal(got, want[:, before:before + sizes[rank]], equal_nan=True)
        assert np.isnan(want[4]).all() and not np.isnan(want[:4]).any()
        if rank == 0:
            np.save(os.path.join(tmp, "ssort_ok.npy"), np.ones(1))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("world", [2, 4])
def test_sample_sort_q_values_gloo(tmp_path, world):
    port = _free_port()
    mp.spawn(_sample_sort_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    assert os.path.exists(tmp_path / "ssort_ok.npy")


class _HostStore:
    """Stand-in for BinTrackStore on CPU (the HIP gather is covered by the -m gpu tests): same interface, numpy gather."""

    def __init__(self, x, row_offset=0, row_ranges=None):
        self.x, self.row_offset, self.row_ranges = x, row_offset, row_ranges

    def batch(self, bin_rows, channels_first=True, out_dtype="f32"):
        xb = torch.as_tensor(self.x[np.asarray(bin_rows) - self.row_offset])
        return xb.transpose(1, 2).contiguous() if channels_first else xb


def _predict_worker(rank, world, port, tmp):
    """predict_sharded (OutputGenerator.predict over a process group) == the single-process predict: replicated store
    (contiguous pieces of the bin list) and a store sharded by bin ranges (a bin goes to the rank that holds its rows; the
    bin list is in random order), predictions + 16-d features back in the order of the list on every rank.  Then the GP on
    sharded features: standardisation statistics by rank-ordered sums, fit on rank 0, broadcast predictor, per-rank
    prediction == the single-process GPTrainer on all rows."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from test_region_and_sequence_models import _golden_net
        from digdriver_amd.region_model.predict import predict, predict_sharded
        from digdriver_amd.region_model.trainers import gp_trainer
        net, d = _golden_net()
        rng = np.random.default_rng(5)
        N, L, T = 61, 100, int(d["shape"][2])
        x = (np.round(rng.uniform(0, 1, (N, L, T)), 2) * 100).astype(np.float32)
        rows = rng.permutation(N)[:47]
        labels = [rng.poisson(20, N).astype(float) for _ in range(3)]
        want_p, want_f, want_r2 = predict(net, _HostStore(x), rows, labels=labels, batch_size=16)
        got = predict_sharded(net, _HostStore(x), rows, labels=labels, batch_size=16)
        ranges = [parallel.shard_rows(N, r, world) for r in range(world)]
        lo, hi = ranges[rank]
        got2 = predict_sharded(net, _HostStore(x[lo:hi], row_offset=lo, row_ranges=ranges), rows, labels=labels, batch_size=16)
        for p, f, r2 in (got, got2):
            assert p.shape == want_p.shape and f.shape == want_f.shape
            np.testing.assert_allclose(p, want_p, rtol=1e-5, atol=1e-6)
            np.testing.assert_allclose(f, want_f, rtol=1e-5, atol=1e-6)
            np.testing.assert_allclose(r2, want_r2, rtol=1e-4, atol=1e-7)

        # ---- GP on sharded rows ----
        g = np.random.default_rng(6)
        def make(n):
            X = g.normal(size=(n, 16)) * g.uniform(0.5, 3, 16) + g.normal(size=16)
            X[:, 7] = 0.0                                               # an all-zero feature is dropped (gp_trainer.py:79)
            y = 3.0 + np.sin(X[:, 0]) + 0.5 * X[:, 1] + 0.1 * g.normal(size=n)
            return X, y
        train, val, held = make(500), make(120), make(171)
        cut = lambda tup, r: tuple(a[(len(a) * r) // world + (7 if 0 < r else 0):(len(a) * (r + 1)) // world + (7 if r + 1 < world else 0)] for a in tup)
        mean, std, ym, ys, n = parallel.standardisation_stats(*cut(train, rank))
        assert n == 500
        np.testing.assert_allclose(mean, train[0].mean(0), rtol=1e-12, atol=1e-14)
        sd = train[0].std(0)
        np.testing.assert_allclose(std, np.where(sd == 0, 1.0, sd), rtol=1e-12)
        np.testing.assert_allclose([ym, ys], [train[1].mean(), train[1].std()], rtol=1e-12)
        both = [torch.empty(16, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(both, torch.as_tensor(std))
        assert all(torch.equal(both[0], b) for b in both)                # the same bits on every rank
        torch.manual_seed(0)
        res, means, stds = gp_trainer.run_gp_sharded(torch.device("cpu"), cut(train, rank), cut(val, rank), cut(held, rank),
                                                     n_runs=1, n_iter=12, n_inducing=40, nn_r2=0.2)
        single = gp_trainer.GPTrainer(torch.device("cpu"), train, val, held, n_iter=12, n_inducing=40)
        v, h = single.run()
        mine = cut((h["gp_mean"], h["gp_std"]), rank)
        np.testing.assert_allclose(means, mine[0], rtol=1e-7)
        np.testing.assert_allclose(stds, mine[1], rtol=1e-6, atol=1e-9)
        np.testing.assert_allclose(res[0]["r2"], h["r2"], rtol=1e-8)
        np.testing.assert_allclose(res[0]["val"]["r2"], v["r2"], rtol=1e-8)
        assert h["r2"] > 0.1 and len(means) == len(cut(held, rank)[1])
        if rank == 0:
            np.save(os.path.join(tmp, "predict_ok.npy"), np.ones(1))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_sharded_predict_and_gp_gloo(tmp_path):
    port = _free_port()
    mp.spawn(_predict_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert os.path.exists(tmp_path / "predict_ok.npy")
