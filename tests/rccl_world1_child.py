"""Child process of tests/test_gpu_rccl_world1.py (not collected by pytest).

A fresh process -- nothing has touched the GPU before dist.init_process_group("nccl", world_size=1) -- that sends EVERY
exchange step of the multi-GPU paths through RCCL on one MI355X (parallel.FORCE_COLLECTIVES: a world of one normally
skips them) and compares the results, bit for bit, with the same calls made with the exchange skipped.  What it covers
is what a one-GPU box can show about the N > 1 code: the RCCL calls get device tensors of the right shape and dtype on
the right device, and they are ordered with the HIP kernels around them on the stream the caller names.

    python tests/rccl_world1_child.py <master_port> <report.json>
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def _eq(a, b):
    import torch
    return a.shape == b.shape and a.dtype == b.dtype and bool(torch.equal(torch.nan_to_num(a.double(), nan=-7.0), torch.nan_to_num(b.double(), nan=-7.0)))


def main(port, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import numpy as np
    import torch
    import torch.distributed as dist
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    torch.cuda.set_device(dev)
    from bench import make_workload
    from digdriver_amd import engine, parallel
    report = {"backend": dist.get_backend(), "world": dist.get_world_size(), "checks": {}}
    ok = report["checks"]

    def forced(flag):
        parallel.FORCE_COLLECTIVES = bool(flag)

    # ---- 0. the small helpers ---------------------------------------------------------------------------------
    forced(True)
    assert parallel.collectives_on() and parallel.comm_device(dev) == dev
    t = torch.arange(12, dtype=torch.float64, device=dev).reshape(3, 4) / 7
    ok["rank_ordered_sum"] = _eq(parallel.rank_ordered_sum(t), t)
    ok["all_gather_rows"] = _eq(parallel.all_gather_rows(t), t) and parallel.all_gather_rows(t[:0]).shape == (0, 4)
    ok["gather_to_rank0"] = _eq(parallel.gather_to_rank0(t), t)
    part = torch.rand(3, 37, dtype=torch.float64, device=dev) + 1
    forced(False)
    want = parallel.scale_factors_from_part(part)
    forced(True)
    got = parallel.scale_factors_from_part(part)
    ok["scale_factors_from_part"] = _eq(got[0], want[0]) and _eq(got[1], want[1])
    ok["broadcast_flag"] = parallel.broadcast_flag(True, dev) is True and parallel.broadcast_flag(False, dev) is False
    n_rows, bs = 37, 8
    feats = torch.rand(n_rows, 16, device=dev)
    ok["gather_visiting_order"] = _eq(parallel.gather_visiting_order(feats, n_rows, bs), feats)
    bn = torch.nn.BatchNorm1d(4).to(dev)
    bn.running_mean.fill_(3.0)
    parallel.broadcast_module_buffers(bn, 0)
    ok["broadcast_module_buffers"] = bool((bn.running_mean == 3.0).all())

    # ---- 1. ShardedPipeline.step(stream=side), the side stream NOT current, the current stream kept busy -----------
    w = make_workload(n_bins=6400, n_elements=5003, n_cohorts=37, seed=17)
    plan = parallel.plan_shards(w["ov_ptr"], w["ov_idx"], 6400, 1)[0]
    shard = parallel.shard_inputs(w, plan, 1)
    forced(False)
    base = parallel.ShardedPipeline(shard, dev)
    assert not base.scale.exchange
    base.step()
    torch.cuda.synchronize()
    forced(True)
    pipe = parallel.ShardedPipeline(shard, dev)
    assert pipe.scale.exchange and pipe.scale.all is not None
    side = torch.cuda.Stream(device=dev)
    busy = torch.rand(4096, 4096, device=dev)
    good = True
    for step in range(6):
        pipe.scale.all.fill_(float("nan"))            # what a finish() that overtook its all-gather would read
        pipe.cj.fill_(float("nan"))
        torch.cuda.synchronize()
        for _ in range(8):                            # ~ms of work on the CURRENT stream: a collective issued there would be late
            busy = (busy @ busy).clamp_(0, 1)
        assert torch.cuda.current_stream(dev) != side
        pipe.step(stream=side)
        side.synchronize()                            # only the side stream: the current one may still be running
        good &= _eq(pipe.cj, base.cj) and _eq(pipe.cj_indel, base.cj_indel) and _eq(pipe.out_stats, base.out_stats)
        good &= all(_eq(pipe.out_acc[k], base.out_acc[k]) for k in base.out_acc)
    torch.cuda.synchronize()
    ok["ShardedPipeline.step(stream=side)"] = bool(good)
    # the plain (non-chunked) scale-factor plan of engine.py (dig_scale_suffstats + all-gather + dig_scale_factors)
    td = pipe.td
    sp = engine.ScaleFactorPlan(td["bin_mu"], td["bin_flag"], td["n_snv_obs"], td["n_ind_obs"])
    outs = {}
    for flag in (False, True):
        forced(flag)
        prt = torch.zeros(3, 37, dtype=torch.float64, device=dev)
        prt[1], prt[2] = td["n_snv_obs"], td["n_ind_obs"]
        cj_, cji_ = torch.empty(37, dtype=torch.float64, device=dev), torch.empty(37, dtype=torch.float64, device=dev)
        sp.run_sharded(prt, cj_, cji_)
        torch.cuda.synchronize()
        outs[flag] = (cj_, cji_)
    ok["ScaleFactorPlan.run_sharded"] = _eq(outs[True][0], outs[False][0]) and _eq(outs[True][1], outs[False][1])

    # ---- 2. ShardedTiles.run() + q_values() -----------------------------------------------------------------------
    from test_gpu_tiles import _tile_problem
    seqs, genome, chroms, starts, ends, S, mu, sg, mc, ms, me, co = _tile_problem(C=3)
    res = {}
    for flag in (False, True):
        forced(flag)
        sh = parallel.ShardedTiles(genome, chroms, starts, ends, S, mu, sg, mc, ms, me, co, 50, dev, 0, 1)
        r = sh.run()
        torch.cuda.synchronize()
        r = {k: v.clone() for k, v in r.items()}
        r["pval"] = torch.nan_to_num(r["pval"], nan=0.5)       # (an all-N bin's NaN makes every q-value NaN: also compare finite ones)
        sh.result = r
        res[flag] = (r, [sh.q_values(c) for c in range(3)], sh.q_values_all())
    ok["ShardedTiles.run"] = all(_eq(res[True][0][k], res[False][0][k]) for k in res[False][0])
    ok["ShardedTiles.q_values"] = all(_eq(a, b) for a, b in zip(res[True][1], res[False][1])) and \
        bool(torch.isfinite(res[True][1][0][res[True][0]["n_valid"] > 0][:, 0]).all())
    ok["ShardedTiles.q_values_all"] = _eq(res[True][2], res[False][2]) and _eq(res[True][2], torch.stack(res[True][1]))

    # ---- 3. predict_sharded on the real BinTrackStore ---------------------------------------------------------------
    from test_region_and_sequence_models import _golden_net
    from digdriver_amd.region_model.data_aux.dataset_generator import BinTrackStore
    from digdriver_amd.region_model.predict import predict, predict_sharded
    net, d = _golden_net()
    net = net.to(dev)
    rng = np.random.default_rng(5)
    N, L, T = 301, 100, int(d["shape"][2])
    x = torch.as_tensor((np.round(rng.uniform(0, 1, (N, L, T)), 2) * 100).astype(np.float32), device=dev)
    rows = rng.permutation(N)[:211]
    labels = [rng.poisson(20, N).astype(float) for _ in range(3)]
    forced(False)
    want_p, want_f, want_r2 = predict(net, BinTrackStore(x), rows, labels=labels, batch_size=64)
    forced(True)
    got_p, got_f, got_r2 = predict_sharded(net, BinTrackStore(x), rows, labels=labels, batch_size=64)
    got2 = predict_sharded(net, BinTrackStore(x, row_offset=0, row_ranges=[(0, N)]), rows, labels=labels, batch_size=64)
    ok["predict_sharded"] = bool(np.array_equal(got_p, want_p) and np.array_equal(got_f, want_f) and np.array_equal(got_r2, want_r2)
                                 and np.array_equal(got2[0], want_p) and np.array_equal(got2[1], want_f))

    # ---- 4. run_gp_sharded (standardisation sums, row gathers, flag + predictor broadcast over RCCL) ----------------
    from digdriver_amd.region_model.trainers import gp_trainer
    g = np.random.default_rng(6)

    def make(n):
        X = g.normal(size=(n, 16)) * g.uniform(0.5, 3, 16) + g.normal(size=16)
        X[:, 7] = 0.0
        y = 3.0 + np.sin(X[:, 0]) + 0.5 * X[:, 1] + 0.1 * g.normal(size=n)
        return X, y
    train, val, held = make(3000), make(400), make(571)
    runs = {}
    for flag in (False, True, "again"):
        forced(flag is True)
        torch.manual_seed(0)
        runs[flag] = gp_trainer.run_gp_sharded(dev, train, val, held, n_runs=2, n_iter=15, n_inducing=60, nn_r2=0.2, seed=3)
    same = lambda a, b: bool(np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]) and a[0][0]["r2"] == b[0][0]["r2"]
                             and a[0][1]["val"]["r2"] == b[0][1]["val"]["r2"])
    report["gp_fit_is_deterministic"] = same(runs[False], runs["again"])
    report["gp_forced_vs_plain_max_abs"] = [float(np.max(np.abs(runs[True][1] - runs[False][1]))), float(np.max(np.abs(runs[True][2] - runs[False][2])))]
    if report["gp_fit_is_deterministic"]:
        ok["run_gp_sharded"] = same(runs[True], runs[False])
    else:                                              # atomics in the fit: the two no-group runs differ too; compare to that spread
        spread = float(np.max(np.abs(runs[False][1] - runs["again"][1])))
        ok["run_gp_sharded"] = float(np.max(np.abs(runs[True][1] - runs[False][1]))) <= max(10 * spread, 1e-9)
    ok["run_gp_sharded_r2"] = runs[True][0][0]["r2"] > 0.5
    mean, std, ym, ys, n = parallel.standardisation_stats(train[0], train[1], comm=dev)
    forced(False)
    mean0, std0, ym0, ys0, n0 = parallel.standardisation_stats(train[0], train[1])
    forced(True)
    ok["standardisation_stats"] = bool(np.array_equal(mean, mean0) and np.array_equal(std, std0) and (ym, ys, n) == (ym0, ys0, n0))

    # ---- 5. average_gradients + a data-parallel NNTrainer epoch -------------------------------------------------------
    torch.manual_seed(1)
    lin = torch.nn.Sequential(torch.nn.Linear(8, 5), torch.nn.ReLU(), torch.nn.Linear(5, 1)).to(dev)
    lin(torch.rand(32, 8, device=dev)).sum().backward()
    before = [p.grad.clone() for p in lin.parameters()]
    parallel.average_gradients(list(lin.parameters()))
    ok["average_gradients"] = all(_eq(a, p.grad) for a, p in zip(before, lin.parameters()))
    from torch import nn, optim
    from digdriver_amd.region_model.nets.cnn_predictors import SimpleMultiTaskResNet
    from digdriver_amd.region_model.trainers.nn_trainer import NNTrainer
    gd = np.load(os.path.join(ROOT, "tests", "golden", "nn_training_golden.npz"))
    Tn, Ln, Cn, n_train, n_val, bsn = [int(v) for v in gd["shape"]]
    store = BinTrackStore(torch.tensor(gd["x"]).to(dev))
    epochs = {}
    for flag in (False, True, "again"):
        forced(flag is True)
        torch.manual_seed(5)
        m = SimpleMultiTaskResNet((n_train, Ln, Tn), Cn)
        tr = NNTrainer(m, optim.Adam(m.parameters(), lr=1e-3), nn.MSELoss(), bsn, ["a", "b"], store, np.arange(n_train),
                       np.arange(n_train, n_train + n_val), list(gd["labels"]), dev, seed=9)
        losses, accs, f, p, tt = tr.train(1)
        epochs[flag] = (losses, np.stack(p), np.stack(f), tr.last_train_rows.copy())
    report["cnn_epoch_is_deterministic"] = bool(np.array_equal(epochs[False][1], epochs["again"][1]))
    ok["NNTrainer.rows"] = bool(np.array_equal(epochs[True][3], epochs[False][3]))
    if report["cnn_epoch_is_deterministic"]:
        ok["NNTrainer.train"] = bool(np.array_equal(epochs[True][1], epochs[False][1]) and np.array_equal(epochs[True][2], epochs[False][2])
                                     and np.array_equal(epochs[True][0], epochs[False][0]))
    else:
        ok["NNTrainer.train"] = bool(np.allclose(epochs[True][1], epochs[False][1], rtol=1e-3, atol=1e-4))

    torch.cuda.synchronize()
    dist.barrier()
    dist.destroy_process_group()
    report["all_ok"] = all(bool(v) for v in ok.values())
    with open(out_path, "w") as f:
        json.dump(report, f, indent=1)
    print(json.dumps(report))
    return 0 if report["all_ok"] else 1


if __name__ == "__main__":
    sys.exit(main(int(sys.argv[1]), sys.argv[2]))
