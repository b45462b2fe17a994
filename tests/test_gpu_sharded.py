"""Bin-sharded burden-test path (BASELINE configs[3]): plan_shards -> per-rank tables incl. halo ->
chunked sufficient statistics -> scale factors -> dig_element_pipeline on the shard -> results back in global order.
The ranks of a plan are walked one after the other on ONE device (no process group needed: the exchange is a stack of
the per-rank parts); every output -- scale factors included -- must have the SAME BITS as the unsharded run, for 2 and
for 8 ranks.  The collective itself is covered on CPU by tests/test_distributed_gloo.py."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _run_world(w, world, dev):
    import torch
    from digdriver_amd import engine, parallel
    n_bins = w["bin_mu"].shape[0]
    E, C = w["L"].shape[0], w["d_pr"].shape[0]
    plans = parallel.plan_shards(w["ov_ptr"], w["ov_idx"], n_bins, world)
    # (no process group here: every plan is told how many ranks there are, the "all-gather" is a stack of the parts)
    ranks = [parallel.ShardedPipeline(parallel.shard_inputs(w, p, world), dev, world=world) for p in plans]
    parts = torch.stack([r.scale.enqueue_part().clone() for r in ranks])          # what the all-gather delivers
    out = {"cj": None}
    stats = torch.full((len(engine.ES_PLANES), E, C), float("nan"), dtype=torch.float64, device=dev)
    acc = {k: None for k in ("MU", "SIGMA", "R_OBS", "FLAG", "P", "R_SIZE", "ELT_SIZE", "P_INDEL")}
    for r in ranks:
        r.scale.finish(parts, r.cj, r.cj_indel)
        if out["cj"] is None:
            out["cj"], out["cj_indel"] = r.cj.clone(), r.cj_indel.clone()
        assert torch.equal(out["cj"], r.cj) and torch.equal(out["cj_indel"], r.cj_indel)       # every rank: same factors
        if r.pipe is not None:
            r.pipe.run(r.cj, r.cj_indel, stages=7)
        idx = torch.as_tensor(r.elements, device=dev)
        stats[:, idx] = r.out_stats
        for k in acc:
            if acc[k] is None:
                acc[k] = torch.zeros((E,) + tuple(r.out_acc[k].shape[1:]), dtype=r.out_acc[k].dtype, device=dev)
            acc[k][idx] = r.out_acc[k]
    torch.cuda.synchronize()
    halo = sum(p["n_halo"] for p in plans)
    return out, stats, acc, halo


def test_sharded_equals_unsharded_bit_for_bit():
    import torch
    from bench import make_workload
    dev = torch.device("cuda:0")
    w = make_workload(n_bins=40_000, n_elements=30_011, n_cohorts=37, seed=11)
    base, stats1, acc1, _ = _run_world(w, 1, dev)
    np.testing.assert_allclose(base["cj"].cpu().numpy(), w["cj"], rtol=1e-4)       # the generator plants N_obs = rint(cj * sum)
    assert bool(torch.isfinite(stats1[1]).all())
    for world in (2, 8):
        got, stats, acc, halo = _run_world(w, world, dev)
        assert torch.equal(got["cj"], base["cj"]) and torch.equal(got["cj_indel"], base["cj_indel"]), world
        assert torch.equal(torch.nan_to_num(stats, nan=-7.0), torch.nan_to_num(stats1, nan=-7.0)), world
        for k in acc1:
            assert torch.equal(acc[k], acc1[k]), (world, k)
        assert 0 < halo < 0.05 * 40_000
    # and the unsharded chunked scale factors agree with the plain reduction to rounding
    cj_plain, cji_plain, _ = engine_scale_local(w, dev)
    np.testing.assert_allclose(base["cj"].cpu().numpy(), cj_plain.cpu().numpy(), rtol=1e-13)
    np.testing.assert_allclose(base["cj_indel"].cpu().numpy(), cji_plain.cpu().numpy(), rtol=1e-13)


def engine_scale_local(w, dev):
    import torch
    from digdriver_amd import engine
    t = lambda a: torch.as_tensor(a, device=dev)
    return engine.scale_factors_local(t(w["bin_mu"]), t(w["bin_flag"]), t(w["n_snv_obs"]), t(w["n_ind_obs"]))


def test_premasked_rate_table_gives_the_same_scale_factors():
    """ChunkedScaleFactorPlan sums a plan-time copy of Y_PRED that holds +0.0 in the flagged entries (bin_flag = NULL in
    dig_scale_suffstats_chunked: 8 instead of 9 bytes per (bin, cohort) and step).  Same chunk sums, same scale factors, bit
    for bit, as the plan that reads the flags every step; a table changed in place needs remask()."""
    import torch
    from bench import make_workload
    from digdriver_amd import engine, parallel
    dev = torch.device("cuda:0")
    for (nb, C, seed) in ((40_000, 37, 3), (777, 1, 4), (5000, 100, 5)):
        w = make_workload(n_bins=nb, n_elements=500, n_cohorts=C, seed=seed)
        w["bin_mu"][::11] *= -1.0                          # (negative and zero rates: -0.0 + 0.0 must come out the same way too)
        w["bin_mu"][5] = 0.0
        t = lambda a: torch.as_tensor(a, device=dev)
        rows = parallel.canonical_chunks(nb)
        out = []
        for premask in (False, True):
            plan = engine.ChunkedScaleFactorPlan(t(w["bin_mu"]), t(w["bin_flag"]), t(w["n_snv_obs"]), t(w["n_ind_obs"]), rows,
                                                 parallel.N_CHUNKS, world=1, premask=premask)
            cj, cji = torch.empty(C, dtype=torch.float64, device=dev), torch.empty(C, dtype=torch.float64, device=dev)
            plan.run(cj, cji)
            out.append((plan.enqueue_part().clone(), cj.clone(), cji.clone(), plan))
        torch.cuda.synchronize()
        for a, b in zip(out[0][:3], out[1][:3]):
            assert torch.equal(a, b), (nb, C)
        plan = out[1][3]
        plan.mu[: nb // 3] *= 2.0
        plan.flag[1::5] ^= 1
        plan.remask()
        ref = engine.ChunkedScaleFactorPlan(plan.mu, plan.flag, t(w["n_snv_obs"]), t(w["n_ind_obs"]), rows, parallel.N_CHUNKS, world=1,
                                            premask=False)
        assert torch.equal(plan.enqueue_part(), ref.enqueue_part())


def test_ragged_shards_and_single_cohort():
    """Bin counts that no rank count divides, one cohort (C = 1 takes the no-fastdiv path), elements spanning shard
    boundaries (long elements: up to 12 bins -> the halo is used and the stream kernel's long-element loop runs)."""
    import torch
    from bench import make_workload
    dev = torch.device("cuda:0")
    w = make_workload(n_bins=9_973, n_elements=5_003, n_cohorts=1, seed=12, max_blocks=9)
    base, stats1, acc1, _ = _run_world(w, 1, dev)
    assert int(np.diff(w["ov_ptr"]).max()) > 3
    for world in (2, 8):
        got, stats, acc, halo = _run_world(w, world, dev)
        assert torch.equal(got["cj"], base["cj"])
        assert torch.equal(torch.nan_to_num(stats, nan=-7.0), torch.nan_to_num(stats1, nan=-7.0))
        assert torch.equal(acc["MU"], acc1["MU"]) and torch.equal(acc["P"], acc1["P"]) and halo > 0
