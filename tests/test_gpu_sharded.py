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


def _walk_against_unsharded_and_oracle(w, world, dev, n_per_shard, min_halo_elts):
    """The `world` ranks of a plan walked on one device: bits of the unsharded run; then every halo element (one that reads
    a neighbour's bins; at most 200 per shard) and `n_per_shard` others per shard against the ORACLE on the global tables."""
    import torch
    from conftest import rel_close
    from digdriver_amd import engine, parallel
    from oracle import dig_oracle as O
    N, C = w["bin_mu"].shape
    base, stats1, acc1, _ = _run_world(w, 1, dev)
    got, stats, acc, halo = _run_world(w, world, dev)
    assert torch.equal(got["cj"], base["cj"]) and torch.equal(got["cj_indel"], base["cj_indel"])
    assert torch.equal(torch.nan_to_num(stats, nan=-7.0), torch.nan_to_num(stats1, nan=-7.0))
    for k in acc1:
        assert torch.equal(acc[k], acc1[k]), k
    plans = parallel.plan_shards(w["ov_ptr"], w["ov_idx"], N, world)
    ranges = parallel.bin_ranges(N, world)
    rng = np.random.default_rng(8)
    pick, n_halo_elts = [], 0
    for p, (lo, hi) in zip(plans, ranges):
        elts = p["elements"]
        assert len(elts) > 0.8 * len(w["L"]) / world                  # every shard holds about its share of the elements
        rows = p["bin_rows"][p["ov_idx"]]                             # global bin rows of the shard's CSR
        foreign = (rows < lo) | (rows >= hi)
        owner_of_entry = np.repeat(np.arange(len(elts)), np.diff(p["ov_ptr"]))
        halo_elts = elts[np.unique(owner_of_entry[foreign])]          # elements that need a neighbour's bins
        n_halo_elts += len(halo_elts)
        rest = np.setdiff1d(elts, halo_elts)
        pick.append(np.concatenate([halo_elts[:200], rng.choice(rest, n_per_shard, replace=False)]))
    assert n_halo_elts >= min_halo_elts and halo >= 1
    pick = np.sort(np.concatenate(pick))
    ptr = np.concatenate([[0], np.cumsum(np.diff(w["ov_ptr"])[pick])])
    idx = np.concatenate([w["ov_idx"][w["ov_ptr"][e]:w["ov_ptr"][e + 1]] for e in pick])
    want = O.accumulate_elements(w["bin_mu"], w["bin_std"], w["bin_y"], w["bin_flag"], w["bin_ctx"], ptr, idx, w["L"][pick],
                                 w["strand_minus"][pick].astype(bool), w["d_pr"])
    pk = torch.as_tensor(pick, device=dev)
    np.testing.assert_allclose(acc["MU"][pk].cpu().numpy(), want["MU"], rtol=1e-12)
    np.testing.assert_allclose(acc["SIGMA"][pk].cpu().numpy(), want["SIGMA"], rtol=1e-12)
    rel_close(acc["P"][pk].cpu().numpy(), want["P"], 1e-11)
    for k in ("R_OBS", "FLAG", "R_SIZE", "ELT_SIZE"):
        assert np.array_equal(acc[k][pk].cpu().numpy(), want[k]), k
    cj_o, cji_o = zip(*[O.scale_factor_genome(w["bin_mu"][:, c], w["bin_flag"][:, c], w["n_snv_obs"][c], w["n_ind_obs"][c])
                        for c in range(C)])
    np.testing.assert_allclose(got["cj"].cpu().numpy(), np.array(cj_o), rtol=1e-12)
    np.testing.assert_allclose(got["cj_indel"].cpu().numpy(), np.array(cji_o), rtol=1e-12)
    ref_st = O.element_stats(want["MU"], want["SIGMA"], want["P"][:, 0, :], want["P_INDEL"][:, None], w["obs_snv"][pick],
                             w["obs_samples"][pick], w["obs_indel"][pick], np.array(cj_o)[None, :], np.array(cji_o)[None, :])
    st_h = stats[:, pk].cpu().numpy()
    for j, name in enumerate(engine.ES_PLANES):
        rel_close(st_h[j], ref_st[name], rtol=1e-6)
    return len(pick), n_halo_elts


def _plant_boundary_elements(w, world, per_boundary, seed=5):
    """Re-point the overlapped-bin lists of `per_boundary` multi-bin elements per inner shard boundary to consecutive bins
    that straddle it (the first bin stays on the left: the element belongs to the left rank and reads the right rank's
    bins through the halo).  Only the CSR entries change; every other input keeps its value."""
    rng = np.random.default_rng(seed)
    N = w["bin_mu"].shape[0]
    nov = np.diff(w["ov_ptr"])
    cand = rng.permutation(np.flatnonzero(nov >= 2))
    n = 0
    for r, (lo, hi) in enumerate(parallel_bin_ranges(N, world)[:-1]):
        for e in cand[n:n + per_boundary]:
            k = int(nov[e])
            j0 = int(rng.integers(1, k))                       # bins on the left of the boundary: 1 .. k - 1
            w["ov_idx"][w["ov_ptr"][e]:w["ov_ptr"][e + 1]] = hi - j0 + np.arange(k)
        n += per_boundary
    return n


def parallel_bin_ranges(n_bins, world):
    from digdriver_amd import parallel
    return parallel.bin_ranges(n_bins, world)


@pytest.mark.timeout(1500)
def test_configs3_full_size_eight_way_walk_with_oracle_sample():
    """BASELINE configs[3] at FULL size -- 288 000 bins x 37 cohorts x 120 091 elements, bin-sharded 8 ways (strong mode:
    one genome, one element set) -- the eight ranks walked on one device: every output, scale factors included, has the
    bits of the unsharded run, and > 1 000 elements spread over all eight shards agree with the ORACLE evaluated on the
    unsharded global tables.  The bench workload's short elements cross a shard boundary once or twice per genome, so a
    second whole-genome walk uses long elements (up to 12 blocks) of which 40 per inner boundary are moved across it:
    280 halo elements, all checked."""
    import torch
    from bench import make_workload
    dev = torch.device("cuda:0")
    n, n_halo = _walk_against_unsharded_and_oracle(make_workload(288_000, 120_091, 37, seed=3), 8, dev, 130, 1)
    assert n >= 1040
    w = make_workload(288_000, 40_009, 37, seed=4, max_blocks=12)
    planted = _plant_boundary_elements(w, 8, 40)
    n, n_halo = _walk_against_unsharded_and_oracle(w, 8, dev, 40, planted)
    assert n_halo >= 7 * 40
