"""BASELINE configs[1] at FULL size inside the driver-run suite: whole genome, ~288 000 10-kb bins, ONE cohort --
sequence_model CNN forward over every bin from the HBM-resident int16 track matrix (T = 735: 42 GB), then the NB burden
test over 120 091 elements (C = 1 takes the no-fastdiv / single-cohort paths of the kernels).  Checked against the oracle
on a sample and through size-independent properties."""
import numpy as np
import pytest

from conftest import rel_close

pytestmark = pytest.mark.gpu


@pytest.mark.timeout(1500)
def test_configs1_whole_genome_cnn_forward_and_burden_test():
    import torch
    from bench import make_workload
    from digdriver_amd import engine
    from digdriver_amd.region_model.data_aux.dataset_generator import BinTrackStore
    from digdriver_amd.region_model.nets.cnn_predictors import SimpleMultiTaskResNet
    from digdriver_amd.region_model.predict import predict
    from oracle import dig_oracle as O
    dev = torch.device("cuda:0")
    N, L, T, C, E = 288_000, 100, 735, 1, 120_091
    # ---- CNN forward over the whole genome ----
    gen = torch.Generator(device=dev).manual_seed(1)
    x = torch.empty((N, L, T), dtype=torch.int16, device=dev)
    for s in range(0, N, 16_000):                       # round(U, 2) * 100 values (DataExtractor.py:220), filled in slabs
        x[s:s + 16_000] = (torch.rand((min(16_000, N - s), L, T), device=dev, generator=gen) * 100).round().to(torch.int16)
    store = BinTrackStore(x)
    torch.manual_seed(0)
    net = SimpleMultiTaskResNet((2048, L, T), C).eval().to(dev)
    rows = np.arange(N)
    preds, feats, _ = predict(net, store, rows, batch_size=4096)
    assert preds.shape == (C, N) and feats.shape == (C, N, 16) and np.isfinite(preds).all() and np.isfinite(feats).all()
    # a sample of bins through the plain module (no BatchNorm folding, MIOpen convolutions, fp32 gather)
    pick = np.random.default_rng(3).choice(N, 96, replace=False)
    with torch.no_grad():
        xb = x[torch.as_tensor(pick, device=dev)].float()            # [B, L, T], the layout the reference module takes
        want_o, want_f, _ = net(xb)
    # (fp32 both ways; the reference-seeded golden at T = 735 is tests/test_gpu_pipeline.py::test_cnn_forward_at_735_tracks_...)
    np.testing.assert_allclose(preds[0, pick], want_o[0].cpu().numpy(), rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(feats[0, pick], want_f[0].cpu().numpy(), rtol=1e-4, atol=2e-5)
    # batch composition must not matter: the same bins in another batch size give the same numbers
    p2, _, _ = predict(net, store, pick, batch_size=32)
    np.testing.assert_allclose(p2[0], preds[0, pick], rtol=1e-4, atol=1e-5)
    del x, store
    torch.cuda.empty_cache()
    # ---- the burden test, one cohort, full element set ----
    w = make_workload(N, E, C, seed=2)
    td = {k: torch.as_tensor(v, device=dev) for k, v in w.items() if isinstance(v, np.ndarray)}
    acc, st = engine.element_pipeline(td["bin_mu"], td["bin_std"], td["bin_y"], td["bin_flag"], td["bin_ctx"], td["ov_ptr"],
                                      td["ov_idx"], td["L"], td["strand_minus"], td["d_pr"], td["obs_snv"], td["obs_samples"],
                                      td["obs_indel"], td["cj"], td["cj_indel"])
    torch.cuda.synchronize()
    st_h = st.cpu().numpy()
    assert np.isfinite(st_h[1]).all() and (st_h[1] >= 0).all() and (st_h[1] <= 1).all()
    n = 4000                                            # oracle on the first 4 000 elements
    ptr = w["ov_ptr"][: n + 1]
    ref_acc = O.accumulate_elements(w["bin_mu"], w["bin_std"], w["bin_y"], w["bin_flag"], w["bin_ctx"], ptr, w["ov_idx"][: ptr[-1]],
                                    w["L"][:n], w["strand_minus"][:n].astype(bool), w["d_pr"])
    np.testing.assert_allclose(acc["MU"][:n].cpu().numpy(), ref_acc["MU"], rtol=1e-13)
    np.testing.assert_allclose(acc["SIGMA"][:n].cpu().numpy(), ref_acc["SIGMA"], rtol=1e-13)
    np.testing.assert_allclose(acc["P"][:n].cpu().numpy(), ref_acc["P"], rtol=1e-11)
    assert np.array_equal(acc["R_OBS"][:n].cpu().numpy(), ref_acc["R_OBS"]) and np.array_equal(acc["R_SIZE"][:n].cpu().numpy(), ref_acc["R_SIZE"])
    ref_st = O.element_stats(ref_acc["MU"], ref_acc["SIGMA"], ref_acc["P"][:, 0, :], ref_acc["P_INDEL"][:, None], w["obs_snv"][:n],
                             w["obs_samples"][:n], w["obs_indel"][:n], w["cj"][None, :], w["cj_indel"][None, :])
    for j, name in enumerate(engine.ES_PLANES):
        rel_close(st_h[j, :n], ref_st[name], rtol=1e-6)
    # the same statistics from the two-call form (accumulate, then element_stats): identical bits
    acc2 = engine.accumulate_elements(td["bin_mu"], td["bin_std"], td["bin_y"], td["bin_flag"], td["bin_ctx"], td["ov_ptr"], td["ov_idx"],
                                      td["L"], td["strand_minus"], td["d_pr"])
    st2 = engine.element_stats(acc2["MU"], acc2["SIGMA"], acc2["P"][:, 0, :].contiguous(), acc2["P_INDEL"], td["obs_snv"],
                               td["obs_samples"], td["obs_indel"], td["cj"], td["cj_indel"])
    for j, name in enumerate(engine.ES_PLANES):
        assert torch.equal(torch.nan_to_num(st[j], nan=-7.0), torch.nan_to_num(st2[name], nan=-7.0)), name


@pytest.mark.timeout(600)
def test_configs2_compact_form_at_full_size_against_general_form():
    """BASELINE configs[2] at full size (288 000 bins, 37 cohorts, 120 091 elements): the compact form of the accumulation
    (what bench.py's plans run) against the general 192-substitution form -- rate sums, sizes and P_INDEL bit-identical, P
    within 1e-13, every statistics plane within the contract; and a sample of elements against the oracle."""
    import torch
    from bench import make_workload
    from digdriver_amd import engine
    from oracle import dig_oracle as O
    from conftest import rel_close
    dev = torch.device("cuda:0")
    w = make_workload(288_000, 120_091, 37, seed=3)
    td = {k: torch.as_tensor(v, device=dev) for k, v in w.items() if isinstance(v, np.ndarray)}
    args = (td["bin_mu"], td["bin_std"], td["bin_y"], td["bin_flag"], td["bin_ctx"], td["ov_ptr"], td["ov_idx"], td["L"],
            td["strand_minus"], td["d_pr"], td["obs_snv"], td["obs_samples"], td["obs_indel"])
    acc_g, st_g = engine.element_pipeline(*args, td["cj"], td["cj_indel"])
    plan = engine.PipelinePlan(*args)
    assert plan.compact
    acc_c, st_c = plan.run(td["cj"], td["cj_indel"])
    torch.cuda.synchronize()
    for k in ("MU", "SIGMA", "R_OBS", "FLAG", "R_SIZE", "ELT_SIZE", "P_INDEL"):
        assert torch.equal(acc_g[k], acc_c[k]), k
    rel = ((acc_c["P"] - acc_g["P"]).abs() / acc_g["P"].abs()).max().item()
    assert rel <= 1e-13, rel
    sg, sc = st_g.cpu().numpy(), st_c.cpu().numpy()
    for j, name in enumerate(engine.ES_PLANES):
        rel_close(sc[j], sg[j], 2e-7 if name.startswith("PVAL") else 1e-12)
    pick = np.sort(np.random.default_rng(5).choice(120_091, 1500, replace=False))
    ptr = np.concatenate([[0], np.cumsum(np.diff(w["ov_ptr"])[pick])])
    idx = np.concatenate([w["ov_idx"][w["ov_ptr"][e]:w["ov_ptr"][e + 1]] for e in pick])
    want = O.accumulate_elements(w["bin_mu"], w["bin_std"], w["bin_y"], w["bin_flag"], w["bin_ctx"], ptr, idx, w["L"][pick],
                                 w["strand_minus"][pick].astype(bool), w["d_pr"])
    rel_close(acc_c["P"].cpu().numpy()[pick], want["P"], 1e-11)
    assert np.array_equal(acc_c["R_SIZE"].cpu().numpy()[pick], want["R_SIZE"])
