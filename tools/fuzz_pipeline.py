#!/usr/bin/env python
"""Random shapes through dig_element_pipeline against dig_accumulate_elements + dig_element_stats (bit for bit): element
and cohort counts from one to a few thousand, from no pair to every pair needing the second pass, with the cases that fill
a workgroup's 1024-record queue exactly, by one less and by one tile more.  (developer tool)

    python tools/fuzz_pipeline.py [n_cases] [seed]
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import make_workload          # noqa: E402
from digdriver_amd import engine         # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
dev = torch.device("cuda:0")
shapes = [(4096, 1, 1.0), (4096 + 64, 1, 1.0), (4096 - 1, 1, 1.0), (1024, 4, 1.0), (1, 1, 1.0), (63, 1, 0.5), (1025, 1, 1.0),
          (2048, 37, 1.0), (7000, 37, 0.07), (7000, 37, 0.055)]
bad = 0
for case in range(n_cases):
    if case < len(shapes):
        E, C, frac = shapes[case]
    else:
        E = int(rng.integers(1, 6000))
        C = int(rng.choice([1, 2, 5, 16, 37, 48, 70]))
        frac = float(rng.choice([0.0, 0.01, 0.06, 0.3, 1.0]))
    nb = int(rng.integers(max(8, E // 4), max(16, 2 * E)))
    w = make_workload(n_bins=nb, n_elements=E, n_cohorts=C, seed=int(rng.integers(1 << 30)))
    td = {k: torch.as_tensor(v, device=dev) for k, v in w.items() if isinstance(v, np.ndarray)}
    mask = torch.as_tensor(rng.uniform(size=(E, C)) < frac, device=dev)
    td["obs_snv"] += mask.to(torch.int32) * 300
    td["obs_indel"] += (mask & torch.as_tensor(rng.uniform(size=(E, C)) < 0.3, device=dev)).to(torch.int32) * 200
    acc = engine.accumulate_elements(td["bin_mu"], td["bin_std"], td["bin_y"], td["bin_flag"], td["bin_ctx"], td["ov_ptr"],
                                     td["ov_idx"], td["L"], td["strand_minus"], td["d_pr"])
    st = engine.element_stats(acc["MU"], acc["SIGMA"], acc["P"].view(E, C), acc["P_INDEL"], td["obs_snv"], td["obs_samples"],
                              td["obs_indel"], td["cj"], td["cj_indel"])
    acc2, st2 = engine.element_pipeline(td["bin_mu"], td["bin_std"], td["bin_y"], td["bin_flag"], td["bin_ctx"],
                                        td["ov_ptr"], td["ov_idx"], td["L"], td["strand_minus"], td["d_pr"], td["obs_snv"],
                                        td["obs_samples"], td["obs_indel"], td["cj"], td["cj_indel"])
    ok = all(torch.equal(torch.nan_to_num(st[name], nan=-7.0), torch.nan_to_num(st2[j], nan=-7.0))
             for j, name in enumerate(engine.ES_PLANES))
    ok = ok and all(torch.equal(torch.nan_to_num(acc[k].double(), nan=-7.0), torch.nan_to_num(acc2[k].double(), nan=-7.0)) for k in acc)
    # a plan with the packed bin records (and, L being context-repeated, the compact accumulation): rate outputs and the
    # statistics that do not depend on P bit for bit, the others through P within its tolerance
    plan = engine.PipelinePlan(td["bin_mu"], td["bin_std"], td["bin_y"], td["bin_flag"], td["bin_ctx"], td["ov_ptr"], td["ov_idx"], td["L"],
                               td["strand_minus"], td["d_pr"], td["obs_snv"], td["obs_samples"], td["obs_indel"], compact=False)
    acc3, st3 = plan.run(td["cj"], td["cj_indel"])
    torch.cuda.synchronize()
    ok = ok and torch.equal(torch.nan_to_num(st3, nan=-7.0), torch.nan_to_num(st2, nan=-7.0))
    ok = ok and all(torch.equal(torch.nan_to_num(acc[k].double(), nan=-7.0), torch.nan_to_num(acc3[k].double(), nan=-7.0)) for k in acc)
    # the record form of the statistics stage (DIG_PIPE_RECORDS, round 5), general and compact accumulation: unpacked, the bits of the planes
    for compact in (False, "auto"):
        ref = plan if compact is False else engine.PipelinePlan(td["bin_mu"], td["bin_std"], td["bin_y"], td["bin_flag"], td["bin_ctx"],
                                                                td["ov_ptr"], td["ov_idx"], td["L"], td["strand_minus"], td["d_pr"], td["obs_snv"],
                                                                td["obs_samples"], td["obs_indel"], pack_bins=plan)
        acc_r, st_r = ref.run(td["cj"], td["cj_indel"])
        rec = engine.PipelinePlan(td["bin_mu"], td["bin_std"], td["bin_y"], td["bin_flag"], td["bin_ctx"], td["ov_ptr"], td["ov_idx"], td["L"],
                                  td["strand_minus"], td["d_pr"], td["obs_snv"], td["obs_samples"], td["obs_indel"], compact=compact,
                                  pack_bins=plan, records_out=True)
        rec.out_records.fill_(float("nan"))
        rec.run(td["cj"], td["cj_indel"])
        acc4, st4 = rec.unpack()
        torch.cuda.synchronize()
        ok = ok and torch.equal(torch.nan_to_num(st4, nan=-7.0), torch.nan_to_num(st_r, nan=-7.0))
        ok = ok and all(torch.equal(torch.nan_to_num(acc_r[k].double(), nan=-7.0), torch.nan_to_num(acc4[k].double(), nan=-7.0)) for k in acc_r)
    neg = int((torch.nan_to_num(st2[1], nan=0.0) < 0).sum() + (torch.nan_to_num(st2[5], nan=0.0) < 0).sum())
    if not ok or neg:
        bad += 1
        print("MISMATCH case", case, "E", E, "C", C, "bins", nb, "frac", frac, "markers left", neg, flush=True)
print("cases", n_cases, "mismatches", bad)
sys.exit(1 if bad else 0)
