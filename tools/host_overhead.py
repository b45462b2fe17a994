#!/usr/bin/env python
"""How long does the host take to ENQUEUE one bench step (no GPU sync inside the loop)?  If this is close to the GPU
time of a step the pipeline is host-bound and launch gaps appear.  Developer tool."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import make_workload                      # noqa: E402
from digdriver_amd import engine, parallel           # noqa: E402

dev = torch.device("cuda:0")
E, C = 120091, 37
w = make_workload(288000, E, C, seed=3)
td = {k: torch.as_tensor(v, device=dev) for k, v in w.items() if isinstance(v, np.ndarray)}
out_acc = engine.alloc_accumulate_outputs(E, C, 1, dev)
out_st = torch.empty((7, E, C), dtype=torch.float64, device=dev)
part = torch.stack([torch.zeros_like(td["n_snv_obs"]), td["n_snv_obs"], td["n_ind_obs"]]).contiguous()
cj_out = (torch.empty(C, dtype=torch.float64, device=dev), torch.empty(C, dtype=torch.float64, device=dev))


def step():
    engine.scale_suffstats(td["bin_mu"], td["bin_flag"], out=part[0])
    cj, cji = parallel.scale_factors_from_part(part, out=cj_out)
    acc = engine.accumulate_elements(td["bin_mu"], td["bin_std"], td["bin_y"], td["bin_flag"], td["bin_ctx"], td["ov_ptr"],
                                     td["ov_idx"], td["L"], td["strand_minus"], td["d_pr"], out=out_acc)
    engine.element_stats(acc["MU"], acc["SIGMA"], acc["P"].view(E, C), acc["P_INDEL"], td["obs_snv"], td["obs_samples"],
                         td["obs_indel"], cj, cji, out=out_st)


for _ in range(5):
    step()
torch.cuda.synchronize()
n = 200
t0 = time.perf_counter()
for _ in range(n):
    step()
t_enq = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print("host enqueue %.1f us/step, total %.1f us/step" % (t_enq / n * 1e6, t_all / n * 1e6))
import cProfile, pstats
pr = cProfile.Profile()
pr.enable()
for _ in range(200):
    step()
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
