#!/usr/bin/env python
"""What does the compacted pass cost as a function of what is in the worklist?  Developer tool."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import make_workload                      # noqa: E402
from digdriver_amd import engine                     # noqa: E402

dev = torch.device("cuda:0")
E, C = 120091, 37
w = make_workload(288000, E, C, seed=3)
td = {k: torch.as_tensor(v, device=dev) for k, v in w.items() if isinstance(v, np.ndarray)}
acc = engine.alloc_accumulate_outputs(E, C, 1, dev)
engine.accumulate_elements(td["bin_mu"], td["bin_std"], td["bin_y"], td["bin_flag"], td["bin_ctx"], td["ov_ptr"],
                           td["ov_idx"], td["L"], td["strand_minus"], td["d_pr"], out=acc)
out = torch.empty((7, E, C), dtype=torch.float64, device=dev)


def run(tag, snv, smp, ind):
    for _ in range(4):
        engine.element_stats(acc["MU"], acc["SIGMA"], acc["P"].view(E, C), acc["P_INDEL"], snv, smp, ind, td["cj"],
                             td["cj_indel"], out=out)
    torch.cuda.synchronize()
    ws = engine._WS_CACHE[("element_stats", 0)]
    print(tag, "slow pairs:", int(ws[:4].view(torch.int32)[0]))


run("bench workload", td["obs_snv"], td["obs_samples"], td["obs_indel"])
cap = lambda t: t.clamp(max=64)
run("counts capped at 64 (only small tails remain)", cap(td["obs_snv"]), cap(td["obs_samples"]), td["obs_indel"])
z = torch.zeros_like(td["obs_snv"])
run("all counts zero (empty worklist)", z, z, z)
big = torch.where(td["obs_snv"] > 64, td["obs_snv"], torch.minimum(td["obs_snv"], torch.full_like(td["obs_snv"], 20)))
run("only k > 64 items and their mates", big, torch.minimum(td["obs_samples"], big), td["obs_indel"])
