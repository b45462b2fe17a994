#!/usr/bin/env python
"""Run only dig_element_stats on the bench workload a few times (for rocprofv3 --pmc passes).  Developer tool."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import make_workload                      # noqa: E402
from digdriver_amd import engine                     # noqa: E402

dev = torch.device("cuda:0")
E, C = int(os.environ.get("KB_E", 120091)), int(os.environ.get("KB_C", 37))
w = make_workload(288000, E, C, seed=3)
td = {k: torch.as_tensor(v, device=dev) for k, v in w.items() if isinstance(v, np.ndarray)}
out_acc = engine.alloc_accumulate_outputs(E, C, 1, dev)
out_st = torch.empty((7, E, C), dtype=torch.float64, device=dev)
engine.accumulate_elements(td["bin_mu"], td["bin_std"], td["bin_y"], td["bin_flag"], td["bin_ctx"], td["ov_ptr"],
                           td["ov_idx"], td["L"], td["strand_minus"], td["d_pr"], out=out_acc)
for _ in range(int(os.environ.get("KB_N", 4))):
    engine.element_stats(out_acc["MU"], out_acc["SIGMA"], out_acc["P"].view(E, C), out_acc["P_INDEL"], td["obs_snv"],
                         td["obs_samples"], td["obs_indel"], td["cj"], td["cj_indel"], out=out_st)
torch.cuda.synchronize()
k = td["obs_snv"].double()
print("k_snv mean %.2f  max %d;  k_ind mean %.2f max %d" % (k.mean().item(), int(k.max()), td["obs_indel"].double().mean().item(), int(td["obs_indel"].max())))
kk = torch.maximum(td["obs_snv"], td["obs_samples"]).clamp(max=64).view(-1)
pad = (-kk.numel()) % 64
kw = torch.nn.functional.pad(kk, (0, pad)).view(-1, 64)
print("mean over waves of max k (snv/samples, capped 64): %.2f ; mean k %.2f" % (kw.max(dim=1).values.double().mean().item(), kk.double().mean().item()))
ki = td["obs_indel"].clamp(max=64).view(-1)
kw = torch.nn.functional.pad(ki, (0, pad)).view(-1, 64)
print("mean over waves of max k_indel: %.2f ; mean %.2f" % (kw.max(dim=1).values.double().mean().item(), ki.double().mean().item()))
