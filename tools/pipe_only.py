#!/usr/bin/env python
"""Run only dig_element_pipeline on the bench workload a few times (for rocprofv3 passes).  Developer tool."""
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
oa = engine.alloc_accumulate_outputs(E, C, 1, dev)
st = torch.empty((7, E, C), dtype=torch.float64, device=dev)
for _ in range(int(os.environ.get("KB_N", 6))):
    engine.element_pipeline(td["bin_mu"], td["bin_std"], td["bin_y"], td["bin_flag"], td["bin_ctx"], td["ov_ptr"], td["ov_idx"],
                            td["L"], td["strand_minus"], td["d_pr"], td["obs_snv"], td["obs_samples"], td["obs_indel"], td["cj"],
                            td["cj_indel"], out_acc=oa, out_stats=st)
torch.cuda.synchronize()
