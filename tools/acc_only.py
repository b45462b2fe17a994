#!/usr/bin/env python
"""Run only dig_accumulate_elements on the bench workload a few times (for rocprofv3 passes).  Developer tool."""
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
for _ in range(int(os.environ.get("KB_N", 6))):
    engine.accumulate_elements(td["bin_mu"], td["bin_std"], td["bin_y"], td["bin_flag"], td["bin_ctx"], td["ov_ptr"],
                               td["ov_idx"], td["L"], td["strand_minus"], td["d_pr"], out=out_acc)
torch.cuda.synchronize()
