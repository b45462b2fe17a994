#!/usr/bin/env python
"""Kernel micro-bench on the bench workload: per-kernel times (HIP events on torch's stream) for a
few tuning knobs.  Developer tool; not part of the product or of the judged bench line."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import make_workload, algorithmic_bytes   # noqa: E402
from digdriver_amd import engine                     # noqa: E402


def timeit(fn, n=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3   # us


def main():
    dev = torch.device("cuda:0")
    E = int(os.environ.get("KB_E", 120091))
    C = int(os.environ.get("KB_C", 37))
    w = make_workload(288000, E, C, seed=3)
    td = {k: torch.as_tensor(v, device=dev) for k, v in w.items() if isinstance(v, np.ndarray)}
    nbar = len(w["ov_idx"]) / E
    b_acc, b_stat = algorithmic_bytes(E, C, nbar)
    out_acc = engine.alloc_accumulate_outputs(E, C, 1, dev)
    out_st = torch.empty((7, E, C), dtype=torch.float64, device=dev)

    def acc():
        engine.accumulate_elements(td["bin_mu"], td["bin_std"], td["bin_y"], td["bin_flag"], td["bin_ctx"], td["ov_ptr"],
                                   td["ov_idx"], td["L"], td["strand_minus"], td["d_pr"], out=out_acc)

    def stats(ws=True):
        engine.element_stats(out_acc["MU"], out_acc["SIGMA"], out_acc["P"].view(E, C), out_acc["P_INDEL"], td["obs_snv"],
                             td["obs_samples"], td["obs_indel"], td["cj"], td["cj_indel"], out=out_st, use_workspace=ws)

    acc()
    us = timeit(acc)
    print("accumulate %8.1f us  %7.1f GB/s algorithmic (%.1f%% of 8 TB/s)" % (us, b_acc / us / 1e3, b_acc / us / 1e3 / 80))
    for ws in (True, False):
        us = timeit(lambda: stats(ws))
        print("element_stats workspace=%-5s %8.1f us  %7.1f GB/s algorithmic (%.1f%% of 8 TB/s)" % (ws, us, b_stat / us / 1e3, b_stat / us / 1e3 / 80))
    pipe_acc = engine.alloc_accumulate_outputs(E, C, 1, dev)
    us = timeit(lambda: engine.element_pipeline(td["bin_mu"], td["bin_std"], td["bin_y"], td["bin_flag"], td["bin_ctx"],
                                                td["ov_ptr"], td["ov_idx"], td["L"], td["strand_minus"], td["d_pr"],
                                                td["obs_snv"], td["obs_samples"], td["obs_indel"], td["cj"], td["cj_indel"],
                                                out_acc=pipe_acc, out_stats=out_st))
    print("element_pipeline (fused) %8.1f us  %7.1f GB/s algorithmic (%.1f%% of 8 TB/s)" % (
        us, (b_acc + b_stat) / us / 1e3, (b_acc + b_stat) / us / 1e3 / 80))
    us = timeit(lambda: engine.scale_suffstats(td["bin_mu"], td["bin_flag"]))
    print("scale_suffstats %8.1f us  %7.1f GB/s" % (us, td["bin_mu"].numel() * 9 / us / 1e3))
    us = timeit(lambda: engine.accumulate_elements(td["bin_mu"], td["bin_std"], td["bin_y"], td["bin_flag"], td["bin_ctx"],
                                                   td["ov_ptr"], td["ov_idx"], td["L"], td["strand_minus"], td["d_pr"],
                                                   out=out_acc, use_workspace=False), n=5, warm=1)
    print("accumulate v1 (LDS, no workspace) %8.1f us" % us)


if __name__ == "__main__":
    main()
