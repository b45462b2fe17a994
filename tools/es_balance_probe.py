#!/usr/bin/env python
"""When do the workgroups of the statistics stream pass finish?  (developer probe; needs a -DDIG_ES_TIMING build)

    DIG_HIP_LIB=.../es_timing.so python tools/es_balance_probe.py
"""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import make_workload          # noqa: E402
from digdriver_amd import _lib, engine   # noqa: E402

dev = torch.device("cuda:0")
E, C = 120091, 37
w = make_workload(288000, E, C, seed=3)
td = {k: torch.as_tensor(v, device=dev) for k, v in w.items() if isinstance(v, np.ndarray)}
plan = engine.PipelinePlan(td["bin_mu"], td["bin_std"], td["bin_y"], td["bin_flag"], td["bin_ctx"], td["ov_ptr"], td["ov_idx"],
                           td["L"], td["strand_minus"], td["d_pr"], td["obs_snv"], td["obs_samples"], td["obs_indel"])
s = torch.cuda.current_stream(dev)
lib = _lib.load()
fn = lib.dig_debug_es_timing
fn.argtypes = [ctypes.c_void_p]
buf = np.zeros(4096, np.uint64)
fq = lib.dig_debug_es_queue
fq.argtypes = [ctypes.c_void_p]
qbuf = np.zeros(1024, np.uint64)
for _ in range(50):
    plan.run(td["cj"], td["cj_indel"], stages=7, stream=s)
torch.cuda.synchronize()
fn(buf.ctypes.data)
for rep in range(3):
    plan.run(td["cj"], td["cj_indel"], stages=7, stream=s)
    torch.cuda.synchronize()
    fn(buf.ctypes.data)
    t0, t1 = buf[:256].astype(np.int64), buf[1024:1280].astype(np.int64)
    base = t0.min()
    start = (t0 - base) / 100.0          # us
    end = (t1 - base) / 100.0
    print("starts: min %.1f max %.1f | ends: min %.1f p10 %.1f median %.1f p90 %.1f max %.1f us | mean idle at the end %.1f us"
          % (start.min(), start.max(), end.min(), np.percentile(end, 10), np.median(end), np.percentile(end, 90), end.max(),
             (end.max() - end).mean()))
    byx = [(end[x::8].mean(), end[x::8].max()) for x in range(8)]
    print("  per XCD (mean, max):", " ".join("%.0f/%.0f" % b for b in byx))
    b0, b1 = (buf[2048:2304].astype(np.int64) - base) / 100.0, (buf[3072:3328].astype(np.int64) - base) / 100.0
    print("  within a workgroup: first wave out of tiles %.1f, last %.1f (spread %.1f), then %.1f us to the end (means over workgroups)"
          % (b0.mean(), b1.mean(), (b1 - b0).mean(), (end - b1).mean()))
    late = np.argsort(-end)[:8]
    print("  the eight last workgroups: end", np.round(end[late], 1), "last wave out of tiles", np.round(b1[late], 1), "then", np.round((end - b1)[late], 1))
    print("  from the barrier to the end, all workgroups: p10 %.1f median %.1f p90 %.1f max %.1f us" % tuple(np.percentile(end - b1, [10, 50, 90, 100])))
    fq(qbuf.ctypes.data)
    recs, tests = (qbuf[:256] & np.uint64(0xffffffff)).astype(np.int64), (qbuf[:256] >> np.uint64(32)).astype(np.int64)
    print("  queue records per workgroup: min %d median %d p90 %d max %d | open tests: median %d max %d | corr(records, barrier-to-end) %.2f"
          % (recs.min(), np.median(recs), np.percentile(recs, 90), recs.max(), np.median(tests), tests.max(), np.corrcoef(recs, end - b1)[0, 1]))
    print("  the eight last: records", recs[late], "tests", tests[late])
    print("  corr(last wave out of tiles, end) %.2f ; spread of 'last wave out of tiles': min %.1f median %.1f max %.1f"
          % (np.corrcoef(b1, end)[0, 1], b1.min(), np.median(b1), b1.max()))
