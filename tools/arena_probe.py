#!/usr/bin/env python
"""Developer probe: every array of the statistics stage carved from ONE device allocation made FIRST THING in the process
(before any other device memory exists), against the default one-allocation-per-array layout made afterwards."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from bench import make_workload
from digdriver_amd import engine, _lib
dev = torch.device("cuda:0")
E, C = 120091, 37
arena = torch.empty(int(os.environ.get("AP_GB", "3")) << 30, dtype=torch.uint8, device=dev)        # first allocation of the process
w = make_workload(288000, E, C, seed=3)
s = torch.cuda.current_stream(dev)
def timed(plan, t, n=30):
    for _ in range(3): plan.run(t["cj"], t["cj_indel"], stages=7, stream=s)
    torch.cuda.synchronize()
    evs = []
    for _ in range(n):
        plan.run(t["cj"], t["cj_indel"], stages=2, stream=s)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(s); plan.run(t["cj"], t["cj_indel"], stages=4, stream=s); b.record(s)
        evs.append((a, b))
    torch.cuda.synchronize()
    return sorted(a.elapsed_time(b) for a, b in evs)[n // 2] * 1e3
off = [0]
align = int(os.environ.get("AP_ALIGN", "256"))
def carve(shape, dtype):
    nb = int(np.prod(shape)) * torch.empty(0, dtype=dtype).element_size()
    o = off[0]; off[0] = (o + nb + align - 1) // align * align
    return arena[o:o + nb].view(dtype).view(shape)
t2 = {}
for k, v in w.items():
    if isinstance(v, np.ndarray):
        tv = torch.as_tensor(v)
        t2[k] = carve(tuple(tv.shape), tv.dtype); t2[k].copy_(tv)
acc_like = dict(MU=((E, C), torch.float64), SIGMA=((E, C), torch.float64), R_OBS=((E, C), torch.int32), FLAG=((E, C), torch.int32),
                P=((E, 1, C), torch.float64), R_SIZE=((E,), torch.int32), ELT_SIZE=((E,), torch.int32), P_INDEL=((E,), torch.float64))
acc2 = {k: carve(*v) for k, v in acc_like.items()}
st2 = carve((7, E, C), torch.float64)
ws2 = carve((_lib.workspace_bytes("pipeline", E, C),), torch.uint8)
def mk(t, acc, st, ws, pack=True):
    return engine.PipelinePlan(t["bin_mu"], t["bin_std"], t["bin_y"], t["bin_flag"], t["bin_ctx"], t["ov_ptr"], t["ov_idx"],
                               t["L"], t["strand_minus"], t["d_pr"], t["obs_snv"], t["obs_samples"], t["obs_indel"],
                               out_acc=acc, out_stats=st, workspace=ws, pack_bins=pack)
p2 = mk(t2, acc2, st2, ws2)
print("arena made first (%d-byte alignment, %.0f MB used; records allocated later): %s" % (align, off[0] / 1e6, [round(timed(p2, t2), 1) for _ in range(3)]), flush=True)
td = {k: torch.as_tensor(v, device=dev) for k, v in w.items() if isinstance(v, np.ndarray)}
p0 = engine.PipelinePlan(td["bin_mu"], td["bin_std"], td["bin_y"], td["bin_flag"], td["bin_ctx"], td["ov_ptr"], td["ov_idx"], td["L"],
                         td["strand_minus"], td["d_pr"], td["obs_snv"], td["obs_samples"], td["obs_indel"])
print("default layout, allocated afterwards:", [round(timed(p0, td), 1) for _ in range(3)], flush=True)
p3 = mk(td, acc2, st2, ws2, pack=p0)
print("inputs default, outputs + workspace in the arena:", [round(timed(p3, td), 1) for _ in range(2)], flush=True)
print("arena again:", round(timed(p2, t2), 1))
