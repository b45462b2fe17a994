#!/usr/bin/env python
"""Developer probe: every array of the statistics stage (inputs, bin tables, outputs, workspace) carved from ONE device
allocation against the default one-allocation-per-array layout, same process."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from bench import make_workload
from digdriver_amd import engine, _lib
dev = torch.device("cuda:0")
w = make_workload(288000, 120091, 37, seed=3)
td = {k: torch.as_tensor(v, device=dev) for k, v in w.items() if isinstance(v, np.ndarray)}
E, C = 120091, 37
s = torch.cuda.current_stream(dev)
def timed(plan, t, n=24):
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
wsb = _lib.workspace_bytes("pipeline", E, C)
def mk(t, acc, st, ws):
    return engine.PipelinePlan(t["bin_mu"], t["bin_std"], t["bin_y"], t["bin_flag"], t["bin_ctx"], t["ov_ptr"], t["ov_idx"],
                               t["L"], t["strand_minus"], t["d_pr"], t["obs_snv"], t["obs_samples"], t["obs_indel"],
                               out_acc=acc, out_stats=st, workspace=ws)
acc0 = engine.alloc_accumulate_outputs(E, C, 1, dev)
st0 = torch.empty((7, E, C), dtype=torch.float64, device=dev)
ws0 = torch.empty(wsb, dtype=torch.uint8, device=dev)
p0 = mk(td, acc0, st0, ws0)
print("default layout:", [round(timed(p0, td), 1) for _ in range(2)], flush=True)
for align in (2 << 20, 4096, 256):
    arena = torch.empty(3 << 30, dtype=torch.uint8, device=dev)
    off = [0]
    def carve(like):
        nb = like.numel() * like.element_size()
        o = off[0]; off[0] = (o + nb + align - 1) // align * align
        return arena[o:o + nb].view(like.dtype).view(like.shape)
    t2 = {}
    for k, v in td.items():
        t2[k] = carve(v); t2[k].copy_(v)
    acc2 = {k: carve(v) for k, v in acc0.items()}
    st2 = carve(st0); ws2 = carve(ws0)
    p2 = mk(t2, acc2, st2, ws2)
    print("one arena, align %d: %s   (used %.0f MB)" % (align, [round(timed(p2, t2), 1) for _ in range(2)], off[0] / 1e6), flush=True)
    same = all(torch.equal(acc2[k], acc0[k]) for k in ("MU", "R_OBS")) and torch.equal(torch.nan_to_num(st2), torch.nan_to_num(st0))
    print("   same results:", same)
    del p2, t2, acc2, st2, ws2, arena
print("default again:", round(timed(p0, td), 1))
