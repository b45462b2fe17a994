#!/usr/bin/env python
"""Developer probe: six buffer sets, the fastest and the slowest found, then hybrids of the two (accumulation outputs /
statistics planes / workspace swapped one at a time), each timed three times round-robin."""
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
def timed(plan, n=30):
    for _ in range(4): plan.run(td["cj"], td["cj_indel"], stages=7, stream=s)
    torch.cuda.synchronize()
    evs = []
    for _ in range(n):
        plan.run(td["cj"], td["cj_indel"], stages=2, stream=s)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(s); plan.run(td["cj"], td["cj_indel"], stages=4, stream=s); b.record(s)
        evs.append((a, b))
    torch.cuda.synchronize()
    return sorted(a.elapsed_time(b) for a, b in evs)[n // 2] * 1e3
wsb = _lib.workspace_bytes("pipeline", E, C)
def mk(acc, st, ws):
    return engine.PipelinePlan(td["bin_mu"], td["bin_std"], td["bin_y"], td["bin_flag"], td["bin_ctx"], td["ov_ptr"], td["ov_idx"],
                               td["L"], td["strand_minus"], td["d_pr"], td["obs_snv"], td["obs_samples"], td["obs_indel"], out_acc=acc, out_stats=st, workspace=ws)
sets, keep = [], []
for k in range(6):
    if k % 2 == 1:
        keep.append(torch.empty(int(37e6) + 4096 * k, dtype=torch.uint8, device=dev))
    acc = engine.alloc_accumulate_outputs(E, C, 1, dev)
    st = torch.empty((7, E, C), dtype=torch.float64, device=dev)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    sets.append((acc, st, ws, mk(acc, st, ws)))
t1 = [timed(x[3]) for x in sets]; t2 = [timed(x[3]) for x in sets]
print("round 1", [round(v, 1) for v in t1]); print("round 2", [round(v, 1) for v in t2], flush=True)
F = int(np.argmin(t2)); S = int(np.argmax(t2))
aF, sF, wF, _ = sets[F]; aS, sS, wS, _ = sets[S]
hy = {"F": sets[F][3], "S": sets[S][3], "accF stS wsS": mk(aF, sS, wS), "accS stF wsS": mk(aS, sF, wS), "accS stS wsF": mk(aS, sS, wF),
      "accF stF wsS": mk(aF, sF, wS), "accS stF wsF": mk(aS, sF, wF), "accF stS wsF": mk(aF, sS, wF)}
for name in ("MU", "SIGMA", "R_OBS", "FLAG", "P"):
    mix = dict(aS); mix[name] = aF[name]
    hy["S with F." + name] = mk(mix, sS, wS)
for rnd in range(3):
    print("hybrids round", rnd, {k: round(timed(p), 1) for k, p in hy.items()}, flush=True)
print("addresses F: stats@%x MU@%x P@%x ws@%x | S: stats@%x MU@%x P@%x ws@%x" % (sF.data_ptr(), aF["MU"].data_ptr(), aF["P"].data_ptr(), wF.data_ptr(),
      sS.data_ptr(), aS["MU"].data_ptr(), aS["P"].data_ptr(), wS.data_ptr()))
