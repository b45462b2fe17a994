#!/usr/bin/env python
"""Developer probe: does the ORDER in which a plan's outputs are allocated decide the statistics stage's time?  (bench.py's loop
plan -- planes first, then the rate outputs -- ran 145 us where the sequential plan -- rate outputs first -- ran 135, same
process, common kind of GPU.)  Several plans over the same inputs and records, outputs allocated in different orders."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from bench import make_workload
from digdriver_amd import engine
dev = torch.device("cuda:0")
w = make_workload(288000, 120091, 37, seed=3)
E, C = 120091, 37
s = torch.cuda.current_stream(dev)
def timed(plan, n=30):
    for _ in range(3): plan.run(td["cj"], td["cj_indel"], stages=7, stream=s)
    torch.cuda.synchronize()
    evs = []
    for _ in range(n):
        plan.run(td["cj"], td["cj_indel"], stages=2, stream=s)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(s); plan.run(td["cj"], td["cj_indel"], stages=4, stream=s); b.record(s)
        evs.append((a, b))
    torch.cuda.synchronize()
    return sorted(a.elapsed_time(b) for a, b in evs)[n // 2] * 1e3
order = os.environ.get("PP_FIRST", "stats")       # what bench.py allocates before the workload goes up
pre_stats = torch.empty((7, E, C), dtype=torch.float64, device=dev) if order == "stats" else None
pre_acc = engine.alloc_accumulate_outputs(E, C, 1, dev) if order == "stats" else None
td = {k: torch.as_tensor(v, device=dev) for k, v in w.items() if isinstance(v, np.ndarray)}
args = (td["bin_mu"], td["bin_std"], td["bin_y"], td["bin_flag"], td["bin_ctx"], td["ov_ptr"], td["ov_idx"], td["L"], td["strand_minus"],
        td["d_pr"], td["obs_snv"], td["obs_samples"], td["obs_indel"])
plans = []
if pre_stats is not None:
    plans.append(("allocated before the inputs, planes first", engine.PipelinePlan(*args, out_acc=pre_acc, out_stats=pre_stats)))
first = plans[0][1] if plans else True
plans.append(("plan's own (rate outputs, then planes)", engine.PipelinePlan(*args, pack_bins=first)))
first = plans[0][1]
st = torch.empty((7, E, C), dtype=torch.float64, device=dev); acc = engine.alloc_accumulate_outputs(E, C, 1, dev)
plans.append(("after the inputs, planes first", engine.PipelinePlan(*args, out_acc=acc, out_stats=st, pack_bins=first)))
acc = engine.alloc_accumulate_outputs(E, C, 1, dev); pad = torch.empty(3 << 20, dtype=torch.uint8, device=dev); st = torch.empty((7, E, C), dtype=torch.float64, device=dev)
plans.append(("rate outputs, 3 MB pad, planes", engine.PipelinePlan(*args, out_acc=acc, out_stats=st, pack_bins=first)))
n = E * C
def slab_plan(order, pad_doubles=0, with_p=False):
    """outputs carved from ONE allocation in `order` (names; 'planes' = the seven statistics planes), `pad_doubles` between them"""
    total = sum({"MU": n, "SIGMA": n, "planes": 7 * n, "R_OBS": n // 2 + 8, "FLAG": n // 2 + 8, "P": n}[k] + pad_doubles for k in order) + 64
    slab = torch.empty(total, dtype=torch.float64, device=dev)
    acc = engine.alloc_accumulate_outputs(E, C, 1, dev)
    st, off = None, 0
    for k in order:
        if k == "planes":
            st = slab[off:off + 7 * n].view(7, E, C); off += 7 * n
        elif k in ("MU", "SIGMA"):
            acc[k] = slab[off:off + n].view(E, C); off += n
        elif k == "P":
            acc[k] = slab[off:off + n].view(E, 1, C); off += n
        else:
            acc[k] = slab[off:off + n // 2 + 8].view(torch.int32)[:n].view(E, C); off += n // 2 + 8
        off += pad_doubles
    return engine.PipelinePlan(*args, out_acc=acc, out_stats=st, pack_bins=first)
plans.append(("slab: MU SIGMA planes R_OBS FLAG", slab_plan(["MU", "SIGMA", "planes", "R_OBS", "FLAG"])))
def staggered_plan(step):
    """separate allocations like the default, the base of output k moved by k * step bytes (de-correlates the low address bits)"""
    acc = engine.alloc_accumulate_outputs(E, C, 1, dev)
    keep = []
    def moved(k, shape, dtype):
        nb = int(np.prod(shape)) * torch.empty(0, dtype=dtype).element_size()
        raw = torch.empty(nb + (2 << 20), dtype=torch.uint8, device=dev)
        keep.append(raw)
        o = (k * step) % (2 << 20) // 8 * 8
        return raw[o:o + nb].view(dtype).view(shape)
    acc["MU"], acc["SIGMA"] = moved(1, (E, C), torch.float64), moved(2, (E, C), torch.float64)
    acc["R_OBS"], acc["FLAG"] = moved(3, (E, C), torch.int32), moved(4, (E, C), torch.int32)
    acc["P"] = moved(5, (E, 1, C), torch.float64)
    st = moved(6, (7, E, C), torch.float64)
    pl = engine.PipelinePlan(*args, out_acc=acc, out_stats=st, pack_bins=first)
    pl._keep_raw = keep
    return pl
for step in (0, 256, 4352, 69888, 266496):
    plans.append(("separate, bases moved by k x %d" % step, staggered_plan(step)))
for rnd in range(2):
    for name, p in plans:
        print("round %d  %-45s %6.1f us   MU@%x planes@%x" % (rnd, name, timed(p), p.acc["MU"].data_ptr(), p.stats.data_ptr()), flush=True)
