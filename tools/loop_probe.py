#!/usr/bin/env python
"""Developer probe: what in bench.py's loop makes a step longer than the bare pipeline pass?  One plan, passes back to back on
one stream (a), then the loop's ingredients added one at a time."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from bench import make_workload
from digdriver_amd import engine, parallel
dev = torch.device("cuda:0")
E, C = 120091, 37
wg = make_workload(288000, E, C, seed=3)
w = parallel.shard_inputs(wg, parallel.plan_shards(wg["ov_ptr"], wg["ov_idx"], 288000, 1)[0], 1)
w["cj"], w["cj_indel"] = wg["cj"], wg["cj_indel"]
td = {k: torch.as_tensor(v, device=dev) for k, v in w.items() if isinstance(v, np.ndarray) and k not in ("chunk_rows", "elements")}
plan = engine.PipelinePlan(td["bin_mu"], td["bin_std"], td["bin_y"], td["bin_flag"], td["bin_ctx"], td["ov_ptr"], td["ov_idx"],
                           td["L"], td["strand_minus"], td["d_pr"], td["obs_snv"], td["obs_samples"], td["obs_indel"])
div = int(os.environ.get("LP_DIV", "1"))          # 1 / div of the bins in the side stream's sums (developer: what does the side work cost per byte?)
scale = engine.ChunkedScaleFactorPlan(td["bin_mu"], td["bin_flag"], td["n_snv_obs"], td["n_ind_obs"], np.asarray(w["chunk_rows"]) // div, parallel.N_CHUNKS, world=1)
pr = os.environ.get("LP_PRIO") == "1"
main, side = torch.cuda.Stream(dev, priority=-1 if pr else 0), torch.cuda.Stream(dev, priority=0)
print("stream priorities (main, side):", main.priority, side.priority, torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else "")
RING = 64
cjs = [(torch.empty(C, dtype=torch.float64, device=dev), torch.empty(C, dtype=torch.float64, device=dev)) for _ in range(RING)]
done = [torch.cuda.Event() for _ in range(RING)]
thr = [torch.cuda.Event() for _ in range(8)]
for b in range(RING):
    scale.run(cjs[b][0], cjs[b][1], stream=main)
torch.cuda.synchronize()

def loop(n, side_work, wait, throttle, lead=3):
    torch.cuda.synchronize()
    a, z = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    queued = -1
    a.record(main)
    for t in range(n):
        b = t % RING
        if throttle and t % 16 == 0:
            k = t // 16
            thr[k % 8].record(main)
            if k >= 3:
                thr[(k - 3) % 8].synchronize()
        while queued < min(t + lead, n - 1):
            queued += 1
            qb = queued % RING
            if side_work:
                with torch.cuda.stream(side):
                    scale.run(cjs[qb][0], cjs[qb][1], stream=side)
                    done[qb].record(side)
            elif wait:
                done[qb].record(side)
        if wait:
            main.wait_event(done[b])
        plan.run(cjs[b][0], cjs[b][1], stages=7, stream=main)
    z.record(main)
    torch.cuda.synchronize()
    return a.elapsed_time(z) / n * 1e3

loop(200, True, True, True)
print("library:", os.environ.get("DIG_HIP_LIB"))
for name, cfg in (("bare passes", (False, False, False)), ("+ throttle events", (False, False, True)), ("+ wait on a side-stream event", (False, True, True)),
                  ("+ scale factors on the side stream (bench.py's loop)", (True, True, True)), ("side work, no throttle", (True, True, False)),
                  ("bare passes again", (False, False, False))):
    print("%-55s %s us per step" % (name, [round(loop(600, *cfg), 1) for _ in range(3)]), flush=True)
