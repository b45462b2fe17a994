import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_workload
from digdriver_amd import engine, _lib
dev = torch.device("cuda:0")
w = make_workload(288000, 120091, 37, seed=3)
td = {k: torch.as_tensor(v, device=dev) for k, v in w.items() if isinstance(v, np.ndarray)}
plan = engine.PipelinePlan(td["bin_mu"], td["bin_std"], td["bin_y"], td["bin_flag"], td["bin_ctx"], td["ov_ptr"], td["ov_idx"],
                           td["L"], td["strand_minus"], td["d_pr"], td["obs_snv"], td["obs_samples"], td["obs_indel"])
s = torch.cuda.current_stream(dev)
for _ in range(20): plan.run(td["cj"], td["cj_indel"], stages=7, stream=s)
fn = _lib.load().dig_debug_slow_profile
buf = (ctypes.c_ulonglong * 8)()
fn(buf)
plan.run(td["cj"], td["cj_indel"], stages=7, stream=s)
fn(buf)
v = list(buf)
print("waves", v[7], "span cycles (first start -> last end)", v[4], "last wave start after first", v[5])
print("cycles per wave: init %.0f | load %.0f tests %.0f write %.0f" % (v[0]/v[7], v[1]/v[7], v[2]/v[7], v[3]/v[7]))
ev=[torch.cuda.Event(enable_timing=True) for _ in range(2)]
ev[0].record(s)
for _ in range(50): plan.run(td["cj"], td["cj_indel"], stages=7, stream=s)
ev[1].record(s); torch.cuda.synchronize(); print("pipe us", ev[0].elapsed_time(ev[1])*20)
