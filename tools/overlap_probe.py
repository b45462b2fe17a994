#!/usr/bin/env python
"""Do two pipeline passes over different batches overlap on one GPU?  (developer probe, not the judged bench)

    python tools/overlap_probe.py lib.so[:ENV=VAL,...] ...

Per library / environment: time per pass of ONE PipelinePlan on one stream, and of TWO plans (own outputs and workspaces)
alternating on two streams, so that the dot kernel (matrix pipe) of one pass can run beside the statistics kernel (vector
ALU + memory) of the other when their workgroups fit on a CU together."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child():
    import numpy as np
    import torch
    sys.path.insert(0, ROOT)
    from bench import make_workload
    from digdriver_amd import engine
    dev = torch.device("cuda:0")
    E, C = 120091, 37
    w = make_workload(288000, E, C, seed=3)
    td = {k: torch.as_tensor(v, device=dev) for k, v in w.items() if isinstance(v, np.ndarray)}

    def plan():
        return engine.PipelinePlan(td["bin_mu"], td["bin_std"], td["bin_y"], td["bin_flag"], td["bin_ctx"], td["ov_ptr"], td["ov_idx"],
                                   td["L"], td["strand_minus"], td["d_pr"], td["obs_snv"], td["obs_samples"], td["obs_indel"])
    plans = [plan(), plan()]
    streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]

    def run(n, dual):
        torch.cuda.synchronize()
        a = [torch.cuda.Event(enable_timing=True) for _ in streams]
        b = [torch.cuda.Event(enable_timing=True) for _ in streams]
        for s, e in zip(streams, a):
            e.record(s)
        for t in range(n):
            k = t % 2 if dual else 0
            plans[k].run(td["cj"], td["cj_indel"], stages=7, stream=streams[k])
        for s, e in zip(streams, b):
            e.record(s)
        torch.cuda.synchronize()
        return max(a[0].elapsed_time(x) for x in b) / n * 1e3

    run(100, False), run(100, True)
    res = {"single": [], "dual": []}
    for _ in range(4):
        res["single"].append(run(200, False))
        res["dual"].append(run(200, True))
    same = bool(torch.equal(plans[0].stats, plans[1].stats))
    print("RESULT " + json.dumps({"single_us": round(min(res["single"]), 1), "dual_us": round(min(res["dual"]), 1),
                                  "dual_all": [round(x, 1) for x in res["dual"]], "plans_agree": same}), flush=True)


def main():
    if sys.argv[1] == "--child":
        return child()
    for spec in sys.argv[1:]:
        lib, _, envs = spec.partition(":")
        env = dict(os.environ)
        if lib != "default":
            env["DIG_HIP_LIB"] = lib if os.path.isabs(lib) else os.path.join(ROOT, "digdriver_amd", "lib", "variants", lib)
        env.update(dict(kv.split("=") for kv in envs.split(",") if kv))
        p = subprocess.run([sys.executable, __file__, "--child"], env=env, capture_output=True, text=True)
        line = [l for l in p.stdout.split("\n") if l.startswith("RESULT ")]
        print("%-60s %s" % (spec, line[0][7:] if line else "FAILED\n" + p.stdout[-1500:] + p.stderr[-3000:]), flush=True)


if __name__ == "__main__":
    main()
