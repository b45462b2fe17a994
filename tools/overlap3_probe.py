#!/usr/bin/env python
"""Developer probe: N statistics kernels on one stream and N accumulation kernels (independent plan) on another, no events
between them: wall time of both together against each alone.  Tells whether the two kernels are ever co-resident."""
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")


def child():
    import numpy as np, torch
    sys.path.insert(0, ROOT)
    from bench import make_workload
    from digdriver_amd import engine
    dev = torch.device("cuda:0")
    w = make_workload(288000, 120091, 37, seed=3)
    td = {k: torch.as_tensor(v, device=dev) for k, v in w.items() if isinstance(v, np.ndarray)}
    mk = lambda: engine.PipelinePlan(td["bin_mu"], td["bin_std"], td["bin_y"], td["bin_flag"], td["bin_ctx"], td["ov_ptr"], td["ov_idx"],
                                     td["L"], td["strand_minus"], td["d_pr"], td["obs_snv"], td["obs_samples"], td["obs_indel"])
    A, B = mk(), mk()
    main = torch.cuda.current_stream(dev)
    side = torch.cuda.Stream(device=dev, priority=-1)
    cj, cji = td["cj"], td["cj_indel"]
    A.run(cj, cji, stages=7, stream=main)
    B.run(cj, cji, stages=7, stream=main)
    torch.cuda.synchronize()
    n = 200
    def wall(fn):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e6
    stats_only = wall(lambda: [A.run(cj, cji, stages=4, stream=main) for _ in range(n)])
    dot_only = wall(lambda: [B.run(cj, cji, stages=2, stream=side) for _ in range(n)])
    def both():
        for _ in range(n):
            A.run(cj, cji, stages=4, stream=main)
            B.run(cj, cji, stages=2, stream=side)
    together = wall(both)
    print("RESULT " + json.dumps({"stats_only_us": stats_only, "dot_only_us": dot_only, "both_us_per_pair": together,
                                  "env": {k: os.environ.get(k) for k in ("DIG_ES_TICKETS", "DIG_CTX_LIGHT")}}), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        child()
    else:
        for env in ({}, {"DIG_ES_TICKETS": "768", "DIG_CTX_LIGHT": "1"}, {"DIG_ES_TICKETS": "768"}, {"DIG_CTX_LIGHT": "1"}):
            p = subprocess.run([sys.executable, __file__, "--child"], env=dict(os.environ, **env), capture_output=True, text=True)
            line = [l for l in p.stdout.split("\n") if l.startswith("RESULT ")]
            print(line[0][7:] if line else "FAILED " + p.stdout[-800:] + p.stderr[-2000:], flush=True)
