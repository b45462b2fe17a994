#!/usr/bin/env python
"""A/B builds of dig_tiled_nb_test (developer tool):  python tools/tiled_nb_bench.py base.so other.so ...
Workload: 37 cohorts x 36 000 bins x 200 tiles, counts Poisson(mu pt) as bench.py's aux leg draws them; preallocated outputs."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

def child(ref, write):
    import numpy as np, torch
    sys.path.insert(0, ROOT)
    from digdriver_amd import _lib
    dev = torch.device("cuda:0")
    torch.manual_seed(5)
    g = torch.Generator(device=dev).manual_seed(5)
    C, nb, nt = 37, int(os.environ.get("TN_BINS", 36000)), 200
    pt = torch.rand((C, nb, nt), device=dev, dtype=torch.float64, generator=g) * 2e-3 + 4e-3
    mu = torch.rand((C, nb), device=dev, dtype=torch.float64, generator=g) * 40 + 5
    sg = torch.rand((C, nb), device=dev, dtype=torch.float64, generator=g) * 6 + 1
    k = torch.poisson(mu[:, :, None] * pt).to(torch.int32)
    pv, ex = torch.empty_like(pt), torch.empty_like(pt)
    p = _lib.dev_ptr
    def run():
        _lib.call("dig_tiled_nb_test", p(pt), 1, p(k), p(mu), p(sg), p(pv), p(ex), C, nb, nt, _lib.stream_ptr())
    run(); torch.cuda.synchronize()
    ts = []
    for _ in range(7):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); run(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    ts.sort()
    out = {"ms": ts[len(ts) // 2], "frac_hbm": (28.0 * C * nb * nt + 16.0 * C * nb) / (ts[len(ts) // 2] * 1e-3) / 8e12}
    got = pv[:, :2000].cpu().numpy()
    if write: np.save(ref, got)
    else:
        want = np.load(ref)
        out["bit_equal"] = bool(np.array_equal(got, want, equal_nan=True))
        out["max_rel"] = float(np.nanmax(np.abs(got - want) / np.maximum(np.abs(want), 1e-300)))
    print(json.dumps(out))

if __name__ == "__main__":
    if sys.argv[1] == "--child":
        child(sys.argv[2], sys.argv[3] == "1"); sys.exit(0)
    ref = "/tmp/tn_ref.npy"
    for i, spec in enumerate(sys.argv[1:]):
        path = spec if os.path.isabs(spec) else os.path.join(ROOT, "tools/variants", spec)
        env = dict(os.environ, DIG_HIP_LIB=path)
        r = subprocess.run([sys.executable, __file__, "--child", ref, "1" if i == 0 else "0"], env=env, capture_output=True, text=True)
        print("%-28s %s" % (spec, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-400:]), flush=True)
