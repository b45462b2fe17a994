#!/usr/bin/env python
"""Developer probe: the q-value step of the per-base route -- 37 cohorts x 7.2 M p-values (an eighth of the genome) -- through
the library's radix sort + Benjamini-Hochberg pass (dig_bh_qvalues_ragged) against round 5's torch.sort + dig_bh_qvalues_sorted +
scatter_, same bits required."""
import os, sys, time, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from digdriver_amd import _lib
from digdriver_amd.sequence_model import nb_model
dev = torch.device("cuda:0")
rows, n = int(os.environ.get("ROWS", 37)), int(os.environ.get("N", 7_200_000))
g = torch.Generator(device=dev).manual_seed(1)
# p-values of the tiled test: most of them just below 1 (tiles without a mutation), a tail of small ones
p = torch.exp(-torch.rand((rows, n), device=dev, generator=g, dtype=torch.float64) * 0.1)
small = torch.rand((rows, n), device=dev, generator=g) < 0.01
p[small] = torch.rand(int(small.sum()), device=dev, generator=g, dtype=torch.float64) ** 4
if os.environ.get("ROUTE"):                                  # like the route's mid-p values: most of them in [0.5, 0.6), a few small ones
    p = 0.5 + 0.1 * torch.rand((rows, n), device=dev, generator=g, dtype=torch.float64)
    p[small] = torch.rand(int(small.sum()), device=dev, generator=g, dtype=torch.float64) ** 4
if os.environ.get("TIES"):                                   # what the route has: p = 1 wherever a tile has no mutation (99.8 %)
    p[~small] = 1.0
    p[small & (torch.rand((rows, n), device=dev, generator=g) < 0.8)] = 1.0

def timeit(fn, reps=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3, out

def old(p_rows):
    ps, order = torch.sort(p_rows, dim=1, stable=True)
    q = torch.empty_like(ps)
    wsb = int(_lib.load().dig_bh_workspace(n, rows))
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    _lib.call("dig_bh_qvalues_sorted", _lib.dev_ptr(ps), n, rows, _lib.dev_ptr(q), _lib.dev_ptr(ws), wsb, _lib.stream_ptr())
    out = torch.empty_like(q)
    out.scatter_(1, order, q)
    return out

t_new, q_new = timeit(lambda: nb_model.get_q_vals_rows(p))
t_old, q_old = timeit(lambda: old(p))
print(json.dumps({"rows": rows, "n": n, "ms_library_sort_bh": t_new, "ms_torch_sort_bh_scatter": t_old, "same_bits": bool(torch.equal(q_new, q_old)),
                  "bytes_per_element": 176, "tb_per_s": rows * n * 176.0 / (t_new * 1e-3) / 1e12}))
