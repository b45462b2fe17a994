#!/usr/bin/env python
"""Developer probe: where do the 22 ms of one cohort's Benjamini-Hochberg pass (nb_model.get_q_vals on 7.2 M p-values) go?"""
import sys, time, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
dev = torch.device("cuda:0")
n = 7_200_000
g = torch.Generator(device=dev).manual_seed(1)
p = torch.rand(n, device=dev, generator=g, dtype=torch.float64)
mask = torch.ones(n, dtype=torch.bool, device=dev)

def t(name, fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    print("%-28s %8.3f ms" % (name, (time.perf_counter() - t0) / reps * 1e3))
    return out

ps, order = t("sort stable", lambda: torch.sort(p, stable=True))
t("sort unstable", lambda: torch.sort(p))
ar = t("arange / n", lambda: torch.arange(1, n + 1, device=dev, dtype=torch.float64) / torch.full((), float(n), device=dev, dtype=torch.float64))
q = t("divide", lambda: ps / ar)
qf = t("flip", lambda: torch.flip(q, [0]))
cm = t("cummin", lambda: torch.cummin(qf, 0).values)
t("flip back + clamp", lambda: torch.clamp(torch.flip(cm, [0]), max=1.0))
out = torch.empty_like(q)
def scat():
    out[order] = q
    return out
t("scatter out[order] = q", scat)
t("masked select p[mask]", lambda: p[mask])
def mset():
    o = torch.full((n,), float("nan"), dtype=torch.float64, device=dev)
    o[mask] = q
    return o
t("masked assign", mset)
from digdriver_amd.sequence_model import nb_model
t("get_q_vals whole", lambda: nb_model.get_q_vals(p))
