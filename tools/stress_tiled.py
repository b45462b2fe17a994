#!/usr/bin/env python
"""Randomised stress of dig_tiled_nb_test (nb_pvalue_exact, two-sided) against the oracle over a wide range (developer tool)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from digdriver_amd import engine                   # noqa: E402
from oracle import dig_oracle as O                 # noqa: E402

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
rng = np.random.default_rng(seed)
C, nb, nt = 4, 400, 50
mu = 10 ** rng.uniform(-1, 5, (C, nb))
alpha = 10 ** rng.uniform(-1.5, 5, (C, nb))
sigma = mu / np.sqrt(alpha)
pt = rng.dirichlet(np.ones(nt) * 0.3, size=nb)
mean = mu[:, :, None] * pt[None]
sd = np.sqrt(mean * (1 + mean / alpha[:, :, None]))
k = np.clip(np.rint(mean + rng.uniform(-4, 12, mean.shape) * sd), 0, 1e6).astype(np.int32)
pval, ex = engine.tiled_nb_test(pt, k, mu, sigma)
worst = 0.0
for c in range(C):
    wp, we = O.tiled_nb_test(pt, k[c], mu[c], sigma[c])
    assert np.array_equal(ex[c], we)
    g = np.asarray(pval[c], float)
    assert (np.isnan(g) == np.isnan(wp)).all()
    ok = np.isfinite(wp) & (np.abs(wp) >= 1e-250)
    rel = np.where(ok, np.abs(g - wp) / np.maximum(np.abs(wp), 1e-300), 0)
    i = np.unravel_index(np.argmax(rel), rel.shape)
    if rel[i] > worst:
        worst = rel[i]
        info = (c, i, mu[c, i[0]], alpha[c, i[0]], pt[i], k[c][i], g[i], wp[i])
    small = np.isfinite(wp) & (np.abs(wp) < 1e-250)
    assert (np.abs(g[small]) < 1.0001e-250).all()
print("seed", seed, "worst rel", worst, info)
