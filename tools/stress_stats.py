#!/usr/bin/env python
"""Randomised stress of the statistics block against the oracle over a wide parameter range (developer tool): rates from
1e-2 to 1e5, dispersion alpha from 1e-2 to 1e6, counts from far below to far above the mean.  Prints the worst relative
difference per plane and the share of pairs the compacted pass handled."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import rel_close                     # noqa: E402
from digdriver_amd import engine                   # noqa: E402
from oracle import dig_oracle as O                 # noqa: E402


def main():
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    rng = np.random.default_rng(seed)
    E, C = 6000, 37
    mean = 10 ** rng.uniform(-2, 5, (E, C))
    alpha = 10 ** rng.uniform(-2, 6, (E, C))
    mu = mean.copy()
    sigma = mu / np.sqrt(alpha)                                  # alpha = mu^2 / sigma^2
    pi = np.ones((E, C))
    pii = np.ones(E)
    cj, cji = np.ones(C), np.ones(C)
    sd = np.sqrt(mean * (1 + mean / alpha))
    z = rng.uniform(-4, 14, (E, C))
    k1 = np.clip(np.rint(mean + z * sd), 0, 2e6).astype(np.int32)
    k2 = np.clip(np.rint(k1 * rng.uniform(0.5, 1.0, (E, C))), 0, None).astype(np.int32)
    k3 = np.clip(np.rint(mean + rng.uniform(-3, 8, (E, C)) * sd), 0, 2e6).astype(np.int32)
    got = engine.element_stats(mu, sigma, pi, pii, k1, k2, k3, cj, cji)
    want = O.element_stats(mu, sigma, pi, pii[:, None], k1, k2, k3, cj[None, :], cji[None, :])
    worst = {}
    for name in engine.ES_PLANES:
        g, w = np.asarray(got[name], float), np.asarray(want[name], float)
        ok = np.isfinite(w) & (np.abs(w) >= 1e-250)
        rel = np.abs(g[ok] - w[ok]) / np.abs(w[ok])
        worst[name] = float(rel.max()) if rel.size else 0.0
        i = np.unravel_index(np.argmax(np.where(ok, np.abs(g - w) / np.maximum(np.abs(w), 1e-300), 0)), w.shape)
        if worst[name] > 1e-7:
            print("BAD", name, worst[name], "at", i, "mu", mu[i], "alpha", alpha[i], "k", k1[i], k2[i], k3[i], "got", g[i], "want", w[i])
        small = np.isfinite(w) & (np.abs(w) < 1e-250)
        assert (np.abs(g[small]) < 1.0001e-250).all(), name
        assert (np.isnan(g) == np.isnan(w)).all(), name
    print("seed", seed, "worst rel per plane:", {k: "%.2e" % v for k, v in worst.items()})


if __name__ == "__main__":
    main()
