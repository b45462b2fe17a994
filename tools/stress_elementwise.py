#!/usr/bin/env python
"""Randomised stress of the elementwise NB entry points (host mirror sequence_model.nb_model) against the oracle over a wide
parameter range (developer tool): mid-p upper, greater, exact, two-sided mid-p, Fisher."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from digdriver_amd.sequence_model import nb_model as M     # noqa: E402
from oracle import dig_oracle as O                         # noqa: E402

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
rng = np.random.default_rng(seed)
n = 200_000
mean = 10 ** rng.uniform(-2, 5, n)
alpha = 10 ** rng.uniform(-2, 6, n)
p = alpha / (alpha + mean)
sd = np.sqrt(mean / p)
k = np.clip(np.rint(mean + rng.uniform(-4, 14, n) * sd), 0, 2e6)


def worst(name, got, want):
    got, want = np.asarray(got, float), np.asarray(want, float)
    assert (np.isnan(got) == np.isnan(want)).all(), name
    ok = np.isfinite(want) & (np.abs(want) >= 1e-250)
    rel = np.where(ok, np.abs(got - want) / np.maximum(np.abs(want), 1e-300), 0)
    i = int(np.argmax(rel))
    small = np.isfinite(want) & (np.abs(want) < 1e-250)
    assert (np.abs(got[small]) < 1.0001e-250).all(), name
    print("%-24s worst rel %.2e at k=%g alpha=%g p=%.17g got=%g want=%g" % (name, rel[i], k[i], alpha[i], p[i], got[i], want[i]))


worst("nb_pvalue_greater_midp", M.nb_pvalue_greater_midp(k, alpha, p), O.nb_pvalue_greater_midp(k, alpha, p))
worst("nb_pvalue_greater", M.nb_pvalue_greater(k, alpha, p), O.nb_pvalue_greater(k, alpha, p))
worst("nb_pvalue_exact", M.nb_pvalue_exact(k, alpha, p), O.nb_pvalue_exact(k, alpha, p))
worst("nb_pvalue_midp", M.nb_pvalue_midp(k, alpha, p), O.nb_pvalue_midp(k, alpha, p))
p1, p2 = 10 ** rng.uniform(-300, 0, n), 10 ** rng.uniform(-30, 0, n)
worst("fisher_combine", M.fisher_combine(p1, p2), O.fisher_combine(p1, p2))
