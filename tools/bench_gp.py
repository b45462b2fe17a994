#!/usr/bin/env python
"""GP calibration stage (a5/a6) on MI355X: one SGPR fit + predictions at the reference's sizes (gp_trainer.py:54-204:
150 000 training rows cap, 16 CNN features, m inducing points, n_iter Adam steps; held-out ~ N / k bins).  Developer tool;
prints one JSON object (committed under profiles/)."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from digdriver_amd.region_model.trainers.gp_trainer import GPTrainer     # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--train", type=int, default=150_000)
    ap.add_argument("--heldout", type=int, default=57_600)
    ap.add_argument("--inducing", type=int, default=400)
    ap.add_argument("--iters", type=int, default=50)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(3)
    n = a.train + 2000 + a.heldout
    X = rng.normal(size=(n, 16))
    w = rng.normal(size=16)
    y = 30 + 8 * np.tanh(X @ w / 3) + rng.normal(0, 1.0, n)
    tr, va, ho = slice(0, a.train), slice(a.train, a.train + 2000), slice(a.train + 2000, n)
    out = []
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        g = GPTrainer(dev, (X[tr], y[tr]), (X[va], y[va]), (X[ho], y[ho]), n_iter=a.iters, n_inducing=a.inducing)
        val, hld = g.run()
        torch.cuda.synchronize()
        out.append(time.perf_counter() - t0)
    flops = a.iters * 3.0 * (2.0 * a.train * a.inducing * (16 + a.inducing))     # K_nm + A = L^-1 K_mn + A A^T, fwd + bwd
    print(json.dumps({"stage": "SGPR fit + predict (GPTrainer.run)", "train_rows": a.train, "heldout_rows": a.heldout,
                      "inducing": a.inducing, "iters": a.iters, "seconds": out[-1], "first_run_seconds": out[0],
                      "heldout_r2": float(hld["r2"]), "approx_tflops": flops / out[-1] / 1e12,
                      "per_cohort_x37_x5folds_s": out[-1] * 37 * 5}))


if __name__ == "__main__":
    main()
