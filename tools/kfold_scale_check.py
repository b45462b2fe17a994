#!/usr/bin/env python
"""Mid-scale run of the region-model orchestration (k-fold CNN training + SGPR calibration + fold assembly) on synthetic
tracks: 40 000 bins x 100 positions x T tracks, several cohorts.  Developer tool: catches what the tiny test matrix
cannot (the SGPR at tens of thousands of rows, the HBM-resident track store at GB scale)."""
import argparse
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bins", type=int, default=40000)
    ap.add_argument("--tracks", type=int, default=120)
    ap.add_argument("--cohorts", type=int, default=3)
    ap.add_argument("--epochs", type=int, default=2)
    a = ap.parse_args()
    from digdriver_amd.io import mapfile
    from digdriver_amd.region_model import kfold_mutations_main as kf, region_model_tools
    rng = np.random.default_rng(4)
    N, L, T = a.bins, 100, a.tracks
    base = rng.uniform(0, 1, (N, 1, T)).astype(np.float32)
    x = np.round(np.clip(base + 0.15 * rng.normal(size=(N, L, T)).astype(np.float32), 0, 1), 2) * 100
    tmp = tempfile.mkdtemp(prefix="kfold_scale_")
    data = os.path.join(tmp, "train.map")
    mapfile.write_array(data, "x_data", x.astype(np.float32))
    mapfile.write_array(data, "idx", np.stack([np.ones(N, int), np.arange(N) * 10000, (np.arange(N) + 1) * 10000], 1))
    mapfile.write_array(data, "mappability", rng.uniform(0.3, 1.0, N))
    names = ["COHORT_%d" % c for c in range(a.cohorts)]
    for c, nme in enumerate(names):
        y = np.rint((30 + 10 * c) * base[:, 0, c] + 25 * base[:, 0, c + 1] ** 2 + rng.normal(0, 1.0, N) + 5).clip(0)
        mapfile.write_array(data, nme, y)
    t0 = time.time()
    args = kf.get_cmd_arguments("-c %s -d %s -o %s -k 2 -e %d -b 512 -gp 1 -nd 400 -nt 50 -gd 0.5 -u --seed 1" %
                                (" ".join(names), data, tmp, a.epochs))
    out_dir = kf.main(args)
    dt = time.time() - t0
    for nme in names:
        df = region_model_tools.kfold_results(out_dir, nme)
        ok = ~df.FLAG.values.astype(bool)
        r = np.corrcoef(df.Y_TRUE.values[ok], df.Y_PRED.values[ok])[0, 1]
        print("%s: %d bins, Pearson r %.3f, mean STD %.2f" % (nme, len(df), r, df.STD.mean()))
        assert len(df) == N and np.isfinite(df.Y_PRED.values).all() and (df.STD.values > 0).all() and r > 0.5
    print("k-fold orchestration on %d bins x %d tracks x %d cohorts: %.1f s" % (N, T, a.cohorts, dt))


if __name__ == "__main__":
    main()
