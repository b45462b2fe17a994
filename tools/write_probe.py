#!/usr/bin/env python
"""Where does writing the C results.txt files go?  (developer probe for the e2e stage `write_results_txt`)"""
import os, sys, time
import numpy as np, pandas as pd
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from digdriver_amd.driver_model import cohort_batch
from digdriver_amd.io import mapfile
E, C = 120_091, 37
rng = np.random.default_rng(0)
idx = pd.Index(["ELT%06d" % i for i in range(E)], name="ELT")
cols = ["R_SIZE", "R_OBS", "R_INDEL", "MU", "SIGMA", "ALPHA", "THETA", "MU_INDEL", "SIGMA_INDEL", "ALPHA_INDEL", "THETA_INDEL", "FLAG", "Pi_SUM", "Pi_INDEL",
        "OBS_SAMPLES", "OBS_SNV", "OBS_INDEL", "EXP_SNV", "EXP_INDEL", "PVAL_SNV_BURDEN", "PVAL_INDEL_BURDEN", "PVAL_MUT_BURDEN", "PVAL_SAMPLE_BURDEN", "N_SAMP"]
def frame():
    d = {}
    for c in cols:
        if c.startswith("OBS") or c in ("R_SIZE", "R_OBS", "R_INDEL", "N_SAMP"): d[c] = rng.integers(0, 50, E)
        elif c == "FLAG": d[c] = rng.random(E) < 0.1
        else: d[c] = rng.random(E) * 10.0 ** rng.integers(-8, 3, E)
    return pd.DataFrame(d, index=idx)
frames = [frame() for _ in range(C)]
for outdir in ("/tmp/dig_write_probe", "/dev/shm/dig_write_probe"):
    for workers in (1, 8, 32):
        t0 = time.perf_counter()
        out = cohort_batch.write_results(frames, outdir, ["c%02d" % i for i in range(C)], workers=workers)
        dt = time.perf_counter() - t0
        mb = sum(os.path.getsize(p) for p in out) / 1e6
        print("%s workers=%2d: %.3f s for %.0f MB (%.2f GB/s)" % (outdir, workers, dt, mb, mb / dt / 1e3))
    import shutil; shutil.rmtree(outdir, ignore_errors=True)
# the pieces of one file
df = frames[0]
t0 = time.perf_counter(); ints = {c: df[c].astype(int) for c in ('OBS_SAMPLES', 'OBS_SNV', 'OBS_INDEL')}; d2 = df.assign(**ints); t1 = time.perf_counter()
mapfile.write_results_tsv(d2, "/dev/shm/one.txt"); t2 = time.perf_counter()
mapfile.write_results_tsv(d2, "/dev/shm/one.txt", threads=16); t3 = time.perf_counter()
print("one file: assign %.3f s, native write (8 threads) %.3f s, (16 threads) %.3f s" % (t1 - t0, t2 - t1, t3 - t2))
os.remove("/dev/shm/one.txt")
