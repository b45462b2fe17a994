#!/usr/bin/env python
"""Mid-scale run of the many-cohort element driver (driver_model/cohort_batch.run_element_cohorts) from files: a synthetic
FASTA genome, per-cohort pretrained maps and mutation TSVs, an element bed12.  Developer tool: exercises the host side
(readers, context counting from the packed genome, tabulation, result frames) at ~100x the size of the unit test."""
import argparse
import os
import sys
import tempfile
import time

import numpy as np
import pandas as pd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--chroms", type=int, default=4)
    ap.add_argument("--bins-per-chrom", type=int, default=500)
    ap.add_argument("--elements", type=int, default=5000)
    ap.add_argument("--cohorts", type=int, default=37)
    ap.add_argument("--muts-per-cohort", type=int, default=20000)
    a = ap.parse_args()
    from digdriver_amd.driver_model import cohort_batch
    from digdriver_amd.io import mapfile
    from digdriver_amd.sequence_model import sequence_tools
    from oracle import dig_oracle as O          # (developer tool: only the 192 model row labels are taken from it)
    rng = np.random.default_rng(31)
    tmp = tempfile.mkdtemp(prefix="cohort_scale_")
    window = 10000
    t0 = time.time()
    fa = os.path.join(tmp, "genome.fa")
    with open(fa, "w") as f:
        for c in range(1, a.chroms + 1):
            s = rng.choice(np.frombuffer(b"ACGTN", np.uint8), a.bins_per_chrom * window, p=[.245, .245, .245, .245, .02]).tobytes().decode()
            f.write(">chr%d\n" % c)
            f.write("\n".join(s[i:i + 60] for i in range(0, len(s), 60)) + "\n")
    rows = [(c, b * window, (b + 1) * window) for c in range(1, a.chroms + 1) for b in range(a.bins_per_chrom)]
    grid = pd.DataFrame(rows, columns=["CHROM", "START", "END"])
    lines, elts = [], []
    for i in range(a.elements):
        c = int(rng.integers(1, a.chroms + 1))
        st = int(rng.integers(0, a.bins_per_chrom * window - 6000))
        nb = int(rng.integers(1, 4))
        blocks, off = [], 0
        for _ in range(nb):
            z = int(rng.integers(100, 1500))
            blocks.append((off, z))
            off += z + int(rng.integers(10, 400))
        end = st + blocks[-1][0] + blocks[-1][1]
        lines.append("%d\t%d\t%d\tE%05d\t0\t%s\t%d\t%d\t.\t%d\t%s,\t%s,\n" % (
            c, st, end, i, "+-"[i & 1], st, st, nb, ",".join(str(z) for _, z in blocks), ",".join(str(o) for o, _ in blocks)))
        elts.append((c, [(st + o, st + o + z) for o, z in blocks]))
    bed = os.path.join(tmp, "elts.bed")
    open(bed, "w").write("".join(lines))
    seq_rows = O.model_rows192()
    pres, muts = [], []
    for k in range(a.cohorts):
        rp = grid.copy()
        n = len(rp)
        rp["Y_PRED"] = rng.gamma(9.0, 3.0, n)
        rp["Y_TRUE"] = rng.poisson(rp.Y_PRED.values)
        rp["STD"] = rng.gamma(4.0, 1.0, n)
        rp["MAPP"], rp["QUANT"], rp["FLAG"] = 1.0, 0.5, rng.uniform(size=n) < 0.1
        rp.index = ["chr{}:{}-{}".format(c, s, e) for c, s, e in zip(rp.CHROM, rp.START, rp.END)]
        sm = pd.DataFrame({"MUT_TYPE": [m for m, _ in seq_rows], "CONTEXT": [c for _, c in seq_rows],
                           "COUNT": rng.integers(1, 1000, 192), "FREQ": rng.dirichlet(np.ones(192)) * 1e-3})
        pre = os.path.join(tmp, "cohort%d.map" % k)
        mapfile.write_frame(pre, "region_params", rp)
        mapfile.write_frame(pre, "sequence_model_192", sm)
        mapfile.write_array(pre, "idx", rp[["CHROM", "START", "END"]].values.astype(np.int32))
        mapfile.write_attrs(pre, cohort_name="c%d" % k, mappability_threshold=0.5)
        m = a.muts_per_cohort
        pick = rng.integers(0, a.elements, m // 2)
        in_c = np.array([elts[i][0] for i in pick])
        in_p = np.array([rng.integers(*elts[i][1][0]) for i in pick])
        bg_c = rng.integers(1, a.chroms + 1, m - m // 2)
        bg_p = rng.integers(0, a.bins_per_chrom * window - 2, m - m // 2)
        ch, pos = np.concatenate([in_c, bg_c]), np.concatenate([in_p, bg_p])
        indel = rng.uniform(size=m) < 0.08
        df = pd.DataFrame({0: ch, 1: pos, 2: pos + np.where(indel, 3, 1), 3: np.where(indel, "AGG", "A"), 4: np.where(indel, "A", "T"),
                           5: ["S%d" % s for s in rng.integers(0, 200, m)], 6: ".", 7: np.where(indel, "INDEL", "Noncoding"),
                           8: np.where(indel, "DEL", "A>T"), 9: np.where(indel, ".", "CAG")})
        f = os.path.join(tmp, "muts%d.tsv" % k)
        df.to_csv(f, sep="\t", header=False, index=False)
        pres.append(pre)
        muts.append(f)
    t1 = time.time()
    gc, ed = os.path.join(tmp, "gc.map"), os.path.join(tmp, "ed.map")
    win = sequence_tools.count_contexts_in_bed(fa, grid, n_up=1, n_down=1)
    mapfile.write_frame(gc, "all_window_genome_counts", win)
    mapfile.write_array(gc, "idx", grid.values.astype(np.int32))
    sequence_tools.initialize_nonc_data(ed, gc, window)
    L = sequence_tools.precount_region_contexts_parallel(bed, fa, 1, window, True)
    sequence_tools.preprocess_nonc(bed, ed, pres[0], L, "elts", window)
    t2 = time.time()
    frames = cohort_batch.run_element_cohorts(muts, pres, ed, "elts", scale_factors=None)
    t3 = time.time()
    assert len(frames) == a.cohorts
    f0 = frames[0]
    assert len(f0) == a.elements and np.isfinite(f0.PVAL_SNV_BURDEN.values).all() and f0.OBS_SNV.sum() > 0
    # the per-cohort route of the reference CLI for one cohort: DigPretrain elementModel -> DigDriver elementDriver
    from digdriver_amd.driver_model import transfer_tools
    from digdriver_amd.sequence_model import genic_driver_tools
    t4 = time.time()
    frame = genic_driver_tools.nonc_model_parallel(pres[0], ed, "elts", 1)
    mapfile.write_frame(pres[0], "elts", frame)
    t5 = time.time()
    one = transfer_tools.run_element_region_model(muts[0], bed, pres[0], "elts", scale_by_expectation=False, scale_type="genome",
                                                  fused=True)
    t6 = time.time()
    both = one.index.intersection(f0.index)
    assert len(both) == a.elements and np.array_equal(one.loc[both].OBS_SNV.values.astype(int), f0.loc[both].OBS_SNV.values.astype(int))
    # quickDriver (driver_model/onthefly_tools.DIG_onthefly): everything from the FASTA, no element-data container
    from digdriver_amd.driver_model import onthefly_tools
    t7 = time.time()
    quick = onthefly_tools.DIG_onthefly(pres[0], muts[0], fa, f_elts_bed=bed, scale_factor=1.0, scale_factor_indel=1.0,
                                        scale_by_expectation=False)
    t8 = time.time()
    assert len(quick) == a.elements and np.isfinite(quick.PVAL_SNV_BURDEN.values.astype(float)).all()
    print("quickDriver, cohort 0: %.1f s" % (t8 - t7))
    print("per-cohort route, cohort 0: nonc_model_parallel %.1f s, run_element_region_model %.1f s" % (t5 - t4, t6 - t5))
    print("inputs written %.1f s | context counting + element data %.1f s | run_element_cohorts (%d cohorts x %d elements, "
          "%d mutations) %.1f s" % (t1 - t0, t2 - t1, a.cohorts, a.elements, a.cohorts * a.muts_per_cohort, t3 - t2))


if __name__ == "__main__":
    main()
