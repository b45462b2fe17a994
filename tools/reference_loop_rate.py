#!/usr/bin/env python
"""Rate of the REFERENCE's own per-element loop and statistics block, measured in the build container (it needs
/root/reference; nothing of this runs on the GPU box).  Writes profiles/r04_reference_loop_rate.json, which
tools/e2e_bench.py prints beside the drop-in's wall-clock.

    PYTHONDONTWRITEBYTECODE=1 python tools/reference_loop_rate.py [--elements 3000]

What is timed: genic_driver_tools.nonc_model (DIGDriver/sequence_model/genic_driver_tools.py:300-431 -- what
`DigPretrain.py elementModel` runs per worker process) on N synthetic elements against a whole-genome region_params frame
(288 000 bins), with the HDF5 files replaced by in-memory dictionaries (tests/golden/make_golden.py's stand-ins: the three
h5 reads per element the real loop does are FREE here, so this is an upper bound of the reference's rate); and
transfer_tools' statistics block (element_expected_muts_nb + the three burden tests + Fisher, transfer_tools.py:272-302,343-344,
473-482,594-615,731-747,1086-1087) on a 120 091-row frame."""
import argparse
import json
import os
import sys
import time

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

import numpy as np
import pandas as pd


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--elements", type=int, default=3000)
    args = ap.parse_args()
    import make_golden as G                      # installs the stand-ins and imports the reference from /root/reference
    rng = np.random.default_rng(5)
    window = 10000
    chroms = list(range(1, 23))
    idx, df_reg = G.synth_region_params(rng, chroms, 288_000 // 22, window)
    n = len(idx)
    ctx_p = rng.dirichlet(np.ones(64))
    bin_ctx = rng.multinomial(window, ctx_p, size=n).astype(np.int64)
    subst_idx = sorted(G.ref_seq.mk_trans_idx(n_up=1, n_down=1, collapse=False))
    keys = list(G.ref_seq.mk_mutation_context(n_up=1, n_down=1, collapse=False).keys())
    df_seq = pd.DataFrame({"MUT_TYPE": [k[0] for k in keys], "CONTEXT": [k[1] for k in keys], "COUNT": rng.integers(0, 1000, 192),
                           "FREQ": rng.dirichlet(np.ones(192)) * 1e-6 * 192})
    f_pre, f_dat = "mem://rate_pretrained.h5", "mem://rate_element_data.h5"
    G._HDF_FRAMES[(f_pre, "region_params")] = df_reg
    G._HDF_FRAMES[(f_pre, "sequence_model_192")] = df_seq
    idx_dict = {tuple(int(v) for v in r): i for i, r in enumerate(idx)}
    per_chrom = 288_000 // 22
    save_key = "elts"
    tree = {"window_%d" % window: {save_key: {}, "full_window_si_values": bin_ctx, "full_window_si_index": idx}}
    names = []
    for e in range(args.elements):
        chrom = int(rng.choice(chroms))
        nb = int(rng.integers(1, 4))
        pos = int(rng.integers(0, (per_chrom - 3) * window))
        starts, ends = [], []
        for _ in range(nb):
            s = pos + int(rng.integers(0, 4000))
            ln = int(rng.integers(200, 3000))
            starts.append(s)
            ends.append(s + ln)
            pos = s + ln
        overlaps = G.ref_gdt.get_ideal_overlaps(chrom, np.vstack((starts, ends)), window)
        region_counts = np.array([np.repeat(bin_ctx[idx_dict[r], :], 3) for r in overlaps]).sum(axis=0)
        L = np.repeat(rng.multinomial(int(sum(b - a for a, b in zip(starts, ends))), ctx_p), 3).astype(np.float64)
        name = "elt_%06d" % e
        tree["window_%d" % window][save_key][name] = {"L_counts": L, "region_counts": region_counts, "__attrs__": {"overlaps": np.array(overlaps)}}
        names.append(name)
    G._H5_FILES[f_dat] = tree
    t0 = time.perf_counter()
    df_out = G.ref_gdt.nonc_model(names, f_pre, f_dat, save_key, False)
    t_loop = time.perf_counter() - t0
    assert len(df_out) == args.elements
    # the statistics block on a frame of the bench's size
    E = 120_091
    df = pd.DataFrame({"MU": rng.gamma(9, 3, E), "SIGMA": rng.gamma(4, 1, E), "Pi_SUM": rng.uniform(1e-5, 1e-2, E),
                       "Pi_INDEL": rng.uniform(1e-5, 1e-2, E), "OBS_SNV": rng.poisson(3, E), "OBS_SAMPLES": rng.poisson(2, E),
                       "OBS_INDEL": rng.poisson(0.3, E)})
    df["ALPHA"], df["THETA"] = G.ref_nb.normal_params_to_gamma(df.MU, df.SIGMA)
    df["THETA"] = df.THETA * 1.3
    df["MU_INDEL"], df["SIGMA_INDEL"], df["ALPHA_INDEL"], df["THETA_INDEL"] = df.MU, df.SIGMA, df.ALPHA, df.THETA * 0.1
    t0 = time.perf_counter()
    df = G.ref_tt.element_expected_muts_nb(df)
    df = G.ref_tt.element_pvalue_burden_nb(df)
    df = G.ref_tt.element_pvalue_burden_nb_by_sample(df)
    df = G.ref_tt.element_pvalue_indel(df, 0.1)
    df["PVAL_MUT_BURDEN"] = [G.scipy.stats.combine_pvalues([a, b], method="fisher")[1] for a, b in zip(df.PVAL_SNV_BURDEN.values[:2000], df.PVAL_INDEL_BURDEN.values[:2000])] + [np.nan] * (E - 2000)
    t_stats = time.perf_counter() - t0
    out = {"what": "the reference's own code, timed in the build container (8 cores, 1 process); HDF5 reads replaced by in-memory "
                   "dictionaries (an upper bound of its real rate)",
           "nonc_model_elements": args.elements, "nonc_model_s": t_loop, "nonc_model_elements_per_s_per_process": args.elements / t_loop,
           "seconds_per_cohort_of_120091_elements_one_process": 120_091 / (args.elements / t_loop),
           "statistics_block_rows": E, "statistics_block_s": t_stats, "reference_default_processes": "min(max(1, ncpu - 2), 20) (auxilaries/utils.py:3-8)",
           "numpy": np.__version__, "pandas": pd.__version__, "scipy": G.scipy.__version__}
    path = os.path.join(ROOT, "profiles", "r04_reference_loop_rate.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
