#!/usr/bin/env python
"""End-to-end wall-clock of the drop-in on BASELINE configs[2] WRITTEN AS FILES.

    python tools/e2e_bench.py [--bins 288000 --elements 120091 --cohorts 37 --mut-rows 300000] [--workdir DIR] [--json OUT]

Inputs (synthetic, seeded; written once, untimed): C pretrained maps as real HDF5 files (region_params and
sequence_model_192 as pandas 'fixed' frames, idx, attributes -- io/h5lite.py), one element-data container (window context
counts + the element set, the layout scripts/DigPreprocess.py writes), one bed12 file, C annotated mutation files of
`mut-rows` rows each.  Timed: driver_model.cohort_batch.run_element_cohorts (maps + element data + mutation files ->
one result frame per cohort) and cohort_batch.write_results (C x <prefix>.results.txt, DigDriver.py's format), with the
seconds of every stage.  Then, for ONE cohort, the two command lines a user of the reference runs:
`scripts/DigPretrain.py elementModel` + `scripts/DigDriver.py elementDriver` (two fresh processes, wall-clock).
Beside it: the reference's own per-element loop rate measured in the build container (profiles/r04_reference_loop_rate.json;
/root/reference does not exist on the GPU box)."""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import pandas as pd


def write_inputs(args, work):
    import bench
    from digdriver_amd.io import mapfile
    from digdriver_amd.sequence_model import sequence_tools
    t = bench._workload_tables(args.bins, args.elements, args.cohorts, args.seed, 10_000, 3)
    N, E, C = args.bins, args.elements, args.cohorts
    rng = np.random.default_rng(args.seed + 1)
    keys = list(sequence_tools.mk_mutation_context(n_up=1, n_down=1, collapse=False).keys())     # (MUT_TYPE, CONTEXT) in model order
    paths = dict(pre=[], mut=[], ed=os.path.join(work, "element_data.map"), bed=os.path.join(work, "elements.bed"))
    idx = np.stack([t["bin_chrom"], t["bin_start"], t["bin_start"] + 10_000], 1).astype(np.int32)
    names = np.array(["chr%d:%d-%d" % tuple(r) for r in idx])
    for c in range(C):
        f = os.path.join(work, "cohort%02d.Pretrained.h5" % c)
        paths["pre"].append(f)
        if os.path.exists(f):
            continue
        rp = pd.DataFrame({"CHROM": t["bin_chrom"].astype(np.int64), "START": t["bin_start"], "END": t["bin_start"] + 10_000,
                           "Y_TRUE": t["bin_y"][:, c].astype(np.int64), "Y_PRED": t["bin_mu"][:, c], "STD": t["bin_std"][:, c],
                           "FLAG": t["bin_flag"][:, c].astype(bool)}, index=names)
        sm = pd.DataFrame({"MUT_TYPE": [k[0] for k in keys], "CONTEXT": [k[1] for k in keys], "FREQ": rng.dirichlet(np.ones(192)) * 1e-6 * 192})
        with mapfile.batch(f):
            mapfile.write_frame(f, "region_params", rp)
            mapfile.write_frame(f, "sequence_model_192", sm)
            mapfile.write_array(f, "idx", idx)
            mapfile.write_attrs(f, cohort_name="cohort%02d" % c, mappability_threshold=0.5)
    # elements: geometry from the bench generator, names, strands, L from the block-seeded draws
    n_blocks = (E + bench.ELEMENT_BLOCK - 1) // bench.ELEMENT_BLOCK
    _, ov_ptr, ov_idx, L, _, _, _ = bench._element_blocks(t, np.arange(n_blocks))
    elt_names = np.array(["ELT%06d" % i for i in range(E)])
    strand = np.where(t["strand_minus"] != 0, "-", "+")
    bp = t["blk_ptr"]
    if not os.path.exists(paths["ed"]):
        base = "window_10000/elts/"
        with mapfile.batch(paths["ed"]):
            mapfile.write_array(paths["ed"], "window_10000/full_window_si_index", idx)
            mapfile.write_array(paths["ed"], "window_10000/full_window_si_values", t["bin_ctx"])
            mapfile.write_array(paths["ed"], base + "names", elt_names)
            mapfile.write_array(paths["ed"], base + "chrom", t["elt_chrom"].astype(np.int32))
            mapfile.write_array(paths["ed"], base + "strand", strand)
            mapfile.write_array(paths["ed"], base + "blk_ptr", bp)
            mapfile.write_array(paths["ed"], base + "blk_start", t["blk_start"])
            mapfile.write_array(paths["ed"], base + "blk_end", t["blk_end"])
            mapfile.write_array(paths["ed"], base + "L", L[:, 0, :])
    if not os.path.exists(paths["bed"]):
        with open(paths["bed"], "w") as f:
            for e in range(E):
                s, en = t["blk_start"][bp[e]:bp[e + 1]], t["blk_end"][bp[e]:bp[e + 1]]
                f.write("%d\t%d\t%d\t%s\t0\t%s\t%d\t%d\t.\t%d\t%s,\t%s,\n" % (
                    t["elt_chrom"][e], s[0], en[-1], elt_names[e], strand[e], s[0], s[0], len(s), ",".join(map(str, en - s)),
                    ",".join(map(str, s - s[0]))))
    # mutation files: `mut-rows` rows per cohort, 70 % of them inside element blocks, 8 % indels, 3 % annotated twice
    nb = len(t["blk_start"])
    blk_chrom = np.repeat(t["elt_chrom"], np.diff(bp))
    chrom_bins = np.bincount(t["bin_chrom"], minlength=24)
    for c in range(C):
        f = os.path.join(work, "cohort%02d.annot.txt" % c)
        paths["mut"].append(f)
        if os.path.exists(f):
            continue
        r = np.random.default_rng([args.seed, 77, c])
        n = args.mut_rows
        inside = r.uniform(size=n) < 0.7
        b = r.integers(0, nb, n)
        pos_in = t["blk_start"][b] + (r.uniform(size=n) * (t["blk_end"][b] - t["blk_start"][b])).astype(np.int64)
        ch_out = r.integers(1, 23, n)
        pos_out = (r.uniform(size=n) * (chrom_bins[ch_out] * 10_000 - 10)).astype(np.int64)
        chrom = np.where(inside, blk_chrom[b], ch_out)
        pos = np.where(inside, pos_in, pos_out)
        indel = r.uniform(size=n) < 0.08
        ln = np.where(indel, r.integers(2, 12, n), 1)
        ref = np.where(indel, "ACGTACGTACGT", np.array(list("ACGT"))[r.integers(0, 4, n)])
        alt = np.where(indel, "A", np.array(list("ACGT"))[r.integers(0, 4, n)])
        df = pd.DataFrame({0: chrom, 1: pos, 2: pos + ln, 3: ref, 4: alt, 5: np.char.add("S", r.integers(0, 400, n).astype(str)),
                           6: ".", 7: np.where(indel, "INDEL", "Noncoding"), 8: np.where(indel, "DEL", "A>T"), 9: np.where(indel, ".", "CAG")})
        dup = df.iloc[r.integers(0, n, int(0.03 * n))].copy()
        dup[6] = "G2"
        pd.concat([df, dup]).to_csv(f, sep="\t", header=False, index=False)
    return paths


def run_e2e(bins=288_000, elements=120_091, cohorts=37, mut_rows=300_000, seed=3, workdir="/tmp/dig_e2e", read_workers=None,
            skip_cli=False, reps=5, keep=False):
    """Write the inputs (untimed), run the many-cohort pipeline `reps` times with stage timings, then the two per-cohort command
    lines once; returns the record tools/e2e_bench.py prints and bench.py embeds as `e2e`."""
    import shutil
    import types
    args = types.SimpleNamespace(bins=bins, elements=elements, cohorts=cohorts, mut_rows=mut_rows, seed=seed)
    os.makedirs(workdir, exist_ok=True)
    t0 = time.perf_counter()
    paths = write_inputs(args, workdir)
    t_inputs = time.perf_counter() - t0
    sizes = {"maps_MB": sum(os.path.getsize(f) for f in paths["pre"]) / 1e6, "mutation_files_MB": sum(os.path.getsize(f) for f in paths["mut"]) / 1e6,
             "element_data_MB": sum(os.path.getsize(os.path.join(dp, f)) for dp, _, fs in os.walk(paths["ed"]) for f in fs) / 1e6
             if os.path.isdir(paths["ed"]) else os.path.getsize(paths["ed"]) / 1e6}
    import torch
    from digdriver_amd import _lib
    from digdriver_amd.driver_model import cohort_batch
    _lib.require_device()
    torch.zeros(1, device="cuda:0")                      # device and library initialisation are not the pipeline's
    torch.cuda.synchronize()
    res = {"what": "BASELINE configs[2] as FILES -> %d results.txt: driver_model.cohort_batch.run_and_write_element_cohorts (the maps, the element "
                   "container and the mutation files read side by side; the result files written while the frames behind them are assembled), "
                   "wall-clock by stage (the device is drained at every stage boundary); inputs written beforehand, untimed" % cohorts,
           "config": {"bins": bins, "elements": elements, "cohorts": cohorts, "mutation_rows_per_cohort": int(mut_rows * 1.03)},
           "host_cores": os.cpu_count(), "input_files": sizes, "inputs_written_s": t_inputs, "runs": []}
    outdir = os.path.join(workdir, "results")
    out = []
    for rep in range(reps):                               # the second run has the files in the page cache and the kernels loaded
        stages = {}
        t0 = time.perf_counter()
        frames, out = cohort_batch.run_and_write_element_cohorts(paths["mut"], paths["pre"], paths["ed"], "elts", outdir,
                                                                 ["cohort%02d" % c for c in range(cohorts)], timings=stages,
                                                                 read_workers=read_workers)
        t2 = time.perf_counter()
        total = t2 - t0
        inside = stages.pop("inside_read_parse_upload", None)
        stages["results_txt_behind_the_last_frame"] = total - sum(stages.values())
        res["runs"].append({"total_s": total, "stages_s": {k: round(v, 4) for k, v in stages.items()},
                            "inside_read_parse_upload_s": inside,
                            "element_cohort_tests_per_s": elements * cohorts / total, "seconds_per_cohort": total / cohorts})
        del frames
    res["results_files"] = len(out)
    tot = sorted(r["total_s"] for r in res["runs"])
    res["total_s_median"] = tot[len(tot) // 2]
    # the overlapped route writes the bytes the plain route writes (run_element_cohorts, then write_results; untimed)
    serial_dir = os.path.join(workdir, "results_serial")
    frames = cohort_batch.run_element_cohorts(paths["mut"], paths["pre"], paths["ed"], "elts", read_workers=1)
    ser = cohort_batch.write_results(frames, serial_dir, ["cohort%02d" % c for c in range(cohorts)])
    del frames
    res["results_identical_to_the_serial_route"] = all(open(a_, "rb").read() == open(b_, "rb").read() for a_, b_ in zip(out, ser))
    if not skip_cli:
        # the per-cohort command lines of the reference's workflow, one cohort, fresh processes
        env = dict(os.environ, PYTHONPATH=ROOT)
        pre0 = os.path.join(workdir, "cli_cohort00.Pretrained.h5")
        shutil.copy(paths["pre"][0], pre0)
        t0 = time.perf_counter()
        subprocess.check_call([sys.executable, os.path.join(ROOT, "scripts", "DigPretrain.py"), "elementModel", pre0, paths["ed"], "elts"],
                              env=env, stdout=subprocess.DEVNULL)
        t1 = time.perf_counter()
        subprocess.check_call([sys.executable, os.path.join(ROOT, "scripts", "DigDriver.py"), "elementDriver", paths["mut"][0], pre0, "elts",
                               "--f-bed", paths["bed"], "--scale-factor-manual", "1.0", "--scale-factor-indel-manual", "0.1",
                               "--outdir", os.path.join(workdir, "cli_out"), "--outpfx", "cohort00"],
                              env=env, stdout=subprocess.DEVNULL)
        t2 = time.perf_counter()
        if os.environ.get("DIG_E2E_CLI_PROFILE"):                 # developer: where the elementDriver process spends its time
            import pstats
            prof = os.path.join(workdir, "cli.prof")
            subprocess.check_call([sys.executable, "-m", "cProfile", "-o", prof, os.path.join(ROOT, "scripts", "DigDriver.py"), "elementDriver",
                                   paths["mut"][0], pre0, "elts", "--f-bed", paths["bed"], "--scale-factor-manual", "1.0",
                                   "--scale-factor-indel-manual", "0.1", "--outdir", os.path.join(workdir, "cli_out"), "--outpfx", "cohort00"],
                                  env=env, stdout=subprocess.DEVNULL)
            pstats.Stats(prof, stream=sys.stderr).sort_stats("cumulative").print_stats(45)
        res["cli_one_cohort"] = {"what": "scripts/DigPretrain.py elementModel + scripts/DigDriver.py elementDriver, one cohort, two fresh processes "
                                         "(interpreter start, imports and device initialisation included)",
                                 "DigPretrain_elementModel_s": t1 - t0, "DigDriver_elementDriver_s": t2 - t1, "elements_per_s": elements / (t2 - t0)}
    ref = os.path.join(ROOT, "profiles", "r04_reference_loop_rate.json")
    if os.path.exists(ref):
        res["reference_in_container"] = json.load(open(ref))
    if not keep:
        shutil.rmtree(workdir, ignore_errors=True)
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bins", type=int, default=288_000)
    ap.add_argument("--elements", type=int, default=120_091)
    ap.add_argument("--cohorts", type=int, default=37)
    ap.add_argument("--mut-rows", type=int, default=300_000)
    ap.add_argument("--seed", type=int, default=3)
    ap.add_argument("--workdir", default="/tmp/dig_e2e")
    ap.add_argument("--json", default=None)
    ap.add_argument("--read-workers", type=int, default=None)
    ap.add_argument("--skip-cli", action="store_true")
    ap.add_argument("--keep", action="store_true")
    args = ap.parse_args()
    res = run_e2e(args.bins, args.elements, args.cohorts, args.mut_rows, args.seed, args.workdir, args.read_workers, args.skip_cli,
                  keep=args.keep)
    txt = json.dumps(res, indent=1)
    print(txt)
    if args.json:
        with open(args.json, "w") as f:
            f.write(txt + "\n")


if __name__ == "__main__":
    main()
