#!/usr/bin/env python
"""Turn the rocprofv3 outputs of one round (gpurun_out/rNNa = --kernel-trace --stats, rNNf = --pmc FETCH_SIZE,
rNNw = --pmc WRITE_SIZE, all of `python3 bench.py ...`) into the committed summaries under profiles/:

    profiles/rNN_kernel_stats.csv   rocprofv3 kernel_stats.csv as is (dig:: kernels + everything else)
    profiles/rNN_traffic.json       per-kernel HBM bytes per launch from the PMC passes, corrected as
                                    MI355X_MICROARCH.md (HBM / rocprofv3) prescribes: counters are in KiB;
                                    FETCH_SIZE reads 1/2 of the streamed bytes on gfx950 -> x2 (calibrated here on
                                    dig::suffstats_stage1, whose read volume N*C*9 B is known); WRITE_SIZE is exact.
"""
import collections
import csv
import glob
import json
import shutil
import sys


def per_kernel(dirname, counter):
    acc = collections.defaultdict(list)
    for f in glob.glob(dirname + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


def main(tag, known_suffstats_bytes):
    base = "gpurun_out/" + tag
    stats = glob.glob(base + "a/**/*kernel_stats.csv", recursive=True)[0]
    shutil.copy(stats, "profiles/%s_kernel_stats.csv" % tag)
    fetch, write = per_kernel(base + "f", "FETCH_SIZE"), per_kernel(base + "w", "WRITE_SIZE")
    cal = [k for k in fetch if "suffstats_stage1" in k][0]
    factor = known_suffstats_bytes / (fetch[cal] * 1024.0)
    out = {"unit": "bytes per launch", "fetch_correction": 2.0, "calibration": {
        "kernel": cal, "known_read_bytes": known_suffstats_bytes, "raw_FETCH_SIZE_KiB": fetch[cal],
        "measured_factor": factor}, "kernels": {}}
    for k in sorted(set(fetch) | set(write)):
        if "dig" not in k:
            continue
        rd = fetch.get(k, 0.0) * 1024.0 * 2.0
        wr = write.get(k, 0.0) * 1024.0
        out["kernels"][k] = {"read_bytes": rd, "write_bytes": wr, "hbm_bytes": rd + wr}
    json.dump(out, open("profiles/%s_traffic.json" % tag, "w"), indent=1)
    print(json.dumps(out["calibration"]))
    for k, v in out["kernels"].items():
        print("%-50s read %8.1f MB  write %8.1f MB" % (k[-50:], v["read_bytes"] / 1e6, v["write_bytes"] / 1e6))


if __name__ == "__main__":
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 288000 * 37 * 9.0)
