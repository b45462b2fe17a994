#!/usr/bin/env python
"""Turn the rocprofv3 outputs of one round into the committed summaries under profiles/ (all of `python3 bench.py ...`;
--output-format csv):

    gpurun_out/rNNa   --kernel-trace --stats          -> profiles/rNN_kernel_stats.csv   (kernel_stats.csv as is)
    gpurun_out/rNNf   --pmc FETCH_SIZE                 -> profiles/rNN_traffic.json       per-kernel HBM bytes per launch,
    gpurun_out/rNNw   --pmc WRITE_SIZE                    corrected as MI355X_MICROARCH.md (HBM / rocprofv3) prescribes:
                                                          counters are in KiB; FETCH_SIZE reads half of the streamed
                                                          bytes on gfx950 -> x2 (calibrated here on the sufficient-
                                                          statistics kernel, whose read volume N*C*9 B is known);
                                                          WRITE_SIZE is exact
    gpurun_out/rNNv/* --pmc <SQ_* sets, <= 4 per pass> -> profiles/rNN_valu.json          VALU instruction counts, active
                                                          cycles, waits per kernel (mean per launch) + derived figures
"""
import collections
import csv
import glob
import json
import shutil
import sys

N_PAIRS = 120091 * 37


def counters(dirname):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(dirname + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(x) / len(x) for c, x in v.items()} for k, v in acc.items()}


def main(tag, known_suffstats_bytes):
    base = "gpurun_out/" + tag
    stats = glob.glob(base + "a/**/*kernel_stats.csv", recursive=True)[0]
    shutil.copy(stats, "profiles/%s_kernel_stats.csv" % tag)
    # the kernels of bench.py's aux_rooflines legs (gather, CNN forward GEMMs, per-base tiles, context counting), same trace
    # (the library's own kernels of those legs in full -- the sort and Benjamini-Hochberg kernels of the per-base route included --;
    #  of the GEMM kernels the thirty with the most time: the GEMM tuner's trials are a thousand more names)
    with open(stats) as f, open("profiles/%s_aux_kernel_stats.csv" % tag, "w") as g:
        gemm = []
        for i, line in enumerate(f):
            if i == 0 or any(k in line for k in ("gather_", "base_tile_probs", "tiled_nb", "context_count", "tile_mut", "dig::sort_", "dig::bhr_",
                                                 "dig::bh_", "batch_norm", "multi_tensor_apply")):
                g.write(line)
            elif "Cijk_" in line:
                gemm.append(line)
        gemm.sort(key=lambda l: -float(l.rsplit('",', 1)[1].split(",")[1]))
        g.writelines(gemm[:30])
    fetch = {k: v.get("FETCH_SIZE", 0.0) for k, v in counters(base + "f").items()}
    write = {k: v.get("WRITE_SIZE", 0.0) for k, v in counters(base + "w").items()}
    cal = [k for k in fetch if "suffstats_chunk_stage1" in k or "suffstats_stage1" in k][0]
    factor = known_suffstats_bytes / (fetch[cal] * 1024.0)
    out = {"unit": "bytes per launch", "fetch_correction": 2.0, "calibration": {
        "kernel": cal, "known_read_bytes": known_suffstats_bytes, "raw_FETCH_SIZE_KiB": fetch[cal], "measured_factor": factor},
        "kernels": {}}
    for k in sorted(set(fetch) | set(write)):
        if "dig" not in k:
            continue
        rd, wr = fetch.get(k, 0.0) * 1024.0 * 2.0, write.get(k, 0.0) * 1024.0
        out["kernels"][k] = {"read_bytes": rd, "write_bytes": wr, "hbm_bytes": rd + wr}
    json.dump(out, open("profiles/%s_traffic.json" % tag, "w"), indent=1)
    print(json.dumps(out["calibration"]))
    for k, v in out["kernels"].items():
        print("%-60s read %8.1f MB  write %8.1f MB" % (k[-60:], v["read_bytes"] / 1e6, v["write_bytes"] / 1e6))
    # ---- VALU counters ----
    valu = {"source": "rocprofv3 --pmc (one pass per set of <= 4 counters) of `python3 bench.py --cpu-sample 0 --steps 40 --warmup 5`; "
                      "mean per launch",
            "units": "SQ_WAVE_CYCLES, SQ_ACTIVE_INST_*, SQ_WAIT_*, SQ_BUSY_CYCLES are in quad-cycles (4 clocks), summed over waves; "
                     "SQ_INSTS_* are wave-level instruction counts; 1024 SIMDs",
            "kernels": {}}
    for k, d in counters(base + "v").items():
        if "dig::" not in k:
            continue
        if "SQ_INSTS_VALU" in d and d.get("SQ_WAVES"):
            d["derived_valu_insts_per_wave"] = d["SQ_INSTS_VALU"] / d["SQ_WAVES"]
        if d.get("SQ_ACTIVE_INST_VALU") and "SQ_THREAD_CYCLES_VALU" in d:
            d["derived_lanes_active_per_valu_inst"] = d["SQ_THREAD_CYCLES_VALU"] / d["SQ_ACTIVE_INST_VALU"]
        if d.get("SQ_WAVE_CYCLES"):
            d["derived_valu_active_frac_of_wave_cycles"] = d.get("SQ_ACTIVE_INST_VALU", 0.0) / d["SQ_WAVE_CYCLES"]
            d["derived_wait_any_frac_of_wave_cycles"] = d.get("SQ_WAIT_ANY", 0.0) / d["SQ_WAVE_CYCLES"]
        if d.get("SQ_ACTIVE_INST_VALU") and d.get("SQ_INSTS_VALU"):
            d["derived_quadcycles_per_valu_inst"] = d["SQ_ACTIVE_INST_VALU"] / d["SQ_INSTS_VALU"]
            d["derived_valu_active_quadcycles_per_simd"] = d["SQ_ACTIVE_INST_VALU"] / 1024.0
        if "element_stats_stream" in k and "SQ_INSTS_VALU" in d:
            d["derived_valu_insts_per_64_pair_tile"] = d["SQ_INSTS_VALU"] / ((N_PAIRS + 63) // 64)
        valu["kernels"][k] = d
    json.dump(valu, open("profiles/%s_valu.json" % tag, "w"), indent=1)
    for k, d in valu["kernels"].items():
        if "element_stats" in k:
            print(k, {c: round(x, 3) for c, x in d.items() if c.startswith("derived")})


if __name__ == "__main__":
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 288000 * 37 * 8.0)      # (round 3: the chunk sums read the pre-masked table, 8 B per entry; 9 before)
