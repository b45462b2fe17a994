#!/usr/bin/env python
"""Aggregate rocprofv3 --pmc counter_collection CSVs per kernel (mean over dispatches)."""
import collections
import csv
import glob
import sys


def main(dirs):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in dirs:
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                agg[r["Kernel_Name"].split("(")[0][-48:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        if "dig::" not in k and "dig" not in k:
            continue
        d = {c: sum(x) / len(x) for c, x in v.items()}
        print(k)
        wc = d.get("SQ_WAVE_CYCLES")
        for c in sorted(d):
            extra = ""
            if wc and c.startswith(("SQ_WAIT", "SQ_ACTIVE", "SQ_INST_CYCLES")):
                extra = "  (%.1f%% of wave cycles)" % (100 * d[c] / wc)
            print("    %-28s %.4g%s" % (c, d[c], extra))
        if "SQ_THREAD_CYCLES_VALU" in d and "SQ_ACTIVE_INST_VALU" in d and d["SQ_ACTIVE_INST_VALU"]:
            print("    lanes active per VALU cycle: %.1f / 64" % (d["SQ_THREAD_CYCLES_VALU"] / d["SQ_ACTIVE_INST_VALU"]))


if __name__ == "__main__":
    main(sys.argv[1:] or ["."])
