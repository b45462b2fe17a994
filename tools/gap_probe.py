#!/usr/bin/env python
"""Gaps between the kernels of the step in a rocprofv3 --kernel-trace csv (developer tool).

    python tools/gap_probe.py <kernel_trace.csv>

Prints, for the two kernels of the step (dot and statistics), their durations and the idle time between the end of one and the
start of the next (statistics -> dot of the next step, dot -> statistics), medians over the trace."""
import csv
import statistics
import sys

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        name = r.get("Kernel_Name") or r.get("kernel_name")
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name))
rows.sort()
step = [(a, b, "dot" if "acc_dot_ctx" in n else "stats") for a, b, n in rows if "acc_dot_ctx" in n or "element_stats_stream_fused" in n]
others = [(a, b, n) for a, b, n in rows if not ("acc_dot_ctx" in n or "element_stats_stream_fused" in n)]
dur = {"dot": [], "stats": []}
gap = {"dot->stats": [], "stats->dot": []}
for i, (a, b, k) in enumerate(step):
    dur[k].append(b - a)
    if i + 1 < len(step):
        a2, _, k2 = step[i + 1]
        if k != k2:
            gap[k + "->" + k2].append(a2 - b)
skip = len(step) // 5
for k, v in dur.items():
    v = v[skip:]
    print("%-12s n=%d median %.2f us  mean %.2f" % (k, len(v), statistics.median(v) / 1e3, statistics.mean(v) / 1e3))
for k, v in gap.items():
    v = v[skip:]
    print("gap %-10s n=%d median %.2f us  mean %.2f  p10 %.2f p90 %.2f" % (k, len(v), statistics.median(v) / 1e3, statistics.mean(v) / 1e3,
                                                                     sorted(v)[len(v) // 10] / 1e3, sorted(v)[len(v) * 9 // 10] / 1e3))
period = [step[i + 2][0] - step[i][0] for i in range(skip, len(step) - 2) if step[i][2] == "dot"]
print("period (dot start to next dot start): median %.2f us" % (statistics.median(period) / 1e3))
names = {}
for a, b, n in others:
    names.setdefault(n[:60], []).append(b - a)
for n, v in sorted(names.items(), key=lambda kv: -sum(kv[1]))[:8]:
    print("other %-60s n=%d mean %.2f us" % (n, len(v), statistics.mean(v) / 1e3))
