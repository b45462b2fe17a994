#!/usr/bin/env python
"""Developer tool: statistics-kernel durations of a rocprofv3 kernel trace of bench.py, split into the settle phase (sequential
evaluation on seq_plan) and the timed loop (the last STEPS + WARMUP launches).   python tools/trace_phases.py trace.csv 320"""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "element_stats_stream_fused" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
n = int(sys.argv[2])
def st(x): x = sorted(x); return "n=%d median %.1f mean %.1f min %.1f p90 %.1f" % (len(x), x[len(x) // 2], sum(x) / len(x), x[0], x[int(len(x) * 0.9)])
print("settle phase:", st(d[:-n])); print("loop        :", st(d[-n:]))
# gap between the end of the preceding kernel on the same stream and the start of the statistics kernel, loop only
allr = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
prev_end = {}
gaps = []
for r in allr:
    q = r.get("Queue_Id") or r.get("Stream_Id")
    if "element_stats_stream_fused" in r["Kernel_Name"] and q in prev_end:
        gaps.append((int(r["Start_Timestamp"]) - prev_end[q]) / 1e3)
    prev_end[q] = int(r["End_Timestamp"])
print("gap before the statistics kernel (same queue), last %d:" % n, st(gaps[-n:]))
# medians over time (all launches in start order, ten buckets per phase)
def buckets(x, k=10):
    m = max(1, len(x) // k)
    return [round(sorted(x[i:i + m])[len(x[i:i + m]) // 2], 1) for i in range(0, len(x) - m + 1, m)]
t0 = int(rows[0]["Start_Timestamp"])
print("settle phase medians over time:", buckets(d[:-n]), " (%.0f ms long)" % ((int(rows[-n - 1]["End_Timestamp"]) - t0) / 1e6))
print("loop medians over time        :", buckets(d[-n:]), " (%.0f ms long)" % ((int(rows[-1]["End_Timestamp"]) - int(rows[-n]["Start_Timestamp"])) / 1e6))
# the loop's timeline on the main stream: gap in front of every accumulation kernel (= between two steps) and step period
acc = [r for r in allr if "acc_dot" in r["Kernel_Name"]][-n:]
ends = {}
per, gap2 = [], []
prev_start = None
for r in allr:
    q = r.get("Queue_Id") or r.get("Stream_Id")
    if "acc_dot" in r["Kernel_Name"]:
        if q in ends: gap2.append((int(r["Start_Timestamp"]) - ends[q]) / 1e3)
        if prev_start is not None: per.append((int(r["Start_Timestamp"]) - prev_start) / 1e3)
        prev_start = int(r["Start_Timestamp"])
    ends[q] = int(r["End_Timestamp"])
print("gap before the accumulation kernel, last %d:" % n, st(gap2[-n:]))
print("step period (start to start of the accumulation kernel), last %d:" % n, st(per[-n:]))
dd = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in acc]
print("accumulation kernel, last %d:" % n, st(dd))
