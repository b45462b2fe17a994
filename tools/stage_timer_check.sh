#!/bin/bash
# bench.py's stage timers beside rocprofv3's kernel trace of the same run (run through gpurun from the repo root;
# profiles/r04_stage_timer_check.txt holds five such runs).
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/cmp -- python3 bench.py --cpu-sample 0 --e2e 0 --aux 0 --steps 300 --warmup 20 > gpurun_out/cmp.json 2> gpurun_out/cmp.err
python3 - <<PY
import json, glob, csv
import numpy as np
d = json.loads(open("gpurun_out/cmp.json").read().strip().splitlines()[-1]); r = d["roofline"]
print({k: r.get(k) for k in ("avg_launch_ms", "frac", "timer_of_an_empty_kernel_us", "launches_timed")}, "dot", d["roofline_other_stages"][1]["avg_launch_ms"])
f = glob.glob("gpurun_out/cmp/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
for key in ("element_stats_stream_fused", "acc_dot_ctx"):
    st = sorted((int(x["Start_Timestamp"]), int(x["End_Timestamp"])) for x in rows if key in x["Kernel_Name"])
    dd = np.array([e - s for s, e in st]) / 1e3
    print(key, "rocprofv3: all %.1f us, the last 330 launches (the loop) %.1f us" % (dd.mean(), dd[-330:].mean()),
          "by position in a group of eight steps:", [round(float(dd[-330:][m::8].mean()), 1) for m in range(8)])
PY
rocm-smi --showserial | grep -i "serial n"
