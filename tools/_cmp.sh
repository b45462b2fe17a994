cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/cmp -- python3 bench.py --cpu-sample 0 --e2e 0 --aux 0 --steps 300 --warmup 20 > gpurun_out/cmp.json 2> gpurun_out/cmp.err
python3 - <<PY
import json,glob,csv
d=json.loads(open("gpurun_out/cmp.json").read().strip().splitlines()[-1]); r=d["roofline"]
print("bench: stats", r["avg_launch_ms"], "n", r["launches_timed"], "dot", [ (x or {}).get("avg_launch_ms") for x in d["roofline_other_stages"]])
f=glob.glob("gpurun_out/cmp/**/*kernel_stats.csv", recursive=True)[0]
for row in csv.DictReader(open(f)):
    if "element_stats_stream_fused" in row["Name"] or "acc_dot_ctx" in row["Name"]: print(row["Name"][:50], row["Calls"], float(row["AverageNs"])/1e3, row["MinNs"], row["MaxNs"])
PY
rocm-smi --showserial | grep -i "serial n"
