#!/bin/bash
# Collect the round's rocprofv3 evidence on the GPU box (run through gpurun from the repo root):
#   tools/collect_profiles.sh r03        -> gpurun_out/r03{a,f,w,v/*} ; then `python tools/make_profile_summary.py r03` here.
# Kernel trace and counters in SEPARATE runs (counters with --kernel-trace only, as the pool requires).
set -u
TAG=${1:-r06}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out
B="python3 bench.py --cpu-sample 0 --e2e 0"
rocm-smi --showserial --showmemvendor 2>/dev/null | grep -i "Serial N\|vendor" > $OUT/${TAG}_gpu.txt      # which GPU of the pool
export DIG_NN_TUNE_FILE=/tmp/dig_tune_$TAG/t.csv      # the GEMM tuner's results of the unprofiled run below serve the profiled one (its
                                                      # trials were 417 000 extra launches in round 6's trace)
python3 bench.py --steps 20 --warmup 5 > $OUT/${TAG}_bench_steps20.json 2> $OUT/${TAG}_b20.err      # the driver's command: e2e and aux legs included
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}a -- $B --steps 300 --warmup 20 > $OUT/${TAG}_bench_under_rocprofv3.json 2> $OUT/${TAG}a.err
python3 bench.py --cpu-sample 0 --aux 0 --e2e 0 --steps 1000 --warmup 50 > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_b1000.err
timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}f -- $B --aux 0 --steps 40 --warmup 5 > /dev/null 2> $OUT/${TAG}f.err
timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}w -- $B --aux 0 --steps 40 --warmup 5 > /dev/null 2> $OUT/${TAG}w.err
i=0
for SET in "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY" \
           "GRBM_GUI_ACTIVE SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64" \
           "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_LDS_BANK_CONFLICT" "SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  timeout 600 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $OUT/${TAG}v/$i -- $B --aux 0 --steps 40 --warmup 5 > /dev/null 2> $OUT/${TAG}v$i.err
  i=$((i+1))
done
# keep only the CSV summaries (the merged-back directory is capped)
find $OUT/${TAG}a $OUT/${TAG}f $OUT/${TAG}w $OUT/${TAG}v -type f ! -name '*kernel_stats.csv' ! -name '*counter_collection.csv' -delete 2>/dev/null
for d in $OUT/${TAG}f $OUT/${TAG}w $OUT/${TAG}v; do
  for f in $(find $d -name '*counter_collection.csv'); do
    python3 - "$f" <<'PY'
import csv, sys, collections
f = sys.argv[1]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    acc[(r["Kernel_Name"], r["Counter_Name"])].append(float(r["Counter_Value"]))
with open(f, "w", newline="") as out:
    w = csv.writer(out)
    w.writerow(["Kernel_Name", "Counter_Name", "Counter_Value", "Launches"])
    for (k, c), v in acc.items():
        w.writerow([k, c, sum(v) / len(v), len(v)])
PY
  done
done
ls -la $OUT/${TAG}a/*/ 2>/dev/null | head; cat $OUT/${TAG}_bench.json | head -c 400
