set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out
TAG=r04b
rocm-smi --showserial 2>/dev/null | grep -i "Serial N" > $OUT/${TAG}_gpu.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}a -- python3 bench.py --cpu-sample 0 --e2e 0 --steps 300 --warmup 20 > $OUT/${TAG}_bench_under_rocprofv3.json 2> $OUT/${TAG}a.err
python3 bench.py --steps 20 --warmup 5 > $OUT/${TAG}_bench_steps20.json 2> $OUT/${TAG}_b20.err
find $OUT/${TAG}a -type f ! -name '*kernel_stats.csv' -delete 2>/dev/null
ls $OUT/${TAG}a/*/ | head; head -c 300 $OUT/${TAG}_bench_steps20.json
