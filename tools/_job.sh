cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $R/dist
BENCH_FORCE_DIST=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-sample 0 > $R/dist/b1.json 2> $R/dist/b1.err; echo rc=$?; cut -c1-400 $R/dist/b1.json; tail -3 $R/dist/b1.err
BENCH_FORCE_DIST=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-sample 0 --mode strong > $R/dist/b2.json 2> $R/dist/b2.err; echo rc=$?; cut -c1-300 $R/dist/b2.json; tail -3 $R/dist/b2.err
