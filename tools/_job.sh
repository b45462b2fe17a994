cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $R/full
timeout 1700 python -m pytest tests -q -m gpu -x > $R/full/pytest_gpu.log 2>&1
echo "pytest exit $?" >> $R/full/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $R/full/smoke.log 2>&1
tail -4 $R/full/pytest_gpu.log; tail -2 $R/full/smoke.log
python bench.py --cpu-sample 0 > $R/full/bench1000.json 2> $R/full/bench1000.err; cat $R/full/bench1000.json | cut -c1-900
python bench.py --cpu-sample 0 --steps 20 --warmup 5 > $R/full/bench20.json 2> $R/full/bench20.err; cat $R/full/bench20.json | cut -c1-700
