cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $R/full
timeout 1700 python -m pytest tests -q -m gpu -x --durations=15 > $R/full/pytest_gpu.log 2>&1
echo "pytest exit $?" >> $R/full/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $R/full/smoke.log 2>&1
echo "smoke exit $?" >> $R/full/smoke.log
tail -5 $R/full/pytest_gpu.log; tail -3 $R/full/smoke.log
