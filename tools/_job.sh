cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $R/gp
timeout 900 python -m pytest tests/test_gp_oracle.py tests/test_gpu_pipeline.py -q -x > $R/gp/pytest.log 2>&1; tail -4 $R/gp/pytest.log
timeout 600 python tools/bench_gp.py > $R/gp/bench_gp.json 2> $R/gp/bench_gp.err; tail -1 $R/gp/bench_gp.json
