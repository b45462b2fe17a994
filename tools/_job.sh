cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $R/mm
timeout 900 python -m pytest tests/test_gpu_pipeline.py -q -x -k "single_split or kfold" > $R/mm/pytest.log 2>&1
tail -30 $R/mm/pytest.log
