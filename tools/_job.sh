cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02m
python -m pytest tests/test_gpu_tiles.py tests/test_gpu_onthefly.py -m gpu -x -q 2>&1 | tail -40 > gpurun_out/r02m/pytest.log
