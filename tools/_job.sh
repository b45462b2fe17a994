cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02o
python -m pytest tests/test_gpu_sharded.py -m gpu -x -q 2>&1 | tail -30 > gpurun_out/r02o/pytest.log
