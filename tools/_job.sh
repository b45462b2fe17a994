cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02r
python -m pytest tests/test_gpu_fullsize.py tests/test_gp_oracle.py -m gpu -x -q 2>&1 | tail -40 > gpurun_out/r02r/pytest.log
