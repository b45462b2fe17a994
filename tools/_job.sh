cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02q
python -m pytest tests -m gpu -q 2>&1 | tail -60 > gpurun_out/r02q/pytest.log
