cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02y
python -m pytest tests/test_gpu_tiles.py -m gpu -x -q 2>&1 | tail -5 > gpurun_out/r02y/pytest.log
python tools/bench_aux.py > gpurun_out/r02y/aux.json 2> gpurun_out/r02y/aux.err
