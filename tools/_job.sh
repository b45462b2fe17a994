cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02h
python tools/variant_bench.py base.so pipe3.so k128.so pipe3.so k128.so > gpurun_out/r02h/v.txt 2>&1
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r02h/pytest.log
