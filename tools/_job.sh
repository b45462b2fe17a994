cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02w2
python tools/variant_bench.py head.so nofence.so head.so nofence.so > gpurun_out/r02w2/v.txt 2>&1
python -m pytest tests/test_gpu_parity.py tests/test_gpu_pipeline.py -m gpu -x -q 2>&1 | tail -5 > gpurun_out/r02w2/pytest.log
python tools/bench_aux.py > gpurun_out/r02w2/aux.txt 2>&1
