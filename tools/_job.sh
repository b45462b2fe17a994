cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $R/grp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_host_mirror.py tests/test_gpu_sharded.py tests/test_gpu_cohort_batch.py -q > $R/grp/pytest5.log 2>&1
tail -5 $R/grp/pytest5.log
timeout 600 python tools/variant_bench.py nb_xcd1.so nb_slow_a3.so nb_quad2.so nb_quad.so nb_quad2.so > $R/grp/ab9.log 2>&1
cut -c1-420 $R/grp/ab9.log
