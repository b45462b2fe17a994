cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $R/grp
timeout 1400 python tools/variant_bench.py nb_slow_a3.so nb_quad2.so nb_quad3.so nb_xcd1.so nb_slow_a3.so nb_quad2.so nb_quad3.so nb_xcd1.so nb_slow_a3.so > $R/grp/ab14.log 2>&1
cut -c1-130 $R/grp/ab14.log
