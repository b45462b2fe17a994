cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $R/grp
timeout 1400 python tools/variant_bench.py nb_base.so nb_pre.so nb_base.so nb_pre.so nb_base.so nb_pre.so > $R/grp/ab16.log 2>&1
cut -c1-130 $R/grp/ab16.log
