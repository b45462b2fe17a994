cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $R/grp
timeout 1400 python tools/variant_bench.py nb_quad3.so nb_mark2.so nb_quad3.so nb_mark2.so nb_quad3.so nb_mark2.so > $R/grp/ab15.log 2>&1
cut -c1-330 $R/grp/ab15.log
