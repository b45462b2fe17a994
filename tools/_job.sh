cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02l
python tools/variant_bench.py k128.so tick.so:DIG_ES_TICKETS=0 tick.so:DIG_ES_TICKETS=1024 tick.so:DIG_ES_TICKETS=256 k128.so tick.so:DIG_ES_TICKETS=1024 tick.so:DIG_ES_TICKETS=256 tick.so:DIG_ES_TICKETS=256,DIG_ES_BLOCKS_PER_CU=4 > gpurun_out/r02l/v.txt 2>&1
