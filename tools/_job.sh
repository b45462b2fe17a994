cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $R/tiles
( timeout 300 python tools/tile_variant_bench.py tiles_final.so tiles_timing.so; timeout 300 python tools/tile_variant_bench.py tiles_final.so:TB_C=5 tiles_timing.so:TB_C=5 ) > $R/tiles/ab9.log 2>&1
cut -c1-900 $R/tiles/ab9.log
