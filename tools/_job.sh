cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $R/r02p
python bench.py > $R/r02p/bench.json 2> $R/r02p/bench.err
python bench.py --steps 20 --warmup 5 --cpu-sample 0 > $R/r02p/bench20.json 2> $R/r02p/bench20.err
python bench.py --mode strong --cpu-sample 0 --steps 200 > $R/r02p/bench_strong1.json 2> $R/r02p/bench_strong1.err
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/r02a -o kt -- python3 $GRAFT_REPO_ROOT/bench.py --cpu-sample 0 > $R/r02p/bench_under_rocprofv3.json 2> $R/r02p/kt.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/r02f -o pmc -- python3 $GRAFT_REPO_ROOT/bench.py --cpu-sample 0 --steps 40 --warmup 5 > /dev/null 2> $R/r02p/f.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/r02w -o pmc -- python3 $GRAFT_REPO_ROOT/bench.py --cpu-sample 0 --steps 40 --warmup 5 > /dev/null 2> $R/r02p/w.err
for set in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_THREAD_CYCLES_VALU SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_INT32" "GRBM_GUI_ACTIVE SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_VALU_INT64"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set --output-format csv -d $R/r02v/$tag -o pmc -- python3 $GRAFT_REPO_ROOT/bench.py --cpu-sample 0 --steps 40 --warmup 5 > /dev/null 2> $R/r02p/v_$tag.err
done
find $R/r02a $R/r02f $R/r02w $R/r02v -name "*.db" -delete
find $R/r02a -name "*kernel_trace.csv" -delete
