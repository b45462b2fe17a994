cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $R/full
timeout 1700 python -m pytest tests -q -m gpu -x --durations=8 > $R/full/pytest_gpu.log 2>&1
echo "pytest exit $?" >> $R/full/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $R/full/smoke.log 2>&1
tail -14 $R/full/pytest_gpu.log; tail -2 $R/full/smoke.log
timeout 900 python tools/bench_aux.py > $R/full/aux.json 2> $R/full/aux.err; python - <<'PY'
import json,os
d=json.load(open(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/full/aux.json"))
print(d["base_tile_probs"]); print(d["tiled_nb_test"][0]["ms"], d["count_contexts"][0]["ms"])
PY
