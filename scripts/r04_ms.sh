# GPU box: tests of the hoisted path, the full suite, then the DAG legs against a library variant (prev)
set -u
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-run35}; mkdir -p $O
export PYTHONPATH=$GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_gpu_hoist.py -x -q > $O/hoist.log 2>&1; rc=$?; tail -15 $O/hoist.log
[ $rc -ne 0 ] && exit 1
timeout 500 python -m pytest tests -x -q -m gpu > $O/gputests.log 2>&1; rc=$?; tail -8 $O/gputests.log
[ $rc -ne 0 ] && exit 1
BASE=$GRAFT_REPO_ROOT/eva_amd/lib/libeva_hip.so
cp $BASE /tmp/new.so
for v in new prev new prev; do
  if [ $v = prev ]; then cp $GRAFT_REPO_ROOT/eva_amd/lib/variants/prev/libeva_hip.so $BASE; else cp /tmp/new.so $BASE; fi
  timeout 200 python scripts/prof_legs.py harris 30 > $O/harris_$v.json 2>$O/harris_$v.err
  python - $O/harris_$v.json $v <<'PY'
import json,sys
j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("harris %s"%sys.argv[2], {k:j[k] for k in j if k.endswith("_ms") and "cpu" not in k or "exact" in k})
PY
  timeout 200 python scripts/prof_legs.py batch 5 > $O/batch_$v.json 2>$O/batch_$v.err
  python - $O/batch_$v.json $v <<'PY'
import json,sys
j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("batch %s"%sys.argv[2], j.get("dags_per_s"), j.get("best_dags_per_s"), j.get("bit_exact_vs_oracle"))
PY
done
cp /tmp/new.so $BASE
