set -u
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-run40}; mkdir -p $O
export PYTHONPATH=$GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_gpu_execute_abi.py -x -q -k "windows" > $O/win.log 2>&1; rc=$?; tail -15 $O/win.log
[ $rc -ne 0 ] && exit 1
for i in 1 2; do
EVA_BATCH_TIMING=1 timeout 200 python scripts/prof_legs.py batch 7 > $O/batch.json 2>$O/batch.err
grep "EVA:" $O/batch.err | tail -3
python - $O/batch.json <<'PY'
import json,sys
j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("batch", j.get("dags_per_s"), j.get("best_dags_per_s"), j.get("bit_exact_vs_oracle"))
PY
done
