set -u
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-run33}; mkdir -p $O
export PYTHONPATH=$GRAFT_REPO_ROOT
EVA_BATCH_TIMING=1 timeout 200 python scripts/prof_legs.py batch 5 > $O/batch.json 2>$O/batch.err
grep "EVA:" $O/batch.err | tail -8
python - $O/batch.json <<'PY'
import json,sys
j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("batch", j.get("dags_per_s"), j.get("best_dags_per_s"), j.get("bit_exact_vs_oracle"))
PY
