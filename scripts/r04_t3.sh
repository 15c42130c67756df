set -u
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-run41}; mkdir -p $O
export PYTHONPATH=$GRAFT_REPO_ROOT
for v in 1 0 1 0; do
EVA_BATCH_STAGGER=$v EVA_BATCH_TIMING=1 timeout 200 python scripts/prof_legs.py batch 7 > $O/batch_$v.json 2>$O/batch_$v.err
grep "EVA:" $O/batch_$v.err | tail -2
python - $O/batch_$v.json $v <<'PY'
import json,sys
j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("batch stagger=%s"%sys.argv[2], j.get("dags_per_s"), j.get("best_dags_per_s"), j.get("bit_exact_vs_oracle"))
PY
done
