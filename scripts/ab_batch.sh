#!/bin/bash
# Run on the GPU box (via gpurun): A/B of the config-4 batch leg only (scripts/prof_legs.py batch) under EVA_* / EVAH_* settings.
#   scripts/ab_batch.sh <out> <reps> label1:VAR=val,VAR=val label2: ...
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$1; shift
REPS=$1; shift
mkdir -p $O
export PYTHONPATH=$R
cd $R
for spec in "$@"; do
  label=${spec%%:*}; rest=${spec#*:}
  envs=$(echo "$rest" | tr ',' ' ')
  timeout 200 env $envs EVA_BATCH_TIMING=1 python scripts/prof_legs.py batch $REPS > $O/batch_$label.json 2> $O/batch_$label.err
  python - $label $O/batch_$label.json <<'PY'
import json, sys
try:
    b = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    print(f"{sys.argv[1]:24s} batch {b.get('dags_per_s')} (best {b.get('best_dags_per_s')}) DAGs/s exact={b.get('bit_exact_vs_oracle')} calls={b.get('ms_calls')}")
except Exception as e:
    print(sys.argv[1], "FAILED", repr(e))
PY
done
