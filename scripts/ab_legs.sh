#!/bin/bash
# Run on the GPU box (via gpurun): A/B of the DAG legs (scripts/prof_legs.py harris / batch) under different EVAH_* / EVA_*
# environments and, optionally, library variants (eva_amd/lib/variants/<name>/libeva_hip.so, scripts/build_variant.sh).
#   scripts/ab_legs.sh <out> label1:VAR=val,VAR=val[@variant] label2:[@variant] ...
# Starts with the hoisting / window tests (bounded) so that a broken build costs seconds, not the call.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$1; shift
mkdir -p $O
export PYTHONPATH=$R
cd $R
timeout 300 python -m pytest tests/test_gpu_hoist.py tests/test_gpu_execute_abi.py -x -q -k "not random" > $O/tests.log 2>&1; rc=$?; tail -3 $O/tests.log
[ $rc -ne 0 ] && exit 1
BASE=$R/eva_amd/lib/libeva_hip.so
cp $BASE /tmp/libeva_hip.base.so
for spec in "$@"; do
  label=${spec%%:*}; rest=${spec#*:}
  variant=""
  if [[ "$rest" == *@* ]]; then variant=${rest##*@}; rest=${rest%@*}; fi
  if [ -n "$variant" ]; then cp $R/eva_amd/lib/variants/$variant/libeva_hip.so $BASE; else cp /tmp/libeva_hip.base.so $BASE; fi
  envs=$(echo "$rest" | tr ',' ' ')
  timeout 200 env $envs python scripts/prof_legs.py harris 30 > $O/harris_$label.json 2> $O/harris_$label.err
  timeout 200 env $envs EVA_BATCH_TIMING=1 python scripts/prof_legs.py batch 5 > $O/batch_$label.json 2> $O/batch_$label.err
  python - $label $O/harris_$label.json $O/batch_$label.json <<'PY'
import json, sys
def last(p):
    try:
        return json.loads(open(p).read().strip().splitlines()[-1])
    except Exception as e:
        return {"error": repr(e)}
h, b = last(sys.argv[2]), last(sys.argv[3])
print(f"{sys.argv[1]:20s} harris {h.get('gpu_execute_ms')} / {h.get('gpu_execute_resident_ms')} ms exact={h.get('bit_exact_vs_oracle')}   "
      f"batch {b.get('dags_per_s')} (best {b.get('best_dags_per_s')}) DAGs/s exact={b.get('bit_exact_vs_oracle')}")
PY
done
cp /tmp/libeva_hip.base.so $BASE
