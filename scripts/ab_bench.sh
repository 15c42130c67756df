#!/bin/bash
# Run on the GPU box (via gpurun): A/B of bench.py under different EVAH_* environments (and, optionally, library
# variants under eva_amd/lib/variants/<name>/libeva_hip.so), one JSON line per configuration in
# gpurun_out/<out>/<label>.json plus a one-line summary per configuration in gpurun_out/<out>/summary.txt.
#   scripts/ab_bench.sh <out> "<bench args>" label1:VAR=val,VAR=val[@variant] label2:... 
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$1; shift
ARGS=$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BASE=$R/eva_amd/lib/libeva_hip.so
cp $BASE /tmp/libeva_hip.base.so
for spec in "$@"; do
  label=${spec%%:*}; rest=${spec#*:}
  variant=""
  if [[ "$rest" == *@* ]]; then variant=${rest##*@}; rest=${rest%@*}; fi
  if [ -n "$variant" ]; then cp $R/eva_amd/lib/variants/$variant/libeva_hip.so $BASE; else cp /tmp/libeva_hip.base.so $BASE; fi
  envs=$(echo "$rest" | tr ',' ' ')
  timeout 240 env $envs python $R/bench.py $ARGS > $OUT/$label.json 2> $OUT/$label.err
  python - "$label" $OUT/$label.json >> $OUT/summary.txt <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    r = j.get("roofline", {})
    alone = r.get("by_class_us_alone") or r.get("by_class_us") or {}
    print(f"{sys.argv[1]:28s} {j['value']:9.1f} /s  {j['ms_per_step']:.3f} ms  sum_alone={sum(alone.values()):7.1f}us  " +
          " ".join(f"{k}={v:.1f}" for k, v in sorted(alone.items())))
except Exception as e:
    print(f"{sys.argv[1]:28s} FAILED {e!r}")
PY
done
cp /tmp/libeva_hip.base.so $BASE
cat $OUT/summary.txt
