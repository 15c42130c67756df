#!/bin/bash
# A/B of DAG legs under environment variants (one GPU call, same box):  bash scripts/r06_ab.sh <out> "<legs>" "VAR=.. VAR=..|VAR=..|..."
# legs: dag dag_batch dag_harris_batch dag_configs ; variants separated by '|' ("" = defaults)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-ab}; mkdir -p $O
LEGS=${2:-"dag_harris_batch dag_batch dag dag_configs"}
IFS='|' read -ra VARIANTS <<< "${3:-}"
[ ${#VARIANTS[@]} -eq 0 ] && VARIANTS=("")
export PYTHONPATH=$R
cd $R
BASE=$R/eva_amd/lib/libeva_hip.so
cp $BASE /tmp/libeva_hip.base.so
trap 'cp /tmp/libeva_hip.base.so $BASE' EXIT
for vi in "${!VARIANTS[@]}"; do
  v="${VARIANTS[$vi]}"
  # "ENV=.. ENV=..@name": the library variant eva_amd/lib/variants/<name>/libeva_hip.so (scripts/build_variant.sh)
  if [[ "$v" == *@* ]]; then cp $R/eva_amd/lib/variants/${v##*@}/libeva_hip.so $BASE; vlabel="$v"; v="${v%@*}"; else cp /tmp/libeva_hip.base.so $BASE; vlabel="$v"; fi
  for leg in $LEGS; do
    f=$O/v${vi}_$leg.json
    env $v timeout 600 python bench.py --only-leg $leg ${EXTRA:-} > $f 2> $O/v${vi}_$leg.err || tail -5 $O/v${vi}_$leg.err
    python - "$f" "$leg" "$vlabel" <<'PY'
import json, sys
f, leg, v = sys.argv[1:4]
try:
    d = json.load(open(f))[leg]
except Exception as e:
    print(f"[{v}] {leg}: FAILED {e}"); sys.exit(0)
if leg in ("dag_batch", "dag_harris_batch"):
    print(f"[{v}] {leg}: {d.get('dags_per_s')} DAGs/s best {d.get('best_dags_per_s')} chunk {d.get('instances_per_device_handle')} frac {d.get('roofline',{}).get('frac')} comp {d.get('roofline',{}).get('launch_compulsory_frac')} exact {d.get('bit_exact_vs_oracle')} {d.get('error','')}")
elif leg == "dag":
    print(f"[{v}] dag: host {d.get('gpu_execute_ms')} ms resident {d.get('gpu_execute_resident_ms')} b2b {d.get('resident_back_to_back_ms')} exact {d.get('bit_exact_vs_oracle')} {d.get('error','')}")
else:
    print(f"[{v}] dag_configs: " + "  ".join(f"{k}: host {x.get('host_ms')} res {x.get('resident_ms')} exact {x.get('bit_exact_vs_oracle')} {x.get('error','')}" for k, x in d.items()))
PY
  done
done
