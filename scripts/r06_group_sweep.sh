#!/bin/bash
# (group size, issue queues) sweep of the Harris batch leg: bash scripts/r06_group_sweep.sh <out> "c:d c:d ..."
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-gsweep}; mkdir -p $O
export PYTHONPATH=$R
cd $R
for cd in ${2:-"12:4 8:4 11:3 11:6 13:5 16:4 16:2 22:3"}; do
  c=${cd%%:*}; d=${cd##*:}
  EVA_BATCH_DEPTH=$d timeout 300 python bench.py --only-leg dag_harris_batch --harris-chunk $c > $O/hb_${c}_$d.json 2> $O/hb_${c}_$d.err
  python -c "
import json; d=json.load(open('$O/hb_${c}_$d.json'))['dag_harris_batch']; print('harris chunk $c depth $d:', d.get('dags_per_s'), d.get('best_dags_per_s'), d.get('bit_exact_vs_oracle'), d.get('error',''))"
done
for cd in ${3:-}; do
  c=${cd%%:*}; d=${cd##*:}
  EVA_BATCH_DEPTH=$d EVA_BATCH_CHUNK=$c timeout 300 python bench.py --only-leg dag_batch > $O/sb_${c}_$d.json 2> $O/sb_${c}_$d.err
  python -c "
import json; d=json.load(open('$O/sb_${c}_$d.json'))['dag_batch']; print('sobel chunk $c depth $d:', d.get('dags_per_s'), d.get('best_dags_per_s'), d.get('bit_exact_vs_oracle'), d.get('error',''))"
done
