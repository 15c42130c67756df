#!/bin/bash
# group-size sweep of the two batch legs: bash scripts/r06_chunk_sweep.sh <out>
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-sweep}; mkdir -p $O
export PYTHONPATH=$R
cd $R
for c in 4 6 8 12 16; do
  timeout 300 python bench.py --only-leg dag_harris_batch --harris-chunk $c > $O/hb_$c.json 2> $O/hb_$c.err
  python -c "
import json; d=json.load(open('$O/hb_$c.json'))['dag_harris_batch']; print('harris chunk $c:', d.get('dags_per_s'), d.get('best_dags_per_s'), d.get('roofline',{}).get('frac'), d.get('bit_exact_vs_oracle'), d.get('error',''))"
done
for c in 16 20 24 28 32; do
  EVA_BATCH_CHUNK=$c timeout 300 python bench.py --only-leg dag_batch > $O/sb_$c.json 2> $O/sb_$c.err
  python -c "
import json; d=json.load(open('$O/sb_$c.json'))['dag_batch']; print('sobel chunk $c:', d.get('dags_per_s'), d.get('best_dags_per_s'), d.get('instances_per_device_handle'), d.get('bit_exact_vs_oracle'), d.get('error',''))"
done
