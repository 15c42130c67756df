mkdir -p gpurun_out/s9
for v in "--fused-multiply --streams 2" "--fused-multiply --streams 3" "--fused-multiply --streams 4" "--fused-multiply --streams 2 --group 16" "--fused-multiply --streams 2 --batch 128 --group 32" "--fused-multiply --streams 4 --batch 128 --group 32"; do
  n=$(echo "x$v" | tr -d ' -')
  timeout 300 python bench.py --steps 60 --warmup 5 --no-legs --no-cpu-baseline $v > gpurun_out/s9/bench_$n.json 2> gpurun_out/s9/bench_$n.err
  python -c "
import json,sys
d=json.loads(open('gpurun_out/s9/bench_$n.json').read().strip().splitlines()[-1]); print('$n', d['value'])"
done
for f in 0 1; do
EVAH_FUSE_MUL=$f timeout 300 python scripts/prof_legs.py harris 15 > gpurun_out/s9/harris_fm$f.json 2>/dev/null
EVAH_FUSE_MUL=$f timeout 300 python scripts/prof_legs.py batch 5 > gpurun_out/s9/batch_fm$f.json 2>/dev/null
EVAH_FUSE_MUL=$f timeout 300 python scripts/prof_legs.py execute 10 > gpurun_out/s9/execute_fm$f.json 2>/dev/null
python - <<PY
import json
for w in ('harris','batch','execute'):
    d=json.loads(open('gpurun_out/s9/%s_fm$f.json'%w).read().strip().splitlines()[-1])
    print('fuse_mul=$f',w,{k:d[k] for k in d if k in ('gpu_execute_ms','gpu_execute_resident_ms','dags_per_s','best_dags_per_s','triples_per_s','bit_exact_vs_oracle')})
PY
done
(timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3)
