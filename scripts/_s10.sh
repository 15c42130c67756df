mkdir -p gpurun_out/s10
L=eva_amd/lib
(timeout 900 python -m pytest tests/test_gpu_hoist.py tests/test_gpu_configs.py tests/test_gpu_e2e.py tests/test_gpu_subdag.py tests/test_gpu_batched.py -x -q 2>&1 | tail -4) > gpurun_out/s10/tests.log
for v in g1 g4 g1 g4; do
  cp $L/variants/libeva_hip_$v.so $L/libeva_hip.so
  timeout 300 python scripts/prof_legs.py harris 15 > gpurun_out/s10/harris_$v.json 2>/dev/null
  timeout 300 python scripts/prof_legs.py batch 5 > gpurun_out/s10/batch_$v.json 2>/dev/null
  python - <<PY
import json
for w in ('harris','batch'):
    d=json.loads(open('gpurun_out/s10/%s_$v.json'%w).read().strip().splitlines()[-1])
    print('$v',w,{k:d[k] for k in d if k in ('gpu_execute_ms','gpu_execute_resident_ms','dags_per_s','best_dags_per_s','bit_exact_vs_oracle')})
PY
done
cat gpurun_out/s10/tests.log
