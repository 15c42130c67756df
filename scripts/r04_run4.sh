set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/run4; mkdir -p $O
export PYTHONPATH=$GRAFT_REPO_ROOT
( time timeout 600 python -m pytest tests -x -q -m gpu ) > $O/gputests.log 2>&1 || { echo GPU TESTS FAILED; tail -40 $O/gputests.log; exit 1; }
tail -4 $O/gputests.log
timeout 900 scripts/ab_bench.sh run4 "--steps 60 --warmup 5 --no-legs --no-cpu-baseline" \
  loop0:EVAH_LOOP_N=0 loop8:EVAH_LOOP_N=8 loop0b:EVAH_LOOP_N=0 loop8b:EVAH_LOOP_N=8 loop4:EVAH_LOOP_N=4 loop16:EVAH_LOOP_N=16 loop32:EVAH_LOOP_N=32
for f in 0 8; do
  EVAH_LOOP_N=$f timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_loop$f.json 2> $O/bench_loop$f.err
  python - $O/bench_loop$f.json <<'PY'
import json,sys
j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1].split('/')[-1], j['value'], 'exec_path', j.get('execute_path',{}).get('triples_per_s'), 'dag', j.get('dag',{}).get('gpu_execute_ms'), j.get('dag',{}).get('gpu_execute_resident_ms'), 'batch', j.get('dag_batch',{}).get('dags_per_s'), j.get('dag_batch',{}).get('best_dags_per_s'), j.get('dag',{}).get('error'), j.get('dag_batch',{}).get('error'))
PY
done
