set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/run3; mkdir -p $O
export PYTHONPATH=$GRAFT_REPO_ROOT
( time timeout 600 python -m pytest tests -x -q -m gpu ) > $O/gputests.log 2>&1 || { echo GPU TESTS FAILED; tail -40 $O/gputests.log; exit 1; }
tail -4 $O/gputests.log
for f in 0 1; do
  EVAH_FOLD_PA=$f timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_fold$f.json 2> $O/bench_fold$f.err
  python - $O/bench_fold$f.json <<'PY'
import json,sys
j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1].split('/')[-1], j['value'], 'exec_path', j.get('execute_path',{}).get('triples_per_s'), 'dag', j.get('dag',{}).get('gpu_execute_ms'), j.get('dag',{}).get('gpu_execute_resident_ms'), 'batch', j.get('dag_batch',{}).get('dags_per_s'), j.get('dag_batch',{}).get('best_dags_per_s'), j.get('dag',{}).get('error'), j.get('dag_batch',{}).get('error'))
PY
done
timeout 300 python bench.py --gpus 1 --shard dag --steps 5 > $O/shard_dag.json 2> $O/shard_dag.err; tail -c 400 $O/shard_dag.json
timeout 300 python bench.py --gpus 1 --shard dag --members 2 --steps 5 > $O/shard_dag_m2.json 2> $O/shard_dag_m2.err; tail -c 300 $O/shard_dag_m2.json
