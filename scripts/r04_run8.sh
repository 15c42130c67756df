set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/run8; mkdir -p $O
export PYTHONPATH=$GRAFT_REPO_ROOT
( time timeout 600 python -m pytest tests -x -q -m gpu ) > $O/gputests.log 2>&1 || { echo GPU TESTS FAILED; tail -40 $O/gputests.log; exit 1; }
tail -4 $O/gputests.log
timeout 900 scripts/ab_bench.sh run8 "--steps 60 --warmup 5 --no-legs --no-cpu-baseline" \
  tb:X=1 notb:X=1@notb tb_b:X=1 notb_b:X=1@notb
for v in tb notb tb notb; do
  if [ $v = notb ]; then cp eva_amd/lib/libeva_hip.so /tmp/keep.so; cp eva_amd/lib/variants/notb/libeva_hip.so eva_amd/lib/libeva_hip.so; fi
  for leg in harris batch; do timeout 200 python scripts/prof_legs.py $leg 9 > $O/${leg}_$v.json 2> $O/${leg}_$v.err; done
  if [ $v = notb ]; then cp /tmp/keep.so eva_amd/lib/libeva_hip.so; fi
  python - $v $O <<'PY'
import json,sys
l,o=sys.argv[1:3]
h=json.loads(open(f"{o}/harris_{l}.json").read().strip().splitlines()[-1]); b=json.loads(open(f"{o}/batch_{l}.json").read().strip().splitlines()[-1])
print(f"{l:10s} harris {h['gpu_execute_ms']} / {h['gpu_execute_resident_ms']} ms  batch {b['dags_per_s']} (best {b['best_dags_per_s']}) ok={h['bit_exact_vs_oracle']},{b['bit_exact_vs_oracle']}")
PY
done
