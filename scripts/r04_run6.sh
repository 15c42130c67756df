set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/run6; mkdir -p $O
export PYTHONPATH=$GRAFT_REPO_ROOT
( time timeout 600 python -m pytest tests -x -q -m gpu ) > $O/gputests.log 2>&1 || { echo GPU TESTS FAILED; tail -40 $O/gputests.log; exit 1; }
tail -4 $O/gputests.log
for rep in 1 2; do
for cfg in "off:EVAH_LOOP_N=0" "thr2048:EVAH_LOOP_N=8" "always:EVAH_LOOP_N=8 EVAH_LOOP_MIN_WGS=0" "thr8192:EVAH_LOOP_N=8 EVAH_LOOP_MIN_WGS=8192"; do
  label=${cfg%%:*}; envs=${cfg#*:}
  for leg in harris batch; do
    env $envs timeout 200 python scripts/prof_legs.py $leg 9 > $O/${leg}_${label}_$rep.json 2> $O/${leg}_${label}_$rep.err
  done
  python - $label $rep $O <<'PY'
import json,sys
l,r,o=sys.argv[1:4]
h=json.loads(open(f"{o}/harris_{l}_{r}.json").read().strip().splitlines()[-1]); b=json.loads(open(f"{o}/batch_{l}_{r}.json").read().strip().splitlines()[-1])
print(f"{l:10s} rep{r} harris {h['gpu_execute_ms']} / {h['gpu_execute_resident_ms']} ms  batch {b['dags_per_s']} (best {b['best_dags_per_s']}) ok={h['bit_exact_vs_oracle']},{b['bit_exact_vs_oracle']}")
PY
done
done
