set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/run9; mkdir -p $O
export PYTHONPATH=$GRAFT_REPO_ROOT
( time timeout 600 python -m pytest tests -x -q -m gpu ) > $O/gputests.log 2>&1 || { echo GPU TESTS FAILED; tail -40 $O/gputests.log; exit 1; }
tail -4 $O/gputests.log
timeout 900 scripts/ab_bench.sh run9 "--steps 60 --warmup 5 --no-legs --no-cpu-baseline" \
  lazy:X=1 prev:X=1@prev lazy_b:X=1 prev_b:X=1@prev lazy_c:X=1 prev_c:X=1@prev
