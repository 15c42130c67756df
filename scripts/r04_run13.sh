set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/run13; mkdir -p $O
export PYTHONPATH=$GRAFT_REPO_ROOT
( time timeout 600 python -m pytest tests -x -q -m gpu ) > $O/gputests.log 2>&1 || { echo GPU TESTS FAILED; tail -40 $O/gputests.log; exit 1; }
tail -5 $O/gputests.log | head -2
timeout 900 scripts/ab_bench.sh run13 "--steps 60 --warmup 5 --no-legs --no-cpu-baseline" \
  new:X=1 prev:X=1@prev new_b:X=1 prev_b:X=1@prev new_c:X=1 prev_c:X=1@prev
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_WAVES --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_sq -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-legs > $GRAFT_REPO_ROOT/$O/pmc_sq.log 2>&1
python $GRAFT_REPO_ROOT/scripts/pmc_sq_summary.py $GRAFT_REPO_ROOT/$O/pmc_sq "r04" | grep "loop_kernel\|ks_inner"
