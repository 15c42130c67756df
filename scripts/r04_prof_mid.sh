set -u
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/run12; mkdir -p $O
export PYTHONPATH=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_WAVES --output-format csv -d $O/pmc_sq -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-legs > $O/pmc_sq.log 2>&1
python $GRAFT_REPO_ROOT/scripts/pmc_sq_summary.py $O/pmc_sq "r04 mid-round" > $O/sq.md; cat $O/sq.md
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python $GRAFT_REPO_ROOT/scripts/prof_legs.py batch 2 > $O/trace.log 2>&1
python $GRAFT_REPO_ROOT/scripts/rocprof_summary.py $O/trace "batch" > $O/batch_trace.md 2>&1; head -40 $O/batch_trace.md
