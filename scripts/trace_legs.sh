# Run on the GPU box (via gpurun): kernel traces of the DAG legs, bash scripts/trace_legs.sh <out> (per-dispatch CSVs kept under gpurun_out/<out>/)
set -u
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-run31}; mkdir -p $O
export PYTHONPATH=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for leg in batch harris; do
  reps=2; [ $leg = harris ] && reps=16
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$leg -- python $GRAFT_REPO_ROOT/scripts/prof_legs.py $leg $reps > $O/$leg.log 2>&1
  python $GRAFT_REPO_ROOT/scripts/rocprof_summary.py $O/$leg "$leg" > $O/${leg}_trace.md 2>&1
done
find $O -name "*_agent_info.csv" -delete; ls -R $O | head -30
