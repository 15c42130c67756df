#!/bin/bash
# kernel traces of the DAG legs: bash scripts/r06_traces.sh <out dir under gpurun_out> "<legs: harris_batch harris batch c5>" [tag]
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-traces}; mkdir -p $O
TAG=${3:-r06}
export PYTHONPATH=$R
cd /tmp && export TMPDIR=/tmp
for leg in ${2:-"harris_batch harris batch c5"}; do
  reps=2; [ $leg = harris ] && reps=16; [ $leg = c5 ] && reps=5
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$leg -- python $R/scripts/prof_legs.py $leg $reps > $O/$leg.log 2>&1
  python $R/scripts/rocprof_summary.py $O/$leg "$TAG $leg (scripts/prof_legs.py $leg $reps): rocprofv3 --kernel-trace --stats" > $O/${TAG}_${leg}_kernel_trace.md 2>&1
  cp $O/$leg/*/*kernel_trace.csv $O/${leg}_kernel_trace.csv 2>/dev/null
  rm -rf $O/$leg
done
ls $O
