mkdir -p gpurun_out/s2
(timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -25) > gpurun_out/s2/tests.log
timeout 60 ./scripts/microbench_bfly > gpurun_out/s2/mb.log 2>&1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/s2/prof_batch -- python $R/scripts/prof_legs.py batch 2 > $R/gpurun_out/s2/prof_batch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/s2/prof_harris -- python $R/scripts/dag_profile.py harris8 > $R/gpurun_out/s2/prof_harris.log 2>&1
cd $R
python scripts/rocprof_summary.py gpurun_out/s2/prof_batch "r03 config 4 batch (256 Sobel DAGs, N=2^14, l=5), 1 warm-up of 32 + 2 timed calls" > gpurun_out/s2/prof_batch.md
python scripts/rocprof_summary.py gpurun_out/s2/prof_harris "r03 Harris N=2^15 L=8" > gpurun_out/s2/prof_harris.md
rm -rf gpurun_out/s2/prof_batch gpurun_out/s2/prof_harris
tail -5 gpurun_out/s2/tests.log; cat gpurun_out/s2/mb.log
