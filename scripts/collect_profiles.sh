#!/bin/bash
# Run on the GPU box (via gpurun): bench line + rocprofv3 kernel trace + PMC traffic passes of the
# same bench.py command.  Outputs under gpurun_out/prof_bench/ ; summaries are made by
# scripts/rocprof_summary.py and scripts/pmc_traffic.py and committed under profiles/.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_bench
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 20 --warmup 3 --no-legs"
python $R/bench.py > $OUT/bench.json 2> $OUT/bench.err
tail -c 600 $OUT/bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python $R/bench.py $ARGS --no-cpu-baseline > $OUT/trace.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-legs > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-legs > $OUT/pmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_WAVES --output-format csv -d $OUT/pmc_sq -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-legs > $OUT/pmc_sq.log 2>&1
ls $OUT/trace/*/ | head
