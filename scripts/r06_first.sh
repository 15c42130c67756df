#!/bin/bash
# round 6, first GPU call: the new parity tests, a group-size sweep of the Harris batch leg, C1/C2/C5 and a C5 trace
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06a; rm -rf $O; mkdir -p $O
export PYTHONPATH=$R
cd $R
timeout 900 python -m pytest tests/test_gpu_configs.py -x -q -m gpu > $O/pytest_configs.log 2>&1; tail -3 $O/pytest_configs.log
for c in 4 8 16 24 32; do
  timeout 300 python bench.py --only-leg dag_harris_batch --harris-chunk $c > $O/harris_batch_$c.json 2> $O/harris_batch_$c.err
  python - <<PY
import json
try:
    d = json.load(open("$O/harris_batch_$c.json"))["dag_harris_batch"]
    print("chunk $c:", d.get("dags_per_s"), d.get("ms_calls"), d.get("roofline", {}).get("frac"), d.get("roofline", {}).get("launch_compulsory_frac"), d.get("bit_exact_vs_oracle"), d.get("error"))
except Exception as e:
    print("chunk $c failed", e); print(open("$O/harris_batch_$c.err").read()[-1500:])
PY
done
timeout 600 python bench.py --only-leg dag_configs > $O/dag_configs.json 2> $O/dag_configs.err; cat $O/dag_configs.json | head -c 3000
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_c5 -- python $R/scripts/prof_legs.py c5 5 > $O/trace_c5.log 2>&1
python $R/scripts/rocprof_summary.py $O/trace_c5 "r06 C5 (3x3 convolution + depth-8 squaring chain, N=2^16, 13 primes): eager walk, capture, 5 resident replays: rocprofv3 --kernel-trace --stats" > $O/r06_c5_kernel_trace_start.md
timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_WAVES --output-format csv -d $O/pmc_sq_c5 -- python $R/scripts/prof_legs.py c5 2 > $O/pmc_sq_c5.log 2>&1
python $R/scripts/pmc_sq_summary.py $O/pmc_sq_c5 "r06 C5, 2 resident replays (+ eager walk and capture): SQ counters" > $O/r06_c5_sq_counters_start.md
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_hb -- python $R/scripts/prof_legs.py harris_batch 2 16 > $O/trace_hb.log 2>&1
python $R/scripts/rocprof_summary.py $O/trace_hb "r06 Harris batch (64 DAGs, N=2^15 L=8, groups of 16, 3 warm-up + 2 timed calls): rocprofv3 --kernel-trace --stats" > $O/r06_harris_batch_kernel_trace_start.md
# keep the per-dispatch csv of c5 (small) for the launch-by-launch reading; drop the rest of the raw output
cp $O/trace_c5/*/*kernel_trace.csv $O/c5_kernel_trace.csv 2>/dev/null
rm -rf $O/trace_c5 $O/pmc_sq_c5 $O/trace_hb
ls $O
