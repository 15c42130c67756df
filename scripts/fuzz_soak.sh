#!/bin/bash
# Fuzz soak on the GPU box: thousands of random parameter sets / random programs against the oracle, MANY processes
# sharing the one GPU (pytest-xdist).  The contention is the point: r6's soak found two transfer races that no run with
# the GPU to itself had ever shown (profiles/r06_tuning_notes.md section 14).
#   bash scripts/fuzz_soak.sh [op seeds] [dag seeds] [workers]      (defaults 12000 12000 32; ~4 + ~5 minutes)
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/soak; mkdir -p $O
cd $R
EVA_FUZZ_SEEDS=${1:-12000} timeout 1500 python -m pytest tests/test_gpu_op_fuzz.py -q -n ${3:-32} -p no:cacheprovider > $O/fuzz_op.log 2>&1
grep -E "^FAILED|passed|failed|MISMATCH" $O/fuzz_op.log | tail -12
EVA_FUZZ_SEEDS=${2:-12000} timeout 1500 python -m pytest tests/test_gpu_fuzz.py -q -n ${3:-32} -p no:cacheprovider > $O/fuzz_dag.log 2>&1
grep -E "^FAILED|passed|failed|MISMATCH" $O/fuzz_dag.log | tail -12
