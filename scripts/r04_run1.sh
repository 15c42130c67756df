set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/run1
( time python -m pytest tests/test_gpu_parity.py -x -q -m gpu ) > gpurun_out/run1/parity.log 2>&1
tail -5 gpurun_out/run1/parity.log
scripts/ab_bench.sh run1 "--steps 60 --warmup 5 --no-legs --no-cpu-baseline" \
  r03form:EVAH_FOLD_PA=0 fold:EVAH_FOLD_PA=1 r03form2:EVAH_FOLD_PA=0 fold2:EVAH_FOLD_PA=1 \
  fold_g2:EVAH_FOLD_PA=1,EVAH_KS_GROUPS=2 fold_g4:EVAH_FOLD_PA=1,EVAH_KS_GROUPS=4 fold_g11:EVAH_FOLD_PA=1,EVAH_KS_GROUPS=11
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $GRAFT_REPO_ROOT/gpurun_out/run1/counters.txt 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/run1/pmc_lds -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-legs > $GRAFT_REPO_ROOT/gpurun_out/run1/pmc_lds.log 2>&1
tail -3 $GRAFT_REPO_ROOT/gpurun_out/run1/pmc_lds.log
