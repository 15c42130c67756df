set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/run2; mkdir -p $O
export PYTHONPATH=$GRAFT_REPO_ROOT
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 || { echo SMOKE FAILED; tail -5 $O/smoke.log; exit 1; }
tail -1 $O/smoke.log
( time timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "relinearize or op_triple or rotate_bit_exact or more_than_16 or key_switch_extremes" ) > $O/parity.log 2>&1 || { echo PARITY FAILED; tail -30 $O/parity.log; exit 1; }
tail -4 $O/parity.log
timeout 900 scripts/ab_bench.sh run2 "--steps 60 --warmup 5 --no-legs --no-cpu-baseline" \
  r03form:EVAH_FOLD_PA=0 fold:EVAH_FOLD_PA=1 r03form2:EVAH_FOLD_PA=0 fold2:EVAH_FOLD_PA=1 \
  fold_g2:EVAH_FOLD_PA=1,EVAH_KS_GROUPS=2 fold_g4:EVAH_FOLD_PA=1,EVAH_KS_GROUPS=4 fold_g11:EVAH_FOLD_PA=1,EVAH_KS_GROUPS=11
cd /tmp && export TMPDIR=/tmp
timeout 60 rocprofv3 -L > $GRAFT_REPO_ROOT/$O/counters.txt 2>&1
timeout 240 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_lds -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-legs > $GRAFT_REPO_ROOT/$O/pmc_lds.log 2>&1
tail -2 $GRAFT_REPO_ROOT/$O/pmc_lds.log
