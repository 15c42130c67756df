set -u
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
timeout 200 python -m pytest tests/test_gpu_parity.py -x -q > gpurun_out/tb_parity.log 2>&1; rc=$?; tail -3 gpurun_out/tb_parity.log
[ $rc -ne 0 ] && exit 1
bash scripts/ab_bench.sh ${1:-run42} "--steps 60 --warmup 10 --no-cpu-baseline --no-legs" new1: old1:@tbmul0 new2: old2:@tbmul0 new3: old3:@tbmul0
