mkdir -p gpurun_out/s7
(timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -8) > gpurun_out/s7/tests_all.log
timeout 300 python bench.py --steps 60 --warmup 5 --no-legs --no-cpu-baseline > gpurun_out/s7/bench.json 2> gpurun_out/s7/bench.err
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/s7/smoke.log 2>&1
tail -4 gpurun_out/s7/tests_all.log; tail -c 300 gpurun_out/s7/bench.json; tail -2 gpurun_out/s7/smoke.log
