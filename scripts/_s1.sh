mkdir -p gpurun_out/s1
(timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -25) > gpurun_out/s1/tests.log
timeout 60 ./scripts/microbench_bfly > gpurun_out/s1/mb.log 2>&1
EVAH_FAST_REDUCE=0 timeout 300 python bench.py --steps 60 --warmup 5 --no-legs --no-cpu-baseline > gpurun_out/s1/bench_base.json 2> gpurun_out/s1/bench_base.err
timeout 300 python bench.py --steps 60 --warmup 5 --no-legs --no-cpu-baseline > gpurun_out/s1/bench_fast.json 2> gpurun_out/s1/bench_fast.err
timeout 300 python bench.py --steps 60 --warmup 5 --no-legs --no-cpu-baseline --streams 2 > gpurun_out/s1/bench_fast_s2.json 2> gpurun_out/s1/bench_fast_s2.err
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/s1/bench_full.json 2> gpurun_out/s1/bench_full.err
tail -c 300 gpurun_out/s1/tests.log; cat gpurun_out/s1/mb.log
