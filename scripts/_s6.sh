mkdir -p gpurun_out/s6
(timeout 900 python -m pytest tests/test_gpu_subdag.py tests/test_gpu_resident.py -x -q 2>&1 | tail -30) > gpurun_out/s6/tests_multi.log
(timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -8) > gpurun_out/s6/tests_all.log
timeout 300 python bench.py --shard subdag --shards 3 --steps 20 --warmup 3 > gpurun_out/s6/bench_subdag.json 2> gpurun_out/s6/bench_subdag.err
timeout 300 python bench.py --shard limb --shards 4 --steps 5 --warmup 1 > gpurun_out/s6/bench_limb4.json 2> gpurun_out/s6/bench_limb4.err
cat gpurun_out/s6/tests_multi.log; tail -3 gpurun_out/s6/tests_all.log; tail -c 600 gpurun_out/s6/bench_subdag.json; tail -3 gpurun_out/s6/bench_subdag.err
