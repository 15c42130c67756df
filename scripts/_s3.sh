mkdir -p gpurun_out/s3
L=eva_amd/lib
run_bench() { timeout 300 python bench.py --steps 60 --warmup 5 --no-legs --no-cpu-baseline > gpurun_out/s3/bench_$1.json 2> gpurun_out/s3/bench_$1.err; }
cp $L/variants/libeva_hip_base.so $L/libeva_hip.so; run_bench base
cp $L/variants/libeva_hip_madlo.so $L/libeva_hip.so; run_bench madlo
(timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_hoist.py tests/test_gpu_configs.py -x -q 2>&1 | tail -5) > gpurun_out/s3/tests_madlo.log
timeout 300 python bench.py --steps 60 --warmup 5 --no-legs --no-cpu-baseline --streams 2 > gpurun_out/s3/bench_madlo_s2.json 2> gpurun_out/s3/bench_madlo_s2.err
for v in 1024 2048 4096 8192; do
  EVAH_FUSE_SMALL=$v timeout 300 python scripts/prof_legs.py batch 3 > gpurun_out/s3/batch_madlo_fs$v.json 2>/dev/null
  EVAH_FUSE_SMALL=$v timeout 300 python scripts/prof_legs.py harris 9 > gpurun_out/s3/harris_madlo_fs$v.json 2>/dev/null
done
cp $L/variants/libeva_hip_base.so $L/libeva_hip.so
for v in 2048 8192; do
  EVAH_FUSE_SMALL=$v timeout 300 python scripts/prof_legs.py batch 3 > gpurun_out/s3/batch_base_fs$v.json 2>/dev/null
  EVAH_FUSE_SMALL=$v timeout 300 python scripts/prof_legs.py harris 9 > gpurun_out/s3/harris_base_fs$v.json 2>/dev/null
done
cat gpurun_out/s3/tests_madlo.log
