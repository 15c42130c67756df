mkdir -p gpurun_out/s8
for v in "" "--fused-multiply" "--group 64" "--streams 2 --fused-multiply"; do
  n=$(echo "x$v" | tr -d ' -')
  timeout 300 python bench.py --steps 60 --warmup 5 --no-legs --no-cpu-baseline $v > gpurun_out/s8/bench_$n.json 2> gpurun_out/s8/bench_$n.err
  python -c "
import json,sys
d=json.loads(open('gpurun_out/s8/bench_$n.json').read().strip().splitlines()[-1]); print('$n', d['value'], d['roofline']['by_class_us'])"
done
