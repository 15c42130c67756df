"""execute() wall time for the BASELINE DAG configs (Sobel N=2^13, Harris N=2^15) on the GPU vs
the CPU oracle walking the same compiled DAG (1 core).  usage: dag_bench.py [reps] [--no-cpu]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from eva.ckks import CKKSCompiler
from eva.seal import generate_keys
from eva.metric import valuation_mse
from eva import evaluate
from test_compiler import _sobel
from test_gpu_e2e import _harris, _image

reps = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 5
no_cpu = "--no-cpu" in sys.argv

def run(name, prog, N):
    compiled, params, sig = CKKSCompiler(config={'warn_vec_size': 'false'}).compile(prog)
    params.poly_modulus_degree = N
    pub, sec = generate_keys(params, 1)
    inputs = _image(4096)
    enc = pub.encrypt(inputs, sig)
    ops = {}
    for d in compiled._dump():
        ops[str(d["op"]).split(".")[-1]] = ops.get(str(d["op"]).split(".")[-1], 0) + 1
    out = pub.execute(compiled, enc)  # warm-up: device ctx, key upload, pool
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); out = pub.execute(compiled, enc); ts.append(time.perf_counter() - t0)
    tm = pub.last_timing
    mse = valuation_mse(sec.decrypt(out, sig), evaluate(compiled, inputs))
    line = f"{name}: N={N} primes={list(params.prime_bits)} terms={sum(ops.values())} {ops}\n  GPU execute(): min {min(ts)*1e3:.2f} ms  median {sorted(ts)[len(ts)//2]*1e3:.2f} ms   MSE {mse:.2e}  [upload {tm[0]:.2f} | host enqueue {tm[1]:.2f} | drain+download {tm[2]:.2f} ms]"
    if not no_cpu:
        from evatest import oracle_execute
        t0 = time.perf_counter(); oracle_execute(pub, compiled, enc); tc = time.perf_counter() - t0
        line += f"\n  CPU oracle walk (1 core): {tc*1e3:.1f} ms   -> GPU speed-up {tc/min(ts):.1f}x"
    print(line, flush=True)

sob = _sobel(64, 64, 4096); sob.set_input_scales(25); sob.set_output_ranges(10)
run("sobel", sob, 8192)
run("harris", _harris(), 32768)
