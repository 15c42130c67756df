"""execute() wall time for the BASELINE DAG configs on one GPU vs the CPU oracle walking the same
compiled DAG (1 core): C1 README polynomial, C2 Sobel N=2^13, C3 Harris N=2^15, one instance of
C4 (Sobel N=2^14) and C5 (3x3 convolution + depth-8 squaring chain, N=2^16, 13 primes).
usage: dag_bench.py [reps] [--cpu] [--only C1,C5]

The GPU legs use only the product.  --cpu adds the reported CPU baseline of each DAG — the same role
as bench.py's cpu_baseline leg — by walking the compiled DAG in C over the CPU oracle (oracle/eva_oracle_dag.c through
tests/oracle_executor.c_walk); nothing of it is on the measured GPU path."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from eva.ckks import CKKSCompiler
from eva.seal import generate_keys
from eva.metric import valuation_mse
from eva import evaluate
from eva_amd.workloads import sobel as _sobel
from eva_amd.workloads import harris as _harris, image as _image

reps = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 5
no_cpu = "--cpu" not in sys.argv
only = sys.argv[sys.argv.index("--only") + 1].split(",") if "--only" in sys.argv else None

def run(name, prog, N, inputs=None, pad_primes=0):
    if only and name.split()[0] not in only:
        return
    compiled, params, sig = CKKSCompiler(config={'warn_vec_size': 'false'}).compile(prog)
    if N:
        params.poly_modulus_degree = N
    if pad_primes and len(params.prime_bits) < pad_primes:  # pad with 60-bit primes to the stated L
        pb = list(params.prime_bits)
        params.prime_bits = pb[:1] + [60] * (pad_primes - len(pb)) + pb[1:]
    N = params.poly_modulus_degree
    pub, sec = generate_keys(params, 1)
    inputs = inputs if inputs is not None else _image(4096)
    enc = pub.encrypt(inputs, sig)
    ops = {}
    for d in compiled._dump():
        ops[str(d["op"]).split(".")[-1]] = ops.get(str(d["op"]).split(".")[-1], 0) + 1
    # resident valuations (the default): execute() enqueues and returns — a call's latency is enqueue + synchronize
    for _ in range(3):
        out = pub.execute(compiled, enc)  # warm-up: device ctx, key upload, pool; graph capture
    pub.synchronize()
    tr = []
    for _ in range(reps):
        t0 = time.perf_counter(); out = pub.execute(compiled, enc); pub.synchronize(); tr.append(time.perf_counter() - t0)
    # host valuations: upload + run + download per call, as the reference hands values over
    enc.to_host(True)
    pub.resident = False
    for _ in range(2):
        out = pub.execute(compiled, enc)
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); out = pub.execute(compiled, enc); ts.append(time.perf_counter() - t0)
    tm = pub.last_timing
    mse = valuation_mse(sec.decrypt(out, sig), evaluate(compiled, inputs))
    line = f"{name}: N={N} primes={list(params.prime_bits)} terms={sum(ops.values())} {ops}\n  GPU execute(): min {min(ts)*1e3:.2f} ms  median {sorted(ts)[len(ts)//2]*1e3:.2f} ms (host valuations)   resident: median {sorted(tr)[len(tr)//2]*1e3:.3f} ms   MSE {mse:.2e}  [upload {tm[0]:.2f} | host enqueue {tm[1]:.2f} | drain+download {tm[2]:.2f} ms]"
    if not no_cpu:
        # the compiled DAG walked in C over the oracle (oracle/eva_oracle_dag.c): serial forwardPass and the
        # dependency-counting traversal on pthreads — lowering / encoding excluded, as key and plaintext
        # preparation is for the GPU figure
        from oracle_executor import c_walk
        import numpy as np
        ref, tc = c_walk(pub, compiled, enc, threads=1)
        same = all(np.array_equal(out.get(n)[4], ref[n]) for n in ref)
        line += f"\n  CPU walk in C over the oracle (1 core): {tc*1e3:.1f} ms   -> GPU speed-up {tc/min(ts):.1f}x   outputs bit-identical: {same}"
        nthr = min(os.cpu_count() or 1, 64)
        _, tp = c_walk(pub, compiled, enc, threads=nthr)
        line += f"\n  CPU walk in C, dependency-counting on {nthr} threads: {tp*1e3:.1f} ms   -> GPU speed-up {tp/min(ts):.1f}x"
    print(line, flush=True)

from eva import EvaProgram, Input, Output
poly = EvaProgram('Polynomial', vec_size=1024)
with poly:
    x = Input('x')
    Output('y', 3 * x ** 2 + 5 * x - 2)
poly.set_output_ranges(30); poly.set_input_scales(30)
run("C1 readme-poly", poly, None, {'x': [i / 1024.0 for i in range(1024)]})

sob = _sobel(64, 64, 4096); sob.set_input_scales(25); sob.set_output_ranges(10)
run("C2 sobel", sob, 8192)
run("C3 harris", _harris(), 32768)
run("C4 sobel (one of the batch)", sob, 16384)

deep = EvaProgram('conv+depth8', vec_size=4096)
with deep:
    image = Input('image')
    acc = None
    for i in range(3):
        for j in range(3):
            t = (image << (i * 64 + j)) * (1.0 / 9.0)
            acc = t if acc is None else acc + t
    for _ in range(8):
        acc = acc * acc
    Output('y', acc)
deep.set_input_scales(30); deep.set_output_ranges(20)
run("C5 conv+depth-8", deep, 65536, pad_primes=13)
