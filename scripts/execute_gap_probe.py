"""Where do the microseconds between the raw C-ABI op-triple and the same work through public_ctx.execute() go?
python scripts/execute_gap_probe.py [n_products] [reps]   (run on the GPU box)

Prints per-call host time of pub.execute() (the call returns before the GPU is done), the C++ split of it
(HipPublic::last_timing: set-up + inputs / DAG enqueue / outputs), the wall time per call with the queue kept
full, and the same with one synchronize per call."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: F401,E402
from eva import EvaProgram, Input, Output  # noqa: E402
from eva.ckks import CKKSCompiler  # noqa: E402
from eva.seal import generate_keys  # noqa: E402

n_products = int(sys.argv[1]) if len(sys.argv) > 1 else 32
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
N, l = 1 << 16, 10
prog = EvaProgram('op_triples', vec_size=1024)
with prog:
    for i in range(n_products):
        Output(f'z{i}', Input(f'x{i}') * Input(f'y{i}'))
prog.set_input_scales(60)
prog.set_output_ranges(20)
compiled, params, sig = CKKSCompiler(config={'warn_vec_size': 'false', 'lazy_relinearize': 'false'}).compile(prog)
params.poly_modulus_degree = N
params.prime_bits = [60] * (l + 1)
pub, sec = generate_keys(params, 17)
rng = np.random.default_rng(5)
inputs = {}
for i in range(n_products):
    inputs[f'x{i}'] = list(rng.uniform(-1, 1, 1024))
    inputs[f'y{i}'] = list(rng.uniform(-1, 1, 1024))
enc = pub.encrypt(inputs, sig)
for _ in range(4):
    out = pub.execute(compiled, enc)
pub.synchronize()


def run(n, sync_each=False, keep=False):
    host, parts, held = [], [], []
    t0 = time.perf_counter()
    for _ in range(n):
        h0 = time.perf_counter()
        out = pub.execute(compiled, enc)
        host.append(time.perf_counter() - h0)
        parts.append(list(pub.last_timing))
        if keep:
            held.append(out)
        if sync_each:
            pub.synchronize()
    pub.synchronize()
    dt = (time.perf_counter() - t0) / n
    med = lambda xs: sorted(xs)[len(xs) // 2]  # noqa: E731
    return dt, med(host), [med([p[j] for p in parts]) for j in range(3)]


for label, kw in (("queue kept full", {}), ("queue kept full (again)", {}), ("one synchronize per call", {"sync_each": True}),
                  ("outputs of 4 calls held", {"keep": True})):
    n = 4 if kw.get("keep") else reps
    dt, host, parts = run(n, **kw)
    print(f"{label:28s} {n_products / dt:9.1f} triples/s  {dt * 1e3:7.3f} ms/call  host {host * 1e3:6.3f} ms/call "
          f"(C++: inputs {parts[0]:.3f}, enqueue {parts[1]:.3f}, outputs {parts[2]:.3f})", flush=True)
# longer run: does the rate drift with the length of the timed region (clocks)?
for n in (12, 100):
    dt, host, parts = run(n)
    print(f"{n:4d} calls: {n_products / dt:9.1f} triples/s", flush=True)
