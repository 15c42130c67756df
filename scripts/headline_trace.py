"""The headline's execute() calls alone, for a kernel trace (rocprofv3 --kernel-trace --output-format csv -- python scripts/headline_trace.py [steps]).
Same program / sizes as bench.py's timed region (32 products per call, N = 2^16, L = 10, resident valuations)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: F401,E402
import bench  # noqa: E402
from eva.seal import generate_keys  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
G, N, l = 32, 1 << 16, 10
compiled, params, sig = bench.triple_program(G, N, l)
pub, sec = generate_keys(params, 17)
rng = np.random.default_rng(1)
vals = []
for _ in range(2):
    inputs = {}
    for i in range(G):
        inputs[f'x{i}'] = list(rng.uniform(-1, 1, 1024))
        inputs[f'y{i}'] = list(rng.uniform(-1, 1, 1024))
    vals.append(pub.encrypt(inputs, sig))
for _ in range(3):
    outs = [pub.execute(compiled, v) for v in vals]
pub.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    outs = [pub.execute(compiled, v) for v in vals]
pub.synchronize()
dt = time.perf_counter() - t0
print(f"{steps} steps, {dt / steps * 1e3:.3f} ms per step, {steps * 2 * G / dt:.1f} triples/s")
