"""probe: replay time of a captured chain of K small dependent launches (evah_negate at N = 8192, 3 limbs),
to see how hipGraph replay cost scales with the node count on this runtime"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from eva_amd import backend
from eva_amd.hostref import coeff_modulus_create
N = 8192
primes = coeff_modulus_create(N, [60, 60, 60, 60])
g = backend.Context(N, primes)
rng = np.random.default_rng(0)
a = np.stack([np.stack([rng.integers(0, primes[i], size=N, dtype=np.uint64) for i in range(3)]) for _ in range(2)])
x0 = g.upload_ct(a, 2.0 ** 20)
for K in (8, 16, 24, 32, 40, 48, 64, 96, 128):
    g.sync()
    g.capture_begin()
    x = x0
    keep = []
    for _ in range(K):
        x = g.negate(x)
        keep.append(x)
    gr = g.capture_end()
    for _ in range(3):
        g.graph_launch(gr); g.sync()
    ts = []
    for _ in range(20):
        t0 = time.perf_counter(); g.graph_launch(gr); g.sync(); ts.append(time.perf_counter() - t0)
    ts.sort()
    print(f"K={K:4d}: replay {ts[10]*1e6:8.1f} us  = {ts[10]*1e6/K:6.2f} us per node", flush=True)
    g.graph_free(gr)
    del keep
