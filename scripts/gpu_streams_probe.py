"""Throughput of independent op-triples issued round-robin over S forked contexts (HIP streams)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from eva_amd import backend
from eva_amd.hostref import coeff_modulus_create

N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
k = int(sys.argv[2]) if len(sys.argv) > 2 else 11
primes = coeff_modulus_create(N, [60] * k)
g = backend.Context(N, primes)
l = k - 1
rng = np.random.default_rng(0)
def rand(prefix, nl):
    return np.stack([rng.integers(0, primes[i], size=prefix + (N,), dtype=np.uint64) for i in range(nl)], axis=len(prefix))
g.upload_relin_key(rand((l, 2), k))
a = g.upload_ct(rand((2,), l), 2.0**40)
b = g.upload_ct(rand((2,), l), 2.0**40)
for S in (int(x) for x in (sys.argv[3].split(",") if len(sys.argv) > 3 else "1,2,4,8,16".split(","))):
    ctxs = [g] + [g.fork() for _ in range(S - 1)]
    def run(n):
        for i in range(n):
            c = ctxs[i % S]
            m = c.multiply(a, b); r = c.relinearize(m); o = c.rescale(r, 60)
            m.free(); r.free(); o.free()
    run(2 * S)
    for c in ctxs: c.sync()
    n = max(128, 8 * S)
    t0 = time.perf_counter()
    run(n)
    t_host = time.perf_counter() - t0
    for c in ctxs: c.sync()
    dt = time.perf_counter() - t0
    print(f"S={S:2d}: {n/dt:9.1f} triples/s   {dt/n*1e6:7.1f} us/triple   host enqueue {t_host/n*1e6:6.1f} us/triple")
    for c in ctxs[1:]: c.close()
