import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from eva_amd import backend
from eva_amd.hostref import coeff_modulus_create
N, k = 65536, 11
primes = coeff_modulus_create(N, [60] * k)
g = backend.Context(N, primes); l = k - 1
rng = np.random.default_rng(0)
rand = lambda pre, nl: np.stack([rng.integers(0, primes[i], size=pre + (N,), dtype=np.uint64) for i in range(nl)], axis=len(pre))
for st in (1, 2, 3, 4):
    g.upload_galois_key(g.galois_elt_from_step(st), rand((l, 2), k))
a = g.upload_ct(rand((2,), l), 2.0**40)
def timeit(name, fn, reps=20):
    for _ in range(3): fn()
    g.sync(); g.timer_start()
    for _ in range(reps): fn()
    ms = g.timer_stop() / reps
    print(f"{name:34s} {ms*1000:9.1f} us")
timeit("rotate x1", lambda: g.rotate(a, 1))
timeit("rotate_many [1]", lambda: g.rotate_many(a, [1]))
timeit("rotate_many [1,1] (same key)", lambda: g.rotate_many(a, [1, 1]))
timeit("rotate_many [1,2] (2 keys)", lambda: g.rotate_many(a, [1, 2]))
timeit("rotate_many [1,1,1,1] (same key)", lambda: g.rotate_many(a, [1, 1, 1, 1]))
timeit("rotate_many [1,2,3,4] (4 keys)", lambda: g.rotate_many(a, [1, 2, 3, 4]))
