"""probe: n rotations of one ciphertext (B instances) through evah_rotate_many, hoisted vs not
(EVAH_HOIST), device time per call measured with HIP events.  usage: hoist_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from eva_amd import backend
from eva_amd.hostref import coeff_modulus_create

def ctx(N, primes, hoist):
    os.environ["EVAH_HOIST"] = "1" if hoist else "0"
    os.environ["EVAH_HOIST_MIN_TILES"] = "0"
    return backend.Context(N, primes)

def run(N, k, B, steps, reps=10):
    primes = coeff_modulus_create(N, [60] * k)
    rng = np.random.default_rng(1)
    l = k - 1
    a = rng.integers(0, 1 << 59, size=(B, 2, l, N), dtype=np.uint64)
    key = rng.integers(0, 1 << 59, size=(l, 2, k, N), dtype=np.uint64)
    res = []
    for hoist in (0, 1):
        g = ctx(N, primes, hoist)
        for st in steps:
            g.upload_galois_key(g.galois_elt_from_step(st), key)
        A = g.upload_ct_batch(a, 2.0 ** 20) if B > 1 else g.upload_ct(a[0], 2.0 ** 20)
        for _ in range(2):
            outs = g.rotate_many(A, steps); del outs
        g.sync()
        g.timer_start()
        for _ in range(reps):
            outs = g.rotate_many(A, steps); del outs
        res.append(g.timer_stop() / reps)
        if hoist:
            g.profile(True); g.profile_reset()
            outs = g.rotate_many(A, steps); del outs
            g.sync()
            prof = {k_: (n_, round(ms * 1e3, 1)) for k_, (n_, ms) in g.profile_get().items() if n_}
            g.profile(False)
        del A, g
    print(f"N=2^{int(np.log2(N))} l={l} B={B} n={len(steps)}: unhoisted {res[0]*1e3:8.1f} us  hoisted {res[1]*1e3:8.1f} us  ({res[0]/res[1]:.2f}x)  hoisted classes (launches, us): {prof}", flush=True)

S8 = [1, 2, 64, 65, 66, 128, 129, 130]
run(8192, 5, 1, S8)
run(16384, 5, 1, S8)
run(16384, 5, 32, S8, reps=5)
run(32768, 5, 1, S8)
run(32768, 9, 1, S8)
run(65536, 13, 1, S8, reps=5)
run(65536, 11, 1, S8[:2], reps=5)
run(65536, 11, 4, S8, reps=3)
