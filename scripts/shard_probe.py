"""probe: limb-sharded relinearize for a given (N, k, G) vs the oracle (debug helper)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from eva_amd.shard import ShardedEvaluator
from eva_amd.hostref import coeff_modulus_create
N, G = int(sys.argv[1]), int(sys.argv[2])
bits = [60, 30, 60, 60, 60]
primes = coeff_modulus_create(N, bits)
k, l = len(primes), len(primes) - 1
ev = ShardedEvaluator.in_process(N, primes, G)
rng = np.random.default_rng(1)
rand = lambda prefix, nl: np.stack([rng.integers(0, primes[i], size=prefix + (N,), dtype=np.uint64) for i in range(nl)], axis=len(prefix))
ev.upload_relin_key(rand((l, 2), k))
A = ev.upload_ct(rand((3,), l), 2.0 ** 20)
print("relinearize...", flush=True)
R = ev.relinearize(A)
ev.sync()
print("ok", ev.download(R).shape)
