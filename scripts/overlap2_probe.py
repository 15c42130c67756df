"""Do a single-ciphertext relinearize (l = 11) and a size-3 rescale (l = 12) at N = 2^16 overlap when they are issued on two
queues?  (r6: the question behind splitting config 5's rescale -> relinearize step over two streams.)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa: F401
from eva_amd import backend
from eva_amd.hostref import coeff_modulus_create

N = 1 << 16
primes = coeff_modulus_create(N, [60] * 13)
k = len(primes); L = k - 1
rng = np.random.default_rng(1)
def rand(prefix, nl):
    return np.stack([rng.integers(0, primes[i], size=prefix + (N,), dtype=np.uint64) for i in range(nl)], axis=len(prefix))
g = backend.Context(N, primes)
q1 = g.fork()
g.upload_relin_key(rand((L, 2), k))
X = g.upload_ct(rand((3,), 11), 2.0 ** 40)
Y = g.upload_ct(rand((3,), 12), 2.0 ** 40)
Y2 = g.upload_ct(rand((2,), 12), 2.0 ** 40)
def timed(fn, reps=200):
    for _ in range(10): fn()
    g.sync(); q1.sync()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    g.sync(); q1.sync()
    return (time.perf_counter() - t0) / reps * 1e6
def relin(q=g): q.relinearize(X).free()
def resc(q=g): q.rescale(Y, 60).free()
def resc2(q=g): q.rescale(Y2, 60).free()
print("relinearize l=11 alone          %.1f us" % timed(lambda: relin()))
print("rescale size 3 l=12 alone       %.1f us" % timed(lambda: resc()))
print("rescale size 2 l=12 alone       %.1f us" % timed(lambda: resc2()))
print("both, one queue                 %.1f us" % timed(lambda: (relin(), resc())))
print("both, two queues                %.1f us" % timed(lambda: (relin(g), resc(q1))))
print("relin + rescale size 2, 1 queue %.1f us" % timed(lambda: (relin(), resc2())))
print("relin + rescale size 2, 2 queues %.1f us" % timed(lambda: (relin(g), resc2(q1))))
