"""probe: do host<->device copies on one queue overlap kernels on another?  Queue A runs a long chain of
rotations; queue B uploads / downloads 32 instances (pinned host memory, stream-ordered calls).
Prints A alone, B alone, and both together (wall clock between syncs)."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from eva_amd import backend

from eva_amd.hostref import coeff_modulus_create

N, k, B = 16384, 5, 32
primes = coeff_modulus_create(N, [60] * k)
g = backend.Context(N, primes)
_lib = backend._lib
f = g.fork()
rng = np.random.default_rng(1)
l = k - 1
key = rng.integers(0, 1 << 59, size=(l, 2, k, N), dtype=np.uint64)
steps = [1, 2, 64, 65]
for st in steps:
    g.upload_galois_key(g.galois_elt_from_step(st), key)
a = rng.integers(0, 1 << 59, size=(B, 2, l, N), dtype=np.uint64)
A = g.upload_ct_batch(a, 2.0 ** 20)
words = 2 * l * N
_lib.evah_host_alloc.restype = C.c_void_p
_lib.evah_host_alloc.argtypes = [C.c_size_t]
bufs = []
for b in range(B):
    p = _lib.evah_host_alloc(words * 8)
    arr = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint64)), shape=(words,))
    arr[:] = a[b].reshape(-1)
    bufs.append((p, arr))
ptrs = (C.POINTER(C.c_uint64) * B)(*[C.cast(p, C.POINTER(C.c_uint64)) for p, _ in bufs])

def compute():
    for _ in range(6):
        outs = g.rotate_many(A, steps)
        del outs

def copies(n=3):
    hs = []
    for _ in range(n):
        h = C.c_void_p()
        backend._chk(_lib.evah_ct_upload_instances_async(f.h, B, 2, l, C.c_double(2.0 ** 20), ptrs, C.byref(h)))
        backend._chk(_lib.evah_ct_download_instances_async(f.h, h, ptrs))
        hs.append(h)
    return hs

def timed(fn):
    g.sync(); f.sync()
    t0 = time.perf_counter()
    keep = fn()
    g.sync(); f.sync()
    dt = time.perf_counter() - t0
    for h in (keep or []):
        _lib.evah_ct_free(f.h, h)
    return dt * 1e3

for _ in range(2):
    timed(compute); timed(copies)
ta = min(timed(compute) for _ in range(3))
tb = min(timed(copies) for _ in range(3))
def both():
    hs = copies()
    compute()
    return hs
tab = min(timed(both) for _ in range(3))
print(f"compute alone {ta:.2f} ms, copies alone {tb:.2f} ms ({3 * B * words * 8 * 2 / tb / 1e6:.1f} GB/s both directions), together {tab:.2f} ms (sum {ta + tb:.2f})")
