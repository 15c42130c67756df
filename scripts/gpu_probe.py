"""Quick on-GPU timing probe of the evaluator ops (HIP events on the context stream)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from eva_amd import backend
from eva_amd.hostref import coeff_modulus_create


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
    k = int(sys.argv[2]) if len(sys.argv) > 2 else 11
    primes = coeff_modulus_create(N, [60] * k)
    t0 = time.time()
    g = backend.Context(N, primes)
    print(f"ctx create {time.time()-t0:.2f}s  N={N} k={k}")
    l = k - 1
    rng = np.random.default_rng(0)
    def rand(shape_prefix, nl):
        return np.stack([rng.integers(0, primes[i], size=shape_prefix + (N,), dtype=np.uint64) for i in range(nl)], axis=len(shape_prefix))
    key = rand((l, 2), k)
    g.upload_relin_key(key)
    a = g.upload_ct(rand((2,), l), 2.0**40)
    b = g.upload_ct(rand((2,), l), 2.0**40)
    def timeit(name, fn, reps=20, bytes_=None):
        for _ in range(3): fn()
        g.sync()
        g.timer_start()
        for _ in range(reps): fn()
        ms = g.timer_stop() / reps
        extra = f"  {bytes_/ms/1e6:.1f} GB/s (algorithmic)" if bytes_ else ""
        print(f"{name:14s} {ms*1000:9.1f} us{extra}")
        return ms
    P = l * N * 8
    m3 = g.multiply(a, b)
    r2 = g.relinearize(m3)
    timeit("add", lambda: g.add(a, b), bytes_=6 * P)
    timeit("multiply", lambda: g.multiply(a, b), bytes_=7 * P)
    timeit("square", lambda: g.square(a), bytes_=5 * P)
    timeit("relinearize", lambda: g.relinearize(m3), bytes_=5 * P + 2 * l * (l + 1) * N * 8)
    timeit("rescale", lambda: g.rescale(r2, 60), bytes_=2 * P + 2 * (l - 1) * N * 8)
    tb = 7 * P + 5 * P + 2 * l * (l + 1) * N * 8 + 2 * P + 2 * (l - 1) * N * 8
    ms = timeit("op_triple", lambda: g.rescale(g.relinearize(g.multiply(a, b)), 60), bytes_=tb)
    print(f"op-triples/s: {1000.0/ms:.1f}")
    x = rng.integers(0, primes[0], size=N, dtype=np.uint64)
    print("mem in_use/cached MB:", [v / 1e6 for v in g.mem_info()])


if __name__ == "__main__":
    main()
