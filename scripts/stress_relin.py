"""Stress of single-context calls under contention: P processes, each looping random parameter sets (as tests/test_gpu_op_fuzz.py
draws them), relinearize / rotate / weighted_sum against the oracle; a mismatch prints WHERE the words differ and whether the
same call repeated on the same handles gives the oracle's words.   python scripts/stress_relin.py [procs] [iterations] [seed0]"""
import multiprocessing as mp
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def describe(tag, got, want):
    d = got != want
    idx = d.nonzero()
    rows = sorted(set(zip(idx[0].tolist(), idx[1].tolist())))
    out = [f"{tag}: {int(d.sum())} of {d.size} words differ, rows (poly, limb) {rows}"]
    for (p, i) in rows[:6]:
        n = d[p, i].nonzero()[0]
        out.append(f"   row ({p},{i}): {n.size} words, n in [{n.min()}, {n.max()}], first {n[:8].tolist()}, blocks of 256: {sorted(set((n // 256).tolist()))[:12]}")
    return "\n".join(out)


def worker(w, iters, seed0, q):
    import numpy as np
    from test_gpu_parity import Env
    bad = 0
    for it in range(iters):
        seed = seed0 + w * 100000 + it
        rng = random.Random(seed)
        logn = rng.randint(13, 16)
        N = 1 << logn
        k = rng.randint(2, 7)
        bits = [max(rng.choice([20, 25, 30, 36, 40, 45, 50, 55, 58, 60]), logn + 8) for _ in range(k)]
        e = Env(N, bits)
        e.rng = np.random.default_rng(seed)
        l_top = k - 1
        drop = rng.randint(0, max(0, l_top - 1))
        l = l_top - drop

        def up(h, scale=2.0 ** 8):
            ct = e.g.upload_ct(h, scale)
            for _ in range(drop):
                ct = e.g.mod_switch(ct)
            return ct
        a3f = e.rand(3, l_top)
        a3 = a3f[:, :l, :].copy()
        a2f = e.rand(2, l_top)
        a2 = a2f[:, :l, :].copy()
        A3, A2 = up(a3f), up(a2f)
        key = e.rand_key()
        e.g.upload_relin_key(key)
        relin = e.o.relinearize(a3, key)
        got = e.g.relinearize(A3).download()
        if not np.array_equal(got, relin):
            bad += 1
            again = e.g.relinearize(A3).download()
            src = A3.download()
            q.put(f"[w{w} seed {seed} N=2^{logn} bits {bits} l={l}] " + describe("relinearize", got, relin) +
                  f"\n   repeated call equals oracle: {np.array_equal(again, relin)}; source intact: {np.array_equal(src, a3)}")
        steps = rng.choice([1, -1, 3, -7, N // 4, -(N // 2 - 1)])
        gk = e.rand_key()
        e.g.upload_galois_key(e.g.galois_elt_from_step(steps), gk)
        rot = e.o.rotate(a2, steps, gk)
        got = e.g.rotate(A2, steps).download()
        if not np.array_equal(got, rot):
            bad += 1
            again = e.g.rotate(A2, steps).download()
            q.put(f"[w{w} seed {seed} N=2^{logn} bits {bits} l={l} step {steps}] " + describe("rotate", got, rot) +
                  f"\n   repeated call equals oracle: {np.array_equal(again, rot)}")
        pt = e.rand(1, l_top)[0][:l].copy()
        PT = e.g.upload_pt(pt, 2.0 ** 8)
        ws_ref = e.o.add(e.o.multiply_plain(a2, pt), e.o.multiply_plain(a2, pt))
        got = e.g.weighted_sum([A2, A2], [PT, PT]).download()
        if not np.array_equal(got, ws_ref):
            bad += 1
            again = e.g.weighted_sum([A2, A2], [PT, PT]).download()
            q.put(f"[w{w} seed {seed} N=2^{logn} bits {bits} l={l}] " + describe("weighted_sum", got, ws_ref) +
                  f"\n   repeated call equals oracle: {np.array_equal(again, ws_ref)}")
    q.put(f"worker {w}: {iters} iterations, {bad} mismatches")


if __name__ == "__main__":
    procs = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    seed0 = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=worker, args=(w, iters, seed0, q)) for w in range(procs)]
    for p in ps:
        p.start()
    done = 0
    while done < procs:
        m = q.get()
        print(m, flush=True)
        if m.startswith("worker "):
            done += 1
    for p in ps:
        p.join()
