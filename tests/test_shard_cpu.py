"""Limb-sharded execution (tests/shard_harness.py) on CPU: ShardedEvaluator driving
the CPU shard of tests/shard_cpu_backend.py, in one process for G = 2, 3, 4 and across two gloo
ranks — every assembled ciphertext must equal the UNSHARDED oracle's, bit for bit."""
import json
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

from shard_harness import ShardedEvaluator, local_limbs, rows_for
from oracle import pyoracle as po
from shard_cpu_backend import OracleShard

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N, BITS = 1024, [40, 30, 40, 30, 40, 41]


def _setup(G):
    primes = po.coeff_modulus_create(N, BITS)
    o = po.Oracle(N, primes)
    ev = ShardedEvaluator.in_process(N, primes, G, make_shard=lambda s: OracleShard(N, primes, s, G))
    return primes, o, ev


def _rand(rng, primes, prefix, nl):
    return np.stack([rng.integers(0, primes[i], size=prefix + (N,), dtype=np.uint64) for i in range(nl)], axis=len(prefix))


def test_partition_helpers():
    for G in (1, 2, 3, 4, 8):
        for l in range(1, 14):
            owned = sorted(i for s in range(G) for i in local_limbs(l, s, G))
            assert owned == list(range(l))
            assert rows_for(l, G) * G >= l and max(len(local_limbs(l, s, G)) for s in range(G)) == rows_for(l, G)


@pytest.mark.parametrize("G", [2, 3, 4])
def test_sharded_ops_equal_unsharded_oracle(G):
    primes, o, ev = _setup(G)
    k, l = len(primes), len(primes) - 1
    rng = np.random.default_rng(G)
    relin = _rand(rng, primes, (l, 2), k)
    ev.upload_relin_key(relin)
    a2, b2, pt = _rand(rng, primes, (2,), l), _rand(rng, primes, (2,), l), _rand(rng, primes, (), l)
    A, B, P = ev.upload_ct(a2, 2.0 ** 20), ev.upload_ct(b2, 2.0 ** 20), ev.upload_pt(pt, 2.0 ** 20)
    assert np.array_equal(ev.download(A), a2)
    assert np.array_equal(ev.download(ev.add(A, B)), o.add(a2, b2))
    assert np.array_equal(ev.download(ev.sub(A, B)), o.sub(a2, b2))
    assert np.array_equal(ev.download(ev.negate(A)), o.negate(a2))
    assert np.array_equal(ev.download(ev.add_plain(A, P)), o.add_plain(a2, pt))
    assert np.array_equal(ev.download(ev.multiply_plain(A, P)), o.multiply_plain(a2, pt))
    M = ev.multiply(A, B)
    m = o.multiply(a2, b2)
    assert np.array_equal(ev.download(M), m)
    R = ev.relinearize(M)
    r = o.relinearize(m, relin)
    assert np.array_equal(ev.download(R), r), "sharded relinearize differs"
    S = ev.rescale(R, 30)
    s = o.rescale(r)
    assert S.limbs == l - 1 and np.array_equal(ev.download(S), s), "sharded rescale differs"
    assert np.array_equal(ev.download(ev.rescale(M, 30)), o.rescale(m)), "sharded size-3 rescale differs"
    # rotations at this level and one level down (through a mod-switched value)
    for steps in (1, -3, N // 2 - 1):
        elt = ev.galois_elt_from_step(steps)
        assert elt == po.galois_elt_from_step(N, steps)
        gk = _rand(rng, primes, (l, 2), k)
        ev.upload_galois_key(elt, gk)
        assert np.array_equal(ev.download(ev.rotate(A, steps)), o.rotate(a2, steps, gk)), f"sharded rotate({steps}) differs"
        down = ev.mod_switch(ev.mod_switch(A))
        want = o.rotate(o.mod_switch(o.mod_switch(a2)), steps, gk)
        assert down.limbs == l - 2 and np.array_equal(ev.download(ev.rotate(down, steps)), want)
    # down the whole chain: the owner of the special limb changes with the level, shards run out of limbs
    cur, ref = S, s
    while cur.limbs >= 2:
        sq = ev.square(cur) if cur.limbs * 30 > 40 else None
        cur, ref = ev.rescale(cur, 30), o.rescale(ref)
        assert np.array_equal(ev.download(cur), ref)
        if cur.limbs >= 1:
            assert np.array_equal(ev.download(ev.relinearize(ev.multiply(cur, cur)))[:, :, :], o.relinearize(o.multiply(ref, ref), relin))
        del sq


WORKER = textwrap.dedent("""
    import json, os, sys
    sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
    import numpy as np
    from eva_amd.dist import Dist
    from shard_harness import ShardedEvaluator
    from oracle import pyoracle as po
    from shard_cpu_backend import OracleShard
    d = Dist(backend="gloo")
    N, bits = 1024, [40, 30, 40, 30, 41]
    primes = po.coeff_modulus_create(N, bits)
    k, l = len(primes), len(primes) - 1
    ev = ShardedEvaluator.distributed(N, primes, d, make_shard=lambda s: OracleShard(N, primes, s, d.world))
    rng = np.random.default_rng(99)   # the same inputs on every rank
    rand = lambda prefix, nl: np.stack([rng.integers(0, primes[i], size=prefix + (N,), dtype=np.uint64) for i in range(nl)], axis=len(prefix))
    relin, gk = rand((l, 2), k), rand((l, 2), k)
    a2, b2 = rand((2,), l), rand((2,), l)
    ev.upload_relin_key(relin)
    ev.upload_galois_key(ev.galois_elt_from_step(5), gk)
    A, B = ev.upload_ct(a2, 2.0 ** 20), ev.upload_ct(b2, 2.0 ** 20)
    out = ev.rotate(ev.rescale(ev.relinearize(ev.multiply(A, B)), 30), 5)
    got = ev.gather(out, d)
    o = po.Oracle(N, primes)
    want = o.rotate(o.rescale(o.relinearize(o.multiply(a2, b2), relin)), 5, gk)
    if d.rank == 0:
        print("RESULT " + json.dumps({"equal": bool(np.array_equal(got, want)), "limbs": out.limbs, "world": d.world,
                                      "local": [s for s, p in out.parts.items() if p is not None]}))
    d.close()
""") % (ROOT, ROOT)


def test_two_gloo_ranks_one_shard_each(tmp_path):
    """one process per shard, exchange steps through torch.distributed (gloo): multiply -> relinearize
    -> rescale -> rotate, assembled across the ranks, equals the unsharded oracle"""
    script = tmp_path / "shard_worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", "29641", str(script)]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    r = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("RESULT ")][0][len("RESULT "):])
    assert r == {"equal": True, "limbs": 3, "world": 2, "local": [0]}
