"""GPU parity: every evaluator entry point of libeva_hip.so (through the C-ABI) vs the CPU oracle,
bit-exact on seeded inputs.  Mirrors the op coverage of the reference's tests/features.py
(binary ops, unary ops, rotations, mixed sizes) at the SEAL-call level
(/root/reference/eva/seal/seal_executor.h:114-243)."""
import numpy as np
import pytest

from eva_amd import backend
from oracle import pyoracle as po

pytestmark = pytest.mark.gpu

# (N, bit sizes incl. special prime last) — small primes exercise the generic Barrett paths
CONFIGS = [
    (1024, [30, 30, 31]),
    (2048, [40, 20, 40, 41]),
    (4096, [60, 20, 60, 60]),
    (8192, [60, 30, 60, 60, 60]),
    (16384, [60, 20, 60, 60, 60, 60]),
    (32768, [60] * 5),
    (65536, [60] * 4),
]


class Env:
    def __init__(self, N, bits):
        self.N = N
        self.primes = po.coeff_modulus_create(N, bits)
        self.k = len(self.primes)
        self.o = po.Oracle(N, self.primes)
        self.g = backend.Context(N, self.primes)
        self.rng = np.random.default_rng(N + len(bits))

    def rand(self, size, l):
        return np.stack([np.stack([self.rng.integers(0, self.primes[i], size=self.N, dtype=np.uint64)
                                   for i in range(l)]) for _ in range(size)])

    def rand_key(self):
        return np.stack([np.stack([np.stack([self.rng.integers(0, self.primes[i], size=self.N, dtype=np.uint64)
                                             for i in range(self.k)]) for _ in range(2)])
                         for _ in range(self.k - 1)])


_envs = {}


def env(cfg):
    key = (cfg[0], tuple(cfg[1]))
    if key not in _envs:
        _envs[key] = Env(*cfg)
    return _envs[key]


@pytest.mark.parametrize("cfg", CONFIGS + [(131072, [60, 60])], ids=lambda c: f"N{c[0]}")
def test_ntt_intt_bit_exact(cfg):
    e = env(cfg) if cfg[0] != 131072 else Env(*cfg)
    for i, q in enumerate(e.primes):
        a = e.rng.integers(0, q, size=e.N, dtype=np.uint64)
        f = e.g.test_ntt(i, a)
        assert np.array_equal(f, e.o.ntt(i, a)), f"forward NTT mismatch prime {i}"
        b = e.g.test_ntt(i, f, inverse=True)
        assert np.array_equal(b, a), f"INTT(NTT(a)) != a prime {i}"
        assert np.array_equal(e.g.test_ntt(i, a, inverse=True), e.o.intt(i, a))


@pytest.mark.parametrize("cfg", CONFIGS, ids=lambda c: f"N{c[0]}")
def test_elementwise_bit_exact(cfg):
    e = env(cfg)
    l = e.k - 1
    a2, b2, b3 = e.rand(2, l), e.rand(2, l), e.rand(3, l)
    a2[0, 0, :7] = 0
    pt = e.rand(1, l)[0]
    A2, B2, B3 = (e.g.upload_ct(x, 2.0 ** 20) for x in (a2, b2, b3))
    PT = e.g.upload_pt(pt, 2.0 ** 20)
    assert np.array_equal(A2.download(), a2)
    assert np.array_equal(PT.download(), pt)
    assert np.array_equal(e.g.add(A2, B2).download(), e.o.add(a2, b2))
    assert np.array_equal(e.g.add(A2, B3).download(), e.o.add(a2, b3))
    assert np.array_equal(e.g.add(B3, A2).download(), e.o.add(b3, a2))
    assert np.array_equal(e.g.sub(A2, B2).download(), e.o.sub(a2, b2))
    assert np.array_equal(e.g.sub(A2, B3).download(), e.o.sub(a2, b3))
    assert np.array_equal(e.g.sub(B3, A2).download(), e.o.sub(b3, a2))
    assert np.array_equal(e.g.negate(B3).download(), e.o.negate(b3))
    assert np.array_equal(e.g.add_plain(B3, PT).download(), e.o.add_plain(b3, pt))
    assert np.array_equal(e.g.sub_plain(A2, PT).download(), e.o.sub_plain(a2, pt))
    m = e.g.multiply(A2, B2)
    assert np.array_equal(m.download(), e.o.multiply(a2, b2))
    assert m.info() == (3, l, 2.0 ** 40)
    assert np.array_equal(e.g.square(A2).download(), e.o.square(a2))
    assert np.array_equal(e.g.multiply_plain(B3, PT).download(), e.o.multiply_plain(b3, pt))
    ms = e.g.mod_switch(B3)
    assert ms.info() == (3, l - 1, 2.0 ** 20)
    assert np.array_equal(ms.download(), e.o.mod_switch(b3))
    # ops on a mod-switched view (poly stride != limbs*N)
    ms2 = e.g.mod_switch(A2)
    assert np.array_equal(e.g.add(ms, ms2).download(), e.o.add(e.o.mod_switch(b3), e.o.mod_switch(a2)))


def test_downloads_of_views_are_the_leading_limbs():
    """A mod-switched handle is a view (poly stride > limbs * N): single and batched handles, one and two levels down,
    download as the leading limbs of every polynomial — one linear copy per polynomial (r6: a 2-D copy into pageable
    memory could return before its last row had arrived), also through the per-instance form."""
    e = env(CONFIGS[4])
    l = e.k - 1
    for size in (2, 3):
        one = e.rand(size, l)
        many = np.stack([e.rand(size, l) for _ in range(3)])
        V, VB = e.g.upload_ct(one, 2.0 ** 20), e.g.upload_ct_batch(many, 2.0 ** 20)
        for drop in (1, 2):
            V, VB = e.g.mod_switch(V), e.g.mod_switch(VB)
            for _ in range(4):
                assert np.array_equal(V.download(), one[:, :l - drop, :])
                assert np.array_equal(VB.download(), many[:, :, :l - drop, :])
            assert np.array_equal(VB.unstack(2).download(), many[2, :, :l - drop, :])


@pytest.mark.parametrize("cfg", CONFIGS, ids=lambda c: f"N{c[0]}")
def test_rescale_bit_exact(cfg):
    e = env(cfg)
    l = e.k - 1
    for size in (2, 3):
        a = e.rand(size, l)
        r = e.g.rescale(e.g.upload_ct(a, 2.0 ** 50), 30)
        assert r.info() == (size, l - 1, 2.0 ** 20)
        assert np.array_equal(r.download(), e.o.rescale(a))


@pytest.mark.parametrize("cfg", CONFIGS, ids=lambda c: f"N{c[0]}")
def test_relinearize_bit_exact(cfg):
    e = env(cfg)
    key = e.rand_key()
    e.g.upload_relin_key(key)
    for l in sorted({e.k - 1, max(1, e.k - 2)}):
        a3 = e.rand(3, l)
        out = e.g.relinearize(e.g.upload_ct(a3, 2.0 ** 30)).download()
        assert np.array_equal(out, e.o.relinearize(a3, key)), f"relinearize mismatch at l={l}"


@pytest.mark.parametrize("cfg", CONFIGS, ids=lambda c: f"N{c[0]}")
def test_relinearize_rescale_fused_bit_exact(cfg):
    """The fused call must equal rescale_to_next(relinearize(x)) exactly."""
    e = env(cfg)
    key = e.rand_key()
    e.g.upload_relin_key(key)
    for l in sorted({e.k - 1, max(2, e.k - 2)}):
        a3 = e.rand(3, l)
        A3 = e.g.upload_ct(a3, 2.0 ** 50)
        out = e.g.relinearize_rescale(A3, 30)
        assert out.info() == (2, l - 1, 2.0 ** 20)
        ref = e.o.rescale(e.o.relinearize(a3, key))
        assert np.array_equal(out.download(), ref), f"fused relin+rescale mismatch at l={l}"
        assert np.array_equal(e.g.rescale(e.g.relinearize(A3), 30).download(), ref)


@pytest.mark.parametrize("cfg", CONFIGS, ids=lambda c: f"N{c[0]}")
def test_multiply_relinearize_rescale_fused_bit_exact(cfg):
    """evah_multiply_relinearize_rescale(_many): the Mul -> Relinearize -> Rescale chain without the
    size-3 product in memory == the oracle's three separate calls; operands are separate
    allocations, one pair shares an operand, one operand is a mod-switched view."""
    e = env(cfg)
    key = e.rand_key()
    e.g.upload_relin_key(key)
    for l in sorted({e.k - 1, max(2, e.k - 2)}):
        hosts = [(e.rand(2, l), e.rand(2, l)) for _ in range(3)]
        A = [e.g.upload_ct(a, 2.0 ** 25) for a, _ in hosts]
        B = [e.g.upload_ct(b, 2.0 ** 25) for _, b in hosts]
        want = [e.o.rescale(e.o.relinearize(e.o.multiply(a, b), key)) for a, b in hosts]
        one = e.g.multiply_relinearize_rescale(A[0], B[0], 30)
        assert one.info() == (2, l - 1, 2.0 ** 20)
        assert np.array_equal(one.download(), want[0]), f"fused multiply+relin+rescale mismatch at l={l}"
        outs = e.g.multiply_relinearize_rescale_many(A + [A[1]], B + [B[2]], 30)
        for o, w in zip(outs, want):
            assert np.array_equal(o.download(), w)
        assert np.array_equal(outs[3].download(), e.o.rescale(e.o.relinearize(e.o.multiply(hosts[1][0], hosts[2][1]), key)))
        if l + 1 <= e.k - 1:  # operands that are mod-switched views of longer ciphertexts
            big = e.rand(2, l + 1)
            V = e.g.mod_switch(e.g.upload_ct(big, 2.0 ** 25))
            got = e.g.multiply_relinearize_rescale(V, B[0], 30).download()
            assert np.array_equal(got, e.o.rescale(e.o.relinearize(e.o.multiply(e.o.mod_switch(big), hosts[0][1]), key)))
    with pytest.raises(backend.EvaHipError, match="size-2"):
        e.g.multiply_relinearize_rescale(e.g.upload_ct(e.rand(3, e.k - 1), 2.0 ** 20), e.g.upload_ct(e.rand(2, e.k - 1), 2.0 ** 20), 30)


# (EVAH_SIDE_STREAM, EVAH_CHAIN_STEP, EVAH_CHAIN_FUSE_BLOCKS): the r6 launch set with / without the side stream, and the
# six-launch chain step (ntt_chain.hip.h) with t_J recomputed per output limb (default) and stored once (0)
CHAIN_MODES = [("1", "0", None), ("0", "0", None), ("0", "1", None), ("0", "1", "0")]


@pytest.mark.parametrize("cfg", CONFIGS, ids=lambda c: f"N{c[0]}")
@pytest.mark.parametrize("mode", CHAIN_MODES, ids=lambda m: f"side{m[0]}-chain{m[1]}-fuse{m[2]}")
def test_multiply_rescale_relinearize_fused_bit_exact(cfg, mode, monkeypatch):
    """r6: evah_multiply_rescale_relinearize(_many) — the Mul -> Rescale -> Relinearize chain of lazy relinearization, the
    size-3 product never in memory — == the oracle's three separate calls; squares (a is b), a shared operand, a
    mod-switched view, every level down to the last key switch (one limb left).  Launch forms: the rescale of d0 / d1 on
    the queue's side stream beside the key switch of d2, the same on one stream, and the six-launch chain step (the
    rescaled d2 formed in coefficient form, d0 / d1 rescaled by the mod-down's forward transform) in both of its shapes."""
    side, chain, fuse = mode
    monkeypatch.setenv("EVAH_SIDE_STREAM", side)
    monkeypatch.setenv("EVAH_CHAIN_STEP", chain)
    if fuse is not None:
        monkeypatch.setenv("EVAH_CHAIN_FUSE_BLOCKS", fuse)
    e = Env(*cfg)
    key = e.rand_key()
    e.g.upload_relin_key(key)
    for l in sorted({e.k - 1, max(2, e.k - 2), 2}):
        hosts = [(e.rand(2, l), e.rand(2, l)) for _ in range(3)]
        A = [e.g.upload_ct(a, 2.0 ** 25) for a, _ in hosts]
        B = [e.g.upload_ct(b, 2.0 ** 25) for _, b in hosts]
        want = [e.o.relinearize(e.o.rescale(e.o.multiply(a, b)), key) for a, b in hosts]
        one = e.g.multiply_rescale_relinearize(A[0], B[0], 30)
        assert one.info() == (2, l - 1, 2.0 ** 20)
        assert np.array_equal(one.download(), want[0]), f"fused multiply+rescale+relinearize mismatch at l={l}"
        sq = e.g.multiply_rescale_relinearize(A[1], A[1], 30)
        assert np.array_equal(sq.download(), e.o.relinearize(e.o.rescale(e.o.square(hosts[1][0])), key)), f"square at l={l}"
        outs = e.g.multiply_rescale_relinearize_many(A + [A[1], B[2]], B + [B[2], B[2]], 30)
        for o, w in zip(outs, want):
            assert np.array_equal(o.download(), w)
        assert np.array_equal(outs[3].download(), e.o.relinearize(e.o.rescale(e.o.multiply(hosts[1][0], hosts[2][1])), key))
        assert np.array_equal(outs[4].download(), e.o.relinearize(e.o.rescale(e.o.square(hosts[2][1])), key))
        # the same words as the three separate calls of this library
        sep = e.g.relinearize(e.g.rescale(e.g.multiply(A[0], B[0]), 30))
        assert np.array_equal(sep.download(), want[0])
        if l + 1 <= e.k - 1:  # operands that are mod-switched views of longer ciphertexts
            big = e.rand(2, l + 1)
            V = e.g.mod_switch(e.g.upload_ct(big, 2.0 ** 25))
            got = e.g.multiply_rescale_relinearize(V, B[0], 30).download()
            assert np.array_equal(got, e.o.relinearize(e.o.rescale(e.o.multiply(e.o.mod_switch(big), hosts[0][1])), key))
    with pytest.raises(backend.EvaHipError, match="size-2"):
        e.g.multiply_rescale_relinearize(e.g.upload_ct(e.rand(3, e.k - 1), 2.0 ** 20), e.g.upload_ct(e.rand(2, e.k - 1), 2.0 ** 20), 30)
    with pytest.raises(backend.EvaHipError, match="end of modulus switching chain"):
        e.g.multiply_rescale_relinearize(e.g.upload_ct(e.rand(2, 1), 2.0 ** 10), e.g.upload_ct(e.rand(2, 1), 2.0 ** 10), 10)
    e.g.close()


@pytest.mark.parametrize("cfg", [c for c in CONFIGS if c[0] >= 4096], ids=lambda c: f"N{c[0]}")
@pytest.mark.parametrize("chain", ["1", "0"])
def test_multiply_rescale_relinearize_batched_handles_bit_exact(cfg, chain, monkeypatch):
    """r6: batched handles through evah_multiply_rescale_relinearize(_many) — every operand holds B instances, the outputs are
    batched handles again (instance b of pair i == the oracle's three calls on instance b of its operands); a square and a
    pair sharing an operand in one call; unequal instance counts are an error"""
    monkeypatch.setenv("EVAH_CHAIN_STEP", chain)
    e = Env(*cfg)
    key = e.rand_key()
    e.g.upload_relin_key(key)
    B, l = 3, e.k - 1
    xs = [np.stack([e.rand(2, l) for _ in range(B)]) for _ in range(3)]
    X = [e.g.upload_ct_batch(x, 2.0 ** 25) for x in xs]
    outs = e.g.multiply_rescale_relinearize_many([X[0], X[1], X[2]], [X[1], X[1], X[0]], 30)
    pairs = [(0, 1), (1, 1), (2, 0)]
    for o, (i, j) in zip(outs, pairs):
        assert o.batch == B and o.info() == (2, l - 1, 2.0 ** 20)
        got = o.download()
        for b in range(B):
            prod = e.o.square(xs[i][b]) if i == j else e.o.multiply(xs[i][b], xs[j][b])
            assert np.array_equal(got[b], e.o.relinearize(e.o.rescale(prod), key)), f"pair {i}x{j} instance {b}"
    one = e.g.multiply_rescale_relinearize(X[0], X[2], 30)
    assert np.array_equal(one.download()[B - 1], e.o.relinearize(e.o.rescale(e.o.multiply(xs[0][B - 1], xs[2][B - 1])), key))
    with pytest.raises(backend.EvaHipError, match="same number of instances"):
        e.g.multiply_rescale_relinearize(X[0], e.g.upload_ct(e.rand(2, l), 2.0 ** 25), 30)
    e.g.close()


@pytest.mark.parametrize("cfg", CONFIGS, ids=lambda c: f"N{c[0]}")
@pytest.mark.parametrize("chain", ["1", "0"])
def test_rescale_relinearize_of_stored_size3_bit_exact(cfg, chain, monkeypatch):
    """r6: evah_rescale_relinearize(_many) — rescale_to_next of a size-3 ciphertext, then relinearize (what lazy relinearization
    leaves after a SUM of products: Sobel's Ix^2 + Iy^2, Harris' response), as the chain step on the stored polynomials — ==
    the oracle's two calls; every level down to the last key switch, a mod-switched view, several handles in one call,
    batched handles; EVAH_CHAIN_STEP=0 is the two separate calls behind the same entry point"""
    monkeypatch.setenv("EVAH_CHAIN_STEP", chain)
    e = Env(*cfg)
    key = e.rand_key()
    e.g.upload_relin_key(key)
    for l in sorted({e.k - 1, max(2, e.k - 2), 2}):
        hosts = [e.rand(3, l) for _ in range(3)]
        H = [e.g.upload_ct(h, 2.0 ** 40) for h in hosts]
        want = [e.o.relinearize(e.o.rescale(h), key) for h in hosts]
        one = e.g.rescale_relinearize(H[0], 30)
        assert one.info() == (2, l - 1, 2.0 ** 10)
        assert np.array_equal(one.download(), want[0]), f"rescale+relinearize mismatch at l={l}"
        outs = e.g.rescale_relinearize_many(H + [H[1]], 30)
        for o, w in zip(outs, want + [want[1]]):
            assert np.array_equal(o.download(), w)
        assert np.array_equal(e.g.relinearize(e.g.rescale(H[2], 30)).download(), want[2])  # the two calls of this library
        if l + 1 <= e.k - 1:
            big = e.rand(3, l + 1)
            V = e.g.mod_switch(e.g.upload_ct(big, 2.0 ** 40))
            assert np.array_equal(e.g.rescale_relinearize(V, 30).download(), e.o.relinearize(e.o.rescale(e.o.mod_switch(big)), key))
        if e.N >= 4096:
            B = 3
            xs = np.stack([e.rand(3, l) for _ in range(B)])
            got = e.g.rescale_relinearize(e.g.upload_ct_batch(xs, 2.0 ** 40), 30)
            assert got.batch == B
            d = got.download()
            for b in range(B):
                assert np.array_equal(d[b], e.o.relinearize(e.o.rescale(xs[b]), key)), f"instance {b} at l={l}"
    with pytest.raises(backend.EvaHipError, match="size-3"):
        e.g.rescale_relinearize(e.g.upload_ct(e.rand(2, e.k - 1), 2.0 ** 40), 30)
    with pytest.raises(backend.EvaHipError, match="end of modulus switching chain"):
        e.g.rescale_relinearize(e.g.upload_ct(e.rand(3, 1), 2.0 ** 10), 10)
    e.g.close()


@pytest.mark.parametrize("chain", ["1", "0"])
def test_multiply_rescale_relinearize_config5_shape_bit_exact(chain, monkeypatch):
    """the shape it was built for: one square at N = 2^16, l = 12 of 13 primes (BASELINE config 5's chain step), eager and
    from inside a captured graph replayed three times; then four products in one call (the chain step's launches are no
    longer latency-sized: stored t, separate inverse passes before the shared forward transform)"""
    monkeypatch.setenv("EVAH_CHAIN_STEP", chain)
    N, bits = 65536, [60] * 13
    e = Env(N, bits)
    key = e.rand_key()
    e.g.upload_relin_key(key)
    a = e.rand(2, 12)
    A = e.g.upload_ct(a, 2.0 ** 40)
    want = e.o.relinearize(e.o.rescale(e.o.square(a)), key)
    assert np.array_equal(e.g.multiply_rescale_relinearize(A, A, 60).download(), want)
    e.g.capture_begin()
    out = e.g.multiply_rescale_relinearize(A, A, 60)
    graph = e.g.capture_end()
    for _ in range(3):
        e.g.graph_launch(graph)
    e.g.sync()
    assert np.array_equal(out.download(), want)
    e.g.graph_free(graph)
    b = e.rand(2, 12)
    B = e.g.upload_ct(b, 2.0 ** 40)
    outs = e.g.multiply_rescale_relinearize_many([A, A, B, B], [A, B, B, A], 60)
    ab = e.o.relinearize(e.o.rescale(e.o.multiply(a, b)), key)
    assert np.array_equal(outs[0].download(), want)
    assert np.array_equal(outs[1].download(), ab)
    assert np.array_equal(outs[2].download(), e.o.relinearize(e.o.rescale(e.o.square(b)), key))
    assert np.array_equal(outs[3].download(), e.o.relinearize(e.o.rescale(e.o.multiply(b, a)), key))
    e.g.close()


@pytest.mark.parametrize("cfg", CONFIGS, ids=lambda c: f"N{c[0]}")
def test_rotate_bit_exact(cfg):
    e = env(cfg)
    l = e.k - 1
    a2 = e.rand(2, l)
    A2 = e.g.upload_ct(a2, 2.0 ** 20)
    assert np.array_equal(e.g.rotate(A2, 0).download(), a2)
    for steps in (1, -1, 5, e.N // 2 - 1, -(e.N // 4)):
        key = e.rand_key()
        elt = e.g.galois_elt_from_step(steps)
        assert elt == po.galois_elt_from_step(e.N, steps)
        e.g.upload_galois_key(elt, key)
        out = e.g.rotate(A2, steps).download()
        assert np.array_equal(out, e.o.rotate(a2, steps, key)), f"rotate({steps}) mismatch"
    # one level down, via a mod-switched view
    ms = e.g.mod_switch(A2)
    if l > 1:
        out = e.g.rotate(ms, -(e.N // 4)).download()
        assert np.array_equal(out, e.o.rotate(e.o.mod_switch(a2), -(e.N // 4), key))


def test_key_switch_at_max_degree_n131072():
    """SEAL's largest poly_modulus_degree (2^17: 9-stage strided pass, 8-stage fused pass) through
    relinearize, the fused relinearize+rescale, rescale and a rotation."""
    e = Env(131072, [60, 60, 60])
    l = e.k - 1
    key = e.rand_key()
    e.g.upload_relin_key(key)
    a3 = e.rand(3, l)
    A3 = e.g.upload_ct(a3, 2.0 ** 50)
    relin = e.o.relinearize(a3, key)
    assert np.array_equal(e.g.relinearize(A3).download(), relin)
    assert np.array_equal(e.g.relinearize_rescale(A3, 30).download(), e.o.rescale(relin))
    assert np.array_equal(e.g.rescale(A3, 30).download(), e.o.rescale(a3))
    gk = e.rand_key()
    e.g.upload_galois_key(e.g.galois_elt_from_step(3), gk)
    a2 = e.rand(2, l)
    assert np.array_equal(e.g.rotate(e.g.upload_ct(a2, 2.0 ** 20), 3).download(), e.o.rotate(a2, 3, gk))


@pytest.mark.parametrize("cfg", CONFIGS, ids=lambda c: f"N{c[0]}")
def test_relinearize_rescale_many_equals_single_calls(cfg):
    """A batch of independent size-3 ciphertexts through one wide launch set == the oracle's
    rescale(relinearize(x)) on each; inputs are separate allocations, one a mod-switched view."""
    e = env(cfg)
    key = e.rand_key()
    e.g.upload_relin_key(key)
    l = e.k - 1
    if l < 3:
        pytest.skip("needs three data limbs for the view case")
    lv = l - 1
    hosts = [e.rand(3, lv) for _ in range(4)]
    wide = e.rand(3, l)
    cts = [e.g.upload_ct(h, 2.0 ** 50) for h in hosts] + [e.g.mod_switch(e.g.upload_ct(wide, 2.0 ** 50))]
    hosts.append(e.o.mod_switch(wide))
    outs = e.g.relinearize_rescale_many(cts, 30)
    for h, o in zip(hosts, outs):
        assert o.info() == (2, lv - 1, 2.0 ** 20)
        assert np.array_equal(o.download(), e.o.rescale(e.o.relinearize(h, key)))
    one = e.g.relinearize_rescale_many(cts[:1], 30)[0]
    assert np.array_equal(one.download(), e.o.rescale(e.o.relinearize(hosts[0], key)))


@pytest.mark.parametrize("cfg", CONFIGS[:6], ids=lambda c: f"N{c[0]}")
def test_weighted_sum_equals_multiply_plain_add_chain(cfg):
    """sum_j ct_j (*) pt_j (+ bare ciphertext terms) in one pass == the oracle's multiply_plain /
    add chain; terms are separate allocations, one a mod-switched view; sizes 2 and 3."""
    e = env(cfg)
    l = e.k - 1
    lv = l - 1 if l >= 2 else l
    for size in (2, 3):
        hs = [e.rand(size, lv) for _ in range(4)]
        ws = [e.rand(1, lv)[0] for _ in range(4)]
        cts = [e.g.upload_ct(h, 2.0 ** 8) for h in hs]
        if l >= 2:
            wide = e.rand(size, l)
            cts.append(e.g.mod_switch(e.g.upload_ct(wide, 2.0 ** 8)))
            hs.append(e.o.mod_switch(wide))
            ws.append(e.rand(1, lv)[0])
        pts = [e.g.upload_pt(w, 2.0 ** 4) for w in ws]
        bare_h = e.rand(size, lv)
        bare = e.g.upload_ct(bare_h, 2.0 ** 12)
        out = e.g.weighted_sum(cts + [bare], pts + [None])
        ref = e.o.multiply_plain(hs[0], ws[0])
        for h, w in zip(hs[1:], ws[1:]):
            ref = e.o.add(ref, e.o.multiply_plain(h, w))
        ref = e.o.add(ref, bare_h)
        assert out.info() == (size, lv, 2.0 ** 12)
        assert np.array_equal(out.download(), ref)
    with pytest.raises(backend.EvaHipError, match="scale mismatch"):
        e.g.weighted_sum(cts[:2], [pts[0], None])


@pytest.mark.parametrize("cfg", CONFIGS[:6], ids=lambda c: f"N{c[0]}")
def test_multiply_many_equals_single_calls(cfg):
    """A batch of independent 2x2 products as one launch == the oracle's multiply on each pair;
    operands are separate allocations, one of them a mod-switched view (larger poly stride)."""
    e = env(cfg)
    l = e.k - 1
    if l < 2:
        pytest.skip("needs two data limbs for the view case")
    lv = l - 1
    ha = [e.rand(2, lv) for _ in range(3)]
    hb = [e.rand(2, lv) for _ in range(3)]
    wide = e.rand(2, l)
    ca = [e.g.upload_ct(h, 2.0 ** 10) for h in ha] + [e.g.mod_switch(e.g.upload_ct(wide, 2.0 ** 10))]
    cb = [e.g.upload_ct(h, 2.0 ** 10) for h in hb] + [ca[0]]
    ha.append(e.o.mod_switch(wide))
    hb.append(ha[0])
    outs = e.g.multiply_many(ca, cb)
    for x, y, o in zip(ha, hb, outs):
        assert o.info() == (3, lv, 2.0 ** 20)
        assert np.array_equal(o.download(), e.o.multiply(x, y))
    with pytest.raises(RuntimeError):
        e.g.multiply_many([outs[0]], [ca[0]])  # size-3 operand


@pytest.mark.parametrize("cfg", CONFIGS[:6], ids=lambda c: f"N{c[0]}")
def test_rotate_many_equals_single_rotations(cfg):
    """Sibling rotations issued as one wide launch set == the individual rotate_vector calls."""
    e = env(cfg)
    l = e.k - 1
    a2 = e.rand(2, l)
    A2 = e.g.upload_ct(a2, 2.0 ** 20)
    steps = [1, 2, 64, 65, 66, 128, 129, 130, -3]
    keys = {}
    for st in steps:
        elt = e.g.galois_elt_from_step(st)
        keys[st] = e.rand_key()
        e.g.upload_galois_key(elt, keys[st])
    outs = e.g.rotate_many(A2, steps)
    for st, o in zip(steps, outs):
        assert o.info() == (2, l, 2.0 ** 20)
        assert np.array_equal(o.download(), e.o.rotate(a2, st, keys[st])), f"rotate_many step {st}"
    del outs
    # from a mod-switched view, and a batch of one
    if l > 1:
        ms = e.g.mod_switch(A2)
        o1 = e.g.rotate_many(ms, [65])[0]
        assert np.array_equal(o1.download(), e.o.rotate(e.o.mod_switch(a2), 65, keys[65]))
    with pytest.raises(backend.EvaHipError, match="zero steps"):
        e.g.rotate_many(A2, [1, 0])


@pytest.mark.parametrize("cfg", [(4096, [40] * 19 + [41]), (131072, [60, 50, 60, 60])], ids=["l19_unfused", "N131072"])
def test_key_switch_extremes_bit_exact(cfg):
    """l > 16 takes the unfused digit-NTT + MAC path (128-bit accumulator headroom); N = 2^17 is
    the largest degree the kernels support."""
    e = Env(*cfg)
    l = e.k - 1
    key = e.rand_key()
    e.g.upload_relin_key(key)
    a3 = e.rand(3, l)
    A3 = e.g.upload_ct(a3, 2.0 ** 35)
    ref = e.o.relinearize(a3, key)
    assert np.array_equal(e.g.relinearize(A3).download(), ref)
    assert np.array_equal(e.g.relinearize_rescale(A3, 20).download(), e.o.rescale(ref))
    gk = e.rand_key()
    e.g.upload_galois_key(e.g.galois_elt_from_step(7), gk)
    a2 = e.rand(2, l)
    A2 = e.g.upload_ct(a2, 2.0 ** 20)
    want = e.o.rotate(a2, 7, gk)
    assert np.array_equal(e.g.rotate(A2, 7).download(), want)
    assert np.array_equal(e.g.rotate_many(A2, [7])[0].download(), want)


def test_op_triple_metric_config_bit_exact():
    """BASELINE metric unit: multiply + relinearize + rescale at N=2^16, L=10."""
    N, bits = 65536, [60] * 11
    e = Env(N, bits)
    l = 10
    key = e.rand_key()
    e.g.upload_relin_key(key)
    a, b = e.rand(2, l), e.rand(2, l)
    A, B = e.g.upload_ct(a, 2.0 ** 40), e.g.upload_ct(b, 2.0 ** 40)
    out = e.g.rescale(e.g.relinearize(e.g.multiply(A, B)), 60)
    assert out.info() == (2, l - 1, 2.0 ** 20)
    ref = e.o.op_triple(a, b, key)
    assert np.array_equal(out.download(), ref)
    fused = e.g.relinearize_rescale(e.g.multiply(A, B), 60)
    assert fused.info() == (2, l - 1, 2.0 ** 20) and np.array_equal(fused.download(), ref)
    # what bench.py issues: the fully fused form, batched
    a2, b2 = e.rand(2, l), e.rand(2, l)
    outs = e.g.multiply_relinearize_rescale_many([A, e.g.upload_ct(a2, 2.0 ** 40)], [B, e.g.upload_ct(b2, 2.0 ** 40)], 60)
    assert outs[0].info() == (2, l - 1, 2.0 ** 20) and np.array_equal(outs[0].download(), ref)
    assert np.array_equal(outs[1].download(), e.o.op_triple(a2, b2, key))


def test_error_behaviour_matches_reference_preconditions():
    """SEAL throws on level/scale mismatch, missing keys, end of chain."""
    e = env(CONFIGS[0])
    l = e.k - 1
    a = e.g.upload_ct(e.rand(2, l), 2.0 ** 20)
    b = e.g.upload_ct(e.rand(2, l), 2.0 ** 21)
    with pytest.raises(backend.EvaHipError, match="scale mismatch"):
        e.g.add(a, b)
    with pytest.raises(backend.EvaHipError, match="parameter mismatch"):
        e.g.add(a, e.g.mod_switch(a))
    with pytest.raises(backend.EvaHipError, match="scale out of bounds"):
        big = e.g.upload_ct(e.rand(2, l), 2.0 ** 40)
        e.g.multiply(big, big)
    with pytest.raises(backend.EvaHipError, match="Galois key not present"):
        e.g.rotate(a, 3)
    with pytest.raises(backend.EvaHipError, match="step count too large"):
        e.g.rotate(a, e.N // 2)
    one = e.g.mod_switch(a) if l == 2 else a
    while one.limbs > 1:
        one = e.g.mod_switch(one)
    with pytest.raises(backend.EvaHipError, match="end of modulus switching chain"):
        e.g.rescale(one, 10)


@pytest.mark.parametrize("logn", [10, 12, 13, 15, 16])
def test_device_encoder_equals_host_encoder(logn):
    """evah_pt_encode (FP64 special FFT, rounding, residues, NTT on the device) gives the plaintext
    of the product's host encoder (the one the DAG-level parity tests feed the oracle with), bit
    for bit; the oracle's own encoder is pinned separately through the canonical embedding
    (tests/test_oracle_kat.py)."""
    from eva import EvaProgram, Input, Output
    from eva.ckks import CKKSCompiler
    from eva.seal import generate_keys
    N = 1 << logn
    prog = EvaProgram('enc', vec_size=8)
    with prog:
        Output('y', Input('x') * 0.5)
    prog.set_input_scales(30)
    prog.set_output_ranges(20)
    compiled, params, sig = CKKSCompiler(config={'warn_vec_size': 'false'}).compile(prog)
    params.poly_modulus_degree = N
    params.prime_bits = [60, 40, 60]
    pub, sec = generate_keys(params, 2)
    g = backend.Context(N, list(pub.primes))
    rng = np.random.default_rng(logn)
    for n_vals, scale_bits, level in ((N // 2, 30, 0), (8, 40, 0), (1, 20, 1), (64, 55, 0)):
        vals = rng.uniform(-3, 3, n_vals)
        limbs = len(pub.primes) - 1 - level
        host = pub._encode(list(vals) * ((N // 2) // n_vals), scale_bits, level)
        dev = g.encode_pt(vals, limbs, 2.0 ** scale_bits).download()
        assert dev.shape == host.shape
        assert np.array_equal(dev, host), f"device encoding differs (N=2^{logn}, {n_vals} values, scale 2^{scale_bits})"


@pytest.mark.parametrize("cfg", CONFIGS[:6], ids=lambda c: f"N{c[0]}")
def test_level_batched_forms_equal_single_calls(cfg):
    """evah_rescale_many / evah_relinearize_many / evah_rotate_pairs: independent nodes of one DAG
    level as one launch set == the oracle's result for each; operands are separate allocations,
    one a mod-switched view."""
    e = Env(*cfg)
    l = e.k - 1
    if l < 2:
        pytest.skip("needs two data limbs")
    lv = l - 1
    key = e.rand_key()
    e.g.upload_relin_key(key)
    for size in (2, 3):
        hs = [e.rand(size, lv) for _ in range(3)]
        wide = e.rand(size, l)
        cts = [e.g.upload_ct(h, 2.0 ** 12) for h in hs] + [e.g.mod_switch(e.g.upload_ct(wide, 2.0 ** 12))]
        hs.append(e.o.mod_switch(wide))
        if lv >= 2:
            outs = e.g.rescale_many(cts, 5)
            for h, o in zip(hs, outs):
                assert o.info() == (size, lv - 1, 2.0 ** 7)
                assert np.array_equal(o.download(), e.o.rescale(h))
        if size == 3:
            outs = e.g.relinearize_many(cts)
            for h, o in zip(hs, outs):
                assert o.info() == (2, lv, 2.0 ** 12)
                assert np.array_equal(o.download(), e.o.relinearize(h, key))
    steps = [1, -2, 1, 7]
    keys = {}
    for s in set(steps):
        keys[s] = e.rand_key()
        e.g.upload_galois_key(e.g.galois_elt_from_step(s), keys[s])
    hs = [e.rand(2, lv) for _ in range(3)]
    wide = e.rand(2, l)
    cts = [e.g.upload_ct(h, 2.0 ** 12) for h in hs] + [e.g.mod_switch(e.g.upload_ct(wide, 2.0 ** 12))]
    hs.append(e.o.mod_switch(wide))
    outs = e.g.rotate_pairs(cts, steps)
    for h, s, o in zip(hs, steps, outs):
        assert np.array_equal(o.download(), e.o.rotate(h, s, keys[s]))
    with pytest.raises(backend.EvaHipError, match="zero steps"):
        e.g.rotate_pairs(cts[:1], [0])


def test_more_than_16_limbs():
    """More than 16 data limbs (possible at N >= 2^15 / 2^16 within the security tables): 16 lazy
    products fill the fused kernel's 128-bit accumulators, so it folds them every 16 digits."""
    e = Env(2048, [30] * 17 + [31, 31])   # 19 primes: 18 data limbs + the special prime
    l = e.k - 1
    assert l == 18
    key = e.rand_key()
    e.g.upload_relin_key(key)
    a3 = e.rand(3, l)
    A3 = e.g.upload_ct(a3, 2.0 ** 20)
    relin = e.o.relinearize(a3, key)
    assert np.array_equal(e.g.relinearize(A3).download(), relin)
    assert np.array_equal(e.g.relinearize_rescale(A3, 10).download(), e.o.rescale(relin))
    gk = e.rand_key()
    e.g.upload_galois_key(e.g.galois_elt_from_step(-5), gk)
    a2 = e.rand(2, l)
    assert np.array_equal(e.g.rotate(e.g.upload_ct(a2, 2.0 ** 20), -5).download(), e.o.rotate(a2, -5, gk))
    # and one level down, where l = 17 still exceeds 16
    ms = e.g.mod_switch(A3)
    assert np.array_equal(e.g.relinearize(ms).download(), e.o.relinearize(e.o.mod_switch(a3), key))
    # the chain step (Mul -> Rescale -> Relinearize) with 17 digits: the 128-bit sums fold once, then take (P L^-1) d_K
    b2 = e.rand(2, l)
    got = e.g.multiply_rescale_relinearize(e.g.upload_ct(a2, 2.0 ** 10), e.g.upload_ct(b2, 2.0 ** 10), 10)
    assert np.array_equal(got.download(), e.o.relinearize(e.o.rescale(e.o.multiply(a2, b2)), key))


@pytest.mark.parametrize("cfg", [CONFIGS[1], CONFIGS[3], CONFIGS[5]], ids=lambda c: f"N{c[0]}")
def test_multiply_plain_many_and_squares_as_products(cfg):
    """evah_multiply_plain_many == the multiply_plain calls it stands for (sizes 2 and 3, one a
    mod-switched view); a square issued as the product (a, a) through evah_multiply_many has the
    residues of evaluator.square (seal_executor.h:162)."""
    e = env(cfg)
    l = e.k - 1
    for size in (2, 3):
        cts = [e.rand(size, l) for _ in range(5)]
        pts = [e.rand(1, l)[0] for _ in range(5)]
        hc = [e.g.upload_ct(x, 2.0 ** 20) for x in cts]
        hp = [e.g.upload_pt(p, 2.0 ** 10) for p in pts]
        outs = e.g.multiply_plain_many(hc, hp)
        for x, p, o in zip(cts, pts, outs):
            assert o.info() == (size, l, 2.0 ** 30)
            assert np.array_equal(o.download(), e.o.multiply_plain(x, p))
    if l > 1:
        big = e.rand(2, l)
        view = e.g.mod_switch(e.g.upload_ct(big, 2.0 ** 20))
        other = e.rand(2, l - 1)
        ptl = [e.rand(1, l - 1)[0] for _ in range(2)]
        outs = e.g.multiply_plain_many([view, e.g.upload_ct(other, 2.0 ** 20)], [e.g.upload_pt(p, 2.0 ** 10) for p in ptl])
        assert np.array_equal(outs[0].download(), e.o.multiply_plain(e.o.mod_switch(big), ptl[0]))
        assert np.array_equal(outs[1].download(), e.o.multiply_plain(other, ptl[1]))
        with pytest.raises(backend.EvaHipError, match="mismatch"):
            e.g.multiply_plain_many([view], [e.g.upload_pt(e.rand(1, l)[0], 2.0 ** 10)])
    a, b = e.rand(2, l), e.rand(2, l)
    A, B = e.g.upload_ct(a, 2.0 ** 20), e.g.upload_ct(b, 2.0 ** 20)
    outs = e.g.multiply_many([A, A, B], [A, B, B])
    assert np.array_equal(outs[0].download(), e.o.square(a))
    assert np.array_equal(outs[1].download(), e.o.multiply(a, b))
    assert np.array_equal(outs[2].download(), e.o.square(b))


def _is_prime(n):
    if n < 2:
        return False
    for p in (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37):
        if n % p == 0:
            return n == p
    d, s = n - 1, 0
    while d % 2 == 0:
        d //= 2
        s += 1
    for a in (2, 325, 9375, 28178, 450775, 9780504, 1795265022):  # deterministic for n < 2^64
        a %= n
        if a == 0:
            continue
        x = pow(a, d, n)
        if x in (1, n - 1):
            continue
        for _ in range(s - 1):
            x = x * x % n
            if x == n - 1:
                break
        else:
            return False
    return True


def _ntt_prime_below(bound, N):
    q = bound - (bound % (2 * N)) + 1
    while q >= bound or not _is_prime(q):
        q -= 2 * N
    return q


def test_primes_of_top_bit_width_but_far_from_the_power_of_two():
    """evah_ctx_create accepts a caller's primes, not only CoeffModulus::Create's.  A 33- or 34-bit prime far below 2^b
    (q = 2^b - c with c close to 2^32) has c < 2^32 yet breaks the bound the top-bit butterflies rest on
    (q + 16 c + 12 q < 16 q): such primes must take the compare-and-subtract path.  Transforms and a whole key switch
    (relinearize, rescale, rotation) against the oracle, next to primes of the ordinary shape in the same chain."""
    N = 2048
    far33 = _ntt_prime_below(int(1.2 * 2 ** 32), N)   # 33 bits, c = 2^33 - q ~ 0.8 * 2^32
    far34 = _ntt_prime_below(int(3.05 * 2 ** 32), N)  # 34 bits, c ~ 0.95 * 2^32
    near = po.coeff_modulus_create(N, [60, 60])
    primes = [far33, near[0], far34, near[1]]
    assert far33.bit_length() == 33 and (1 << 33) - far33 < 2 ** 32 and far34.bit_length() == 34 and (1 << 34) - far34 < 2 ** 32
    g = backend.Context(N, primes)
    o = po.Oracle(N, primes)
    rng = np.random.default_rng(99)
    k, l = len(primes), len(primes) - 1
    for i, q in enumerate(primes):
        for a in (rng.integers(0, q, size=N, dtype=np.uint64), np.full(N, q - 1, dtype=np.uint64)):
            f = g.test_ntt(i, a)
            assert np.array_equal(f, o.ntt(i, a)), f"forward NTT, prime {i} ({q})"
            assert np.array_equal(g.test_ntt(i, f, inverse=True), a), f"inverse NTT, prime {i} ({q})"
    rand = lambda prefix, nl: np.stack([rng.integers(0, primes[i], size=prefix + (N,), dtype=np.uint64)  # noqa: E731
                                        for i in range(nl)], axis=len(prefix))
    rk, gk = rand((l, 2), k), rand((l, 2), k)
    g.upload_relin_key(rk)
    g.upload_galois_key(g.galois_elt_from_step(5), gk)
    a, b = rand((2,), l), rand((2,), l)
    A, B = g.upload_ct(a, 2.0 ** 20), g.upload_ct(b, 2.0 ** 20)
    m = o.multiply(a, b)
    assert np.array_equal(g.multiply(A, B).download(), m)
    r = o.relinearize(m, rk)
    R = g.relinearize(g.multiply(A, B))
    assert np.array_equal(R.download(), r)
    assert np.array_equal(g.rescale(R, 34).download(), o.rescale(r))
    assert np.array_equal(g.rotate(A, 5).download(), o.rotate(a, 5, gk))
    assert np.array_equal(g.multiply_relinearize_rescale_many([A, A], [B, B], 34)[1].download(), o.op_triple(a, b, rk))
    assert np.array_equal(g.multiply_rescale_relinearize(A, B, 34).download(), o.relinearize(o.rescale(m), rk))
    g.close()
    # a 60-bit prime far below 2^60 (c >= 2^32: no top-bit shape, too large for reduction-free radix-2^30 rows): the whole
    # context takes the 128-bit inner products, the chain step included
    far60 = _ntt_prime_below(int(0.7 * 2 ** 60), N)
    primes = [near[0], far60, _ntt_prime_below(near[0] - 1, N), near[1]]
    g, o = backend.Context(N, primes), po.Oracle(N, primes)
    k, l = len(primes), len(primes) - 1
    rk = rand((l, 2), k)
    g.upload_relin_key(rk)
    for lv in (l, l - 1):
        a, b = rand((2,), lv), rand((2,), lv)
        A, B = g.upload_ct(a, 2.0 ** 20), g.upload_ct(b, 2.0 ** 20)
        assert np.array_equal(g.multiply_rescale_relinearize(A, B, 60).download(), o.relinearize(o.rescale(o.multiply(a, b)), rk))
        assert np.array_equal(g.multiply_relinearize_rescale(A, B, 60).download(), o.op_triple(a, b, rk))
    g.close()
