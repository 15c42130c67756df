"""Pins oracle/ (the CPU restatement of SEAL 3.6 used by EVA's executor) against
library-independent known answers: SURVEY.md Appendix B constants and pure-Python big-integer
algebra (tests/pyref.py).  CPU only."""
import random

import numpy as np
import pytest

import pyref
from oracle import pyoracle as po

# SURVEY.md Appendix B — CoeffModulus::Create results (reference call site
# /root/reference/eva/seal/seal.cpp:181-182; KAT shapes from tests/bug_fixes.py:68,
# tests/features.py:129)
APPENDIX_B = [
    (8192, [60, 20, 60, 60], [0xFFFFFFFFFFD8001, 0xFC001, 0xFFFFFFFFFFE8001, 0xFFFFFFFFFFFC001]),
    (8192, [60, 30, 60, 60], [0xFFFFFFFFFFD8001, 0x3FFF4001, 0xFFFFFFFFFFE8001, 0xFFFFFFFFFFFC001]),
    (16384, [60, 20, 60, 60, 60, 60],
     [0xFFFFFFFFFE38001, 0xC0001, 0xFFFFFFFFFF28001, 0xFFFFFFFFFFC0001, 0xFFFFFFFFFFD8001, 0xFFFFFFFFFFE8001]),
    (32768, [60] * 9,
     [0xFFFFFFFFEFE0001, 0xFFFFFFFFF240001, 0xFFFFFFFFF2A0001, 0xFFFFFFFFF330001, 0xFFFFFFFFF550001,
      0xFFFFFFFFF5A0001, 0xFFFFFFFFF6A0001, 0xFFFFFFFFF840001, 0xFFFFFFFFFFC0001]),
    (8192, [60] * 4, [0xFFFFFFFFFFC4001, 0xFFFFFFFFFFD8001, 0xFFFFFFFFFFE8001, 0xFFFFFFFFFFFC001]),
    (65536, [60] * 11,
     [0xFFFFFFFFE740001, 0xFFFFFFFFE7C0001, 0xFFFFFFFFE9E0001, 0xFFFFFFFFECA0001, 0xFFFFFFFFEFE0001,
      0xFFFFFFFFF240001, 0xFFFFFFFFF2A0001, 0xFFFFFFFFF5A0001, 0xFFFFFFFFF6A0001, 0xFFFFFFFFF840001,
      0xFFFFFFFFFFC0001]),
]


@pytest.mark.parametrize("N,bits,expect", APPENDIX_B)
def test_coeff_modulus_create(N, bits, expect):
    got = po.coeff_modulus_create(N, bits)
    assert got == expect
    for p in got:
        assert p % (2 * N) == 1 and po.lib.evo_is_prime(p)


def test_minimal_primitive_root_constants():
    assert po.lib.evo_minimal_primitive_root(8192, 0xFFFFFFFFFFFC001) == 25959043411404
    assert po.lib.evo_minimal_primitive_root(4096, 1073692673) == 236231


def test_minimal_root_is_minimal_bruteforce():
    # tiny case, brute force over all primitive 2N-th roots
    N, q = 8, 97  # 97 = 6*16+1
    psi = po.lib.evo_minimal_primitive_root(N, q)
    prim = [g for g in range(2, q) if pow(g, N, q) == q - 1]
    assert psi == min(prim)


SMALL = [(32, [30, 30, 30]), (64, [40, 25, 40]), (128, [50, 50]), (256, [60, 60, 60])]


def _ctx(N, bits):
    primes = po.coeff_modulus_create(N, bits)
    return po.Oracle(N, primes), primes


@pytest.mark.parametrize("N,bits", SMALL)
def test_ntt_matches_definition(N, bits):
    o, primes = _ctx(N, bits)
    rng = random.Random(N)
    for i, q in enumerate(primes):
        a = [rng.randrange(q) for _ in range(N)]
        got = o.ntt(i, np.array(a, dtype=np.uint64))
        exp = pyref.naive_ntt(a, o.psi(i), q)
        assert [int(x) for x in got] == exp
        back = o.intt(i, got)
        assert [int(x) for x in back] == a


def test_root_power_table_layout():
    N = 64
    o, primes = _ctx(N, [40])
    q, psi = primes[0], o.psi(0)
    rp = o.root_powers(0)
    irp = o.inv_root_powers(0)
    for i in range(N):
        assert int(rp[pyref.bitrev(i, 6)]) == pow(psi, i, q)
        assert int(rp[i]) * int(irp[i]) % q == 1


def test_ntt_roundtrip_large():
    N = 8192
    o, primes = _ctx(N, [60, 20, 60])
    rng = np.random.default_rng(1)
    for i, q in enumerate(primes):
        a = rng.integers(0, q, size=N, dtype=np.uint64)
        assert np.array_equal(o.intt(i, o.ntt(i, a)), a)


@pytest.mark.parametrize("N,bits", SMALL[:3])
def test_dyadic_product_is_negacyclic_convolution(N, bits):
    o, primes = _ctx(N, bits + [bits[0]])  # one extra prime as the (unused) special prime
    l = len(primes) - 1
    rng = random.Random(7)
    # coefficient-form polys -> NTT form ciphertexts
    A = [[[rng.randrange(primes[i]) for _ in range(N)] for i in range(l)] for _ in range(2)]
    B = [[[rng.randrange(primes[i]) for _ in range(N)] for i in range(l)] for _ in range(2)]
    a = np.array([[o.ntt(i, np.array(A[p][i], dtype=np.uint64)) for i in range(l)] for p in range(2)])
    b = np.array([[o.ntt(i, np.array(B[p][i], dtype=np.uint64)) for i in range(l)] for p in range(2)])
    d = o.multiply(a, b)
    s = o.square(a)
    for i in range(l):
        q = primes[i]
        a0b0 = pyref.negacyclic_mul(A[0][i], B[0][i], q)
        a1b1 = pyref.negacyclic_mul(A[1][i], B[1][i], q)
        x = pyref.negacyclic_mul(A[0][i], B[1][i], q)
        y = pyref.negacyclic_mul(A[1][i], B[0][i], q)
        assert [int(v) for v in o.intt(i, d[0, i])] == a0b0
        assert [int(v) for v in o.intt(i, d[1, i])] == [(u + v) % q for u, v in zip(x, y)]
        assert [int(v) for v in o.intt(i, d[2, i])] == a1b1
        aa = pyref.negacyclic_mul(A[0][i], A[1][i], q)
        assert [int(v) for v in o.intt(i, s[0, i])] == pyref.negacyclic_mul(A[0][i], A[0][i], q)
        assert [int(v) for v in o.intt(i, s[1, i])] == [2 * v % q for v in aa]
        assert [int(v) for v in o.intt(i, s[2, i])] == pyref.negacyclic_mul(A[1][i], A[1][i], q)


def _rand_ct(o, primes, size, l, seed):
    rng = np.random.default_rng(seed)
    return np.stack([np.stack([rng.integers(0, primes[i], size=o.N, dtype=np.uint64) for i in range(l)])
                     for _ in range(size)])


def test_elementwise_semantics():
    N = 64
    o, primes = _ctx(N, [40, 30, 40, 40])
    l = 3
    a2 = _rand_ct(o, primes, 2, l, 1)
    b3 = _rand_ct(o, primes, 3, l, 2)
    pt = _rand_ct(o, primes, 1, l, 3)[0]
    a2[0, 0, 0] = 0  # negate(0) = 0
    q = np.array(primes[:l], dtype=object).reshape(1, l, 1)
    A, B, P = a2.astype(object), b3.astype(object), pt.astype(object)
    s = o.add(a2, b3)
    assert s.shape == (3, l, N)
    assert np.array_equal(s[:2].astype(object), (A + B[:2]) % q)
    assert np.array_equal(s[2], b3[2])  # extra poly copied
    d = o.sub(a2, b3)
    assert np.array_equal(d[:2].astype(object), (A - B[:2]) % q)
    assert np.array_equal(d[2:].astype(object), (-B[2:]) % q)  # extra poly of subtrahend negated
    d2 = o.sub(b3, a2)
    assert np.array_equal(d2[2], b3[2])
    assert np.array_equal(o.negate(a2).astype(object), (-A) % q)
    ap = o.add_plain(b3, pt)
    assert np.array_equal(ap[0].astype(object), (B[0] + P) % q[0]) and np.array_equal(ap[1:], b3[1:])
    sp = o.sub_plain(b3, pt)
    assert np.array_equal(sp[0].astype(object), (B[0] - P) % q[0]) and np.array_equal(sp[1:], b3[1:])
    mp = o.multiply_plain(b3, pt)
    assert np.array_equal(mp.astype(object), (B * P.reshape(1, l, N)) % q)
    ms = o.mod_switch(b3)
    assert np.array_equal(ms, b3[:, : l - 1])


@pytest.mark.parametrize("N,bits", [(32, [30, 30, 30, 30]), (64, [40, 20, 40, 40]), (64, [30, 50, 40, 45])])
def test_rescale_is_exact_divide_and_round(N, bits):
    """A.5: rescale == floor((X + q_last/2) / q_last) on the CRT-composed coefficients."""
    o, primes = _ctx(N, bits)
    l = len(primes) - 1
    ct = _rand_ct(o, primes, 3, l, 11)
    out = o.rescale(ct)
    assert out.shape == (3, l - 1, N)
    for p in range(3):
        coeff = [[int(v) for v in o.intt(i, ct[p, i])] for i in range(l)]
        got = [[int(v) for v in o.intt(i, out[p, i])] for i in range(l - 1)]
        for j in range(N):
            X, _ = pyref.crt([coeff[i][j] for i in range(l)], primes[:l])
            Y = pyref.divide_round(X, primes[l - 1])
            for i in range(l - 1):
                assert got[i][j] == Y % primes[i]


def _keygen_py(o, primes, N, l_digits, s_coeff, sprime_coeff, rng):
    """A.10 key-switch key for s' under s, in NTT form: key[J][K][i][N]."""
    k = len(primes)
    P = primes[-1]
    key = np.zeros((l_digits, 2, k, N), dtype=np.uint64)
    for J in range(l_digits):
        e = [rng.randrange(-3, 4) for _ in range(N)]
        for i, q in enumerate(primes):
            a = np.array([rng.randrange(q) for _ in range(N)], dtype=np.uint64)  # NTT form
            s_ntt = o.ntt(i, np.array([v % q for v in s_coeff], dtype=np.uint64))
            e_ntt = o.ntt(i, np.array([v % q for v in e], dtype=np.uint64))
            c0 = [(-(int(a[j]) * int(s_ntt[j]) + int(e_ntt[j]))) % q for j in range(N)]
            if i == J:
                sp_ntt = o.ntt(i, np.array([v % q for v in sprime_coeff], dtype=np.uint64))
                c0 = [(c0[j] + (P % q) * int(sp_ntt[j])) % q for j in range(N)]
            key[J, 0, i] = np.array(c0, dtype=np.uint64)
            key[J, 1, i] = a
    return key


@pytest.mark.parametrize("N,bits", [(32, [30, 30, 30, 31]), (64, [40, 25, 40, 41]), (64, [36, 40, 30, 40])])
def test_switch_key_exact_and_decrypts(N, bits):
    """A.6: (1) bit-exact vs a pure-Python big-int evaluation of the definition;
    (2) <ks(c), (1,s)> = c*s' + small error."""
    o, primes = _ctx(N, bits)
    k = len(primes)
    l = k - 1
    rng = random.Random(5)
    s = [rng.randrange(-1, 2) for _ in range(N)]
    sp = [rng.randrange(-1, 2) for _ in range(N)]
    key = _keygen_py(o, primes, N, l, s, sp, rng)
    for lv in (l, l - 1):  # also one level down (fewer limbs than digits in the key)
        target = _rand_ct(o, primes, 1, lv, 21)[0]
        ct = _rand_ct(o, primes, 2, lv, 22)
        out = o.switch_key(ct, target, key)
        # ---- (1) exact big-int restatement
        P = primes[-1]
        t = [[int(v) for v in o.intt(J, target[J])] for J in range(lv)]
        mods = primes[:lv] + [P]
        kidx = list(range(lv)) + [k - 1]
        prod = [[None] * (lv + 1) for _ in range(2)]
        for I in range(lv + 1):
            q = mods[I]
            for K in range(2):
                acc = [0] * N
                for J in range(lv):
                    kc = [int(v) for v in o.intt(kidx[I], key[J, K, kidx[I]])]
                    tj = [v % q for v in t[J]]
                    pr = pyref.negacyclic_mul(tj, kc, q)
                    acc = [(x + y) % q for x, y in zip(acc, pr)]
                prod[K][I] = acc
        for K in range(2):
            got = [[int(v) for v in o.intt(J, out[K, J])] for J in range(lv)]
            base = [[int(v) for v in o.intt(J, ct[K, J])] for J in range(lv)]
            for j in range(N):
                r = prod[K][lv][j]
                rr = (r + (P >> 1)) % P
                for J in range(lv):
                    q = primes[J]
                    u = (rr % q - (P >> 1) % q) % q
                    exp = (base[J][j] + (prod[K][J][j] - u) * pow(P, -1, q)) % q
                    assert got[J][j] == exp
        # ---- (2) decrypt relation
        delta = o.sub(out, ct)  # = keyswitch(target)
        Q = 1
        for q in primes[:lv]:
            Q *= q
        d0 = [[int(v) for v in o.intt(J, delta[0, J])] for J in range(lv)]
        d1 = [[int(v) for v in o.intt(J, delta[1, J])] for J in range(lv)]
        for J in range(lv):
            q = primes[J]
            lhs = [(x + y) % q for x, y in zip(d0[J], pyref.negacyclic_mul(d1[J], [v % q for v in s], q))]
            rhs = pyref.negacyclic_mul(t[J], [v % q for v in sp], q)
            d0[J] = [(x - y) % q for x, y in zip(lhs, rhs)]
        for j in range(N):
            err, _ = pyref.crt([d0[J][j] for J in range(lv)], primes[:lv])
            assert abs(pyref.centered(err, Q)) < (1 << 24)


def test_relinearize_is_switch_key_on_c2():
    N = 32
    o, primes = _ctx(N, [30, 30, 31])
    rng = random.Random(3)
    key = _rand_ct(o, primes, 2 * 2, 3, 9).reshape(2, 2, 3, N)
    a3 = _rand_ct(o, primes, 3, 2, 10)
    out = o.relinearize(a3, key)
    assert np.array_equal(out, o.switch_key(a3[:2].copy(), a3[2], key))


@pytest.mark.parametrize("N", [16, 64])
def test_galois_table_is_ntt_domain_automorphism(N):
    o, primes = _ctx(N, [30])
    q = primes[0]
    rng = random.Random(N)
    a = [rng.randrange(q) for _ in range(N)]
    a_ntt = o.ntt(0, np.array(a, dtype=np.uint64))
    for steps in (1, 2, -1, N // 2 - 1, -(N // 2 - 1)):
        elt = po.galois_elt_from_step(N, steps)
        assert elt == pow(3, steps if steps > 0 else N // 2 + steps, 2 * N)
        tab = po.galois_table(N, elt)
        lhs = a_ntt[tab]
        rhs = o.ntt(0, np.array(pyref.apply_galois_coeff(a, elt, q), dtype=np.uint64))
        assert np.array_equal(lhs, rhs)
    assert po.galois_elt_from_step(N, N // 2) == 0  # "step count too large"


def test_rotate_zero_is_copy_and_general_matches_definition():
    N = 32
    o, primes = _ctx(N, [30, 30, 31])
    key = _rand_ct(o, primes, 4, 3, 9).reshape(2, 2, 3, N)
    a2 = _rand_ct(o, primes, 2, 2, 10)
    assert np.array_equal(o.rotate(a2, 0, None), a2)
    steps = 3
    tab = po.galois_table(N, po.galois_elt_from_step(N, steps))
    base = np.stack([a2[0][:, tab], np.zeros_like(a2[1])])
    exp = o.switch_key(np.ascontiguousarray(base), np.ascontiguousarray(a2[1][:, tab]), key)
    assert np.array_equal(o.rotate(a2, steps, key), exp)


def test_encode_uniform_constant_is_constant_poly():
    """A.9: a uniform vector encodes to round(c*scale) in every NTT slot."""
    N = 64
    o, primes = _ctx(N, [40, 30, 40])
    for c, scale in ((1.0, 2.0 ** 20), (-2.5, 2.0 ** 30), (0.04, 2.0 ** 35)):
        pt = o.encode(2, np.full(N // 2, c), scale)
        v = round(c * scale)
        for i in range(2):
            assert np.all(pt[i] == np.uint64(v % primes[i]))


def test_encode_decodes_through_canonical_embedding():
    """Evaluate the encoded polynomial at zeta^(3^i) — must return scale*values."""
    N = 32
    rng = np.random.default_rng(0)
    vals = rng.uniform(-2, 2, N // 2)
    scale = 2.0 ** 30
    co = po.encode_coeffs(N, vals, scale)
    pos = 1
    for i in range(N // 2):
        z = np.exp(2j * np.pi * pos / (2 * N))
        ev = sum(co[j] * z ** j for j in range(N))
        assert abs(ev.real / scale - vals[i]) < 1e-6 and abs(ev.imag / scale) < 1e-6
        pos = pos * 3 % (2 * N)
