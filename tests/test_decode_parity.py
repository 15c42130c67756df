"""Decrypt + decode parity (SEALSecret::decrypt, /root/reference/eva/seal/seal.cpp:124-146: decryptor.decrypt
then encoder.decode).  decrypt is integer work (canonical residues): bit-exact.  decode is FP64, so the
contract is the ORDER of operations — the oracle's evo_decode restates SEAL 3.6's
CKKSEncoder::decode_internal (inverse NTT, CRT composition to base-2^64 words, words to one double
least significant first with 1/scale folded in, signed per-word differences above (Q+1)/2,
DWTHandler::transform_to_rev with root_powers_, no FMA) and both product decoders must return the SAME
doubles: the host decoder (CPU test below) and evah_decrypt_decode (-m gpu).  Tolerance: none —
np.array_equal on the float64 bit patterns (libm's cos/sin feed the root table on every side alike)."""
import os

import numpy as np
import pytest

from eva import EvaProgram, Input, Output
from eva.ckks import CKKSCompiler
from eva.seal import generate_keys
from oracle import pyoracle as po


def _bits_equal(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return a.shape == b.shape and np.array_equal(a.view(np.uint64), b.view(np.uint64))


# ---- the oracle itself: algebra that does not depend on any implementation ----------------------

@pytest.mark.parametrize("N,bits", [(1024, [40, 30, 41]), (4096, [60, 20, 60, 60]), (8192, [60, 60, 60, 60])])
def test_oracle_decode_inverts_oracle_encode(N, bits):
    primes = po.coeff_modulus_create(N, bits)
    o = po.Oracle(N, primes)
    rng = np.random.default_rng(N)
    for l in range(1, len(primes)):
        for scale_bits in (20, 30):
            v = rng.uniform(-8, 8, N // 2)
            pt = o.encode(l, v, 2.0 ** scale_bits)
            got = o.decode(pt, 2.0 ** scale_bits)
            assert np.abs(got - v).max() < 2.0 ** -(scale_bits - 12), (l, scale_bits)


def test_oracle_decode_negative_and_multiword_coefficients():
    """coefficients of both signs and beyond 64 bits: the composed integer against Python's big ints"""
    N, bits = 1024, [50, 50, 50, 51]
    primes = po.coeff_modulus_create(N, bits)
    o = po.Oracle(N, primes)
    l = 3
    Q = 1
    for q in primes[:l]:
        Q *= q
    rng = np.random.default_rng(7)
    # a plaintext whose coefficient-form integers are chosen: x_j uniform in (-Q/2, Q/2)
    xs = [int(rng.integers(-2 ** 62, 2 ** 62)) * int(rng.integers(1, 2 ** 62)) * int(rng.integers(1, 2 ** 20)) % Q - Q // 2 for _ in range(N)]
    pt = np.stack([o.ntt(i, np.array([x % primes[i] for x in xs], dtype=np.uint64)) for i in range(l)])
    scale = 2.0 ** 100
    got = o.decode(pt, scale)
    # expected slot values from the exact integers: forward special FFT in numpy complex128 (tolerance, FP order differs)
    logN = 10
    m = 2 * N
    zeta = np.exp(2j * np.pi * np.arange(m) / m)
    pos, want = 1, []
    coeffs = np.array([x / scale for x in xs], dtype=np.float64)
    for i in range(N // 2):
        want.append(np.sum(coeffs * zeta[(pos * np.arange(N)) % m]).real)
        pos = (pos * 3) % m
    want = np.array(want)
    assert np.abs(got - want).max() <= 1e-9 * max(1.0, np.abs(want).max())


# ---- host decoder == oracle decoder (CPU) --------------------------------------------------------

def _flow(n_vec, N, scale, out_range=20):
    prog = EvaProgram('dec', vec_size=n_vec)
    with prog:
        x, y = Input('x'), Input('y')
        Output('z', x * y + x)
        Output('w', x - y)
    prog.set_input_scales(scale)
    prog.set_output_ranges(out_range)
    compiled, params, sig = CKKSCompiler(config={'warn_vec_size': 'false'}).compile(prog)
    if N:
        params.poly_modulus_degree = N
    return compiled, params, sig


@pytest.mark.parametrize("N,scale", [(2048, 30), (8192, 40)])
def test_host_decoder_equals_oracle_decoder(N, scale, monkeypatch):
    """encrypt on the host, decrypt on the host (EVA_DEVICE_CLIENT=0): no GPU anywhere"""
    monkeypatch.setenv("EVA_DEVICE_CLIENT", "0")
    compiled, params, sig = _flow(N // 2, N, scale)
    pub, sec = generate_keys(params, 21)
    rng = np.random.default_rng(N)
    inputs = {'x': list(rng.uniform(-2, 2, N // 2)), 'y': list(rng.uniform(-2, 2, N // 2))}
    enc = pub.encrypt(inputs, sig)
    o = po.Oracle(N, list(pub.primes))
    got = sec.decrypt(enc, sig)
    sk = sec._secret_key_ntt()
    for name in ('x', 'y'):
        kind, size, limbs, sc, data = enc.get(name)
        want = o.decode(o.decrypt(data, sk), sc)[:N // 2]
        assert _bits_equal(got[name], want), f"host decoder differs from the oracle's decode on input {name}"
        assert np.abs(np.array(got[name]) - np.array(inputs[name])).max() < 1e-4


# ---- device decoder == oracle decoder (-m gpu) ----------------------------------------------------

@pytest.mark.gpu
@pytest.mark.parametrize("N,bits,scale_bits", [(1024, [40, 30, 40, 41], 30), (8192, [60, 40, 60, 60], 40),
                                               (32768, [60, 60, 60, 60, 60], 50), (65536, [60, 60, 60], 40)],
                         ids=lambda v: str(v) if isinstance(v, int) else None)
def test_device_decrypt_decode_equals_oracle(N, bits, scale_bits):
    from eva_amd import backend
    primes = po.coeff_modulus_create(N, bits)
    k = len(primes)
    o = po.Oracle(N, primes)
    g = backend.Context(N, primes)
    rng = np.random.default_rng(N + scale_bits)
    # the same ternary secret under every prime
    small = rng.integers(-1, 2, size=N)
    sk = np.stack([o.ntt(i, np.array([int(v) % primes[i] for v in small], dtype=np.uint64)) for i in range(k)])
    g.upload_secret_key(sk)
    for l in range(1, k):
        for size in (2, 3):
            # a ciphertext of a known message: c0 = m - c1 s - c2 s^2 with an encoded m, so decode is meaningful;
            # plus noise-like garbage in the low bits through random c1, c2
            v = rng.uniform(-4, 4, N // 2)
            m = o.encode(l, v, 2.0 ** scale_bits)
            ct = np.stack([np.stack([rng.integers(0, primes[i], size=N, dtype=np.uint64) for i in range(l)]) for _ in range(size)])
            other = ct.copy()
            other[0] = 0
            # c0 := m - (c1 s + c2 s^2): decrypt(other) with c0 = 0 gives c1 s + c2 s^2
            rest = o.decrypt(other, sk)
            ct[0] = np.stack([(m[i].astype(object) - rest[i].astype(object)) % primes[i] for i in range(l)]).astype(np.uint64)
            want_pt = o.decrypt(ct, sk)
            assert np.array_equal(want_pt, m)
            want = o.decode(want_pt, 2.0 ** scale_bits)
            got = g.decrypt_decode(g.upload_ct(ct, 2.0 ** scale_bits), N // 2)
            assert _bits_equal(got, want), f"device decrypt+decode differs from the oracle (N={N}, l={l}, size={size})"
            assert np.abs(got - v).max() < 1e-3
    # a ciphertext of uniformly random residues: every coefficient is a random element of [0, Q) — both signs,
    # all word counts — the integer part of the pipeline at full width
    l = k - 1
    ct = np.stack([np.stack([rng.integers(0, primes[i], size=N, dtype=np.uint64) for i in range(l)]) for _ in range(2)])
    for sb in (scale_bits, sum(bits[:l]) - 8):  # up to just inside SEAL's "scale out of bounds"
        want = o.decode(o.decrypt(ct, sk), 2.0 ** sb)
        got = g.decrypt_decode(g.upload_ct(ct, 2.0 ** sb), N // 2)
        assert _bits_equal(got, want), f"random ciphertext at scale 2^{sb}"
    g.close()


@pytest.mark.gpu
def test_product_decrypt_paths_return_the_oracles_doubles():
    """sec.decrypt on the device (resident handle and host words) and on the host: all three == oracle"""
    compiled, params, sig = _flow(2048, 16384, 40)
    pub, sec = generate_keys(params, 22)
    rng = np.random.default_rng(9)
    inputs = {'x': list(rng.uniform(-3, 3, 2048)), 'y': list(rng.uniform(-3, 3, 2048))}
    enc = pub.encrypt(inputs, sig)
    out = pub.execute(compiled, enc)
    dev_resident = sec.decrypt(out, sig)
    o = po.Oracle(16384, list(pub.primes))
    sk = sec._secret_key_ntt()
    out.to_host(True)
    dev_host_words = sec.decrypt(out, sig)
    os.environ["EVA_DEVICE_CLIENT"] = "0"
    try:
        pub0, sec0 = generate_keys(params, 22)  # same seed: same secret key
        host = sec0.decrypt(out, sig)
    finally:
        os.environ.pop("EVA_DEVICE_CLIENT")
    for name in out.names():
        kind, size, limbs, sc, data = out.get(name)
        want = o.decode(o.decrypt(data, sk), sc)[:2048]
        for label, got in (("device, resident", dev_resident), ("device, host words", dev_host_words), ("host", host)):
            assert _bits_equal(got[name], want), f"{label}: output {name} differs from the oracle's decode"
