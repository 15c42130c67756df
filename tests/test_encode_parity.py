"""Run-time Encode (/root/reference/eva/seal/seal_executor.h:217-243,
303-309) as an oracle-checked op: the product's host encoder (CPU part) and device encoder
(`-m gpu` part, through the C-ABI's evah_pt_encode) must give the plaintext of the oracle's
restatement of SEAL 3.6's CKKSEncoder::encode bit for bit — FP64 special FFT included — over
N = 2^10 .. 2^16 and scales 2^20 .. 2^60, plus scales whose coefficients need two or more words."""
import numpy as np
import pytest

from eva import EvaProgram, Input, Output
from eva.ckks import CKKSCompiler
from eva.seal import generate_keys
from oracle import pyoracle as po

LOGNS = [10, 11, 12, 13, 14, 15, 16]
# (number of values, scale bits, level)
CASES = [(None, 20, 0), (None, 30, 0), (8, 40, 0), (64, 50, 1), (None, 55, 0), (None, 60, 0), (1, 35, 1)]


def _public(logn, bits=(60, 40, 60)):
    prog = EvaProgram('enc', vec_size=8)
    with prog:
        Output('y', Input('x') * 0.5)
    prog.set_input_scales(30)
    prog.set_output_ranges(20)
    compiled, params, sig = CKKSCompiler(config={'warn_vec_size': 'false'}).compile(prog)
    params.poly_modulus_degree = 1 << logn
    params.prime_bits = list(bits)
    pub, _ = generate_keys(params, 2)
    return pub


def _values(rng, N, n_vals):
    n_vals = N // 2 if n_vals is None else n_vals
    return rng.uniform(-3, 3, n_vals)


@pytest.mark.parametrize("logn", LOGNS)
def test_host_encoder_equals_oracle_encoder(logn):
    N = 1 << logn
    pub = _public(logn)
    o = po.Oracle(N, list(pub.primes))
    rng = np.random.default_rng(100 + logn)
    for n_vals, scale_bits, level in CASES:
        vals = _values(rng, N, n_vals)
        limbs = len(pub.primes) - 1 - level
        rep = np.tile(vals, (N // 2) // len(vals))
        want = o.encode(limbs, rep, 2.0 ** scale_bits)
        got = pub._encode(list(vals), scale_bits, level)
        assert got.shape == want.shape
        assert np.array_equal(got, want), f"host encoding differs from the oracle (N=2^{logn}, scale 2^{scale_bits})"


@pytest.mark.parametrize("scale_bits", [70, 100, 130, 170])
def test_host_encoder_multiword_coefficients_equal_oracle(scale_bits):
    """coefficients beyond 64 / 128 bits: SEAL's two-word and multi-precision decomposition paths"""
    logn = 11
    N = 1 << logn
    pub = _public(logn, bits=(60, 60, 60, 60, 60))
    o = po.Oracle(N, list(pub.primes))
    rng = np.random.default_rng(scale_bits)
    vals = rng.uniform(-3, 3, N // 2)
    want = o.encode(4, vals, 2.0 ** scale_bits)
    got = pub._encode(list(vals), scale_bits, 0)
    assert np.array_equal(got, want)


def test_uniform_constant_equals_general_encoder():
    """a constant vector through the general encoder == round(c * scale) in every NTT slot (what the
    executor's uniform-constant shortcut uploads), on the oracle and on the host encoder"""
    logn = 12
    N = 1 << logn
    pub = _public(logn)
    o = po.Oracle(N, list(pub.primes))
    for c, scale_bits in ((0.5, 30), (-1.0 / 9.0, 40), (3.25, 60), (1.0, 20)):
        want = o.encode(2, np.full(N // 2, c), 2.0 ** scale_bits)
        got = pub._encode([c], scale_bits, 0)
        assert np.array_equal(got, want)
        r = int(round(c * 2.0 ** scale_bits))
        for i in range(2):
            assert np.all(want[i] == np.uint64(r % pub.primes[i]))


@pytest.mark.gpu
@pytest.mark.parametrize("logn", LOGNS)
def test_device_encoder_equals_oracle_encoder(logn):
    from eva_amd import backend
    N = 1 << logn
    primes = po.coeff_modulus_create(N, [60, 40, 60])
    o = po.Oracle(N, primes)
    g = backend.Context(N, primes)
    rng = np.random.default_rng(100 + logn)
    for n_vals, scale_bits, level in CASES:
        vals = _values(rng, N, n_vals)
        limbs = len(primes) - 1 - level
        want = o.encode(limbs, np.tile(vals, (N // 2) // len(vals)), 2.0 ** scale_bits)
        got = g.encode_pt(vals, limbs, 2.0 ** scale_bits).download()
        assert np.array_equal(got, want), f"device encoding differs from the oracle (N=2^{logn}, scale 2^{scale_bits})"
