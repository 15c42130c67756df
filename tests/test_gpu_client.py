"""The neighbours of execute() on the device: evah_encrypt against an
oracle-built expectation for the same randomness (bit-exact), evah_decrypt_decode against the host
decoder and the encrypted values (FP64 tolerance 1e-6 absolute at scale 2^40: decode is not a
bit-level contract — the reference checks MSE, tests/common.py:34), and the whole flow
encrypt -> execute -> decrypt with both client paths."""
import os

import numpy as np
import pytest

from eva import EvaProgram, Input, Output, evaluate
from eva.ckks import CKKSCompiler
from eva.seal import generate_keys
from eva_amd import backend
from oracle import pyoracle as po

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("cfg", [(1024, [40, 30, 40, 41]), (8192, [60, 40, 60, 60]), (65536, [60, 60, 60])], ids=lambda c: f"N{c[0]}")
def test_device_encrypt_equals_oracle_built_ciphertext(cfg):
    N, bits = cfg
    primes = po.coeff_modulus_create(N, bits)
    k = len(primes)
    o = po.Oracle(N, primes)
    g = backend.Context(N, primes)
    rng = np.random.default_rng(N)
    pk = np.stack([np.stack([rng.integers(0, primes[i], size=N, dtype=np.uint64) for i in range(k)]) for _ in range(2)])
    g.upload_public_key(pk)
    for l in range(1, k - 1 + 1):
        if l + 1 > k:
            continue
        up = l + 1
        small = np.stack([rng.integers(-1, 2, size=N), rng.integers(-20, 21, size=N), rng.integers(-20, 21, size=N)]).astype(np.int8)
        ptd = np.stack([rng.integers(0, primes[i], size=N, dtype=np.uint64) for i in range(l)])
        got = g.encrypt(g.upload_pt(ptd, 2.0 ** 30), small).download()
        c = np.zeros((2, up, N), dtype=np.uint64)
        for i in range(up):
            q = primes[i]
            sm = [o.ntt(i, np.array([int(v) % q for v in small[j]], dtype=np.uint64)).astype(object) for j in range(3)]
            for K in range(2):
                c[K, i] = ((pk[K, i].astype(object) * sm[0] + sm[1 + K]) % q).astype(np.uint64)
        want = o.add_plain(o.rescale(c), ptd)
        assert np.array_equal(got, want), f"device encryption differs from the oracle-built ciphertext (l={l})"


def _flow(n_vec, N, scale):
    prog = EvaProgram('flow', vec_size=n_vec)
    with prog:
        x, y = Input('x'), Input('y')
        Output('z', x * y + x)
        Output('w', x - y)
    prog.set_input_scales(scale)
    prog.set_output_ranges(20)
    compiled, params, sig = CKKSCompiler(config={'warn_vec_size': 'false'}).compile(prog)
    if N:
        params.poly_modulus_degree = N
    return compiled, params, sig


@pytest.mark.parametrize("N", [None, 32768])
def test_decrypt_decode_on_device_matches_host_and_inputs(N):
    compiled, params, sig = _flow(512, N, 40)
    rng = np.random.default_rng(3)
    inputs = {'x': list(rng.uniform(-2, 2, 512)), 'y': list(rng.uniform(-2, 2, 512))}
    ref = evaluate(compiled, inputs)
    results = {}
    for mode in ("1", "0"):
        os.environ["EVA_DEVICE_CLIENT"] = mode
        pub, sec = generate_keys(params, 5)
        enc = pub.encrypt(inputs, sig)
        out = pub.execute(compiled, enc)
        results[mode] = sec.decrypt(out, sig)
        for name in ref:
            err = np.abs(np.array(results[mode][name]) - np.array(ref[name])).max()
            assert err < 1e-4, (mode, name, err)
    os.environ.pop("EVA_DEVICE_CLIENT")
    for name in ref:  # the two client paths decode the (differently randomised) results to the same values
        assert np.abs(np.array(results["1"][name]) - np.array(results["0"][name])).max() < 1e-4


def test_device_decoder_equals_host_decoder_on_one_ciphertext():
    """same ciphertext through both decoders: only FP64 summation order differs"""
    compiled, params, sig = _flow(1024, 16384, 40)
    pub, sec = generate_keys(params, 9)
    rng = np.random.default_rng(4)
    inputs = {'x': list(rng.uniform(-3, 3, 1024)), 'y': list(rng.uniform(-3, 3, 1024))}
    enc = pub.encrypt(inputs, sig)
    out = pub.execute(compiled, enc)
    os.environ["EVA_DEVICE_CLIENT"] = "1"
    pub1, sec1 = generate_keys(params, 9)
    dev = sec1.decrypt(out, sig)
    os.environ["EVA_DEVICE_CLIENT"] = "0"
    pub0, sec0 = generate_keys(params, 9)   # same seed: same secret key
    host = sec0.decrypt(out, sig)
    os.environ.pop("EVA_DEVICE_CLIENT")
    for name in host:
        assert np.abs(np.array(dev[name]) - np.array(host[name])).max() < 1e-6
