"""Committed fixtures of tests/golden/ (made by tests/golden/make_golden.py).

CPU: the oracle reproduces every stored output and the reference-derived constants; the compiler
reproduces the `prime_bits` vectors the reference's own tests assert.
GPU: every evaluator entry point of libeva_hip.so reproduces the stored outputs bit for bit."""
import json
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
N = 1024


@pytest.fixture(scope="module")
def vec():
    return dict(np.load(os.path.join(GOLD, "ops_n1024.npz")))


@pytest.fixture(scope="module")
def consts():
    return json.load(open(os.path.join(GOLD, "kat_constants.json")))


def _ops(o, v, rotate_key):
    """(name, result) for every stored output, computed by `o` (oracle or GPU adapter)"""
    a2, b2, a3, pt, rk = v["a2"], v["b2"], v["a3"], v["pt"], v["relin_key"]
    yield "out_add", o.add(a2, b2)
    yield "out_add_32", o.add(a3, b2)
    yield "out_sub", o.sub(a2, b2)
    yield "out_sub_23", o.sub(a2, a3)
    yield "out_negate", o.negate(a3)
    yield "out_add_plain", o.add_plain(a2, pt)
    yield "out_sub_plain", o.sub_plain(a2, pt)
    yield "out_multiply", o.multiply(a2, b2)
    yield "out_square", o.square(a2)
    yield "out_multiply_plain", o.multiply_plain(a3, pt)
    yield "out_relinearize", o.relinearize(a3, rk)
    yield "out_rescale", o.rescale(a2)
    yield "out_rescale3", o.rescale(a3)
    yield "out_relin_rescale", o.rescale(o.relinearize(a3, rk))
    yield "out_mod_switch", o.mod_switch(a3)
    for s in v["rot_steps"]:
        yield f"out_rotate_{int(s)}", o.rotate(a2, int(s), rotate_key(int(s)))
    yield "out_triple", o.op_triple(a2, b2, rk)


def test_oracle_reproduces_golden_vectors(vec):
    from oracle import pyoracle as po
    primes = [int(p) for p in vec["primes"]]
    assert primes == po.coeff_modulus_create(N, [60, 40, 60])
    o = po.Oracle(N, primes)
    assert np.array_equal(o.ntt(0, vec["poly"]), vec["out_ntt0"])
    assert np.array_equal(o.intt(0, vec["poly"]), vec["out_intt0"])
    n = 0
    for name, got in _ops(o, vec, lambda s: vec[f"galois_key_{s}"]):
        assert np.array_equal(got, vec[name]), name
        n += 1
    assert n == 19
    for c, sb in enumerate(vec["enc_scale_bits"]):  # CKKSEncoder::encode vectors
        assert np.array_equal(o.encode(2, vec[f"enc_values_{c}"], 2.0 ** int(sb)), vec[f"out_encode_{c}"]), f"encode {c}"
    assert [o.psi(i) for i in range(3)] == [int(x) for x in vec["psi"]]
    # Decryptor::decrypt / CKKSEncoder::decode vectors (float64 results compared as bit patterns)
    assert np.array_equal(o.decrypt(vec["a2"], vec["sk_ntt"]), vec["out_decrypt2"])
    assert np.array_equal(o.decrypt(vec["a3"], vec["sk_ntt"]), vec["out_decrypt3"])
    for name, pt, sb in _decode_cases(vec):
        assert np.array_equal(o.decode(pt, 2.0 ** sb).view(np.uint64), vec[name].view(np.uint64)), name


def _decode_cases(vec):
    for c, sb in enumerate(vec["enc_scale_bits"]):
        yield f"out_decode_{c}", vec[f"out_encode_{c}"], int(sb)
    yield "out_decode_pt", vec["pt"], 10
    yield "out_decode_dec3", vec["out_decrypt3"], 10


def test_reference_constants(consts):
    from oracle import pyoracle as po
    from eva_amd.hostref import coeff_modulus_create
    for row in consts["coeff_modulus_create"]:
        assert po.coeff_modulus_create(row["N"], row["bits"]) == row["primes"]
        assert coeff_modulus_create(row["N"], row["bits"]) == row["primes"]  # the product's own generator
    for row in consts["minimal_primitive_root"]:
        assert po.lib.evo_minimal_primitive_root(row["N"], row["q"]) == row["psi"]


def test_compiler_reproduces_reference_prime_bits(consts):
    from eva import EvaProgram, Input, Output
    from eva.ckks import CKKSCompiler
    for row in consts["compiler_prime_bits"]:
        prog = EvaProgram('kat', vec_size=row["vec_size"])
        with prog:
            if row["program"] == "y = x*x":
                x = Input('x')
                Output('y', x * x)
            else:
                x1, x2, x3, x4 = (Input(f'x{i}') for i in range(1, 5))
                Output('y', (x1 * (x2 * (x3 * x4))) + (x1 + (x2 + (x3 + x4))))
        prog.set_output_ranges(row["output_ranges"])
        prog.set_input_scales(row["input_scales"])
        _, params, _ = CKKSCompiler(config=row["config"]).compile(prog)
        assert list(params.prime_bits) == row["prime_bits"], row["test"]


class _GpuAdapter:
    """the oracle's call shapes over libeva_hip.so"""

    def __init__(self, vec):
        from eva_amd import backend
        self.g = backend.Context(N, [int(p) for p in vec["primes"]])
        self.g.upload_relin_key(vec["relin_key"])
        for s in vec["rot_steps"]:
            self.g.upload_galois_key(self.g.galois_elt_from_step(int(s)), vec[f"galois_key_{int(s)}"])
        self.s = 2.0 ** 10

    def _c(self, x):
        return self.g.upload_ct(x, self.s)

    def add(self, a, b): return self.g.add(self._c(a), self._c(b)).download()
    def sub(self, a, b): return self.g.sub(self._c(a), self._c(b)).download()
    def negate(self, a): return self.g.negate(self._c(a)).download()
    def add_plain(self, a, p): return self.g.add_plain(self._c(a), self.g.upload_pt(p, self.s)).download()
    def sub_plain(self, a, p): return self.g.sub_plain(self._c(a), self.g.upload_pt(p, self.s)).download()
    def multiply(self, a, b): return self.g.multiply(self._c(a), self._c(b)).download()
    def square(self, a): return self.g.square(self._c(a)).download()
    def multiply_plain(self, a, p): return self.g.multiply_plain(self._c(a), self.g.upload_pt(p, self.s)).download()
    def relinearize(self, a, key): return self.g.relinearize(self._c(a)).download()
    def rescale(self, a): return self.g.rescale(self._c(a), 5).download()
    def mod_switch(self, a): return self.g.mod_switch(self._c(a)).download()
    def rotate(self, a, steps, key): return self.g.rotate(self._c(a), steps).download()

    def op_triple(self, a, b, key):
        return self.g.relinearize_rescale(self.g.multiply(self._c(a), self._c(b)), 5).download()


@pytest.mark.gpu
def test_gpu_reproduces_golden_vectors(vec):
    o = _GpuAdapter(vec)
    assert np.array_equal(o.g.test_ntt(0, vec["poly"]), vec["out_ntt0"])
    assert np.array_equal(o.g.test_ntt(0, vec["poly"], inverse=True), vec["out_intt0"])
    for name, got in _ops(o, vec, lambda s: None):
        assert np.array_equal(got, vec[name]), name
    # the fused and batched forms against the same vectors
    A3 = o._c(vec["a3"])
    assert np.array_equal(o.g.relinearize_rescale(A3, 5).download(), vec["out_relin_rescale"])
    outs = o.g.rotate_many(o._c(vec["a2"]), [int(s) for s in vec["rot_steps"]])
    for s, ct in zip(vec["rot_steps"], outs):
        assert np.array_equal(ct.download(), vec[f"out_rotate_{int(s)}"])
    for c, sb in enumerate(vec["enc_scale_bits"]):  # the device encoder against the stored plaintexts
        got = o.g.encode_pt(vec[f"enc_values_{c}"], 2, 2.0 ** int(sb)).download()
        assert np.array_equal(got, vec[f"out_encode_{c}"]), f"encode {c}"
    # the device decryptor + decoder: a plaintext is its own message as a size-1 ciphertext
    o.g.upload_secret_key(vec["sk_ntt"])
    for name, pt, sb in _decode_cases(vec):
        got = o.g.decrypt_decode(o.g.upload_ct(pt[None], 2.0 ** sb), N // 2)
        assert np.array_equal(got.view(np.uint64), vec[name].view(np.uint64)), name
    for ct, want in ((vec["a2"], "out_decrypt2"), (vec["a3"], "out_decrypt3")):
        got = o.g.decrypt_decode(o.g.upload_ct(ct, 2.0 ** 10), N // 2)
        assert np.array_equal(got.view(np.uint64), o_decode(vec[want], 10).view(np.uint64)), want


def o_decode(pt, scale_bits):
    from oracle import pyoracle as po
    return po.Oracle(N, po.coeff_modulus_create(N, [60, 40, 60])).decode(pt, 2.0 ** scale_bits)
