"""save / load of all six object kinds (reference: tests/features.py:154-217,
examples/serialization.py).  CPU part: round trips + the execute step walked over the oracle;
GPU part: the reference's full client / server flow through execute() on the MI355X."""
import os

import numpy as np
import pytest

from eva import EvaProgram, Input, Output, evaluate, save, load, Op
from eva.ckks import CKKSCompiler
from eva.metric import valuation_mse
from eva.seal import generate_keys
from evatest import oracle_execute


def _flow(tmp_path, execute):
    poly = EvaProgram('Polynomial', vec_size=1024)
    with poly:
        x = Input('x')
        Output('y', 3 * x ** 2 + 5 * x - 2)
    poly.set_output_ranges(20)
    poly.set_input_scales(30)
    inputs = {'x': [i / 100.0 for i in range(poly.vec_size)]}
    reference = evaluate(poly, inputs)
    poly, params, signature = CKKSCompiler(config={'warn_vec_size': 'false'}).compile(poly)
    p = lambda n: os.path.join(tmp_path, n)
    save(poly, p('poly.eva'))
    save(params, p('poly.evaparams'))
    save(signature, p('poly.evasignature'))
    params = load(p('poly.evaparams'))
    public_ctx, secret_ctx = generate_keys(params)
    save(public_ctx, p('poly.sealpublic'))
    save(secret_ctx, p('poly.sealsecret'))
    signature = load(p('poly.evasignature'))
    public_ctx = load(p('poly.sealpublic'))
    enc_inputs = public_ctx.encrypt(inputs, signature)
    save(enc_inputs, p('poly_inputs.sealvals'))
    poly = load(p('poly.eva'))
    public_ctx = load(p('poly.sealpublic'))
    enc_inputs = load(p('poly_inputs.sealvals'))
    enc_outputs = execute(public_ctx, poly, enc_inputs)
    save(enc_outputs, p('poly_outputs.sealvals'))
    secret_ctx = load(p('poly.sealsecret'))
    enc_outputs = load(p('poly_outputs.sealvals'))
    outputs = secret_ctx.decrypt(enc_outputs, signature)
    assert valuation_mse(reference, evaluate(poly, inputs)) < 1e-10
    assert valuation_mse(outputs, reference) < 0.01


def test_serialization_flow_cpu(tmp_path):
    _flow(str(tmp_path), lambda pub, prog, enc: oracle_execute(pub, prog, enc))


@pytest.mark.gpu
def test_serialization_flow_gpu(tmp_path):
    _flow(str(tmp_path), lambda pub, prog, enc: pub.execute(prog, enc))


def test_round_trips_are_exact(tmp_path):
    prog = EvaProgram('p', vec_size=64)
    with prog:
        x = Input('x')
        y = Input('y', is_encrypted=False)
        Output('z', (x << 3) * [float(i) for i in range(64)] + y * 0.25 - (x >> 1))
    prog.set_output_ranges(20)
    prog.set_input_scales(30)
    compiled, params, sig = CKKSCompiler(config={'warn_vec_size': 'false'}).compile(prog)
    f = lambda n: os.path.join(str(tmp_path), n)
    save(compiled, f('a'))
    back = load(f('a'))
    # term ids are re-densified on save, so the topological listing may come back in another
    # (equally valid) order: compare as multisets of (op, arity, attributes)
    strip = lambda dump: sorted(repr(sorted(({k: v for k, v in d.items() if k != 'id'} | {'operands': len(d['operands'])}).items(), key=str)) for d in dump)
    assert back.name == compiled.name and back.vec_size == 64
    assert strip(back._dump()) == strip(compiled._dump())
    inputs = {'x': [i * 0.1 for i in range(64)], 'y': [1.0] * 64}
    assert evaluate(back, inputs) == evaluate(compiled, inputs)
    save(params, f('b'))
    p2 = load(f('b'))
    assert list(p2.prime_bits) == list(params.prime_bits) and set(p2.rotations) == set(params.rotations)
    assert p2.poly_modulus_degree == params.poly_modulus_degree
    save(sig, f('c'))
    s2 = load(f('c'))
    assert s2.vec_size == sig.vec_size and {k: (v.input_type, v.scale, v.level) for k, v in s2.inputs.items()} == \
        {k: (v.input_type, v.scale, v.level) for k, v in sig.inputs.items()}
    pub, sec = generate_keys(params, 5)
    enc = pub.encrypt(inputs, sig)
    save(enc, f('d'))
    e2 = load(f('d'))
    for name in enc.names():
        a, b = enc.get(name), e2.get(name)
        assert a[:4] == b[:4] and np.array_equal(np.asarray(a[4]), np.asarray(b[4]))
    save(pub, f('e'))
    pub2 = load(f('e'))
    assert np.array_equal(pub.relin_key(), pub2.relin_key()) and pub.primes == pub2.primes
    assert all(np.array_equal(v, pub2.galois_keys()[k]) for k, v in pub.galois_keys().items())
    with pytest.raises(RuntimeError):
        open(f('bad'), 'wb').write(b'nonsense')
        load(f('bad'))


@pytest.mark.parametrize("seed", [2001, 2004, 3002, 3005, 4201, 4203, 17, 29])
def test_random_programs_survive_save_and_load(tmp_path, seed):
    """Random programs (tests/test_gpu_fuzz.py's generator: constants, vector constants, rotations,
    unencrypted inputs): the compiled program, its parameters, signature and an encrypted valuation
    are unchanged by save -> load (same term dump, same prime bits / rotations, same words)."""
    from test_gpu_fuzz import _random_program
    prog, inputs = _random_program(seed, 32)
    compiled, params, sig = CKKSCompiler(config={'warn_vec_size': 'false'}).compile(prog)
    p = lambda n: os.path.join(str(tmp_path), n)
    save(compiled, p('c.eva')); save(params, p('c.evaparams')); save(sig, p('c.evasignature'))
    c2, p2, s2 = load(p('c.eva')), load(p('c.evaparams')), load(p('c.evasignature'))

    def shape(pr):  # ids may be renumbered by the round trip: compare the multiset of node descriptions
        d = {x["id"]: x for x in pr._dump()}
        def desc(x):
            return (str(x["op"]), tuple(str(d[o]["op"]) for o in x["operands"]), x.get("rotation"), x.get("rescale_divisor"),
                    x.get("encode_scale"), x.get("encode_level"), tuple(x.get("constant") or ()))
        return sorted(map(desc, d.values()), key=repr)
    assert shape(compiled) == shape(c2)
    assert list(params.prime_bits) == list(p2.prime_bits) and set(params.rotations) == set(p2.rotations)
    assert params.poly_modulus_degree == p2.poly_modulus_degree and sig.vec_size == s2.vec_size
    assert valuation_mse(evaluate(compiled, inputs), evaluate(c2, inputs)) == 0
    pub, sec = generate_keys(params, seed)
    enc = pub.encrypt(inputs, sig)
    save(enc, p('v.sealvals'))
    enc2 = load(p('v.sealvals'))
    for name in enc.names():
        a, b = enc.get(name), enc2.get(name)
        assert a[:4] == b[:4]
        assert np.array_equal(np.asarray(a[4]), np.asarray(b[4]))


def test_malformed_files_and_values_are_rejected(tmp_path):
    """load() validates structure (op codes, operand counts, bindings, key / value sizes) and the
    executor re-checks value shapes before any upload: crafted inputs raise instead of reading out
    of bounds."""
    import struct
    from eva import EvaProgram, Input, Output, save, load
    from eva.seal import SEALValuation
    prog = EvaProgram('p', vec_size=8)
    with prog:
        Output('y', Input('x') + Input('z'))
    path = str(tmp_path / "p.eva")
    save(prog, path, format="native")
    raw = bytearray(open(path, "rb").read())
    # find the Add term's record: op code 11 followed by operand count 2 -> make it 1
    needle = struct.pack("<iI", 11, 2)
    at = bytes(raw).find(needle)
    assert at > 0
    bad = bytearray(raw)
    bad[at + 4:at + 8] = struct.pack("<I", 1)
    open(path, "wb").write(bad)
    with pytest.raises(RuntimeError, match="operand count"):
        load(path)
    bad = bytearray(raw)
    bad[at:at + 4] = struct.pack("<i", 99)
    open(path, "wb").write(bad)
    with pytest.raises(RuntimeError, match="op code"):
        load(path)
    open(path, "wb").write(raw[:len(raw) // 2])
    with pytest.raises(RuntimeError, match="truncated|parse"):
        load(path)
    # a length field near 2^64 must not wrap the bounds check
    bad = bytearray(raw[:12]) + struct.pack("<Q", 2 ** 64 - 4) + bytearray(raw[20:])
    open(path, "wb").write(bad)
    with pytest.raises(RuntimeError, match="truncated|parse"):
        load(path)
    # a valuation whose declared shape disagrees with its data
    v = SEALValuation()
    v._set_cipher('x', np.zeros((2, 2, 1024), dtype=np.uint64), 2.0 ** 30)
    vpath = str(tmp_path / "v.eva")
    save(v, vpath)
    rawv = bytearray(open(vpath, "rb").read())
    at = bytes(rawv).find(struct.pack("<III", 1, 2, 2))   # kind, size, limbs
    rawv[at + 8:at + 12] = struct.pack("<I", 7)          # limbs := 7
    open(vpath, "wb").write(rawv)
    with pytest.raises(RuntimeError, match="shape"):
        load(vpath)


def test_key_files_with_unreduced_words_are_rejected(tmp_path):
    """the kernels' lazy-reduction bounds assume canonical residues: a key file with a word >= its prime, or a
    secret key that is not ternary, is an error at load time (r2 advisor finding), not a silently wrong result"""
    import struct
    from eva import EvaProgram, Input, Output, save, load
    from eva.ckks import CKKSCompiler
    from eva.seal import generate_keys
    prog = EvaProgram('p', vec_size=8)
    with prog:
        Output('y', Input('x') * Input('x'))
    prog.set_input_scales(30)
    prog.set_output_ranges(20)
    compiled, params, sig = CKKSCompiler(config={'warn_vec_size': 'false'}).compile(prog)
    pub, sec = generate_keys(params, 3)
    pp, sp = str(tmp_path / "pub"), str(tmp_path / "sec")
    save(pub, pp)
    save(sec, sp)
    assert load(pp).poly_modulus_degree == pub.poly_modulus_degree
    raw = bytearray(open(pp, "rb").read())
    assert struct.unpack("<Q", raw[-8:])[0] == 0   # no Galois keys in this program: the file ends with their count
    raw[-16:-8] = struct.pack("<Q", 2 ** 64 - 1)  # the last word of the relinearization key: above any 60-bit prime
    open(pp, "wb").write(raw)
    with pytest.raises(RuntimeError, match="not reduced modulo its prime"):
        load(pp)
    raw = bytearray(open(sp, "rb").read())
    N = pub.poly_modulus_degree
    at = 12 + 4 + 8 + 8 * len(list(pub.primes)) + 8   # header, N, prime vector, length of s
    raw[at] = 5                                        # first coefficient of s := 5
    open(sp, "wb").write(raw)
    with pytest.raises(RuntimeError, match="ternary"):
        load(sp)
    raw = bytearray(open(sp, "rb").read())
    raw[at] = 1
    raw[-8:] = struct.pack("<Q", 2 ** 63)
    open(sp, "wb").write(raw)
    with pytest.raises(RuntimeError, match="not reduced modulo its prime"):
        load(sp)


def test_integral_exponents_of_any_type():
    from eva import EvaProgram, Input, Output, evaluate
    prog = EvaProgram('pow', vec_size=4)
    with prog:
        x = Input('x')
        Output('a', x ** np.int64(3))
        Output('b', x ** 2)
        for bad in (0, -1, 2.0, True, "2"):
            with pytest.raises(ValueError):
                x ** bad
    out = evaluate(prog, {'x': [1.0, 2.0, 3.0, -1.0]})
    assert out['a'] == [1.0, 8.0, 27.0, -1.0] and out['b'] == [1.0, 4.0, 9.0, 1.0]
