"""Device-resident valuations (the valuation "may hold device handles"; reference flow
/root/reference/eva/seal/seal.cpp:24-146, seal_executor.h:264-277, 420-435): encrypt -> execute ->
decrypt passes ciphertexts by handle.  Asserted here: no ciphertext crosses the host boundary in
between (the library's transfer counter), the ciphertexts are the ones the host-valuation path and
the CPU oracle produce (bit for bit), a result survives later calls (graph replays reuse their
buffers), values outlive the contexts that made them, and files still hold host words."""
import gc
import os

import numpy as np
import pytest

from eva import EvaProgram, Input, Output, evaluate, save, load
from eva.ckks import CKKSCompiler
from eva.seal import generate_keys
from evatest import oracle_execute

pytestmark = pytest.mark.gpu


def _prog(n_vec=512, N=8192, scale=40):
    prog = EvaProgram('resident', vec_size=n_vec)
    with prog:
        x, y = Input('x'), Input('y')
        Output('z', (x * y + x) << 3)
        Output('w', x - y)
    prog.set_input_scales(scale)
    prog.set_output_ranges(20)
    compiled, params, sig = CKKSCompiler(config={'warn_vec_size': 'false'}).compile(prog)
    params.poly_modulus_degree = N
    return compiled, params, sig


def _inputs(seed, n=512):
    rng = np.random.default_rng(seed)
    return {'x': list(rng.uniform(-2, 2, n)), 'y': list(rng.uniform(-2, 2, n))}


def _same(a, b):
    assert sorted(a.names()) == sorted(b.names())
    for name in a.names():
        g, o = a.get(name), b.get(name)
        assert g[:4] == o[:4], (name, g[:4], o[:4])
        assert np.array_equal(g[4], o[4]), f"{name}: ciphertext words differ"


def test_encrypt_execute_decrypt_moves_no_ciphertext_over_pcie():
    compiled, params, sig = _prog()
    pub, sec = generate_keys(params, 11)
    assert pub.resident
    inputs = _inputs(1)
    enc = pub.encrypt(inputs, sig)
    assert all(enc.is_resident(n) and not enc.on_host(n) for n in ('x', 'y'))
    pub.execute(compiled, enc)  # first walk: encodes the program's constants (plaintext uploads are allowed there)
    before = pub.transfer_stats()
    ref = evaluate(compiled, inputs)
    for call in range(4):  # eager walk, graph capture, two replays
        enc = pub.encrypt(inputs, sig)
        out = pub.execute(compiled, enc)
        assert all(out.is_resident(n) and not out.on_host(n) for n in out.names())
        res = sec.decrypt(out, sig)
        for name in ref:
            assert np.abs(np.array(res[name]) - np.array(ref[name])).max() < 1e-4, (call, name)
    after = pub.transfer_stats()
    assert after["ct_uploads"] == before["ct_uploads"] and after["ct_downloads"] == before["ct_downloads"], (before, after)
    assert after["pt_uploads"] == before["pt_uploads"], (before, after)
    # asking for the words is what downloads (once: the host copy is kept)
    out.get('z'); out.get('z')
    assert pub.transfer_stats()["ct_downloads"] == before["ct_downloads"] + 1
    assert out.on_host('z') and out.is_resident('z')


@pytest.mark.parametrize("copy_limit", [0, 1 << 40], ids=["eager-on-handles", "graph-with-device-slots"])
def test_resident_results_equal_host_results_and_oracle(copy_limit):
    compiled, params, sig = _prog()
    pub, sec = generate_keys(params, 12)
    pub.graph_copy_limit = copy_limit
    enc = pub.encrypt(_inputs(2), sig)
    want = oracle_execute(pub, compiled, enc)  # CPU oracle walk on the same encrypted inputs (downloads them)
    for call in range(4):
        _same(pub.execute(compiled, enc), want)
    # host valuations through the same context: identical ciphertexts
    pub.resident = False
    host_enc = load_roundtrip(enc)
    assert not host_enc.is_resident('x')
    for call in range(3):
        out = pub.execute(compiled, host_enc)
        assert not out.is_resident('z')
        _same(out, want)
    # host inputs, resident outputs (uploads of one call overlap the kernels of the previous one)
    pub.resident = True
    outs = [pub.execute(compiled, host_enc) for _ in range(4)]
    for out in outs:
        assert out.is_resident('z')
        _same(out, want)


def load_roundtrip(val, tmp="/tmp/_eva_resident_val.bin"):
    save(val, tmp)
    got = load(tmp)
    os.remove(tmp)
    return got


def test_results_survive_later_calls_and_saved_files_hold_words():
    compiled, params, sig = _prog()
    pub, sec = generate_keys(params, 13)
    encs = [pub.encrypt(_inputs(10 + i), sig) for i in range(3)]
    wants = [oracle_execute(pub, compiled, e) for e in encs]
    outs = []
    for rep in range(2):           # second round replays the captured graph into the same buffers
        for e in encs:
            outs.append(pub.execute(compiled, e))
    for i, out in enumerate(outs):
        _same(out, wants[i % 3])
    back = load_roundtrip(outs[0])  # a file holds host words whatever the valuation held
    assert not back.is_resident('z')
    _same(back, wants[0])
    res = sec.decrypt(back, sig)
    ref = evaluate(compiled, _inputs(10))
    assert np.abs(np.array(res['z']) - np.array(ref['z'])).max() < 1e-4


def test_values_outlive_their_contexts():
    compiled, params, sig = _prog()
    pub, sec = generate_keys(params, 14)
    enc = pub.encrypt(_inputs(5), sig)
    want = oracle_execute(pub, compiled, enc)
    outs = [pub.execute(compiled, enc) for _ in range(3)]
    del pub, sec
    gc.collect()
    for out in outs:
        _same(out, want)


def test_decrypt_by_a_separately_loaded_secret_context():
    """a secret context that does not share the key pair's device state gets the words through the host"""
    compiled, params, sig = _prog()
    pub, sec = generate_keys(params, 15)
    tmp = "/tmp/_eva_resident_sec.bin"
    save(sec, tmp)
    sec2 = load(tmp)
    os.remove(tmp)
    inputs = _inputs(6)
    out = pub.execute(compiled, pub.encrypt(inputs, sig))
    res = sec2.decrypt(out, sig)
    ref = evaluate(compiled, inputs)
    for name in ref:
        assert np.abs(np.array(res[name]) - np.array(ref[name])).max() < 1e-4


def test_output_that_is_an_input_keeps_its_handle():
    prog = EvaProgram('passthrough', vec_size=256)
    with prog:
        x = Input('x')
        Output('same', x)
        Output('twice', x + x)
    prog.set_input_scales(30)
    prog.set_output_ranges(20)
    compiled, params, sig = CKKSCompiler(config={'warn_vec_size': 'false'}).compile(prog)
    pub, sec = generate_keys(params, 16)
    inputs = {'x': list(np.linspace(-1, 1, 256))}
    enc = pub.encrypt(inputs, sig)
    for call in range(3):
        out = pub.execute(compiled, enc)
        res = sec.decrypt(out, sig)
        assert np.abs(np.array(res['same']) - np.array(inputs['x'])).max() < 1e-4
        assert np.abs(np.array(res['twice']) - 2 * np.array(inputs['x'])).max() < 1e-4
    assert np.array_equal(out.get('same')[4], enc.get('x')[4])


def test_unreduced_input_words_are_an_error_at_execute():
    """a ciphertext handed over as words (a file, a numpy array) is untrusted: a word >= its prime breaks the
    kernels' lazy-reduction bounds, so execute() refuses it (r2 advisor finding) — checked once per value"""
    from eva.seal import SEALValuation
    compiled, params, sig = _prog()
    pub, sec = generate_keys(params, 5)
    enc = pub.encrypt(_inputs(1), sig)
    good = enc.get('x')
    words = np.array(good[4], dtype=np.uint64).reshape(good[1], good[2], -1)
    v = SEALValuation()
    v._set_cipher('y', np.array(enc.get('y')[4], dtype=np.uint64).reshape(words.shape), enc.get('y')[3])
    bad = words.copy()
    bad[1, 0, 17] = np.uint64(2 ** 63)
    v._set_cipher('x', bad, good[3])
    with pytest.raises(RuntimeError, match="not reduced modulo its prime"):
        pub.execute(compiled, v)
    v._set_cipher('x', words, good[3])
    out = pub.execute(compiled, v)
    _same(out, pub.execute(compiled, enc))


def test_back_to_back_calls_use_twin_graph_plans_and_change_no_bit():
    """r6: a caller that issues execute() while the previous replay of the program is still running gets a twin plan (the walk
    captured a second time, own slots and buffers, own queue) and the calls go to whichever plan is idle.  Every result is
    the oracle walk's of ITS input, whatever plan produced it; a caller that synchronises after every call never has two."""
    from eva_amd.workloads import harris, image
    from oracle_executor import c_walk
    compiled, params, sig = CKKSCompiler(config={'warn_vec_size': 'false'}).compile(harris())
    params.poly_modulus_degree = 16384
    pub, sec = generate_keys(params, 9)
    encs = [pub.encrypt(image(4096, shift=7 * u), sig) for u in range(3)]
    refs = [c_walk(pub, compiled, e, threads=8)[0] for e in encs]
    for u in (0, 1, 2):  # eager walk, capture, first replay — each waited for: one plan
        out = pub.execute(compiled, encs[u])
        pub.synchronize()
        for name, words in refs[u].items():
            assert np.array_equal(out.get(name)[4], words)
    n_plans = lambda: pub._graph_plans()
    one = n_plans()
    outs = [pub.execute(compiled, encs[i % 3]) for i in range(12)]  # nothing waits in between
    pub.synchronize()
    assert n_plans() == one + 1, "a busy plan did not get its twin"
    for i, out in enumerate(outs):
        for name, words in refs[i % 3].items():
            assert np.array_equal(out.get(name)[4], words), f"call {i}"
    # and with the twin switched off: the same words from the one plan
    pub.twin_plans = False
    outs = [pub.execute(compiled, encs[i % 3]) for i in range(6)]
    pub.synchronize()
    for i, out in enumerate(outs):
        for name, words in refs[i % 3].items():
            assert np.array_equal(out.get(name)[4], words)
