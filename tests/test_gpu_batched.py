"""Batched handles (`batch` independent ciphertexts in one evah_ct): every evaluator entry point
applied to a batched handle must give, for each instance, bit for bit what the CPU oracle gives
for that instance alone — the unit BASELINE config 4 (a batch of independent Sobel DAGs) runs on."""
import numpy as np
import pytest

from test_gpu_parity import CONFIGS, Env, env

pytestmark = pytest.mark.gpu
B = 3


def _each(out, ref_fn):
    d = out.download()
    assert d.shape[0] == B
    for b in range(B):
        assert np.array_equal(d[b], ref_fn(b)), f"instance {b} differs from the oracle"


@pytest.mark.parametrize("cfg", CONFIGS[:6], ids=lambda c: f"N{c[0]}")
def test_batched_elementwise_and_rescale(cfg):
    e = env(cfg)
    l = e.k - 1
    a2 = np.stack([e.rand(2, l) for _ in range(B)])
    b2 = np.stack([e.rand(2, l) for _ in range(B)])
    b3 = np.stack([e.rand(3, l) for _ in range(B)])
    pt = e.rand(1, l)[0]
    s = 2.0 ** 10
    A2, B2, B3 = (e.g.upload_ct_batch(x, s) for x in (a2, b2, b3))
    PT = e.g.upload_pt(pt, s)
    assert A2.batch == B and A2.info() == (2, l, s)
    assert np.array_equal(A2.download(), a2)
    _each(e.g.add(A2, B2), lambda b: e.o.add(a2[b], b2[b]))
    _each(e.g.add(A2, B3), lambda b: e.o.add(a2[b], b3[b]))
    _each(e.g.sub(B3, A2), lambda b: e.o.sub(b3[b], a2[b]))
    _each(e.g.sub(A2, B3), lambda b: e.o.sub(a2[b], b3[b]))
    _each(e.g.negate(B3), lambda b: e.o.negate(b3[b]))
    _each(e.g.add_plain(B3, PT), lambda b: e.o.add_plain(b3[b], pt))
    _each(e.g.sub_plain(A2, PT), lambda b: e.o.sub_plain(a2[b], pt))
    m = e.g.multiply(A2, B2)
    assert m.info() == (3, l, s * s) and m.batch == B
    _each(m, lambda b: e.o.multiply(a2[b], b2[b]))
    _each(e.g.square(A2), lambda b: e.o.square(a2[b]))
    _each(e.g.multiply_plain(B3, PT), lambda b: e.o.multiply_plain(b3[b], pt))
    if l >= 2:
        ms = e.g.mod_switch(B3)
        assert ms.info() == (3, l - 1, s) and ms.batch == B
        _each(ms, lambda b: e.o.mod_switch(b3[b]))
        _each(e.g.add(ms, e.g.mod_switch(A2)), lambda b: e.o.add(e.o.mod_switch(b3[b]), e.o.mod_switch(a2[b])))
        for x, X in ((a2, A2), (b3, B3)):
            r = e.g.rescale(X, 5)
            assert r.info() == (x.shape[1], l - 1, s / 32) and r.batch == B
            _each(r, lambda b: e.o.rescale(x[b]))
        if l >= 3:
            _each(e.g.rescale(ms, 5), lambda b: e.o.rescale(e.o.mod_switch(b3[b])))
    # stack / unstack
    singles = [e.g.upload_ct(a2[b], s) for b in range(B)]
    st = e.g.stack(singles)
    assert st.batch == B and np.array_equal(st.download(), a2)
    if l >= 2:  # r6 (one gather launch): sources with their own polynomial stride — mod-switched views next to plain ones
        mixed = [e.g.mod_switch(singles[0]), e.g.upload_ct(a2[1][:, :l - 1], s), e.g.mod_switch(e.g.upload_ct(b3[2][:2], s))]
        sm = e.g.stack(mixed)
        assert sm.batch == 3 and sm.info()[:2] == (2, l - 1)
        assert np.array_equal(sm.download(), np.stack([a2[0][:, :l - 1], a2[1][:, :l - 1], b3[2][:2, :l - 1]]))
    u = A2.unstack(1)
    assert u.batch == 1 and np.array_equal(u.download(), a2[1])
    assert np.array_equal(e.g.add(u, singles[0]).download(), e.o.add(a2[1], a2[0]))
    with pytest.raises(RuntimeError, match="batch size mismatch"):
        e.g.add(A2, singles[0])


@pytest.mark.parametrize("cfg", CONFIGS[:6], ids=lambda c: f"N{c[0]}")
def test_batched_key_switching(cfg):
    e = Env(*cfg)  # own context: this test installs Galois keys
    l = e.k - 1
    key = e.rand_key()
    e.g.upload_relin_key(key)
    a3 = np.stack([e.rand(3, l) for _ in range(B)])
    A3 = e.g.upload_ct_batch(a3, 2.0 ** 10)
    r = e.g.relinearize(A3)
    assert r.info() == (2, l, 2.0 ** 10) and r.batch == B
    _each(r, lambda b: e.o.relinearize(a3[b], key))
    if l >= 2:
        rr = e.g.relinearize_rescale(A3, 4)
        assert rr.info() == (2, l - 1, 2.0 ** 6) and rr.batch == B
        _each(rr, lambda b: e.o.rescale(e.o.relinearize(a3[b], key)))
        ms = e.g.mod_switch(A3)  # view: instance stride stays 3 * l * N
        _each(e.g.relinearize(ms), lambda b: e.o.relinearize(e.o.mod_switch(a3[b]), key))
        if l >= 3:
            _each(e.g.relinearize_rescale(ms, 4), lambda b: e.o.rescale(e.o.relinearize(e.o.mod_switch(a3[b]), key)))
    # rotations: one step, several steps (more (step, instance) pairs than one launch set takes at
    # B = 3 when there are > 21 steps is covered by the chunking test below), and step 0
    a2 = np.stack([e.rand(2, l) for _ in range(B)])
    A2 = e.g.upload_ct_batch(a2, 2.0 ** 10)
    steps = [1, -2, 5]
    keys = {}
    for st in steps:
        elt = e.g.galois_elt_from_step(st)
        keys[st] = e.rand_key()
        e.g.upload_galois_key(elt, keys[st])
    _each(e.g.rotate(A2, 1), lambda b: e.o.rotate(a2[b], 1, keys[1]))
    outs = e.g.rotate_many(A2, steps)
    for st, o in zip(steps, outs):
        assert o.batch == B
        _each(o, lambda b: e.o.rotate(a2[b], st, keys[st]))
    _each(e.g.rotate(A2, 0), lambda b: a2[b])


def test_batched_rotations_are_chunked():
    """24 instances x 9 steps = 216 (step, instance) pairs > 64 per launch set"""
    e = Env(*CONFIGS[0])
    l = e.k - 1
    nb = 24
    a2 = np.stack([e.rand(2, l) for _ in range(nb)])
    A2 = e.g.upload_ct_batch(a2, 2.0 ** 10)
    steps = [1, 2, 3, 4, 5, 6, 7, 8, 9]
    keys = {}
    for st in steps:
        keys[st] = e.rand_key()
        e.g.upload_galois_key(e.g.galois_elt_from_step(st), keys[st])
    outs = e.g.rotate_many(A2, steps)
    for st, o in zip(steps, outs):
        d = o.download()
        for b in (0, 7, 23):
            assert np.array_equal(d[b], e.o.rotate(a2[b], st, keys[st]))
    key = e.rand_key()
    e.g.upload_relin_key(key)
    nb = 70  # > KS_BATCH_MAX instances are not accepted in one handle
    with pytest.raises(RuntimeError, match="batch must be"):
        e.g.upload_ct_batch(np.zeros((nb, 2, l, e.N), dtype=np.uint64), 1.0)
