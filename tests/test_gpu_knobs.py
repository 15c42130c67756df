"""Every launch knob of libeva_hip.so (Tunables, eva_amd/csrc/internal.hip.h) decides HOW a call is launched, never
what it computes: under each non-default setting a compact set of calls — the fused and batched key-switch forms, a
rotation set, a convolution window, the same on a transparent source — must return the oracle's words.  (The knobs
are read once per context, so each setting gets a context of its own.)  The SEAL calls behind these entry points:
/root/reference/eva/seal/seal_executor.h:164, :181/:188, :200, :213."""
import os

import numpy as np
import pytest

from eva_amd import backend
from oracle import pyoracle as po

pytestmark = pytest.mark.gpu

KNOBS = [
    {"EVAH_FUSE_MAC": 0},  # the unfused reference path: separate digit transforms + inner product kernel
    {"EVAH_FUSE_MAC": 0, "EVAH_FOLD_PA": 0},
    {"EVAH_FOLD_PA": 0},
    {"EVAH_FUSE_MUL": 0},
    {"EVAH_MAC3": 0},
    {"EVAH_LOOP_N": 0},
    {"EVAH_MAC3": 0, "EVAH_LOOP_N": 0},
    {"EVAH_FUSE_SMALL": 0},
    {"EVAH_FUSE_SPECIAL_INV": 0},
    {"EVAH_SMALL_LR": 3},
    {"EVAH_HOIST": 0},
    {"EVAH_WIN_FUSE": 0},
    {"EVAH_FB_PERSIST": 0},
    {"EVAH_KS_THREADS": 256},
    {"EVAH_KS_GROUPS": 2},
    {"EVAH_HOIST_TABLE_FAIL": 1},
]
# all primes of the top-bit shape (MAC3 and the top-bit butterflies apply) / a chain with a small prime (they do not)
CHAINS = [(4096, [60, 60, 60, 60, 60]), (8192, [60, 30, 60, 60])]


def _ctx(N, primes, knobs):
    env = dict({"EVAH_HOIST_MIN_TILES": 0}, **knobs)
    old = {k: os.environ.get(k) for k in env}
    os.environ.update({k: str(v) for k, v in env.items()})
    try:
        return backend.Context(N, primes)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


@pytest.mark.parametrize("chain", CHAINS, ids=lambda c: f"N{c[0]}_k{len(c[1])}")
@pytest.mark.parametrize("knobs", KNOBS, ids=lambda kn: ",".join(f"{k[5:]}={v}" for k, v in kn.items()))
def test_every_knob_gives_the_oracles_words(knobs, chain):
    N, bits = chain
    primes = po.coeff_modulus_create(N, bits)
    k, l = len(primes), len(primes) - 1
    o = po.Oracle(N, primes)
    g = _ctx(N, primes, knobs)
    rng = np.random.default_rng(31 * N + len(knobs))

    def rand(prefix, nl):
        return np.stack([rng.integers(0, primes[i], size=prefix + (N,), dtype=np.uint64) for i in range(nl)], axis=len(prefix))
    rk = rand((l, 2), k)
    g.upload_relin_key(rk)
    steps = [1, 65, -3]
    gks = {}
    for st in steps:
        gks[st] = rand((l, 2), k)
        g.upload_galois_key(g.galois_elt_from_step(st), gks[st])
    a, b = rand((2,), l), rand((2,), l)
    A, B = g.upload_ct(a, 2.0 ** 20), g.upload_ct(b, 2.0 ** 20)
    div = bits[-2]
    triple = o.op_triple(a, b, rk)
    # the fused op-triple (single and batched), relinearize + rescale of a stored product (single, many, batched handle)
    for got in g.multiply_relinearize_rescale_many([A, B, A], [B, A, B], div):
        assert np.array_equal(got.download(), triple)
    M = g.multiply(A, B)
    m = o.multiply(a, b)
    assert np.array_equal(g.relinearize_rescale(M, div).download(), triple)
    for got in g.relinearize_rescale_many([M, M], div):
        assert np.array_equal(got.download(), triple)
    MB = g.upload_ct_batch(np.stack([m, m, m]), 2.0 ** 40)
    got = g.relinearize_rescale(MB, div).download()
    assert np.array_equal(got[0], triple) and np.array_equal(got[2], triple)
    assert np.array_equal(g.relinearize(M).download(), o.relinearize(m, rk))
    for got in g.relinearize_many([M, M]):
        assert np.array_equal(got.download(), o.relinearize(m, rk))
    # rotations: one, a set of one source, a window with two sums; then the same set on a transparent source
    assert np.array_equal(g.rotate(A, 65).download(), o.rotate(a, 65, gks[65]))
    for st, got in zip(steps, g.rotate_many(A, steps)):
        assert np.array_equal(got.download(), o.rotate(a, st, gks[st])), f"rotate_many step {st}"
    wts = [[rand((), l) for _ in [0] + steps] for _ in range(2)]
    W = [[g.upload_pt(w, 2.0 ** 10) for w in row] for row in wts]

    def window(src):
        rot = [src] + [o.rotate(src, st, gks[st]) for st in steps]
        outs = []
        for row in wts:
            acc = None
            for r, w in zip(rot, row):
                t = o.multiply_plain(r, w)
                acc = t if acc is None else o.add(acc, t)
            outs.append(acc)
        return outs
    for got, want in zip(g.rotate_weighted_sums([([(A, st) for st in [0] + steps], W)]), window(a)):
        assert np.array_equal(got.download(), want), "window sums"
    z = a.copy()
    z[1] = 0
    Z = g.upload_ct(z, 2.0 ** 20)
    for st, got in zip(steps, g.rotate_many(Z, steps)):
        assert np.array_equal(got.download(), o.rotate(z, st, gks[st])), f"transparent source, step {st}"
    for got, want in zip(g.rotate_weighted_sums([([(Z, st) for st in [0] + steps], W)]), window(z)):
        assert np.array_equal(got.download(), want), "transparent source, window sums"
    g.close()
