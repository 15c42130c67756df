"""Hoisted rotations (several rotations of ONE ciphertext share the digit decomposition of c1;
eva_amd/csrc/rotation_sets.hip.h: k_hoist_mac) against the CPU oracle's rotate-then-switch-key, which
follows SEAL's Evaluator::rotate_internal (/root/reference/eva/seal/seal_executor.h:181,188).
The contexts are created with EVAH_HOIST_MIN_TILES=0, so every rotate_many of >= 2 steps takes
the hoisted path; the zero-coefficient cases exercise the per-coefficient correction (k_hoist_fix)
and, beyond its capacity, the guarded unhoisted fallback."""
import os

import numpy as np
import pytest

from eva_amd import backend
from oracle import pyoracle as po

pytestmark = pytest.mark.gpu

CONFIGS = [
    (2048, [40, 20, 40, 41]),
    (4096, [60, 20, 60, 60]),
    (8192, [60, 30, 60, 60, 60]),
    (16384, [60, 20, 60, 60, 60, 60]),
    (32768, [60] * 5),
    (65536, [60] * 4),
    (4096, [40] * 19 + [41]),  # l = 19 digits
    (4096, [50, 60]),  # l = 1: one digit, data limb + special prime only
]
STEPS = [1, 2, 64, 65, -3, 129, -64, 1000]


class Env:
    def __init__(self, N, bits, hoist=True):
        old = {k: os.environ.get(k) for k in ("EVAH_HOIST", "EVAH_HOIST_MIN_TILES")}
        os.environ["EVAH_HOIST"] = "1" if hoist else "0"
        os.environ["EVAH_HOIST_MIN_TILES"] = "0"
        try:
            self.N = N
            self.primes = po.coeff_modulus_create(N, bits)
            self.k = len(self.primes)
            self.o = po.Oracle(N, self.primes)
            self.g = backend.Context(N, self.primes)
        finally:
            for k, v in old.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
        self.rng = np.random.default_rng(7 * N + len(bits))
        self.keys = {}

    def rand(self, size, l):
        return np.stack([np.stack([self.rng.integers(0, self.primes[i], size=self.N, dtype=np.uint64)
                                   for i in range(l)]) for _ in range(size)])

    def key_for(self, step):
        if step not in self.keys:
            key = np.stack([np.stack([np.stack([self.rng.integers(0, self.primes[i], size=self.N, dtype=np.uint64)
                                                for i in range(self.k)]) for _ in range(2)])
                            for _ in range(self.k - 1)])
            self.g.upload_galois_key(self.g.galois_elt_from_step(step), key)
            self.keys[step] = key
        return self.keys[step]


_envs = {}


def env(cfg):
    key = (cfg[0], tuple(cfg[1]))
    if key not in _envs:
        _envs[key] = Env(*cfg)
    return _envs[key]


@pytest.mark.parametrize("cfg", CONFIGS, ids=lambda c: f"N{c[0]}_k{len(c[1])}")
def test_hoisted_rotations_bit_exact(cfg):
    e = env(cfg)
    l = e.k - 1
    a2 = e.rand(2, l)
    A2 = e.g.upload_ct(a2, 2.0 ** 20)
    steps = STEPS if e.N >= 4096 else STEPS[:-1]
    for st in steps:
        e.key_for(st)
    outs = e.g.rotate_many(A2, steps)
    for st, o in zip(steps, outs):
        assert o.info() == (2, l, 2.0 ** 20)
        assert np.array_equal(o.download(), e.o.rotate(a2, st, e.keys[st])), f"hoisted rotate step {st}"
    del outs
    # one level down through a mod-switched view (different level => different constants), two steps
    if l > 1:
        ms = e.g.mod_switch(A2)
        o = e.g.rotate_many(ms, [65, -3])
        low = e.o.mod_switch(a2)
        assert np.array_equal(o[0].download(), e.o.rotate(low, 65, e.keys[65]))
        assert np.array_equal(o[1].download(), e.o.rotate(low, -3, e.keys[-3]))


@pytest.mark.parametrize("cfg", [CONFIGS[2], CONFIGS[3]], ids=lambda c: f"N{c[0]}")
def test_hoisted_rotations_of_a_batched_handle(cfg):
    """B instances x n rotations: more (rotation, instance) pairs than one launch set holds."""
    e = env(cfg)
    l = e.k - 1
    B = 24
    inst = [e.rand(2, l) for _ in range(B)]
    H = e.g.upload_ct_batch(np.stack(inst), 2.0 ** 30)
    steps = [1, 64, 65, -3, 129]
    for st in steps:
        e.key_for(st)
    outs = e.g.rotate_many(H, steps)
    for st, o in zip(steps, outs):
        got = o.download()
        for b in (0, 7, B - 1):
            assert np.array_equal(got[b], e.o.rotate(inst[b], st, e.keys[st])), f"step {st} instance {b}"


@pytest.mark.parametrize("case", ["transparent", "one_zero_flipped", "one_zero_unflipped", "zero_limb", "several_zeros"])
def test_zero_digit_coefficients_stay_exact(case):
    """The hoisting identity needs non-zero digit coefficients at the sign-flipped positions; zero
    coefficients must still give SEAL's bits: a few through the per-coefficient correction, many
    (transparent ciphertext, a zero limb) through the guarded unhoisted launches."""
    e = env(CONFIGS[2])
    l = e.k - 1
    N = e.N
    a2 = e.rand(2, l)
    steps = [1, 65, -3]
    for st in steps:
        e.key_for(st)
    if case == "transparent":
        a2[1] = 0
    elif case == "zero_limb":
        a2[1][1] = 0
    elif case == "several_zeros":
        for limb, count in ((0, 5), (2, 3), (l - 1, 1)):
            x = e.rng.integers(1, e.primes[limb], size=N, dtype=np.uint64)
            x[e.rng.choice(N, size=count, replace=False)] = 0
            a2[1][limb] = e.o.ntt(limb, x)
    else:
        # coefficient form of limb 0 of c1 with exactly one zero, at a position whose image under the
        # first rotation is (or is not) sign-flipped: index k maps to k*elt mod 2N, flipped iff >= N
        elt = po.galois_elt_from_step(N, steps[0])
        want = case == "one_zero_flipped"
        kpos = next(k for k in range(1, N) if (((k * elt) >> int(np.log2(N))) & 1 == 1) == want)
        x = e.rng.integers(1, e.primes[0], size=N, dtype=np.uint64)
        x[kpos] = 0
        a2[1][0] = e.o.ntt(0, x)
    A2 = e.g.upload_ct(a2, 2.0 ** 20)
    outs = e.g.rotate_many(A2, steps)
    for st, o in zip(steps, outs):
        assert np.array_equal(o.download(), e.o.rotate(a2, st, e.keys[st])), f"{case}: step {st}"


def test_zero_coefficients_in_some_instances_of_a_batched_handle():
    e = env(CONFIGS[2])
    l = e.k - 1
    B = 6
    inst = [e.rand(2, l) for _ in range(B)]
    for b, limb, count in ((1, 0, 2), (4, 1, 1), (4, 3, 2)):
        x = e.rng.integers(1, e.primes[limb], size=e.N, dtype=np.uint64)
        x[e.rng.choice(e.N, size=count, replace=False)] = 0
        inst[b][1][limb] = e.o.ntt(limb, x)
    steps = [1, 2, 65, -3, -64]
    for st in steps:
        e.key_for(st)
    outs = e.g.rotate_many(e.g.upload_ct_batch(np.stack(inst), 2.0 ** 30), steps)
    for st, o in zip(steps, outs):
        got = o.download()
        for b in range(B):
            assert np.array_equal(got[b], e.o.rotate(inst[b], st, e.keys[st])), f"step {st} instance {b}"


@pytest.mark.parametrize("cfg", [CONFIGS[2], CONFIGS[4]], ids=lambda c: f"N{c[0]}")
def test_rotation_set_with_several_sources_shares_each_sources_digits(cfg):
    """evah_rotate_pairs: three convolutions' worth of sibling rotations plus a lone rotation in one
    set (what a level of Harris looks like); one source carries zero digit coefficients."""
    e = env(cfg)
    l = e.k - 1
    srcs = [e.rand(2, l) for _ in range(4)]
    x = e.rng.integers(1, e.primes[1], size=e.N, dtype=np.uint64)
    x[e.rng.choice(e.N, size=3, replace=False)] = 0
    srcs[1][1][1] = e.o.ntt(1, x)
    hs = [e.g.upload_ct(a, 2.0 ** 20) for a in srcs]
    plan = [(0, 1), (1, 1), (2, 1), (0, 65), (1, 65), (2, 65), (0, -3), (1, -64), (2, 129), (3, 2), (1, 2)]
    for _, st in plan:
        e.key_for(st)
    outs = e.g.rotate_pairs([hs[i] for i, _ in plan], [st for _, st in plan])
    for (i, st), o in zip(plan, outs):
        assert np.array_equal(o.download(), e.o.rotate(srcs[i], st, e.keys[st])), f"source {i} step {st}"


def test_replacing_a_galois_key_invalidates_the_hoisting_constants():
    e = Env(*CONFIGS[1])
    l = e.k - 1
    a2 = e.rand(2, l)
    A2 = e.g.upload_ct(a2, 2.0 ** 20)
    steps = [1, 65]
    for round_ in range(2):
        e.keys.clear()  # fresh random keys for the same Galois elements
        for st in steps:
            e.key_for(st)
        outs = e.g.rotate_many(A2, steps)
        for st, o in zip(steps, outs):
            assert np.array_equal(o.download(), e.o.rotate(a2, st, e.keys[st])), f"round {round_} step {st}"


def _launch_classes(e, A2, steps):
    e.g.profile(True)
    e.g.profile_reset()
    outs = e.g.rotate_many(A2, steps)
    e.g.sync()
    prof = e.g.profile_get()
    e.g.profile(False)
    return outs, prof


def test_the_hoisted_path_is_the_one_that_runs_and_can_be_disabled():
    """Launch accounting: hoisted = ONE inverse transform set of c1 and full digit transforms
    (ksdigit_pass2 launches exist) whatever the number of rotations; disabled = the fused
    per-rotation key switch (no stand-alone second digit pass)."""
    cfg = CONFIGS[2]
    steps = [1, 2, 65]
    e1 = env(cfg)
    for st in steps:
        e1.key_for(st)
    l = e1.k - 1
    a2 = e1.rand(2, l)
    outs, prof = _launch_classes(e1, e1.g.upload_ct(a2, 2.0 ** 20), steps)
    assert prof["ksdigit_pass2"][0] >= 1, prof
    for st, o in zip(steps, outs):
        assert np.array_equal(o.download(), e1.o.rotate(a2, st, e1.keys[st]))
    e0 = Env(*cfg, hoist=False)
    for st in steps:
        e0.key_for(st)
    outs, prof = _launch_classes(e0, e0.g.upload_ct(a2, 2.0 ** 20), steps)
    assert prof["ksdigit_pass2"][0] == 0, prof
    for st, o in zip(steps, outs):
        assert np.array_equal(o.download(), e0.o.rotate(a2, st, e0.keys[st]))


def test_exported_shortcut_vectors_through_the_hoisted_and_fused_paths():
    """The vectors tests/golden/export_seal_vectors.py writes for tools/seal_parity.cpp section 4b — what a SEAL host
    pins first — run through the product's two shortcuts: ONE hoisted rotation set per source (dense, zero limb in c1,
    transparent) and ONE fused multiply / square -> relinearize -> rescale.  Same residues as the oracle's SEAL-order
    evaluation stored in the vectors."""
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.join(here, "golden"))
    import export_seal_vectors as ex
    d = ex.add_shortcut_vectors(dict(np.load(os.path.join(here, "golden", "ops_n1024.npz"))), 1024)
    primes = [int(q) for q in d["primes"]]
    old = {k: os.environ.get(k) for k in ("EVAH_HOIST", "EVAH_HOIST_MIN_TILES")}
    os.environ["EVAH_HOIST"], os.environ["EVAH_HOIST_MIN_TILES"] = "1", "0"
    try:
        g = backend.Context(1024, primes)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    steps = [int(s) for s in d["hoist_steps"]]
    for s in steps:
        g.upload_galois_key(g.galois_elt_from_step(s), d[f"galois_key_h{s}"])
    g.upload_relin_key(d["relin_key"])
    for name in ("dense", "zero_limb", "transparent"):
        src = g.upload_ct(d[f"hoist_src_{name}"], 2.0 ** 10)
        outs = g.rotate_many(src, steps)
        for s, o in zip(steps, outs):
            assert np.array_equal(o.download(), d[f"out_hoist_{name}_{s}"]), (name, s)
    a, b = g.upload_ct(d["a2"], 2.0 ** 10), g.upload_ct(d["b2"], 2.0 ** 10)
    assert np.array_equal(g.multiply_relinearize_rescale_many([a], [b], 60)[0].download(), d["out_triple"])
    assert np.array_equal(g.multiply_relinearize_rescale_many([a], [a], 60)[0].download(), d["out_triple_square"])
    g.close()


# ---- window sums: evah_rotate_weighted_sums (rotations of a window fused with the weighted sums that consume them)
def _window_oracle(e, terms, weights):
    """terms: [(ciphertext words, step)], weights: rows of plaintext words (None = 1) -> the sums, op by op"""
    rot = [a if st == 0 else e.o.rotate(a, st, e.keys[st]) for a, st in terms]
    outs = []
    for row in weights:
        acc = None
        for r, w in zip(rot, row):
            t = r if w is None else e.o.multiply_plain(r, w)
            acc = t if acc is None else e.o.add(acc, t)
        outs.append(acc)
    return outs


def _rand_pt(e, l):
    return np.stack([e.rng.integers(0, e.primes[i], size=e.N, dtype=np.uint64) for i in range(l)])


@pytest.mark.parametrize("cfg", [CONFIGS[0], CONFIGS[2], CONFIGS[3], CONFIGS[4], CONFIGS[5], CONFIGS[6], CONFIGS[7]],
                         ids=lambda c: f"N{c[0]}_k{len(c[1])}")
def test_window_sums_bit_exact(cfg):
    """One 3x3 window with two filters over the same rotations (convolutionXY) and, in the same call, windows with one
    sum, an unrotated term in the middle / absent, and weights of 1."""
    e = env(cfg)
    l = e.k - 1
    steps9 = [0, 1, 2, 64, 65, 66, 128, 129, 130]
    for st in steps9 + [-3]:
        if st:
            e.key_for(st)
    img, x1, x2 = e.rand(2, l), e.rand(2, l), e.rand(2, l)
    I, X1, X2 = (e.g.upload_ct(a, 2.0 ** 20) for a in (img, x1, x2))
    wts = [[_rand_pt(e, l) for _ in steps9] for _ in range(2)]
    W = [[e.g.upload_pt(w, 2.0 ** 10) for w in row] for row in wts]
    pool = [_rand_pt(e, l) for _ in range(4)]
    P = [e.g.upload_pt(w, 2.0 ** 10) for w in pool]
    windows = [
        ([(I, st) for st in steps9], W),                                             # two sums, unrotated term first
        ([(X1, 1), (X1, 65), (X1, 0), (X1, -3)], [P]),                               # one sum, unrotated term in the middle
        ([(X2, 1), (X2, 2), (X2, 129)], [[None, None, None]]),                       # weights of 1 (a rotate-and-sum)
        ([(X1, 2), (X2, 64)], [[P[0], P[1]], [P[2], P[3]]]),                        # two sources in one window
    ]
    outs = e.g.rotate_weighted_sums(windows)
    ref = (_window_oracle(e, [(img, st) for st in steps9], wts)
           + _window_oracle(e, [(x1, 1), (x1, 65), (x1, 0), (x1, -3)], [pool])
           + _window_oracle(e, [(x2, 1), (x2, 2), (x2, 129)], [[None] * 3])
           + _window_oracle(e, [(x1, 2), (x2, 64)], [[pool[0], pool[1]], [pool[2], pool[3]]]))
    assert len(outs) == len(ref) == 6
    for i, (o, r) in enumerate(zip(outs, ref)):
        size, limbs, scale = o.info()
        assert (size, limbs) == (2, l) and scale == (2.0 ** 30 if i != 3 else 2.0 ** 20), (i, o.info())
        assert np.array_equal(o.download(), r), f"sum {i}"


def _uniform_pt(e, l):
    """a uniform plaintext: one residue per limb in every slot (evah_pt_uniform, the encoding of a scalar constant)"""
    vals = np.array([int(e.rng.integers(0, e.primes[i])) for i in range(l)], dtype=np.uint64)
    return np.repeat(vals[:, None], e.N, axis=1), vals


@pytest.mark.parametrize("cfg", [CONFIGS[0], CONFIGS[2], CONFIGS[3], CONFIGS[4], CONFIGS[5], CONFIGS[6], CONFIGS[7]],
                         ids=lambda c: f"N{c[0]}_k{len(c[1])}")
def test_window_sums_with_uniform_weights_bit_exact(cfg):
    """r6: windows whose rotated terms are weighted by uniform plaintexts (EVA's scalar filter taps) run ONE forward transform
    per sum in their mod-down (ntt_window_lin.hip.h).  Same words as rotate -> multiply_plain -> add one by one: two sums over
    nine taps, one sum with the unrotated term in the middle and a general plaintext on it, weights of 1, two sources, and —
    in the same call — a window with general weights (which keeps the per-rotation form)."""
    e = env(cfg)
    l = e.k - 1
    steps9 = [0, 1, 2, 64, 65, 66, 128, 129, 130]
    for st in steps9 + [-3]:
        if st:
            e.key_for(st)
    img, x1, x2 = e.rand(2, l), e.rand(2, l), e.rand(2, l)
    I, X1, X2 = (e.g.upload_ct(a, 2.0 ** 20) for a in (img, x1, x2))
    uw = [[_uniform_pt(e, l) for _ in steps9] for _ in range(2)]
    W = [[e.g.uniform_pt(v, 2.0 ** 10) for _, v in row] for row in uw]
    wts = [[full for full, _ in row] for row in uw]
    up = [_uniform_pt(e, l) for _ in range(4)]
    P = [e.g.uniform_pt(v, 2.0 ** 10) for _, v in up]
    pool = [full for full, _ in up]
    gen = _rand_pt(e, l)  # a general plaintext on the UNROTATED term only
    G = e.g.upload_pt(gen, 2.0 ** 10)
    gw = [_rand_pt(e, l) for _ in range(2)]
    GW = [e.g.upload_pt(w, 2.0 ** 10) for w in gw]
    windows = [
        ([(I, st) for st in steps9], W),
        ([(X1, 1), (X1, 65), (X1, 0), (X1, -3)], [[P[0], P[1], G, P[3]]]),
        ([(X2, 1), (X2, 2), (X2, 129)], [[P[1], P[2], P[3]]]),
        ([(X1, 2), (X2, 64)], [[P[0], P[1]], [P[2], P[3]]]),
        ([(X2, 66), (X2, 130)], [GW]),
        ([(X2, 1), (X2, 2)], [[None, None]]),
    ]
    outs = e.g.rotate_weighted_sums(windows)
    ref = (_window_oracle(e, [(img, st) for st in steps9], wts)
           + _window_oracle(e, [(x1, 1), (x1, 65), (x1, 0), (x1, -3)], [[pool[0], pool[1], gen, pool[3]]])
           + _window_oracle(e, [(x2, 1), (x2, 2), (x2, 129)], [[pool[1], pool[2], pool[3]]])
           + _window_oracle(e, [(x1, 2), (x2, 64)], [[pool[0], pool[1]], [pool[2], pool[3]]])
           + _window_oracle(e, [(x2, 66), (x2, 130)], [gw])
           + _window_oracle(e, [(x2, 1), (x2, 2)], [[None, None]]))
    assert len(outs) == len(ref) == 8
    for i, (o, r) in enumerate(zip(outs, ref)):
        assert np.array_equal(o.download(), r), f"sum {i}"
    # the same windows with the linear form switched off give the same words (A/B of the two forms)
    os.environ["EVAH_WIN_LINEAR"] = "0"
    try:
        x = Env(*cfg)
        for st in steps9 + [-3]:
            if st:
                x.g.upload_galois_key(x.g.galois_elt_from_step(st), e.keys[st])
        Ix = x.g.upload_ct(img, 2.0 ** 20)
        Wx = [[x.g.uniform_pt(v, 2.0 ** 10) for _, v in row] for row in uw]
        outs0 = x.g.rotate_weighted_sums([([(Ix, st) for st in steps9], Wx)])
        for o, r in zip(outs0, ref[:2]):
            assert np.array_equal(o.download(), r)
        x.g.close()
    finally:
        del os.environ["EVAH_WIN_LINEAR"]


def test_a_rewritten_uniform_plaintext_is_not_treated_as_uniform():
    """evah_pt_write replaces the words of a handle: the uniform mark must go with them"""
    e = env(CONFIGS[2])
    l = e.k - 1
    for st in (1, 2):
        e.key_for(st)
    a = e.rand(2, l)
    A = e.g.upload_ct(a, 2.0 ** 20)
    _, v = _uniform_pt(e, l)
    w = [e.g.uniform_pt(v, 2.0 ** 10) for _ in range(2)]
    gen = _rand_pt(e, l)
    w[1].write(gen)
    outs = e.g.rotate_weighted_sums([([(A, 1), (A, 2)], [w])])
    full = np.repeat(v[:, None], e.N, axis=1)
    ref = _window_oracle(e, [(a, 1), (a, 2)], [[full, gen]])
    assert np.array_equal(outs[0].download(), ref[0])


def test_window_sums_general_shapes_take_the_unfused_path():
    """Three sums over one window and two unrotated terms are outside the fused kernel's tables: same results."""
    e = env(CONFIGS[2])
    l = e.k - 1
    for st in (1, 65, -3):
        e.key_for(st)
    a = e.rand(2, l)
    A = e.g.upload_ct(a, 2.0 ** 20)
    wts = [[_rand_pt(e, l) for _ in range(4)] for _ in range(3)]
    W = [[e.g.upload_pt(w, 2.0 ** 10) for w in row] for row in wts]
    terms = [(A, 1), (A, 0), (A, 65), (A, 0)]
    outs = e.g.rotate_weighted_sums([(terms, W)])
    ref = _window_oracle(e, [(a, st) for _, st in terms], wts)
    for o, r in zip(outs, ref):
        assert np.array_equal(o.download(), r)
    with pytest.raises(Exception, match="scale mismatch"):
        e.g.rotate_weighted_sums([([(A, 1), (A, 65)], [[W[0][0], None]])])


@pytest.mark.parametrize("sums", [1, 2])
@pytest.mark.parametrize("case", ["transparent", "several_zeros"])
@pytest.mark.parametrize("cfg", [CONFIGS[2], CONFIGS[4]], ids=lambda c: f"N{c[0]}")
def test_uniform_weight_window_sums_with_zero_digit_coefficients(cfg, case, sums):
    """r6: the same two cases through the linear mod-down (uniform weights): k_hoist_fix's corrections land in the products
    the weighted sums are formed from, the guarded fallback recomputes a transparent source; also as a batched handle of
    three instances with zeros in two of them"""
    e = env(cfg)
    l = e.k - 1
    steps = [0, 1, 65, -3]
    for st in steps[1:]:
        e.key_for(st)
    insts = []
    for b in range(3):
        a = e.rand(2, l)
        if b != 1:
            if case == "transparent":
                a[1] = 0
            else:
                for limb, count in ((0, 5), (min(2, l - 1), 3), (l - 1, 1)):
                    x = e.rng.integers(1, e.primes[limb], size=e.N, dtype=np.uint64)
                    x[e.rng.choice(e.N, size=count, replace=False)] = 0
                    a[1][limb] = e.o.ntt(limb, x)
        insts.append(a)
    uw = [[_uniform_pt(e, l) for _ in steps] for _ in range(sums)]
    W = [[e.g.uniform_pt(v, 2.0 ** 10) for _, v in row] for row in uw]
    wts = [[full for full, _ in row] for row in uw]
    # one instance
    A = e.g.upload_ct(insts[0], 2.0 ** 20)
    outs = e.g.rotate_weighted_sums([([(A, st) for st in steps], W)])
    for o, r in zip(outs, _window_oracle(e, [(insts[0], st) for st in steps], wts)):
        assert np.array_equal(o.download(), r), case
    # a batched handle of the three
    H = e.g.stack([e.g.upload_ct(a, 2.0 ** 20) for a in insts])
    outs = e.g.rotate_weighted_sums([([(H, st) for st in steps], W)])
    for f, o in enumerate(outs):
        for b in range(3):
            want = _window_oracle(e, [(insts[b], st) for st in steps], wts)[f]
            assert np.array_equal(o.unstack(b).download(), want), (case, f, b)


@pytest.mark.parametrize("case", ["transparent", "several_zeros"])
def test_window_sums_with_zero_digit_coefficients(case):
    """k_hoist_fix's correction and, beyond its capacity, the guarded fallback (unhoisted rotations + k_window_sums)"""
    e = env(CONFIGS[2])
    l = e.k - 1
    steps = [0, 1, 65, -3]
    for st in steps[1:]:
        e.key_for(st)
    a = e.rand(2, l)
    if case == "transparent":
        a[1] = 0
    else:
        for limb, count in ((0, 5), (2, 3), (l - 1, 1)):
            x = e.rng.integers(1, e.primes[limb], size=e.N, dtype=np.uint64)
            x[e.rng.choice(e.N, size=count, replace=False)] = 0
            a[1][limb] = e.o.ntt(limb, x)
    A = e.g.upload_ct(a, 2.0 ** 20)
    wts = [[_rand_pt(e, l) for _ in steps] for _ in range(2)]
    W = [[e.g.upload_pt(w, 2.0 ** 10) for w in row] for row in wts]
    outs = e.g.rotate_weighted_sums([([(A, st) for st in steps], W)])
    for o, r in zip(outs, _window_oracle(e, [(a, st) for st in steps], wts)):
        assert np.array_equal(o.download(), r), case


def test_window_sums_of_a_batched_handle():
    """B instances of one window (config 4: a batch of DAG instances): more pairs than one launch set holds"""
    e = env(CONFIGS[3])
    l = e.k - 1
    B = 24
    steps = [0, 1, 2, 64, 65, 66, 128, 129, 130]
    for st in steps[1:]:
        e.key_for(st)
    inst = [e.rand(2, l) for _ in range(B)]
    x = e.rng.integers(1, e.primes[1], size=e.N, dtype=np.uint64)
    x[e.rng.choice(e.N, size=2, replace=False)] = 0
    inst[5][1][1] = e.o.ntt(1, x)
    H = e.g.upload_ct_batch(np.stack(inst), 2.0 ** 30)
    wts = [[_rand_pt(e, l) for _ in steps] for _ in range(2)]
    W = [[e.g.upload_pt(w, 2.0 ** 10) for w in row] for row in wts]
    outs = e.g.rotate_weighted_sums([([(H, st) for st in steps], W)])
    got = [o.download() for o in outs]
    for b in (0, 5, 9, B - 1):
        ref = _window_oracle(e, [(inst[b], st) for st in steps], wts)
        for s in range(2):
            assert np.array_equal(got[s][b], ref[s]), (b, s)


def test_window_sums_fused_path_is_the_one_that_runs():
    """Launch accounting: fused = no weighted-sum launch and no permuted copy of c0; EVAH_WIN_FUSE=0 = the same bits
    through rotate_pairs + weighted_sum."""
    cfg = CONFIGS[2]
    e = env(cfg)
    l = e.k - 1
    steps = [0, 1, 65, -3]
    a = e.rand(2, l)
    wts = [[_rand_pt(e, l) for _ in steps] for _ in range(2)]

    def run(ctx_env):
        for st in steps[1:]:
            ctx_env.g.upload_galois_key(ctx_env.g.galois_elt_from_step(st), e.key_for(st))
        A = ctx_env.g.upload_ct(a, 2.0 ** 20)
        W = [[ctx_env.g.upload_pt(w, 2.0 ** 10) for w in row] for row in wts]
        ctx_env.g.rotate_weighted_sums([([(A, st) for st in steps], W)])  # tables of the first use
        ctx_env.g.profile(True)
        ctx_env.g.profile_reset()
        outs = ctx_env.g.rotate_weighted_sums([([(A, st) for st in steps], W)])
        ctx_env.g.sync()
        prof = ctx_env.g.profile_get()
        ctx_env.g.profile(False)
        return [o.download() for o in outs], prof

    def with_env(settings):
        saved = {k: os.environ.get(k) for k in settings}
        os.environ.update(settings)
        try:
            return Env(*cfg)
        finally:
            for k, v in saved.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v

    # both contexts pin the knobs they compare (the suite is also run with the knobs set globally)
    fused, pf = run(with_env({"EVAH_WIN_FUSE": "1", "EVAH_FOLD_PA": "1", "EVAH_HOIST": "1", "EVAH_FB_PERSIST": "1"}))
    unfused, pu = run(with_env({"EVAH_WIN_FUSE": "0", "EVAH_FOLD_PA": "1", "EVAH_HOIST": "1", "EVAH_FB_PERSIST": "1"}))
    for x, y in zip(fused, unfused):
        assert np.array_equal(x, y)
    # unfused: the weighted sums are elementwise launches of their own (the guarded fallback's launches are counted on
    # both sides); fused: the sums ride the mod-down's second pass
    assert pf["elementwise"][0] < pu["elementwise"][0], (pf, pu)


def test_fallback_of_a_batched_handle_spans_several_chunks():
    """One transparent instance among B: every chunk of the set takes the exact fallback (k_rot_fallback, one persistent
    launch per chunk), the other instances' results must come out the same either way."""
    e = env(CONFIGS[2])
    l = e.k - 1
    B = 24
    inst = [e.rand(2, l) for _ in range(B)]
    inst[3][1] = 0
    steps = [1, 2, 65, -3, -64]
    for st in steps:
        e.key_for(st)
    outs = e.g.rotate_many(e.g.upload_ct_batch(np.stack(inst), 2.0 ** 30), steps)
    for st, o in zip(steps, outs):
        got = o.download()
        for b in (0, 3, 11, B - 1):
            assert np.array_equal(got[b], e.o.rotate(inst[b], st, e.keys[st])), f"step {st} instance {b}"


def test_fallback_forms_agree():
    """EVAH_FB_PERSIST=0 (the guarded unhoisted launches) and the persistent kernel: same bits on a transparent source,
    for a plain set and for a window with two sums."""
    cfg = CONFIGS[3]
    e = env(cfg)
    l = e.k - 1
    steps = [0, 1, 65, -3]
    a = e.rand(2, l)
    a[1] = 0
    wts = [[_rand_pt(e, l) for _ in steps] for _ in range(2)]
    res = []
    for persist in ("1", "0"):
        old = os.environ.get("EVAH_FB_PERSIST")
        os.environ["EVAH_FB_PERSIST"] = persist
        try:
            x = Env(*cfg)
        finally:
            if old is None:
                os.environ.pop("EVAH_FB_PERSIST", None)
            else:
                os.environ["EVAH_FB_PERSIST"] = old
        for st in steps[1:]:
            x.g.upload_galois_key(x.g.galois_elt_from_step(st), e.key_for(st))
        A = x.g.upload_ct(a, 2.0 ** 20)
        W = [[x.g.upload_pt(w, 2.0 ** 10) for w in row] for row in wts]
        rot = [o.download() for o in x.g.rotate_many(A, steps[1:])]
        sums = [o.download() for o in x.g.rotate_weighted_sums([([(A, st) for st in steps], W)])]
        res.append(rot + sums)
    for u, v in zip(*res):
        assert np.array_equal(u, v)
    for st, got in zip(steps[1:], res[0]):
        assert np.array_equal(got, e.o.rotate(a, st, e.keys[st]))


def _env_with(cfg, **knobs):
    old = {k: os.environ.get(k) for k in knobs}
    os.environ.update({k: str(v) for k, v in knobs.items()})
    try:
        return Env(*cfg)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


@pytest.mark.parametrize("grid", [1, 3, 2000])
def test_fallback_is_independent_of_its_grid(grid):
    """The persistent fallback orders its phases by tickets, not by a barrier over resident workgroups
    (rot_fallback.hip.h): ONE workgroup walks every chunk of every phase alone, three share them, two thousand (far
    more than fit the chip at once) compete for them — the same bits every time."""
    cfg = CONFIGS[2]
    e = env(cfg)
    x = _env_with(cfg, EVAH_FB_GRID=grid)
    l = e.k - 1
    steps = [0, 1, 65, -3]
    a = e.rand(2, l)
    a[1] = 0
    for st in steps[1:]:
        x.g.upload_galois_key(x.g.galois_elt_from_step(st), e.key_for(st))
    wts = [[_rand_pt(e, l) for _ in steps] for _ in range(2)]
    A = x.g.upload_ct(a, 2.0 ** 20)
    W = [[x.g.upload_pt(w, 2.0 ** 10) for w in row] for row in wts]
    for st, o in zip(steps[1:], x.g.rotate_many(A, steps[1:])):
        assert np.array_equal(o.download(), e.o.rotate(a, st, e.keys[st])), f"grid {grid}: step {st}"
    sums = [o.download() for o in x.g.rotate_weighted_sums([([(A, st) for st in steps], W)])]
    want = _window_oracle(e, [(a, st) for st in steps], wts)
    for got, w in zip(sums, want):
        assert np.array_equal(got, w), f"grid {grid}: window sums"


def test_two_active_fallbacks_on_two_queues_under_load():
    """Two ACTIVE fallbacks (transparent sources: every digit coefficient is zero) on two issue queues of one device
    state while a third queue of the same GPU is kept full with the N = 2^16, L = 10 op-triple (32 triples per launch
    set: kernels of ~1 ms that fill every CU).  The r4 fallback was a grid-wide spin barrier that relied on all of its
    workgroups being resident, which co-running queues do not guarantee; the ticket-ordered form needs no such thing.
    Everything is enqueued before anything is waited for; every result is compared with the oracle."""
    cfg = CONFIGS[3]
    e = env(cfg)
    x = _env_with(cfg)
    q = [x.g.fork(), x.g.fork()]
    l = e.k - 1
    steps = [1, 65, -3, 2]
    for st in steps:
        x.g.upload_galois_key(x.g.galois_elt_from_step(st), e.key_for(st))
    srcs = []
    for i in range(2):
        a = e.rand(2, l)
        a[1] = 0
        srcs.append(a)
    # the load: op-triples on a context of their own (another N), 3 launch sets of 32
    N2, l2 = 65536, 10
    primes2 = po.coeff_modulus_create(N2, [60] * (l2 + 1))
    big = backend.Context(N2, primes2)
    rng = np.random.default_rng(11)
    rand2 = lambda prefix, nl: np.stack([rng.integers(0, primes2[i], size=prefix + (N2,), dtype=np.uint64)  # noqa: E731
                                         for i in range(nl)], axis=len(prefix))
    key2 = rand2((l2, 2), l2 + 1)
    big.upload_relin_key(key2)
    ha, hb = rand2((2,), l2), rand2((2,), l2)
    As = [big.upload_ct(ha, 2.0 ** 40) for _ in range(32)]
    Bs = [big.upload_ct(hb, 2.0 ** 40) for _ in range(32)]
    handles = [qq.upload_ct(a, 2.0 ** 20) for qq, a in zip(q, srcs)]
    for qq in q:
        qq.sync()
    big.sync()
    rounds = []
    for _ in range(3):
        trip = big.multiply_relinearize_rescale_many(As, Bs, 60)
        rot = [qq.rotate_many(h, steps) for qq, h in zip(q, handles)]
        rounds.append((trip, rot))
    for trip, rot in rounds:
        for qi, outs in enumerate(rot):
            for st, o in zip(steps, outs):
                assert np.array_equal(o.download(), e.o.rotate(srcs[qi], st, e.keys[st])), f"queue {qi}: step {st}"
    want = po.Oracle(N2, primes2).op_triple(ha, hb, key2)
    for trip, _ in rounds:
        assert np.array_equal(trip[0].download(), want) and np.array_equal(trip[31].download(), want)
    for qq in q:
        qq.close()
    big.close()


def test_hoisting_tables_that_do_not_fit_degrade_to_the_unhoisted_path():
    """A hoisted set keeps a permuted copy of every Galois key it uses and a constant per (element, level); when the
    device has no room for one (EVAH_HOIST_TABLE_FAIL=1 refuses every new table) the set — a rotation set, sibling
    rotations of several sources, a convolution window — must run unhoisted with the same bits instead of failing,
    and the copies that exist are accounted for (evah_ctx_key_bytes_detail)."""
    cfg = CONFIGS[2]
    e = env(cfg)
    l = e.k - 1
    steps = [1, 65, -3]
    a, b = e.rand(2, l), e.rand(2, l)
    wts = [[_rand_pt(e, l) for _ in [0] + steps]]
    for fail in (1, 0):
        x = _env_with(cfg, EVAH_HOIST_TABLE_FAIL=fail)
        for st in steps:
            x.g.upload_galois_key(x.g.galois_elt_from_step(st), e.key_for(st))
        A, B = x.g.upload_ct(a, 2.0 ** 20), x.g.upload_ct(b, 2.0 ** 20)
        for st, o in zip(steps, x.g.rotate_many(A, steps)):
            assert np.array_equal(o.download(), e.o.rotate(a, st, e.keys[st])), f"fail={fail}: rotate_many step {st}"
        outs = x.g.rotate_pairs([A, B, A, B], [1, 1, 65, -3])
        for (src, st), o in zip([(a, 1), (b, 1), (a, 65), (b, -3)], outs):
            assert np.array_equal(o.download(), e.o.rotate(src, st, e.keys[st])), f"fail={fail}: rotate_pairs step {st}"
        W = [[x.g.upload_pt(w, 2.0 ** 10) for w in row] for row in wts]
        sums = [o.download() for o in x.g.rotate_weighted_sums([([(A, st) for st in [0] + steps], W)])]
        for got, w in zip(sums, _window_oracle(e, [(a, st) for st in [0] + steps], wts)):
            assert np.array_equal(got, w), f"fail={fail}: window sums"
        words, split, perm = x.g.key_bytes_detail()
        assert words == x.g.key_bytes() == 3 * l * 2 * e.k * e.N * 8
        # one permuted copy per key a hoisted set used; r6: in blocks of (2 digits + 1 pad) x 2 KiB per (prime row, 256 coefficients)
        assert perm == (0 if fail else 3 * (2 * l + 1) * e.k * e.N * 8), (fail, words, split, perm)
