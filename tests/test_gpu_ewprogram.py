"""evah_elementwise_program (eva_amd/csrc/ewprogram.hip): a straight-line program of elementwise evaluator calls — add,
sub, negate, multiply, square, add_plain, sub_plain, multiply_plain as SEALExecutor dispatches them
(/root/reference/eva/seal/seal_executor.h:114-175, :191-195) — evaluated in ONE launch must return, bit for bit, what
the oracle (and the separate entry points) return for the same calls one by one: random programs, Harris' response and
Sobel's tail as written in /root/reference/examples/image_processing.py:39-100, batched handles, mod-switched views,
programs too large for one launch, and the error cases with the separate entry points' messages.  The scheduler side
(evah_execute builds these programs itself) is pinned at DAG level: EVAH_EW_FUSE=0 and 1 give the same ciphertexts as
the oracle walk, with fewer elementwise launches."""
import os

import numpy as np
import pytest

from eva_amd import backend
from oracle import pyoracle as po

pytestmark = pytest.mark.gpu
NEG, ADD, SUB, MUL = 10, 11, 12, 13


class Env:
    def __init__(self, N, bits):
        self.N, self.primes = N, po.coeff_modulus_create(N, bits)
        self.k, self.l = len(self.primes), len(self.primes) - 1
        self.o = po.Oracle(N, self.primes)
        self.g = backend.Context(N, self.primes)
        self.rng = np.random.default_rng(5 * N + len(bits))

    def ct(self, size, l=None):
        l = l or self.l
        return np.stack([np.stack([self.rng.integers(0, self.primes[i], size=self.N, dtype=np.uint64) for i in range(l)]) for _ in range(size)])

    def pt(self, l=None):
        return self.ct(1, l)[0]


_envs = {}


def env(cfg):
    key = (cfg[0], tuple(cfg[1]))
    if key not in _envs:
        _envs[key] = Env(*cfg)
    return _envs[key]


def oracle_run(e, inputs, ops):
    """inputs: [("ct", words, scale) | ("pt", words, scale)] -> every value of the program by the oracle, op by op, with
    SEALExecutor's dispatch rules"""
    vals = [(k, w.copy(), s) for k, w, s in inputs]
    for op, a, b in ops:
        if op == NEG:
            k, w, s = vals[a]
            vals.append(("ct", e.o.negate(w), s))
            continue
        if op != SUB and vals[a][0] == "pt":
            a, b = b, a
        (ka, wa, sa), (kb, wb, sb) = vals[a], vals[b]
        assert ka == "ct"
        if kb == "ct":
            if op == ADD: r, s = e.o.add(wa, wb), sa
            elif op == SUB: r, s = e.o.sub(wa, wb), sa
            elif a == b: r, s = e.o.square(wa), sa * sa
            else: r, s = e.o.multiply(wa, wb), sa * sb
        else:
            if op == ADD: r, s = e.o.add_plain(wa, wb), sa
            elif op == SUB: r, s = e.o.sub_plain(wa, wb), sa
            else: r, s = e.o.multiply_plain(wa, wb), sa * sb
        vals.append(("ct", r, s))
    return vals


def random_program(rng, kinds, sizes, scales, n_ops, scale_cap):
    """a valid random program over inputs of the given kinds / sizes / scales (log2)"""
    kinds, sizes, scales = list(kinds), list(sizes), list(scales)
    ops = []
    while len(ops) < n_ops:
        op = int(rng.choice([NEG, ADD, ADD, SUB, MUL, MUL]))
        n = len(kinds)
        a, b = int(rng.integers(n)), int(rng.integers(n))
        if op == NEG:
            if kinds[a] != "ct": continue
            ops.append((op, a, a)); kinds.append("ct"); sizes.append(sizes[a]); scales.append(scales[a]); continue
        ca, cb = (b, a) if (op != SUB and kinds[a] == "pt") else (a, b)
        if kinds[ca] != "ct": continue
        if op == MUL:
            if scales[ca] + scales[cb] > scale_cap: continue
            if kinds[cb] == "ct" and (sizes[ca] != 2 or sizes[cb] != 2): continue
            ops.append((op, a, b)); kinds.append("ct"); sizes.append(3 if kinds[cb] == "ct" else sizes[ca]); scales.append(scales[ca] + scales[cb])
        else:
            if scales[ca] != scales[cb]: continue
            ops.append((op, a, b)); kinds.append("ct")
            sizes.append(max(sizes[ca], sizes[cb]) if kinds[cb] == "ct" else sizes[ca]); scales.append(scales[ca])
    return ops


def upload(e, inputs, g=None):
    g = g or e.g
    return [g.upload_ct(w, s) if k == "ct" else g.upload_pt(w, s) for k, w, s in inputs]


def uniform_input(e, scale, l=None):
    """a uniform plaintext (evah_pt_uniform: one residue per limb in every slot) -> (oracle-side input, per-limb values)"""
    l = l or e.l
    uv = np.array([int(e.rng.integers(0, e.primes[i])) for i in range(l)], dtype=np.uint64)
    return ("pt", np.repeat(uv[:, None], e.N, axis=1), scale), uv


@pytest.mark.parametrize("on", ["1", "0"])
def test_uniform_plaintexts_multiply_as_scalars(on, monkeypatch):
    """r6: a product with a uniform plaintext (EVA's scalar constants) is a Shoup product with a per-limb scalar inside the
    interpreter (EW_MULU: the plaintext's polynomial is never loaded) — the same canonical residues as multiply_plain;
    mixed with general plaintexts, a mod-switched uniform plaintext (fewer limbs), zero and q - 1 as scalars, batched
    handles, the throughput-sized two-coefficient launch, and more distinct scalars than one launch holds (separate calls)"""
    monkeypatch.setenv("EVAH_EW_UNIFORM", on)
    for cfg, batch in (((4096, [60, 20, 60, 60]), 1), ((8192, [60, 30, 60, 60, 60]), 3), ((32768, [60] * 5), 8)):
        e = Env(*cfg)
        l = e.l
        us = [uniform_input(e, 2.0 ** 10) for _ in range(3)]
        us[1][1][0] = 0                      # the scalar 0
        us[2][1][:] = np.array(e.primes[:l], dtype=np.uint64) - 1  # q - 1 in every limb
        for (k, w, sc), uv in us:
            w[:] = np.repeat(uv[:, None], e.N, axis=1)
        inputs = [("ct", e.ct(2), 2.0 ** 10), ("ct", e.ct(3), 2.0 ** 10), ("pt", e.pt(), 2.0 ** 10)] + [u[0] for u in us]
        #       6: a*u0   7: b*u1   8: a*pt   9: 6+8     10: 9*u2   11: 7*u0   12: u1*a (plaintext first)  13: 12 - 6
        ops = [(MUL, 0, 3), (MUL, 1, 4), (MUL, 0, 2), (ADD, 6, 8), (MUL, 9, 5), (MUL, 7, 3), (MUL, 4, 0), (SUB, 12, 6)]
        ref = oracle_run(e, inputs, ops)

        def handles(b):
            hs = []
            for j, (k, w, sc) in enumerate(inputs):
                if k == "ct":
                    hs.append(e.g.upload_ct(w, sc) if b == 1 else e.g.upload_ct_batch(np.stack([w] * b), sc))
                elif j >= 3:
                    hs.append(e.g.uniform_pt(us[j - 3][1], sc))
                else:
                    hs.append(e.g.upload_pt(w, sc))
            return hs
        outs = [10, 11, 13]
        got = e.g.elementwise_program(handles(batch), ops, outs)
        for v, h in zip(outs, got):
            d = h.download()
            for b in range(batch):
                assert np.array_equal(d[b] if batch > 1 else d, ref[v][1]), f"N={cfg[0]} value {v} instance {b}"
        # more distinct uniform plaintexts than one launch's scalar table (16): the separate entry points, same words
        many = [uniform_input(e, 2.0 ** 2) for _ in range(18)]
        inputs2 = [("ct", e.ct(2), 2.0 ** 2)] + [m[0] for m in many]
        ops2 = [(MUL, 0 if j == 0 else 19 + j - 1, 1 + j) for j in range(18)]
        ref2 = oracle_run(e, inputs2, ops2)
        h2 = [e.g.upload_ct(inputs2[0][1], inputs2[0][2])] + [e.g.uniform_pt(m[1], 2.0 ** 2) for m in many]
        got2 = e.g.elementwise_program(h2, ops2, [19 + 17])[0]
        assert np.array_equal(got2.download(), ref2[-1][1])
        # a uniform plaintext of fewer limbs on a mod-switched view
        if l >= 3:
            (k, w, sc), uv = uniform_input(e, 2.0 ** 10, l - 1)
            big = e.ct(2)
            V = e.g.mod_switch(e.g.upload_ct(big, 2.0 ** 10))
            got3 = e.g.elementwise_program([V, e.g.uniform_pt(uv, sc)], [(MUL, 0, 1), (ADD, 2, 2)], [3])[0]
            want = e.o.multiply_plain(e.o.mod_switch(big), w)
            assert np.array_equal(got3.download(), e.o.add(want, want))
        e.g.close()


@pytest.mark.parametrize("cfg", [(1024, [30, 30, 31]), (4096, [60, 20, 60, 60]), (8192, [60, 60, 60, 60, 60]), (32768, [60, 60, 60])],
                         ids=lambda c: f"N{c[0]}_k{len(c[1])}")
def test_random_programs_bit_exact(cfg):
    e = env(cfg)
    cap = sum(cfg[1][:-1]) - 2
    for seed in range(int(os.environ.get("EVA_FUZZ_SEEDS", "6"))):  # EVA_FUZZ_SEEDS=n: a longer one-off run
        rng = np.random.default_rng(100 * seed + cfg[0])
        inputs = [("ct", e.ct(2), 2.0 ** 10), ("ct", e.ct(2), 2.0 ** 10), ("ct", e.ct(3), 2.0 ** 10), ("ct", e.ct(2), 2.0 ** 20),
                  ("pt", e.pt(), 2.0 ** 10), ("pt", e.pt(), 2.0 ** 20)]
        ops = random_program(rng, [i[0] for i in inputs], [len(i[1]) if i[0] == "ct" else 1 for i in inputs], [10, 10, 10, 20, 10, 20],
                             int(rng.integers(3, 14 if seed < 64 else 40)), cap)
        ref = oracle_run(e, inputs, ops)
        n_in = len(inputs)
        outs = sorted(set([n_in + len(ops) - 1] + [int(x) for x in rng.integers(n_in, n_in + len(ops), size=2)]))
        H = upload(e, inputs)
        got = e.g.elementwise_program(H, ops, outs)
        for v, h in zip(outs, got):
            assert np.array_equal(h.download(), ref[v][1]), f"seed {seed}: value {v} of {ops}"
            assert h.scale == ref[v][2] and h.size == len(ref[v][1])


def test_harris_response_and_sobel_tail_as_one_launch():
    """det - k trace^2 of Harris (examples/image_processing.py:92-100: seven calls on three ciphertexts) and the tail of
    Sobel's polynomial (:39-63) exactly as the compiled programs hold them; the launch profile shows ONE elementwise launch"""
    e = env((8192, [60, 60, 60, 60, 60]))
    sxx, syy, sxy = (("ct", e.ct(2), 2.0 ** 30) for _ in range(3))
    k, one = ("pt", e.pt(), 2.0 ** 30), ("pt", e.pt(), 2.0 ** 30)
    inputs = [sxx, syy, sxy, k, one]
    #       5: Sxx*Syy   6: trace     7: Sxy^2      8: det       9: k*trace   10: (k trace) trace  11: det*1  12: response
    ops = [(MUL, 0, 1), (ADD, 0, 1), (MUL, 2, 2), (SUB, 5, 7), (MUL, 3, 6), (MUL, 9, 6), (MUL, 8, 4), (SUB, 11, 10)]
    ref = oracle_run(e, inputs, ops)
    H = upload(e, inputs)
    e.g.profile(True)
    e.g.profile_reset()
    got = e.g.elementwise_program(H, ops, [12])[0]
    e.g.sync()
    prof = e.g.profile_get()
    e.g.profile(False)
    assert np.array_equal(got.download(), ref[12][1]) and got.size == 3 and got.scale == ref[12][2]
    assert prof["elementwise"][0] == 1, prof
    # Sobel's tail: products with plaintexts, a size-2 + size-3 sum, two ciphertext products (values 7 .. 15; scales
    # 60, 60, 90, 60, 120, 90, 90, 120, 120 bits)
    x, x2 = ("ct", e.ct(2), 2.0 ** 30), ("ct", e.ct(2), 2.0 ** 60)
    inputs = [x, x2] + [("pt", e.pt(), 2.0 ** 30) for _ in range(5)]
    ops = [(MUL, 0, 2), (MUL, 3, 0), (MUL, 8, 0), (MUL, 4, 0), (MUL, 10, 1), (MUL, 7, 5), (ADD, 12, 9), (MUL, 13, 6), (ADD, 14, 11)]
    ref = oracle_run(e, inputs, ops)
    got = e.g.elementwise_program(upload(e, inputs), ops, [15, 13])
    assert np.array_equal(got[0].download(), ref[15][1]) and np.array_equal(got[1].download(), ref[13][1])
    assert got[0].size == 3 and got[0].scale == 2.0 ** 120


def test_outputs_that_are_inputs_or_share_polynomials_with_them():
    e = env((4096, [60, 20, 60, 60]))
    a2, b3 = ("ct", e.ct(2), 2.0 ** 20), ("ct", e.ct(3), 2.0 ** 20)
    inputs = [a2, b3]
    ops = [(ADD, 0, 1), (SUB, 0, 1), (NEG, 1, 1)]  # size 2 (+/-) size 3: the third polynomial is b's own / its negation
    ref = oracle_run(e, inputs, ops)
    got = e.g.elementwise_program(upload(e, inputs), ops, [2, 3, 4, 1, 2])
    for v, h in zip([2, 3, 4, 1, 2], got):
        assert np.array_equal(h.download(), ref[v][1]), v


def test_batched_handles_and_mod_switched_views():
    e = env((4096, [60, 20, 60, 60]))
    B = 5
    xs, ys = [e.ct(2) for _ in range(B)], [e.ct(2) for _ in range(B)]
    w = e.pt()
    X, Y, W = e.g.upload_ct_batch(np.stack(xs), 2.0 ** 20), e.g.upload_ct_batch(np.stack(ys), 2.0 ** 20), e.g.upload_pt(w, 2.0 ** 20)
    ops = [(MUL, 0, 0), (MUL, 0, 1), (ADD, 3, 4), (MUL, 2, 1), (SUB, 5, 6)]  # x^2 + x y - w y... at matching scales
    ops[3] = (MUL, 1, 2)
    got = e.g.elementwise_program([X, Y, W], ops, [7, 5])
    d7, d5 = got[0].download(), got[1].download()
    for b in range(B):
        ref = oracle_run(e, [("ct", xs[b], 2.0 ** 20), ("ct", ys[b], 2.0 ** 20), ("pt", w, 2.0 ** 20)], ops)
        assert np.array_equal(d7[b], ref[7][1]) and np.array_equal(d5[b], ref[5][1]), b
    # views: the operands after mod_switch_to_next (same buffers, one limb fewer, the poly stride of the parent)
    a, b2 = e.ct(2), e.ct(2)
    A, Bh = e.g.mod_switch(e.g.upload_ct(a, 2.0 ** 20)), e.g.mod_switch(e.g.upload_ct(b2, 2.0 ** 20))
    wl = e.pt(e.l - 1)
    Wl = e.g.upload_pt(wl, 2.0 ** 20)
    ops = [(MUL, 0, 1), (MUL, 0, 2), (NEG, 4, 4)]
    ref = oracle_run(e, [("ct", a[:, :e.l - 1], 2.0 ** 20), ("ct", b2[:, :e.l - 1], 2.0 ** 20), ("pt", wl, 2.0 ** 20)], ops)
    got = e.g.elementwise_program([A, Bh, Wl], ops, [3, 5])
    assert np.array_equal(got[0].download(), ref[3][1]) and np.array_equal(got[1].download(), ref[5][1])


def test_a_program_beyond_one_launch_runs_as_the_separate_calls():
    """more live polynomials / instructions than one launch holds: the same results through the entry points, call by call"""
    e = env((1024, [30, 30, 31]))
    inputs = [("ct", e.ct(2), 2.0 ** 5) for _ in range(24)]
    ops = []
    n = len(inputs)
    for i in range(0, 24, 2):  # 12 products alive at once (36 polynomials), then their sum, then a long chain
        ops.append((MUL, i, i + 1))
    acc = n
    for j in range(1, 12):
        ops.append((ADD, acc, n + j))
        acc = n + len(ops) - 1
    for _ in range(60):
        ops.append((NEG, acc, acc))
        acc = n + len(ops) - 1
    ref = oracle_run(e, inputs, ops)
    got = e.g.elementwise_program(upload(e, inputs), ops, [acc, n + 3])
    assert np.array_equal(got[0].download(), ref[acc][1]) and np.array_equal(got[1].download(), ref[n + 3][1])


def test_errors_are_those_of_the_separate_entry_points():
    e = env((4096, [60, 20, 60, 60]))
    a, b = e.g.upload_ct(e.ct(2), 2.0 ** 20), e.g.upload_ct(e.ct(2), 2.0 ** 30)
    c3 = e.g.upload_ct(e.ct(3), 2.0 ** 20)
    low = e.g.mod_switch(e.g.upload_ct(e.ct(2), 2.0 ** 20))
    w = e.g.upload_pt(e.pt(), 2.0 ** 20)
    big = e.g.upload_ct(e.ct(2), 2.0 ** 100)
    cases = [([a, b], [(ADD, 0, 1)], "scale mismatch", lambda: e.g.add(a, b)),
             ([a, c3], [(MUL, 0, 1)], "size-2 operands only", lambda: e.g.multiply(a, c3)),
             ([c3], [(MUL, 0, 0)], "square supports size-2", lambda: e.g.square(c3)),
             ([a, low], [(SUB, 0, 1)], "parameter mismatch", lambda: e.g.sub(a, low)),
             ([low, w], [(MUL, 0, 1)], "encrypted and plain parameter mismatch", lambda: e.g.multiply_plain(low, w)),
             ([big, big], [(MUL, 0, 1)], "scale out of bounds", lambda: e.g.multiply(big, big)),
             ([w, a], [(SUB, 0, 1)], "Unsupported operation", None),
             ([w, w], [(ADD, 0, 1)], "Unsupported operation", None)]
    for inputs, ops, msg, direct in cases:
        with pytest.raises(backend.EvaHipError, match=msg):
            e.g.elementwise_program(inputs, ops, [len(inputs) + len(ops) - 1])
        if direct:
            with pytest.raises(backend.EvaHipError, match=msg):
                direct()
    # an error in a value nobody stores is still an error (checks run in program order, as the separate calls would)
    with pytest.raises(backend.EvaHipError, match="scale mismatch"):
        e.g.elementwise_program([a, b], [(ADD, 0, 1), (NEG, 0, 0)], [3])


def _with_env(knobs, fn):
    old = {k: os.environ.get(k) for k in knobs}
    os.environ.update({k: str(v) for k, v in knobs.items()})
    try:
        return fn()
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


@pytest.mark.parametrize("which", ["harris", "sobel"])
def test_the_scheduler_fuses_elementwise_runs_and_changes_no_bit(which):
    """evah_execute with EVAH_EW_FUSE=1 (default) and 0: the same ciphertexts as the oracle walk, fewer elementwise launches"""
    from eva.ckks import CKKSCompiler
    from eva.seal import generate_keys
    from oracle_executor import c_walk
    from eva_amd.workloads import sobel as _sobel
    from eva_amd.workloads import harris as _harris, image as _image
    if which == "harris":
        prog, N = _harris(), 8192
    else:
        prog, N = _sobel(64, 64, 4096), 8192
        prog.set_input_scales(25)
        prog.set_output_ranges(10)
    compiled, params, sig = CKKSCompiler(config={'warn_vec_size': 'false'}).compile(prog)
    params.poly_modulus_degree = N
    image = _image(4096) if which == "harris" else {'image': [((37 * i) % 256) / 255.0 for i in range(4096)]}
    launches, outs = {}, {}
    for fuse in (1, 0):
        def run():
            pub, sec = generate_keys(params, 1)
            pub.use_graphs = False
            enc = pub.encrypt(image, sig)
            pub.execute(compiled, enc)  # constants, tables
            pub.synchronize()
            pub.profile(True)
            pub.profile_reset()
            out = pub.execute(compiled, enc)
            pub.synchronize()
            prof = pub.profile_get()
            pub.profile(False)
            ref, _ = c_walk(pub, compiled, enc, threads=4)
            for name in ref:
                assert np.array_equal(out.get(name)[4], ref[name]), f"EVAH_EW_FUSE={fuse}: output {name}"
            return prof["elementwise"][0]
        launches[fuse] = _with_env({"EVAH_EW_FUSE": fuse}, run)
    assert launches[1] < launches[0], launches
    if which == "harris":
        assert launches[0] - launches[1] >= 5, launches  # the seven calls of the response are one launch


def test_a_very_long_dependent_elementwise_chain_is_cut_not_recursed():
    """r5 advisor: nothing bounded the depth of a deferred expression — 400 dependent negate / subtract ops used to build a
    400-deep graph of nodes (recursive visit, recursive destructors).  The scheduler now evaluates a chain every 48 nodes;
    the result is the oracle walk's, bit for bit."""
    from eva import EvaProgram, Input, Output
    from eva.ckks import CKKSCompiler
    from eva.seal import generate_keys
    from oracle_executor import c_walk
    prog = EvaProgram('long_chain', vec_size=1024)
    with prog:
        x, y = Input('x'), Input('y')
        acc = x
        for i in range(400):
            acc = (y - acc) if i % 2 else (-acc)
        Output('z', acc * y)
    prog.set_input_scales(30)
    prog.set_output_ranges(20)
    compiled, params, sig = CKKSCompiler(config={'warn_vec_size': 'false'}).compile(prog)
    pub, sec = generate_keys(params, 3)
    enc = pub.encrypt({'x': [i / 1024.0 for i in range(1024)], 'y': [1.0 - i / 2048.0 for i in range(1024)]}, sig)
    out = pub.execute(compiled, enc)
    ref, _ = c_walk(pub, compiled, enc, threads=4)
    for name in ref:
        assert np.array_equal(out.get(name)[4], ref[name])


@pytest.mark.parametrize("live", [2, 6], ids=["wide_256_threads", "narrow_128_threads"])
def test_throughput_sized_launches_two_coefficients_per_thread(live):
    """>= 2^20 coefficients per launch (here 2^21): the interpreter runs two coefficients per thread — 256-thread workgroups while the
    program needs at most 14 polynomial registers, 128-thread workgroups above (`live` products of 3 polynomials each alive
    at once: 6 and 18 registers)."""
    e = env((32768, [60, 60, 60]))
    B = 32  # 32768 x 2 limbs x 32 instances = 2^21 coefficients
    xs = [[e.ct(2) for _ in range(B)] for _ in range(live + 1)]
    H = [e.g.upload_ct_batch(np.stack(x), 2.0 ** 20) for x in xs]
    n_in = len(H)
    ops = [(MUL, i, i + 1) for i in range(live)]          # values n_in .. n_in + live - 1, all alive until the sums below
    acc = n_in
    for j in range(1, live):
        ops.append((ADD, acc, n_in + j))
        acc = n_in + len(ops) - 1
    ops.append((NEG, acc, acc))
    out_v = n_in + len(ops) - 1
    got = e.g.elementwise_program(H, ops, [out_v, n_in])
    d_out, d_first = got[0].download(), got[1].download()
    for b in (0, 13, B - 1):
        ref = oracle_run(e, [("ct", xs[i][b], 2.0 ** 20) for i in range(n_in)], ops)
        assert np.array_equal(d_out[b], ref[out_v][1]) and np.array_equal(d_first[b], ref[n_in][1]), b
