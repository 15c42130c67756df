"""evah_execute (the whole-DAG submit of SURVEY.md §8(b)): a compiled program lowered to the
flat evah_op list over a value table and run through the C-ABI in ONE call must give, bit for
bit, the ciphertexts the CPU oracle gets walking the same DAG (tests/oracle_executor.py), with
last-use frees, the fused Relinearize->Rescale form and batched sibling rotations exercised."""
import numpy as np
import pytest

from eva import EvaProgram, Input, Output, Op
from eva.ckks import CKKSCompiler
from eva.seal import generate_keys
from eva_amd import backend as be
from eva_amd.workloads import sobel as _sobel

pytestmark = pytest.mark.gpu


def _lower(compiled, enc_inputs, pub, g):
    """term list -> (ops, values): encrypted part only; raw (vector<double>) nodes are host work
    (seal_executor.h:63-112) and are folded here, Encode nodes are encoded by the product's host
    encoder and uploaded as plaintexts."""
    dump = compiled._dump()
    uses = {}
    for d in dump:
        for a in d["operands"]:
            uses[a] = uses.get(a, 0) + 1
    raw, values, ops = {}, {}, []
    inputs = {name: t.index for name, t in compiled.inputs.items()}
    for name in enc_inputs.names():
        kind, size, limbs, scale, data = enc_inputs.get(name)
        t = inputs[name]
        if kind == "cipher":
            values[t] = g.upload_ct(data, scale)
        elif kind == "plain":
            values[t] = g.upload_pt(data, scale)
        else:
            raw[t] = list(data) * (compiled.vec_size // len(data))
    seen = {}
    for d in dump:
        t, op, a = d["id"], d["op"], d["operands"]
        if op == Op.Input:
            continue
        if op == Op.Constant:
            raw[t] = list(d["constant"]) * (compiled.vec_size // len(d["constant"]))
            continue
        if op == Op.Encode:
            data = pub._encode(raw[a[0]], d["encode_scale"], d["encode_level"])
            values[t] = g.upload_pt(data, 2.0 ** d["encode_scale"])
            continue
        if all(x in raw for x in a):
            x = [raw[i] for i in a]
            if op == Op.Add: raw[t] = [u + v for u, v in zip(*x)]
            elif op == Op.Sub: raw[t] = [u - v for u, v in zip(*x)]
            elif op == Op.Mul: raw[t] = [u * v for u, v in zip(*x)]
            elif op == Op.Negate: raw[t] = [-u for u in x[0]]
            elif op in (Op.Rescale, Op.Relinearize, Op.ModSwitch): raw[t] = list(x[0])  # scale management of a raw value: a copy
            elif op in (Op.RotateLeftConst, Op.RotateRightConst):
                r = d["rotation"] % len(x[0])
                raw[t] = x[0][r:] + x[0][:r] if op == Op.RotateLeftConst else x[0][len(x[0]) - r:] + x[0][:len(x[0]) - r]
            elif op == Op.Output: raw[t] = x[0]
            else: raise RuntimeError("raw op not needed by these programs")
            continue
        imm = d.get("rotation", d.get("rescale_divisor", 0)) or 0
        flags = 0
        for pos, s in enumerate(a):
            seen[s] = seen.get(s, 0) + 1
        for pos, s in enumerate(a[:2]):
            # last use of an intermediate (never an input or a plaintext the caller still holds)
            if seen[s] == uses[s] and s not in values and not (pos == 1 and a[0] == a[1]):
                flags |= be.OPF_FREE_SRC0 if pos == 0 else be.OPF_FREE_SRC1
        ops.append((int(op), t, a[0], a[1] if len(a) > 1 else 0, int(imm), flags))
    outs = {name: t.index for name, t in compiled.outputs.items()}
    return ops, values, outs


def _check(prog, inputs, N=None):
    compiled, params, sig = CKKSCompiler(config={'warn_vec_size': 'false'}).compile(prog)
    if N:
        params.poly_modulus_degree = N
    pub, sec = generate_keys(params, 3)
    enc = pub.encrypt(inputs, sig)
    g = be.Context(pub.poly_modulus_degree, list(pub.primes))
    g.upload_relin_key(pub.relin_key())
    for elt, key in pub.galois_keys().items():
        g.upload_galois_key(elt, key)
    ops, values, outs = _lower(compiled, enc, pub, g)
    res = g.execute(ops, values)
    from oracle_executor import OracleExecutor
    ref = OracleExecutor(pub).execute(compiled, enc)
    for name, t in outs.items():
        if isinstance(ref[name], list):
            continue   # an output that depends on unencrypted values only: host work, not part of the submit
        assert np.array_equal(res[t].download(), ref[name].data), f"output {name} differs from the oracle walk"
        assert res[t].scale == ref[name].scale
    # every intermediate was released at its last use: only caller-placed values and outputs remain
    assert set(res) <= set(values) | set(outs.values())
    return ops


def test_execute_polynomial_and_fused_relin_rescale():
    poly = EvaProgram('p', vec_size=64)
    with poly:
        x = Input('x')
        Output('y', 3 * x ** 2 + 5 * x - 2)
    poly.set_output_ranges(20)
    poly.set_input_scales(30)
    ops = _check(poly, {'x': [i / 64.0 for i in range(64)]})
    assert any(o[0] == int(Op.Relinearize) for o in ops)


def test_execute_rotations_and_mixed_ops():
    p = EvaProgram('r', vec_size=32)
    with p:
        x, y = Input('x'), Input('y')
        s = (x << 1) + (x << 2) + (x >> 3) + (x << 0)
        Output('a', s * y - x)
        Output('b', -(x * x) + 0.5)
    p.set_output_ranges(20)
    p.set_input_scales(25)
    rng = np.random.default_rng(5)
    _check(p, {'x': list(rng.uniform(-1, 1, 32)), 'y': list(rng.uniform(-1, 1, 32))})


def test_execute_sobel_n8192():
    """BASELINE config 2 through the one-call submit"""
    sob = _sobel(64, 64, 4096)
    sob.set_input_scales(25)
    sob.set_output_ranges(10)
    img = [((37 * i) % 256) / 255.0 for i in range(4096)]
    ops = _check(sob, {'image': img}, N=8192)
    assert sum(1 for o in ops if o[0] in (int(Op.RotateLeftConst), int(Op.RotateRightConst))) >= 8


def test_execute_harris_level_batching():
    """BASELINE config 3's DAG (three independent convolution chains) through the one-call submit:
    its level scheduler batches them; results still equal the oracle walk."""
    from eva_amd.workloads import harris as _harris, image as _image
    _check(_harris(), _image(4096), N=16384)


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("EVA_FUZZ_SEEDS", "48"))))
def test_execute_random_programs(seed):
    """the generator of tests/test_gpu_fuzz.py through evah_execute"""
    from test_gpu_fuzz import _random_program
    prog, inputs = _random_program(1000 + seed, 64)
    _check(prog, inputs)


def test_execute_captured_into_a_graph_and_replayed():
    """evah_capture_begin / evah_execute / evah_capture_end: the whole program becomes one hipGraph;
    refilling the input slot (evah_ct_write) and launching the graph gives the oracle's ciphertext
    for the new input."""
    from eva.ckks import CKKSCompiler
    from eva.seal import generate_keys
    from oracle_executor import OracleExecutor
    sob = _sobel(32, 32, 1024)
    sob.set_input_scales(25)
    sob.set_output_ranges(10)
    compiled, params, sig = CKKSCompiler(config={'warn_vec_size': 'false'}).compile(sob)
    pub, sec = generate_keys(params, 4)
    g = be.Context(pub.poly_modulus_degree, list(pub.primes))
    g.upload_relin_key(pub.relin_key())
    for elt, key in pub.galois_keys().items():
        g.upload_galois_key(elt, key)
    img = lambda u: {'image': [((37 * i + 13 * u) % 256) / 255.0 for i in range(1024)]}
    enc0, enc1 = pub.encrypt(img(0), sig), pub.encrypt(img(1), sig)
    ops, values, outs = _lower(compiled, enc0, pub, g)
    # the caller keeps its inputs: strip the release flags of caller-placed slots (none are set by _lower)
    warm = g.execute(ops, dict(values))          # eager: builds permutation tables, fills the pool
    for t, h in warm.items():
        if t not in values:
            h.free()
    g.capture_begin()
    res = g.execute(ops, dict(values))
    graph = g.capture_end()
    (name, t_out), = outs.items()
    in_slot = compiled.inputs['image'].index
    for enc in (enc0, enc1, enc0):
        kind, size, limbs, scale, data = enc.get('image')
        values[in_slot].write(data)
        g.graph_launch(graph)
        ref = OracleExecutor(pub).execute(compiled, enc)
        assert np.array_equal(res[t_out].download(), ref[name].data)
    g.graph_free(graph)


def test_execute_on_batched_handles():
    """The same op list over batched input handles (three encryptions in one handle): every
    instance of every output equals the oracle walk of that instance."""
    from eva.ckks import CKKSCompiler
    from eva.seal import generate_keys
    from oracle_executor import OracleExecutor
    sob = _sobel(32, 32, 1024)
    sob.set_input_scales(25)
    sob.set_output_ranges(10)
    compiled, params, sig = CKKSCompiler(config={'warn_vec_size': 'false'}).compile(sob)
    pub, sec = generate_keys(params, 6)
    g = be.Context(pub.poly_modulus_degree, list(pub.primes))
    g.upload_relin_key(pub.relin_key())
    for elt, key in pub.galois_keys().items():
        g.upload_galois_key(elt, key)
    encs = [pub.encrypt({'image': [((37 * i + 29 * u) % 256) / 255.0 for i in range(1024)]}, sig) for u in range(3)]
    ops, values, outs = _lower(compiled, encs[0], pub, g)
    in_slot = compiled.inputs['image'].index
    datas = [e.get('image') for e in encs]
    values[in_slot] = g.upload_ct_batch(np.stack([d[4] for d in datas]), datas[0][3])
    res = g.execute(ops, values)
    (name, t_out), = outs.items()
    got = res[t_out].download()
    assert got.shape[0] == 3
    for u, enc in enumerate(encs):
        assert np.array_equal(got[u], OracleExecutor(pub).execute(compiled, enc)[name].data), f"instance {u}"


def test_execute_reports_errors():
    g = be.Context(1024, be.default_test_primes(1024)) if hasattr(be, "default_test_primes") else None
    if g is None:
        from eva_amd.hostref import coeff_modulus_create
        g = be.Context(1024, coeff_modulus_create(1024, [30, 30, 30]))
    rng = np.random.default_rng(1)
    a = g.upload_ct(rng.integers(0, 1 << 20, size=(2, 2, 1024), dtype=np.uint64), 2.0 ** 10)
    with pytest.raises(RuntimeError, match="not a ciphertext|out of range|before it is produced"):
        g.execute([(int(Op.Negate), 1, 5, 0, 0, 0)], {0: a}, n_vals=3)
    with pytest.raises(RuntimeError, match="written by exactly one op"):
        g.execute([(int(Op.Negate), 1, 0, 0, 0, 0), (int(Op.Negate), 1, 0, 0, 0, 0)], {0: a}, n_vals=3)
    with pytest.raises(RuntimeError, match="Unhandled op"):
        g.execute([(99, 1, 0, 0, 0, 0)], {0: a}, n_vals=3)
    with pytest.raises(RuntimeError, match="relinearization key not present|size-3"):
        g.execute([(int(Op.Relinearize), 1, 0, 0, 0, 0)], {0: a}, n_vals=3)
    assert a.h is not None and a.info()[0] == 2  # a failed submit leaves caller values alone


# ---- convolution windows in the scheduler: Rotate -> Mul(plain) -> Add chains whose rotations are deferred into
# evah_rotate_weighted_sums (eva_amd/csrc/scheduler.hip); the shapes below are the ones its bookkeeping must get right
def _window_programs():
    progs = {}

    p = EvaProgram('two_filters', vec_size=64)  # convolutionXY: two sums over the same rotations (one window, F = 2)
    with p:
        x = Input('x')
        taps = [x << 1, x << 2, x << 5, x]
        a = taps[0] * 0.5 + taps[1] * -1.5 + taps[2] * 0.25 + taps[3] * 2.0
        b = taps[0] * 1.5 + taps[1] * 0.75 + taps[2] * -0.5 + taps[3] * 0.125
        Output('a', a)
        Output('b', b)
    progs['two_filters'] = p

    p = EvaProgram('shared_rotation_different_sums', vec_size=64)  # r2 feeds two sums with different term lists
    with p:
        x = Input('x')
        r1, r2, r3 = x << 1, x << 2, x << 3
        Output('a', r1 * 0.5 + r2 * -1.5)
        Output('b', r2 * 0.75 + r3 * 0.25)
    progs['shared_rotation_different_sums'] = p

    p = EvaProgram('uneven_chains', vec_size=64)  # the second sum ends one level later than the first
    with p:
        x = Input('x')
        r1, r2 = x << 1, x << 4
        Output('a', r1 * 0.5 + r2 * -1.5)
        Output('b', r1 * 1.25 + r2 * 0.75 + x * 0.5 + (x >> 2) * 0.25)
    progs['uneven_chains'] = p

    p = EvaProgram('rotation_with_other_readers', vec_size=64)  # a tap that is also an output / squared: not deferred
    with p:
        x = Input('x')
        r1, r2 = x << 1, x << 2
        Output('a', r1 * 0.5 + r2 * -1.5)
        Output('r', r1)
        Output('s', r2 * r2)
    progs['rotation_with_other_readers'] = p

    p = EvaProgram('three_windows_two_levels', vec_size=64)  # Harris in miniature: windows of products of windows
    with p:
        x = Input('x')
        def conv(v, ws):
            acc = None
            for k, w in enumerate(ws):
                t = (v << k) * w
                acc = t if acc is None else acc + t
            return acc
        ix, iy = conv(x, [0.5, -1.0, 0.25]), conv(x, [0.125, 0.75, -0.5])
        Output('y', conv(ix * ix, [1.0, 1.0, 1.0]) + conv(ix * iy, [1.0, 0.5, 1.0]) - conv(iy * iy, [0.25, 1.0, 1.0]))
    progs['three_windows_two_levels'] = p
    for q in progs.values():
        q.set_output_ranges(20)
        q.set_input_scales(30)
    return progs


@pytest.mark.parametrize("name", ['two_filters', 'shared_rotation_different_sums', 'uneven_chains', 'rotation_with_other_readers',
                                  'three_windows_two_levels'])
@pytest.mark.parametrize("min_tiles", ["0", None], ids=["fused", "default"])
def test_execute_convolution_windows(name, min_tiles, monkeypatch):
    """min_tiles = 0 makes every window large enough for the hoisted + fused path; the default leaves these small
    programs on the general path of evah_rotate_weighted_sums.  Same ciphertexts as the oracle walk either way."""
    if min_tiles is not None:
        monkeypatch.setenv("EVAH_HOIST_MIN_TILES", min_tiles)
    rng = np.random.default_rng(11)
    _check(_window_programs()[name], {'x': list(rng.uniform(-1, 1, 64))}, N=4096)
