"""GPU end-to-end: the reference's own test programs (tests/features.py, large_programs.py,
bug_fixes.py, std.py; examples/image_processing.py) through compile -> generate_keys -> encrypt ->
public_ctx.execute (MI355X) -> decrypt, with the reference's thresholds (tests/common.py:25,34)
and, where marked, the output ciphertexts compared bit-for-bit with the CPU oracle walking the
same compiled DAG on the same encrypted inputs and keys."""
import math

import pytest

from eva import EvaProgram, Input, Output
from eva.std.numeric import horizontal_sum
from evatest import compile_and_check
from eva_amd.workloads import sobel as _sobel

pytestmark = pytest.mark.gpu


def test_readme_polynomial_bit_exact():
    """BASELINE config 1 (README.md:120-150)"""
    poly = EvaProgram('Polynomial', vec_size=1024)
    with poly:
        x = Input('x')
        Output('y', 3 * x ** 2 + 5 * x - 2)
    poly.set_output_ranges(30)
    poly.set_input_scales(30)
    compile_and_check(poly, {'x': [i / 64.0 for i in range(1024)]}, check_bit_exact=True)


def test_bin_ops():
    for binop in (lambda a, b: a + b, lambda a, b: a - b, lambda a, b: a * b):
        for enc1 in (False, True):
            for enc2 in (False, True):
                prog = EvaProgram('BinOp', vec_size=64)
                with prog:
                    a = Input('a', enc1)
                    b = Input('b', enc2)
                    Output('y', binop(a, b))
                prog.set_output_ranges(20)
                prog.set_input_scales(30)
                compile_and_check(prog, check_bit_exact=True)


def test_unary_ops():
    for unop in (lambda x: x, lambda x: -x, lambda x: x ** 3, lambda x: 42):
        for enc in (False, True):
            prog = EvaProgram('UnOp', vec_size=64)
            with prog:
                x = Input('x', enc)
                Output('y', unop(x))
            prog.set_output_ranges(20)
            prog.set_input_scales(30)
            compile_and_check(prog, check_bit_exact=True)


def test_rotations():
    for rotop in (lambda x, r: x << r, lambda x, r: x >> r):
        for rot in range(-2, 2):
            prog = EvaProgram('RotOp', vec_size=8)
            with prog:
                x = Input('x')
                Output('y', rotop(x, rot))
            prog.set_output_ranges(20)
            prog.set_input_scales(30)
            compile_and_check(prog, check_bit_exact=True)


def test_unencrypted_computation():
    for enc1 in (False, True):
        for enc2 in (False, True):
            prog = EvaProgram('UnencryptedInputs', vec_size=128)
            with prog:
                x1 = Input('x1', enc1)
                x2 = Input('x2', enc2)
                Output('y', pow(x2, 3) + x1 * x2)
            prog.set_output_ranges(20)
            prog.set_input_scales(30)
            compile_and_check(prog)


def test_eager_relinearize_uses_fused_relin_rescale_bit_exact():
    """lazy_relinearize=false puts Relinearize directly under Rescale: the executor evaluates the
    pair with one fused call; the oracle walk evaluates them separately — outputs must be identical."""
    from eva import Op
    prog = EvaProgram('poly', vec_size=1024)
    with prog:
        x = Input('x')
        y = Input('y')
        Output('z', (x * y + x) * (x * x) + 0.5)
    prog.set_output_ranges(20)
    prog.set_input_scales(60)
    compiled, _, _ = compile_and_check(prog, config={'lazy_relinearize': 'false'}, check_bit_exact=True)
    terms = {d["id"]: d for d in compiled._dump()}
    assert any(d["op"] == Op.Rescale and terms[d["operands"][0]]["op"] == Op.Relinearize for d in terms.values())


def test_transparent_ciphertext():
    prog = EvaProgram('Transparent', vec_size=4096)
    with prog:
        x = Input('x')
        Output('y', x - x + x * 0)
    prog.set_output_ranges(20)
    prog.set_input_scales(30)
    compile_and_check(prog, check_bit_exact=True)


def test_x1x1x2_scale60():
    prog = EvaProgram('prog', vec_size=128)
    with prog:
        x1, x2 = Input('x1'), Input('x2')
        Output('y', x1 * x1 * x2)
    prog.set_output_ranges(20)
    prog.set_input_scales(60)
    compile_and_check(prog, check_bit_exact=True)


@pytest.mark.parametrize("enc", [True, False])
def test_horizontal_sum(enc):
    prog = EvaProgram('HorizontalSum', vec_size=2048)
    with prog:
        x = Input('x', is_encrypted=enc)
        Output('y', horizontal_sum(x))
    prog.set_output_ranges(25)
    prog.set_input_scales(33)
    compile_and_check(prog, check_bit_exact=enc)


@pytest.mark.parametrize("rescaler", ['lazy_waterline', 'eager_waterline', 'always'])
@pytest.mark.parametrize("balance", ['true', 'false'])
def test_sobel_configs(rescaler, balance):
    """tests/large_programs.py:10-53 — Sobel on a 90x90 image padded to vec 8192"""
    sobel = _sobel(90, 90, 2 ** math.ceil(math.log(90 * 90, 2)))
    sobel.set_input_scales(45)
    sobel.set_output_ranges(20)
    compile_and_check(sobel, config={'rescaler': rescaler, 'balance_reductions': balance},
                      check_bit_exact=(rescaler == 'lazy_waterline' and balance == 'true'))


from eva_amd.workloads import harris as _harris, image as _image  # noqa: E402  (shared with bench.py and scripts/)


def test_sobel_example_n8192_bit_exact():
    """BASELINE config 2: examples/image_processing.py Sobel 64x64, N = 2^13"""
    sobel = _sobel(64, 64, 4096)
    sobel.set_input_scales(25)
    sobel.set_output_ranges(10)
    def force(params):  # N forced to 2^13 (legal: seal.cpp:169 uses sec_level none)
        params.poly_modulus_degree = 8192
    compile_and_check(sobel, _image(4096), check_bit_exact=True, params_hook=force)


def test_harris_example():
    """BASELINE config 3: examples/image_processing.py Harris 64x64, N forced to 2^15"""
    def force(params):
        params.poly_modulus_degree = 32768
    compiled, params, _ = compile_and_check(_harris(), _image(4096), check_bit_exact=True, params_hook=force)
    assert list(params.prime_bits) == [60] * 5 and len(params.rotations) == 9


def test_regression_programs():
    """tests/large_programs.py:55-146 — deterministic inputs"""
    linreg = EvaProgram('linear_regression', vec_size=2048)
    with linreg:
        p = 63
        x = [Input(f'x{i}') for i in range(p)]
        e = Input('e')
        y = e + 6.56
        for i in range(p):
            y += x[i] * (i * 0.732)
        Output('y', y)
    linreg.set_input_scales(40)
    linreg.set_output_ranges(30)
    inputs = {'e': [(2048 - i) * 0.001 for i in range(2048)]}
    for i in range(63):
        inputs[f'x{i}'] = [i * j * 0.01 for j in range(2048)]
    compile_and_check(linreg, inputs)

    polyreg = EvaProgram('polynomial_regression', vec_size=4096)
    with polyreg:
        x, e = Input('x'), Input('e')
        y = e + 6.56
        for i in range(4):
            x_i = x
            for j in range(i):
                x_i = x_i * x
            y += x_i * (i * 0.732)
        Output('y', y)
    polyreg.set_input_scales(40)
    polyreg.set_output_ranges(30)
    compile_and_check(polyreg, {'x': [i * 0.01 for i in range(4096)],
                                'e': [(4096 - i) * 0.001 for i in range(4096)]})


def test_graph_replay_matches_eager_and_oracle():
    """Repeated execute() of one compiled program replays a captured hipGraph (multi-queue);
    results must be bit-identical to the eager walk and to the CPU oracle, for fresh inputs too."""
    import numpy as np
    from eva.ckks import CKKSCompiler
    from eva.seal import generate_keys
    from evatest import oracle_execute
    sobel = _sobel(64, 64, 4096)
    sobel.set_input_scales(25)
    sobel.set_output_ranges(10)
    compiled, params, sig = CKKSCompiler(config={'warn_vec_size': 'false'}).compile(sobel)
    params.poly_modulus_degree = 8192
    pub, sec = generate_keys(params, 3)
    for harris_too in (False, True):
        if harris_too:
            compiled, params, sig = CKKSCompiler(config={'warn_vec_size': 'false'}).compile(_harris())
            params.poly_modulus_degree = 16384
            pub, sec = generate_keys(params, 4)
        enc_a = pub.encrypt(_image(4096), sig)
        enc_b = pub.encrypt({'image': [((91 * i) % 256) / 255.0 for i in range(4096)]}, sig)
        pub.use_graphs = False
        eager_a = pub.execute(compiled, enc_a).get('image')
        pub.use_graphs = True
        outs = [pub.execute(compiled, e).get('image') for e in (enc_a, enc_a, enc_a, enc_b, enc_a)]
        ref_a = oracle_execute(pub, compiled, enc_a).get('image')
        ref_b = oracle_execute(pub, compiled, enc_b).get('image')
        assert np.array_equal(eager_a[4], ref_a[4])
        for i, o in enumerate(outs):
            ref = ref_b if i == 3 else ref_a
            assert o[:4] == ref[:4]
            assert np.array_equal(o[4], ref[4]), f"execute() call {i} (graph replay from call 1 on) differs from the oracle"
        assert pub.last_timing[1] < 5.0


def test_executor_options_give_identical_ciphertexts():
    """Issue-queue count, the library scheduler vs the node-by-node host walk, rotation batching and
    relinearize+rescale fusion are scheduling choices: every combination must produce the same
    output ciphertext."""
    import numpy as np
    from eva.ckks import CKKSCompiler
    from eva.seal import generate_keys
    compiled, params, sig = CKKSCompiler(config={'warn_vec_size': 'false', 'lazy_relinearize': 'false'}).compile(_harris())
    params.poly_modulus_degree = 16384
    pub, sec = generate_keys(params, 9)
    enc = pub.encrypt(_image(4096), sig)
    pub.use_graphs = False
    base = None
    import os
    for queues in (1, 3, 8):
        for batch in ("1", "0"):
            for fuse in ("1", "0"):
                for lib in ("1", "0"):
                    os.environ["EVA_BATCH_ROTATIONS"], os.environ["EVA_FUSE_RELIN_RESCALE"] = batch, fuse
                    pub.library_scheduler = lib == "1"
                    pub.num_queues = queues
                    out = pub.execute(compiled, enc).get('image')
                    if base is None:
                        base = out
                    assert out[:4] == base[:4] and np.array_equal(out[4], base[4]), (queues, batch, fuse, lib)
    os.environ.pop("EVA_BATCH_ROTATIONS"); os.environ.pop("EVA_FUSE_RELIN_RESCALE")


def test_execute_batch_equals_execute_per_instance():
    """BASELINE config 4's unit: a batch of independent input valuations of one program through
    execute_batch (batched device handles, one launch set per node per group) gives, instance by
    instance, the ciphertexts of execute() — which the other tests pin to the oracle."""
    import numpy as np
    from eva.ckks import CKKSCompiler
    from eva.seal import generate_keys
    from eva.metric import valuation_mse
    from eva import evaluate
    sob = _sobel(32, 32, 1024)
    sob.set_input_scales(25)
    sob.set_output_ranges(10)
    compiled, params, sig = CKKSCompiler(config={'warn_vec_size': 'false'}).compile(sob)
    pub, sec = generate_keys(params, 7)
    imgs = [{'image': [((37 * i + 11 * u) % 256) / 255.0 for i in range(1024)]} for u in range(5)]
    encs = [pub.encrypt(x, sig) for x in imgs]
    pub.batch_chunk = 3   # 5 instances -> groups of 3 and 2
    outs = pub.execute_batch(compiled, encs)
    assert len(outs) == 5
    pub.use_graphs = False
    for x, e, o in zip(imgs, encs, outs):
        one = pub.execute(compiled, e)
        for name in one.names():
            g, r = o.get(name), one.get(name)
            assert g[:4] == r[:4]
            assert np.array_equal(g[4], r[4]), f"{name}: batched instance differs from the single execute()"
        assert valuation_mse(sec.decrypt(o, sig), evaluate(compiled, x)) < 0.01
    with pytest.raises(RuntimeError, match="batch_chunk"):
        pub.batch_chunk = 65
        pub.execute_batch(compiled, encs)


def test_execute_batch_called_repeatedly_with_fresh_inputs_and_a_partial_last_group():
    """Repeated execute_batch calls of one program (resident constants, recycled pools, both issue
    queues, a partial last group): every instance equals execute() on it."""
    import numpy as np
    from eva.ckks import CKKSCompiler
    from eva.seal import generate_keys
    sob = _sobel(32, 32, 1024)
    sob.set_input_scales(25)
    sob.set_output_ranges(10)
    compiled, params, sig = CKKSCompiler(config={'warn_vec_size': 'false'}).compile(sob)
    pub, sec = generate_keys(params, 9)
    pub.batch_chunk = 4

    def batch(seed, n):
        return [pub.encrypt({'image': [((31 * i + 7 * u + seed) % 256) / 255.0 for i in range(1024)]}, sig) for u in range(n)]

    pub.execute_batch(compiled, batch(0, 9))
    encs = batch(100, 19)                         # 4 full groups + 3 instances
    for round_ in range(2):
        outs = pub.execute_batch(compiled, encs)
        assert len(outs) == 19
        pub.use_graphs = False
        for u in (0, 3, 4, 7, 8, 12, 15, 16, 18):
            one = pub.execute(compiled, encs[u])
            for name in one.names():
                g, r = outs[u].get(name), one.get(name)
                assert g[:4] == r[:4]
                assert np.array_equal(g[4], r[4]), f"round {round_}, instance {u}, {name}: batched instance differs from execute()"
        pub.use_graphs = True
        encs = batch(200 + round_, 19)


def test_product_of_several_constants_with_a_ciphertext():
    """see tests/test_host_e2e_cpu.py: a Rescale lands on an unencrypted constant product"""
    from test_host_e2e_cpu import _constant_chain_program
    compile_and_check(_constant_chain_program(), {'x': [i / 16.0 for i in range(16)]}, check_bit_exact=True)
