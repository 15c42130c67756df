"""CPU end-to-end: compiler + host crypto (keygen / encode / encrypt / decrypt) with the compiled
DAG walked over the CPU oracle — the reference's statistical oracle (MSE < 0.01,
/root/reference/tests/common.py:34) on the reference's own programs, without a GPU.  Also pins the
oracle's evaluator semantics end to end (rotation direction, rescale, relinearize decrypt right)."""
import numpy as np
import pytest

from eva import EvaProgram, Input, Output
from eva.std.numeric import horizontal_sum
from evatest import compile_and_check


def test_readme_polynomial():
    poly = EvaProgram('Polynomial', vec_size=1024)
    with poly:
        x = Input('x')
        Output('y', 3 * x ** 2 + 5 * x - 2)
    poly.set_output_ranges(30)
    poly.set_input_scales(30)
    compile_and_check(poly, {'x': [i / 64.0 for i in range(1024)]}, executor="oracle")


@pytest.mark.parametrize("rot", [-2, -1, 0, 1])
@pytest.mark.parametrize("left", [True, False])
def test_rotations_small(rot, left):
    prog = EvaProgram('RotOp', vec_size=8)
    with prog:
        x = Input('x')
        Output('y', (x << rot) if left else (x >> rot))
    prog.set_output_ranges(20)
    prog.set_input_scales(30)
    compile_and_check(prog, executor="oracle")


def test_mixed_raw_cipher_and_cube():
    for enc1 in (False, True):
        prog = EvaProgram('UnencryptedInputs', vec_size=128)
        with prog:
            x1 = Input('x1', enc1)
            x2 = Input('x2', True)
            Output('y', pow(x2, 3) + x1 * x2)
        prog.set_output_ranges(20)
        prog.set_input_scales(30)
        compile_and_check(prog, executor="oracle")


def test_horizontal_sum_small():
    prog = EvaProgram('HorizontalSum', vec_size=64)
    with prog:
        x = Input('x')
        Output('y', horizontal_sum(x))
    prog.set_output_ranges(25)
    prog.set_input_scales(25)
    compile_and_check(prog, executor="oracle")


def test_transparent_ciphertext():
    prog = EvaProgram('Transparent', vec_size=512)
    with prog:
        x = Input('x')
        Output('y', x - x + x * 0)
    prog.set_output_ranges(20)
    prog.set_input_scales(30)
    compile_and_check(prog, executor="oracle")


def test_constant_at_large_scale_encodes():
    """A constant matched up to a 160-bit scale exceeds 128-bit coefficients: the encoder's
    multi-precision path (mantissa * 2^e per residue) must handle it."""
    prog = EvaProgram('poly', vec_size=256)
    with prog:
        x = Input('x')
        y = Input('y')
        Output('z', (x * y + x) * (x * x) + 0.5)
    prog.set_output_ranges(20)
    prog.set_input_scales(40)
    compile_and_check(prog, config={'lazy_relinearize': 'false'}, executor="oracle")


@pytest.mark.parametrize("seed", range(6))
def test_random_programs_compile_and_decrypt_on_the_oracle(seed):
    """The generator of tests/test_gpu_fuzz.py on the CPU: compiled semantics == source semantics,
    and the compiled DAG walked over the oracle decrypts to the reference (tests/common.py:34)."""
    from test_gpu_fuzz import _random_program
    prog, inputs = _random_program(100 + seed, 32)
    compile_and_check(prog, inputs, executor="oracle", seed=seed + 1)


def test_threaded_oracle_walk_equals_serial_walk():
    """The node-parallel CPU walk (the reported multi-core baseline of the DAG configs) produces the
    serial walk's ciphertexts."""
    import numpy as np
    from eva.ckks import CKKSCompiler
    from eva.seal import generate_keys
    from evatest import oracle_execute
    from eva_amd.workloads import sobel as _sobel
    sob = _sobel(16, 16, 256)
    sob.set_input_scales(25)
    sob.set_output_ranges(10)
    compiled, params, sig = CKKSCompiler(config={'warn_vec_size': 'false'}).compile(sob)
    pub, sec = generate_keys(params, 5)
    enc = pub.encrypt({'image': [((37 * i) % 256) / 255.0 for i in range(256)]}, sig)
    a, b = oracle_execute(pub, compiled, enc), oracle_execute(pub, compiled, enc, threads=4)
    for name in a.names():
        x, y = a.get(name), b.get(name)
        assert x[:4] == y[:4] and np.array_equal(x[4], y[4])


def _constant_chain_program():
    prog = EvaProgram('consts', vec_size=16)
    with prog:
        x = Input('x')
        Output('y', x * 0.5 * 0.25 * 2.0 * 1.5 + x)
    prog.set_input_scales(30)
    prog.set_output_ranges(20)
    return prog


def test_product_of_several_constants_with_a_ciphertext():
    """x*c1*c2*c3*c4: the reduction balancer pairs the constants (raw x raw products created after
    type deduction), the rescaler then inserts a Rescale on an unencrypted value — found by the
    randomised test (seed 418).  The reference's SEALExecutor cannot run that node
    (seal_executor.h:209-215 takes a Ciphertext); here it is the copy its semantic executor makes."""
    prog = _constant_chain_program()
    compiled, params, sig = compile_and_check(prog, {'x': [i / 16.0 for i in range(16)]}, executor="oracle")
    from eva import Op
    dump = {d["id"]: d for d in compiled._dump()}

    def unencrypted(t):
        d = dump[t]
        return d["op"] == Op.Constant or (d["op"] == Op.Mul and all(unencrypted(o) for o in d["operands"]))
    assert any(d["op"] == Op.Rescale and unencrypted(d["operands"][0]) for d in dump.values()), \
        "this program is meant to put a Rescale on a constant product"


@pytest.mark.parametrize("threads", [1, 4])
def test_c_dag_walk_equals_python_walk(threads):
    """oracle/eva_oracle_dag.c (the CPU baseline of the DAG configs: serial forwardPass and the
    dependency-counting multicore traversal) gives the ciphertexts of the node-by-node Python walk."""
    import numpy as np
    from eva.ckks import CKKSCompiler
    from eva.seal import generate_keys
    from oracle_executor import OracleExecutor, c_walk
    from eva_amd.workloads import sobel as _sobel
    for prog, inputs in ((_sobel(16, 16, 256), {'image': [((37 * i) % 256) / 255.0 for i in range(256)]}),
                         (_constant_chain_program(), {'x': [i / 16.0 for i in range(16)]})):
        if prog.name != 'chain':
            prog.set_input_scales(25)
            prog.set_output_ranges(10)
        compiled, params, sig = CKKSCompiler(config={'warn_vec_size': 'false'}).compile(prog)
        pub, sec = generate_keys(params, 5)
        enc = pub.encrypt(inputs, sig)
        ref = OracleExecutor(pub).execute(compiled, enc)
        got, dt = c_walk(pub, compiled, enc, threads=threads)
        assert dt > 0
        for name, v in ref.items():
            if isinstance(v, list):
                continue
            assert np.array_equal(got[name], v.data), name
