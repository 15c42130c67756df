"""CPU: compiler known answers and semantic preservation on the reference's own programs
(/root/reference/tests/features.py, bug_fixes.py, large_programs.py, std.py)."""
import math

import pytest

from eva import EvaProgram, Input, Output, Op, Type, evaluate
from eva.ckks import CKKSCompiler
from eva.std.numeric import horizontal_sum
from evatest import compile_and_check


def test_kat_prime_bits_square_scale60():
    """tests/bug_fixes.py:51-68: x*x, scale 60, range 20, lazy_waterline -> [60,20,60,60]"""
    prog = EvaProgram('x2', vec_size=4096)
    with prog:
        x = Input('x')
        Output('y', x * x)
    prog.set_output_ranges(20)
    prog.set_input_scales(60)
    _, params, _ = compile_and_check(prog, executor=None)
    assert list(params.prime_bits) == [60, 20, 60, 60]


def test_kat_prime_bits_reduction_balancer():
    """tests/features.py:113-133"""
    prog = EvaProgram('ReductionTree', vec_size=16384)
    with prog:
        x1, x2, x3, x4 = (Input(f'x{i}') for i in range(1, 5))
        Output('y', (x1 * (x2 * (x3 * x4))) + (x1 + (x2 + (x3 + x4))))
    prog.set_output_ranges(20)
    prog.set_input_scales(60)
    _, params, _ = compile_and_check(prog, executor=None, config={'rescaler': 'always', 'balance_reductions': 'false'})
    assert list(params.prime_bits) == [60, 20, 60, 60, 60, 60]
    _, params, _ = compile_and_check(prog, executor=None, config={'rescaler': 'always', 'balance_reductions': 'true'})
    assert list(params.prime_bits) == [60, 20, 60, 60, 60]


def test_x1x1x2_scale60():
    """tests/bug_fixes.py:10-26"""
    prog = EvaProgram('prog', vec_size=128)
    with prog:
        x1, x2 = Input('x1'), Input('x2')
        Output('y', x1 * x1 * x2)
    prog.set_output_ranges(20)
    prog.set_input_scales(60)
    compile_and_check(prog, executor=None)


from eva_amd.workloads import sobel as _sobel  # noqa: E402  (shared with bench.py and scripts/)


@pytest.mark.parametrize("rescaler", ['lazy_waterline', 'eager_waterline', 'always'])
@pytest.mark.parametrize("balance", ['true', 'false'])
def test_sobel_configs_compile(rescaler, balance):
    """tests/large_programs.py:10-53 (compile + plain-semantics check on CPU)"""
    sobel = _sobel(90, 90, 2 ** math.ceil(math.log(90 * 90, 2)))
    sobel.set_input_scales(45)
    sobel.set_output_ranges(20)
    compiled, params, sig = compile_and_check(sobel, executor=None, config={'rescaler': rescaler, 'balance_reductions': balance})
    ops = [d["op"] for d in compiled._dump()]
    assert Op.Relinearize in ops and Op.Rescale in ops and Op.Encode in ops
    assert sorted(params.rotations) == sorted({i * 90 + j for i in range(3) for j in range(3)})


def test_compiled_program_invariants():
    """Appendix C of SURVEY.md: facts the executor relies on."""
    sobel = _sobel(64, 64, 4096)
    sobel.set_input_scales(25)
    sobel.set_output_ranges(10)
    compiled, params, sig = compile_and_check(sobel, executor=None)
    terms = {d["id"]: d for d in compiled._dump()}
    # types by propagation
    ty = {}
    for d in compiled._dump():
        if d["op"] == Op.Input:
            ty[d["id"]] = d["type"]
        elif d["op"] == Op.Constant:
            ty[d["id"]] = Type.Raw
        elif d["op"] == Op.Encode:
            ty[d["id"]] = Type.Plain
        else:
            ty[d["id"]] = Type.Cipher if any(ty[o] == Type.Cipher for o in d["operands"]) else Type.Raw
    for d in compiled._dump():
        if d["op"] == Op.Sub:  # plain - cipher never reaches the executor
            assert not (ty[d["operands"][0]] != Type.Cipher and ty[d["operands"][1]] == Type.Cipher)
        if d["op"] == Op.Mul and all(ty[o] == Type.Cipher for o in d["operands"]):
            for o in d["operands"]:  # cipher x cipher only on size-2 operands: no pending Mul above
                assert terms[o]["op"] != Op.Mul or not all(ty[q] == Type.Cipher for q in terms[o]["operands"])
        if d["op"] == Op.Rescale:
            assert d["rescale_divisor"] == 60
        if d["op"] in (Op.Input, Op.Encode):
            assert "encode_level" in d and "encode_scale" in d
    assert params.poly_modulus_degree >= 2 * 4096 and sig.vec_size == 4096


def test_horizontal_sum_variants():
    """tests/std.py:10-36"""
    for enc in (True, False):
        prog = EvaProgram('HorizontalSum', vec_size=2048)
        with prog:
            x = Input('x', is_encrypted=enc)
            Output('y', horizontal_sum(x))
        prog.set_output_ranges(25)
        prog.set_input_scales(33)
        compile_and_check(prog, executor=None)
    prog = EvaProgram('HorizontalSumConstant', vec_size=2048)
    with prog:
        Output('y', horizontal_sum([1 for _ in range(prog.vec_size)]))
    prog.set_output_ranges(25)
    prog.set_input_scales(33)
    compile_and_check(prog, executor=None)


def test_security_levels_and_errors():
    """tests/features.py:79-111"""
    degrees = {}
    for s in ('128', '192', '256'):
        for q in ('false', 'true'):
            prog = EvaProgram('SecurityLevel', vec_size=512)
            with prog:
                x = Input('x')
                Output('y', 5 * x * x + 3 * x + x << 12 + 10)
            prog.set_output_ranges(20)
            prog.set_input_scales(30)
            _, params, _ = compile_and_check(prog, executor=None, config={'security_level': s, 'quantum_safe': q})
            degrees[(s, q)] = params.poly_modulus_degree
    assert degrees[('128', 'false')] <= degrees[('192', 'false')] <= degrees[('256', 'false')]
    prog = EvaProgram('SecurityLevel', vec_size=512)
    with prog:
        x = Input('x')
        Output('y', 5 * x * x + 3 * x)
    prog.set_output_ranges(20)
    prog.set_input_scales(30)
    with pytest.raises(RuntimeError, match="up to 256 bit security"):
        CKKSCompiler(config={'security_level': '1024'}).compile(prog)


def test_missing_scale_is_an_error():
    prog = EvaProgram('NoScale', vec_size=8)
    with prog:
        Output('y', Input('x') * 2)
    with pytest.raises(RuntimeError, match="scale for"):
        CKKSCompiler().compile(prog)


def test_program_api_errors():
    with pytest.raises(RuntimeError, match="power-of-two"):
        EvaProgram('bad', vec_size=3)
    with pytest.raises(RuntimeError, match="No Program in context"):
        Input('x')
    p = EvaProgram('p', vec_size=4)
    with p:
        x = Input('x')
        with pytest.raises(ValueError):
            x ** 0
        Output('y', x)
    with pytest.raises(RuntimeError, match="length of all inputs"):
        evaluate(p, {'x': [1, 2, 3]})


def test_repeated_squaring_is_not_flattened():
    """reduction_balancer.h:44-45: a term used twice by the same product (acc*acc) has two use
    edges and is not merged into its user — a depth-8 squaring chain stays 8 multiplications."""
    from eva import EvaProgram, Input, Output, Op
    from eva.ckks import CKKSCompiler
    prog = EvaProgram('chain', vec_size=64)
    with prog:
        acc = Input('x')
        for _ in range(8):
            acc = acc * acc
        Output('y', acc)
    prog.set_input_scales(30)
    prog.set_output_ranges(20)
    compiled, params, sig = CKKSCompiler(config={'warn_vec_size': 'false'}).compile(prog)
    ops = [d["op"] for d in compiled._dump()]
    assert ops.count(Op.Relinearize) == 8
    assert sum(1 for d in compiled._dump() if d["op"] == Op.Mul and d["operands"][0] == d["operands"][1]) == 8
    assert len(params.prime_bits) == 10  # 8 rescale primes + one output prime + special


def test_every_plain_minus_cipher_is_lowered():
    """seal_lowering.h:24-30 turns plaintext - ciphertext into (-ciphertext) + plaintext; the
    reference's traversal never reaches terms downstream of the Negate/Add it creates, so a second
    such Sub further down survives there.  Here every one is lowered."""
    from eva import EvaProgram, Input, Output, Op
    from eva.ckks import CKKSCompiler
    prog = EvaProgram('subs', vec_size=8)
    with prog:
        x = Input('x')
        p = Input('p', False)
        y = (p - x) + 0.5
        Output('y', p - (y + y))
    prog.set_input_scales(30)
    prog.set_output_ranges(20)
    compiled, params, sig = CKKSCompiler(config={'warn_vec_size': 'false'}).compile(prog)
    dump = {d["id"]: d for d in compiled._dump()}
    assert not any(d["op"] == Op.Sub for d in dump.values())
    assert sum(1 for d in dump.values() if d["op"] == Op.Negate) == 2
