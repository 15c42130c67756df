"""Independent sub-DAGs of one program on several devices (eva_amd/subdag.py):
the Harris corner detector's three convolutions run as three evah_execute submits on three contexts
(three devices on a multi-GPU node; three contexts of the one GPU here), with evah_ct_copy at the
cuts — the output ciphertext must equal the oracle's walk of the same DAG and the single-context
execute(), bit for bit."""
import numpy as np
import pytest

from eva import EvaProgram, Input, Output
from eva.ckks import CKKSCompiler
from eva.seal import generate_keys
from eva_amd import subdag
from oracle_executor import c_walk
from test_gpu_e2e import _harris, _image

pytestmark = pytest.mark.gpu


def test_plan_finds_the_three_convolutions_of_harris():
    compiled, params, sig = CKKSCompiler(config={'warn_vec_size': 'false'}).compile(_harris())
    params.poly_modulus_degree = 8192
    pub, sec = generate_keys(params, 1)
    enc = pub.encrypt(_image(4096), sig)
    ops, placed, outs, raw = subdag.lower(compiled, enc, pub._encode)
    pre, comps, suf = subdag.plan(ops, placed, 3)
    assert sorted(d for d, _ in comps) == [0, 1, 2] and len({len(c) for _, c in comps}) == 1
    assert sorted(pre + [i for _, c in comps for i in c] + suf) == list(range(len(ops)))  # a partition of the op list
    assert subdag.plan(ops, placed, 1) == (list(range(len(ops))), [], [])


@pytest.mark.parametrize("n_ctx", [2, 3])
def test_harris_split_over_contexts_bit_exact(n_ctx):
    compiled, params, sig = CKKSCompiler(config={'warn_vec_size': 'false'}).compile(_harris())
    params.poly_modulus_degree = 16384
    pub, sec = generate_keys(params, 4)
    enc = pub.encrypt(_image(4096), sig)
    ex = subdag.SubDagExecutor(pub, [0] * n_ctx)
    out = ex.execute(compiled, enc)
    assert len(ex.last_plan["components"]) >= 2
    ref, _ = c_walk(pub, compiled, enc, threads=4)
    single = pub.execute(compiled, enc)
    for name, (data, scale) in out.items():
        assert np.array_equal(data, ref[name]), f"{name}: split execution differs from the oracle walk"
        assert np.array_equal(data, single.get(name)[4]) and scale == single.get(name)[3]
    again = ex.execute(compiled, enc)   # a second run reuses the contexts
    assert all(np.array_equal(again[k][0], out[k][0]) for k in out)
    ex.close()


def test_program_without_parallel_branches_runs_on_one_context():
    prog = EvaProgram('chain', vec_size=64)
    with prog:
        x = Input('x')
        Output('y', (x * x + x) * x)
    prog.set_input_scales(30)
    prog.set_output_ranges(20)
    compiled, params, sig = CKKSCompiler(config={'warn_vec_size': 'false'}).compile(prog)
    pub, sec = generate_keys(params, 2)
    enc = pub.encrypt({'x': [i / 64.0 for i in range(64)]}, sig)
    ex = subdag.SubDagExecutor(pub, [0, 0])
    out = ex.execute(compiled, enc)
    assert ex.last_plan["components"] == []
    ref, _ = c_walk(pub, compiled, enc)
    assert np.array_equal(out['y'][0], ref['y'])
    ex.close()
