"""BASELINE.json's DAG configurations at their stated sizes, output ciphertexts of
public_ctx.execute / execute_batch compared bit for bit with the CPU oracle walking the same
compiled DAG on the same encrypted inputs and keys (SURVEY.md section 8(d) describes the padding of
the prime chain to the stated number of data limbs L):
  C3  Harris corner detector (examples/image_processing.py:65-100), N = 2^15, L = 8
  C4  256 independent Sobel DAGs (examples/image_processing.py:39-63), N = 2^14, L = 5 — execute_batch
  C5  3x3 convolution + depth-8 squaring chain (tests/large_programs.py:10-53 style), N = 2^16, L = 12
"""
import numpy as np
import pytest

from eva import EvaProgram, Input, Output, evaluate
from eva.ckks import CKKSCompiler
from eva.metric import valuation_mse
from eva.seal import generate_keys
from evatest import compile_and_check, oracle_execute
from eva_amd.workloads import conv_depth8, harris as _harris, image as _image, pad_chain, sobel as _sobel

pytestmark = pytest.mark.gpu


def test_config5_conv_depth8_n65536_l12_bit_exact():
    _, params, _ = compile_and_check(conv_depth8(), _image(4096), check_bit_exact=True,
                                     params_hook=lambda p: pad_chain(p, 13, 65536))
    assert params.poly_modulus_degree == 65536 and len(params.prime_bits) == 13


def test_config3_harris_n32768_l8_bit_exact():
    _, params, _ = compile_and_check(_harris(), _image(4096), check_bit_exact=True,
                                     params_hook=lambda p: pad_chain(p, 9, 32768))
    assert params.poly_modulus_degree == 32768 and len(params.prime_bits) == 9 and len(params.rotations) == 9


def _c_walk_valuation(pub, compiled, enc, threads=64):
    from oracle_executor import c_walk
    return c_walk(pub, compiled, enc, threads=threads)[0]


def test_config3_harris_batch_n32768_l8_every_instance_bit_exact():
    """north_star's target workload as a batch: independent Harris DAGs at N = 2^15, L = 8 through execute_batch
    (what bench.py's dag_harris_batch leg times) — three groups over three issue queues, a last group that is not full,
    EVERY instance against the oracle's walk of the same DAG on the same ciphertexts and keys"""
    compiled, params, sig = CKKSCompiler(config={'warn_vec_size': 'false'}).compile(_harris())
    pad_chain(params, 9, 32768)
    pub, sec = generate_keys(params, 5)
    pub.batch_chunk = 4
    imgs = [_image(4096, shift=11 * u, scale=255.0 if u % 2 else 300.0) for u in range(11)]
    encs = [pub.encrypt(x, sig) for x in imgs]
    outs = pub.execute_batch(compiled, encs)
    assert len(outs) == 11
    for u in range(11):
        ref = _c_walk_valuation(pub, compiled, encs[u])
        for name, words in ref.items():
            assert np.array_equal(outs[u].get(name)[4], words), f"instance {u}, output {name}: execute_batch differs from the oracle walk"
    for u in (0, 5, 10):
        assert valuation_mse(sec.decrypt(outs[u], sig), evaluate(compiled, imgs[u])) < 0.01
    # the same instances one execute() at a time give the same words (the batched handle changes nothing)
    single = pub.execute(compiled, encs[6])
    for name in single.names():
        assert np.array_equal(single.get(name)[4], outs[6].get(name)[4])
    # r6: the call above ran on resident valuations (encrypt() leaves them in HBM, the outputs are views of the batched
    # output: nothing crossed PCIe).  Host valuations — inputs as host words, outputs downloaded — give the same words
    assert all(o.is_resident(n) for o in outs for n in o.names())
    st0 = pub.transfer_stats()
    again = pub.execute_batch(compiled, encs)
    st1 = pub.transfer_stats()
    assert st1["ct_uploads"] == st0["ct_uploads"] and st1["ct_downloads"] == st0["ct_downloads"], (st0, st1)
    for e in encs:
        e.to_host(True)
    pub.resident = False
    houts = pub.execute_batch(compiled, encs)
    for u in range(11):
        for name in houts[u].names():
            assert not houts[u].is_resident(name)
            assert np.array_equal(houts[u].get(name)[4], outs[u].get(name)[4])
            assert np.array_equal(again[u].get(name)[4], outs[u].get(name)[4])


def test_config3_harris_batch_default_groups_bit_exact():
    """the bench leg's own shape: groups of 12 instances (three full groups and a ragged one), every instance checked"""
    compiled, params, sig = CKKSCompiler(config={'warn_vec_size': 'false'}).compile(_harris())
    pad_chain(params, 9, 32768)
    pub, sec = generate_keys(params, 6)
    pub.batch_chunk = 12
    encs = [pub.encrypt(_image(4096, shift=u), sig) for u in range(4)]
    inputs = [encs[b % 4] for b in range(37)]
    outs = pub.execute_batch(compiled, inputs)
    refs = [_c_walk_valuation(pub, compiled, e) for e in encs]
    for u in range(37):
        for name, words in refs[u % 4].items():
            assert np.array_equal(outs[u].get(name)[4], words), f"instance {u}, output {name}"


def test_config4_256_sobel_n16384_every_instance_against_oracle():
    sob = _sobel(64, 64, 4096)
    sob.set_input_scales(25)
    sob.set_output_ranges(10)
    compiled, params, sig = CKKSCompiler(config={'warn_vec_size': 'false'}).compile(sob)
    pad_chain(params, 6, 16384)
    pub, sec = generate_keys(params, 11)
    # half-intensity images: at N = 2^14 with the padded chain the noise of a scale-2^25 encryption,
    # amplified by the cubic square-root approximation at the image's wrap-around pixel, alone exceeds
    # the reference's MSE threshold on the full-intensity image (the oracle walk shows the same 0.13)
    imgs = [{'image': [((37 * i + 13 * u) % 256) / 510.0 for i in range(4096)]} for u in range(256)]
    encs = [pub.encrypt(x, sig) for x in imgs]
    outs = pub.execute_batch(compiled, encs)
    assert len(outs) == 256
    for u in range(256):  # every instance (r6; the C walk of the oracle on the host's cores does one in ~30 ms)
        ref = _c_walk_valuation(pub, compiled, encs[u])
        for name, words in ref.items():
            assert np.array_equal(outs[u].get(name)[4], words), f"instance {u}, output {name}: execute_batch differs from the oracle walk"
    for u in (0, 1, 31, 32, 63, 64, 127, 200, 255):  # shapes, scales and the decrypted values of a sample
        ref = oracle_execute(pub, compiled, encs[u]) if u in (0, 255) else None
        for name in (ref.names() if ref else ()):
            g, o = outs[u].get(name), ref.get(name)
            assert g[:4] == o[:4]
            assert np.array_equal(g[4], o[4])
        assert valuation_mse(sec.decrypt(outs[u], sig), evaluate(compiled, imgs[u])) < 0.01
