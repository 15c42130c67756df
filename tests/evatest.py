"""Shared helpers for the end-to-end tests: the reference's one assertion helper
(/root/reference/tests/common.py:12-36) re-expressed over eva_amd, with a seeded RNG and a
choice of executor: the GPU (public_ctx.execute) or the CPU oracle walked by
tests/oracle_executor.py (CPU-only suites; never the product path)."""
import random

import numpy as np

from eva import evaluate
from eva.ckks import CKKSCompiler
from eva.metric import valuation_mse
from eva.seal import generate_keys, SEALValuation


def oracle_execute(public_ctx, compiled, enc_inputs, threads=1):
    from oracle_executor import OracleExecutor, Cipher, Plain
    outs = OracleExecutor(public_ctx).execute(compiled, enc_inputs, threads=threads)
    val = SEALValuation()
    for name, v in outs.items():
        if isinstance(v, Cipher):
            val._set_cipher(name, v.data, v.scale)
        elif isinstance(v, Plain):
            val._set_plain(name, v.data, v.scale)
        else:
            val._set_raw(name, v)
    return val


def compile_and_check(prog, inputs=None, config=None, executor="gpu", seed=1, params_hook=None,
                      check_bit_exact=False):
    """reference -> compile -> reference(compiled) MSE < 1e-10 -> keygen/encrypt/execute/decrypt
    MSE < 0.01.  Returns (compiled, params, signature)."""
    config = dict(config or {})
    rng = random.Random(seed)
    if inputs is None:
        inputs = {name: [rng.uniform(-2, 2) for _ in range(prog.vec_size)] for name in prog.inputs}
    config['warn_vec_size'] = 'false'
    reference = evaluate(prog, inputs)
    compiled, params, signature = CKKSCompiler(config=config).compile(prog)
    ref_mse = valuation_mse(reference, evaluate(compiled, inputs))
    assert ref_mse < 1e-10, f"compiled program changed semantics: MSE {ref_mse}"
    if executor is None:
        return compiled, params, signature
    if params_hook:
        params_hook(params)
    public_ctx, secret_ctx = generate_keys(params, seed)
    enc_inputs = public_ctx.encrypt(inputs, signature)
    if executor == "gpu":
        enc_outputs = public_ctx.execute(compiled, enc_inputs)
        if check_bit_exact:
            ref_out = oracle_execute(public_ctx, compiled, enc_inputs)
            for name in enc_outputs.names():
                g, o = enc_outputs.get(name), ref_out.get(name)
                assert g[0] == o[0] and g[1:4] == o[1:4], (name, g[:4], o[:4])
                if g[0] != "raw":
                    assert np.array_equal(g[4], o[4]), f"output {name}: GPU ciphertext differs from the CPU oracle"
    else:
        enc_outputs = oracle_execute(public_ctx, compiled, enc_inputs)
    outputs = secret_ctx.decrypt(enc_outputs, signature)
    he_mse = valuation_mse(outputs, reference)
    assert he_mse < 0.01, f"Mean squared error was {he_mse}"
    return compiled, params, signature
