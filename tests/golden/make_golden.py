#!/usr/bin/env python
"""Generates the committed fixtures of tests/golden/.

  kat_constants.json  reference-derived known answers: the CoeffModulus::Create prime chains and
                      minimal 2N-th roots of SURVEY.md Appendix B, and the CKKSCompiler `prime_bits`
                      vectors the reference's own tests assert (/root/reference/tests/bug_fixes.py:68,
                      tests/features.py:129,133), with the program and configuration of each.
  ops_n1024.npz       inputs and expected outputs of every evaluator call on the path (N = 1024,
                      primes [60, 40, 60] -> 2 data limbs + the special prime), from seeded inputs,
                      of CKKSEncoder::encode at five scales, and of Decryptor::decrypt / CKKSEncoder::decode
                      (a ternary secret key, sizes 2 and 3, decode of encoded and of random plaintexts).
                      tests/golden/export_seal_vectors.py
                      turns the same vectors into raw files for tools/seal_parity.cpp, which diffs
                      them against a real SEAL >= 3.6 wherever one is installed.

The reference itself cannot run in this container (SEAL, protobuf and Galois are absent: SURVEY.md
§8(c)), so the ciphertext-level vectors are produced by this repo's CPU oracle (oracle/), whose
arithmetic is pinned to library-independent algebra by tests/test_oracle_kat.py.  They fix the
bits across rounds: a change of the oracle OR of the HIP kernels that alters any output word
fails tests/test_golden.py.

usage: python tests/golden/make_golden.py      (rewrites the two files next to this script)
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import pyoracle as po  # noqa: E402

N, BITS = 1024, [60, 40, 60]


def main():
    consts = {
        "coeff_modulus_create": [
            {"N": 8192, "bits": [60, 20, 60, 60], "primes": [0xFFFFFFFFFFD8001, 0xFC001, 0xFFFFFFFFFFE8001, 0xFFFFFFFFFFFC001]},
            {"N": 8192, "bits": [60, 30, 60, 60], "primes": [0xFFFFFFFFFFD8001, 0x3FFF4001, 0xFFFFFFFFFFE8001, 0xFFFFFFFFFFFC001]},
            {"N": 16384, "bits": [60, 20, 60, 60, 60, 60],
             "primes": [0xFFFFFFFFFE38001, 0xC0001, 0xFFFFFFFFFF28001, 0xFFFFFFFFFFC0001, 0xFFFFFFFFFFD8001, 0xFFFFFFFFFFE8001]},
            {"N": 8192, "bits": [60] * 4, "primes": [0xFFFFFFFFFFC4001, 0xFFFFFFFFFFD8001, 0xFFFFFFFFFFE8001, 0xFFFFFFFFFFFC001]},
            {"N": 65536, "bits": [60] * 11,
             "primes": [0xFFFFFFFFE740001, 0xFFFFFFFFE7C0001, 0xFFFFFFFFE9E0001, 0xFFFFFFFFECA0001, 0xFFFFFFFFEFE0001,
                        0xFFFFFFFFF240001, 0xFFFFFFFFF2A0001, 0xFFFFFFFFF5A0001, 0xFFFFFFFFF6A0001, 0xFFFFFFFFF840001,
                        0xFFFFFFFFFFC0001]},
        ],
        "minimal_primitive_root": [
            {"N": 8192, "q": 0xFFFFFFFFFFFC001, "psi": 25959043411404},
            {"N": 4096, "q": 1073692673, "psi": 236231},
        ],
        # program shape -> prime_bits the reference's tests expect from CKKSCompiler.compile
        "compiler_prime_bits": [
            {"test": "tests/bug_fixes.py:50-68 test_output_rescaled", "vec_size": 4,
             "program": "y = x*x", "input_scales": 60, "output_ranges": 20,
             "config": {"rescaler": "lazy_waterline", "warn_vec_size": "false"}, "prime_bits": [60, 20, 60, 60]},
            {"test": "tests/features.py:112-129 test_reduction_balancer (off)", "vec_size": 16384,
             "program": "y = (x1*(x2*(x3*x4))) + (x1+(x2+(x3+x4)))", "input_scales": 60, "output_ranges": 20,
             "config": {"rescaler": "always", "balance_reductions": "false", "warn_vec_size": "false"},
             "prime_bits": [60, 20, 60, 60, 60, 60]},
            {"test": "tests/features.py:131-133 test_reduction_balancer (on)", "vec_size": 16384,
             "program": "y = (x1*(x2*(x3*x4))) + (x1+(x2+(x3+x4)))", "input_scales": 60, "output_ranges": 20,
             "config": {"rescaler": "always", "balance_reductions": "true", "warn_vec_size": "false"},
             "prime_bits": [60, 20, 60, 60, 60]},
        ],
    }
    with open(os.path.join(HERE, "kat_constants.json"), "w") as f:
        json.dump(consts, f, indent=1)

    primes = po.coeff_modulus_create(N, BITS)
    o = po.Oracle(N, primes)
    k, l = len(primes), len(primes) - 1
    rng = np.random.default_rng(20260926)

    def rand(shape_prefix, nl):
        return np.stack([rng.integers(0, primes[i], size=shape_prefix + (N,), dtype=np.uint64) for i in range(nl)],
                        axis=len(shape_prefix))

    d = {"primes": np.array(primes, dtype=np.uint64)}
    d["a2"], d["b2"], d["a3"] = rand((2,), l), rand((2,), l), rand((3,), l)
    d["pt"] = rand((), l)
    d["relin_key"] = rand((l, 2), k)
    d["rot_steps"] = np.array([1, -3, 511], dtype=np.int64)
    for s in d["rot_steps"]:
        d[f"galois_key_{int(s)}"] = rand((l, 2), k)
    d["poly"] = rand((), 1)[0]
    a2, b2, a3, pt = d["a2"], d["b2"], d["a3"], d["pt"]
    d["out_ntt0"] = o.ntt(0, d["poly"])
    d["out_intt0"] = o.intt(0, d["poly"])
    d["out_add"] = o.add(a2, b2)
    d["out_add_32"] = o.add(a3, b2)
    d["out_sub"] = o.sub(a2, b2)
    d["out_sub_23"] = o.sub(a2, a3)
    d["out_negate"] = o.negate(a3)
    d["out_add_plain"] = o.add_plain(a2, pt)
    d["out_sub_plain"] = o.sub_plain(a2, pt)
    d["out_multiply"] = o.multiply(a2, b2)
    d["out_square"] = o.square(a2)
    d["out_multiply_plain"] = o.multiply_plain(a3, pt)
    d["out_relinearize"] = o.relinearize(a3, d["relin_key"])
    d["out_rescale"] = o.rescale(a2)
    d["out_rescale3"] = o.rescale(a3)
    d["out_relin_rescale"] = o.rescale(d["out_relinearize"])
    d["out_mod_switch"] = o.mod_switch(a3)
    for s in d["rot_steps"]:
        d[f"out_rotate_{int(s)}"] = o.rotate(a2, int(s), d[f"galois_key_{int(s)}"])
    d["out_triple"] = o.op_triple(a2, b2, d["relin_key"])
    # CKKSEncoder::encode (seal_executor.h:242): full slot vectors at the first data level; drawn
    # after everything else so the vectors above keep their values
    d["psi"] = np.array([o.psi(i) for i in range(k)], dtype=np.uint64)
    d["enc_scale_bits"] = np.array([20, 30, 40, 55, 60], dtype=np.int64)
    for c, sb in enumerate(d["enc_scale_bits"]):
        d[f"enc_values_{c}"] = rng.uniform(-3, 3, N // 2)
        d[f"out_encode_{c}"] = o.encode(l, d[f"enc_values_{c}"], 2.0 ** int(sb))
    # Decryptor::decrypt + CKKSEncoder::decode (/root/reference/eva/seal/seal.cpp:132-135); drawn last, so
    # every vector above keeps its value.  The secret key is ternary under every key prime, NTT form.
    small = rng.integers(-1, 2, size=N)
    d["sk_ntt"] = np.stack([o.ntt(i, np.array([int(v) % primes[i] for v in small], dtype=np.uint64)) for i in range(k)])
    d["out_decrypt2"] = o.decrypt(a2, d["sk_ntt"])
    d["out_decrypt3"] = o.decrypt(a3, d["sk_ntt"])
    for c, sb in enumerate(d["enc_scale_bits"]):  # decode of the encoder's own plaintexts: small coefficients of both signs
        d[f"out_decode_{c}"] = o.decode(d[f"out_encode_{c}"], 2.0 ** int(sb))
    # decode of uniformly random residues: every coefficient a random element of [0, Q) — multi-word, both halves
    d["out_decode_pt"] = o.decode(pt, 2.0 ** 10)
    d["out_decode_dec3"] = o.decode(d["out_decrypt3"], 2.0 ** 10)
    np.savez_compressed(os.path.join(HERE, "ops_n1024.npz"), **d)
    print("wrote", sorted(os.listdir(HERE)))


if __name__ == "__main__":
    main()
