#!/usr/bin/env python
"""Writes the golden vectors as raw little-endian files for tools/seal_parity.cpp (which diffs
them against a real Microsoft SEAL >= 3.6 — the reference's arithmetic, absent from the
development container).

  python tests/golden/export_seal_vectors.py OUT_DIR            # the committed N = 1024 set
  python tests/golden/export_seal_vectors.py OUT_DIR N b0,b1,.. # a fresh set from the CPU oracle
                                                                # (e.g. 65536 60,60,...,60; 8192 60,20,60,60 puts
                                                                # one of EVA's 20-bit output primes in the chain)
OUT_DIR/manifest.txt  : `key v0 v1 ...` lines (N, bits, scale_log2, rot_steps, enc_scale_bits)
OUT_DIR/<name>.u64    : uint64 arrays, C order ([size][limbs][N] for ciphertexts,
                        [digit][2][k][N] for keys); OUT_DIR/<name>.f64 : float64 arrays
OUT_DIR/seal_*.bin    : the same parameters, values and keys as SEAL objects, written by THIS REPO's writer of
                        SEAL's binary format (eva_amd/host/seal_format.h) — seal_parity section 7 loads them with
                        SEAL and compares SEAL's own save() with them byte for byte
Test infrastructure only (it may import oracle/)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))


def fresh(N, bits, seed=20260927):
    sys.path.insert(0, ROOT)
    from oracle import pyoracle as po
    primes = po.coeff_modulus_create(N, bits)
    o = po.Oracle(N, primes)
    k, l = len(primes), len(primes) - 1
    rng = np.random.default_rng(seed)

    def rand(prefix, nl):
        return np.stack([rng.integers(0, primes[i], size=prefix + (N,), dtype=np.uint64) for i in range(nl)], axis=len(prefix))
    d = {"primes": np.array(primes, dtype=np.uint64), "psi": np.array([o.psi(i) for i in range(k)], dtype=np.uint64)}
    d["a2"], d["b2"], d["a3"], d["pt"] = rand((2,), l), rand((2,), l), rand((3,), l), rand((), l)
    d["relin_key"] = rand((l, 2), k)
    d["rot_steps"] = np.array([1, -3, N // 2 - 1], dtype=np.int64)
    for s in d["rot_steps"]:
        d[f"galois_key_{int(s)}"] = rand((l, 2), k)
    d["poly"] = rand((), 1)[0]
    a2, b2, a3, pt = d["a2"], d["b2"], d["a3"], d["pt"]
    d["out_ntt0"], d["out_intt0"] = o.ntt(0, d["poly"]), o.intt(0, d["poly"])
    d["out_add"], d["out_add_32"] = o.add(a2, b2), o.add(a3, b2)
    d["out_sub"], d["out_sub_23"] = o.sub(a2, b2), o.sub(a2, a3)
    d["out_negate"] = o.negate(a3)
    d["out_add_plain"], d["out_sub_plain"] = o.add_plain(a2, pt), o.sub_plain(a2, pt)
    d["out_multiply"], d["out_square"] = o.multiply(a2, b2), o.square(a2)
    d["out_multiply_plain"] = o.multiply_plain(a3, pt)
    d["out_relinearize"] = o.relinearize(a3, d["relin_key"])
    d["out_rescale"], d["out_rescale3"] = o.rescale(a2), o.rescale(a3)
    d["out_relin_rescale"] = o.rescale(d["out_relinearize"])
    d["out_mod_switch"] = o.mod_switch(a3)
    for s in d["rot_steps"]:
        d[f"out_rotate_{int(s)}"] = o.rotate(a2, int(s), d[f"galois_key_{int(s)}"])
    d["out_triple"] = o.op_triple(a2, b2, d["relin_key"])
    d["enc_scale_bits"] = np.array([20, 30, 40, 55, 60], dtype=np.int64)
    for c, sb in enumerate(d["enc_scale_bits"]):
        d[f"enc_values_{c}"] = rng.uniform(-3, 3, N // 2)
        d[f"out_encode_{c}"] = o.encode(l, d[f"enc_values_{c}"], 2.0 ** int(sb))
    # Decryptor::decrypt / CKKSEncoder::decode (as in make_golden.py)
    small = rng.integers(-1, 2, size=N)
    d["sk_ntt"] = np.stack([o.ntt(i, np.array([int(v) % primes[i] for v in small], dtype=np.uint64)) for i in range(k)])
    d["out_decrypt2"], d["out_decrypt3"] = o.decrypt(a2, d["sk_ntt"]), o.decrypt(a3, d["sk_ntt"])
    for c, sb in enumerate(d["enc_scale_bits"]):
        d[f"out_decode_{c}"] = o.decode(d[f"out_encode_{c}"], 2.0 ** int(sb))
    d["out_decode_pt"] = o.decode(pt, 2.0 ** 10)
    d["out_decode_dec3"] = o.decode(d["out_decrypt3"], 2.0 ** 10)
    return d, bits


def add_shortcut_vectors(d, N, seed=4):
    """The two places where the HIP product does NOT follow SEAL's order of operations and claims the same bits anyway
    (DESIGN.md 4.1 and 4): a hoisted rotation set — 8 rotations of ONE source, decomposed once — and the fused
    multiply -> relinearize -> rescale.  The oracle computes them the way SEAL does (rotate, then switch keys, one call
    per step; three separate evaluator calls), so these are the first vectors a SEAL host pins.  Sources: a dense
    ciphertext, one whose c1 has a zero limb (more zero digit coefficients than the hoisted path corrects one by one:
    its guarded exact fallback), and a transparent one (c1 = 0)."""
    sys.path.insert(0, ROOT)
    from oracle import pyoracle as po
    primes = [int(q) for q in d["primes"]]
    o = po.Oracle(N, primes)
    k, l = len(primes), len(primes) - 1
    rng = np.random.default_rng(seed)
    steps = np.array([1, 2, 3, 5, -1, -2, 7, N // 2 - 3], dtype=np.int64)  # distinct Galois elements (N/2 - s is the element of -s)
    d["hoist_steps"] = steps
    for s in steps:
        d[f"galois_key_h{int(s)}"] = np.stack([rng.integers(0, primes[i], size=(l, 2, N), dtype=np.uint64) for i in range(k)], axis=2)
    dense = np.array(d["a2"])
    zero_limb = np.array(d["b2"])
    zero_limb[1, 0, :] = 0
    transparent = np.array(d["a2"])
    transparent[1] = 0
    d["hoist_src_dense"], d["hoist_src_zero_limb"], d["hoist_src_transparent"] = dense, zero_limb, transparent
    for name in ("dense", "zero_limb", "transparent"):
        for s in steps:
            d[f"out_hoist_{name}_{int(s)}"] = o.rotate(d[f"hoist_src_{name}"], int(s), d[f"galois_key_h{int(s)}"])
    # fused multiply -> relinearize -> rescale, also as a square (seal_executor.h:161-164, :200, :213)
    d["out_triple_square"] = o.rescale(o.relinearize(o.square(d["a2"]), d["relin_key"]))
    return d


def write_seal_objects(out, d, N):
    """parameters, two ciphertexts, a plaintext, a public / secret key and both key sets in SEAL's object format"""
    sys.path.insert(0, ROOT)
    try:
        from eva_amd import _eva
    except Exception as e:  # the host module is not built: the arrays are still exported
        print(f"SEAL objects not written ({e})")
        return 0
    primes = [int(q) for q in d["primes"]]
    k = len(primes)
    scale = 2.0 ** 10
    if "pk" not in d:  # any residues serve: the public key is only carried, never used
        rng = np.random.default_rng(7)
        d["pk"] = np.stack([rng.integers(0, primes[i], size=(2, N), dtype=np.uint64) for i in range(k)], axis=1)

    def elt_of(step):
        return pow(3, step if step > 0 else N // 2 + step, 2 * N)
    blobs = {
        "seal_parms": _eva._seal_blob("parms", N, primes, np.zeros((1,), dtype=np.uint64)),
        "seal_ct_a2": _eva._seal_blob("ciphertext", N, primes, d["a2"], scale),
        "seal_ct_a3": _eva._seal_blob("ciphertext", N, primes, d["a3"], scale),
        "seal_pt": _eva._seal_blob("plaintext", N, primes, d["pt"], scale),
        "seal_pk": _eva._seal_blob("public_key", N, primes, d["pk"]),
        "seal_sk": _eva._seal_blob("secret_key", N, primes, d["sk_ntt"]),
        "seal_relin": _eva._seal_blob("relin_keys", N, primes, d["relin_key"]),
        "seal_galois": _eva._seal_galois_blob(N, primes, {elt_of(int(s)): d[f"galois_key_{int(s)}"] for s in d["rot_steps"]}),
    }
    for name, b in blobs.items():
        open(os.path.join(out, name + ".bin"), "wb").write(b)
    return len(blobs)


def main():
    if len(sys.argv) < 2:
        raise SystemExit(__doc__)
    out = sys.argv[1]
    os.makedirs(out, exist_ok=True)
    if len(sys.argv) >= 4:
        d, bits = fresh(int(sys.argv[2]), [int(b) for b in sys.argv[3].split(",")])
        N = int(sys.argv[2])
    else:
        d, bits, N = dict(np.load(os.path.join(HERE, "ops_n1024.npz"))), [60, 40, 60], 1024
    d = add_shortcut_vectors(d, N)
    with open(os.path.join(out, "manifest.txt"), "w") as f:
        f.write("# vectors for tools/seal_parity.cpp; every ciphertext / plaintext carries scale 2^scale_log2\n")
        f.write(f"N {N}\nbits {' '.join(str(b) for b in bits)}\nscale_log2 10\n")
        f.write("rot_steps " + " ".join(str(int(s)) for s in d["rot_steps"]) + "\n")
        f.write("enc_scale_bits " + " ".join(str(int(s)) for s in d["enc_scale_bits"]) + "\n")
        f.write("hoist_steps " + " ".join(str(int(s)) for s in d["hoist_steps"]) + "\n")
    n_blobs = write_seal_objects(out, d, N)
    n = 0
    for name, a in d.items():
        if name in ("rot_steps", "enc_scale_bits", "hoist_steps"):
            continue
        ext = ".f64" if a.dtype == np.float64 else ".u64"
        np.ascontiguousarray(a).astype("<f8" if ext == ".f64" else "<u8").tofile(os.path.join(out, name + ext))
        n += 1
    print(f"wrote {n} arrays + {n_blobs} SEAL objects + manifest.txt to {out}")


if __name__ == "__main__":
    main()
