"""Algorithmic HBM bytes of a compiled program: every input word of an op read
once, every output word written once, key words read once; transforms inside an op count as zero.
With P = l * N * 8 bytes per polynomial at l limbs:

    add / sub (sizes sa, sb)      (sa + sb + max(sa, sb)) P      (6P for 2 + 2)
    add_plain / sub_plain (s)     (s + 1 + s) P
    negate (s)                    2 s P
    multiply_plain (s)            (2 s + 1) P
    multiply 2 x 2                7 P           square 5 P
    relinearize                   5 P + 2 l (l + 1) N 8
    rotate (step != 0)            4 P + 2 l (l + 1) N 8
    rescale (s)                   s P + s (l - 1) N 8
    mod_switch (s)                2 s (l - 1) N 8   (the library returns a view: the bytes the reference copies)

`dag_bytes` walks the compiler's output the way the executor does — shapes (size, limbs) are
propagated from the signature's input levels — and returns the total plus a per-op breakdown.  It is
what bench.py divides by the measured execute() time to get `roofline.achieved` for the DAG legs
(reference walk: /root/reference/eva/seal/seal_executor.h:279-404)."""
from . import Op


def op_bytes(kind, N, l, sa=2, sb=2):
    P = l * N * 8
    key = 2 * l * (l + 1) * N * 8
    if kind in ("add", "sub"):
        return (sa + sb + max(sa, sb)) * P
    if kind in ("add_plain", "sub_plain"):
        return (2 * sa + 1) * P
    if kind == "negate":
        return 2 * sa * P
    if kind == "multiply_plain":
        return (2 * sa + 1) * P
    if kind == "multiply":
        return 7 * P
    if kind == "square":
        return 5 * P
    if kind == "relinearize":
        return 5 * P + key
    if kind == "rotate":
        return 4 * P + key
    if kind == "rescale":
        return sa * P + sa * (l - 1) * N * 8
    if kind == "mod_switch":
        return 2 * sa * (l - 1) * N * 8
    raise ValueError(kind)


def _walk(compiled, signature, N, n_primes):
    """yields (kind, bytes, key) per ciphertext op of the compiled list; key = None or (level, which, l) naming the
    evaluation key the op reads and the scheduler level it runs at (depth = longest path from the placed values, as
    csrc/scheduler.hip buckets its op list): ops of one level that read one key form one launch set"""
    k = n_primes
    shape = {}   # term -> ("ct", size, limbs) | ("pt", limbs) | ("raw",)
    depth = {}
    in_ids = {t.index: name for name, t in compiled.inputs.items()}
    for d in compiled._dump():
        t, op, a = d["id"], d["op"], d["operands"]
        depth[t] = 0
        if op == Op.Input:
            info = signature.inputs[in_ids[t]]
            kind = str(info.input_type).split(".")[-1]
            limbs = k - 1 - info.level
            shape[t] = ("ct", 2, limbs) if kind == "Cipher" else ("pt", limbs) if kind == "Plain" else ("raw",)
            continue
        if op == Op.Constant:
            shape[t] = ("raw",)
            continue
        if op == Op.Encode:
            shape[t] = ("pt", k - 1 - d["encode_level"])
            continue
        x = [shape[i] for i in a]
        cts = [s for s in x if s[0] == "ct"]
        if not cts:  # arithmetic on unencrypted values / their outputs: host work, no HBM traffic
            shape[t] = x[0] if op == Op.Output else ("raw",)
            continue
        depth[t] = 1 + max(depth[i] for i in a)
        if op == Op.Output:
            shape[t] = x[0]
            continue
        c0 = cts[0]
        l = c0[2]
        if op in (Op.Add, Op.Sub):
            name = "add" if op == Op.Add else "sub"
            if len(cts) == 2:
                yield name, op_bytes(name, N, l, cts[0][1], cts[1][1]), None
                shape[t] = ("ct", max(cts[0][1], cts[1][1]), l)
            else:
                yield name + "_plain", op_bytes(name + "_plain", N, l, c0[1]), None
                shape[t] = c0
        elif op == Op.Mul:
            if len(cts) == 2:
                same = a[0] == a[1]
                yield ("square" if same else "multiply"), op_bytes("square" if same else "multiply", N, l), None
                shape[t] = ("ct", 3, l)
            else:
                yield "multiply_plain", op_bytes("multiply_plain", N, l, c0[1]), None
                shape[t] = c0
        elif op == Op.Negate:
            yield "negate", op_bytes("negate", N, l, c0[1]), None
            shape[t] = c0
        elif op in (Op.RotateLeftConst, Op.RotateRightConst):
            r = d.get("rotation", 0)
            if r:
                step = (r if op == Op.RotateLeftConst else -r) % (N // 2)
                yield "rotate", op_bytes("rotate", N, l), (depth[t], ("galois", step), l)
            shape[t] = c0
        elif op == Op.Relinearize:
            yield "relinearize", op_bytes("relinearize", N, l), (depth[t], "relin", l)
            shape[t] = ("ct", 2, l)
        elif op == Op.Rescale:
            yield "rescale", op_bytes("rescale", N, l, c0[1]), None
            shape[t] = ("ct", c0[1], l - 1)
        elif op == Op.ModSwitch:
            yield "mod_switch", op_bytes("mod_switch", N, l, c0[1]), None
            shape[t] = ("ct", c0[1], l - 1)
        else:
            raise ValueError(f"unexpected op {op}")


def key_bytes(N, l):
    """one evaluation key as a key switch at l limbs reads it: l digits x 2 polynomials x (l + 1) primes"""
    return 2 * l * (l + 1) * N * 8


def dag_bytes(compiled, signature, N, n_primes):
    """-> (total algorithmic bytes of one execute(), {op kind: (count, bytes)})"""
    by = {}
    total = 0
    for kind, b, _ in _walk(compiled, signature, N, n_primes):
        c, s = by.get(kind, (0, 0))
        by[kind] = (c + 1, s + b)
        total += b
    return total, by


def dag_compulsory_bytes(compiled, signature, N, n_primes, instances=1):
    """Bytes `instances` DAGs of ONE launch set must move: SURVEY.md 8(d) charges an evaluation key to every key
    switch; here each distinct (scheduler level, key, limb count) is charged ONCE — the sibling rotations of a hoisted
    set, the relinearizations of one level and the instances of a batched handle read one copy of their key.
    -> (bytes of the whole set, {"keys_charged_8d": n, "keys_distinct": m, "key_bytes_8d": …, "key_bytes_once": …})"""
    data = 0
    charged, distinct = 0, {}
    kb_all = 0
    for kind, b, key in _walk(compiled, signature, N, n_primes):
        if key is None:
            data += b
            continue
        kb = key_bytes(N, key[2])
        data += b - kb
        kb_all += kb
        charged += 1
        distinct[key] = kb
    once = sum(distinct.values())
    return instances * data + once, {"keys_charged_8d": charged, "keys_distinct": len(distinct),
                                     "key_bytes_8d": kb_all, "key_bytes_once": once, "instances": instances}


def roofline(total_bytes, seconds, peak_gbps=8000.0, compulsory=None):
    """the `roofline` object of a bench line for one unit of `total_bytes` done in `seconds`; `compulsory` =
    (bytes, detail) of dag_compulsory_bytes for the SAME unit adds the fraction with every key charged once per launch set"""
    ach = total_bytes / seconds / 1e9
    out = {"bound": "hbm", "achieved": round(ach, 1), "peak": peak_gbps, "unit": "GB/s",
           "frac": round(ach / peak_gbps, 4), "bytes_per_unit": int(total_bytes),
           "basis": "SURVEY.md 8(d) algorithmic bytes summed over the compiled op list (eva_amd/roofline.py) / measured time"}
    if compulsory is not None:
        cb, detail = compulsory
        out.update({"launch_compulsory_bytes": int(cb), "launch_compulsory_frac": round(cb / seconds / 1e9 / peak_gbps, 4),
                    "launch_compulsory": dict(detail, note="8(d) charges an evaluation key to every key switch; here each distinct "
                                                            "(scheduler level, key, limbs) is read once per launch set")})
    return out


def csrc_tree_hash(root=None):
    """sha256 (first 16 hex digits) over the device sources the library is built from (eva_amd/csrc/*, sorted by name).
    The committed counter summaries (profiles/bench_pmc_traffic.json, bench_valu_issue.json) carry the hash of the
    tree they were measured on; bench.py recomputes it and marks them stale when the kernels have changed since."""
    import hashlib
    import os
    root = root or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    d = os.path.join(root, "eva_amd", "csrc")
    h = hashlib.sha256()
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".h")):
            h.update(name.encode())
            h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()[:16]
