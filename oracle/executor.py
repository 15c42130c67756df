"""Checker-side restatement of SEALExecutor's dispatch over the CPU oracle
(/root/reference/eva/seal/seal_executor.h:279-404): walks a compiled program's term list and
calls one oracle op per node.  Used to check execute() on the GPU bit-for-bit at the DAG level.
Test infrastructure only: imported by tests/ and by bench.py's checker / cpu_baseline code, never by the product."""
import numpy as np

from . import pyoracle as po


class Cipher:
    def __init__(self, data, scale):
        self.data, self.scale = data, scale


class Plain:
    def __init__(self, data, scale):
        self.data, self.scale = data, scale


def _rot(v, s, left):
    n = len(v)
    s %= n
    return v[s:] + v[:s] if left else v[n - s:] + v[:n - s]


class OracleExecutor:
    def __init__(self, public_ctx):
        self.pub = public_ctx
        self.N = public_ctx.poly_modulus_degree
        self.primes = list(public_ctx.primes)
        self.k = len(self.primes)
        self.o = po.Oracle(self.N, self.primes)
        self.relin = public_ctx.relin_key()
        self.galois = public_ctx.galois_keys()

    def _eval(self, program, d, vals):
        """value of one node given its operands' values (SEALExecutor::operator(), :279-404)"""
        from eva_amd import Op
        t, op, a = d["id"], d["op"], d["operands"]
        if op == Op.Constant:
            c = d["constant"]
            return list(c) * (program.vec_size // len(c))
        elif op == Op.Encode:
            # encoder.encode at 2^scale and the level's limb count (seal_executor.h:217-243), by the
            # ORACLE's encoder: the walk borrows nothing from the product
            v = list(vals[a[0]])
            slots = self.N // 2
            if not v or slots % len(v):
                raise RuntimeError("Size must exactly divide slots")
            limbs = self.k - 1 - d["encode_level"]
            data = self.o.encode(limbs, np.array(v * (slots // len(v)), dtype=np.float64), 2.0 ** d["encode_scale"])
            return Plain(data, 2.0 ** d["encode_scale"])
        elif op in (Op.Add, Op.Sub, Op.Mul):
            x, y = vals[a[0]], vals[a[1]]
            if isinstance(x, list) and isinstance(y, list):
                f = {Op.Add: lambda u, v: u + v, Op.Sub: lambda u, v: u - v, Op.Mul: lambda u, v: u * v}[op]
                return [f(u, v) for u, v in zip(x, y)]
            elif op == Op.Add:
                if not isinstance(x, Cipher):
                    x, y = y, x
                if isinstance(y, Cipher):
                    return Cipher(self.o.add(x.data, y.data), x.scale)
                else:
                    return Cipher(self.o.add_plain(x.data, y.data), x.scale)
            elif op == Op.Sub:
                if not isinstance(x, Cipher):  # seal_executor.h:139: std::get<Ciphertext>(args1)
                    raise RuntimeError("Unsupported operation encountered")
                if isinstance(y, Cipher):
                    return Cipher(self.o.sub(x.data, y.data), x.scale)
                else:
                    return Cipher(self.o.sub_plain(x.data, y.data), x.scale)
            else:
                same = a[0] == a[1]
                if not isinstance(x, Cipher):
                    x, y = y, x
                if isinstance(y, Cipher):
                    out = self.o.square(x.data) if same else self.o.multiply(x.data, y.data)
                else:
                    out = self.o.multiply_plain(x.data, y.data)
                return Cipher(out, x.scale * y.scale)
        elif op in (Op.RotateLeftConst, Op.RotateRightConst):
            x = vals[a[0]]
            if isinstance(x, list):
                return _rot(x, d["rotation"], op == Op.RotateLeftConst)
            else:
                steps = d["rotation"] if op == Op.RotateLeftConst else -d["rotation"]
                key = None
                if steps != 0:
                    key = self.galois[po.galois_elt_from_step(self.N, steps)]
                return Cipher(self.o.rotate(x.data, steps, key), x.scale)
        elif op == Op.Negate:
            x = vals[a[0]]
            return [-u for u in x] if isinstance(x, list) else Cipher(self.o.negate(x.data), x.scale)
        elif op in (Op.Relinearize, Op.ModSwitch, Op.Rescale) and isinstance(vals[a[0]], list):
            # a scale-management node on an unencrypted value (the reduction balancer can pair
            # constants: raw x raw products the rescaler then treats like any product) is a copy,
            # as in the reference's semantic executor (eva/common/reference_executor.cpp)
            return vals[a[0]]
        elif op == Op.Relinearize:
            x = vals[a[0]]
            return Cipher(self.o.relinearize(x.data, self.relin), x.scale)
        elif op == Op.ModSwitch:
            x = vals[a[0]]
            return Cipher(self.o.mod_switch(x.data), x.scale)
        elif op == Op.Rescale:
            x = vals[a[0]]
            return Cipher(self.o.rescale(x.data), x.scale / 2.0 ** d["rescale_divisor"])
        elif op == Op.Output:
            return vals[a[0]]
        else:
            raise RuntimeError(f"Unhandled op {op}")

    def execute(self, program, enc_inputs, threads=1):
        vals = {}
        inputs = {name: t.index for name, t in program.inputs.items()}
        for name in enc_inputs.names():
            kind, size, limbs, scale, data = enc_inputs.get(name)
            t = inputs[name]
            if kind == "cipher":
                vals[t] = Cipher(data, scale)
            elif kind == "plain":
                vals[t] = Plain(data, scale)
            else:
                vals[t] = list(data) * (program.vec_size // len(data))
        from eva_amd import Op
        dump = program._dump()
        if threads <= 1:
            for d in dump:
                if d["op"] != Op.Input:
                    vals[d["id"]] = self._eval(program, d, vals)
        else:
            # node-level parallel walk (dependency counting over a thread pool; ctypes releases the
            # GIL inside the oracle) — the CPU analogue of MulticoreProgramTraversal::forwardPass
            import threading
            from concurrent.futures import ThreadPoolExecutor
            nodes = {d["id"]: d for d in dump if d["op"] != Op.Input}
            waiting = {t: sum(1 for o in set(d["operands"]) if o in nodes) for t, d in nodes.items()}
            users = {}
            for t, d in nodes.items():
                for o in set(d["operands"]):
                    if o in nodes:
                        users.setdefault(o, []).append(t)
            lock, done = threading.Lock(), threading.Event()
            left = [len(nodes)]
            errors = []
            pool = ThreadPoolExecutor(max_workers=threads)

            def run(t):
                try:
                    v = self._eval(program, nodes[t], vals)
                except Exception as e:  # noqa: BLE001
                    errors.append(e)
                    done.set()
                    return
                ready = []
                with lock:
                    vals[t] = v
                    left[0] -= 1
                    for u in users.get(t, ()):
                        waiting[u] -= 1
                        if waiting[u] == 0:
                            ready.append(u)
                    if left[0] == 0:
                        done.set()
                for u in ready:
                    pool.submit(run, u)
            for t in [t for t, w in waiting.items() if w == 0]:
                pool.submit(run, t)
            done.wait()
            pool.shutdown(wait=True)
            if errors:
                raise errors[0]
        return {name: vals[t.index] for name, t in program.outputs.items()}


def lower(program, enc_inputs, oracle, N, k):
    """compiled program -> (ops, values, outputs) for Oracle.dag_walk: the encrypted part as a flat
    list of the reference's op codes; unencrypted (vector<double>) nodes are evaluated here as
    SEALExecutor does on the host (seal_executor.h:63-112), Encode nodes by the oracle's encoder."""
    from eva_amd import Op
    dump = program._dump()
    raw, values, ops = {}, {}, []
    inputs = {name: t.index for name, t in program.inputs.items()}
    for name in enc_inputs.names():
        kind, size, limbs, scale, data = enc_inputs.get(name)
        t = inputs[name]
        if kind == "cipher":
            values[t] = ("ct", data)
        elif kind == "plain":
            values[t] = ("pt", data)
        else:
            raw[t] = list(data) * (program.vec_size // len(data))
    slots = N // 2
    for d in dump:
        t, op, a = d["id"], d["op"], d["operands"]
        if op == Op.Input:
            continue
        if op == Op.Constant:
            raw[t] = list(d["constant"]) * (program.vec_size // len(d["constant"]))
        elif op == Op.Encode:
            v = raw[a[0]]
            limbs = k - 1 - d["encode_level"]
            values[t] = ("pt", oracle.encode(limbs, np.array(v * (slots // len(v)), dtype=np.float64), 2.0 ** d["encode_scale"]))
        elif all(x in raw for x in a):
            x = [raw[i] for i in a]
            if op == Op.Add: raw[t] = [u + v for u, v in zip(*x)]
            elif op == Op.Sub: raw[t] = [u - v for u, v in zip(*x)]
            elif op == Op.Mul: raw[t] = [u * v for u, v in zip(*x)]
            elif op == Op.Negate: raw[t] = [-u for u in x[0]]
            elif op in (Op.RotateLeftConst, Op.RotateRightConst): raw[t] = _rot(x[0], d["rotation"], op == Op.RotateLeftConst)
            else: raw[t] = list(x[0])  # Output / scale management of an unencrypted value: a copy
        else:
            imm = d.get("rotation", d.get("rescale_divisor", 0)) or 0
            ops.append((int(op), t, a[0], a[1] if len(a) > 1 else 0, int(imm)))
    outs = {name: t.index for name, t in program.outputs.items()}
    return ops, values, outs, max(d["id"] for d in dump) + 1


def c_walk(public_ctx, program, enc_inputs, threads=1):
    """The compiled DAG walked in C over the oracle (oracle/eva_oracle_dag.c): serial forwardPass or
    the dependency-counting multicore traversal.  Returns ({output name: ciphertext array}, seconds
    inside the walk — lowering and encoding excluded, as key / plaintext preparation is for the GPU)."""
    N, primes = public_ctx.poly_modulus_degree, list(public_ctx.primes)
    o = po.Oracle(N, primes)
    ops, values, outs, n_vals = lower(program, enc_inputs, o, N, len(primes))
    res, dt = o.dag_walk(ops, values, n_vals, public_ctx.relin_key(), public_ctx.galois_keys(), threads)
    return {name: res[t] for name, t in outs.items() if t in res}, dt
