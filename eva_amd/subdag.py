"""Independent sub-DAGs of ONE program on several GPUs: e.g. the three 3x3
convolutions Sxx, Syy, Sxy of the Harris corner detector
(/root/reference/examples/image_processing.py:65-100) are independent between the products Ix^2,
Iy^2, Ix*Iy and the determinant / trace that joins them.  The reference's only parallelism inside a
program is node-level on the host (MulticoreProgramTraversal,
/root/reference/eva/common/multicore_program_traversal.h:55-78); here the encrypted part of the
compiled program is cut into

    prefix (device 0)  ->  independent components, dealt over the devices  ->  suffix (device 0)

each piece one evah_execute submit (asynchronous, so the devices work concurrently), with a peer
copy (evah_ct_copy, xGMI) for every ciphertext that crosses a cut: the operands of the components
and the two-or-so ciphertexts at the join.  Results are the ciphertexts of the single-device run,
bit for bit — the partition only decides where each node runs.

The devices are any list of device indices; repeating an index (e.g. [0, 0, 0]) gives several
contexts on one GPU, which is how the path is validated on a single-GPU box.
"""
import numpy as np

from . import Op, backend

HEAVY = {int(Op.RotateLeftConst), int(Op.RotateRightConst), int(Op.Relinearize), int(Op.Rescale)}


def lower(program, enc_inputs, encode):
    """compiled program -> (ops, placed, outs, raw): the encrypted part as a flat list
    [(op, dst, src0, src1, imm)] of the reference's op codes (eva/ir/ops.h:11-25) over value slots;
    `placed` = {slot: ("ct" | "pt", array, scale)} are the inputs and the plaintexts of Encode nodes
    (encode(values, scale_bits, level) -> residues); unencrypted nodes are evaluated here, as
    SEALExecutor does on the host (seal_executor.h:63-112).  raw = {slot: list} for outputs that
    depend on unencrypted values only."""
    dump = program._dump()
    raw, placed, ops = {}, {}, []
    inputs = {name: t.index for name, t in program.inputs.items()}
    for name in enc_inputs.names():
        kind, size, limbs, scale, data = enc_inputs.get(name)
        t = inputs[name]
        if kind == "cipher":
            placed[t] = ("ct", data, scale)
        elif kind == "plain":
            placed[t] = ("pt", data, scale)
        else:
            raw[t] = list(data) * (program.vec_size // len(data))

    def rot(v, s, left):
        s %= len(v)
        return v[s:] + v[:s] if left else v[len(v) - s:] + v[:len(v) - s]

    for d in dump:
        t, op, a = d["id"], d["op"], d["operands"]
        if op == Op.Input:
            continue
        if op == Op.Constant:
            raw[t] = list(d["constant"]) * (program.vec_size // len(d["constant"]))
        elif op == Op.Encode:
            placed[t] = ("pt", encode(raw[a[0]], d["encode_scale"], d["encode_level"]), 2.0 ** d["encode_scale"])
        elif all(x in raw for x in a):
            x = [raw[i] for i in a]
            if op == Op.Add: raw[t] = [u + v for u, v in zip(*x)]
            elif op == Op.Sub: raw[t] = [u - v for u, v in zip(*x)]
            elif op == Op.Mul: raw[t] = [u * v for u, v in zip(*x)]
            elif op == Op.Negate: raw[t] = [-u for u in x[0]]
            elif op in (Op.RotateLeftConst, Op.RotateRightConst): raw[t] = rot(x[0], d["rotation"], op == Op.RotateLeftConst)
            else: raw[t] = list(x[0])  # Output / scale management of an unencrypted value
        else:
            imm = d.get("rotation", d.get("rescale_divisor", 0)) or 0
            ops.append((int(op), t, a[0], a[1] if len(a) > 1 else 0, int(imm)))
    outs = {name: t.index for name, t in program.outputs.items()}
    return ops, placed, outs, raw


def _arity(op):
    return 2 if op in (int(Op.Add), int(Op.Sub), int(Op.Mul)) else 1


def plan(ops, placed, n_dev):
    """-> (prefix, components, suffix): lists of op indices; components is a list of
    (device, [op indices]).  The cut is the pair of levels between which the op DAG falls into
    the most evenly loaded independent components; n_dev == 1 or no worthwhile cut -> everything
    in the prefix."""
    n = len(ops)
    producer = {o[1]: i for i, o in enumerate(ops)}
    level = [0] * n
    for i, o in enumerate(ops):
        for s in (o[2], o[3])[:_arity(o[0])]:
            if s in producer:
                level[i] = max(level[i], level[producer[s]] + 1)
    cost = [10 if (o[0] in HEAVY or (o[0] == int(Op.Mul) and o[2] in producer and o[3] in producer and
                                     _is_ct(o[2], producer, placed) and _is_ct(o[3], producer, placed))) else 1 for o in ops]
    depth = max(level) + 1 if n else 0
    serial = sum(cost)
    best = (serial, None)
    if n_dev > 1:
        for lo in range(depth):
            for hi in range(lo + 1, depth + 1):
                region = [i for i in range(n) if lo <= level[i] < hi]
                if len(region) < 2:
                    continue
                parent = {i: i for i in region}

                def find(x):
                    while parent[x] != x:
                        parent[x] = parent[parent[x]]
                        x = parent[x]
                    return x
                inreg = set(region)
                for i in region:
                    for s in (ops[i][2], ops[i][3])[:_arity(ops[i][0])]:
                        j = producer.get(s)
                        if j is not None and j in inreg:
                            parent[find(i)] = find(j)
                comps = {}
                for i in region:
                    comps.setdefault(find(i), []).append(i)
                if len(comps) < 2:
                    continue
                loads = [0] * n_dev
                assign = []
                for c in sorted(comps.values(), key=lambda c: -sum(cost[i] for i in c)):  # longest first
                    d = loads.index(min(loads))
                    loads[d] += sum(cost[i] for i in c)
                    assign.append((d, sorted(c)))
                crossing = sum(1 for d, c in assign if d != 0 for i in c for s in (ops[i][2], ops[i][3])[:_arity(ops[i][0])]
                               if s not in {ops[j][1] for j in c})
                est = sum(cost[i] for i in range(n) if level[i] < lo or level[i] >= hi) + max(loads) + 2 * crossing
                if est < best[0]:
                    best = (est, (lo, hi, assign))
    if best[1] is None or best[0] > 0.9 * serial:
        return list(range(n)), [], []
    lo, hi, assign = best[1]
    return [i for i in range(n) if level[i] < lo], assign, [i for i in range(n) if level[i] >= hi]


def _is_ct(slot, producer, placed):
    return slot in producer or (slot in placed and placed[slot][0] == "ct")


class SubDagExecutor:
    """execute() of a compiled program with its independent sub-DAGs on several devices."""

    def __init__(self, pub, devices):
        self.pub, self.devices = pub, list(devices)
        N, primes = pub.poly_modulus_degree, list(pub.primes)
        self.ctx = []
        for i, d in enumerate(self.devices):
            same = next((self.ctx[j] for j in range(i) if self.devices[j] == d), None)
            # contexts on one device share tables and keys (forks); a new device gets its own copy
            self.ctx.append(same.fork() if same is not None else backend.Context(N, primes, device=d))
            if same is None:
                self.ctx[-1].upload_relin_key(pub.relin_key())
                for elt, key in pub.galois_keys().items():
                    self.ctx[-1].upload_galois_key(elt, key)
        self.last_plan = None

    def close(self):
        for c in reversed(self.ctx):
            c.close()

    def _run(self, d, idx, ops, vals, consumers):
        """one evah_execute on device d over ops[idx]; vals[d] gains the produced slots"""
        if not idx:
            return
        c = self.ctx[d]
        inside = {ops[i][1] for i in idx}
        seen, sub = {}, []
        reads = {}
        for i in idx:
            for s in (ops[i][2], ops[i][3])[:_arity(ops[i][0])]:
                reads[s] = reads.get(s, 0) + 1
        for i in idx:
            o = ops[i]
            flags = 0
            srcs = (o[2], o[3])[:_arity(o[0])]
            for pos, s in enumerate(srcs):
                if pos == 1 and o[2] == o[3]:
                    continue
                seen[s] = seen.get(s, 0) + (2 if len(srcs) == 2 and o[2] == o[3] else 1)
                # an intermediate of this piece whose every reader is in this piece: released at its last use
                if s in inside and consumers.get(s, 0) == reads[s] and seen[s] == reads[s]:
                    flags |= backend.OPF_FREE_SRC0 if pos == 0 else backend.OPF_FREE_SRC1
            sub.append((o[0], o[1], o[2], o[3], o[4], flags))
        needed = {s for i in idx for s in (ops[i][2], ops[i][3])[:_arity(ops[i][0])] if s not in inside}
        res = c.execute(sub, {s: vals[d][s] for s in needed}, n_vals=self.n_vals)
        for s, h in res.items():
            vals[d][s] = h

    def execute(self, program, enc_inputs):
        """-> {output name: (array, scale)} for ciphertext outputs, {name: list} for unencrypted ones"""
        ops, placed, outs, raw = lower(program, enc_inputs, self.pub._encode)
        self.n_vals = 1 + max([max(o[1], o[2], o[3]) for o in ops] + list(placed) + [0])
        prefix, comps, suffix = plan(ops, placed, len(self.ctx))
        self.last_plan = {"prefix": len(prefix), "components": [(d, len(c)) for d, c in comps], "suffix": len(suffix)}
        consumers = {}
        for o in ops:
            for s in (o[2], o[3])[:_arity(o[0])]:
                consumers[s] = consumers.get(s, 0) + 1
        vals = [dict() for _ in self.ctx]

        def fetch(d, s):
            """value of slot s on device d: upload a placed value, or copy it from the device that holds it"""
            if s in vals[d]:
                return
            if s in placed:
                kind, data, scale = placed[s]
                vals[d][s] = self.ctx[d].upload_ct(data, scale) if kind == "ct" else self.ctx[d].upload_pt(data, scale)
                return
            src = next(e for e in range(len(vals)) if s in vals[e])
            vals[d][s] = self.ctx[d].copy_here(vals[src][s])

        def run(d, idx):
            inside = {ops[i][1] for i in idx}
            for i in idx:
                for s in (ops[i][2], ops[i][3])[:_arity(ops[i][0])]:
                    if s not in inside:
                        fetch(d, s)
            self._run(d, idx, ops, vals, consumers)

        run(0, prefix)
        for d, c in comps:
            run(d, c)
        run(0, suffix)
        result = {}
        for name, s in outs.items():
            if s in raw:
                result[name] = raw[s]
                continue
            d = next(e for e in range(len(vals)) if s in vals[e])
            h = vals[d][s]
            result[name] = (h.download(), h.scale)
        return result
