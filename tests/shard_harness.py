"""TEST HARNESS (r5: moved here from the shipped package, where it was a second driver of the protocol): the phases
of limb-sharded execution restated in Python over a pluggable shard backend.  The product's driver is the C++
LimbShardEvaluator behind public_ctx.execute (eva_amd/host/multi_device.h, limb_exec.h) — what bench.py --shard limb
times; this file exists so that the partition / exchange / reassembly logic can be checked WITHOUT a GPU (the CPU shard
of tests/shard_cpu_backend.py over the oracle, one process and two gloo ranks) and so that the evah_shard_* entry
points can be driven phase by phase from the tests (tests/test_gpu_shard.py).

Limb-sharded execution over G shards (BASELINE config 5): limb i of every
ciphertext and plaintext lives on shard i mod G, the special prime's limb of the key-switch products
on shard l mod G.  Elementwise operations and the per-limb transforms are local to a shard; a key
switch (relinearize, rotate) costs one all-gather of the l coefficient-form digits (l N 8 bytes)
and one broadcast of the special limb's inverse transform (2 N 8 bytes), a rescale one broadcast of
the last limb's inverse transform (size N 8 bytes).  Every stored word is the canonical residue the
unsharded path stores for that limb, so the assembled ciphertexts are bit-identical.

The reference has no counterpart (its parallelism is node-level,
/root/reference/eva/common/multicore_program_traversal.h:55-78); the SEAL calls being split are
switch_key_inplace behind relinearize / rotate_vector (/root/reference/eva/seal/seal_executor.h:200,
:181/:188) and rescale_to_next (:213).

Two deployments of the same harness:
  * one process, all G shards (`ShardedEvaluator.in_process(N, primes, G)`): the shards are G contexts —
    on G devices with peer access, or on one device — and the exchange steps are device / peer
    copies (LocalExchange).
  * one process per rank (`ShardedEvaluator.distributed(N, primes, Dist(...))`): every rank owns shard `rank`; the
    exchange steps are torch.distributed collectives directly on the library's device buffers
    (backend "nccl" = RCCL) or staged through host memory (backend "gloo": two ranks can then share one GPU, or run
    the CPU backend of the tests).

The shard backend is any object with the methods of HipShard below (libeva_hip.so through eva_amd.backend, or the
CPU backend over the oracle, tests/shard_cpu_backend.py).
"""
import numpy as np

KEY_RELIN, KEY_GALOIS = 0, 1


def local_limbs(l, shard, G):
    """limbs i < l owned by `shard`"""
    return list(range(shard, l, G))


def rows_for(l, G):
    return (l + G - 1) // G


class ShardedValue:
    """A ciphertext (size >= 1) or plaintext (size == 0) dealt over the shards.  parts[s] is shard
    s's local handle — None when the shard owns no limb at this level or is not in this process."""

    def __init__(self, parts, size, limbs, scale):
        self.parts, self.size, self.limbs, self.scale = parts, size, limbs, scale


class HipShard:
    """One shard on the MI355X backend: an evah_ctx with the limb -> prime map (shard, G)."""

    def __init__(self, N, primes, shard, G, device=0):
        from eva_amd import backend
        self.backend = backend
        self.N, self.primes, self.k, self.shard, self.G = N, list(primes), len(primes), shard, G
        # a device state of its own, and the shard map goes in BEFORE any key: evah_key_upload then keeps only
        # this shard's prime rows (its data limbs + the special prime) of every key
        self.ctx = backend.Context(N, primes, device=device)
        self.ctx.set_shard(shard, G)

    # ---- values: arrays hold the LOCAL limbs
    def upload_ct(self, data, scale):
        return self.ctx.upload_ct(np.ascontiguousarray(data), scale)

    def upload_pt(self, data, scale):
        return self.ctx.upload_pt(np.ascontiguousarray(data), scale)

    def download(self, h):
        return h.download()

    def upload_key(self, kind, elt, key):
        if kind == KEY_RELIN:
            self.ctx.upload_relin_key(key)
        else:
            self.ctx.upload_galois_key(elt, key)

    # ---- per-limb operations
    def add(self, a, b): return self.ctx.add(a, b)
    def sub(self, a, b): return self.ctx.sub(a, b)
    def negate(self, a): return self.ctx.negate(a)
    def multiply(self, a, b): return self.ctx.multiply(a, b)
    def square(self, a): return self.ctx.square(a)
    def multiply_plain(self, a, p): return self.ctx.multiply_plain(a, p)
    def add_plain(self, a, p): return self.ctx.add_plain(a, p)
    def sub_plain(self, a, p): return self.ctx.sub_plain(a, p)
    def drop_last_limb(self, a): return self.ctx.mod_switch(a)
    def galois_perm(self, a, elt): return self.ctx.shard_galois_perm(a, elt)

    # ---- phases around the exchange steps
    def buffer(self, words): return self.ctx.buffer(words)
    def ks_digits(self, a, poly, l, digits, rows): self.ctx.shard_ks_digits(a, poly, l, digits, rows)
    def ks_products(self, a, poly, l, digits, rows, kind, elt, prod, r): self.ctx.shard_ks_products(a, poly, l, digits, rows, kind, elt, prod, r)
    def ks_finish(self, l, prod, r, add, add_polys, scale): return self.ctx.shard_ks_finish(l, prod, r, add, add_polys, scale)
    def rescale_last(self, a, l, r): self.ctx.shard_rescale_last(a, l, r)
    def rescale_finish(self, a, l, r, bits): return self.ctx.shard_rescale_finish(a, l, r, bits)
    def sync(self): self.ctx.sync()
    def close(self): self.ctx.close()


class LocalExchange:
    """All shards in this process: the exchange steps are device / peer copies, enqueued on the
    receiving shard's queue and ordered after the producing shard's work by the library."""

    def all_gather(self, bufs, chunk):
        """bufs[s]: buffer of G chunks, chunk s filled by shard s -> every buffer complete; ONE launch per
        receiving shard (evah_buf_gather: a kernel on its queue reading the other shards' chunks as peers)"""
        for d, dst in bufs.items():
            others = [s for s in bufs if s != d]
            if others:
                offs = [s * chunk for s in others]
                dst.gather_from([bufs[s] for s in others], offs, offs, chunk)

    def broadcast(self, bufs, owner, words):
        for d, dst in bufs.items():
            if d != owner:
                dst.gather_from([bufs[owner]], [0], [0], words)


class DistExchange:
    """One shard per process: torch.distributed collectives.  nccl (= RCCL): in place on the
    library's device buffers, on the stream the shard's kernels run on; gloo: staged through host."""

    def __init__(self, dist):
        self.d = dist  # eva_amd.dist.Dist
        self.torch = dist.torch
        self.device_collectives = dist.backend == "nccl"

    def _view(self, buf):
        return self.torch.as_tensor(buf, device="cuda")

    def all_gather(self, bufs, chunk):
        (s, buf), = bufs.items()
        if self.device_collectives:
            full = self._view(buf)[: self.d.world * chunk]
            self.d.dist.all_gather_into_tensor(full, full[s * chunk:(s + 1) * chunk])
        else:
            mine = self.torch.from_numpy(buf.download(s * chunk, chunk).view(np.int64))
            parts = [self.torch.empty_like(mine) for _ in range(self.d.world)]
            self.d.dist.all_gather(parts, mine)
            for r, p in enumerate(parts):
                if r != s:
                    buf.upload(p.numpy().view(np.uint64), r * chunk)

    def broadcast(self, bufs, owner, words):
        (s, buf), = bufs.items()
        if self.device_collectives:
            self.d.dist.broadcast(self._view(buf)[:words], src=owner)
        else:
            t = self.torch.from_numpy(buf.download(0, words).view(np.int64)) if s == owner else self.torch.empty(words, dtype=self.torch.int64)
            self.d.dist.broadcast(t, src=owner)
            if s != owner:
                buf.upload(t.numpy().view(np.uint64), 0)


class ShardedEvaluator:
    """The evaluator calls of SEALExecutor over limb-sharded values.

    shards: {shard index: backend object} — all G of them (one process) or exactly this rank's."""

    def __init__(self, N, primes, G, shards, exchange):
        self.N, self.primes, self.k, self.G = N, list(primes), len(primes), G
        self.shards, self.x = shards, exchange

    # ---- construction helpers
    @classmethod
    def in_process(cls, N, primes, G, make_shard=None, devices=None):
        """All G shards here.  devices: one device index per shard (default: all on device 0)."""
        shards = {}
        if make_shard is None:
            for s in range(G):
                shards[s] = HipShard(N, primes, s, G, device=devices[s] if devices else 0)
        else:
            for s in range(G):
                shards[s] = make_shard(s)
        return cls(N, primes, G, shards, LocalExchange())

    @classmethod
    def distributed(cls, N, primes, dist, make_shard=None):
        """This process is shard dist.rank of dist.world (one process per GPU)."""
        s, G = dist.rank, dist.world
        if make_shard is None:
            shard = HipShard(N, primes, s, G, device=dist.device_index if dist.on_gpu else 0)
            if dist.backend == "nccl":  # kernels and collectives on one stream: no host synchronisation between phases
                shard.ctx.set_stream(dist.torch.cuda.current_stream().cuda_stream)
        else:
            shard = make_shard(s)
        return cls(N, primes, G, {s: shard}, DistExchange(dist))

    def close(self):
        for sh in reversed(list(self.shards.values())):
            sh.close()

    def _buf(self, s, words):
        """exchange buffer of one operation: allocated from the shard's pool and released at the end
        of the operation — the pool recycles it only after the other queues' reads of it"""
        return self.shards[s].buffer(words)

    # ---- values
    def upload_ct(self, data, scale):
        """data: [size][l][N] (all limbs; every process passes the same array)"""
        size, l, _ = data.shape
        parts = {s: (sh.upload_ct(data[:, s::self.G, :], scale) if s < l else None) for s, sh in self.shards.items()}
        return ShardedValue(parts, size, l, scale)

    def upload_pt(self, data, scale):
        l = data.shape[0]
        parts = {s: (sh.upload_pt(data[s::self.G, :], scale) if s < l else None) for s, sh in self.shards.items()}
        return ShardedValue(parts, 0, l, scale)

    def upload_relin_key(self, key):
        for sh in self.shards.values():
            sh.upload_key(KEY_RELIN, 0, key)

    def upload_galois_key(self, elt, key):
        for sh in self.shards.values():
            sh.upload_key(KEY_GALOIS, elt, key)

    def download(self, v):
        """-> [size][l][N] (plaintext: [l][N]) assembled from the shards of THIS process; in the
        distributed deployment use gather() to assemble across ranks."""
        return self._assemble({s: self.shards[s].download(p) for s, p in v.parts.items() if p is not None}, v)

    def _assemble(self, pieces, v):
        shape = (v.size, v.limbs, self.N) if v.size else (v.limbs, self.N)
        out = np.zeros(shape, dtype=np.uint64)
        for s, a in pieces.items():
            if v.size:
                out[:, s::self.G, :] = a
            else:
                out[s::self.G, :] = a
        return out

    def gather(self, v, dist):
        """distributed deployment: the whole value on every rank (test / output path, via host)"""
        pieces = {s: self.shards[s].download(p) for s, p in v.parts.items() if p is not None}
        allp = [None] * dist.world
        dist.dist.all_gather_object(allp, pieces)
        merged = {}
        for p in allp:
            merged.update(p)
        return self._assemble(merged, v)

    # ---- per-limb operations (no exchange)
    def _each(self, fn, *vals):
        return {s: (fn(sh, *[v.parts[s] for v in vals]) if all(v.parts[s] is not None for v in vals) else None)
                for s, sh in self.shards.items()}

    def _same_level(self, a, b, scales=True):
        if a.limbs != b.limbs:
            raise ValueError("encrypted1 and encrypted2 parameter mismatch")
        if scales and a.scale != b.scale:
            raise ValueError("scale mismatch")

    def add(self, a, b):
        self._same_level(a, b)
        return ShardedValue(self._each(lambda sh, x, y: sh.add(x, y), a, b), max(a.size, b.size), a.limbs, a.scale)

    def sub(self, a, b):
        self._same_level(a, b)
        return ShardedValue(self._each(lambda sh, x, y: sh.sub(x, y), a, b), max(a.size, b.size), a.limbs, a.scale)

    def add_plain(self, a, p):
        self._same_level(a, p)
        return ShardedValue(self._each(lambda sh, x, y: sh.add_plain(x, y), a, p), a.size, a.limbs, a.scale)

    def sub_plain(self, a, p):
        self._same_level(a, p)
        return ShardedValue(self._each(lambda sh, x, y: sh.sub_plain(x, y), a, p), a.size, a.limbs, a.scale)

    def negate(self, a):
        return ShardedValue(self._each(lambda sh, x: sh.negate(x), a), a.size, a.limbs, a.scale)

    def _check_scale(self, scale, limbs):
        bits = 1
        for q in self.primes[:limbs]:
            bits *= q
        if not scale > 0 or int(np.log2(scale)) >= bits.bit_length():
            raise ValueError("scale out of bounds")

    def multiply(self, a, b):
        self._same_level(a, b, scales=False)
        if a.size != 2 or b.size != 2:
            raise ValueError("multiply supports size-2 operands only (relinearize first)")
        self._check_scale(a.scale * b.scale, a.limbs)
        return ShardedValue(self._each(lambda sh, x, y: sh.multiply(x, y), a, b), 3, a.limbs, a.scale * b.scale)

    def square(self, a):
        if a.size != 2:
            raise ValueError("square supports size-2 operands only (relinearize first)")
        self._check_scale(a.scale * a.scale, a.limbs)
        return ShardedValue(self._each(lambda sh, x: sh.square(x), a), 3, a.limbs, a.scale * a.scale)

    def multiply_plain(self, a, p):
        self._same_level(a, p, scales=False)
        self._check_scale(a.scale * p.scale, a.limbs)
        return ShardedValue(self._each(lambda sh, x, y: sh.multiply_plain(x, y), a, p), a.size, a.limbs, a.scale * p.scale)

    def mod_switch(self, a):
        """drop the last limb: a view on its owner, nothing on the other shards"""
        if a.limbs < 2:
            raise ValueError("end of modulus switching chain reached")
        owner = (a.limbs - 1) % self.G
        parts = dict(a.parts)
        if owner in parts and parts[owner] is not None:
            parts[owner] = self.shards[owner].drop_last_limb(parts[owner]) if a.limbs - 1 > owner else None
        return ShardedValue(parts, a.size, a.limbs - 1, a.scale)

    # ---- operations with an exchange step
    def _key_switch(self, target, poly, l, kind, elt, add, add_polys, scale):
        """target.parts[s] holds the key-switch target as polynomial `poly`; returns the size-2 result"""
        N, G, rows = self.N, self.G, rows_for(l, self.G)
        chunk = rows * N
        dig = {s: self._buf(s, G * chunk) for s in self.shards}
        for s, sh in self.shards.items():
            if target.parts[s] is not None:
                sh.ks_digits(target.parts[s], poly, l, dig[s], rows)
        self.x.all_gather(dig, chunk)                        # ---- exchange 1: the l digits
        owner = l % G
        prod = {s: self._buf(s, 2 * (len(local_limbs(l, s, G)) + 1) * N) for s in self.shards}
        rbuf = {s: self._buf(s, 3 * N) for s in self.shards}
        for s, sh in self.shards.items():
            if target.parts[s] is not None or s == owner:
                sh.ks_products(target.parts[s], poly, l, dig[s], rows, kind, elt, prod[s], rbuf[s])
        self.x.broadcast(rbuf, owner, 2 * N)                 # ---- exchange 2: INTT of the special limb
        parts = {}
        for s, sh in self.shards.items():
            if s < l:
                parts[s] = sh.ks_finish(l, prod[s], rbuf[s], add.parts[s] if add is not None else None, add_polys, scale)
            else:
                parts[s] = None
        return ShardedValue(parts, 2, l, scale)

    def relinearize(self, a):
        if a.size != 3:
            raise ValueError("relinearize expects a size-3 ciphertext")
        return self._key_switch(a, 2, a.limbs, KEY_RELIN, 0, a, 2, a.scale)

    def galois_elt_from_step(self, steps):
        m = 2 * self.N
        if steps == 0:
            return m - 1
        pos = abs(steps)
        if pos >= self.N // 2:
            raise ValueError("step count too large")
        s = self.N // 2 - pos if steps < 0 else pos
        return pow(3, s, m)

    def rotate(self, a, steps):
        if a.size != 2:
            raise ValueError("rotate expects a size-2 ciphertext (relinearize first)")
        if steps == 0:
            return a
        elt = self.galois_elt_from_step(steps)
        perm = ShardedValue(self._each(lambda sh, x: sh.galois_perm(x, elt), a), 2, a.limbs, a.scale)
        return self._key_switch(perm, 1, a.limbs, KEY_GALOIS, elt, perm, 1, a.scale)

    def rescale(self, a, divisor_bits):
        l = a.limbs
        if l < 2:
            raise ValueError("end of modulus switching chain reached")
        owner = (l - 1) % self.G
        rbuf = {s: self._buf(s, 3 * self.N) for s in self.shards}
        if owner in self.shards:
            self.shards[owner].rescale_last(a.parts[owner], l, rbuf[owner])
        self.x.broadcast(rbuf, owner, a.size * self.N)       # ---- exchange: INTT of the last limb
        parts = {}
        for s, sh in self.shards.items():
            parts[s] = sh.rescale_finish(a.parts[s], l, rbuf[s], divisor_bits) if s < l - 1 else None
        return ShardedValue(parts, a.size, l - 1, a.scale / 2.0 ** divisor_bits)

    def sync(self):
        for sh in self.shards.values():
            sh.sync()


def execute_sharded(ev, program, enc_inputs, encode):
    """SEALPublic::execute (/root/reference/eva/seal/seal.cpp:104-122) over limb-sharded values:
    serial forwardPass of the compiled program, SEALExecutor's dispatch per node
    (/root/reference/eva/seal/seal_executor.h:279-404).  encode(values, scale_bits, level) -> [l][N]
    plaintext residues (the host encoder).  Returns {output name: ShardedValue | list}."""
    from eva_amd import Op
    vals = {}
    inputs = {name: t.index for name, t in program.inputs.items()}
    for name in enc_inputs.names():
        kind, size, limbs, scale, data = enc_inputs.get(name)
        t = inputs[name]
        if kind == "cipher":
            vals[t] = ev.upload_ct(data, scale)
        elif kind == "plain":
            vals[t] = ev.upload_pt(data, scale)
        else:
            vals[t] = list(data) * (program.vec_size // len(data))

    def is_ct(v): return isinstance(v, ShardedValue) and v.size > 0
    def is_pt(v): return isinstance(v, ShardedValue) and v.size == 0

    def rot(v, s, left):
        s %= len(v)
        return v[s:] + v[:s] if left else v[len(v) - s:] + v[:len(v) - s]

    for d in program._dump():
        t, op, a = d["id"], d["op"], d["operands"]
        if op == Op.Input:
            continue
        x = vals[a[0]] if a else None
        y = vals[a[1]] if len(a) > 1 else None
        if op == Op.Constant:
            c = d["constant"]
            vals[t] = list(c) * (program.vec_size // len(c))
        elif op == Op.Encode:
            vals[t] = ev.upload_pt(encode(x, d["encode_scale"], d["encode_level"]), 2.0 ** d["encode_scale"])
        elif op in (Op.Add, Op.Sub, Op.Mul) and isinstance(x, list) and isinstance(y, list):
            f = {Op.Add: lambda u, v: u + v, Op.Sub: lambda u, v: u - v, Op.Mul: lambda u, v: u * v}[op]
            vals[t] = [f(u, v) for u, v in zip(x, y)]
        elif op == Op.Add:
            if not is_ct(x):
                x, y = y, x
            vals[t] = ev.add(x, y) if is_ct(y) else ev.add_plain(x, y)
        elif op == Op.Sub:
            if not is_ct(x):
                raise RuntimeError("Unsupported operation encountered")
            vals[t] = ev.sub(x, y) if is_ct(y) else ev.sub_plain(x, y)
        elif op == Op.Mul:
            same = a[0] == a[1]
            if not is_ct(x):
                x, y = y, x
            vals[t] = (ev.square(x) if same else ev.multiply(x, y)) if is_ct(y) else ev.multiply_plain(x, y)
        elif op in (Op.RotateLeftConst, Op.RotateRightConst):
            if isinstance(x, list):
                vals[t] = rot(x, d["rotation"], op == Op.RotateLeftConst)
            else:
                vals[t] = ev.rotate(x, d["rotation"] if op == Op.RotateLeftConst else -d["rotation"])
        elif op == Op.Negate:
            vals[t] = [-u for u in x] if isinstance(x, list) else ev.negate(x)
        elif op in (Op.Relinearize, Op.ModSwitch, Op.Rescale) and isinstance(x, list):
            vals[t] = x
        elif op == Op.Relinearize:
            vals[t] = ev.relinearize(x)
        elif op == Op.ModSwitch:
            vals[t] = ev.mod_switch(x)
        elif op == Op.Rescale:
            vals[t] = ev.rescale(x, d["rescale_divisor"])
        elif op == Op.Output:
            vals[t] = x
        else:
            raise RuntimeError(f"Unhandled op {op}")
    return {name: vals[t.index] for name, t in program.outputs.items()}
