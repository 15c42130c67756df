"""Test-side CPU shard: the phases of tests/shard_harness.py's limb-sharded key switch and rescale
restated over the CPU oracle's primitives (per-prime NTT / INTT) and exact Python integers, so the
partition / exchange / reassembly logic of ShardedEvaluator can be checked — in one process and
across gloo ranks — without a GPU.  Follows SURVEY.md A.5 / A.6 limb by limb; the result of every
operation is compared with the UNSHARDED oracle (oracle/eva_oracle.c) in the tests.
Test infrastructure only."""
import numpy as np

from oracle import pyoracle as po


class HostBuffer:
    """exchange buffer of the CPU shard: the methods of eva_amd.backend.DeviceBuffer"""

    def __init__(self, words):
        self.words = words
        self.a = np.zeros(words, dtype=np.uint64)

    def copy_from(self, src, dst_off, src_off, words):
        self.a[dst_off:dst_off + words] = src.a[src_off:src_off + words]

    def gather_from(self, srcs, src_offs, dst_offs, words):
        for src, so, do in zip(srcs, src_offs, dst_offs):
            self.copy_from(src, do, so, words)

    def download(self, off=0, words=None):
        words = self.words - off if words is None else words
        return self.a[off:off + words].copy()

    def upload(self, data, off=0):
        d = np.asarray(data, dtype=np.uint64).reshape(-1)
        self.a[off:off + d.size] = d


class Local:
    """a shard-local value: data [size][n_local][N] (plaintext: [n_local][N])"""

    def __init__(self, data):
        self.data = data


def _mulmod(a, b, q):
    return np.array([(int(x) * int(y)) % q for x, y in zip(a, b)], dtype=np.uint64)


class OracleShard:
    def __init__(self, N, primes, shard, G):
        self.N, self.primes, self.k, self.shard, self.G = N, list(primes), len(primes), shard, G
        self.o = po.Oracle(N, primes)  # per-prime transforms under the GLOBAL prime indices
        self.keys = {}

    def prime(self, j):
        return self.shard + j * self.G

    # ---- values
    def upload_ct(self, data, scale): return Local(np.array(data, dtype=np.uint64))
    def upload_pt(self, data, scale): return Local(np.array(data, dtype=np.uint64))
    def download(self, h): return h.data
    def upload_key(self, kind, elt, key): self.keys[(kind, elt if kind else 0)] = np.array(key, dtype=np.uint64)
    def buffer(self, words): return HostBuffer(words)
    def sync(self): pass
    def close(self): pass

    # ---- per-limb operations
    def _limbwise(self, f, *arrs, out_polys=None):
        nl = arrs[0].shape[-2]
        cols = []
        for j in range(nl):
            q = self.primes[self.prime(j)]
            cols.append(f(q, *[a[..., j, :] for a in arrs]))
        return np.stack(cols, axis=-2)

    def add(self, a, b): return Local(self._bin(a.data, b.data, False))
    def sub(self, a, b): return Local(self._bin(a.data, b.data, True))

    def _bin(self, a, b, sub):
        sa, sb, nl = a.shape[0], b.shape[0], a.shape[1]
        out = np.zeros((max(sa, sb), nl, self.N), dtype=np.uint64)
        for j in range(nl):
            q = self.primes[self.prime(j)]
            for p in range(max(sa, sb)):
                x = a[p, j].astype(object) if p < sa else 0
                y = b[p, j].astype(object) if p < sb else 0
                out[p, j] = ((x - y) % q if sub else (x + y) % q).astype(np.uint64)
        return out

    def negate(self, a): return Local(self._limbwise(lambda q, x: ((q - x.astype(object)) % q).astype(np.uint64), a.data))
    def add_plain(self, a, p):
        out = a.data.copy()
        out[0] = self._limbwise(lambda q, x, y: ((x.astype(object) + y.astype(object)) % q).astype(np.uint64), a.data[0], p.data)
        return Local(out)
    def sub_plain(self, a, p):
        out = a.data.copy()
        out[0] = self._limbwise(lambda q, x, y: ((x.astype(object) - y.astype(object)) % q).astype(np.uint64), a.data[0], p.data)
        return Local(out)
    def multiply(self, a, b):
        A, B = a.data.astype(object), b.data.astype(object)
        nl = A.shape[1]
        out = np.zeros((3, nl, self.N), dtype=np.uint64)
        for j in range(nl):
            q = self.primes[self.prime(j)]
            out[0, j] = (A[0, j] * B[0, j] % q).astype(np.uint64)
            out[1, j] = ((A[0, j] * B[1, j] + A[1, j] * B[0, j]) % q).astype(np.uint64)
            out[2, j] = (A[1, j] * B[1, j] % q).astype(np.uint64)
        return Local(out)
    def square(self, a): return self.multiply(a, a)
    def multiply_plain(self, a, p):
        return Local(np.stack([self._limbwise(lambda q, x, y: (x.astype(object) * y.astype(object) % q).astype(np.uint64), a.data[i], p.data)
                               for i in range(a.data.shape[0])]))
    def drop_last_limb(self, a): return Local(a.data[..., :-1, :].copy())
    def galois_perm(self, a, elt):
        tab = po.galois_table(self.N, elt)
        return Local(a.data[..., tab])

    # ---- key switch phases
    def ks_digits(self, a, poly, l, digits, rows):
        for j in range(a.data.shape[1]):
            t = self.o.intt(self.prime(j), a.data[poly, j].copy())
            digits.upload(t, (self.shard * rows + j) * self.N)

    def ks_products(self, a, poly, l, digits, rows, kind, elt, prod, r):
        N, G, s, k = self.N, self.G, self.shard, self.k
        key = self.keys[(kind, elt if kind else 0)]  # [digit][2][k][N]
        outs = list(range(s, l, G)) + ([l] if l % G == s else [])
        ni = len(outs)
        D = digits.download()
        t = {J: D[((J % G) * rows + J // G) * N:((J % G) * rows + J // G + 1) * N] for J in range(l)}
        P = np.zeros((2, max(ni, 1), N), dtype=np.uint64)
        for iy, I in enumerate(outs):
            kap = k - 1 if I == l else I
            q = self.primes[kap]
            acc = [np.zeros(N, dtype=object), np.zeros(N, dtype=object)]
            for J in range(l):
                if I == J:
                    op = a.data[poly, iy].astype(object)
                else:
                    op = self.o.ntt(kap, (t[J].astype(object) % q).astype(np.uint64)).astype(object)
                for K in range(2):
                    acc[K] = acc[K] + op * key[J, K, kap].astype(object)
            for K in range(2):
                P[K, iy] = (acc[K] % q).astype(np.uint64)
        prod.upload(P.reshape(-1))
        prod.shape = (2, max(ni, 1))
        if l % G == s:
            qP = self.primes[k - 1]
            R = np.stack([((self.o.intt(k - 1, P[K, ni - 1].copy()).astype(object) + (qP >> 1)) % qP).astype(np.uint64) for K in range(2)])
            r.upload(R.reshape(-1))

    def ks_finish(self, l, prod, r, add, add_polys, scale):
        N, G, s, k = self.N, self.G, self.shard, self.k
        nl = len(range(s, l, G))
        ni = prod.shape[1]
        P = prod.download(0, 2 * ni * N).reshape(2, ni, N)
        R = r.download(0, 2 * N).reshape(2, N).astype(object)
        qP = self.primes[k - 1]
        out = np.zeros((2, nl, N), dtype=np.uint64)
        for j in range(nl):
            gi = self.prime(j)
            q = self.primes[gi]
            pinv = pow(qP % q, q - 2, q)
            for K in range(2):
                u = ((R[K] % q) - ((qP >> 1) % q)) % q
                U = self.o.ntt(gi, u.astype(np.uint64)).astype(object)
                v = (P[K, j].astype(object) - U) * pinv % q
                if add is not None and K < add_polys:
                    v = (v + add.data[K, j].astype(object)) % q
                out[K, j] = v.astype(np.uint64)
        return Local(out)

    # ---- rescale phases
    def rescale_last(self, a, l, r):
        ql = self.primes[l - 1]
        size, nl = a.data.shape[0], a.data.shape[1]
        R = np.stack([((self.o.intt(l - 1, a.data[p, nl - 1].copy()).astype(object) + (ql >> 1)) % ql).astype(np.uint64) for p in range(size)])
        r.upload(R.reshape(-1))

    def rescale_finish(self, a, l, r, bits):
        N, G, s = self.N, self.G, self.shard
        size = a.data.shape[0]
        nn = len(range(s, l - 1, G))
        ql = self.primes[l - 1]
        R = r.download(0, size * N).reshape(size, N).astype(object)
        out = np.zeros((size, nn, N), dtype=np.uint64)
        for j in range(nn):
            gi = self.prime(j)
            q = self.primes[gi]
            inv = pow(ql % q, q - 2, q)
            for p in range(size):
                u = ((R[p] % q) - ((ql >> 1) % q)) % q
                U = self.o.ntt(gi, u.astype(np.uint64)).astype(object)
                out[p, j] = ((a.data[p, j].astype(object) - U) * inv % q).astype(np.uint64)
        return Local(out)
