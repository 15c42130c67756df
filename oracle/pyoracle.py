"""ctypes binding of oracle/libeva_oracle.so.

TEST INFRASTRUCTURE ONLY (see oracle/eva_oracle.h): imported by tests/, by
__graft_entry__.smoke() and by bench.py's cpu_baseline leg — never by eva_amd/.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libeva_oracle.so")


def build(force=False):
    srcs = [os.path.join(_HERE, f) for f in ("eva_oracle.c", "eva_oracle_dag.c", "eva_oracle.h")]
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < max(os.path.getmtime(s) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s"])


class DagOp(C.Structure):  # evo_dag_op
    _fields_ = [("op", C.c_uint32), ("dst", C.c_uint32), ("src0", C.c_uint32), ("src1", C.c_uint32), ("imm", C.c_int32)]


class DagVal(C.Structure):  # evo_dag_val
    _fields_ = [("kind", C.c_uint32), ("size", C.c_uint32), ("limbs", C.c_uint32), ("alias", C.c_uint32),
                ("data", C.POINTER(C.c_uint64))]


def _load():
    if not os.path.exists(_LIB):
        build()
    lib = C.CDLL(_LIB)
    u64p = C.POINTER(C.c_uint64)
    u32p = C.POINTER(C.c_uint32)
    dblp = C.POINTER(C.c_double)
    vp = C.c_void_p
    sig = {
        "evo_is_prime": (C.c_int, [C.c_uint64]),
        "evo_coeff_modulus_create": (C.c_int, [C.c_uint32, C.POINTER(C.c_int), C.c_uint32, u64p]),
        "evo_minimal_primitive_root": (C.c_uint64, [C.c_uint32, C.c_uint64]),
        "evo_mulmod": (C.c_uint64, [C.c_uint64] * 3),
        "evo_powmod": (C.c_uint64, [C.c_uint64] * 3),
        "evo_invmod": (C.c_uint64, [C.c_uint64] * 2),
        "evo_ctx_create": (vp, [C.c_uint32, C.c_uint32, u64p]),
        "evo_ctx_destroy": (None, [vp]),
        "evo_ctx_psi": (C.c_uint64, [vp, C.c_uint32]),
        "evo_ctx_root_powers": (u64p, [vp, C.c_uint32]),
        "evo_ctx_inv_root_powers": (u64p, [vp, C.c_uint32]),
        "evo_ntt": (None, [vp, C.c_uint32, u64p]),
        "evo_intt": (None, [vp, C.c_uint32, u64p]),
        "evo_add": (None, [vp, C.c_uint32, u64p, C.c_uint32, u64p, C.c_uint32, u64p]),
        "evo_sub": (None, [vp, C.c_uint32, u64p, C.c_uint32, u64p, C.c_uint32, u64p]),
        "evo_add_plain": (None, [vp, C.c_uint32, u64p, C.c_uint32, u64p, u64p]),
        "evo_sub_plain": (None, [vp, C.c_uint32, u64p, C.c_uint32, u64p, u64p]),
        "evo_negate": (None, [vp, C.c_uint32, u64p, C.c_uint32, u64p]),
        "evo_multiply": (None, [vp, C.c_uint32, u64p, u64p, u64p]),
        "evo_square": (None, [vp, C.c_uint32, u64p, u64p]),
        "evo_multiply_plain": (None, [vp, C.c_uint32, u64p, C.c_uint32, u64p, u64p]),
        "evo_rescale": (None, [vp, C.c_uint32, u64p, C.c_uint32, u64p]),
        "evo_mod_switch": (None, [vp, C.c_uint32, u64p, C.c_uint32, u64p]),
        "evo_switch_key": (None, [vp, C.c_uint32, u64p, u64p, u64p]),
        "evo_relinearize": (None, [vp, C.c_uint32, u64p, u64p, u64p]),
        "evo_galois_elt_from_step": (C.c_uint32, [C.c_uint32, C.c_int32]),
        "evo_galois_table": (None, [C.c_uint32, C.c_uint32, u32p]),
        "evo_rotate": (None, [vp, C.c_uint32, u64p, C.c_int32, u64p, u64p]),
        "evo_encode": (C.c_int, [vp, C.c_uint32, dblp, C.c_double, u64p]),
        "evo_encode_coeffs": (None, [C.c_uint32, dblp, C.c_double, dblp]),
        "evo_decrypt": (None, [vp, C.c_uint32, C.c_uint32, u64p, u64p, u64p]),
        "evo_decode": (C.c_int, [vp, C.c_uint32, u64p, C.c_double, dblp]),
        "evo_op_triple": (None, [vp, C.c_uint32, u64p, u64p, u64p, u64p]),
        "evo_dag_walk": (C.c_int, [vp, C.POINTER(DagOp), C.c_uint32, C.POINTER(DagVal), C.c_uint32, u64p, u32p,
                                   C.POINTER(u64p), C.c_uint32, C.c_int]),
        "evo_dag_free": (None, [u64p]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()
_u64p = C.POINTER(C.c_uint64)


def _p(a):
    assert a.dtype == np.uint64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_u64p)


def coeff_modulus_create(N, bit_sizes):
    bits = (C.c_int * len(bit_sizes))(*bit_sizes)
    out = np.zeros(len(bit_sizes), dtype=np.uint64)
    rc = lib.evo_coeff_modulus_create(N, bits, len(bit_sizes), _p(out))
    if rc != 0:
        raise ValueError("failed to find enough qualifying primes")
    return [int(x) for x in out]


def galois_elt_from_step(N, steps):
    return int(lib.evo_galois_elt_from_step(N, steps))


def galois_table(N, elt):
    t = np.zeros(N, dtype=np.uint32)
    lib.evo_galois_table(N, elt, t.ctypes.data_as(C.POINTER(C.c_uint32)))
    return t


class Oracle:
    """One CKKS context: N, key-level primes (special prime last)."""

    def __init__(self, N, primes):
        self.N = N
        self.primes = [int(p) for p in primes]
        self.k = len(primes)
        arr = np.array(self.primes, dtype=np.uint64)
        self._c = lib.evo_ctx_create(N, self.k, _p(arr))
        if not self._c:
            raise ValueError("bad primes for N")

    def __del__(self):
        if getattr(self, "_c", None):
            lib.evo_ctx_destroy(self._c)
            self._c = None

    def psi(self, i):
        return int(lib.evo_ctx_psi(self._c, i))

    def root_powers(self, i):
        return np.ctypeslib.as_array(lib.evo_ctx_root_powers(self._c, i), shape=(self.N,)).copy()

    def inv_root_powers(self, i):
        return np.ctypeslib.as_array(lib.evo_ctx_inv_root_powers(self._c, i), shape=(self.N,)).copy()

    def ntt(self, i, x):
        y = np.ascontiguousarray(x, dtype=np.uint64).copy()
        lib.evo_ntt(self._c, i, _p(y))
        return y

    def intt(self, i, x):
        y = np.ascontiguousarray(x, dtype=np.uint64).copy()
        lib.evo_intt(self._c, i, _p(y))
        return y

    # ciphertexts are numpy uint64 arrays of shape [size, l, N]
    def _bin(self, fn, a, b):
        l = a.shape[1]
        out = np.empty((max(a.shape[0], b.shape[0]), l, self.N), dtype=np.uint64)
        fn(self._c, l, _p(a), a.shape[0], _p(b), b.shape[0], _p(out))
        return out

    def add(self, a, b):
        return self._bin(lib.evo_add, a, b)

    def sub(self, a, b):
        return self._bin(lib.evo_sub, a, b)

    def _plain(self, fn, a, pt):
        out = np.empty_like(a)
        fn(self._c, a.shape[1], _p(a), a.shape[0], _p(pt), _p(out))
        return out

    def add_plain(self, a, pt):
        return self._plain(lib.evo_add_plain, a, pt)

    def sub_plain(self, a, pt):
        return self._plain(lib.evo_sub_plain, a, pt)

    def multiply_plain(self, a, pt):
        return self._plain(lib.evo_multiply_plain, a, pt)

    def negate(self, a):
        out = np.empty_like(a)
        lib.evo_negate(self._c, a.shape[1], _p(a), a.shape[0], _p(out))
        return out

    def multiply(self, a, b):
        out = np.empty((3, a.shape[1], self.N), dtype=np.uint64)
        lib.evo_multiply(self._c, a.shape[1], _p(a), _p(b), _p(out))
        return out

    def square(self, a):
        out = np.empty((3, a.shape[1], self.N), dtype=np.uint64)
        lib.evo_square(self._c, a.shape[1], _p(a), _p(out))
        return out

    def rescale(self, a):
        out = np.empty((a.shape[0], a.shape[1] - 1, self.N), dtype=np.uint64)
        lib.evo_rescale(self._c, a.shape[1], _p(a), a.shape[0], _p(out))
        return out

    def mod_switch(self, a):
        out = np.empty((a.shape[0], a.shape[1] - 1, self.N), dtype=np.uint64)
        lib.evo_mod_switch(self._c, a.shape[1], _p(a), a.shape[0], _p(out))
        return out

    def switch_key(self, ct2, target, key):
        out = ct2.copy()
        lib.evo_switch_key(self._c, ct2.shape[1], _p(out), _p(target), _p(key))
        return out

    def relinearize(self, a3, key):
        out = np.empty((2, a3.shape[1], self.N), dtype=np.uint64)
        lib.evo_relinearize(self._c, a3.shape[1], _p(a3), _p(key), _p(out))
        return out

    def rotate(self, a2, steps, key):
        out = np.empty_like(a2)
        kp = _p(key) if key is not None else None
        lib.evo_rotate(self._c, a2.shape[1], _p(a2), steps, kp, _p(out))
        return out

    def encode(self, l, values, scale):
        v = np.ascontiguousarray(values, dtype=np.float64)
        assert v.shape[0] == self.N // 2
        pt = np.empty((l, self.N), dtype=np.uint64)
        rc = lib.evo_encode(self._c, l, v.ctypes.data_as(C.POINTER(C.c_double)), float(scale), _p(pt))
        if rc != 0:
            raise ValueError("encoded values are too large")
        return pt

    def decrypt(self, ct, sk_ntt):
        """Decryptor::decrypt: ct [size][l][N], sk_ntt [k][N] -> plaintext [l][N] (NTT form)"""
        ct = np.ascontiguousarray(ct, dtype=np.uint64)
        sk = np.ascontiguousarray(sk_ntt, dtype=np.uint64)
        pt = np.empty((ct.shape[1], self.N), dtype=np.uint64)
        lib.evo_decrypt(self._c, ct.shape[1], ct.shape[0], _p(ct), _p(sk), _p(pt))
        return pt

    def decode(self, pt, scale):
        """CKKSEncoder::decode: plaintext [l][N] (NTT form) -> the N/2 slot values (SEAL 3.6's FP64 order)"""
        pt = np.ascontiguousarray(pt, dtype=np.uint64)
        out = np.empty(self.N // 2, dtype=np.float64)
        rc = lib.evo_decode(self._c, pt.shape[0], _p(pt), float(scale), out.ctypes.data_as(C.POINTER(C.c_double)))
        if rc != 0:
            raise ValueError("scale out of bounds")
        return out

    def dag_walk(self, ops, values, n_vals, relin_key, galois_keys, threads=1):
        """Walks a flat op list [(op, dst, src0, src1, imm)] (the reference's op codes) over the
        evaluator, in C: threads <= 1 the serial forwardPass, > 1 the dependency-counting traversal
        on that many pthreads.  values: {slot: ("ct", array [size][l][N]) | ("pt", array [l][N])}.
        Returns ({slot: array} for every slot the walk left filled, seconds spent inside the walk)."""
        import time
        arr = (DagOp * len(ops))(*[DagOp(*[int(x) for x in o[:5]]) for o in ops])
        vals = (DagVal * n_vals)()
        keep = []
        for t, (kind, a) in values.items():
            a = np.ascontiguousarray(a, dtype=np.uint64)
            keep.append(a)
            vals[t].kind = 1 if kind == "ct" else 2
            vals[t].size = a.shape[0] if kind == "ct" else 1
            vals[t].limbs = a.shape[-2]
            vals[t].data = _p(a)
        elts = sorted(galois_keys)
        gk = [np.ascontiguousarray(galois_keys[e], dtype=np.uint64) for e in elts]
        elt_arr = (C.c_uint32 * max(1, len(elts)))(*elts)
        key_arr = (_u64p * max(1, len(elts)))(*[_p(k) for k in gk])
        rk = _p(relin_key) if relin_key is not None else None
        t0 = time.perf_counter()
        rc = lib.evo_dag_walk(self._c, arr, len(ops), vals, n_vals, rk, elt_arr, key_arr, len(elts), int(threads))
        dt = time.perf_counter() - t0
        if rc != 0:
            raise RuntimeError(f"evo_dag_walk failed ({rc})")
        out = {}
        for t in range(n_vals):
            v = vals[t]
            if v.kind == 1 and t not in values:
                n = v.size * v.limbs * self.N
                out[t] = np.ctypeslib.as_array(v.data, shape=(n,)).reshape(v.size, v.limbs, self.N).copy()
        for t in range(n_vals):
            if vals[t].kind == 1 and t not in values and not vals[t].alias:
                lib.evo_dag_free(vals[t].data)
        return out, dt

    def op_triple(self, a2, b2, key):
        l = a2.shape[1]
        out = np.empty((2, l - 1, self.N), dtype=np.uint64)
        lib.evo_op_triple(self._c, l, _p(a2), _p(b2), _p(key), _p(out))
        return out


def encode_coeffs(N, values, scale):
    v = np.ascontiguousarray(values, dtype=np.float64)
    out = np.empty(N, dtype=np.float64)
    dp = C.POINTER(C.c_double)
    lib.evo_encode_coeffs(N, v.ctypes.data_as(dp), float(scale), out.ctypes.data_as(dp))
    return out
