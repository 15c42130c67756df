"""ctypes binding of libeva_hip.so (include/eva_hip.h) — the MI355X CKKS evaluation backend.

This is the drop-in for the seal::Evaluator calls of EVA's SEALExecutor
(/root/reference/eva/seal/seal_executor.h:114-243).  There is no CPU fallback: if the HIP
library is missing or no device is present, construction fails loudly.
"""
import ctypes as C
import os

import numpy as np

from . import _hipruntime  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libeva_hip.so")

KEY_RELIN = 0
KEY_GALOIS = 1


class EvaHipError(RuntimeError):
    pass


_u64p = C.POINTER(C.c_uint64)
_vp = C.c_void_p
_vpp = C.POINTER(C.c_void_p)

# name -> (argtypes); every function returns int status unless listed in _VOID / _OTHER
_SIGS = {
    "evah_device_count": [C.POINTER(C.c_int)],
    "evah_ctx_create": [C.c_uint32, C.c_uint32, _u64p, C.c_int, _vpp],
    "evah_ctx_fork": [_vp, _vpp],
    "evah_ctx_set_stream": [_vp, _vp],
    "evah_ctx_sync": [_vp],
    "evah_ctx_mem_info": [_vp, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)],
    "evah_key_upload": [_vp, C.c_int, C.c_uint32, C.c_uint32, _u64p],
    "evah_galois_elt_from_step": [_vp, C.c_int32, C.POINTER(C.c_uint32)],
    "evah_ct_upload": [_vp, C.c_uint32, C.c_uint32, C.c_double, _u64p, _vpp],
    "evah_ct_write": [_vp, _vp, _u64p],
    "evah_ct_copy": [_vp, _vp, _vpp],
    "evah_ct_assign": [_vp, _vp, _vp],
    "evah_ctx_wait": [_vp, _vp],
    "evah_ctx_transfer_stats": [_vp, _u64p],
    "evah_ctx_key_bytes": [_vp, _u64p],
    "evah_ctx_key_bytes_detail": [_vp, _u64p],
    "evah_pt_copy": [_vp, _vp, _vpp],
    "evah_pt_write": [_vp, _vp, _u64p],
    "evah_capture_begin": [_vp, _vpp, C.c_uint32],
    "evah_capture_end": [_vp, _vpp, C.c_uint32, _vpp],
    "evah_graph_launch": [_vp, _vp],
    "evah_ct_info": [_vp, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_double)],
    "evah_ct_download": [_vp, _vp, _u64p],
    "evah_pt_upload": [_vp, C.c_uint32, C.c_double, _u64p, _vpp],
    "evah_pt_upload_coeff": [_vp, C.c_uint32, C.c_double, _u64p, _vpp],
    "evah_pt_uniform": [_vp, C.c_uint32, C.c_double, _u64p, _vpp],
    "evah_pt_info": [_vp, C.POINTER(C.c_uint32), C.POINTER(C.c_double)],
    "evah_pt_download": [_vp, _vp, _u64p],
    "evah_add": [_vp, _vp, _vp, _vpp],
    "evah_sub": [_vp, _vp, _vp, _vpp],
    "evah_add_plain": [_vp, _vp, _vp, _vpp],
    "evah_sub_plain": [_vp, _vp, _vp, _vpp],
    "evah_negate": [_vp, _vp, _vpp],
    "evah_multiply": [_vp, _vp, _vp, _vpp],
    "evah_square": [_vp, _vp, _vpp],
    "evah_multiply_plain": [_vp, _vp, _vp, _vpp],
    "evah_relinearize": [_vp, _vp, _vpp],
    "evah_relinearize_rescale": [_vp, _vp, C.c_uint32, _vpp],
    "evah_relinearize_rescale_many": [_vp, _vpp, C.c_uint32, C.c_uint32, _vpp],
    "evah_multiply_many": [_vp, _vpp, _vpp, C.c_uint32, _vpp],
    "evah_multiply_relinearize_rescale": [_vp, _vp, _vp, C.c_uint32, _vpp],
    "evah_multiply_relinearize_rescale_many": [_vp, _vpp, _vpp, C.c_uint32, C.c_uint32, _vpp],
    "evah_multiply_rescale_relinearize": [_vp, _vp, _vp, C.c_uint32, _vpp],
    "evah_multiply_rescale_relinearize_many": [_vp, _vpp, _vpp, C.c_uint32, C.c_uint32, _vpp],
    "evah_ctx_busy": [_vp, C.POINTER(C.c_int)],
    "evah_execute": [_vp, _vp, C.c_uint32, _vp, C.c_uint32],
    "evah_elementwise_program": [_vp, _vp, C.c_uint32, _vp, C.c_uint32, C.POINTER(C.c_uint32), C.c_uint32, _vpp],
    "evah_weighted_sum": [_vp, _vpp, _vpp, C.c_uint32, _vpp],
    "evah_rotate_pairs": [_vp, _vpp, C.POINTER(C.c_int32), C.c_uint32, _vpp],
    "evah_rotate_weighted_sums": [_vp, _vpp, C.POINTER(C.c_int32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.c_uint32, _vpp, _vpp],
    "evah_multiply_plain_many": [_vp, _vpp, _vpp, C.c_uint32, _vpp],
    "evah_rescale_many": [_vp, _vpp, C.c_uint32, C.c_uint32, _vpp],
    "evah_relinearize_many": [_vp, _vpp, C.c_uint32, _vpp],
    "evah_rescale_relinearize": [_vp, _vp, C.c_uint32, _vpp],
    "evah_rescale_relinearize_many": [_vp, _vpp, C.c_uint32, C.c_uint32, _vpp],
    "evah_pt_encode": [_vp, C.POINTER(C.c_double), C.c_uint32, C.c_uint32, C.c_double, _vpp],
    "evah_ct_upload_batch": [_vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_double, _u64p, _vpp],
    "evah_ct_upload_instances": [_vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_double, C.POINTER(_u64p), _vpp],
    "evah_ct_download_instances": [_vp, _vp, C.POINTER(_u64p)],
    "evah_ct_upload_instances_async": [_vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_double, C.POINTER(_u64p), _vpp],
    "evah_ct_download_instances_async": [_vp, _vp, C.POINTER(_u64p)],
    "evah_ct_batch": [_vp, C.POINTER(C.c_uint32)],
    "evah_ct_stack": [_vp, _vpp, C.c_uint32, _vpp],
    "evah_ct_unstack": [_vp, _vp, C.c_uint32, _vpp],
    "evah_rotate": [_vp, _vp, C.c_int32, _vpp],
    "evah_rotate_many": [_vp, _vp, C.POINTER(C.c_int32), C.c_uint32, _vpp],
    "evah_rescale": [_vp, _vp, C.c_uint32, _vpp],
    "evah_mod_switch": [_vp, _vp, _vpp],
    "evah_test_ntt": [_vp, C.c_uint32, C.c_int, _u64p],
    "evah_profile_enable": [_vp, C.c_int],
    "evah_profile_reset": [_vp],
    "evah_profile_get": [_vp, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_double)],
    "evah_timer_start": [_vp],
    "evah_timer_stop": [_vp, C.POINTER(C.c_float)],
    "evah_client_key_upload": [_vp, C.c_int, _u64p],
    "evah_encrypt": [_vp, _vp, C.POINTER(C.c_int8), _vpp],
    "evah_decrypt_decode": [_vp, _vp, C.c_uint32, C.POINTER(C.c_double)],
    # limb-sharded execution
    "evah_ctx_set_shard": [_vp, C.c_uint32, C.c_uint32],
    "evah_ctx_shard_info": [_vp, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)],
    "evah_buf_alloc": [_vp, C.c_size_t, _vpp],
    "evah_buf_copy": [_vp, _vp, C.c_size_t, _vp, C.c_size_t, C.c_size_t],
    "evah_buf_gather": [_vp, _vp, C.c_uint32, _vpp, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.c_size_t],
    "evah_ctx_enable_peer": [_vp, _vp],
    "evah_buf_download": [_vp, _vp, C.c_size_t, C.c_size_t, _u64p],
    "evah_buf_upload": [_vp, _vp, C.c_size_t, C.c_size_t, _u64p],
    "evah_shard_galois_perm": [_vp, _vp, C.c_uint32, _vpp],
    "evah_shard_ks_digits": [_vp, _vp, C.c_uint32, C.c_uint32, _vp, C.c_uint32],
    "evah_shard_ks_products": [_vp, _vp, C.c_uint32, C.c_uint32, _vp, C.c_uint32, C.c_int, C.c_uint32, _vp, _vp],
    "evah_shard_ks_finish": [_vp, C.c_uint32, _vp, _vp, _vp, C.c_uint32, C.c_double, _vpp],
    "evah_shard_rescale_last": [_vp, _vp, C.c_uint32, _vp],
    "evah_shard_rescale_finish": [_vp, _vp, C.c_uint32, _vp, C.c_uint32, _vpp],
}
_VOID = {
    "evah_ctx_destroy": [_vp],
    "evah_ct_free": [_vp, _vp],
    "evah_pt_free": [_vp, _vp],
    "evah_graph_free": [_vp],
    "evah_buf_free": [_vp, _vp],
}
EXPORTED_SYMBOLS = sorted(list(_SIGS) + list(_VOID) + ["evah_host_alloc", "evah_host_free", "evah_buf_ptr", "evah_buf_words"] + [
    "evah_last_error", "evah_abi_version", "evah_profile_classes", "evah_profile_class_name"])



class EvahVal(C.Structure):
    """include/eva_hip.h evah_val"""
    _fields_ = [("kind", C.c_uint32), ("h", C.c_void_p)]


class EvahOp(C.Structure):
    """include/eva_hip.h evah_op"""
    _fields_ = [("op", C.c_uint32), ("dst", C.c_uint32), ("src0", C.c_uint32), ("src1", C.c_uint32),
                ("imm", C.c_int32), ("flags", C.c_uint32)]


class EvahEwOp(C.Structure):
    """include/eva_hip.h evah_ew_op"""
    _fields_ = [("op", C.c_uint32), ("a", C.c_uint32), ("b", C.c_uint32)]


VAL_NONE, VAL_CT, VAL_PT = 0, 1, 2
OPF_FREE_SRC0, OPF_FREE_SRC1 = 1, 2

_lib = None


def load():
    """Load libeva_hip.so (once).  Raises EvaHipError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise EvaHipError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(the MI355X backend has no CPU fallback)")
    lib = C.CDLL(LIB_PATH)
    for name, args in _SIGS.items():
        fn = getattr(lib, name)
        fn.restype = C.c_int
        fn.argtypes = args
    for name, args in _VOID.items():
        fn = getattr(lib, name)
        fn.restype = None
        fn.argtypes = args
    lib.evah_last_error.restype = C.c_char_p
    lib.evah_last_error.argtypes = []
    lib.evah_abi_version.restype = C.c_int
    lib.evah_abi_version.argtypes = []
    lib.evah_profile_classes.restype = C.c_int
    lib.evah_profile_classes.argtypes = []
    lib.evah_profile_class_name.restype = C.c_char_p
    lib.evah_profile_class_name.argtypes = [C.c_int]
    lib.evah_buf_ptr.restype = C.c_void_p
    lib.evah_buf_ptr.argtypes = [_vp]
    lib.evah_buf_words.restype = C.c_size_t
    lib.evah_buf_words.argtypes = [_vp]
    _lib = lib
    return lib


def _chk(rc):
    if rc != 0:
        raise EvaHipError(_lib.evah_last_error().decode())


def _p(a):
    assert a.dtype == np.uint64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_u64p)


def device_count():
    lib = load()
    n = C.c_int(0)
    _chk(lib.evah_device_count(C.byref(n)))
    return n.value


class Ciphertext:
    """Device-resident ciphertext handle (evah_ct)."""

    __slots__ = ("ctx", "h")

    def __init__(self, ctx, h):
        self.ctx, self.h = ctx, h

    def info(self):
        s, l, sc = C.c_uint32(), C.c_uint32(), C.c_double()
        _chk(_lib.evah_ct_info(self.h, C.byref(s), C.byref(l), C.byref(sc)))
        return s.value, l.value, sc.value

    size = property(lambda self: self.info()[0])
    limbs = property(lambda self: self.info()[1])
    scale = property(lambda self: self.info()[2])

    @property
    def batch(self):
        b = C.c_uint32()
        _chk(_lib.evah_ct_batch(self.h, C.byref(b)))
        return b.value

    def download(self):
        """[size][limbs][N]; a batched handle downloads as [batch][size][limbs][N]"""
        s, l, _ = self.info()
        b = self.batch
        out = np.empty((b, s, l, self.ctx.N), dtype=np.uint64)
        _chk(_lib.evah_ct_download(self.ctx.h, self.h, _p(out)))
        return out if b > 1 else out[0]

    def write(self, data):
        """overwrite the device words of this handle (same shape): refill of a captured graph's input"""
        data = np.ascontiguousarray(data, dtype=np.uint64)
        _chk(_lib.evah_ct_write(self.ctx.h, self.h, _p(data)))

    def assign(self, src, queue=None):
        """refill this handle from another one of the same shape, device to device (evah_ct_assign)"""
        _chk(_lib.evah_ct_assign((queue or self.ctx).h, self.h, src.h))

    def unstack(self, b):
        h = C.c_void_p()
        _chk(_lib.evah_ct_unstack(self.ctx.h, self.h, int(b), C.byref(h)))
        return Ciphertext(self.ctx, h)

    def free(self):
        if self.h:
            _lib.evah_ct_free(self.ctx.h, self.h)
            self.h = None

    def __del__(self):
        try:
            if self.h and self.ctx.h:
                self.free()
        except Exception:
            pass


class Plaintext:
    """Device-resident plaintext handle (evah_pt)."""

    __slots__ = ("ctx", "h")

    def __init__(self, ctx, h):
        self.ctx, self.h = ctx, h

    def info(self):
        l, sc = C.c_uint32(), C.c_double()
        _chk(_lib.evah_pt_info(self.h, C.byref(l), C.byref(sc)))
        return l.value, sc.value

    limbs = property(lambda self: self.info()[0])
    scale = property(lambda self: self.info()[1])

    def download(self):
        l, _ = self.info()
        out = np.empty((l, self.ctx.N), dtype=np.uint64)
        _chk(_lib.evah_pt_download(self.ctx.h, self.h, _p(out)))
        return out

    def write(self, data):
        """replace the words of this handle (evah_pt_write; [limbs][N] NTT-form residues)"""
        data = np.ascontiguousarray(data, dtype=np.uint64)
        assert data.shape == (self.info()[0], self.ctx.N)
        _chk(_lib.evah_pt_write(self.ctx.h, self.h, _p(data)))

    def free(self):
        if self.h:
            _lib.evah_pt_free(self.ctx.h, self.h)
            self.h = None

    def __del__(self):
        try:
            if self.h and self.ctx.h:
                self.free()
        except Exception:
            pass


class DeviceBuffer:
    """Device memory for the exchange steps of limb-sharded execution (evah_buf).  Exposes
    __cuda_array_interface__, so torch.as_tensor(buf, device='cuda') views it without a copy (RCCL
    collectives then run directly on it)."""

    def __init__(self, ctx, words):
        self.ctx, self.words = ctx, int(words)
        h = C.c_void_p()
        _chk(_lib.evah_buf_alloc(ctx.h, self.words, C.byref(h)))
        self.h = h

    @property
    def ptr(self):
        return _lib.evah_buf_ptr(self.h)

    @property
    def __cuda_array_interface__(self):
        return {"shape": (self.words,), "typestr": "<i8", "data": (self.ptr, False), "version": 2}

    def copy_from(self, src, dst_off, src_off, words):
        """device (or peer) copy on this buffer's queue, ordered after the producer of src"""
        _chk(_lib.evah_buf_copy(self.ctx.h, self.h, int(dst_off), src.h, int(src_off), int(words)))

    def gather_from(self, srcs, src_offs, dst_offs, words):
        """ONE launch on this buffer's queue that pulls a chunk of `words` words from each of srcs (peer reads across
        devices): srcs[j][src_offs[j]..) -> self[dst_offs[j]..)"""
        n = len(srcs)
        hs = (C.c_void_p * n)(*[s.h for s in srcs])
        so = (C.c_size_t * n)(*[int(x) for x in src_offs])
        do = (C.c_size_t * n)(*[int(x) for x in dst_offs])
        _chk(_lib.evah_buf_gather(self.ctx.h, self.h, n, hs, so, do, int(words)))

    def download(self, off=0, words=None):
        words = self.words - off if words is None else words
        out = np.empty(words, dtype=np.uint64)
        _chk(_lib.evah_buf_download(self.ctx.h, self.h, int(off), int(words), _p(out)))
        return out

    def upload(self, data, off=0):
        data = np.ascontiguousarray(data, dtype=np.uint64).reshape(-1)
        _chk(_lib.evah_buf_upload(self.ctx.h, self.h, int(off), data.size, _p(data)))

    def free(self):
        if self.h and self.ctx.h:
            _lib.evah_buf_free(self.ctx.h, self.h)
        self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Context:
    """evah_ctx: N, key-level prime chain (special prime last), tables and keys on one GPU."""

    def __init__(self, N, primes, device=0):
        load()
        self.N = int(N)
        self.primes = [int(p) for p in primes]
        self.k = len(self.primes)
        arr = np.array(self.primes, dtype=np.uint64)
        h = C.c_void_p()
        self.h = None
        _chk(_lib.evah_ctx_create(self.N, self.k, _p(arr), int(device), C.byref(h)))
        self.h = h

    def fork(self):
        """Another issue queue (own stream + pool) sharing this context's tables and keys."""
        child = Context.__new__(Context)
        child.N, child.primes, child.k, child.h = self.N, self.primes, self.k, None
        h = C.c_void_p()
        _chk(_lib.evah_ctx_fork(self.h, C.byref(h)))
        child.h = h
        return child

    def close(self):
        if self.h:
            _lib.evah_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- limb-sharded execution (include/eva_hip.h): this context as shard `shard` of `n_shards`
    def set_shard(self, shard, n_shards):
        _chk(_lib.evah_ctx_set_shard(self.h, int(shard), int(n_shards)))

    def buffer(self, words):
        return DeviceBuffer(self, words)

    def shard_galois_perm(self, a, elt):
        return self._ct1(_lib.evah_shard_galois_perm, a, C.c_uint32(int(elt)))

    def shard_ks_digits(self, a, poly, l, digits, rows):
        _chk(_lib.evah_shard_ks_digits(self.h, a.h, int(poly), int(l), digits.h, int(rows)))

    def shard_ks_products(self, a, poly, l, digits, rows, key_kind, elt, prod, r):
        _chk(_lib.evah_shard_ks_products(self.h, a.h if a is not None else None, int(poly), int(l), digits.h, int(rows),
                                         int(key_kind), int(elt), prod.h, r.h))

    def shard_ks_finish(self, l, prod, r, add, add_polys, scale):
        h = C.c_void_p()
        _chk(_lib.evah_shard_ks_finish(self.h, int(l), prod.h, r.h, add.h if add is not None else None, int(add_polys),
                                       float(scale), C.byref(h)))
        return Ciphertext(self, h)

    def shard_rescale_last(self, a, l, r):
        _chk(_lib.evah_shard_rescale_last(self.h, a.h, int(l), r.h))

    def shard_rescale_finish(self, a, l, r, divisor_bits):
        h = C.c_void_p()
        _chk(_lib.evah_shard_rescale_finish(self.h, a.h, int(l), r.h, int(divisor_bits), C.byref(h)))
        return Ciphertext(self, h)

    # ---- client side on the device (encrypt / decrypt + decode)
    def upload_public_key(self, pk):
        pk = np.ascontiguousarray(pk, dtype=np.uint64)
        _chk(_lib.evah_client_key_upload(self.h, 2, _p(pk)))

    def upload_secret_key(self, sk_ntt):
        sk = np.ascontiguousarray(sk_ntt, dtype=np.uint64)
        _chk(_lib.evah_client_key_upload(self.h, 3, _p(sk)))

    def encrypt(self, pt, small):
        """pt: Plaintext (NTT form); small: int8 [3][N] = (u, e0, e1)"""
        small = np.ascontiguousarray(small, dtype=np.int8)
        h = C.c_void_p()
        _chk(_lib.evah_encrypt(self.h, pt.h, small.ctypes.data_as(C.POINTER(C.c_int8)), C.byref(h)))
        return Ciphertext(self, h)

    def decrypt_decode(self, ct, n_out):
        out = np.empty(n_out, dtype=np.float64)
        _chk(_lib.evah_decrypt_decode(self.h, ct.h, int(n_out), out.ctypes.data_as(C.POINTER(C.c_double))))
        return out

    def copy_here(self, value):
        """a copy of a Ciphertext / Plaintext of another context (another GPU: peer copy), owned by this one"""
        h = C.c_void_p()
        if isinstance(value, Ciphertext):
            _chk(_lib.evah_ct_copy(self.h, value.h, C.byref(h)))
            return Ciphertext(self, h)
        _chk(_lib.evah_pt_copy(self.h, value.h, C.byref(h)))
        return Plaintext(self, h)

    def enable_peer(self, other):
        """peer access between this context's device and `other`'s, both directions (a refusal raises)"""
        _chk(_lib.evah_ctx_enable_peer(self.h, other.h))

    # ---- plumbing
    def set_stream(self, stream_ptr):
        _chk(_lib.evah_ctx_set_stream(self.h, C.c_void_p(stream_ptr)))

    def sync(self):
        _chk(_lib.evah_ctx_sync(self.h))

    def wait_for(self, other):
        """this queue waits (on the device) for everything enqueued so far on `other`"""
        _chk(_lib.evah_ctx_wait(self.h, other.h))

    def transfer_stats(self):
        """(ct uploads, ct downloads, pt uploads, pt downloads, bytes up, bytes down) of this device state"""
        out = (C.c_uint64 * 6)()
        _chk(_lib.evah_ctx_transfer_stats(self.h, out))
        return tuple(int(x) for x in out)

    def key_bytes(self):
        """bytes of HBM the evaluation keys of this device state occupy (a limb shard holds its prime rows only)"""
        out = C.c_uint64()
        _chk(_lib.evah_ctx_key_bytes(self.h, C.byref(out)))
        return int(out.value)

    def key_bytes_detail(self):
        """(key words as uploaded, radix-2^30 split copies, permuted copies used by hoisted rotation sets) in bytes"""
        out = (C.c_uint64 * 3)()
        _chk(_lib.evah_ctx_key_bytes_detail(self.h, out))
        return tuple(int(x) for x in out)

    def mem_info(self):
        a, b = C.c_size_t(), C.c_size_t()
        _chk(_lib.evah_ctx_mem_info(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def timer_start(self):
        _chk(_lib.evah_timer_start(self.h))

    def timer_stop(self):
        ms = C.c_float()
        _chk(_lib.evah_timer_stop(self.h, C.byref(ms)))
        return ms.value

    def profile(self, on):
        _chk(_lib.evah_profile_enable(self.h, 1 if on else 0))

    def profile_reset(self):
        _chk(_lib.evah_profile_reset(self.h))

    def profile_get(self):
        """{kernel class: (launches, total_ms)} measured with HIP events around each launch"""
        out = {}
        for i in range(_lib.evah_profile_classes()):
            n, ms = C.c_uint64(), C.c_double()
            _chk(_lib.evah_profile_get(self.h, i, C.byref(n), C.byref(ms)))
            out[_lib.evah_profile_class_name(i).decode()] = (n.value, ms.value)
        return out

    # ---- keys and values
    def upload_relin_key(self, key):
        key = np.ascontiguousarray(key, dtype=np.uint64)
        assert key.shape[1:] == (2, self.k, self.N)
        _chk(_lib.evah_key_upload(self.h, KEY_RELIN, 0, key.shape[0], _p(key)))

    def upload_galois_key(self, elt, key):
        key = np.ascontiguousarray(key, dtype=np.uint64)
        assert key.shape[1:] == (2, self.k, self.N)
        _chk(_lib.evah_key_upload(self.h, KEY_GALOIS, int(elt), key.shape[0], _p(key)))

    def galois_elt_from_step(self, steps):
        e = C.c_uint32()
        _chk(_lib.evah_galois_elt_from_step(self.h, int(steps), C.byref(e)))
        return e.value

    def upload_ct(self, data, scale):
        data = np.ascontiguousarray(data, dtype=np.uint64)
        size, limbs, n = data.shape
        assert n == self.N
        h = C.c_void_p()
        _chk(_lib.evah_ct_upload(self.h, size, limbs, float(scale), _p(data), C.byref(h)))
        return Ciphertext(self, h)

    def upload_ct_batch(self, data, scale):
        """data [batch][size][limbs][N] -> one batched handle"""
        data = np.ascontiguousarray(data, dtype=np.uint64)
        batch, size, limbs, n = data.shape
        assert n == self.N
        h = C.c_void_p()
        _chk(_lib.evah_ct_upload_batch(self.h, batch, size, limbs, float(scale), _p(data), C.byref(h)))
        return Ciphertext(self, h)

    def stack(self, cts):
        n = len(cts)
        ins = (C.c_void_p * n)(*[ct.h for ct in cts])
        h = C.c_void_p()
        _chk(_lib.evah_ct_stack(self.h, ins, n, C.byref(h)))
        return Ciphertext(self, h)

    def upload_pt(self, data, scale, coeff_form=False):
        data = np.ascontiguousarray(data, dtype=np.uint64)
        limbs, n = data.shape
        assert n == self.N
        h = C.c_void_p()
        fn = _lib.evah_pt_upload_coeff if coeff_form else _lib.evah_pt_upload
        _chk(fn(self.h, limbs, float(scale), _p(data), C.byref(h)))
        return Plaintext(self, h)

    def encode_pt(self, values, limbs, scale):
        """device CKKS encoder: reals (replicated over the slots) -> NTT-form plaintext"""
        v = np.ascontiguousarray(values, dtype=np.float64)
        h = C.c_void_p()
        _chk(_lib.evah_pt_encode(self.h, v.ctypes.data_as(C.POINTER(C.c_double)), v.shape[0], int(limbs), float(scale), C.byref(h)))
        return Plaintext(self, h)

    def uniform_pt(self, values, scale):
        values = np.ascontiguousarray(values, dtype=np.uint64)
        h = C.c_void_p()
        _chk(_lib.evah_pt_uniform(self.h, values.shape[0], float(scale), _p(values), C.byref(h)))
        return Plaintext(self, h)

    # ---- evaluator (one method per SEAL call in SEALExecutor)
    def _ct2(self, fn, a, b):
        h = C.c_void_p()
        _chk(fn(self.h, a.h, b.h, C.byref(h)))
        return Ciphertext(self, h)

    def _ct1(self, fn, a, *extra):
        h = C.c_void_p()
        _chk(fn(self.h, a.h, *extra, C.byref(h)))
        return Ciphertext(self, h)

    def add(self, a, b):
        return self._ct2(_lib.evah_add, a, b)

    def sub(self, a, b):
        return self._ct2(_lib.evah_sub, a, b)

    def add_plain(self, a, pt):
        return self._ct2(_lib.evah_add_plain, a, pt)

    def sub_plain(self, a, pt):
        return self._ct2(_lib.evah_sub_plain, a, pt)

    def multiply(self, a, b):
        return self._ct2(_lib.evah_multiply, a, b)

    def multiply_many(self, cts_a, cts_b):
        n = len(cts_a)
        ia = (C.c_void_p * n)(*[ct.h for ct in cts_a])
        ib = (C.c_void_p * n)(*[ct.h for ct in cts_b])
        outs = (C.c_void_p * n)()
        _chk(_lib.evah_multiply_many(self.h, ia, ib, n, outs))
        return [Ciphertext(self, C.c_void_p(outs[i])) for i in range(n)]

    def multiply_plain_many(self, cts, pts):
        n = len(cts)
        ia = (C.c_void_p * n)(*[ct.h for ct in cts])
        ib = (C.c_void_p * n)(*[pt.h for pt in pts])
        outs = (C.c_void_p * n)()
        _chk(_lib.evah_multiply_plain_many(self.h, ia, ib, n, outs))
        return [Ciphertext(self, C.c_void_p(outs[i])) for i in range(n)]

    # ---- graph capture of a sequence of calls on this context (single queue)
    def capture_begin(self):
        _chk(_lib.evah_capture_begin(self.h, None, 0))

    def capture_end(self):
        g = C.c_void_p()
        _chk(_lib.evah_capture_end(self.h, None, 0, C.byref(g)))
        return g

    def graph_launch(self, g):
        _chk(_lib.evah_graph_launch(self.h, g))

    def graph_free(self, g):
        _lib.evah_graph_free(g)

    def execute(self, ops, values, n_vals=None):
        """evah_execute: `ops` = [(op, dst, src0, src1, imm, flags)], `values` = {slot: Ciphertext |
        Plaintext} placed by the caller.  Returns {slot: handle} for every non-empty slot afterwards;
        wrappers whose handle the library released (EVAH_OPF_FREE_*) or moved (Output) are detached."""
        n_vals = n_vals if n_vals is not None else 1 + max([max(o[1], o[2], o[3]) for o in ops] + list(values))
        tab = (EvahVal * n_vals)()
        for i, v in values.items():
            tab[i].kind = VAL_CT if isinstance(v, Ciphertext) else VAL_PT
            tab[i].h = v.h.value if isinstance(v.h, C.c_void_p) else v.h
        arr = (EvahOp * len(ops))(*[EvahOp(*o) for o in ops])
        before = {i: tab[i].h for i in values}
        rc = _lib.evah_execute(self.h, arr, len(ops), tab, n_vals)
        out = {}
        for i in range(n_vals):
            if tab[i].kind == VAL_NONE:
                continue
            if i in values and tab[i].h == before[i]:
                out[i] = values[i]
            else:
                cls = Ciphertext if tab[i].kind == VAL_CT else Plaintext
                out[i] = cls(self, C.c_void_p(tab[i].h))
        for i, v in values.items():  # released or moved inside the library: this wrapper no longer owns it
            if out.get(i) is not v:
                v.h = None
        _chk(rc)
        return out

    def elementwise_program(self, inputs, ops, outs):
        """evah_elementwise_program: inputs = [Ciphertext | Plaintext], ops = [(op, a, b)] with op in {10 Negate, 11 Add,
        12 Sub, 13 Mul} over value indices (inputs first, then one value per op), outs = value indices -> [Ciphertext]"""
        tab = (EvahVal * len(inputs))()
        for i, v in enumerate(inputs):
            tab[i].kind = VAL_CT if isinstance(v, Ciphertext) else VAL_PT
            tab[i].h = v.h.value if isinstance(v.h, C.c_void_p) else v.h
        arr = (EvahEwOp * max(len(ops), 1))(*[EvahEwOp(*o) for o in ops])
        ov = (C.c_uint32 * len(outs))(*[int(x) for x in outs])
        res = (C.c_void_p * len(outs))()
        _chk(_lib.evah_elementwise_program(self.h, tab, len(inputs), arr, len(ops), ov, len(outs), res))
        return [Ciphertext(self, C.c_void_p(res[i])) for i in range(len(outs))]

    def weighted_sum(self, cts, pts):
        """sum_j cts[j] (*) pts[j]; pts[j] may be None (the ciphertext itself)"""
        n = len(cts)
        ic = (C.c_void_p * n)(*[ct.h for ct in cts])
        ip = (C.c_void_p * n)(*[(pt.h if pt is not None else None) for pt in pts])
        h = C.c_void_p()
        _chk(_lib.evah_weighted_sum(self.h, ic, ip, n, C.byref(h)))
        return Ciphertext(self, h)

    def multiply_plain(self, a, pt):
        return self._ct2(_lib.evah_multiply_plain, a, pt)

    def negate(self, a):
        return self._ct1(_lib.evah_negate, a)

    def square(self, a):
        return self._ct1(_lib.evah_square, a)

    def relinearize(self, a):
        return self._ct1(_lib.evah_relinearize, a)

    def relinearize_rescale(self, a, divisor_bits):
        return self._ct1(_lib.evah_relinearize_rescale, a, C.c_uint32(int(divisor_bits)))

    def relinearize_rescale_many(self, cts, divisor_bits):
        n = len(cts)
        ins = (C.c_void_p * n)(*[ct.h for ct in cts])
        outs = (C.c_void_p * n)()
        _chk(_lib.evah_relinearize_rescale_many(self.h, ins, n, C.c_uint32(int(divisor_bits)), outs))
        return [Ciphertext(self, C.c_void_p(outs[i])) for i in range(n)]

    def multiply_relinearize_rescale(self, a, b, divisor_bits):
        out = C.c_void_p()
        _chk(_lib.evah_multiply_relinearize_rescale(self.h, a.h, b.h, C.c_uint32(int(divisor_bits)), C.byref(out)))
        return Ciphertext(self, out)

    def multiply_relinearize_rescale_many(self, cts_a, cts_b, divisor_bits):
        n = len(cts_a)
        ia = (C.c_void_p * n)(*[ct.h for ct in cts_a])
        ib = (C.c_void_p * n)(*[ct.h for ct in cts_b])
        outs = (C.c_void_p * n)()
        _chk(_lib.evah_multiply_relinearize_rescale_many(self.h, ia, ib, n, C.c_uint32(int(divisor_bits)), outs))
        return [Ciphertext(self, C.c_void_p(outs[i])) for i in range(n)]

    def multiply_rescale_relinearize(self, a, b, divisor_bits):
        """multiply (a is b: square) -> rescale -> relinearize as one call (lazy relinearization's order)"""
        out = C.c_void_p()
        _chk(_lib.evah_multiply_rescale_relinearize(self.h, a.h, b.h, C.c_uint32(int(divisor_bits)), C.byref(out)))
        return Ciphertext(self, out)

    def rescale_relinearize(self, a, divisor_bits):
        """rescale_to_next of a size-3 ciphertext -> relinearize as one call"""
        out = C.c_void_p()
        _chk(_lib.evah_rescale_relinearize(self.h, a.h, C.c_uint32(int(divisor_bits)), C.byref(out)))
        return Ciphertext(self, out)

    def rescale_relinearize_many(self, cts, divisor_bits):
        n = len(cts)
        ins = (C.c_void_p * n)(*[ct.h for ct in cts])
        outs = (C.c_void_p * n)()
        _chk(_lib.evah_rescale_relinearize_many(self.h, ins, n, C.c_uint32(int(divisor_bits)), outs))
        return [Ciphertext(self, C.c_void_p(outs[i])) for i in range(n)]

    def multiply_rescale_relinearize_many(self, cts_a, cts_b, divisor_bits):
        n = len(cts_a)
        ia = (C.c_void_p * n)(*[ct.h for ct in cts_a])
        ib = (C.c_void_p * n)(*[ct.h for ct in cts_b])
        outs = (C.c_void_p * n)()
        _chk(_lib.evah_multiply_rescale_relinearize_many(self.h, ia, ib, n, C.c_uint32(int(divisor_bits)), outs))
        return [Ciphertext(self, C.c_void_p(outs[i])) for i in range(n)]

    def rotate(self, a, steps):
        return self._ct1(_lib.evah_rotate, a, C.c_int32(int(steps)))

    def rotate_many(self, a, steps):
        n = len(steps)
        arr = (C.c_int32 * n)(*[int(x) for x in steps])
        outs = (C.c_void_p * n)()
        _chk(_lib.evah_rotate_many(self.h, a.h, arr, n, outs))
        return [Ciphertext(self, C.c_void_p(outs[i])) for i in range(n)]

    def _many(self, fn, cts, *extra):
        n = len(cts)
        ins = (C.c_void_p * n)(*[ct.h for ct in cts])
        outs = (C.c_void_p * n)()
        _chk(fn(self.h, ins, *extra, outs))
        return [Ciphertext(self, C.c_void_p(outs[i])) for i in range(n)]

    def rotate_pairs(self, cts, steps):
        n = len(cts)
        arr = (C.c_int32 * n)(*[int(x) for x in steps])
        return self._many(_lib.evah_rotate_pairs, cts, arr, C.c_uint32(n))

    def rotate_weighted_sums(self, windows):
        """windows: [(terms, weights)] with terms = [(ciphertext, steps)] and weights = one list per sum of the window,
        each holding a plaintext (or None = 1) per term -> the sums of all windows in order:
        out = sum_t weights[s][t] (*) rotate(terms[t])  (steps 0 = the ciphertext itself)"""
        cts, steps, wt, ws, pts = [], [], [], [], []
        for terms, weights in windows:
            wt.append(len(terms))
            ws.append(len(weights))
            for ct, st in terms:
                cts.append(ct.h)
                steps.append(int(st))
            for row in weights:
                assert len(row) == len(terms)
                pts += [(pt.h if pt is not None else None) for pt in row]
        n_sums = sum(ws)
        outs = (C.c_void_p * n_sums)()
        _chk(_lib.evah_rotate_weighted_sums(self.h, (C.c_void_p * len(cts))(*cts), (C.c_int32 * len(steps))(*steps),
                                            (C.c_uint32 * len(wt))(*wt), (C.c_uint32 * len(ws))(*ws), len(wt),
                                            (C.c_void_p * len(pts))(*pts), outs))
        return [Ciphertext(self, C.c_void_p(outs[i])) for i in range(n_sums)]

    def rescale_many(self, cts, divisor_bits):
        return self._many(_lib.evah_rescale_many, cts, C.c_uint32(len(cts)), C.c_uint32(int(divisor_bits)))

    def relinearize_many(self, cts):
        return self._many(_lib.evah_relinearize_many, cts, C.c_uint32(len(cts)))

    def rescale(self, a, divisor_bits):
        return self._ct1(_lib.evah_rescale, a, C.c_uint32(int(divisor_bits)))

    def mod_switch(self, a):
        return self._ct1(_lib.evah_mod_switch, a)

    # ---- test hook
    def test_ntt(self, prime_idx, x, inverse=False):
        y = np.ascontiguousarray(x, dtype=np.uint64).copy()
        _chk(_lib.evah_test_ntt(self.h, int(prime_idx), 1 if inverse else 0, _p(y)))
        return y
