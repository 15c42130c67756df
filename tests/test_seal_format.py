"""The SEAL-object messages of the reference's wire format (eva_amd/host/seal_format.h; SURVEY.md 8(f) rank 4:
/root/reference/eva/serialization/seal.proto:10-44, seal_serialization.cpp:46-229): SEALValuation / SEALPublic /
SEALSecret inside the KnownType envelope, every SEALObject.data in Microsoft SEAL 3.6's binary object format.

SEAL is not in this image, so these tests pin the C++ writer / reader against a SECOND, independent restatement of
the same format — the `Seal*` helpers below, written with struct, hashlib.blake2b, zlib and the official protobuf
runtime (messages built from the reference's schema) — in both directions, byte for byte:
  * what save(format="seal") writes parses with real protobuf, and every blob equals the Python writer's bytes;
  * what the Python writer produces (as microsoft/EVA + SEAL would, also zlib- and zstd-compressed, also with
    SEAL 4.x's extra ciphertext field) loads here into the same keys and values;
  * hostile blobs (truncated, wrong magic, wrong parms_id, oversize dimensions, zip bombs) are errors.
Agreement of two restatements made from the same knowledge is not agreement with SEAL: that check is
tools/seal_parity.cpp section 7, run where SEAL is installed.  The header of seal_format.h says "parity unpinned"."""
import hashlib
import struct
import zlib

import numpy as np
import pytest

pb = pytest.importorskip("google.protobuf")
from google.protobuf import any_pb2, descriptor_pb2, descriptor_pool, message_factory  # noqa: E402

from eva import EvaProgram, Input, Output, save, load  # noqa: E402
from eva.ckks import CKKSCompiler  # noqa: E402
from eva.seal import generate_keys, SEALValuation  # noqa: E402
from eva_amd import _eva  # noqa: E402

F = descriptor_pb2.FieldDescriptorProto


# ------------------------------------------------------------------ the independent restatement (test side only)
def seal_header(compr, total):
    return struct.pack("<HBBBBHQ", 0xA15E, 0x10, 3, 6, compr, 0, total)


def seal_wrap(members, compr=0, version=(3, 6)):
    body = members if compr == 0 else zlib.compress(members) if compr == 1 else _zstd(members)
    return struct.pack("<HBBBBHQ", 0xA15E, 0x10, version[0], version[1], compr, 0, 16 + len(body)) + body


def _zstd(data):
    import ctypes
    z = ctypes.CDLL("libzstd.so.1")
    z.ZSTD_compressBound.restype = ctypes.c_size_t
    z.ZSTD_compressBound.argtypes = [ctypes.c_size_t]
    z.ZSTD_compress.restype = ctypes.c_size_t
    z.ZSTD_compress.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_int]
    cap = z.ZSTD_compressBound(len(data))
    buf = ctypes.create_string_buffer(cap)
    n = z.ZSTD_compress(buf, cap, data, len(data), 3)
    return buf.raw[:n]


def seal_parms_id(N, primes):
    words = [2, N] + [int(q) for q in primes] + [0]
    return hashlib.blake2b(struct.pack("<%dQ" % len(words), *words), digest_size=32).digest()


def seal_modulus(q):
    return seal_wrap(struct.pack("<Q", int(q)))


def seal_parms(N, primes, compr=0):
    m = struct.pack("<BQQ", 2, N, len(primes)) + b"".join(seal_modulus(q) for q in primes) + seal_modulus(0)
    return seal_wrap(m, compr)


def seal_dynarray(words):
    words = np.ascontiguousarray(words, dtype=np.uint64).reshape(-1)
    return seal_wrap(struct.pack("<Q", words.size) + words.tobytes())


def seal_ciphertext(N, primes, data, scale, compr=0, version=(3, 6)):
    size, limbs, n = data.shape
    m = seal_parms_id(N, primes[:limbs]) + struct.pack("<BQQQd", 1, size, n, limbs, scale)
    if version[0] >= 4:
        m += struct.pack("<Q", 1)  # correction_factor
    return seal_wrap(m + seal_dynarray(data), compr, version)


def seal_plaintext(N, primes, data, scale, compr=0):
    limbs, n = data.shape
    return seal_wrap(seal_parms_id(N, primes[:limbs]) + struct.pack("<Qd", limbs * n, scale) + seal_dynarray(data), compr)


def seal_public_key(N, primes, data, compr=0, nested=False):
    """SEAL >= 3.5: PublicKey::save is pk_.save — the ciphertext object itself (nested: the pre-3.5 double header,
    which the reader still accepts)"""
    if nested:
        return seal_wrap(seal_ciphertext(N, primes, data, 1.0), compr)
    return seal_ciphertext(N, primes, data, 1.0, compr)


def seal_secret_key(N, primes, s_ntt, compr=0, nested=False):
    if nested:
        return seal_wrap(seal_plaintext(N, primes, s_ntt, 1.0), compr)
    return seal_plaintext(N, primes, s_ntt, 1.0, compr)


def seal_kswitch(N, primes, dim1, slots, compr=0, nested=False):
    """slots: {index: key array [digits][2][k][N]}"""
    m = seal_parms_id(N, primes) + struct.pack("<Q", dim1)
    for i in range(dim1):
        key = slots.get(i)
        if key is None:
            m += struct.pack("<Q", 0)
        else:
            m += struct.pack("<Q", key.shape[0]) + b"".join(seal_public_key(N, primes, key[j], nested=nested) for j in range(key.shape[0]))
    return seal_wrap(m, compr)


def _schema():
    pool = descriptor_pool.DescriptorPool()
    pool.Add(descriptor_pb2.FileDescriptorProto.FromString(any_pb2.DESCRIPTOR.serialized_pb))
    fd = descriptor_pb2.FileDescriptorProto(name="seal_all.proto", package="eva.msg", syntax="proto3",
                                            dependency=["google/protobuf/any.proto"])
    S, M = F.LABEL_OPTIONAL, F.LABEL_REPEATED

    def msg(name, fields):
        m = fd.message_type.add(name=name)
        for f in fields:
            m.field.add(**f)
        return m

    def map_entry(parent, name, value_type):
        e = parent.nested_type.add(name=name)
        e.options.map_entry = True
        e.field.add(name="key", number=1, type=F.TYPE_STRING, label=S)
        e.field.add(name="value", number=2, type=F.TYPE_MESSAGE, type_name=value_type, label=S)
    msg("ConstantValue", [dict(name="size", number=1, type=F.TYPE_UINT32, label=S),
                          dict(name="values", number=2, type=F.TYPE_DOUBLE, label=M),
                          dict(name="sparse_indices", number=3, type=F.TYPE_UINT32, label=M)])
    so = msg("SEALObject", [dict(name="seal_type", number=1, type=F.TYPE_ENUM, type_name=".eva.msg.SEALObject.SEALType", label=S),
                            dict(name="data", number=2, type=F.TYPE_BYTES, label=S)])
    en = so.enum_type.add(name="SEALType")
    for i, n in enumerate(("UNKNOWN", "CIPHERTEXT", "PLAINTEXT", "SECRET_KEY", "PUBLIC_KEY", "GALOIS_KEYS", "RELIN_KEYS", "ENCRYPTION_PARAMETERS")):
        en.value.add(name=n, number=i)
    O = ".eva.msg.SEALObject"
    msg("SEALPublic", [dict(name="encryption_parameters", number=1, type=F.TYPE_MESSAGE, type_name=O, label=S),
                       dict(name="public_key", number=2, type=F.TYPE_MESSAGE, type_name=O, label=S),
                       dict(name="galois_keys", number=3, type=F.TYPE_MESSAGE, type_name=O, label=S),
                       dict(name="relin_keys", number=4, type=F.TYPE_MESSAGE, type_name=O, label=S)])
    msg("SEALSecret", [dict(name="encryption_parameters", number=1, type=F.TYPE_MESSAGE, type_name=O, label=S),
                       dict(name="secret_key", number=2, type=F.TYPE_MESSAGE, type_name=O, label=S)])
    val = msg("SEALValuation", [dict(name="encryption_parameters", number=1, type=F.TYPE_MESSAGE, type_name=O, label=S),
                                dict(name="values", number=2, type=F.TYPE_MESSAGE, type_name=".eva.msg.SEALValuation.ValuesEntry", label=M),
                                dict(name="raw_values", number=3, type=F.TYPE_MESSAGE, type_name=".eva.msg.SEALValuation.RawValuesEntry", label=M)])
    map_entry(val, "ValuesEntry", O)
    map_entry(val, "RawValuesEntry", ".eva.msg.ConstantValue")
    msg("KnownType", [dict(name="contents", number=1, type=F.TYPE_MESSAGE, type_name=".google.protobuf.Any", label=S),
                      dict(name="creator", number=2, type=F.TYPE_STRING, label=S)])
    pool.Add(fd)
    return {n: message_factory.GetMessageClass(pool.FindMessageTypeByName("eva.msg." + n))
            for n in ("SEALObject", "SEALPublic", "SEALSecret", "SEALValuation", "KnownType", "ConstantValue")}


@pytest.fixture(scope="module")
def schema():
    return _schema()


@pytest.fixture(scope="module")
def keys():
    prog = EvaProgram('fmt', vec_size=16)
    with prog:
        x, y = Input('x'), Input('y', is_encrypted=False)
        Output('z', (x << 1) * x + y + (x >> 3))
    prog.set_input_scales(30)
    prog.set_output_ranges(20)
    compiled, params, sig = CKKSCompiler(config={'warn_vec_size': 'false'}).compile(prog)
    params.poly_modulus_degree = 1024
    pub, sec = generate_keys(params, 9)
    return compiled, params, sig, pub, sec


def _open(schema, path, inner):
    kt = schema["KnownType"]()
    kt.ParseFromString(open(path, "rb").read())
    assert kt.contents.type_url == "type.googleapis.com/eva.msg." + inner
    m = schema[inner]()
    m.ParseFromString(kt.contents.value)
    return m


def _envelope(schema, m):
    kt = schema["KnownType"](creator="microsoft/EVA")
    kt.contents.Pack(m)
    return kt.SerializeToString()


def _ctx(pub):
    return int(pub.poly_modulus_degree), [int(q) for q in pub.primes]


def test_blake2b_is_rfc7693():
    # RFC 7693 appendix A: BLAKE2b-512("abc") starts BA 80 A5 3F ...; the 256-bit digests against hashlib
    for data in (b"", b"abc", bytes(range(128)), bytes(range(129)), b"x" * 1000):
        assert _eva._blake2b_256(data) == hashlib.blake2b(data, digest_size=32).digest()


def test_public_context_round_trip_and_bytes(schema, keys, tmp_path):
    compiled, params, sig, pub, sec = keys
    N, primes = _ctx(pub)
    k = len(primes)
    path = str(tmp_path / "pub.seal")
    save(pub, path, format="seal")
    m = _open(schema, path, "SEALPublic")
    T = schema["SEALObject"]
    assert (m.encryption_parameters.seal_type, m.public_key.seal_type, m.galois_keys.seal_type, m.relin_keys.seal_type) == (7, 4, 5, 6)
    assert m.encryption_parameters.data == seal_parms(N, primes)
    assert m.public_key.data == seal_public_key(N, primes, pub.public_key())
    assert m.relin_keys.data == seal_kswitch(N, primes, 1, {0: pub.relin_key()})
    gal = {(int(e) - 1) // 2: key for e, key in pub.galois_keys().items()}
    assert len(gal) == 2 and m.galois_keys.data == seal_kswitch(N, primes, N, gal)
    # header fields of the outer object: magic, 16-byte header, 3.6, uncompressed, total size
    magic, hs, vmaj, vmin, compr, _, size = struct.unpack_from("<HBBBBHQ", m.public_key.data)
    assert (magic, hs, vmaj, vmin, compr, size) == (0xA15E, 16, 3, 6, 0, len(m.public_key.data))
    again = load(path)
    assert [int(q) for q in again.primes] == primes and again.poly_modulus_degree == N
    assert np.array_equal(again.public_key(), pub.public_key()) and np.array_equal(again.relin_key(), pub.relin_key())
    assert sorted(again.galois_keys()) == sorted(pub.galois_keys())
    for e, key in pub.galois_keys().items():
        assert np.array_equal(again.galois_keys()[e], key)
    # and the other way: the file microsoft/EVA would write — real protobuf around the Python writer's blobs,
    # compressed the way SEAL's default build does
    for compr in (0, 1, 2):
        if compr == 2 and not _eva._seal_zstd_available():
            continue
        theirs = schema["SEALPublic"](
            encryption_parameters=T(seal_type=7, data=seal_parms(N, primes, compr)),
            public_key=T(seal_type=4, data=seal_public_key(N, primes, pub.public_key(), compr)),
            galois_keys=T(seal_type=5, data=seal_kswitch(N, primes, N, gal, compr)),
            relin_keys=T(seal_type=6, data=seal_kswitch(N, primes, 1, {0: pub.relin_key()}, compr)))
        p2 = str(tmp_path / f"theirs{compr}")
        open(p2, "wb").write(_envelope(schema, theirs))
        got = load(p2)
        assert np.array_equal(got.public_key(), pub.public_key()) and np.array_equal(got.relin_key(), pub.relin_key())
        assert sorted(got.galois_keys()) == sorted(pub.galois_keys())
    # the pre-3.5 nesting (a second header around the key's ciphertext — what r03 of this repo wrote) still loads
    nested_gal = {i: key for i, key in gal.items()}
    old = schema["SEALPublic"](
        encryption_parameters=T(seal_type=7, data=seal_parms(N, primes)),
        public_key=T(seal_type=4, data=seal_public_key(N, primes, pub.public_key(), nested=True)),
        galois_keys=T(seal_type=5, data=seal_kswitch(N, primes, N, nested_gal, nested=True)),
        relin_keys=T(seal_type=6, data=seal_kswitch(N, primes, 1, {0: pub.relin_key()}, nested=True)))
    p_old = str(tmp_path / "nested")
    open(p_old, "wb").write(_envelope(schema, old))
    got = load(p_old)
    assert np.array_equal(got.public_key(), pub.public_key()) and np.array_equal(got.relin_key(), pub.relin_key())
    for fmt in ("seal+zlib",) + (("seal+zstd",) if _eva._seal_zstd_available() else ()):
        p3 = str(tmp_path / fmt)
        save(pub, p3, format=fmt)
        assert np.array_equal(load(p3).relin_key(), pub.relin_key())
        assert len(open(p3, "rb").read()) < len(open(path, "rb").read())


def test_context_without_rotations_carries_n_empty_galois_slots(schema, tmp_path):
    """KeyGenerator::create_galois_keys sizes the key vector to poly_modulus_degree whatever the steps are, and the
    reference calls it for every program (/root/reference/eva/seal/seal.cpp:195)"""
    prog = EvaProgram('norot', vec_size=8)
    with prog:
        x = Input('x')
        Output('y', x * x)
    prog.set_input_scales(30)
    prog.set_output_ranges(20)
    compiled, params, sig = CKKSCompiler(config={'warn_vec_size': 'false'}).compile(prog)
    params.poly_modulus_degree = 1024
    pub, sec = generate_keys(params, 2)
    N, primes = _ctx(pub)
    path = str(tmp_path / "pub")
    save(pub, path, format="seal")
    m = _open(schema, path, "SEALPublic")
    assert m.galois_keys.data == seal_kswitch(N, primes, N, {})
    assert load(path).galois_keys() == {}


def test_secret_context_round_trip_and_bytes(schema, keys, tmp_path):
    compiled, params, sig, pub, sec = keys
    N, primes = _ctx(pub)
    path = str(tmp_path / "sec.seal")
    save(sec, path, format="seal")
    m = _open(schema, path, "SEALSecret")
    assert m.encryption_parameters.data == seal_parms(N, primes)
    assert m.secret_key.seal_type == 3 and m.secret_key.data == seal_secret_key(N, primes, sec._secret_key_ntt())
    again = load(path)
    assert np.array_equal(again._secret_key_ntt(), sec._secret_key_ntt())
    nested = schema["SEALSecret"](encryption_parameters=schema["SEALObject"](seal_type=7, data=seal_parms(N, primes)),
                                  secret_key=schema["SEALObject"](seal_type=3, data=seal_secret_key(N, primes, sec._secret_key_ntt(), nested=True)))
    p_old = str(tmp_path / "nested_secret")
    open(p_old, "wb").write(_envelope(schema, nested))
    assert np.array_equal(load(p_old)._secret_key_ntt(), sec._secret_key_ntt())
    # the reloaded key decrypts what the original public context encrypts
    enc = pub.encrypt({'x': [0.5 * i for i in range(16)], 'y': [1.0] * 16}, sig)
    vals = again.decrypt(enc, sig)
    assert np.abs(np.array(vals['x']) - np.array([0.5 * i for i in range(16)])).max() < 1e-4
    # rows that are not one ternary polynomial are refused
    T = schema["SEALObject"]
    bad = np.array(sec._secret_key_ntt())
    bad[1, 5] ^= np.uint64(1)
    theirs = schema["SEALSecret"](encryption_parameters=T(seal_type=7, data=seal_parms(N, primes)),
                                  secret_key=T(seal_type=3, data=seal_secret_key(N, primes, bad)))
    p2 = str(tmp_path / "bad")
    open(p2, "wb").write(_envelope(schema, theirs))
    with pytest.raises(RuntimeError, match="not one polynomial|not reduced"):
        load(p2)
    bad = np.array(sec._secret_key_ntt())
    bad[0, 0] = (int(bad[0, 0]) + 1) % primes[0]
    theirs.secret_key.data = seal_secret_key(N, primes, bad)
    open(p2, "wb").write(_envelope(schema, theirs))
    with pytest.raises(RuntimeError, match="ternary"):
        load(p2)


def test_valuation_round_trip_and_bytes(schema, keys, tmp_path):
    compiled, params, sig, pub, sec = keys
    N, primes = _ctx(pub)
    inputs = {'x': [0.25 * i for i in range(16)], 'y': [2.0] * 16}
    enc = pub.encrypt(inputs, sig)
    path = str(tmp_path / "val.seal")
    save(enc, path, format="seal")
    m = _open(schema, path, "SEALValuation")
    assert m.encryption_parameters.data == seal_parms(N, primes)
    kinds = {n: enc.get(n)[0] for n in enc.names()}
    assert set(m.values) | set(m.raw_values) == set(kinds)
    for name, kind in kinds.items():
        g = enc.get(name)
        if kind == "cipher":
            assert m.values[name].seal_type == 1 and m.values[name].data == seal_ciphertext(N, primes, np.array(g[4]), g[3])
        elif kind == "plain":
            assert m.values[name].seal_type == 2 and m.values[name].data == seal_plaintext(N, primes, np.array(g[4]), g[3])
        else:
            assert list(m.raw_values[name].values) == list(g[4]) and m.raw_values[name].size == len(g[4])
    again = load(path)
    assert sorted(again.names()) == sorted(enc.names())
    for name in enc.names():
        a, b = again.get(name), enc.get(name)
        assert a[:4] == b[:4] and np.array_equal(a[4], b[4])
    assert np.allclose(sec.decrypt(again, sig)['x'], inputs['x'], atol=1e-4)
    # files as the reference + SEAL would write them: compressed blobs, a mod-switched ciphertext, a sparse raw
    # value, a SEAL 4.x ciphertext (one extra header word)
    T, C = schema["SEALObject"], schema["ConstantValue"]
    cx = enc.get('x')
    ct = np.array(cx[4])
    low = np.ascontiguousarray(ct[:, :ct.shape[1] - 1, :])
    for compr, version in ((1, (3, 6)), (2, (3, 6)), (0, (4, 1)), (0, (3, 5))):
        if compr == 2 and not _eva._seal_zstd_available():
            continue
        theirs = schema["SEALValuation"](encryption_parameters=T(seal_type=7, data=seal_parms(N, primes, compr)))
        theirs.values['x'].CopyFrom(T(seal_type=1, data=seal_ciphertext(N, primes, ct, cx[3], compr, version)))
        theirs.values['low'].CopyFrom(T(seal_type=1, data=seal_ciphertext(N, primes, low, cx[3], compr, version)))
        theirs.raw_values['r'].CopyFrom(C(size=8, values=[1.5, -2.0], sparse_indices=[1, 6]))
        theirs.raw_values['d'].CopyFrom(C(size=4, values=[3.0]))
        p2 = str(tmp_path / f"theirs{compr}{version[0]}")
        open(p2, "wb").write(_envelope(schema, theirs))
        got = load(p2)
        assert np.array_equal(got.get('x')[4], ct) and got.get('x')[3] == cx[3]
        assert got.get('low')[2] == ct.shape[1] - 1 and np.array_equal(got.get('low')[4], low)
        assert got.get('r')[4] == [0, 1.5, 0, 0, 0, 0, -2.0, 0] and got.get('d')[4] == [3.0] * 4
    # a valuation assembled by hand has no encryption parameters to write
    v = SEALValuation()
    v._set_cipher('x', ct, cx[3])
    with pytest.raises(RuntimeError, match="no encryption parameters"):
        save(v, str(tmp_path / "none"), format="seal")
    v._set_params(pub)
    save(v, str(tmp_path / "byhand"), format="seal")
    assert np.array_equal(load(str(tmp_path / "byhand")).get('x')[4], ct)
    # a ciphertext that does not fit the parameters it is filed under is refused at save time
    v._set_cipher('bad', ct[:, :, :512], cx[3])
    with pytest.raises(RuntimeError, match="does not match the valuation's encryption parameters"):
        save(v, str(tmp_path / "bad"), format="seal")


def test_hostile_seal_objects_are_errors(schema, keys, tmp_path):
    compiled, params, sig, pub, sec = keys
    N, primes = _ctx(pub)
    T = schema["SEALObject"]
    enc = pub.encrypt({'x': [1.0] * 16, 'y': [1.0] * 16}, sig)
    ct, scale = np.array(enc.get('x')[4]), enc.get('x')[3]
    good = seal_ciphertext(N, primes, ct, scale)
    path = str(tmp_path / "h")

    def val_with(blob, seal_type=1, parms=None):
        m = schema["SEALValuation"](encryption_parameters=T(seal_type=7, data=parms if parms is not None else seal_parms(N, primes)))
        m.values['x'].CopyFrom(T(seal_type=seal_type, data=blob))
        open(path, "wb").write(_envelope(schema, m))
        return path
    assert np.array_equal(load(val_with(good)).get('x')[4], ct)
    cases = [
        (good[:40], "truncated"),                                                   # cut inside the members
        (b"\x00\x00" + good[2:], "bad magic"),
        (good[:2] + b"\x20" + good[3:], "header size"),
        (good[:3] + b"\x02" + good[4:], "unsupported version"),
        (good[:5] + b"\x07" + good[6:], "compression mode"),
        (good[:8] + struct.pack("<Q", 2 ** 40) + good[16:], "truncated"),           # size field beyond the buffer
        (good[:8] + struct.pack("<Q", 8) + good[16:], "smaller than its header"),
        (seal_ciphertext(N, primes[::-1], ct, scale), "parms_id"),                  # a level of some other context
        (seal_wrap(seal_parms_id(N, primes[:ct.shape[1]]) + struct.pack("<BQQQd", 0, 2, N, ct.shape[1], scale) + seal_dynarray(ct)), "NTT form"),
        (seal_wrap(seal_parms_id(N, primes[:ct.shape[1]]) + struct.pack("<BQQQd", 1, 2 ** 40, N, ct.shape[1], scale) + seal_dynarray(ct)), "shape"),
        (seal_wrap(seal_parms_id(N, primes[:ct.shape[1]]) + struct.pack("<BQQQd", 1, 2, N, ct.shape[1], -1.0) + seal_dynarray(ct)), "scale"),
        (seal_wrap(seal_parms_id(N, primes[:ct.shape[1]]) + struct.pack("<BQQQd", 1, 2, N, ct.shape[1], scale) + seal_dynarray(ct.reshape(-1)[:-5])), "wrong length|seeded"),
        (seal_header(1, 16 + 30) + zlib.compress(b"\0" * (1 << 26))[:30], "zlib|truncated"),
        (seal_wrap(b"\0" * (1 << 27), 1), "expands beyond its bound"),             # zip bomb: 128 MiB of zeros in ~130 KB
    ]
    for blob, msg in cases:
        with pytest.raises(RuntimeError, match=msg):
            load(val_with(blob))
    with pytest.raises(RuntimeError, match="Not a ciphertext or plaintext"):
        load(val_with(good, seal_type=4))
    with pytest.raises(RuntimeError, match="UNKNOWN"):
        load(_no_parms(schema, good, path))
    # parameters: not CKKS, composite modulus, absurd degree
    for parms, msg in ((seal_wrap(struct.pack("<BQQ", 1, N, 2) + seal_modulus(primes[0]) + seal_modulus(primes[1]) + seal_modulus(0)), "not CKKS"),
                       (seal_parms(N, [primes[0], primes[1] - 2]), "coefficient modulus"),
                       (seal_parms(N, [primes[0], primes[0]]), "distinct"),
                       (seal_wrap(struct.pack("<BQQ", 2, 2 ** 40, 2)), "invalid encryption parameters"),
                       (seal_wrap(struct.pack("<BQQ", 2, N, 2 ** 50)), "invalid encryption parameters")):
        with pytest.raises(RuntimeError, match=msg):
            load(val_with(good, parms=parms))
    # key sets: more slots than the degree, a wrong decomposition count, a key of another context
    k = len(primes)
    pk, rk = pub.public_key(), pub.relin_key()

    def pub_with(galois=None, relin=None, public=None):
        m = schema["SEALPublic"](encryption_parameters=T(seal_type=7, data=seal_parms(N, primes)),
                                 public_key=T(seal_type=4, data=public if public is not None else seal_public_key(N, primes, pk)),
                                 galois_keys=T(seal_type=5, data=galois if galois is not None else seal_kswitch(N, primes, 0, {})),
                                 relin_keys=T(seal_type=6, data=relin if relin is not None else seal_kswitch(N, primes, 1, {0: rk})))
        open(path, "wb").write(_envelope(schema, m))
        return path
    assert np.array_equal(load(pub_with()).relin_key(), rk)
    with pytest.raises(RuntimeError, match="too many entries"):
        load(pub_with(galois=seal_wrap(seal_parms_id(N, primes) + struct.pack("<Q", 2 ** 50))))
    with pytest.raises(RuntimeError, match="decomposition count"):
        load(pub_with(relin=seal_kswitch(N, primes, 1, {0: rk[:k - 2]})))
    with pytest.raises(RuntimeError, match="too many entries"):
        load(pub_with(relin=seal_kswitch(N, primes, 2, {0: rk, 1: rk})))
    with pytest.raises(RuntimeError, match="do not belong"):
        load(pub_with(relin=seal_wrap(seal_parms_id(N, primes[:-1]) + struct.pack("<QQ", 1, 0))))
    bad = np.array(pk)
    bad[1, 0, 3] = np.uint64(2 ** 63)
    with pytest.raises(RuntimeError, match="not reduced modulo its prime"):
        load(pub_with(public=seal_public_key(N, primes, bad)))
    with pytest.raises(RuntimeError, match="type mismatch"):
        m = schema["SEALPublic"](encryption_parameters=T(seal_type=7, data=seal_parms(N, primes)), public_key=T(seal_type=3, data=seal_public_key(N, primes, pk)),
                                 galois_keys=T(seal_type=5, data=seal_kswitch(N, primes, 0, {})), relin_keys=T(seal_type=6, data=seal_kswitch(N, primes, 1, {0: rk})))
        open(path, "wb").write(_envelope(schema, m))
        load(path)


def _no_parms(schema, blob, path):
    m = schema["SEALValuation"]()
    m.values['x'].CopyFrom(schema["SEALObject"](seal_type=1, data=blob))
    open(path, "wb").write(_envelope(schema, m))
    return path


@pytest.mark.gpu
def test_all_six_object_kinds_through_the_reference_formats_on_the_gpu(keys, tmp_path):
    """the reference's own round-trip test (/root/reference/tests/features.py:154-217) with every file in the
    reference's formats: program / parameters / signature as EVA protobuf, both contexts and both valuations as
    seal.proto messages around SEAL objects — execute() on the loaded objects gives the same ciphertext bits"""
    from eva import evaluate
    compiled, params, sig, pub, sec = keys
    inputs = {'x': [0.1 * i for i in range(16)], 'y': [0.5] * 16}
    enc = pub.encrypt(inputs, sig)
    out = pub.execute(compiled, enc)
    files = {n: str(tmp_path / n) for n in ("prog", "params", "sig", "pub", "sec", "enc", "out")}
    save(compiled, files["prog"])
    save(params, files["params"])
    save(sig, files["sig"])
    save(pub, files["pub"], format="seal")
    save(sec, files["sec"], format="seal+zlib")
    save(enc, files["enc"], format="seal")
    save(out, files["out"], format="seal")   # a device-resident result is downloaded for the file
    prog2, sig2, pub2, sec2, enc2, out2 = (load(files[n]) for n in ("prog", "sig", "pub", "sec", "enc", "out"))
    assert load(files["params"]).prime_bits == params.prime_bits
    res = pub2.execute(prog2, enc2)
    for name in out.names():
        a, b, c = out.get(name), res.get(name), out2.get(name)
        assert a[:4] == b[:4] == c[:4] and np.array_equal(a[4], b[4]) and np.array_equal(a[4], c[4])
    got = sec2.decrypt(res, sig2)
    want = evaluate(compiled, inputs)
    assert np.mean((np.array(got['z']) - np.array(want['z'])) ** 2) < 0.01   # the reference's threshold (tests/common.py:34)
    assert np.allclose(sec.decrypt(out2, sig)['z'], got['z'], atol=1e-9)
