"""EVA's wire format (eva_amd/host/wire.h) against the OFFICIAL protobuf runtime: the message
classes are built in Python from the reference's schema (/root/reference/eva/serialization/
eva.proto, ckks.proto, known_type.proto — restated below as descriptors, field for field), so
  * what save() writes parses with real protobuf into the expected messages, and
  * what real protobuf serialises (as microsoft/EVA would) loads here into an equivalent object.
No protoc is needed; the descriptors are data."""
import numpy as np
import pytest

pb = pytest.importorskip("google.protobuf")
from google.protobuf import any_pb2, descriptor_pb2, descriptor_pool, message_factory  # noqa: E402

from eva import EvaProgram, Input, Output, evaluate, save, load, Op  # noqa: E402
from eva.ckks import CKKSCompiler  # noqa: E402

F = descriptor_pb2.FieldDescriptorProto


def _schema():
    pool = descriptor_pool.DescriptorPool()
    pool.Add(descriptor_pb2.FileDescriptorProto.FromString(any_pb2.DESCRIPTOR.serialized_pb))
    fd = descriptor_pb2.FileDescriptorProto(name="eva_all.proto", package="eva.msg", syntax="proto3",
                                            dependency=["google/protobuf/any.proto"])

    def msg(name, fields, oneofs=()):
        m = fd.message_type.add(name=name)
        for o in oneofs:
            m.oneof_decl.add(name=o)
        for f in fields:
            m.field.add(**f)
        return m
    S, M = F.LABEL_OPTIONAL, F.LABEL_REPEATED
    msg("ConstantValue", [dict(name="size", number=1, type=F.TYPE_UINT32, label=S),
                          dict(name="values", number=2, type=F.TYPE_DOUBLE, label=M),
                          dict(name="sparse_indices", number=3, type=F.TYPE_UINT32, label=M)])
    msg("Attribute", [dict(name="key", number=1, type=F.TYPE_UINT32, label=S),
                      dict(name="uint32", number=2, type=F.TYPE_UINT32, label=S, oneof_index=0),
                      dict(name="int32", number=3, type=F.TYPE_SINT32, label=S, oneof_index=0),
                      dict(name="type", number=4, type=F.TYPE_UINT32, label=S, oneof_index=0),
                      dict(name="constant_value", number=5, type=F.TYPE_MESSAGE, type_name=".eva.msg.ConstantValue", label=S, oneof_index=0)],
        oneofs=["value"])
    msg("Term", [dict(name="op", number=1, type=F.TYPE_UINT32, label=S),
                 dict(name="operands", number=2, type=F.TYPE_UINT64, label=M),
                 dict(name="attributes", number=3, type=F.TYPE_MESSAGE, type_name=".eva.msg.Attribute", label=M)])
    msg("TermName", [dict(name="term", number=1, type=F.TYPE_UINT64, label=S), dict(name="name", number=2, type=F.TYPE_STRING, label=S)])
    msg("Program", [dict(name="ir_version", number=1, type=F.TYPE_UINT32, label=S),
                    dict(name="name", number=2, type=F.TYPE_STRING, label=S),
                    dict(name="vec_size", number=3, type=F.TYPE_UINT32, label=S),
                    dict(name="terms", number=4, type=F.TYPE_MESSAGE, type_name=".eva.msg.Term", label=M),
                    dict(name="inputs", number=5, type=F.TYPE_MESSAGE, type_name=".eva.msg.TermName", label=M),
                    dict(name="outputs", number=6, type=F.TYPE_MESSAGE, type_name=".eva.msg.TermName", label=M)])
    msg("CKKSParameters", [dict(name="prime_bits", number=1, type=F.TYPE_UINT32, label=M),
                           dict(name="rotations", number=2, type=F.TYPE_INT32, label=M),
                           dict(name="poly_modulus_degree", number=3, type=F.TYPE_UINT32, label=S)])
    msg("CKKSEncodingInfo", [dict(name="input_type", number=1, type=F.TYPE_INT32, label=S),
                             dict(name="scale", number=2, type=F.TYPE_INT32, label=S),
                             dict(name="level", number=3, type=F.TYPE_INT32, label=S)])
    sig = msg("CKKSSignature", [dict(name="vec_size", number=1, type=F.TYPE_INT32, label=S),
                                dict(name="inputs", number=2, type=F.TYPE_MESSAGE, type_name=".eva.msg.CKKSSignature.InputsEntry", label=M)])
    entry = sig.nested_type.add(name="InputsEntry")
    entry.options.map_entry = True
    entry.field.add(name="key", number=1, type=F.TYPE_STRING, label=S)
    entry.field.add(name="value", number=2, type=F.TYPE_MESSAGE, type_name=".eva.msg.CKKSEncodingInfo", label=S)
    msg("KnownType", [dict(name="contents", number=1, type=F.TYPE_MESSAGE, type_name=".google.protobuf.Any", label=S),
                      dict(name="creator", number=2, type=F.TYPE_STRING, label=S)])
    pool.Add(fd)
    get = lambda n: message_factory.GetMessageClass(pool.FindMessageTypeByName("eva.msg." + n))
    return {n: get(n) for n in ("Program", "CKKSParameters", "CKKSSignature", "KnownType", "Term", "Attribute", "ConstantValue")}


@pytest.fixture(scope="module")
def schema():
    return _schema()


def _program():
    prog = EvaProgram('wire', vec_size=8)
    with prog:
        x, y = Input('x'), Input('y', is_encrypted=False)
        z = (x << 2) * y + [1.0, -2.5, 3.0, 0.5, 0, 0, 0, 7] - (x >> 1) * 0.25 + x * 0
        Output('z', z * z)
    prog.set_input_scales(30)
    prog.set_output_ranges(20)
    return prog


def _open(schema, path, inner):
    kt = schema["KnownType"]()
    kt.ParseFromString(open(path, "rb").read())
    assert kt.contents.type_url == "type.googleapis.com/eva.msg." + inner
    m = schema[inner]()
    m.ParseFromString(kt.contents.value)
    return kt, m


def test_saved_files_parse_with_real_protobuf(schema, tmp_path):
    prog = _program()
    compiled, params, sig = CKKSCompiler(config={'warn_vec_size': 'false'}).compile(prog)
    for obj, name, inner in ((compiled, "p", "Program"), (params, "q", "CKKSParameters"), (sig, "s", "CKKSSignature")):
        save(obj, str(tmp_path / name))
    kt, m = _open(schema, str(tmp_path / "p"), "Program")
    assert m.ir_version == 2 and m.name == "wire" and m.vec_size == 8
    dump = compiled._dump()
    assert len(m.terms) == len(dump)
    for t, d in zip(m.terms, dump):
        assert t.op == int(d["op"]) and len(t.operands) == len(d["operands"])
        attrs = {a.key: a for a in t.attributes}
        if "rotation" in d: assert attrs[2].int32 == d["rotation"]
        if "rescale_divisor" in d: assert attrs[1].uint32 == d["rescale_divisor"]
        if "encode_scale" in d: assert attrs[6].uint32 == d["encode_scale"]
        if "encode_level" in d: assert attrs[7].uint32 == d["encode_level"]
        if "type" in d: assert attrs[4].type == int(d["type"])
        if "constant" in d:
            cv = attrs[3].constant_value
            assert cv.size == 8 and (list(cv.values) == list(d["constant"]) or (not cv.values and not any(d["constant"])))
        assert [a.key for a in t.attributes] == sorted(attrs)   # ascending keys, as AttributeList keeps them
    assert {n.name for n in m.inputs} == {"x", "y"} and {n.name for n in m.outputs} == {"z"}
    # re-serialising with real protobuf reproduces our bytes (same field order, packing and defaults)
    assert m.SerializeToString(deterministic=True) == kt.contents.value
    _, q = _open(schema, str(tmp_path / "q"), "CKKSParameters")
    assert list(q.prime_bits) == list(params.prime_bits) and sorted(q.rotations) == sorted(params.rotations)
    assert q.poly_modulus_degree == params.poly_modulus_degree
    _, s = _open(schema, str(tmp_path / "s"), "CKKSSignature")
    assert s.vec_size == sig.vec_size and set(s.inputs) == set(sig.inputs)
    for k, v in sig.inputs.items():
        assert (s.inputs[k].input_type, s.inputs[k].scale, s.inputs[k].level) == (int(v.input_type), v.scale, v.level)


def test_files_written_by_real_protobuf_load_here(schema, tmp_path):
    """a Program / parameters / signature serialised by the official runtime — negative rotations,
    unpacked and packed repeated fields, a sparse and a zero constant — loads and evaluates"""
    P, T, A = schema["Program"], schema["Term"], schema["Attribute"]
    m = P(ir_version=2, name="from_eva", vec_size=4)
    def term(op, operands=(), **attrs):
        t = m.terms.add(op=op, operands=list(operands))
        for key, (field, val) in attrs.items():
            a = t.attributes.add(key=int(key[1:]))
            if field == "constant_value":
                a.constant_value.CopyFrom(val)
            else:
                setattr(a, field, val)
    CV = schema["ConstantValue"]
    term(1, k4=("type", 2))                                                       # 0: Input x (raw)
    term(3, k3=("constant_value", CV(size=4, values=[2.0, 5.0], sparse_indices=[1, 3])))  # 1: sparse constant [0,2,0,5]
    term(3, k3=("constant_value", CV(size=4)))                                    # 2: the zero constant
    term(14, [0], k2=("int32", -1))                                               # 3: x << -1
    term(13, [3, 1])                                                              # 4: * sparse
    term(11, [4, 2])                                                              # 5: + 0
    term(2, [5])                                                                  # 6: Output
    m.inputs.add(term=0, name="x")
    m.outputs.add(term=6, name="y")
    kt = schema["KnownType"](creator="EVA 1.0.1")
    kt.contents.Pack(m)
    path = str(tmp_path / "ref_program")
    open(path, "wb").write(kt.SerializeToString())
    prog = load(path)
    assert prog.name == "from_eva" and prog.vec_size == 4
    out = evaluate(prog, {"x": [1.0, 2.0, 3.0, 4.0]})
    assert out["y"] == [0.0, 2.0, 0.0, 15.0]   # (x rotated left by -1 = [4,1,2,3]) * [0,2,0,5]
    Q = schema["CKKSParameters"](prime_bits=[60, 20, 60], rotations=[-3, 1, 512], poly_modulus_degree=8192)
    kt = schema["KnownType"]()
    kt.contents.Pack(Q)
    open(path, "wb").write(kt.SerializeToString())
    q = load(path)
    assert list(q.prime_bits) == [60, 20, 60] and sorted(q.rotations) == [-3, 1, 512] and q.poly_modulus_degree == 8192
    S = schema["CKKSSignature"](vec_size=16)
    S.inputs["a"].input_type, S.inputs["a"].scale, S.inputs["a"].level = 1, 30, 0
    S.inputs["b"].input_type, S.inputs["b"].scale, S.inputs["b"].level = 2, 25, 1
    kt = schema["KnownType"]()
    kt.contents.Pack(S)
    open(path, "wb").write(kt.SerializeToString())
    s = load(path)
    assert s.vec_size == 16 and (int(s.inputs["b"].input_type), s.inputs["b"].scale, s.inputs["b"].level) == (2, 25, 1)


def test_round_trip_and_rejects(tmp_path):
    prog = _program()
    compiled, params, sig = CKKSCompiler(config={'warn_vec_size': 'false'}).compile(prog)
    path = str(tmp_path / "rt")
    inputs = {'x': [i / 8.0 for i in range(8)], 'y': [1.0 - i for i in range(8)]}
    for fmt in ("eva", "native"):
        save(compiled, path, format=fmt)
        back = load(path)
        assert sorted(int(d["op"]) for d in back._dump()) == sorted(int(d["op"]) for d in compiled._dump())
        assert evaluate(back, inputs) == evaluate(compiled, inputs)
    raw = open(path, "rb").read()
    save(compiled, path, format="eva")
    raw = open(path, "rb").read()
    open(path, "wb").write(raw[:len(raw) - 7])
    with pytest.raises(RuntimeError, match="parse|truncated"):
        load(path)
    with pytest.raises(ValueError):
        save(compiled, path, format="json")


def test_untrusted_sizes_and_signatures_are_bounded_before_use(schema, tmp_path):
    """r2 advisor findings: a ConstantValue's `size` comes from the file and used to size an allocation before it
    was compared with the program's vec_size (a 30-byte file asking for 32 GiB); signatures / parameters were
    loaded without range checks (vec_size 0 divided by zero in encrypt)."""
    P, CV = schema["Program"], schema["ConstantValue"]

    def program_with(cv, vec_size=8):
        m = P(ir_version=2, name="hostile", vec_size=vec_size)
        t = m.terms.add(op=3)
        a = t.attributes.add(key=3)
        a.constant_value.CopyFrom(cv)
        o = m.terms.add(op=2, operands=[0])
        m.outputs.add(term=1, name="y")
        kt = schema["KnownType"](creator="x")
        kt.contents.Pack(m)
        path = str(tmp_path / "hostile")
        open(path, "wb").write(kt.SerializeToString())
        return path
    # sparse constant expanding to 2^32 - 1 doubles: rejected by the size check, nothing allocated
    with pytest.raises(RuntimeError, match="does not fit the vector size"):
        load(program_with(CV(size=2 ** 32 - 1, values=[1.0], sparse_indices=[0])))
    with pytest.raises(RuntimeError, match="does not fit the vector size"):
        load(program_with(CV(size=3, values=[1.0, 2.0, 3.0])))            # 3 does not divide 8
    with pytest.raises(RuntimeError, match="does not divide its size"):
        load(program_with(CV(size=8, values=[1.0, 2.0, 3.0])))            # dense values must tile `size`
    assert load(program_with(CV(size=4, values=[1.0, 2.0]))).vec_size == 8  # the legal shapes still load
    path = str(tmp_path / "sig")
    for bad in (dict(vec_size=0), dict(vec_size=12), dict(vec_size=-8)):
        S = schema["CKKSSignature"](**bad)
        kt = schema["KnownType"]()
        kt.contents.Pack(S)
        open(path, "wb").write(kt.SerializeToString())
        with pytest.raises(RuntimeError, match="power of two"):
            load(path)
    for field, val in (("input_type", 7), ("scale", -1), ("level", -2)):
        S = schema["CKKSSignature"](vec_size=16)
        S.inputs["a"].input_type, S.inputs["a"].scale, S.inputs["a"].level = 1, 30, 0
        setattr(S.inputs["a"], field, val)
        kt = schema["KnownType"]()
        kt.contents.Pack(S)
        open(path, "wb").write(kt.SerializeToString())
        with pytest.raises(RuntimeError, match="invalid encoding info"):
            load(path)
    for kw in (dict(prime_bits=[60, 61], poly_modulus_degree=8192), dict(prime_bits=[60, 60], poly_modulus_degree=3000),
               dict(prime_bits=[], poly_modulus_degree=8192), dict(prime_bits=[60] * 70, poly_modulus_degree=8192)):
        Q = schema["CKKSParameters"](**kw)
        kt = schema["KnownType"]()
        kt.contents.Pack(Q)
        open(path, "wb").write(kt.SerializeToString())
        with pytest.raises(RuntimeError, match="parse message"):
            load(path)
