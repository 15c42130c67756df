// wire.h — EVA's own wire format for Program, CKKSParameters and CKKSSignature: the protobuf
// messages of /root/reference/eva/serialization/{eva,ckks,known_type}.proto inside the KnownType
// envelope (google.protobuf.Any + creator) that eva::save / eva::load write and read
// (/root/reference/eva/serialization/save_load.h:29-62, eva_serialization.cpp:146-289,
// ckks_serialization.cpp:14-88).  Files written here load in microsoft/EVA and the other way round.
// protobuf itself is not available in this image, so the handful of wire rules needed are
// implemented directly: base-128 varints, zigzag for sint32, little-endian fixed64 doubles,
// length-delimited sub-messages, packed repeated scalars (proto3 default), map entries as
// (key = 1, value = 2) messages.  tests/test_wire_format.py checks every byte stream against the
// official Python protobuf runtime built from the same schema.
// The SEAL-object kinds (valuation, public and secret context: seal.proto wraps SEAL's binary
// blobs) are in seal_format.h, which builds on the primitives here.
#pragma once
#include <cstring>
#include <string>

#include "executor.h"

namespace evahost {
namespace wire {

constexpr uint32_t EVA_FORMAT_VERSION = 2; // /root/reference/eva/serialization/eva_format_version.h
// attribute keys: /root/reference/eva/ir/attributes.h:12-19
enum AttrKey : uint32_t { RescaleDivisor = 1, Rotation = 2, ConstantValueAttr = 3, TypeAttr = 4, Range = 5, EncodeAtScale = 6, EncodeAtLevel = 7 };

struct Out {
  std::string b;
  void varint(uint64_t v) {
    while (v >= 0x80) { b.push_back((char)(v | 0x80)); v >>= 7; }
    b.push_back((char)v);
  }
  void tag(uint32_t field, uint32_t wt) { varint(((uint64_t)field << 3) | wt); }
  void u(uint32_t field, uint64_t v, bool always = false) { if (v || always) { tag(field, 0); varint(v); } }
  void i32(uint32_t field, int32_t v, bool always = false) { if (v || always) { tag(field, 0); varint((uint64_t)(int64_t)v); } } // int32: sign-extended
  void s32(uint32_t field, int32_t v, bool always = false) { if (v || always) { tag(field, 0); varint(((uint32_t)v << 1) ^ (uint32_t)(v >> 31)); } }
  void bytes(uint32_t field, const std::string &s, bool always = false) { if (!s.empty() || always) { tag(field, 2); varint(s.size()); b += s; } }
};
struct In {
  const unsigned char *p, *end;
  In(const std::string &s) : p((const unsigned char *)s.data()), end(p + s.size()) {}
  In(const unsigned char *a, const unsigned char *e) : p(a), end(e) {}
  bool done() const { return p >= end; }
  uint64_t varint() {
    uint64_t v = 0;
    for (int shift = 0; shift < 70; shift += 7) {
      if (p >= end) throw std::runtime_error("Could not parse message: truncated varint");
      const unsigned char c = *p++;
      v |= (uint64_t)(c & 0x7f) << shift;
      if (!(c & 0x80)) return v;
    }
    throw std::runtime_error("Could not parse message: varint too long");
  }
  In sub() { // length-delimited payload
    const uint64_t n = varint();
    if (n > (uint64_t)(end - p)) throw std::runtime_error("Could not parse message: truncated field");
    In s(p, p + n);
    p += n;
    return s;
  }
  std::string str() { In s = sub(); return std::string((const char *)s.p, (const char *)s.end); }
  double f64() {
    if (end - p < 8) throw std::runtime_error("Could not parse message: truncated double");
    double d;
    std::memcpy(&d, p, 8);
    p += 8;
    return d;
  }
  void skip(uint32_t wt) {
    if (wt == 0) (void)varint();
    else if (wt == 1) { if (end - p < 8) throw std::runtime_error("Could not parse message: truncated"); p += 8; }
    else if (wt == 2) (void)sub();
    else if (wt == 5) { if (end - p < 4) throw std::runtime_error("Could not parse message: truncated"); p += 4; }
    else throw std::runtime_error("Could not parse message: unsupported wire type");
  }
  // repeated scalar field: packed (wire type 2) or one element per tag
  template <class F> void repeated(uint32_t wt, F &&one) {
    if (wt == 2) { In s = sub(); while (!s.done()) one(s); }
    else one(*this);
  }
};

// ---- eva.msg.Program
inline std::string encode(const Program &p) {
  Out o;
  o.u(1, EVA_FORMAT_VERSION);
  o.bytes(2, p.name());
  o.u(3, p.vec_size());
  auto order = p.topo_order();
  std::vector<TermId> newid(p.size(), NO_TERM);
  for (size_t i = 0; i < order.size(); i++) newid[order[i]] = (TermId)i;
  for (TermId t : order) {
    const Term &x = p.at(t);
    Out tm;
    tm.u(1, (uint32_t)x.op);
    if (!x.operands.empty()) {
      Out ops;
      for (TermId a : x.operands) ops.varint(newid[a]);
      tm.bytes(2, ops.b, true);
    }
    auto attr = [&](uint32_t key, auto &&value) {
      Out a;
      a.u(1, key);
      value(a);
      tm.bytes(3, a.b, true);
    };
    if (x.has_rescale_divisor) attr(RescaleDivisor, [&](Out &a) { a.u(2, x.rescale_divisor, true); });
    if (x.has_rotation) attr(Rotation, [&](Out &a) { a.s32(3, x.rotation, true); });
    if (x.constant) attr(ConstantValueAttr, [&](Out &a) {
      Out c;
      c.u(1, p.vec_size()); // DenseConstantValue(size = vec_size, values): values broadcast over size
      bool zero = true;
      for (double v : x.constant->values) zero = zero && v == 0.0 && !std::signbit(v);
      if (!zero) { // an all-zero constant is the empty `values` list (sparse zero) in the reference too
        Out vals;
        for (double v : x.constant->values) { char raw[8]; std::memcpy(raw, &v, 8); vals.b.append(raw, 8); }
        c.bytes(2, vals.b, true);
      }
      a.bytes(5, c.b, true);
    });
    if (x.has_type) attr(TypeAttr, [&](Out &a) { a.u(4, (uint32_t)x.type_attr, true); });
    if (x.has_range) attr(Range, [&](Out &a) { a.u(2, x.range, true); });
    if (x.has_encode_scale) attr(EncodeAtScale, [&](Out &a) { a.u(2, x.encode_scale, true); });
    if (x.has_encode_level) attr(EncodeAtLevel, [&](Out &a) { a.u(2, x.encode_level, true); });
    o.bytes(4, tm.b, true);
  }
  auto names = [&](uint32_t field, const auto &map) {
    for (auto &kv : map) {
      Out n;
      n.u(1, newid[kv.second]);
      n.bytes(2, kv.first);
      o.bytes(field, n.b, true);
    }
  };
  names(5, p.inputs());
  names(6, p.outputs());
  return o.b;
}

inline std::unique_ptr<Program> decode_program(In in) {
  uint32_t version = 0, vec_size = 0;
  std::string name;
  struct TermMsg { uint32_t op = 0; std::vector<uint64_t> operands; std::vector<std::string> attrs; };
  std::vector<TermMsg> terms;
  std::vector<std::pair<uint64_t, std::string>> ins, outs;
  while (!in.done()) {
    const uint64_t t = in.varint();
    const uint32_t f = (uint32_t)(t >> 3), wt = (uint32_t)(t & 7);
    if (f == 1 && wt == 0) version = (uint32_t)in.varint();
    else if (f == 2 && wt == 2) name = in.str();
    else if (f == 3 && wt == 0) vec_size = (uint32_t)in.varint();
    else if (f == 4 && wt == 2) {
      In tm = in.sub();
      TermMsg m;
      while (!tm.done()) {
        const uint64_t tt = tm.varint();
        const uint32_t ff = (uint32_t)(tt >> 3), ww = (uint32_t)(tt & 7);
        if (ff == 1 && ww == 0) m.op = (uint32_t)tm.varint();
        else if (ff == 2) tm.repeated(ww, [&](In &s) { m.operands.push_back(s.varint()); });
        else if (ff == 3 && ww == 2) m.attrs.push_back(tm.str());
        else tm.skip(ww);
      }
      terms.push_back(std::move(m));
    } else if ((f == 5 || f == 6) && wt == 2) {
      In n = in.sub();
      uint64_t term = 0;
      std::string nm;
      while (!n.done()) {
        const uint64_t tt = n.varint();
        if ((tt >> 3) == 1 && (tt & 7) == 0) term = n.varint();
        else if ((tt >> 3) == 2 && (tt & 7) == 2) nm = n.str();
        else n.skip((uint32_t)(tt & 7));
      }
      (f == 5 ? ins : outs).emplace_back(term, nm);
    } else in.skip(wt);
  }
  if (version != EVA_FORMAT_VERSION) throw std::runtime_error("Serialization format version mismatch");
  auto p = std::make_unique<Program>(name, vec_size);
  for (size_t i = 0; i < terms.size(); i++) {
    const TermMsg &m = terms[i];
    const Op op = (Op)m.op;
    uint32_t want = 0;
    switch (op) {
    case Op::Input: case Op::Constant: want = 0; break;
    case Op::Add: case Op::Sub: case Op::Mul: want = 2; break;
    case Op::Output: case Op::Negate: case Op::RotateLeftConst: case Op::RotateRightConst: case Op::Relinearize:
    case Op::ModSwitch: case Op::Rescale: case Op::Encode: want = 1; break;
    default: throw std::runtime_error("Invalid op encountered");
    }
    if (m.operands.size() != want) throw std::runtime_error("Could not parse message: wrong operand count for op");
    std::vector<TermId> ops;
    for (uint64_t a : m.operands) {
      if (a >= i) throw std::runtime_error("Could not parse message: operand out of order");
      ops.push_back((TermId)a);
    }
    const TermId t = p->make_term(op, ops);
    Term &x = p->at(t);
    for (const std::string &raw : m.attrs) {
      In a(raw);
      uint32_t key = 0;
      while (!a.done()) {
        const uint64_t tt = a.varint();
        const uint32_t ff = (uint32_t)(tt >> 3), ww = (uint32_t)(tt & 7);
        if (ff == 1 && ww == 0) key = (uint32_t)a.varint();
        else if (ff == 2 && ww == 0) { // uint32 value: which attribute it is follows from the key
          const uint32_t v = (uint32_t)a.varint();
          if (key == RescaleDivisor) { x.has_rescale_divisor = true; x.rescale_divisor = v; }
          else if (key == Range) { x.has_range = true; x.range = v; }
          else if (key == EncodeAtScale) { x.has_encode_scale = true; x.encode_scale = v; }
          else if (key == EncodeAtLevel) { x.has_encode_level = true; x.encode_level = v; }
          else throw std::runtime_error("Invalid attribute encountered");
        } else if (ff == 3 && ww == 0) {
          const uint64_t z = a.varint();
          if (key != Rotation) throw std::runtime_error("Invalid attribute encountered");
          x.has_rotation = true;
          x.rotation = (int32_t)((z >> 1) ^ (~(z & 1) + 1));
        } else if (ff == 4 && ww == 0) {
          const uint32_t v = (uint32_t)a.varint();
          if (key != TypeAttr || v > 3) throw std::runtime_error("Invalid attribute encountered");
          x.has_type = true;
          x.type_attr = (Type)v;
        } else if (ff == 5 && ww == 2) {
          if (key != ConstantValueAttr) throw std::runtime_error("Invalid attribute encountered");
          In c = a.sub();
          uint32_t size = 0;
          std::vector<double> values;
          std::vector<uint32_t> sparse;
          while (!c.done()) {
            const uint64_t ct = c.varint();
            const uint32_t cf = (uint32_t)(ct >> 3), cw = (uint32_t)(ct & 7);
            if (cf == 1 && cw == 0) size = (uint32_t)c.varint();
            else if (cf == 2) c.repeated(cw == 1 ? 1u : cw, [&](In &s) { values.push_back(s.f64()); });
            else if (cf == 3) c.repeated(cw, [&](In &s) { sparse.push_back((uint32_t)s.varint()); });
            else c.skip(cw);
          }
          if (size == 0) throw std::runtime_error("Constant must have non-zero size");
          // `size` comes from the file: bound it by the program's vector size BEFORE anything is sized by
          // it (a sparse constant expands to `size` doubles), and hold dense values to the rule the
          // reference's validateSlots applies — the value count divides `size`, `size` divides vec_size
          if (size > vec_size || vec_size % size) throw std::runtime_error("Could not parse message: constant does not fit the vector size");
          if (!values.empty() && sparse.empty() && (values.size() > size || size % values.size()))
            throw std::runtime_error("Could not parse message: constant value count does not divide its size");
          std::vector<double> dense;
          if (values.empty()) dense.assign(1, 0.0);                 // the zero constant
          else if (sparse.empty()) dense = std::move(values);       // dense, broadcast over `size`
          else {                                                    // sparse: expanded to `size` values
            if (sparse.size() != values.size()) throw std::runtime_error("Values and sparse indices count mismatch");
            dense.assign(size, 0.0);
            for (size_t j = 0; j < sparse.size(); j++) {
              if (sparse[j] >= size) throw std::runtime_error("Could not parse message: sparse index out of range");
              dense[sparse[j]] = values[j];
            }
          }
          if (dense.size() > vec_size || vec_size % dense.size()) throw std::runtime_error("Could not parse message: constant does not fit the vector size");
          x.constant = std::make_shared<ConstantValue>(ConstantValue{std::move(dense)});
        } else a.skip(ww);
      }
    }
    if (op == Op::Constant && !x.constant) throw std::runtime_error("Could not parse message: constant without a value");
  }
  auto bind = [&](const std::vector<std::pair<uint64_t, std::string>> &v, Op want, bool input) {
    for (auto &e : v) {
      if (e.first >= terms.size() || p->at((TermId)e.first).op != want)
        throw std::runtime_error("Could not parse message: input / output binding names the wrong term");
      if (input) p->bind_input(e.second, (TermId)e.first);
      else p->bind_output(e.second, (TermId)e.first);
    }
  };
  bind(ins, Op::Input, true);
  bind(outs, Op::Output, false);
  return p;
}

// ---- eva.msg.CKKSParameters / CKKSSignature
inline std::string encode(const CKKSParameters &p) {
  Out o;
  if (!p.prime_bits.empty()) { Out v; for (uint32_t b : p.prime_bits) v.varint(b); o.bytes(1, v.b, true); }
  if (!p.rotations.empty()) { Out v; for (int r : p.rotations) v.varint((uint64_t)(int64_t)r); o.bytes(2, v.b, true); }
  o.u(3, p.poly_modulus_degree);
  return o.b;
}
inline CKKSParameters decode_parameters(In in) {
  CKKSParameters p;
  while (!in.done()) {
    const uint64_t t = in.varint();
    const uint32_t f = (uint32_t)(t >> 3), wt = (uint32_t)(t & 7);
    if (f == 1) in.repeated(wt, [&](In &s) { p.prime_bits.push_back((uint32_t)s.varint()); });
    else if (f == 2) in.repeated(wt, [&](In &s) { p.rotations.insert((int)(int32_t)(uint32_t)s.varint()); });
    else if (f == 3 && wt == 0) p.poly_modulus_degree = (uint32_t)in.varint();
    else in.skip(wt);
  }
  if (p.poly_modulus_degree == 0 || (p.poly_modulus_degree & (p.poly_modulus_degree - 1)) || p.poly_modulus_degree > (1u << 17))
    throw std::runtime_error("Could not parse message: poly_modulus_degree must be a power of two up to 131072");
  if (p.prime_bits.empty() || p.prime_bits.size() > 62) throw std::runtime_error("Could not parse message: invalid prime count");
  for (uint32_t b : p.prime_bits)
    if (b < 2 || b > 60) throw std::runtime_error("Could not parse message: prime bit sizes must be 2..60");
  return p;
}
inline std::string encode(const CKKSSignature &s) {
  Out o;
  o.i32(1, s.vec_size);
  for (auto &kv : s.inputs) {
    Out info;
    info.i32(1, (int32_t)kv.second.input_type);
    info.i32(2, kv.second.scale);
    info.i32(3, kv.second.level);
    Out entry;
    entry.bytes(1, kv.first);
    entry.bytes(2, info.b, true);
    o.bytes(2, entry.b, true);
  }
  return o.b;
}
inline CKKSSignature decode_signature(In in) {
  CKKSSignature s;
  while (!in.done()) {
    const uint64_t t = in.varint();
    const uint32_t f = (uint32_t)(t >> 3), wt = (uint32_t)(t & 7);
    if (f == 1 && wt == 0) s.vec_size = (int32_t)(uint32_t)in.varint();
    else if (f == 2 && wt == 2) {
      In e = in.sub();
      std::string key;
      CKKSEncodingInfo info{Type::Undef, 0, 0};
      while (!e.done()) {
        const uint64_t et = e.varint();
        if ((et >> 3) == 1 && (et & 7) == 2) key = e.str();
        else if ((et >> 3) == 2 && (et & 7) == 2) {
          In m = e.sub();
          while (!m.done()) {
            const uint64_t mt = m.varint();
            const uint32_t mf = (uint32_t)(mt >> 3);
            if ((mt & 7) != 0) { m.skip((uint32_t)(mt & 7)); continue; }
            const int32_t v = (int32_t)(uint32_t)m.varint();
            if (mf == 1) info.input_type = (Type)v;
            else if (mf == 2) info.scale = v;
            else if (mf == 3) info.level = v;
          }
        } else e.skip((uint32_t)(et & 7));
      }
      // a signature is loaded from untrusted files too: the enum and the two counts are range-checked here
      if ((int)info.input_type < 0 || (int)info.input_type > 3 || info.scale < 0 || info.level < 0)
        throw std::runtime_error("Could not parse message: invalid encoding info for input " + key);
      s.inputs.emplace(key, info);
    } else in.skip(wt);
  }
  if (s.vec_size <= 0 || (s.vec_size & (s.vec_size - 1)))
    throw std::runtime_error("Could not parse message: signature vector size must be a positive power of two");
  return s;
}

// ---- eva.msg.KnownType { google.protobuf.Any contents = 1; string creator = 2; }
inline std::string envelope(const std::string &type, const std::string &payload) {
  Out any;
  any.bytes(1, "type.googleapis.com/eva.msg." + type);
  any.bytes(2, payload);
  Out o;
  o.bytes(1, any.b, true);
  o.bytes(2, "eva_amd (MI355X backend), EVA wire format 2");
  return o.b;
}
// -> (message type name, payload); throws if the buffer is not a KnownType envelope
inline std::pair<std::string, std::string> open_envelope(const std::string &buf) {
  In in(buf);
  std::string url, payload;
  bool have = false;
  while (!in.done()) {
    const uint64_t t = in.varint();
    if ((t >> 3) == 1 && (t & 7) == 2) {
      In any = in.sub();
      have = true;
      while (!any.done()) {
        const uint64_t at = any.varint();
        if ((at >> 3) == 1 && (at & 7) == 2) url = any.str();
        else if ((at >> 3) == 2 && (at & 7) == 2) payload = any.str();
        else any.skip((uint32_t)(at & 7));
      }
    } else in.skip((uint32_t)(t & 7));
  }
  const std::string prefix = "eva.msg.";
  const size_t at = url.rfind('/');
  std::string type = at == std::string::npos ? url : url.substr(at + 1);
  if (!have || type.compare(0, prefix.size(), prefix) != 0) throw std::runtime_error("Could not parse message");
  return {type.substr(prefix.size()), payload};
}

} // namespace wire
} // namespace evahost
