// serialization.h — save/load of the six object kinds EVA persists (Program, CKKSParameters,
// CKKSSignature, valuation, public and secret context): the same Python API as
// /root/reference/eva/serialization/save_load.h:36-62 / known_type.cpp:13-25, exercised like
// tests/features.py:154-217.  The container is this repo's own little-endian tagged format, not
// protobuf + SEAL's binary blobs (neither library exists here), so files are not interchangeable
// with microsoft/EVA's.
#pragma once
#include <cstring>
#include <fstream>
#include <variant>

#include "executor.h"

namespace evahost {

constexpr uint32_t FORMAT_MAGIC = 0x48415645; // "EVAH"
constexpr uint32_t FORMAT_VERSION = 1;
enum class Kind : uint32_t { Program = 1, Parameters = 2, Signature = 3, Valuation = 4, Public = 5, Secret = 6 };

class Writer {
public:
  std::vector<char> buf;
  template <class T> void pod(const T &v) {
    const char *p = reinterpret_cast<const char *>(&v);
    buf.insert(buf.end(), p, p + sizeof(T));
  }
  void str(const std::string &s) { pod<uint64_t>(s.size()); buf.insert(buf.end(), s.begin(), s.end()); }
  template <class T, class A> void vec(const std::vector<T, A> &v) {
    pod<uint64_t>(v.size());
    const char *p = reinterpret_cast<const char *>(v.data());
    buf.insert(buf.end(), p, p + v.size() * sizeof(T));
  }
};
class Reader {
public:
  const std::vector<char> &buf;
  size_t pos = 0;
  explicit Reader(const std::vector<char> &b) : buf(b) {}
  // overflow-safe: n is compared with what is left, never added to pos
  void need(uint64_t n) { if (n > buf.size() - pos) throw std::runtime_error("Could not parse message: truncated file"); }
  template <class T> T pod() {
    need(sizeof(T));
    T v;
    std::memcpy(&v, buf.data() + pos, sizeof(T));
    pos += sizeof(T);
    return v;
  }
  std::string str() {
    uint64_t n = pod<uint64_t>();
    need(n);
    std::string s(buf.data() + pos, buf.data() + pos + n);
    pos += n;
    return s;
  }
  template <class T> std::vector<T> vec() {
    uint64_t n = pod<uint64_t>();
    if (n > (buf.size() - pos) / sizeof(T)) throw std::runtime_error("Could not parse message: truncated file");
    need(n * sizeof(T));
    std::vector<T> v(n);
    std::memcpy(v.data(), buf.data() + pos, n * sizeof(T));
    pos += n * sizeof(T);
    return v;
  }
};

// ---- Program: live terms in topological order, ids re-densified
inline void write(Writer &w, const Program &p) {
  w.str(p.name());
  w.pod<uint32_t>(p.vec_size());
  auto order = p.topo_order();
  std::vector<TermId> newid(p.size(), NO_TERM);
  for (size_t i = 0; i < order.size(); i++) newid[order[i]] = (TermId)i;
  w.pod<uint64_t>(order.size());
  for (TermId t : order) {
    const Term &x = p.at(t);
    w.pod<int32_t>((int32_t)x.op);
    w.pod<uint32_t>((uint32_t)x.operands.size());
    for (TermId o : x.operands) w.pod<uint32_t>(newid[o]);
    uint32_t flags = (x.has_rescale_divisor ? 1 : 0) | (x.has_rotation ? 2 : 0) | (x.has_type ? 4 : 0) | (x.has_range ? 8 : 0) |
                     (x.has_encode_scale ? 16 : 0) | (x.has_encode_level ? 32 : 0) | (x.constant ? 64 : 0);
    w.pod(flags);
    w.pod(x.rescale_divisor); w.pod(x.rotation); w.pod<int32_t>((int32_t)x.type_attr);
    w.pod(x.range); w.pod(x.encode_scale); w.pod(x.encode_level);
    if (x.constant) w.vec(x.constant->values);
  }
  w.pod<uint64_t>(p.inputs().size());
  for (auto &kv : p.inputs()) { w.str(kv.first); w.pod<uint32_t>(newid[kv.second]); }
  w.pod<uint64_t>(p.outputs().size());
  for (auto &kv : p.outputs()) { w.str(kv.first); w.pod<uint32_t>(newid[kv.second]); }
}
inline std::unique_ptr<Program> read_program(Reader &r) {
  std::string name = r.str();
  uint32_t vs = r.pod<uint32_t>();
  auto p = std::make_unique<Program>(name, vs);
  uint64_t n = r.pod<uint64_t>();
  for (uint64_t i = 0; i < n; i++) {
    Op op = (Op)r.pod<int32_t>();
    uint32_t no = r.pod<uint32_t>();
    // files come from untrusted clients: the op code and its operand count are checked here, so
    // evaluate() / execute() never index a missing operand (the reference gets this from protobuf
    // parsing plus Program's own checks, eva/serialization/eva_serialization.cpp:146-289)
    uint32_t want = 0;
    switch (op) {
    case Op::Input: case Op::Constant: want = 0; break;
    case Op::Add: case Op::Sub: case Op::Mul: want = 2; break;
    case Op::Output: case Op::Negate: case Op::RotateLeftConst: case Op::RotateRightConst: case Op::Relinearize:
    case Op::ModSwitch: case Op::Rescale: case Op::Encode: want = 1; break;
    default: throw std::runtime_error("Could not parse message: unknown op code");
    }
    if (no != want) throw std::runtime_error("Could not parse message: wrong operand count for op");
    std::vector<TermId> ops;
    for (uint32_t j = 0; j < no; j++) {
      TermId o = r.pod<uint32_t>();
      if (o >= i) throw std::runtime_error("Could not parse message: operand out of order");
      ops.push_back(o);
    }
    TermId t = p->make_term(op, ops);
    Term &x = p->at(t);
    uint32_t flags = r.pod<uint32_t>();
    x.rescale_divisor = r.pod<uint32_t>(); x.rotation = r.pod<int32_t>(); x.type_attr = (Type)r.pod<int32_t>();
    x.range = r.pod<uint32_t>(); x.encode_scale = r.pod<uint32_t>(); x.encode_level = r.pod<uint32_t>();
    x.has_rescale_divisor = flags & 1; x.has_rotation = flags & 2; x.has_type = flags & 4; x.has_range = flags & 8;
    x.has_encode_scale = flags & 16; x.has_encode_level = flags & 32;
    if (flags & 64) x.constant = std::make_shared<ConstantValue>(ConstantValue{r.vec<double>()});
    if (op == Op::Constant && (!x.constant || x.constant->values.empty() || x.constant->values.size() > vs || vs % x.constant->values.size()))
      throw std::runtime_error("Could not parse message: constant without a valid value");
    if ((op == Op::RotateLeftConst || op == Op::RotateRightConst) && !x.has_rotation)
      throw std::runtime_error("Could not parse message: rotation without a step count");
  }
  auto term_of = [&](Op want_op) {
    uint32_t t = r.pod<uint32_t>();
    if (t >= n || p->at(t).op != want_op) throw std::runtime_error("Could not parse message: input / output binding names the wrong term");
    return t;
  };
  uint64_t ni = r.pod<uint64_t>();
  for (uint64_t i = 0; i < ni; i++) { std::string s = r.str(); p->bind_input(s, term_of(Op::Input)); }
  uint64_t nout = r.pod<uint64_t>();
  for (uint64_t i = 0; i < nout; i++) { std::string s = r.str(); p->bind_output(s, term_of(Op::Output)); }
  return p;
}

inline void write(Writer &w, const CKKSParameters &p) {
  w.vec(p.prime_bits);
  std::vector<int32_t> rot(p.rotations.begin(), p.rotations.end());
  w.vec(rot);
  w.pod(p.poly_modulus_degree);
}
inline CKKSParameters read_parameters(Reader &r) {
  CKKSParameters p;
  p.prime_bits = r.vec<uint32_t>();
  auto rot = r.vec<int32_t>();
  p.rotations.insert(rot.begin(), rot.end());
  p.poly_modulus_degree = r.pod<uint32_t>();
  if (p.poly_modulus_degree == 0 || (p.poly_modulus_degree & (p.poly_modulus_degree - 1)) || p.poly_modulus_degree > (1u << 17))
    throw std::runtime_error("Could not parse message: poly_modulus_degree must be a power of two up to 131072");
  if (p.prime_bits.empty() || p.prime_bits.size() > 62) throw std::runtime_error("Could not parse message: invalid prime count");
  for (uint32_t b : p.prime_bits)
    if (b < 2 || b > 60) throw std::runtime_error("Could not parse message: prime bit sizes must be 2..60");
  return p;
}
inline void write(Writer &w, const CKKSSignature &s) {
  w.pod<int32_t>(s.vec_size);
  w.pod<uint64_t>(s.inputs.size());
  for (auto &kv : s.inputs) { w.str(kv.first); w.pod<int32_t>((int32_t)kv.second.input_type); w.pod<int32_t>(kv.second.scale); w.pod<int32_t>(kv.second.level); }
}
inline CKKSSignature read_signature(Reader &r) {
  CKKSSignature s;
  s.vec_size = r.pod<int32_t>();
  uint64_t n = r.pod<uint64_t>();
  for (uint64_t i = 0; i < n; i++) {
    std::string name = r.str();
    Type t = (Type)r.pod<int32_t>();
    int sc = r.pod<int32_t>(), lv = r.pod<int32_t>();
    if ((int)t < 0 || (int)t > 3 || sc < 0 || lv < 0) throw std::runtime_error("Could not parse message: invalid encoding info for input " + name);
    s.inputs.emplace(name, CKKSEncodingInfo{t, sc, lv});
  }
  if (s.vec_size <= 0 || (s.vec_size & (s.vec_size - 1)))
    throw std::runtime_error("Could not parse message: signature vector size must be a positive power of two");
  return s;
}
inline void write(Writer &w, const HipValuation &v) {
  w.pod<uint64_t>(v.values.size());
  for (auto &kv : v.values) {
    w.str(kv.first);
    if (auto *c = std::get_if<HostCipher>(&kv.second)) {
      w.pod<uint32_t>(1); w.pod(c->size); w.pod(c->limbs); w.pod(c->scale);
      w.vec(words(*c)); // a device-resident value is downloaded for the file
    }
    else if (auto *p = std::get_if<HostPlain>(&kv.second)) { w.pod<uint32_t>(2); w.pod(p->limbs); w.pod(p->scale); w.vec(p->data); }
    else { w.pod<uint32_t>(3); w.vec(std::get<std::vector<double>>(kv.second)); }
  }
}
inline HipValuation read_valuation(Reader &r) {
  HipValuation v;
  uint64_t n = r.pod<uint64_t>();
  for (uint64_t i = 0; i < n; i++) {
    std::string name = r.str();
    uint32_t kind = r.pod<uint32_t>();
    // a valuation file carries no context; shapes must at least be self-consistent here (power-of-two
    // degree) and are checked against the context again before every upload (check_shape)
    auto degree_ok = [](size_t words, size_t polys) {
      if (!polys || words % polys) return false;
      size_t n = words / polys;
      return n >= 1024 && n <= 131072 && !(n & (n - 1));
    };
    if (kind == 1) {
      HostCipher c; c.size = r.pod<uint32_t>(); c.limbs = r.pod<uint32_t>(); c.scale = r.pod<double>();
      { auto w = r.vec<u64>(); c.data.assign(w.begin(), w.end()); }
      if (c.size < 1 || c.size > 3 || c.limbs < 1 || c.limbs > 61 || !degree_ok(c.data.size(), (size_t)c.size * c.limbs))
        throw std::runtime_error("Could not parse message: ciphertext shape does not match its data");
      v.values[name] = std::move(c);
    } else if (kind == 2) {
      HostPlain p; p.limbs = r.pod<uint32_t>(); p.scale = r.pod<double>(); p.data = r.vec<u64>();
      if (p.limbs < 1 || p.limbs > 61 || !degree_ok(p.data.size(), p.limbs))
        throw std::runtime_error("Could not parse message: plaintext shape does not match its data");
      v.values[name] = std::move(p);
    }
    else if (kind == 3) v.values[name] = r.vec<double>();
    else throw std::runtime_error("Could not parse message: unknown value kind");
  }
  return v;
}
inline void write_ctx(Writer &w, const HostContext &h) { w.pod(h.N); w.vec(h.primes); }
inline std::shared_ptr<HostContext> read_ctx(Reader &r) {
  uint32_t N = r.pod<uint32_t>();
  auto primes = r.vec<u64>();
  if (N < 1024 || N > 131072 || (N & (N - 1)) || primes.size() < 2 || primes.size() > 62)
    throw std::runtime_error("Could not parse message: invalid encryption parameters");
  for (u64 q : primes)
    if (q < 2 || q >= ((u64)1 << 60) || (q - 1) % (2ull * N) || !evah::is_prime(q))
      throw std::runtime_error("Could not parse message: invalid coefficient modulus");
  return std::make_shared<HostContext>(N, primes);
}
inline void write(Writer &w, const HipPublic &p) {
  write_ctx(w, *p.host);
  w.vec(p.pk.data);
  w.pod(p.relin.n_digits); w.vec(p.relin.data);
  w.pod<uint64_t>(p.galois.size());
  for (auto &kv : p.galois) { w.pod(kv.first); w.pod(kv.second.n_digits); w.vec(kv.second.data); }
}
// every word of a key-level object ([...][k][N]) must be a canonical residue of its prime: the kernels'
// lazy-reduction bounds assume it, so an out-of-range word would give silently wrong results
template <class Vec> inline void check_residues(const Vec &words, const HostContext &h, const char *what) {
  const size_t N = h.N, rows = words.size() / N;
  for (size_t r = 0; r < rows; r++) {
    const u64 q = h.primes[r % h.k];
    const u64 *w = words.data() + r * N;
    for (size_t j = 0; j < N; j++)
      if (w[j] >= q) throw std::runtime_error(std::string("Could not parse message: ") + what + " holds a word that is not reduced modulo its prime");
  }
}
inline void check_switch_key(const SwitchKey &k, const HostContext &h, const char *what) {
  if (k.n_digits == 0 || k.n_digits > h.k - 1 || k.data.size() != (size_t)k.n_digits * 2 * h.k * h.N)
    throw std::runtime_error(std::string("Could not parse message: ") + what + " has the wrong size for its context");
  check_residues(k.data, h, what);
}
inline std::shared_ptr<HipPublic> read_public(Reader &r) {
  auto p = std::make_shared<HipPublic>();
  p->host = read_ctx(r);
  p->pk.data = r.vec<u64>();
  if (p->pk.data.size() != (size_t)2 * p->host->k * p->host->N) throw std::runtime_error("Could not parse message: public key has the wrong size for its context");
  check_residues(p->pk.data, *p->host, "public key");
  p->relin.n_digits = r.pod<uint32_t>(); p->relin.data = r.vec<u64>();
  check_switch_key(p->relin, *p->host, "relinearization key");
  uint64_t n = r.pod<uint64_t>();
  for (uint64_t i = 0; i < n; i++) {
    uint32_t elt = r.pod<uint32_t>();
    SwitchKey k; k.n_digits = r.pod<uint32_t>(); k.data = r.vec<u64>();
    if (!(elt & 1) || elt >= 2 * p->host->N) throw std::runtime_error("Could not parse message: Galois element is not valid");
    check_switch_key(k, *p->host, "Galois key");
    p->galois.emplace(elt, std::move(k));
  }
  return p;
}
inline void write(Writer &w, const HipSecret &s) {
  write_ctx(w, *s.host);
  w.vec(s.sk.s);
  w.vec(s.sk.s_ntt);
}
inline std::shared_ptr<HipSecret> read_secret(Reader &r) {
  auto s = std::make_shared<HipSecret>();
  s->host = read_ctx(r);
  s->sk.s = r.vec<int8_t>();
  s->sk.s_ntt = r.vec<u64>();
  if (s->sk.s.size() != s->host->N || s->sk.s_ntt.size() != (size_t)s->host->k * s->host->N)
    throw std::runtime_error("Could not parse message: secret key has the wrong size for its context");
  for (int8_t v : s->sk.s)
    if (v < -1 || v > 1) throw std::runtime_error("Could not parse message: secret key coefficients must be ternary");
  check_residues(s->sk.s_ntt, *s->host, "secret key");
  return s;
}

template <class T> void save_to_file(Kind kind, const T &obj, const std::string &path) {
  Writer w;
  w.pod(FORMAT_MAGIC); w.pod(FORMAT_VERSION); w.pod<uint32_t>((uint32_t)kind);
  write(w, obj);
  std::ofstream f(path, std::ios::binary);
  if (!f) throw std::runtime_error("Could not open file " + path);
  f.write(w.buf.data(), (std::streamsize)w.buf.size());
}

} // namespace evahost
#include "wire.h"
#include "seal_format.h"
namespace evahost {

// Program / CKKSParameters / CKKSSignature in the reference's own wire format (protobuf KnownType
// envelope, wire.h): what eva::save writes and eva::load reads.  The default for these three kinds.
inline void save_wire_to_file(const std::string &type, const std::string &payload, const std::string &path) {
  std::ofstream f(path, std::ios::binary);
  if (!f) throw std::runtime_error("Could not open file " + path);
  const std::string buf = wire::envelope(type, payload);
  f.write(buf.data(), (std::streamsize)buf.size());
}

using KnownType = std::variant<std::unique_ptr<Program>, CKKSParameters, CKKSSignature, HipValuation, std::shared_ptr<HipPublic>, std::shared_ptr<HipSecret>>;
inline KnownType load_from_file(const std::string &path) {
  std::ifstream f(path, std::ios::binary);
  if (!f) throw std::runtime_error("Could not open file " + path);
  std::vector<char> buf((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
  Reader r(buf);
  uint32_t magic = 0;
  if (buf.size() >= 4) std::memcpy(&magic, buf.data(), 4);
  if (magic != FORMAT_MAGIC) { // not this repo's container: the reference's protobuf envelope
    auto [type, payload] = wire::open_envelope(std::string(buf.begin(), buf.end()));
    if (type == "Program") return wire::decode_program(wire::In(payload));
    if (type == "CKKSParameters") return wire::decode_parameters(wire::In(payload));
    if (type == "CKKSSignature") return wire::decode_signature(wire::In(payload));
    // the SEAL-object messages (seal.proto): SEAL's own binary object format inside, seal_format.h
    if (type == "SEALValuation") return sealfmt::decode_valuation(wire::In(payload));
    if (type == "SEALPublic") return sealfmt::decode_public(wire::In(payload));
    if (type == "SEALSecret") return sealfmt::decode_secret(wire::In(payload));
    throw std::runtime_error("Unknown inner message type eva.msg." + type);
  }
  if (r.pod<uint32_t>() != FORMAT_MAGIC) throw std::runtime_error("Could not parse message: not an eva_amd file");
  if (r.pod<uint32_t>() != FORMAT_VERSION) throw std::runtime_error("Serialization format version is not compatible");
  switch ((Kind)r.pod<uint32_t>()) {
  case Kind::Program: return read_program(r);
  case Kind::Parameters: return read_parameters(r);
  case Kind::Signature: return read_signature(r);
  case Kind::Valuation: return read_valuation(r);
  case Kind::Public: return read_public(r);
  case Kind::Secret: return read_secret(r);
  }
  throw std::runtime_error("Unknown inner message type");
}

} // namespace evahost
