// seal_format.h — the SEAL-object messages of the reference's wire format:
// eva.msg.SEALValuation / SEALPublic / SEALSecret (/root/reference/eva/serialization/seal.proto:10-44) as
// written and read by /root/reference/eva/serialization/seal_serialization.cpp:46-229.  Every SEALObject.data is
// the byte stream of the SEAL object's own save() (seal_serialization.cpp:46-66: obj.save(buffer, size,
// compr_mode_default)), so this file restates Microsoft SEAL's binary object format.
//
// PROVENANCE — parity unpinned.  SEAL (microsoft/SEAL v3.6.x, the version the reference pins: CMakeLists.txt:24)
// is not in this image; the layout below is restated from SEAL 3.6's serialization.h / *.cpp save_members:
//   SEALHeader (16 bytes)   u16 magic 0xA15E | u8 header_size 0x10 | u8 version_major | u8 version_minor |
//                           u8 compr_mode (0 none, 1 zlib, 2 zstd) | u16 reserved 0 | u64 size (header included)
//   after the header        the object's members; compressed as ONE zlib / zstd stream when compr_mode != 0
//   Modulus                 u64 value
//   EncryptionParameters    u8 scheme (2 = CKKS) | u64 poly_modulus_degree | u64 coeff_modulus_size |
//                           coeff_modulus_size x Modulus object (own header, uncompressed) | plain_modulus object
//   DynArray<u64>           u64 size | size x u64
//   Plaintext               parms_id (4 x u64) | u64 coeff_count | f64 scale | DynArray object
//   Ciphertext              parms_id | u8 is_ntt_form | u64 size | u64 poly_modulus_degree | u64 coeff_modulus_size |
//                           f64 scale | (SEAL 4.x only: u64 correction_factor) | DynArray object
//   PublicKey / SecretKey   exactly the bytes of their Ciphertext / Plaintext member at the key level — SEAL >= 3.5:
//                           PublicKey::save is pk_.save(stream, compr_mode), SecretKey::save is sk_.save(...): ONE header
//                           (r03 of this file wrapped them in a second header, as SEAL <= 3.4 did; the reader still
//                           accepts that nesting, the writer no longer produces it)
//   KSwitchKeys             parms_id | u64 dim1 | dim1 x ( u64 dim2 | dim2 x PublicKey object )
//                           RelinKeys: dim1 = 1; GaloisKeys: dim1 = poly_modulus_degree, index (galois_elt - 1) / 2
//   parms_id                BLAKE2b-256 over the u64 words [scheme, degree, q_0 .. q_{n-1}, plain_modulus = 0]
// tests/test_seal_format.py holds an independent Python restatement (struct + hashlib.blake2b + the protobuf
// runtime) that must agree byte for byte, and tools/seal_parity.cpp section 7 loads these files with SEAL itself
// where SEAL is installed.  Until that has run somewhere, interoperability with real SEAL is a claim, not a fact.
//
// Written with compr_mode none (loadable by every SEAL build); zlib (linked) and zstd (libzstd.so.1 through
// dlopen, when present) are understood on input, because SEAL >= 3.6 writes zstd by default.
// Included by serialization.h (after the key checks and wire.h it builds on); not a stand-alone header.
#pragma once
#include <array>
#include <cmath>
#include <dlfcn.h>
#include <map>
#include <zlib.h>

namespace evahost {
namespace sealfmt {

// ------------------------------------------------------------------------------------------- BLAKE2b (RFC 7693)
inline void blake2b(uint8_t *out, size_t outlen, const uint8_t *in, size_t inlen) {
  static const uint64_t IV[8] = {0x6a09e667f3bcc908ULL, 0xbb67ae8584caa73bULL, 0x3c6ef372fe94f82bULL, 0xa54ff53a5f1d36f1ULL,
                                 0x510e527fade682d1ULL, 0x9b05688c2b3e6c1fULL, 0x1f83d9abfb41bd6bULL, 0x5be0cd19137e2179ULL};
  static const uint8_t SIGMA[12][16] = {
      {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3},
      {11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4}, {7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8},
      {9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13}, {2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9},
      {12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11}, {13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10},
      {6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5}, {10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0},
      {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3}};
  if (outlen < 1 || outlen > 64) throw std::invalid_argument("blake2b digest length");
  uint64_t h[8];
  for (int i = 0; i < 8; i++) h[i] = IV[i];
  h[0] ^= 0x01010000ULL ^ (uint64_t)outlen;
  auto rotr = [](uint64_t x, int n) { return (x >> n) | (x << (64 - n)); };
  auto compress = [&](const uint8_t *block, uint64_t t, bool last) {
    uint64_t m[16], v[16];
    std::memcpy(m, block, 128);
    for (int i = 0; i < 8; i++) { v[i] = h[i]; v[i + 8] = IV[i]; }
    v[12] ^= t; // the high counter word stays zero: inputs here are far below 2^64 bytes
    if (last) v[14] = ~v[14];
    auto G = [&](int a, int b, int c, int d, uint64_t x, uint64_t y) {
      v[a] = v[a] + v[b] + x; v[d] = rotr(v[d] ^ v[a], 32);
      v[c] = v[c] + v[d];     v[b] = rotr(v[b] ^ v[c], 24);
      v[a] = v[a] + v[b] + y; v[d] = rotr(v[d] ^ v[a], 16);
      v[c] = v[c] + v[d];     v[b] = rotr(v[b] ^ v[c], 63);
    };
    for (int r = 0; r < 12; r++) {
      const uint8_t *s = SIGMA[r];
      G(0, 4, 8, 12, m[s[0]], m[s[1]]);   G(1, 5, 9, 13, m[s[2]], m[s[3]]);
      G(2, 6, 10, 14, m[s[4]], m[s[5]]);  G(3, 7, 11, 15, m[s[6]], m[s[7]]);
      G(0, 5, 10, 15, m[s[8]], m[s[9]]);  G(1, 6, 11, 12, m[s[10]], m[s[11]]);
      G(2, 7, 8, 13, m[s[12]], m[s[13]]); G(3, 4, 9, 14, m[s[14]], m[s[15]]);
    }
    for (int i = 0; i < 8; i++) h[i] ^= v[i] ^ v[i + 8];
  };
  size_t off = 0;
  while (inlen - off > 128) { // every block but the last
    compress(in + off, (uint64_t)off + 128, false);
    off += 128;
  }
  uint8_t lastb[128] = {0};
  if (inlen > off) std::memcpy(lastb, in + off, inlen - off);
  compress(lastb, (uint64_t)inlen, true);
  std::memcpy(out, h, outlen);
}

using ParmsId = std::array<uint64_t, 4>;
// EncryptionParameters::compute_parms_id of the CKKS parameters (degree N, the first n primes)
inline ParmsId parms_id(uint32_t N, const u64 *primes, uint32_t n) {
  std::vector<uint64_t> w{2 /* scheme_type::ckks */, (uint64_t)N};
  w.insert(w.end(), primes, primes + n);
  w.push_back(0); // plain_modulus: unused by CKKS, one zero word
  ParmsId id;
  blake2b((uint8_t *)id.data(), 32, (const uint8_t *)w.data(), w.size() * 8);
  return id;
}

// ------------------------------------------------------------------------------------------- compression
enum Compr : uint8_t { None = 0, Zlib = 1, Zstd = 2 };
constexpr uint64_t LIMIT_SMALL = (uint64_t)1 << 24; // parameters
constexpr uint64_t LIMIT_KEYS = (uint64_t)1 << 38;  // a Galois key set at N = 2^16 is tens of GB

inline std::vector<uint8_t> zlib_inflate(const uint8_t *src, size_t n, uint64_t limit) {
  std::vector<uint8_t> out;
  z_stream zs{};
  if (inflateInit(&zs) != Z_OK) throw std::runtime_error("Could not parse message: zlib initialisation failed");
  size_t in_off = 0;
  std::vector<uint8_t> chunk(1 << 20);
  int rc = Z_OK;
  try {
    while (rc != Z_STREAM_END) {
      if (zs.avail_in == 0 && in_off < n) {
        const size_t take = std::min<size_t>(n - in_off, (size_t)1 << 30);
        zs.next_in = const_cast<Bytef *>(src + in_off);
        zs.avail_in = (uInt)take;
        in_off += take;
      }
      zs.next_out = chunk.data();
      zs.avail_out = (uInt)chunk.size();
      rc = inflate(&zs, Z_NO_FLUSH);
      if (rc != Z_OK && rc != Z_STREAM_END) throw std::runtime_error("Could not parse message: corrupt zlib stream in a SEAL object");
      const size_t got = chunk.size() - zs.avail_out;
      if (out.size() + got > limit) throw std::runtime_error("Could not parse message: a compressed SEAL object expands beyond its bound");
      out.insert(out.end(), chunk.begin(), chunk.begin() + got);
      if (rc == Z_OK && got == 0 && zs.avail_in == 0 && in_off >= n) throw std::runtime_error("Could not parse message: truncated zlib stream in a SEAL object");
    }
  } catch (...) {
    inflateEnd(&zs);
    throw;
  }
  inflateEnd(&zs);
  return out;
}
inline std::string zlib_deflate(const std::string &src) {
  uLongf cap = compressBound((uLong)src.size());
  std::string out(cap, '\0');
  if (compress2((Bytef *)out.data(), &cap, (const Bytef *)src.data(), (uLong)src.size(), Z_DEFAULT_COMPRESSION) != Z_OK)
    throw std::runtime_error("zlib compression failed");
  out.resize(cap);
  return out;
}

// libzstd has no headers in this image: the handful of (stable, v1.0+) entry points are declared here and bound at run time
struct ZstdLib {
  struct InBuf { const void *src; size_t size, pos; };
  struct OutBuf { void *dst; size_t size, pos; };
  void *lib = nullptr;
  void *(*createDStream)() = nullptr;
  size_t (*initDStream)(void *) = nullptr;
  size_t (*decompressStream)(void *, OutBuf *, InBuf *) = nullptr;
  size_t (*freeDStream)(void *) = nullptr;
  unsigned (*isError)(size_t) = nullptr;
  size_t (*compressBound)(size_t) = nullptr;
  size_t (*compress)(void *, size_t, const void *, size_t, int) = nullptr;
  static const ZstdLib &get() {
    static ZstdLib z = [] {
      ZstdLib l;
      for (const char *name : {"libzstd.so.1", "libzstd.so"}) {
        l.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        if (l.lib) break;
      }
      if (l.lib) {
        auto sym = [&](const char *s) { return dlsym(l.lib, s); };
        l.createDStream = (void *(*)())sym("ZSTD_createDStream");
        l.initDStream = (size_t(*)(void *))sym("ZSTD_initDStream");
        l.decompressStream = (size_t(*)(void *, OutBuf *, InBuf *))sym("ZSTD_decompressStream");
        l.freeDStream = (size_t(*)(void *))sym("ZSTD_freeDStream");
        l.isError = (unsigned (*)(size_t))sym("ZSTD_isError");
        l.compressBound = (size_t(*)(size_t))sym("ZSTD_compressBound");
        l.compress = (size_t(*)(void *, size_t, const void *, size_t, int))sym("ZSTD_compress");
        if (!l.createDStream || !l.initDStream || !l.decompressStream || !l.freeDStream || !l.isError || !l.compressBound || !l.compress) l.lib = nullptr;
      }
      return l;
    }();
    return z;
  }
  bool ok() const { return lib != nullptr; }
};
inline std::vector<uint8_t> zstd_inflate(const uint8_t *src, size_t n, uint64_t limit) {
  const ZstdLib &z = ZstdLib::get();
  if (!z.ok()) throw std::runtime_error("Could not parse message: the SEAL object is Zstandard-compressed and libzstd is not available on this host");
  void *ds = z.createDStream();
  if (!ds || z.isError(z.initDStream(ds))) { if (ds) z.freeDStream(ds); throw std::runtime_error("Could not parse message: zstd initialisation failed"); }
  std::vector<uint8_t> out, chunk(1 << 20);
  ZstdLib::InBuf in{src, n, 0};
  try {
    size_t rc = 1;
    while (in.pos < in.size || rc != 0) {
      ZstdLib::OutBuf ob{chunk.data(), chunk.size(), 0};
      const size_t before = in.pos;
      rc = z.decompressStream(ds, &ob, &in);
      if (z.isError(rc)) throw std::runtime_error("Could not parse message: corrupt zstd stream in a SEAL object");
      if (out.size() + ob.pos > limit) throw std::runtime_error("Could not parse message: a compressed SEAL object expands beyond its bound");
      out.insert(out.end(), chunk.begin(), chunk.begin() + ob.pos);
      if (in.pos == before && ob.pos == 0) {
        if (rc != 0) throw std::runtime_error("Could not parse message: truncated zstd stream in a SEAL object");
        break;
      }
    }
  } catch (...) {
    z.freeDStream(ds);
    throw;
  }
  z.freeDStream(ds);
  return out;
}
inline std::string zstd_deflate(const std::string &src) {
  const ZstdLib &z = ZstdLib::get();
  if (!z.ok()) throw std::runtime_error("zstd compression requested and libzstd is not available on this host");
  std::string out(z.compressBound(src.size()), '\0');
  const size_t got = z.compress(out.data(), out.size(), src.data(), src.size(), 3);
  if (z.isError(got)) throw std::runtime_error("zstd compression failed");
  out.resize(got);
  return out;
}

// ------------------------------------------------------------------------------------------- objects: writing
constexpr uint16_t SEAL_MAGIC = 0xA15E;
constexpr uint8_t SEAL_MAJOR = 3, SEAL_MINOR = 6; // the release the reference pins

template <class T> inline void put(std::string &b, T v) { b.append((const char *)&v, sizeof(T)); }
inline void put_id(std::string &b, const ParmsId &id) { b.append((const char *)id.data(), 32); }

// Serialization::Save: header + members (compressed as one stream when asked)
inline std::string wrap(const std::string &members, Compr compr = None) {
  const std::string body = compr == None ? std::string() : compr == Zlib ? zlib_deflate(members) : zstd_deflate(members);
  const std::string &payload = compr == None ? members : body;
  std::string out;
  out.reserve(16 + payload.size());
  put<uint16_t>(out, SEAL_MAGIC);
  put<uint8_t>(out, 0x10);
  put<uint8_t>(out, SEAL_MAJOR);
  put<uint8_t>(out, SEAL_MINOR);
  put<uint8_t>(out, (uint8_t)compr);
  put<uint16_t>(out, 0);
  put<uint64_t>(out, 16 + (uint64_t)payload.size());
  out += payload;
  return out;
}
inline std::string modulus_obj(u64 value) {
  std::string m;
  put<uint64_t>(m, value);
  return wrap(m);
}
inline std::string parms_obj(const HostContext &h, Compr compr) {
  std::string m;
  put<uint8_t>(m, 2);
  put<uint64_t>(m, h.N);
  put<uint64_t>(m, h.k);
  for (u64 q : h.primes) m += modulus_obj(q);
  m += modulus_obj(0);
  return wrap(m, compr);
}
inline std::string dynarray_obj(const u64 *d, size_t n) {
  std::string m;
  m.reserve(8 + 8 * n);
  put<uint64_t>(m, n);
  m.append((const char *)d, 8 * n);
  return wrap(m);
}
inline std::string ciphertext_obj(const HostContext &h, uint32_t size, uint32_t limbs, double scale, const u64 *data, Compr compr) {
  std::string m;
  put_id(m, parms_id(h.N, h.primes.data(), limbs));
  put<uint8_t>(m, 1); // CKKS ciphertexts are always in NTT form
  put<uint64_t>(m, size);
  put<uint64_t>(m, h.N);
  put<uint64_t>(m, limbs);
  put<double>(m, scale);
  m += dynarray_obj(data, (size_t)size * limbs * h.N);
  return wrap(m, compr);
}
inline std::string plaintext_obj(const HostContext &h, uint32_t limbs, double scale, const u64 *data, Compr compr) {
  std::string m;
  put_id(m, parms_id(h.N, h.primes.data(), limbs));
  put<uint64_t>(m, (uint64_t)limbs * h.N);
  put<double>(m, scale);
  m += dynarray_obj(data, (size_t)limbs * h.N);
  return wrap(m, compr);
}
// PublicKey::save = pk_.save, SecretKey::save = sk_.save (SEAL >= 3.5): the member's own object, nothing around it
inline std::string public_key_obj(const HostContext &h, const u64 *data /* [2][k][N] */, Compr compr) {
  return ciphertext_obj(h, 2, h.k, 1.0, data, compr);
}
inline std::string secret_key_obj(const HostContext &h, const u64 *s_ntt /* [k][N] */, Compr compr) {
  return plaintext_obj(h, h.k, 1.0, s_ntt, compr);
}
// slots: (index into keys_, key) pairs; dim1 = length of keys_
inline std::string kswitch_obj(const HostContext &h, uint64_t dim1, const std::map<uint64_t, const SwitchKey *> &slots, Compr compr) {
  std::string m;
  put_id(m, parms_id(h.N, h.primes.data(), h.k));
  put<uint64_t>(m, dim1);
  auto it = slots.begin();
  for (uint64_t i = 0; i < dim1; i++) {
    if (it != slots.end() && it->first == i) {
      const SwitchKey &k = *it->second;
      put<uint64_t>(m, k.n_digits);
      const size_t each = (size_t)2 * h.k * h.N;
      for (uint32_t j = 0; j < k.n_digits; j++) m += public_key_obj(h, k.data.data() + j * each, None);
      ++it;
    } else {
      put<uint64_t>(m, 0);
    }
  }
  return wrap(m, compr);
}

// ------------------------------------------------------------------------------------------- objects: reading
struct Cur {
  const uint8_t *p, *end;
  void need(uint64_t n) const { if (n > (uint64_t)(end - p)) throw std::runtime_error("Could not parse message: truncated SEAL object"); }
  template <class T> T pod() { need(sizeof(T)); T v; std::memcpy(&v, p, sizeof(T)); p += sizeof(T); return v; }
  ParmsId id() { need(32); ParmsId v; std::memcpy(v.data(), p, 32); p += 32; return v; }
  bool done() const { return p >= end; }
};
// the members of one object: a view into the parent buffer, or the decompressed bytes
struct Obj {
  std::vector<uint8_t> owned;
  Cur cur{nullptr, nullptr};
  uint8_t major = 0, minor = 0;
  Obj() = default;
  Obj(const Obj &) = delete;
  Obj &operator=(const Obj &) = delete;
};
// Serialization::Load: consumes one object from `c`
inline void open(Cur &c, uint64_t limit, Obj &o) {
  if (c.pod<uint16_t>() != SEAL_MAGIC) throw std::runtime_error("Could not parse message: not a SEAL object (bad magic)");
  if (c.pod<uint8_t>() != 0x10) throw std::runtime_error("Could not parse message: unsupported SEAL header size");
  o.major = c.pod<uint8_t>();
  o.minor = c.pod<uint8_t>();
  const uint8_t compr = c.pod<uint8_t>();
  (void)c.pod<uint16_t>();
  const uint64_t size = c.pod<uint64_t>();
  if (!((o.major == 3 && o.minor >= 5) || o.major == 4)) throw std::runtime_error("Could not parse message: SEAL object of an unsupported version " + std::to_string(o.major) + "." + std::to_string(o.minor));
  if (size < 16) throw std::runtime_error("Could not parse message: SEAL object size smaller than its header");
  c.need(size - 16);
  const uint8_t *body = c.p;
  c.p += size - 16;
  if (compr == None) {
    if (size - 16 > limit) throw std::runtime_error("Could not parse message: a SEAL object is larger than its bound");
    o.cur = Cur{body, body + (size - 16)};
  } else if (compr == Zlib || compr == Zstd) {
    // residues are pseudo-random 60-bit words: a genuine object shrinks by a few percent, only the empty slots of a
    // Galois key set (zeros) compress well — so the expansion is bounded by the compressed length as well, which
    // keeps a small hostile file from asking for the memory of a large key set
    limit = std::min<uint64_t>(limit, 8 * (size - 16) + ((uint64_t)32 << 20));
    o.owned = compr == Zlib ? zlib_inflate(body, size - 16, limit) : zstd_inflate(body, size - 16, limit);
    o.cur = Cur{o.owned.data(), o.owned.data() + o.owned.size()};
  } else {
    throw std::runtime_error("Could not parse message: unknown SEAL compression mode");
  }
}
inline u64 read_modulus(Cur &c) {
  Obj o;
  open(c, 64, o);
  return o.cur.pod<uint64_t>();
}
inline std::shared_ptr<HostContext> read_parms(Cur &c) {
  Obj o;
  open(c, LIMIT_SMALL, o);
  if (o.cur.pod<uint8_t>() != 2) throw std::runtime_error("Could not parse message: the encryption parameters are not CKKS parameters");
  const uint64_t N = o.cur.pod<uint64_t>(), k = o.cur.pod<uint64_t>();
  if (N < 1024 || N > 131072 || (N & (N - 1)) || k < 2 || k > 62) throw std::runtime_error("Could not parse message: invalid encryption parameters");
  std::vector<u64> primes;
  for (uint64_t i = 0; i < k; i++) primes.push_back(read_modulus(o.cur));
  (void)read_modulus(o.cur); // plain_modulus
  for (u64 q : primes)
    if (q < 2 || q >= ((u64)1 << 60) || (q - 1) % (2 * N) || !evah::is_prime(q)) throw std::runtime_error("Could not parse message: invalid coefficient modulus");
  for (size_t i = 0; i < primes.size(); i++)
    for (size_t j = 0; j < i; j++)
      if (primes[i] == primes[j]) throw std::runtime_error("Could not parse message: coefficient modulus primes must be distinct");
  return std::make_shared<HostContext>((uint32_t)N, primes);
}
inline std::vector<u64> read_dynarray(Cur &c, uint64_t expect_words) {
  Obj o;
  open(c, 16 + 8 * expect_words, o);
  const uint64_t n = o.cur.pod<uint64_t>();
  if (n != expect_words) throw std::runtime_error("Could not parse message: SEAL object data has the wrong length for its shape (seeded objects are not supported)");
  o.cur.need(8 * n);
  std::vector<u64> v(n);
  std::memcpy(v.data(), o.cur.p, 8 * n);
  return v;
}
struct CtFields {
  uint32_t size = 0, limbs = 0;
  double scale = 1.0;
  std::vector<u64> data;
};
// max_limbs: h.k for a key-level object (public key), h.k - 1 for a value
// does a complete SEAL object start here and fill the rest of the cursor?  (A key object written with the pre-3.5
// nesting holds its member as an inner object; a member's own first bytes are a parms_id hash, which passes this test
// with probability 2^-80.)
inline bool nested_object(const Cur &c) {
  if (c.end - c.p < 16) return false;
  uint16_t magic, reserved;
  uint64_t size;
  std::memcpy(&magic, c.p, 2);
  std::memcpy(&reserved, c.p + 6, 2);
  std::memcpy(&size, c.p + 8, 8);
  return magic == SEAL_MAGIC && c.p[2] == 0x10 && (c.p[3] == 3 || c.p[3] == 4) && c.p[5] <= 2 && reserved == 0 && size == (uint64_t)(c.end - c.p);
}
inline CtFields ciphertext_members(Obj &o, const HostContext &h, uint32_t max_limbs, uint32_t max_size);
inline CtFields read_ciphertext(Cur &c, const HostContext &h, uint32_t max_limbs, uint32_t max_size) {
  Obj o;
  open(c, (uint64_t)max_size * max_limbs * h.N * 8 + 4096, o);
  return ciphertext_members(o, h, max_limbs, max_size);
}
inline CtFields ciphertext_members(Obj &o, const HostContext &h, uint32_t max_limbs, uint32_t max_size) {
  const ParmsId id = o.cur.id();
  const uint8_t ntt = o.cur.pod<uint8_t>();
  const uint64_t size = o.cur.pod<uint64_t>(), N = o.cur.pod<uint64_t>(), cms = o.cur.pod<uint64_t>();
  CtFields f;
  f.scale = o.cur.pod<double>();
  if (o.major >= 4) (void)o.cur.pod<uint64_t>(); // correction_factor (BGV), SEAL 4.x
  if (N != h.N || cms < 1 || cms > max_limbs || size < 1 || size > max_size) throw std::runtime_error("Could not parse message: ciphertext shape does not match the encryption parameters");
  if (!ntt) throw std::runtime_error("Could not parse message: CKKS ciphertexts must be in NTT form");
  if (id != parms_id(h.N, h.primes.data(), (uint32_t)cms)) throw std::runtime_error("Could not parse message: ciphertext parms_id does not belong to the encryption parameters");
  if (!(f.scale > 0) || !std::isfinite(f.scale)) throw std::runtime_error("Could not parse message: invalid ciphertext scale");
  f.size = (uint32_t)size;
  f.limbs = (uint32_t)cms;
  f.data = read_dynarray(o.cur, size * cms * N);
  return f;
}
inline CtFields plaintext_members(Obj &o, const HostContext &h, uint32_t max_limbs);
inline CtFields read_plaintext(Cur &c, const HostContext &h, uint32_t max_limbs) {
  Obj o;
  open(c, (uint64_t)max_limbs * h.N * 8 + 4096, o);
  return plaintext_members(o, h, max_limbs);
}
inline CtFields plaintext_members(Obj &o, const HostContext &h, uint32_t max_limbs) {
  const ParmsId id = o.cur.id();
  const uint64_t cc = o.cur.pod<uint64_t>();
  CtFields f;
  f.scale = o.cur.pod<double>();
  if (cc == 0 || cc % h.N || cc / h.N > max_limbs) throw std::runtime_error("Could not parse message: plaintext shape does not match the encryption parameters");
  f.size = 1;
  f.limbs = (uint32_t)(cc / h.N);
  if (id != parms_id(h.N, h.primes.data(), f.limbs)) throw std::runtime_error("Could not parse message: plaintext parms_id does not belong to the encryption parameters (CKKS plaintexts are in NTT form)");
  if (!(f.scale > 0) || !std::isfinite(f.scale)) throw std::runtime_error("Could not parse message: invalid plaintext scale");
  f.data = read_dynarray(o.cur, cc);
  return f;
}
inline std::vector<u64> read_public_key(Cur &c, const HostContext &h) {
  Obj o;
  open(c, (uint64_t)2 * h.k * h.N * 8 + 8192, o);
  // SEAL >= 3.5: the object IS the ciphertext; the pre-3.5 nesting (and r03 files of this repo) holds it one level down
  CtFields f = nested_object(o.cur) ? read_ciphertext(o.cur, h, h.k, 2) : ciphertext_members(o, h, h.k, 2);
  if (f.size != 2 || f.limbs != h.k) throw std::runtime_error("Could not parse message: public key has the wrong size for its context");
  check_residues(f.data, h, "public key");
  return std::move(f.data);
}
struct KSwitchSet {
  uint64_t dim1 = 0;
  std::map<uint64_t, SwitchKey> slots;
};
inline KSwitchSet read_kswitch(Cur &c, const HostContext &h, uint64_t max_dim1, const char *what) {
  Obj o;
  open(c, LIMIT_KEYS, o);
  if (o.cur.id() != parms_id(h.N, h.primes.data(), h.k)) throw std::runtime_error(std::string("Could not parse message: ") + what + " do not belong to the encryption parameters");
  KSwitchSet s;
  s.dim1 = o.cur.pod<uint64_t>();
  if (s.dim1 > max_dim1) throw std::runtime_error(std::string("Could not parse message: ") + what + " hold too many entries");
  for (uint64_t i = 0; i < s.dim1; i++) {
    const uint64_t dim2 = o.cur.pod<uint64_t>();
    if (dim2 == 0) continue;
    if (dim2 != h.k - 1) throw std::runtime_error(std::string("Could not parse message: ") + what + " have the wrong decomposition count for the encryption parameters");
    SwitchKey k;
    k.n_digits = (uint32_t)dim2;
    k.data.reserve((size_t)dim2 * 2 * h.k * h.N);
    for (uint64_t j = 0; j < dim2; j++) {
      std::vector<u64> pk = read_public_key(o.cur, h);
      k.data.insert(k.data.end(), pk.begin(), pk.end());
    }
    s.slots.emplace(i, std::move(k));
  }
  return s;
}

// ------------------------------------------------------------------------------------------- seal.proto messages
enum SealType : uint32_t { CIPHERTEXT = 1, PLAINTEXT = 2, SECRET_KEY = 3, PUBLIC_KEY = 4, GALOIS_KEYS = 5, RELIN_KEYS = 6, ENCRYPTION_PARAMETERS = 7 };

inline std::string seal_object_msg(SealType t, const std::string &data) {
  wire::Out o;
  o.u(1, t);
  o.bytes(2, data);
  return o.b;
}
// -> the data bytes; the type tag must be `want` (seal_serialization.cpp:69-82: UNKNOWN and mismatches throw)
inline std::string open_seal_object(wire::In in, SealType want, uint32_t *got_type = nullptr) {
  uint32_t t = 0;
  std::string data;
  while (!in.done()) {
    const uint64_t tag = in.varint();
    if ((tag >> 3) == 1 && (tag & 7) == 0) t = (uint32_t)in.varint();
    else if ((tag >> 3) == 2 && (tag & 7) == 2) data = in.str();
    else in.skip((uint32_t)(tag & 7));
  }
  if (got_type) { *got_type = t; return data; }
  if (t == 0) throw std::runtime_error("SEAL message type set to UNKNOWN");
  if (t != (uint32_t)want) throw std::runtime_error("SEAL message type mismatch");
  return data;
}
inline Cur cur_of(const std::string &s) { return Cur{(const uint8_t *)s.data(), (const uint8_t *)s.data() + s.size()}; }

inline std::string encode_public(const HipPublic &p, Compr compr = None) {
  const HostContext &h = *p.host;
  wire::Out o;
  o.bytes(1, seal_object_msg(ENCRYPTION_PARAMETERS, parms_obj(h, compr)), true);
  o.bytes(2, seal_object_msg(PUBLIC_KEY, public_key_obj(h, p.pk.data.data(), compr)), true);
  std::map<uint64_t, const SwitchKey *> gal;
  for (auto &kv : p.galois) gal.emplace((uint64_t)(kv.first - 1) / 2, &kv.second);
  // KeyGenerator::create_galois_keys sizes keys_ to poly_modulus_degree whatever the steps are (seal.cpp:195
  // always calls it): a context without rotations carries N empty slots
  o.bytes(3, seal_object_msg(GALOIS_KEYS, kswitch_obj(h, h.N, gal, compr)), true);
  std::map<uint64_t, const SwitchKey *> rel{{0, &p.relin}};
  o.bytes(4, seal_object_msg(RELIN_KEYS, kswitch_obj(h, 1, rel, compr)), true);
  return o.b;
}
inline std::shared_ptr<HipPublic> decode_public(wire::In in) {
  std::string parts[5];
  bool have[5] = {false, false, false, false, false};
  while (!in.done()) {
    const uint64_t tag = in.varint();
    const uint32_t f = (uint32_t)(tag >> 3);
    if (f >= 1 && f <= 4 && (tag & 7) == 2) { parts[f] = in.str(); have[f] = true; }
    else in.skip((uint32_t)(tag & 7));
  }
  for (int f = 1; f <= 4; f++)
    if (!have[f]) throw std::runtime_error("SEAL message type set to UNKNOWN");
  auto p = std::make_shared<HipPublic>();
  {
    const std::string d = open_seal_object(wire::In(parts[1]), ENCRYPTION_PARAMETERS);
    Cur c = cur_of(d);
    p->host = read_parms(c);
  }
  const HostContext &h = *p->host;
  {
    const std::string d = open_seal_object(wire::In(parts[2]), PUBLIC_KEY);
    Cur c = cur_of(d);
    p->pk.data = read_public_key(c, h);
  }
  {
    const std::string d = open_seal_object(wire::In(parts[3]), GALOIS_KEYS);
    Cur c = cur_of(d);
    KSwitchSet s = read_kswitch(c, h, h.N, "Galois keys");
    for (auto &kv : s.slots) {
      check_switch_key(kv.second, h, "Galois key");
      p->galois.emplace((uint32_t)(2 * kv.first + 1), std::move(kv.second));
    }
  }
  {
    const std::string d = open_seal_object(wire::In(parts[4]), RELIN_KEYS);
    Cur c = cur_of(d);
    KSwitchSet s = read_kswitch(c, h, 1, "relinearization keys");
    auto it = s.slots.find(0);
    if (it == s.slots.end()) throw std::runtime_error("Could not parse message: relinearization key has the wrong size for its context");
    check_switch_key(it->second, h, "relinearization key");
    p->relin = std::move(it->second);
  }
  return p;
}

inline std::string encode_secret(const HipSecret &s, Compr compr = None) {
  wire::Out o;
  o.bytes(1, seal_object_msg(ENCRYPTION_PARAMETERS, parms_obj(*s.host, compr)), true);
  o.bytes(2, seal_object_msg(SECRET_KEY, secret_key_obj(*s.host, s.sk.s_ntt.data(), compr)), true);
  return o.b;
}
inline std::shared_ptr<HipSecret> decode_secret(wire::In in) {
  std::string parts[3];
  bool have[3] = {false, false, false};
  while (!in.done()) {
    const uint64_t tag = in.varint();
    const uint32_t f = (uint32_t)(tag >> 3);
    if (f >= 1 && f <= 2 && (tag & 7) == 2) { parts[f] = in.str(); have[f] = true; }
    else in.skip((uint32_t)(tag & 7));
  }
  if (!have[1] || !have[2]) throw std::runtime_error("SEAL message type set to UNKNOWN");
  auto s = std::make_shared<HipSecret>();
  {
    const std::string d = open_seal_object(wire::In(parts[1]), ENCRYPTION_PARAMETERS);
    Cur c = cur_of(d);
    s->host = read_parms(c);
  }
  const HostContext &h = *s->host;
  const std::string d = open_seal_object(wire::In(parts[2]), SECRET_KEY);
  Cur c = cur_of(d);
  Obj o;
  open(c, (uint64_t)h.k * h.N * 8 + 8192, o);
  CtFields f = nested_object(o.cur) ? read_plaintext(o.cur, h, h.k) : plaintext_members(o, h, h.k);
  if (f.limbs != h.k) throw std::runtime_error("Could not parse message: secret key has the wrong size for its context");
  check_residues(f.data, h, "secret key");
  s->sk.s_ntt = std::move(f.data);
  // SEAL stores the NTT form only; the ternary coefficients are its inverse transform under the first prime —
  // and that polynomial must reproduce every other row, or the rows are not one key
  std::vector<u64> t(s->sk.s_ntt.begin(), s->sk.s_ntt.begin() + h.N);
  h.intt(0, t.data());
  s->sk.s.resize(h.N);
  for (uint32_t j = 0; j < h.N; j++) {
    if (t[j] == 0) s->sk.s[j] = 0;
    else if (t[j] == 1) s->sk.s[j] = 1;
    else if (t[j] == h.primes[0] - 1) s->sk.s[j] = -1;
    else throw std::runtime_error("Could not parse message: secret key coefficients must be ternary");
  }
  for (uint32_t i = 1; i < h.k; i++) {
    for (uint32_t j = 0; j < h.N; j++) t[j] = s->sk.s[j] < 0 ? h.primes[i] - 1 : (u64)s->sk.s[j];
    h.ntt(i, t.data());
    if (std::memcmp(t.data(), s->sk.s_ntt.data() + (size_t)i * h.N, 8 * (size_t)h.N)) throw std::runtime_error("Could not parse message: secret key rows are not one polynomial");
  }
  return s;
}

// eva.msg.ConstantValue (eva.proto:15-22), dense form
inline std::string constant_msg(const std::vector<double> &v) {
  wire::Out c;
  c.u(1, v.size());
  if (!v.empty()) {
    c.tag(2, 2);
    c.varint(8 * v.size());
    c.b.append((const char *)v.data(), 8 * v.size());
  }
  return c.b;
}
inline std::vector<double> decode_constant(wire::In in) {
  uint64_t size = 0;
  std::vector<double> values;
  std::vector<uint32_t> sparse;
  while (!in.done()) {
    const uint64_t tag = in.varint();
    const uint32_t f = (uint32_t)(tag >> 3), wt = (uint32_t)(tag & 7);
    if (f == 1 && wt == 0) size = in.varint();
    else if (f == 2 && (wt == 2 || wt == 1)) in.repeated(wt, [&](wire::In &s) { values.push_back(s.f64()); });
    else if (f == 3 && (wt == 2 || wt == 0)) in.repeated(wt, [&](wire::In &s) { sparse.push_back((uint32_t)s.varint()); });
    else in.skip(wt);
  }
  if (size == 0 || size > ((uint64_t)1 << 20)) throw std::runtime_error("Could not parse message: raw value of an invalid size");
  std::vector<double> out(size, 0.0);
  if (!sparse.empty()) {
    if (sparse.size() != values.size()) throw std::runtime_error("Could not parse message: sparse raw value with mismatched index / value counts");
    for (size_t i = 0; i < sparse.size(); i++) {
      if (sparse[i] >= size) throw std::runtime_error("Could not parse message: sparse index out of range");
      out[sparse[i]] = values[i];
    }
  } else if (!values.empty()) {
    if (size % values.size()) throw std::runtime_error("Could not parse message: raw value count does not divide its size");
    for (uint64_t i = 0; i < size; i++) out[i] = values[i % values.size()];
  }
  return out;
}

inline std::string encode_valuation(const HipValuation &v, Compr compr = None) {
  if (!v.params) throw std::runtime_error("this valuation carries no encryption parameters (it was not made by encrypt() / execute() / load()): it cannot be written in the SEAL format");
  const HostContext &h = *v.params;
  wire::Out o;
  o.bytes(1, seal_object_msg(ENCRYPTION_PARAMETERS, parms_obj(h, compr)), true);
  std::map<std::string, const SchemeValue *> sorted; // deterministic files
  for (auto &kv : v.values) sorted.emplace(kv.first, &kv.second);
  for (auto &kv : sorted) {
    wire::Out e;
    e.bytes(1, kv.first, true);
    if (auto *c = std::get_if<HostCipher>(kv.second)) {
      const CipherWords &w = words(*c);
      if (c->limbs < 1 || c->limbs > h.k - 1 || w.size() != (size_t)c->size * c->limbs * h.N) throw std::runtime_error("ciphertext " + kv.first + " does not match the valuation's encryption parameters");
      e.bytes(2, seal_object_msg(CIPHERTEXT, ciphertext_obj(h, c->size, c->limbs, c->scale, (const u64 *)w.data(), compr)), true);
      o.bytes(2, e.b, true);
    } else if (auto *p = std::get_if<HostPlain>(kv.second)) {
      if (p->limbs < 1 || p->limbs > h.k - 1 || p->data.size() != (size_t)p->limbs * h.N) throw std::runtime_error("plaintext " + kv.first + " does not match the valuation's encryption parameters");
      e.bytes(2, seal_object_msg(PLAINTEXT, plaintext_obj(h, p->limbs, p->scale, p->data.data(), compr)), true);
      o.bytes(2, e.b, true);
    } else {
      e.bytes(2, constant_msg(std::get<std::vector<double>>(*kv.second)), true);
      o.bytes(3, e.b, true);
    }
  }
  return o.b;
}
inline HipValuation decode_valuation(wire::In in) {
  std::string parms;
  std::vector<std::pair<std::string, std::string>> values, raws;
  auto entry = [](wire::In e) {
    std::pair<std::string, std::string> kv;
    while (!e.done()) {
      const uint64_t tag = e.varint();
      if ((tag >> 3) == 1 && (tag & 7) == 2) kv.first = e.str();
      else if ((tag >> 3) == 2 && (tag & 7) == 2) kv.second = e.str();
      else e.skip((uint32_t)(tag & 7));
    }
    return kv;
  };
  bool have_parms = false;
  while (!in.done()) {
    const uint64_t tag = in.varint();
    const uint32_t f = (uint32_t)(tag >> 3);
    if (f == 1 && (tag & 7) == 2) { parms = in.str(); have_parms = true; }
    else if (f == 2 && (tag & 7) == 2) values.push_back(entry(in.sub()));
    else if (f == 3 && (tag & 7) == 2) raws.push_back(entry(in.sub()));
    else in.skip((uint32_t)(tag & 7));
  }
  if (!have_parms) throw std::runtime_error("SEAL message type set to UNKNOWN");
  HipValuation v;
  {
    const std::string d = open_seal_object(wire::In(parms), ENCRYPTION_PARAMETERS);
    Cur c = cur_of(d);
    v.params = read_parms(c);
  }
  const HostContext &h = *v.params;
  for (auto &kv : values) {
    uint32_t t = 0;
    const std::string d = open_seal_object(wire::In(kv.second), CIPHERTEXT, &t);
    Cur c = cur_of(d);
    if (t == CIPHERTEXT) {
      CtFields f = read_ciphertext(c, h, h.k - 1, 3);
      HostCipher hc;
      hc.size = f.size;
      hc.limbs = f.limbs;
      hc.scale = f.scale;
      hc.data.assign(f.data.begin(), f.data.end());
      v.values[kv.first] = std::move(hc);
    } else if (t == PLAINTEXT) {
      CtFields f = read_plaintext(c, h, h.k - 1);
      HostPlain hp;
      hp.limbs = f.limbs;
      hp.scale = f.scale;
      hp.data = std::move(f.data);
      v.values[kv.first] = std::move(hp);
    } else {
      throw std::runtime_error("Not a ciphertext or plaintext"); // seal_serialization.cpp:131-132
    }
  }
  for (auto &kv : raws) v.values[kv.first] = decode_constant(wire::In(kv.second));
  return v;
}

} // namespace sealfmt
} // namespace evahost
