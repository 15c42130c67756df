// seal_parity.cpp — diff this repository's CPU oracle (and therefore the HIP kernels that are
// bit-exact against it) with the REAL Microsoft SEAL, bit for bit.
//
// The reference (microsoft/EVA) takes its arithmetic from Microsoft SEAL >= 3.6
// (/root/reference/CMakeLists.txt:24 `find_package(SEAL 3.6 REQUIRED)`, README.md:28-36), which is
// not present where this repository was developed.  This standalone program builds against an
// installed SEAL 3.6.x ANYWHERE and checks, on the vectors written by
// tests/golden/export_seal_vectors.py:
//   1. CoeffModulus::Create(N, bits)                      == exported prime chain
//   2. the minimal primitive 2N-th root per prime          == exported psi
//   3. ntt_negacyclic_harvey / inverse_...                 == exported transforms
//   4. every Evaluator call EVA's SEALExecutor makes (seal_executor.h:114-243) on the exported
//      inputs and keys                                      == exported outputs
//   5. CKKSEncoder::encode on the exported value vectors   == exported plaintexts
//   6. Decryptor::decrypt / CKKSEncoder::decode            == exported plaintexts / slot values (as bits)
//   7. SEAL's binary object format (what the reference's files carry, seal_serialization.cpp:46-108): the
//      seal_*.bin objects written by this repository's writer (eva_amd/host/seal_format.h) load with SEAL's
//      own load() into the exported values, and SEAL's own save(compr_mode_type::none) of those values
//      reproduces the files byte for byte
// and prints one PASS/FAIL line per item plus a summary; exit status 0 only when all pass.
// With --time-triple it also times multiply + relinearize + rescale_to_next (BASELINE.json's
// metric) on one thread, so bench.py can report a `cpu_baseline` of kind "reference".
//
// build:  cmake -S tools -B build/seal_parity && cmake --build build/seal_parity
//    or:  g++ -O2 -std=c++17 tools/seal_parity.cpp -I<seal include dir> -L<seal lib dir> -lseal-3.6 -o seal_parity
// run:    python tests/golden/export_seal_vectors.py /tmp/vec && ./seal_parity /tmp/vec
//
// NOT compiled in the development container (no SEAL there): written against the SEAL 3.6 public
// API as used by the reference itself (eva/seal/seal.cpp:148-203, seal_executor.h).
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <fstream>
#include <iostream>
#include <map>
#include <sstream>
#include <string>
#include <vector>

#include "seal/seal.h"

using namespace seal;
using u64 = std::uint64_t;

static std::string g_dir;
static int g_fail = 0, g_pass = 0;

static std::vector<u64> load_u64(const std::string &name) {
  std::ifstream f(g_dir + "/" + name + ".u64", std::ios::binary | std::ios::ate);
  if (!f) throw std::runtime_error("missing vector file " + name + ".u64");
  const std::streamsize n = f.tellg();
  std::vector<u64> v((size_t)n / 8);
  f.seekg(0);
  f.read(reinterpret_cast<char *>(v.data()), n);
  return v;
}
static std::vector<double> load_f64(const std::string &name) {
  std::ifstream f(g_dir + "/" + name + ".f64", std::ios::binary | std::ios::ate);
  if (!f) throw std::runtime_error("missing vector file " + name + ".f64");
  const std::streamsize n = f.tellg();
  std::vector<double> v((size_t)n / 8);
  f.seekg(0);
  f.read(reinterpret_cast<char *>(v.data()), n);
  return v;
}
static std::map<std::string, std::vector<long long>> load_manifest() {
  std::ifstream f(g_dir + "/manifest.txt");
  if (!f) throw std::runtime_error("missing manifest.txt in " + g_dir);
  std::map<std::string, std::vector<long long>> m;
  std::string line;
  while (std::getline(f, line)) {
    std::istringstream is(line);
    std::string key;
    if (!(is >> key) || key[0] == '#') continue;
    long long v;
    while (is >> v) m[key].push_back(v);
  }
  return m;
}
static void report(const std::string &what, bool ok) {
  std::cout << (ok ? "PASS  " : "FAIL  ") << what << std::endl;
  (ok ? g_pass : g_fail)++;
}
static bool same(const u64 *got, const std::vector<u64> &want) { return std::memcmp(got, want.data(), want.size() * 8) == 0; }

int main(int argc, char **argv) {
  if (argc < 2) {
    std::cerr << "usage: seal_parity <vector dir> [--time-triple]" << std::endl;
    return 2;
  }
  g_dir = argv[1];
  const bool time_triple = argc > 2 && std::string(argv[2]) == "--time-triple";
  auto mf = load_manifest();
  const size_t N = (size_t)mf.at("N")[0];
  std::vector<int> bits(mf.at("bits").begin(), mf.at("bits").end());
  const double scale = std::pow(2.0, (double)mf.at("scale_log2")[0]);
  const size_t k = bits.size(), l = k - 1;

  std::cout << "SEAL " << SEAL_VERSION_MAJOR << "." << SEAL_VERSION_MINOR << "." << SEAL_VERSION_PATCH << ", N = " << N << ", k = " << k
            << std::endl;

  // ---- the context exactly as the reference builds it (eva/seal/seal.cpp:169, 179-182)
  EncryptionParameters parms(scheme_type::ckks);
  parms.set_poly_modulus_degree(N);
  parms.set_coeff_modulus(CoeffModulus::Create(N, bits));
  SEALContext context(parms, true, sec_level_type::none);
  auto key_data = context.key_context_data();

  // 1. primes   2. psi
  {
    auto want = load_u64("primes"), psi = load_u64("psi");
    bool okp = true, okr = true;
    auto &cm = key_data->parms().coeff_modulus();
    for (size_t i = 0; i < k; i++) {
      okp = okp && cm[i].value() == want[i];
      okr = okr && key_data->small_ntt_tables()[i].get_root() == psi[i];
    }
    report("CoeffModulus::Create prime chain", okp);
    report("minimal primitive 2N-th roots (NTTTables::get_root)", okr);
  }
  // 3. transforms under prime 0
  {
    auto poly = load_u64("poly");
    auto f = poly, b = poly;
    util::ntt_negacyclic_harvey(f.data(), key_data->small_ntt_tables()[0]);
    util::inverse_ntt_negacyclic_harvey(b.data(), key_data->small_ntt_tables()[0]);
    report("ntt_negacyclic_harvey", same(f.data(), load_u64("out_ntt0")));
    report("inverse_ntt_negacyclic_harvey", same(b.data(), load_u64("out_intt0")));
  }

  // ---- values at the first data level (l limbs)
  auto first = context.first_context_data();
  const parms_id_type pid = first->parms_id();
  auto make_ct = [&](const std::string &name, size_t size) {
    auto d = load_u64(name);
    Ciphertext ct;
    ct.resize(context, pid, size);
    ct.is_ntt_form() = true;
    ct.scale() = scale;
    if (d.size() != size * l * N) throw std::runtime_error("unexpected length of " + name);
    std::memcpy(ct.data(), d.data(), d.size() * 8);
    return ct;
  };
  auto make_pt = [&](const std::string &name) {
    auto d = load_u64(name);
    Plaintext pt;
    pt.parms_id() = parms_id_zero;
    pt.resize(l * N);
    std::memcpy(pt.data(), d.data(), d.size() * 8);
    pt.parms_id() = pid;
    pt.scale() = scale;
    return pt;
  };
  Ciphertext a2 = make_ct("a2", 2), b2 = make_ct("b2", 2), a3 = make_ct("a3", 3);
  Plaintext pt = make_pt("pt");

  // ---- keys: generate real key objects for their structure, then overwrite every residue with the
  // exported uniform data (layout [digit][2][k][N] == SEAL's KSwitchKeys: one size-2 key-level
  // ciphertext per digit)
  KeyGenerator keygen(context);
  RelinKeys rk;
  keygen.create_relin_keys(rk);
  {
    auto d = load_u64("relin_key");
    for (size_t J = 0; J < l; J++) std::memcpy(rk.data()[0][J].data().data(), d.data() + J * 2 * k * N, 2 * k * N * 8);
  }
  std::vector<int> steps(mf.at("rot_steps").begin(), mf.at("rot_steps").end());
  // the steps of the hoisted-set vectors (section 4b) get keys of their own ("galois_key_h<step>")
  std::vector<int> hoist_steps;
  if (mf.count("hoist_steps")) hoist_steps.assign(mf.at("hoist_steps").begin(), mf.at("hoist_steps").end());
  GaloisKeys gk, gk_h;
  keygen.create_galois_keys(steps, gk);
  if (!hoist_steps.empty()) keygen.create_galois_keys(hoist_steps, gk_h);
  auto fill_key = [&](GaloisKeys &keys, int s, const std::string &file) {
    auto d = load_u64(file);
    const std::uint32_t elt = key_data->galois_tool()->get_elt_from_step(s);
    auto &kv = keys.data()[GaloisKeys::get_index(elt)];
    for (size_t J = 0; J < l; J++) std::memcpy(kv[J].data().data(), d.data() + J * 2 * k * N, 2 * k * N * 8);
  };
  for (int s : steps) fill_key(gk, s, "galois_key_" + std::to_string(s));
  for (int s : hoist_steps) fill_key(gk_h, s, "galois_key_h" + std::to_string(s));

  // 4. evaluator calls (seal_executor.h line in brackets)
  Evaluator ev(context);
  Ciphertext o;
  ev.add(a2, b2, o);                   report("add 2+2 [:124]", same(o.data(), load_u64("out_add")));
  ev.add(a3, b2, o);                   report("add 3+2 [:124]", same(o.data(), load_u64("out_add_32")));
  ev.sub(a2, b2, o);                   report("sub 2-2 [:140]", same(o.data(), load_u64("out_sub")));
  ev.sub(a2, a3, o);                   report("sub 2-3 [:140]", same(o.data(), load_u64("out_sub_23")));
  ev.negate(a3, o);                    report("negate [:194]", same(o.data(), load_u64("out_negate")));
  ev.add_plain(a2, pt, o);             report("add_plain [:127]", same(o.data(), load_u64("out_add_plain")));
  ev.sub_plain(a2, pt, o);             report("sub_plain [:143]", same(o.data(), load_u64("out_sub_plain")));
  ev.multiply(a2, b2, o);              report("multiply [:164]", same(o.data(), load_u64("out_multiply")));
  ev.square(a2, o);                    report("square [:162]", same(o.data(), load_u64("out_square")));
  ev.multiply_plain(a3, pt, o);        report("multiply_plain [:168]", same(o.data(), load_u64("out_multiply_plain")));
  ev.relinearize(a3, rk, o);           report("relinearize [:200]", same(o.data(), load_u64("out_relinearize")));
  { Ciphertext r; ev.rescale_to_next(o, r); report("rescale_to_next(relinearize) [:200,:213]", same(r.data(), load_u64("out_relin_rescale"))); }
  ev.rescale_to_next(a2, o);           report("rescale_to_next size 2 [:213]", same(o.data(), load_u64("out_rescale")));
  ev.rescale_to_next(a3, o);           report("rescale_to_next size 3 [:213]", same(o.data(), load_u64("out_rescale3")));
  ev.mod_switch_to_next(a3, o);        report("mod_switch_to_next [:206]", same(o.data(), load_u64("out_mod_switch")));
  for (int s : steps) {
    ev.rotate_vector(a2, s, gk, o);
    report("rotate_vector " + std::to_string(s) + " [:181,:188]", same(o.data(), load_u64("out_rotate_" + std::to_string(s))));
  }
  {
    Ciphertext m, r, t;
    ev.multiply(a2, b2, m);
    ev.relinearize(m, rk, r);
    ev.rescale_to_next(r, t);
    report("op-triple multiply+relinearize+rescale", same(t.data(), load_u64("out_triple")));
  }
  // 4b. the product's two "exactness" shortcuts, as SEAL computes them.  (i) what the HIP backend evaluates as ONE fused
  // multiply -> relinearize -> rescale (also as a square); (ii) what it evaluates as ONE hoisted rotation set: 8
  // rotate_vector calls on one source — dense, with a zero limb in c1 (the hoisted path's guarded fallback), and
  // transparent (c1 = 0; a SEAL built with SEAL_THROW_ON_TRANSPARENT_CIPHERTEXT refuses that result: reported as SKIP)
  if (std::ifstream(g_dir + "/out_triple_square.u64")) {
    Ciphertext m, r, t;
    ev.square(a2, m);
    ev.relinearize(m, rk, r);
    ev.rescale_to_next(r, t);
    report("fused form: square+relinearize+rescale", same(t.data(), load_u64("out_triple_square")));
  }
  for (const char *src_name : {"dense", "zero_limb", "transparent"}) {
    if (hoist_steps.empty()) break;
    Ciphertext src = make_ct(std::string("hoist_src_") + src_name, 2);
    for (int s : hoist_steps) {
      const std::string what = std::string("hoisted set: rotate_vector ") + std::to_string(s) + " of the " + src_name + " source [:181,:188]";
      try {
        ev.rotate_vector(src, s, gk_h, o);
        report(what, same(o.data(), load_u64(std::string("out_hoist_") + src_name + "_" + std::to_string(s))));
      } catch (const std::logic_error &e) {
        std::cout << "SKIP  " << what << " (" << e.what() << ")" << std::endl;
      }
    }
  }

  // 5. encoder (seal_executor.h:242): full slot vectors at 2^scale_bits, first data level
  CKKSEncoder encoder(context);
  if (mf.count("enc_scale_bits")) {
    auto &sb = mf.at("enc_scale_bits");
    for (size_t c = 0; c < sb.size(); c++) {
      auto vals = load_f64("enc_values_" + std::to_string(c));
      Plaintext p;
      encoder.encode(vals, pid, std::pow(2.0, (double)sb[c]), p);
      report("CKKSEncoder::encode at 2^" + std::to_string(sb[c]) + " [:242]", same(p.data(), load_u64("out_encode_" + std::to_string(c))));
    }
  }

  // 6. Decryptor::decrypt and CKKSEncoder::decode (eva/seal/seal.cpp:132-135).  The secret key object is
  // generated for its structure and overwritten with the exported NTT-form key (k limbs); decode results
  // are float64 and compared as bit patterns — the oracle restates decode_internal's operation order.
  auto file_exists = [&](const std::string &name) { return (bool)std::ifstream(g_dir + "/" + name); };
  if (file_exists("sk_ntt.u64")) {
    SecretKey sk = keygen.secret_key();
    {
      auto d = load_u64("sk_ntt");
      if (d.size() != k * N) throw std::runtime_error("unexpected length of sk_ntt");
      std::memcpy(sk.data().data(), d.data(), d.size() * 8);
    }
    Decryptor decryptor(context, sk);
    Plaintext p;
    decryptor.decrypt(a2, p);
    report("Decryptor::decrypt size 2 [seal.cpp:132]", same(p.data(), load_u64("out_decrypt2")));
    decryptor.decrypt(a3, p);
    report("Decryptor::decrypt size 3 [seal.cpp:132]", same(p.data(), load_u64("out_decrypt3")));
    auto check_decode = [&](const std::string &what, Plaintext &plain, const std::string &want_name) {
      std::vector<double> got;
      encoder.decode(plain, got);
      auto want = load_f64(want_name);
      report("CKKSEncoder::decode " + what + " [seal.cpp:134]", got.size() == want.size() && std::memcmp(got.data(), want.data(), want.size() * 8) == 0);
    };
    if (mf.count("enc_scale_bits")) {
      auto &sb = mf.at("enc_scale_bits");
      for (size_t c = 0; c < sb.size(); c++) {
        Plaintext q = make_pt("out_encode_" + std::to_string(c));
        q.scale() = std::pow(2.0, (double)sb[c]);
        check_decode("of the encoding at 2^" + std::to_string(sb[c]), q, "out_decode_" + std::to_string(c));
      }
    }
    check_decode("of uniformly random residues", pt, "out_decode_pt");
    decryptor.decrypt(a3, p);
    check_decode("of a decrypted size-3 ciphertext", p, "out_decode_dec3");
  }

  // 7. the object format.  load(context, ...) runs SEAL's own validity checks (parms_id in the chain, data
  // below the moduli, shapes); the byte comparison pins header, field order and widths, nesting and parms_id.
  if (file_exists("seal_parms.bin")) {
    auto load_bytes = [&](const std::string &name) {
      std::ifstream f(g_dir + "/" + name + ".bin", std::ios::binary | std::ios::ate);
      if (!f) throw std::runtime_error("missing object file " + name + ".bin");
      std::vector<seal_byte> v((size_t)f.tellg());
      f.seekg(0);
      f.read(reinterpret_cast<char *>(v.data()), (std::streamsize)v.size());
      return v;
    };
    auto saves_as = [&](const auto &obj, const std::vector<seal_byte> &want) {
      std::vector<seal_byte> buf((size_t)obj.save_size(compr_mode_type::none));
      const auto n = obj.save(buf.data(), buf.size(), compr_mode_type::none);
      return (size_t)n == want.size() && std::memcmp(buf.data(), want.data(), want.size()) == 0;
    };
    {
      auto b = load_bytes("seal_parms");
      EncryptionParameters p2;
      p2.load(b.data(), b.size());
      report("EncryptionParameters::load of this repo's object", p2 == parms);
      report("EncryptionParameters::save == this repo's bytes", saves_as(parms, b));
    }
    auto check_ct = [&](const std::string &file, const Ciphertext &want, const std::string &what) {
      auto b = load_bytes(file);
      Ciphertext c;
      c.load(context, b.data(), b.size());
      report("Ciphertext::load " + what, c.size() == want.size() && c.parms_id() == want.parms_id() && c.is_ntt_form() && c.scale() == want.scale() &&
                                             std::memcmp(c.data(), want.data(), want.size() * l * N * 8) == 0);
      report("Ciphertext::save " + what + " == this repo's bytes", saves_as(want, b));
    };
    check_ct("seal_ct_a2", a2, "size 2");
    check_ct("seal_ct_a3", a3, "size 3");
    {
      auto b = load_bytes("seal_pt");
      Plaintext p;
      p.load(context, b.data(), b.size());
      report("Plaintext::load", p.parms_id() == pt.parms_id() && p.scale() == pt.scale() && p.coeff_count() == l * N && std::memcmp(p.data(), pt.data(), l * N * 8) == 0);
      report("Plaintext::save == this repo's bytes", saves_as(pt, b));
    }
    {
      auto b = load_bytes("seal_pk");
      auto want = load_u64("pk");
      PublicKey pk;
      pk.load(context, b.data(), b.size());
      report("PublicKey::load", pk.data().size() == 2 && pk.parms_id() == key_data->parms_id() && same(pk.data().data(), want));
      report("PublicKey::save == this repo's bytes", saves_as(pk, b));
    }
    if (file_exists("sk_ntt.u64")) {
      auto b = load_bytes("seal_sk");
      auto want = load_u64("sk_ntt");
      SecretKey sk;
      sk.load(context, b.data(), b.size());
      report("SecretKey::load", sk.parms_id() == key_data->parms_id() && same(sk.data().data(), want));
      report("SecretKey::save == this repo's bytes", saves_as(sk, b));
    }
    {
      auto b = load_bytes("seal_relin");
      RelinKeys r2;
      r2.load(context, b.data(), b.size());
      bool ok = r2.data().size() == 1 && r2.data()[0].size() == l;
      for (size_t J = 0; ok && J < l; J++) ok = std::memcmp(r2.data()[0][J].data().data(), rk.data()[0][J].data().data(), 2 * k * N * 8) == 0;
      report("RelinKeys::load", ok);
      report("RelinKeys::save == this repo's bytes", saves_as(rk, b));
    }
    {
      auto b = load_bytes("seal_galois");
      GaloisKeys g2;
      g2.load(context, b.data(), b.size());
      bool ok = g2.data().size() == gk.data().size();
      for (size_t i = 0; ok && i < gk.data().size(); i++) {
        ok = g2.data()[i].size() == gk.data()[i].size();
        for (size_t J = 0; ok && J < gk.data()[i].size(); J++)
          ok = std::memcmp(g2.data()[i][J].data().data(), gk.data()[i][J].data().data(), 2 * k * N * 8) == 0;
      }
      report("GaloisKeys::load", ok);
      report("GaloisKeys::save == this repo's bytes", saves_as(gk, b));
    }
  }

  if (time_triple) {
    Ciphertext m, r, t;
    const int reps = 20;
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < reps; i++) {
      ev.multiply(a2, b2, m);
      ev.relinearize(m, rk, r);
      ev.rescale_to_next(r, t);
    }
    const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::cout << "TIMING op-triples/s " << reps / s << " threads 1 N " << N << " limbs " << l << std::endl;
  }
  std::cout << "SUMMARY " << g_pass << " passed, " << g_fail << " failed" << std::endl;
  return g_fail ? 1 : 0;
}
