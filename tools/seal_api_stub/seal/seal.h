// COMPILE-CHECK ONLY — this is NOT Microsoft SEAL and contains no arithmetic.
//
// tools/seal_parity.cpp is written against the public API of Microsoft SEAL 3.6 (the reference's
// dependency, /root/reference/CMakeLists.txt:24), which is not installed where this repository is
// developed.  This header DECLARES — without defining anything — exactly the SEAL 3.6 names that
// checker uses, so that `g++ -fsyntax-only -I tools/seal_api_stub tools/seal_parity.cpp`
// (tests/test_seal_parity.py) catches typos and type errors in the checker.  Nothing can be linked
// or run against it, it pins nothing, and it is never on an include path of the product, the oracle
// or any test that computes.  The declarations are restated from the SEAL 3.6 API as the reference
// itself uses it (/root/reference/eva/seal/seal.cpp:148-203, seal_executor.h:114-243); a real SEAL
// is the only judge of the checker — see tools/seal_probe.py.
#pragma once
#include <array>
#include <cstddef>
#include <ios>
#include <cstdint>
#include <memory>
#include <vector>

#define SEAL_VERSION_MAJOR 3
#define SEAL_VERSION_MINOR 6
#define SEAL_VERSION_PATCH 0

namespace seal {

using parms_id_type = std::array<std::uint64_t, 4>;
extern const parms_id_type parms_id_zero;
enum class scheme_type : std::uint8_t { none = 0, bfv = 1, ckks = 2 };
enum class sec_level_type : int { none = 0, tc128 = 128, tc192 = 192, tc256 = 256 };
using seal_byte = std::byte;
enum class compr_mode_type : std::uint8_t { none = 0, zlib = 1, zstd = 2 };
class SEALContext;
// every serialisable SEAL class has these three (Serialization::Save / Load behind them)
#define SEAL_STUB_SERIALIZABLE                                                                                          \
  std::streamoff save_size(compr_mode_type compr_mode) const;                                                           \
  std::streamoff save(seal_byte *out, std::size_t size, compr_mode_type compr_mode) const;                              \
  std::streamoff load(const SEALContext &context, const seal_byte *in, std::size_t size);

class Modulus {
public:
  std::uint64_t value() const;
};
class CoeffModulus {
public:
  static std::vector<Modulus> Create(std::size_t poly_modulus_degree, std::vector<int> bit_sizes);
};
class EncryptionParameters {
public:
  EncryptionParameters(scheme_type scheme = scheme_type::none);
  std::streamoff save_size(compr_mode_type compr_mode) const;
  std::streamoff save(seal_byte *out, std::size_t size, compr_mode_type compr_mode) const;
  std::streamoff load(const seal_byte *in, std::size_t size);
  bool operator==(const EncryptionParameters &other) const;
  void set_poly_modulus_degree(std::size_t poly_modulus_degree);
  void set_coeff_modulus(const std::vector<Modulus> &coeff_modulus);
  const std::vector<Modulus> &coeff_modulus() const;
};

namespace util {
class NTTTables {
public:
  std::uint64_t get_root() const;
};
class GaloisTool {
public:
  std::uint32_t get_elt_from_step(int step) const;
};
void ntt_negacyclic_harvey(std::uint64_t *operand, const NTTTables &tables);
void inverse_ntt_negacyclic_harvey(std::uint64_t *operand, const NTTTables &tables);
} // namespace util

class SEALContext {
public:
  class ContextData {
  public:
    const EncryptionParameters &parms() const;
    const parms_id_type &parms_id() const;
    const util::NTTTables *small_ntt_tables() const;
    util::GaloisTool *galois_tool() const;
  };
  SEALContext(const EncryptionParameters &parms, bool expand_mod_chain = true, sec_level_type sec_level = sec_level_type::tc128);
  std::shared_ptr<const ContextData> key_context_data() const;
  std::shared_ptr<const ContextData> first_context_data() const;
};

class Plaintext {
public:
  SEAL_STUB_SERIALIZABLE
  std::size_t coeff_count() const;
  const parms_id_type &parms_id() const;
  const double &scale() const;
  void resize(std::size_t coeff_count);
  std::uint64_t *data();
  const std::uint64_t *data() const;
  parms_id_type &parms_id();
  double &scale();
};
class Ciphertext {
public:
  SEAL_STUB_SERIALIZABLE
  std::size_t size() const;
  const parms_id_type &parms_id() const;
  bool is_ntt_form() const;
  const double &scale() const;
  void resize(const SEALContext &context, parms_id_type parms_id, std::size_t size);
  std::uint64_t *data();
  const std::uint64_t *data() const;
  bool &is_ntt_form();
  double &scale();
};
class SecretKey {
public:
  SEAL_STUB_SERIALIZABLE
  Plaintext &data();
  const parms_id_type &parms_id() const;
};
class PublicKey {
public:
  SEAL_STUB_SERIALIZABLE
  Ciphertext &data();
  const parms_id_type &parms_id() const;
};
class KSwitchKeys {
public:
  SEAL_STUB_SERIALIZABLE
  std::vector<std::vector<PublicKey>> &data();
};
class RelinKeys : public KSwitchKeys {};
class GaloisKeys : public KSwitchKeys {
public:
  static std::size_t get_index(std::uint32_t galois_elt);
};
class KeyGenerator {
public:
  KeyGenerator(const SEALContext &context);
  const SecretKey &secret_key() const;
  void create_relin_keys(RelinKeys &destination);
  void create_galois_keys(const std::vector<int> &steps, GaloisKeys &destination);
};
class Evaluator {
public:
  Evaluator(const SEALContext &context);
  void add(const Ciphertext &a, const Ciphertext &b, Ciphertext &destination) const;
  void sub(const Ciphertext &a, const Ciphertext &b, Ciphertext &destination) const;
  void negate(const Ciphertext &a, Ciphertext &destination) const;
  void add_plain(const Ciphertext &a, const Plaintext &p, Ciphertext &destination) const;
  void sub_plain(const Ciphertext &a, const Plaintext &p, Ciphertext &destination) const;
  void multiply(const Ciphertext &a, const Ciphertext &b, Ciphertext &destination) const;
  void square(const Ciphertext &a, Ciphertext &destination) const;
  void multiply_plain(const Ciphertext &a, const Plaintext &p, Ciphertext &destination) const;
  void relinearize(const Ciphertext &a, const RelinKeys &keys, Ciphertext &destination) const;
  void rescale_to_next(const Ciphertext &a, Ciphertext &destination) const;
  void mod_switch_to_next(const Ciphertext &a, Ciphertext &destination) const;
  void rotate_vector(const Ciphertext &a, int steps, const GaloisKeys &keys, Ciphertext &destination) const;
};
class CKKSEncoder {
public:
  CKKSEncoder(const SEALContext &context);
  void encode(const std::vector<double> &values, parms_id_type parms_id, double scale, Plaintext &destination);
  void decode(const Plaintext &plain, std::vector<double> &destination);
};
class Decryptor {
public:
  Decryptor(const SEALContext &context, const SecretKey &secret_key);
  void decrypt(const Ciphertext &encrypted, Plaintext &destination);
};

} // namespace seal
