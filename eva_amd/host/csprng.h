// csprng.h — the randomness behind keygen and encrypt (host side, outside execute()).
//
// The reference takes its randomness from SEAL's default UniformRandomGeneratorFactory (a BLAKE2
// XOF seeded with 512 bits of OS entropy, reached from /root/reference/eva/seal/seal.cpp:85
// `encryptor.encrypt` and :188-196 `KeyGenerator`).  Here: ChaCha20 (RFC 8439 block function) as
// a counter-mode generator keyed with 256 bits from getrandom(2) — one independent stream per
// keygen and per encrypt call.  A caller-supplied 64-bit seed (generate_keys(params, seed != 0))
// gives a reproducible stream for tests and is NOT secret-grade: 64 bits of key.
#pragma once
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <sys/random.h>

namespace evahost {

class SecureRng {
public:
  using result_type = uint64_t;
  static constexpr result_type min() { return 0; }
  static constexpr result_type max() { return ~(result_type)0; }

  SecureRng() { // 256-bit key + 64-bit nonce from the operating system
    unsigned char seed[40];
    size_t got = 0;
    while (got < sizeof seed) {
      ssize_t r = getrandom(seed + got, sizeof seed - got, 0);
      if (r < 0) throw std::runtime_error("getrandom failed: no entropy source for key generation / encryption");
      got += (size_t)r;
    }
    init(seed, seed + 32);
    wipe(seed, sizeof seed);
  }
  explicit SecureRng(uint64_t test_seed, uint64_t stream = 0) { // reproducible test hook
    unsigned char key[32] = {0}, nonce[8];
    std::memcpy(key, &test_seed, 8);
    std::memcpy(key + 8, "eva_amd test seed: not secret", 24);
    std::memcpy(nonce, &stream, 8);
    init(key, nonce);
  }
  ~SecureRng() { wipe(state_, sizeof state_); wipe(block_, sizeof block_); }
  SecureRng(const SecureRng &) = delete;
  SecureRng &operator=(const SecureRng &) = delete;

  result_type operator()() {
    if (pos_ == 8) refill();
    return block_[pos_++];
  }

private:
  uint32_t state_[16];
  uint64_t block_[8];
  int pos_ = 8;

  static void wipe(void *p, size_t n) {
    volatile unsigned char *v = static_cast<volatile unsigned char *>(p);
    while (n--) *v++ = 0;
  }
  static uint32_t rotl(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }
  static void quarter(uint32_t *s, int a, int b, int c, int d) {
    s[a] += s[b]; s[d] = rotl(s[d] ^ s[a], 16);
    s[c] += s[d]; s[b] = rotl(s[b] ^ s[c], 12);
    s[a] += s[b]; s[d] = rotl(s[d] ^ s[a], 8);
    s[c] += s[d]; s[b] = rotl(s[b] ^ s[c], 7);
  }
  void init(const unsigned char *key, const unsigned char *nonce8) {
    static const uint32_t sigma[4] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u};
    std::memcpy(state_, sigma, 16);
    std::memcpy(state_ + 4, key, 32);
    state_[12] = state_[13] = 0; // 64-bit block counter
    std::memcpy(state_ + 14, nonce8, 8);
  }
  void refill() {
    uint32_t w[16];
    std::memcpy(w, state_, sizeof w);
    for (int r = 0; r < 10; r++) {
      quarter(w, 0, 4, 8, 12); quarter(w, 1, 5, 9, 13); quarter(w, 2, 6, 10, 14); quarter(w, 3, 7, 11, 15);
      quarter(w, 0, 5, 10, 15); quarter(w, 1, 6, 11, 12); quarter(w, 2, 7, 8, 13); quarter(w, 3, 4, 9, 14);
    }
    for (int i = 0; i < 16; i++) w[i] += state_[i];
    std::memcpy(block_, w, sizeof block_);
    wipe(w, sizeof w);
    if (++state_[12] == 0) ++state_[13];
    pos_ = 0;
  }
};

} // namespace evahost
