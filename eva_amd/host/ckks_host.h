// ckks_host.h — host-side CKKS (FP64 encoder, key generation, encryption, decryption) for the
// MI355X backend.  These are the once-per-program / once-per-input steps that the reference runs
// through SEAL on the CPU (/root/reference/eva/seal/seal.cpp:24-102 encrypt, :124-146 decrypt,
// :174-203 generateKeys); they stay on the host here too.  execute() never comes through this
// file for ciphertext arithmetic — that is libeva_hip.so.
//
// Algebra follows SURVEY.md Appendix A.9/A.10 (restating SEAL 3.6 CKKSEncoder / KeyGenerator /
// Encryptor / Decryptor).  Randomness is not SEAL's stream; only the algebraic relations matter.
#pragma once
#include <cmath>
#include <complex>
#include <cstdint>
#include <memory>
#include <random>
#include <stdexcept>
#include <vector>

#include "csprng.h"
#include "hostmath.h"
#include "../../include/eva_hip.h"

namespace evahost {

using evah::u128;
using evah::u64;

struct HostPrime {
  u64 q = 0, r0 = 0, r1 = 0; // floor(2^128/q)
  u64 ninv = 0, ninv_s = 0;
  std::vector<u64> rp, rps, irp, irps; // heap-ordered root powers + Shoup quotients
};

class HostContext {
public:
  uint32_t N, logN, k;
  std::vector<u64> primes;
  std::vector<HostPrime> pm;
  std::vector<int> total_bits; // total_bits[l] = bit length of q_0...q_{l-1}

  HostContext(uint32_t N_, const std::vector<u64> &primes_) : N(N_), logN(evah::ilog2(N_)), k((uint32_t)primes_.size()), primes(primes_) {
    if (N < 2 || (N & (N - 1))) throw std::invalid_argument("poly_modulus_degree must be a power of two");
    pm.resize(k);
    for (uint32_t i = 0; i < k; i++) {
      HostPrime &m = pm[i];
      m.q = primes[i];
      u128 ratio = (~(u128)0) / m.q;
      m.r0 = (u64)ratio;
      m.r1 = (u64)(ratio >> 64);
      u64 psi = evah::minimal_primitive_root(N, m.q), psi_inv = evah::invmod(psi, m.q);
      m.rp = evah::root_power_table(N, m.q, psi);
      m.irp = evah::root_power_table(N, m.q, psi_inv);
      m.rps.resize(N);
      m.irps.resize(N);
      for (uint32_t j = 0; j < N; j++) {
        m.rps[j] = evah::shoup(m.rp[j], m.q);
        m.irps[j] = evah::shoup(m.irp[j], m.q);
      }
      m.ninv = evah::invmod(N % m.q, m.q);
      m.ninv_s = evah::shoup(m.ninv, m.q);
    }
    std::vector<u64> w{1};
    total_bits.push_back(0);
    for (uint32_t i = 0; i < k; i++) {
      u64 carry = 0;
      for (auto &x : w) {
        u128 t = (u128)x * primes[i] + carry;
        x = (u64)t;
        carry = (u64)(t >> 64);
      }
      if (carry) w.push_back(carry);
      int bits = (int)(w.size() - 1) * 64;
      for (u64 top = w.back(); top; top >>= 1) bits++;
      total_bits.push_back(bits);
    }
    pow2_.resize(k);
    for (uint32_t i = 0; i < k; i++) { // 2^e mod q_i for the encoder's multi-precision residues
      pow2_[i].resize(1100);
      u64 p = 1 % primes[i];
      for (auto &v : pow2_[i]) { v = p; p = evah::addmod(p, p, primes[i]); }
    }
    init_encoder();
  }

  // ---- modular helpers
  static inline u64 mul_shoup_lazy(u64 x, u64 w, u64 ws, u64 q) { return x * w - (u64)(((u128)x * ws) >> 64) * q; }
  inline u64 mulm(u64 a, u64 b, uint32_t i) const {
    const HostPrime &m = pm[i];
    u128 x = (u128)a * b;
    u64 x0 = (u64)x, x1 = (u64)(x >> 64);
    u64 carry = (u64)(((u128)x0 * m.r0) >> 64);
    u128 t = (u128)x0 * m.r1;
    u64 tmp1 = (u64)t + carry, tmp3 = (u64)(t >> 64) + (tmp1 < carry);
    t = (u128)x1 * m.r0;
    u64 tmp1b = tmp1 + (u64)t;
    carry = (u64)(t >> 64) + (tmp1b < tmp1);
    u64 r = x0 - (x1 * m.r1 + tmp3 + carry) * m.q;
    while (r >= m.q) r -= m.q;
    return r;
  }

  // negacyclic NTT, natural -> bit-reversed, canonical output
  void ntt(uint32_t i, u64 *x) const {
    const HostPrime &m = pm[i];
    const u64 q = m.q, q2 = 2 * q;
    for (uint32_t mm = 1, gap = N >> 1; mm < N; mm <<= 1, gap >>= 1)
      for (uint32_t g = 0; g < mm; g++) {
        const u64 w = m.rp[mm + g], ws = m.rps[mm + g];
        u64 *a = x + 2 * (size_t)g * gap, *b = a + gap;
        for (uint32_t j = 0; j < gap; j++) {
          u64 X = a[j];
          X -= (X >= q2) ? q2 : 0;
          u64 T = mul_shoup_lazy(b[j], w, ws, q);
          a[j] = X + T;
          b[j] = X + q2 - T;
        }
      }
    for (uint32_t j = 0; j < N; j++) {
      u64 v = x[j];
      v -= (v >= q2) ? q2 : 0;
      v -= (v >= q) ? q : 0;
      x[j] = v;
    }
  }
  void intt(uint32_t i, u64 *x) const {
    const HostPrime &m = pm[i];
    const u64 q = m.q, q2 = 2 * q;
    for (uint32_t mm = N >> 1, gap = 1; mm >= 1; mm >>= 1, gap <<= 1)
      for (uint32_t g = 0; g < mm; g++) {
        const u64 w = m.irp[mm + g], ws = m.irps[mm + g];
        u64 *a = x + 2 * (size_t)g * gap, *b = a + gap;
        for (uint32_t j = 0; j < gap; j++) {
          u64 X = a[j], Y = b[j], S = X + Y;
          a[j] = S - ((S >= q2) ? q2 : 0);
          b[j] = mul_shoup_lazy(X + q2 - Y, w, ws, q);
        }
      }
    for (uint32_t j = 0; j < N; j++) {
      u64 v = mul_shoup_lazy(x[j], m.ninv, m.ninv_s, q);
      x[j] = v >= q ? v - q : v;
    }
  }

  // residues of an integer-valued double of any magnitude: v = m * 2^e exactly (53-bit m), so
  // v mod q = (m mod q) * (2^e mod q) — what SEAL's multi-precision encode path computes
  void residues_of(double v, uint32_t limbs, u64 *out, size_t stride) const {
    const bool neg = std::signbit(v);
    const double a = std::fabs(v);
    int e = 0;
    u64 mant;
    if (a < 9007199254740992.0) { mant = (u64)a; }  // < 2^53: exact
    else {
      double fr = std::frexp(a, &e); // a = fr * 2^e, fr in [0.5,1)
      mant = (u64)std::ldexp(fr, 53);
      e -= 53;
    }
    for (uint32_t i = 0; i < limbs; i++) {
      const u64 q = primes[i];
      u64 r = mant % q;
      if (e > 0) r = evah::mulmod(r, e < (int)pow2_[i].size() ? pow2_[i][e] : evah::powmod(2, (u64)e, q), q);
      out[(size_t)i * stride] = (neg && r) ? q - r : r;
    }
  }

  // ---- CKKS encoder.  values: N/2 slots (already replicated by the caller).
  // Coefficient-form residues [limbs][N]; the forward NTT is done by the caller (device or host).
  // Floating-point operation order is SEAL 3.6's (CKKSEncoder::encode_internal reached from
  // /root/reference/eva/seal/seal_executor.h:242): Gentleman-Sande stages over the sequentially
  // stored inverse roots, and the factor scale/N applied INSIDE the last stage — sums are scaled,
  // differences are multiplied by the pre-scaled root — so the rounded coefficients are SEAL's.
  static std::complex<double> cmul(const std::complex<double> &a, const std::complex<double> &b) {
    // four rounded products, one rounded difference and sum (no fused multiply-add)
    const double p = a.real() * b.real(), q = a.imag() * b.imag();
    const double r = a.real() * b.imag(), t = a.imag() * b.real();
    return {p - q, r + t};
  }
  void encode_coeff(const double *values, double scale, uint32_t limbs, u64 *out) const {
    const uint32_t slots = N >> 1;
    std::vector<std::complex<double>> c(N);
    for (uint32_t i = 0; i < slots; i++) {
      c[slot_map_[i]] = std::complex<double>(values[i], 0.0);
      c[slot_map_[slots + i]] = std::complex<double>(values[i], -0.0); // conj of a real value
    }
    const double fix = scale / (double)N;
    size_t next_root = 1; // inv_seq_[1..N-1] in the order the stages consume them
    uint32_t gap = 1;
    for (uint32_t groups = N >> 1; groups > 1; groups >>= 1, gap <<= 1)
      for (uint32_t g = 0; g < groups; g++) {
        const std::complex<double> w = inv_seq_[next_root++];
        std::complex<double> *lo = c.data() + 2 * (size_t)g * gap, *hi = lo + gap;
        for (uint32_t j = 0; j < gap; j++) {
          const std::complex<double> u = lo[j], v = hi[j];
          lo[j] = u + v;
          hi[j] = cmul(u - v, w);
        }
      }
    { // final stage: one group, gap = N/2, scaling folded in
      const std::complex<double> w = inv_seq_[next_root], ws(w.real() * fix, w.imag() * fix);
      std::complex<double> *lo = c.data(), *hi = lo + gap;
      for (uint32_t j = 0; j < gap; j++) {
        const std::complex<double> u = lo[j], v = hi[j], s = u + v;
        lo[j] = std::complex<double>(s.real() * fix, s.imag() * fix);
        hi[j] = cmul(u - v, ws);
      }
    }
    double max_coeff = 0;
    for (uint32_t j = 0; j < N; j++) max_coeff = std::max(max_coeff, std::fabs(c[j].real()));
    int bitcount = (int)std::ceil(std::log2(std::max(max_coeff, 1.0))) + 1;
    if (bitcount >= total_bits[limbs]) throw std::invalid_argument("encoded values are too large");
    for (uint32_t j = 0; j < N; j++) residues_of(std::round(c[j].real()), limbs, out + j, N);
  }
  // residues of round(c*scale) per limb: the encoding of a uniform constant (every NTT slot)
  void encode_uniform(double value, double scale, uint32_t limbs, u64 *out) const {
    double v = std::round(value * scale);
    int bitcount = (int)std::ceil(std::log2(std::max(std::fabs(v), 1.0))) + 1;
    if (bitcount >= total_bits[limbs]) throw std::invalid_argument("encoded values are too large");
    residues_of(v, limbs, out, 1);
  }
  // coefficient-form plaintext [limbs][N] (already INTT'd) -> N/2 slot values
  void decode_coeff(const u64 *coeff, uint32_t limbs, double scale, std::vector<double> &out) const {
    std::vector<std::complex<double>> c(N);
    // mixed-radix (Garner) composition per coefficient, exact multiword compare against Q/2
    std::vector<std::vector<u64>> prefix(limbs); // prefix[i] = q_0...q_{i-1} as multiword
    prefix[0] = {1};
    for (uint32_t i = 1; i < limbs; i++) prefix[i] = mul_small(prefix[i - 1], primes[i - 1]);
    std::vector<u64> Q = mul_small(prefix[limbs - 1], primes[limbs - 1]);
    std::vector<u64> halfQ = shr1(Q);
    // inv_prefix[i] = (q_0...q_{i-1})^-1 mod q_i ; pre_mod[i][j] = q_0..q_{j-1} mod q_i
    std::vector<u64> inv_prefix(limbs, 1);
    std::vector<std::vector<u64>> pre_mod(limbs);
    for (uint32_t i = 0; i < limbs; i++) {
      u64 acc = 1 % primes[i];
      pre_mod[i].resize(i + 1);
      for (uint32_t j = 0; j < i; j++) {
        pre_mod[i][j] = acc;
        acc = evah::mulmod(acc, primes[j] % primes[i], primes[i]);
      }
      pre_mod[i][i] = acc;
      inv_prefix[i] = evah::invmod(acc, primes[i]);
    }
    std::vector<u64> v(limbs), x;
    const double inv_scale = 1.0 / scale, two_pow_64 = std::pow(2.0, 64);
    for (uint32_t j = 0; j < N; j++) {
      for (uint32_t i = 0; i < limbs; i++) {
        u64 qi = primes[i], acc = 0;
        for (uint32_t t = 0; t < i; t++) acc = evah::addmod(acc, evah::mulmod(v[t] % qi, pre_mod[i][t], qi), qi);
        u64 r = coeff[(size_t)i * N + j];
        v[i] = evah::mulmod(evah::submod(r, acc, qi), inv_prefix[i], qi);
      }
      x.assign(Q.size() + 1, 0);
      for (uint32_t i = 0; i < limbs; i++) add_mul(x, prefix[i], v[i]);
      x.resize(Q.size());
      // SEAL 3.6 CKKSEncoder::decode_internal: the base-2^64 words of the composed coefficient go into
      // ONE double, least significant first, with 1/scale folded into the running power of 2^64; a
      // coefficient above Q/2 is negative and is accumulated as signed per-word differences against the
      // words of Q.  `limbs` words per coefficient (SEAL's coeff_modulus_size), high ones zero.
      const bool negative = cmp(x, halfQ) > 0; // x >= upper_half_threshold = (Q + 1) / 2
      double acc = 0.0, scaled = inv_scale;
      for (uint32_t w = 0; w < limbs; w++, scaled *= two_pow_64) {
        const u64 xw = w < x.size() ? x[w] : 0, qw = w < Q.size() ? Q[w] : 0;
        if (!negative) {
          acc += xw ? (double)xw * scaled : 0.0;
        } else if (xw > qw) {
          const u64 diff = xw - qw;
          acc += diff ? (double)diff * scaled : 0.0;
        } else {
          const u64 diff = qw - xw;
          acc -= diff ? (double)diff * scaled : 0.0;
        }
      }
      c[j] = acc;
    }
    // forward special FFT (Cooley-Tukey, zeta^br(m+g))
    for (uint32_t mm = 1, gap = N >> 1; mm < N; mm <<= 1, gap >>= 1)
      for (uint32_t g = 0; g < mm; g++) {
        const std::complex<double> w = roots_[mm + g];
        std::complex<double> *a = c.data() + 2 * (size_t)g * gap, *b = a + gap;
        for (uint32_t jj = 0; jj < gap; jj++) {
          std::complex<double> u = a[jj], t = b[jj] * w;
          a[jj] = u + t;
          b[jj] = u - t;
        }
      }
    out.resize(N >> 1);
    for (uint32_t i = 0; i < (N >> 1); i++) out[i] = c[slot_map_[i]].real();
  }

  // ---- sampling
  void sample_ternary(SecureRng &rng, std::vector<int8_t> &out) const {
    out.resize(N);
    std::uniform_int_distribution<int> d(-1, 1);
    for (auto &v : out) v = (int8_t)d(rng);
  }
  // centered binomial, 21 + 21 bits: sigma ~ 3.24 (SEAL sample_poly_cbd)
  void sample_error(SecureRng &rng, std::vector<int8_t> &out) const {
    out.resize(N);
    for (auto &v : out) {
      u64 r = rng();
      v = (int8_t)(__builtin_popcountll(r & 0x1FFFFF) - __builtin_popcountll((r >> 21) & 0x1FFFFF));
    }
  }
  void small_to_ntt(const std::vector<int8_t> &s, uint32_t prime_idx, u64 *out) const {
    u64 q = primes[prime_idx];
    for (uint32_t j = 0; j < N; j++) out[j] = s[j] < 0 ? q - (u64)(-s[j]) : (u64)s[j];
    ntt(prime_idx, out);
  }
  void sample_uniform(SecureRng &rng, uint32_t prime_idx, u64 *out) const {
    u64 q = primes[prime_idx];
    u64 lim = ~(u64)0 - (~(u64)0 % q) - 1; // rejection bound
    for (uint32_t j = 0; j < N; j++) {
      u64 r;
      do r = rng(); while (r > lim);
      out[j] = r % q;
    }
  }

private:
  std::vector<std::vector<u64>> pow2_;
  std::vector<uint32_t> slot_map_;
  std::vector<std::complex<double>> roots_;   // roots_[m+g] = zeta^br(m+g), zeta = exp(2 pi i / 2N)
  std::vector<std::complex<double>> inv_seq_; // inverse-transform roots in consumption order

  void init_encoder() {
    const uint32_t slots = N >> 1, m = 2 * N;
    slot_map_.resize(N);
    u64 pos = 1;
    for (uint32_t i = 0; i < slots; i++) {
      slot_map_[i] = evah::bitrev((uint32_t)((pos - 1) >> 1), logN);
      slot_map_[slots + i] = evah::bitrev((uint32_t)((m - pos - 1) >> 1), logN);
      pos = (pos * 3) & (m - 1);
    }
    evah::CkksRoots r = evah::ckks_roots(N); // SEAL ComplexRoots doubles (hostmath.h)
    roots_ = std::move(r.fwd);
    inv_seq_ = std::move(r.inv_seq);
  }

  // ---- tiny multiword helpers (little-endian u64 words)
  static std::vector<u64> mul_small(const std::vector<u64> &a, u64 b) {
    std::vector<u64> r(a.size());
    u64 carry = 0;
    for (size_t i = 0; i < a.size(); i++) {
      u128 t = (u128)a[i] * b + carry;
      r[i] = (u64)t;
      carry = (u64)(t >> 64);
    }
    if (carry) r.push_back(carry);
    return r;
  }
  static void add_mul(std::vector<u64> &acc, const std::vector<u64> &a, u64 b) { // acc += a*b
    u64 carry = 0;
    size_t i = 0;
    for (; i < a.size(); i++) {
      u128 t = (u128)a[i] * b + acc[i] + carry;
      acc[i] = (u64)t;
      carry = (u64)(t >> 64);
    }
    for (; carry && i < acc.size(); i++) {
      u128 t = (u128)acc[i] + carry;
      acc[i] = (u64)t;
      carry = (u64)(t >> 64);
    }
  }
  static std::vector<u64> shr1(const std::vector<u64> &a) {
    std::vector<u64> r(a.size());
    for (size_t i = 0; i < a.size(); i++) r[i] = (a[i] >> 1) | (i + 1 < a.size() ? a[i + 1] << 63 : 0);
    return r;
  }
  static int cmp(const std::vector<u64> &a, const std::vector<u64> &b) {
    for (size_t i = std::max(a.size(), b.size()); i-- > 0;) {
      u64 x = i < a.size() ? a[i] : 0, y = i < b.size() ? b[i] : 0;
      if (x != y) return x > y ? 1 : -1;
    }
    return 0;
  }
  static std::vector<u64> sub(const std::vector<u64> &a, const std::vector<u64> &b) { // a - b, a >= b
    std::vector<u64> r(a.size());
    u64 borrow = 0;
    for (size_t i = 0; i < a.size(); i++) {
      u64 y = i < b.size() ? b[i] : 0;
      u128 t = (u128)a[i] - y - borrow;
      r[i] = (u64)t;
      borrow = (t >> 64) ? 1 : 0;
    }
    return r;
  }
  static double to_double(const std::vector<u64> &a) {
    double d = 0;
    for (size_t i = a.size(); i-- > 0;) d = d * 18446744073709551616.0 + (double)a[i];
    return d;
  }
};

// Key material in the layout libeva_hip.so expects.
struct SwitchKey {
  uint32_t n_digits = 0;
  std::vector<u64> data; // [digit][2][k][N], NTT form
};

struct SecretKey {
  std::vector<int8_t> s;       // ternary coefficients
  std::vector<u64> s_ntt;      // [k][N]
};
struct PublicKey {
  std::vector<u64> data; // [2][k][N], NTT form: (-(a s + e), a)
};

class KeyGenerator {
public:
  const HostContext &cx;
  // two independent ChaCha20 streams: `secret` draws the secret key and the error polynomials,
  // `pub` the uniform polynomials that are published as part of every key — nothing an observer of
  // the keys sees comes from the stream that produced the secret.  seed == 0: both keyed from the
  // OS (getrandom); seed != 0 is the reproducible test hook (csprng.h).
  std::unique_ptr<SecureRng> secret, pub;
  SecretKey sk;
  KeyGenerator(const HostContext &c, uint64_t seed) : cx(c) {
    if (seed) {
      secret = std::make_unique<SecureRng>(seed, 1);
      pub = std::make_unique<SecureRng>(seed, 2);
    } else {
      secret = std::make_unique<SecureRng>();
      pub = std::make_unique<SecureRng>();
    }
    cx.sample_ternary(*secret, sk.s);
    sk.s_ntt.resize((size_t)cx.k * cx.N);
    for (uint32_t i = 0; i < cx.k; i++) cx.small_to_ntt(sk.s, i, sk.s_ntt.data() + (size_t)i * cx.N);
  }
  PublicKey public_key() {
    PublicKey pk;
    pk.data.resize((size_t)2 * cx.k * cx.N);
    encrypt_zero_symmetric(pk.data.data(), pk.data.data() + (size_t)cx.k * cx.N);
    return pk;
  }
  // key-switch key from s' (NTT form over all k primes) to s: digit J carries P * s' in limb J
  SwitchKey switch_key(const std::vector<u64> &sprime_ntt) {
    const uint32_t N = cx.N, k = cx.k, D = k - 1;
    SwitchKey key;
    key.n_digits = D;
    key.data.resize((size_t)D * 2 * k * N);
    const u64 P = cx.primes[k - 1];
    for (uint32_t J = 0; J < D; J++) {
      u64 *c0 = key.data.data() + (size_t)J * 2 * k * N, *c1 = c0 + (size_t)k * N;
      encrypt_zero_symmetric(c0, c1);
      const u64 q = cx.primes[J], f = P % q;
      u64 *dst = c0 + (size_t)J * N;
      const u64 *sp = sprime_ntt.data() + (size_t)J * N;
      for (uint32_t j = 0; j < N; j++) dst[j] = evah::addmod(dst[j], cx.mulm(sp[j], f, J), q);
    }
    return key;
  }
  SwitchKey relin_key() {
    std::vector<u64> s2((size_t)cx.k * cx.N);
    for (uint32_t i = 0; i < cx.k; i++)
      for (uint32_t j = 0; j < cx.N; j++) {
        u64 v = sk.s_ntt[(size_t)i * cx.N + j];
        s2[(size_t)i * cx.N + j] = cx.mulm(v, v, i);
      }
    return switch_key(s2);
  }
  SwitchKey galois_key(uint32_t elt) {
    // s(X^elt) in NTT form = permutation of s_ntt (same table as the device uses)
    std::vector<u64> sp((size_t)cx.k * cx.N);
    for (uint32_t j = 0; j < cx.N; j++) {
      uint32_t reversed = evah::bitrev(cx.N + j, cx.logN + 1);
      u64 raw = (((u64)elt * reversed) >> 1) & (u64)(cx.N - 1);
      uint32_t src = evah::bitrev((uint32_t)raw, cx.logN);
      for (uint32_t i = 0; i < cx.k; i++) sp[(size_t)i * cx.N + j] = sk.s_ntt[(size_t)i * cx.N + src];
    }
    return switch_key(sp);
  }

private:
  // (c0, c1) = (-(a s + e), a) over all k primes, NTT form
  void encrypt_zero_symmetric(u64 *c0, u64 *c1) {
    const uint32_t N = cx.N;
    std::vector<int8_t> e;
    cx.sample_error(*secret, e);
    std::vector<u64> en(N);
    for (uint32_t i = 0; i < cx.k; i++) {
      const u64 q = cx.primes[i];
      u64 *a = c1 + (size_t)i * N, *b = c0 + (size_t)i * N;
      cx.sample_uniform(*pub, i, a);
      cx.small_to_ntt(e, i, en.data());
      const u64 *s = sk.s_ntt.data() + (size_t)i * N;
      for (uint32_t j = 0; j < N; j++) b[j] = evah::negmod(evah::addmod(cx.mulm(a[j], s[j], i), en[j], q), q);
    }
  }
};

// Host ciphertext / plaintext values (what encrypt() hands to execute() and execute() to decrypt())
// Ciphertext words live in pinned host memory when the device library can provide it
// (evah_host_alloc): they are what execute() uploads and downloads, and DMA from pinned pages runs
// at PCIe rate.  Without a device the allocator is plain operator new.
template <class T> struct HostAlloc {
  using value_type = T;
  HostAlloc() = default;
  template <class U> HostAlloc(const HostAlloc<U> &) {}
  T *allocate(size_t n) {
    // 64-byte header in front of the data remembers where the block came from
    const size_t bytes = n * sizeof(T) + 64;
    char *base = static_cast<char *>(evah_host_alloc(bytes));
    const bool pinned = base != nullptr;
    if (!base) base = static_cast<char *>(::operator new(bytes));
    *reinterpret_cast<uint64_t *>(base) = pinned ? 0x50494e4eull : 0x48454150ull;
    return reinterpret_cast<T *>(base + 64);
  }
  void deallocate(T *p, size_t) {
    char *base = reinterpret_cast<char *>(p) - 64;
    if (*reinterpret_cast<uint64_t *>(base) == 0x50494e4eull) evah_host_free(base);
    else ::operator delete(base);
  }
  // resize() leaves new words uninitialised (every user overwrites them: downloads, encrypt): a
  // value-initialising resize of a 256-instance output batch would memset 134 MB on the host
  template <class U, class... A> void construct(U *p, A &&...a) {
    if constexpr (sizeof...(A) == 0) ::new ((void *)p) U;
    else ::new ((void *)p) U(std::forward<A>(a)...);
  }
  template <class U> bool operator==(const HostAlloc<U> &) const { return true; }
  template <class U> bool operator!=(const HostAlloc<U> &) const { return false; }
};
using CipherWords = std::vector<u64, HostAlloc<u64>>;
// A ciphertext value of a valuation.  It may live on the host (data), on the device (dev: a handle
// of a device context, executor.h) or both: SURVEY.md 8(b) keeps the reference's SEALValuation
// opaque so that it "may hold device handles", and encrypt -> execute -> decrypt then never moves a
// ciphertext over PCIe.  The host words are filled on demand (words(): a download) — by get(),
// save(), a context on another device, or the host-only code paths.
struct DeviceResident;
struct HostCipher {
  uint32_t size = 0, limbs = 0;
  double scale = 1.0;
  mutable CipherWords data; // [size][limbs][N]; empty while the value lives only on the device
  std::shared_ptr<DeviceResident> dev;
  mutable bool words_checked = false; // every word < its prime: verified (values from files / Python) or by construction
};
struct HostPlain {
  uint32_t limbs = 0;
  double scale = 1.0;
  std::vector<u64> data; // [limbs][N], NTT form
  mutable bool words_checked = false;
};

// Public-key encryption of an NTT-form plaintext at `limbs` data limbs (A.10): encrypt zero one
// level up (limbs+1 primes), divide-and-round by that extra prime, add the plaintext to c0.
inline HostCipher encrypt(const HostContext &cx, const PublicKey &pk, const HostPlain &pt, SecureRng &rng) {
  const uint32_t N = cx.N, l = pt.limbs, up = l + 1;
  if (up > cx.k) throw std::invalid_argument("plaintext level is not valid for encryption");
  std::vector<int8_t> u, e0, e1;
  cx.sample_ternary(rng, u);
  cx.sample_error(rng, e0);
  cx.sample_error(rng, e1);
  std::vector<u64> c((size_t)2 * up * N), un(N), en(N);
  for (uint32_t i = 0; i < up; i++) {
    const u64 q = cx.primes[i];
    cx.small_to_ntt(u, i, un.data());
    for (uint32_t K = 0; K < 2; K++) {
      cx.small_to_ntt(K ? e1 : e0, i, en.data());
      const u64 *p = pk.data.data() + ((size_t)K * cx.k + i) * N;
      u64 *dst = c.data() + ((size_t)K * up + i) * N;
      for (uint32_t j = 0; j < N; j++) dst[j] = evah::addmod(cx.mulm(p[j], un[j], i), en[j], q);
    }
  }
  // divide and round by the last of the `up` primes (same rule as rescale, A.5)
  HostCipher out;
  out.size = 2;
  out.limbs = l;
  out.scale = pt.scale;
  out.data.resize((size_t)2 * l * N);
  const uint32_t last = up - 1;
  const u64 ql = cx.primes[last], half = ql >> 1;
  std::vector<u64> t(N), w(N);
  for (uint32_t K = 0; K < 2; K++) {
    std::copy_n(c.data() + ((size_t)K * up + last) * N, N, t.data());
    cx.intt(last, t.data());
    for (uint32_t j = 0; j < N; j++) t[j] = evah::addmod(t[j], half, ql);
    for (uint32_t i = 0; i < l; i++) {
      const u64 q = cx.primes[i], hq = half % q, inv = evah::invmod(ql % q, q);
      for (uint32_t j = 0; j < N; j++) w[j] = evah::submod(t[j] % q, hq, q);
      cx.ntt(i, w.data());
      const u64 *src = c.data() + ((size_t)K * up + i) * N;
      u64 *dst = out.data.data() + ((size_t)K * l + i) * N;
      for (uint32_t j = 0; j < N; j++) {
        u64 v = cx.mulm(evah::submod(src[j], w[j], q), inv, i);
        dst[j] = K == 0 ? evah::addmod(v, pt.data[(size_t)i * N + j], q) : v;
      }
    }
  }
  return out;
}

// m = c0 + c1 s (+ c2 s^2), NTT form -> coefficient form per limb
inline std::vector<u64> decrypt_to_coeff(const HostContext &cx, const SecretKey &sk, const HostCipher &ct) {
  const uint32_t N = cx.N, l = ct.limbs;
  std::vector<u64> m((size_t)l * N);
  for (uint32_t i = 0; i < l; i++) {
    const u64 q = cx.primes[i];
    const u64 *s = sk.s_ntt.data() + (size_t)i * N;
    u64 *dst = m.data() + (size_t)i * N;
    for (uint32_t j = 0; j < N; j++) {
      u64 acc = ct.data[((size_t)0 * l + i) * N + j], sp = s[j];
      for (uint32_t p = 1; p < ct.size; p++) {
        acc = evah::addmod(acc, cx.mulm(ct.data[((size_t)p * l + i) * N + j], sp, i), q);
        sp = cx.mulm(sp, s[j], i);
      }
      dst[j] = acc;
    }
    cx.intt(i, dst);
  }
  return m;
}

} // namespace evahost
