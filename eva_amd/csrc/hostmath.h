// hostmath.h — host-side number theory for the MI355X CKKS backend (product code, C++17).
//
// Builds the things the device tables are made of: the SEAL-conformant prime chain
// (CoeffModulus::Create, reference call site /root/reference/eva/seal/seal.cpp:181-182), the
// minimal primitive 2N-th root per prime, bit-reversed root-power tables with Shoup quotients.
// Shared by libeva_hip.so (table upload) and the host module (keygen / encrypt / decrypt).
#pragma once
#include <cmath>
#include <complex>
#include <cstdint>
#include <stdexcept>
#include <vector>

namespace evah {

using u64 = unsigned long long;  // same type as the device code (ulonglong2 members)
using u128 = unsigned __int128;

inline u64 mulmod(u64 a, u64 b, u64 q) { return (u64)(((u128)a * b) % q); }
inline u64 addmod(u64 a, u64 b, u64 q) { u64 s = a + b; return s >= q ? s - q : s; }
inline u64 submod(u64 a, u64 b, u64 q) { return a >= b ? a - b : a + q - b; }
inline u64 negmod(u64 a, u64 q) { return a ? q - a : 0; }

inline u64 powmod(u64 a, u64 e, u64 q) {
  u64 r = 1 % q;
  a %= q;
  for (; e; e >>= 1) {
    if (e & 1) r = mulmod(r, a, q);
    a = mulmod(a, a, q);
  }
  return r;
}
inline u64 invmod(u64 a, u64 q) { return powmod(a % q, q - 2, q); }  // q prime
inline u64 shoup(u64 w, u64 q) { return (u64)(((u128)w << 64) / q); }

inline uint32_t bitrev(uint32_t x, uint32_t bits) {
  uint32_t r = 0;
  for (uint32_t i = 0; i < bits; i++) r |= ((x >> i) & 1u) << (bits - 1 - i);
  return r;
}
inline uint32_t ilog2(uint32_t n) {
  uint32_t l = 0;
  while ((1u << l) < n) l++;
  return l;
}

inline bool is_prime(u64 n) {
  static const u64 bases[] = {2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37};
  if (n < 2) return false;
  for (u64 b : bases) {
    if (n == b) return true;
    if (n % b == 0) return false;
  }
  u64 d = n - 1;
  int r = 0;
  while (!(d & 1)) { d >>= 1; r++; }
  for (u64 b : bases) {
    u64 x = powmod(b, d, n);
    if (x == 1 || x == n - 1) continue;
    bool comp = true;
    for (int j = 1; j < r && comp; j++) {
      x = mulmod(x, x, n);
      if (x == n - 1) comp = false;
    }
    if (comp) return false;
  }
  return true;
}

// NTT primes of the given bit size for degree N, descending from 2^b - 2N + 1 in steps of 2N.
inline std::vector<u64> ntt_primes_descending(uint32_t N, int bits, size_t count) {
  std::vector<u64> out;
  u64 step = 2ull * N;
  u64 v = (((u64)1 << bits) - 1) / step * step + 1, lower = (u64)1 << (bits - 1);
  for (; out.size() < count && v > lower; v -= step)
    if (is_prime(v)) out.push_back(v);
  if (out.size() != count) throw std::logic_error("failed to find enough qualifying primes");
  return out;
}

// Prime chain for (N, bit sizes): equal sizes share one descending list; list order takes
// from the back, so the first occurrence of a size receives its smallest prime.
inline std::vector<u64> coeff_modulus_create(uint32_t N, const std::vector<int> &bit_sizes) {
  size_t cnt[64] = {0}, used[64] = {0};
  std::vector<u64> table[64];
  for (int b : bit_sizes) {
    if (b < 2 || b > 60) throw std::invalid_argument("bit_sizes is invalid"); // SEAL_USER_MOD_BIT_COUNT_MAX = 60
    cnt[b]++;
  }
  for (int b = 2; b <= 60; b++)
    if (cnt[b]) table[b] = ntt_primes_descending(N, b, cnt[b]);
  std::vector<u64> out;
  for (int b : bit_sizes) out.push_back(table[b][cnt[b] - 1 - used[b]++]);
  return out;
}

// numerically smallest primitive 2N-th root of unity mod q
inline u64 minimal_primitive_root(uint32_t N, u64 q) {
  u64 deg = 2ull * N;
  if ((q - 1) % deg) throw std::invalid_argument("prime is not 1 mod 2N");
  u64 e = (q - 1) / deg, root = 0;
  for (u64 g = 2; g < q; g++) {
    u64 r = powmod(g, e, q);
    if (powmod(r, N, q) == q - 1) { root = r; break; }
  }
  u64 sq = mulmod(root, root, q), cur = root, best = root;
  for (u64 i = 0; i < N; i++) {
    if (cur < best) best = cur;
    cur = mulmod(cur, sq, q);
  }
  return best;
}

// rp[bitrev(i)] = psi^i, i < N  (heap order: stage m group i uses rp[m+i])
inline std::vector<u64> root_power_table(uint32_t N, u64 q, u64 psi) {
  std::vector<u64> rp(N);
  uint32_t logN = ilog2(N);
  u64 p = 1;
  for (uint32_t i = 0; i < N; i++) {
    rp[bitrev(i, logN)] = p;
    p = mulmod(p, psi, q);
  }
  return rp;
}

// ---- complex roots of the CKKS encoder (FP64).  The encoder's bits depend on the exact doubles
// used as roots, so they are produced the way SEAL 3.6's util::ComplexRoots produces them: the
// first octant of the 2N-th roots from cos/sin(2*pi*i/2N), all others through the 8-fold
// symmetry.  Only indices <= N (the upper half plane) are ever needed.
struct CkksRoots {
  std::vector<std::complex<double>> fwd;     // fwd[j] = zeta^br(j), j < N (forward special FFT, heap order)
  std::vector<std::complex<double>> inv_seq; // inv_seq[i] = conj(zeta^(br(i-1)+1)), 1 <= i < N: the
                                             // inverse transform's roots in the order its stages use them
};
inline CkksRoots ckks_roots(uint32_t N) {
  const uint32_t logN = ilog2(N);
  const size_t deg = 2 * (size_t)N, oct_n = deg / 8;
  std::vector<std::complex<double>> oct(oct_n + 1);
  const double pi = 3.1415926535897932384626433832795028842;
  // cos and sin through volatile pointers: GCC would fuse the pair into one sincos() call, and
  // glibc's sincos() is not bit-identical to sin() (2N = 8192, i = 487 differs in the last place);
  // the reference recommends clang (README.md:21-25), which makes the two separate libm calls
  double (*volatile libm_cos)(double) = static_cast<double (*)(double)>(std::cos);
  double (*volatile libm_sin)(double) = static_cast<double (*)(double)>(std::sin);
  for (size_t i = 0; i <= oct_n; i++) {
    const double theta = 2 * pi * (double)i / (double)deg;
    oct[i] = std::complex<double>(libm_cos(theta), libm_sin(theta));
  }
  auto first_quadrant = [&](size_t idx) { // idx <= deg/4
    if (idx <= oct_n) return oct[idx];
    const std::complex<double> t = oct[deg / 4 - idx];
    return std::complex<double>(t.imag(), t.real());
  };
  auto root = [&](size_t idx) { // idx <= deg/2
    if (idx <= deg / 4) return first_quadrant(idx);
    return -std::conj(first_quadrant(deg / 2 - idx));
  };
  CkksRoots r;
  r.fwd.resize(N);
  r.inv_seq.resize(N);
  r.fwd[0] = 1.0;
  for (uint32_t j = 1; j < N; j++) {
    r.fwd[j] = root(bitrev(j, logN));
    r.inv_seq[j] = std::conj(root((size_t)bitrev(j - 1, logN) + 1));
  }
  return r;
}

}  // namespace evah
