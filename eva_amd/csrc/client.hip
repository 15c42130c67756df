// client.hip — the host-side neighbours of execute() on the device:
// public-key encryption of an encoded plaintext and decryption + decoding of a result, i.e. the
// arithmetic of SEALPublic::encrypt and SEALSecret::decrypt (/root/reference/eva/seal/seal.cpp:24-102,
// 124-146: encoder.encode + encryptor.encrypt; decryptor.decrypt + encoder.decode).  Randomness stays on the host (csprng.h): the sampled small polynomials
// travel as int8 arrays (3 N bytes per encryption); everything of size N log N or l N runs here.
//   encrypt  : c = (pk0 u + e0, pk1 u + e1) at l+1 limbs, divided-and-rounded by the extra prime
//              (same rule as rescale A.5), plus the plaintext on c0
//   decrypt  : m = c0 + c1 s (+ c2 s^2) per limb, inverse transform, exact recomposition to base-2^64
//              words (mixed-radix digits first), the words to one double in SEAL 3.6's order with
//              1/scale folded in and the sign taken against (Q+1)/2, forward special FFT
//              (CKKSEncoder::decode_internal), slot values out.  FP64 with SEAL's operation order and no
//              FMA contraction: the doubles are those of the oracle's evo_decode and of the host
//              decoder, bit for bit (tests/test_decode_parity.py)

#include "launch.hip.h"

namespace evah {

__global__ void __launch_bounds__(256)
k_small_to_residues(DevCtx cx, const int8_t *small, uint32_t n_polys, uint32_t limbs, u64 *out) {
  // out[p][i][n] = small[p][n] mod primes[i] (negative -> q - |v|)
  const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y, p = blockIdx.z;
  (void)n_polys;
  const int v = small[(size_t)p * cx.N + n];
  const u64 q = cx.primes[cx.prime_of(i)].q;
  out[((size_t)p * limbs + i) * cx.N + n] = v < 0 ? q - (u64)(-v) : (u64)v;
}
// c[K][i] = pk[K][i] * u[i] + e_K[i]; small = NTT forms [3][up][N] of (u, e0, e1); pk [2][k][N]
__global__ void __launch_bounds__(256)
k_encrypt_zero(DevCtx cx, const u64 *pk, const u64 *small, uint32_t up, u64 *c) {
  const size_t n = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t i = blockIdx.y, K = blockIdx.z;
  const DevPrime pm = cx.primes[i];
  const u64 u = small[(size_t)i * cx.N + n], e = small[((size_t)(1 + K) * up + i) * cx.N + n];
  const u64 p = pk[((size_t)K * cx.k + i) * cx.N + n];
  c[((size_t)K * up + i) * cx.N + n] = addmod(mulmod(p, u, pm), e, pm.q);
}
// m[i] = c0 + c1 s + c2 s^2 (size 2 or 3; a size-1 value is its own message)
__global__ void __launch_bounds__(256)
k_decrypt_dot(DevCtx cx, const u64 *ct, size_t ps, uint32_t size, const u64 *sk, u64 *m) {
  const size_t n = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t i = blockIdx.y;
  const DevPrime pm = cx.primes[i];
  const size_t off = (size_t)i * cx.N + n;
  const u64 s = sk[off];
  u64 acc = ct[off], sp = s;
  for (uint32_t p = 1; p < size; p++) {
    acc = addmod(acc, mulmod(ct[p * ps + off], sp, pm), pm.q);
    sp = mulmod(sp, s, pm);
  }
  m[off] = acc;
}
// Garner tables of a level: inv_prefix[i] = (q_0..q_{i-1})^-1 mod q_i, pre_mod[i][t] = q_0..q_{t-1} mod q_i,
// prefix[i][w] = word w of q_0..q_{i-1} (base 2^64, l words), qwords[w] / half[w] = word w of Q / of floor(Q/2)
struct CrtTab {
  const u64 *inv_prefix, *pre_mod, *prefix, *qwords, *half;
};
__device__ __forceinline__ void garner(const DevCtx &cx, const CrtTab &t, uint32_t l, const u64 *r, u64 *v) {
  for (uint32_t i = 0; i < l; i++) {
    const DevPrime pm = cx.primes[i];
    u128_t acc = {0, 0};
    for (uint32_t j = 0; j < i; j++) acc128(acc, v[j] >= pm.q ? barrett64(v[j], pm.q, pm.brt) : v[j], t.pre_mod[i * l + j]);
    const u64 a = barrett128(acc, pm);
    v[i] = i ? mulmod(submod(r[i], a, pm.q), t.inv_prefix[i], pm) : r[0];
  }
}
// SEAL 3.6 CKKSEncoder::decode_internal between the inverse NTTs and the FFT: the composed coefficient
// x in [0, Q) as l base-2^64 words (here from the mixed-radix digits: x = sum_i v_i q_0..q_{i-1}, exact),
// then ONE double from the words, least significant first, with inv_scale folded into the running power
// of 2^64; x >= (Q + 1) / 2 is negative and accumulates the signed per-word differences against Q's
// words.  Same operations in the same order as the oracle's evo_decode and the host decoder: same doubles.
__global__ void __launch_bounds__(256)
k_crt_to_double(DevCtx cx, CrtTab t, uint32_t l, const u64 *coeff, double inv_scale, double2 *out) {
#pragma clang fp contract(off)
  const size_t n = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  u64 r[62], v[62], x[63];
  for (uint32_t i = 0; i < l; i++) r[i] = coeff[(size_t)i * cx.N + n];
  garner(cx, t, l, r, v);
  for (uint32_t w = 0; w <= l; w++) x[w] = 0;
  for (uint32_t i = 0; i < l; i++) { // x += v_i * prefix_i (prefix_i has at most i words; the sum stays below Q)
    u64 carry = 0;
    const u64 *pf = t.prefix + (size_t)i * l;
    for (uint32_t w = 0; w < l; w++) {
      u128_t p = mul128(pf[w], v[i]);
      const u64 lo = p.lo + carry;
      u64 hi = p.hi + (lo < carry);
      const u64 sum = x[w] + lo;
      hi += (sum < lo);
      x[w] = sum;
      carry = hi;
    }
  }
  bool negative = false; // x > floor(Q/2), compared from the most significant word
  for (int w = (int)l - 1; w >= 0; w--)
    if (x[w] != t.half[w]) { negative = x[w] > t.half[w]; break; }
  const double two_pow_64 = 18446744073709551616.0;
  double acc = 0.0, scaled = inv_scale;
  for (uint32_t w = 0; w < l; w++, scaled *= two_pow_64) {
    const u64 xw = x[w], qw = t.qwords[w];
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
  out[n] = make_double2(acc, 0.0);
}
// forward special FFT stage (Cooley-Tukey): group g of `groups` uses roots[groups + g]
// (DWTHandler::transform_to_rev of SEAL 3.6: x = u + v r, y = u - v r; the complex product as four rounded
// multiplies, a rounded difference and a rounded sum — no FMA contraction, as in the encoder)
__global__ void __launch_bounds__(256)
k_dec_fft_stage(double2 *c, const double2 *roots, uint32_t groups, uint32_t log_gap, uint32_t half_n) {
#pragma clang fp contract(off)
  const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= half_n) return;
  const uint32_t gap = 1u << log_gap, g = idx >> log_gap, j = idx & (gap - 1);
  const uint32_t a = 2 * g * gap + j, b = a + gap;
  const double2 w = roots[groups + g], u = c[a], y = c[b];
  const double ac = y.x * w.x, bd = y.y * w.y, ad = y.x * w.y, bc = y.y * w.x;
  const double tx = ac - bd, ty = ad + bc;
  c[a] = make_double2(u.x + tx, u.y + ty);
  c[b] = make_double2(u.x - tx, u.y - ty);
}
__global__ void __launch_bounds__(256)
k_dec_gather(const double2 *c, const uint32_t *slot_map, uint32_t n_out, double *out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_out) out[i] = c[slot_map[i]].x;
}

} // namespace evah

extern "C" {

// the public key [2][k][N] / the secret key in NTT form [k][N] (client side), resident like the other keys
int evah_client_key_upload(evah_ctx *c, int kind, const uint64_t *data) {
  API_BEGIN
  use(c);
  if (kind != EVAH_KEY_PUBLIC && kind != EVAH_KEY_SECRET) throw std::invalid_argument("unknown key kind");
  KeyDev kd;
  kd.n_digits = 1;
  kd.bytes = sizeof(u64) * (size_t)(kind == EVAH_KEY_PUBLIC ? 2 : 1) * c->k * c->N;
  HIPCHK(hipMalloc(&kd.d, kd.bytes));
  h2d_now(c, kd.d, data, kd.bytes);
  KeyDev &slot = kind == EVAH_KEY_PUBLIC ? c->sh->pk : c->sh->sk;
  if (slot.d) {
    if (kind == EVAH_KEY_SECRET) (void)hipMemset(slot.d, 0, slot.bytes); // no key material in freed HBM
    (void)hipFree(slot.d);
  }
  slot = kd;
  API_END
}

// SEAL Encryptor::encrypt of an NTT-form plaintext with the caller's randomness: small = (u ternary,
// e0, e1 error polynomials) as int8 [3][N]
int evah_encrypt(evah_ctx *c, const evah_pt *pt, const int8_t *small, evah_ct **out) {
  API_BEGIN
  use(c);
  if (c->capturing) throw std::logic_error("this call synchronises with the host and cannot be captured into a graph");
  if (!c->sh->pk.d) throw std::invalid_argument("public key not present");
  acquire(c, pt->buf);
  const uint32_t l = pt->limbs, up = l + 1;
  if (up > c->k) throw std::invalid_argument("plaintext level is not valid for encryption");
  const size_t N = c->N;
  Scratch sm8(c, (3 * N + 7) / 8), sm(c, (size_t)3 * up * N), ct(c, (size_t)2 * up * N), r(c, 2 * N);
  HIPCHK(hipMemcpyAsync(sm8.d, small, 3 * N, hipMemcpyHostToDevice, c->stream));
  EW_LAUNCH(k_small_to_residues, dim3(c->N / 256, up, 3), dim3(256), 0, c->stream, c->dev, reinterpret_cast<const int8_t *>(sm8.d), 3u, up, sm.d);
  OpPlain::Params fp{sm.d, sm.d, (size_t)up * N, (size_t)up * N, up, 0, 0, {}};
  ntt_forward<OpPlain>(c, fp, 3 * up);
  EW_LAUNCH(k_encrypt_zero, dim3(c->N / 256, up, 2), dim3(256), 0, c->stream, c->dev, c->sh->pk.d, sm.d, up, ct.d);
  HIPCHK(hipGetLastError());
  // divide and round by prime `l` (the last of the up primes), then add the plaintext to c0
  evah_ct *o = ct_new(c, 2, l, pt->scale);
  try {
    OpPlain::Params ip{ct.d + (size_t)l * N, r.d, (size_t)up * N, N, 1, l, 1, {}};
    ntt_inverse<OpPlain>(c, ip, 2);
    OpModDown::Params mp{r.d, N, ct.d, (size_t)up * N, pt->d, 0, 1, o->d, o->ps, l, l};
    ntt_forward<OpModDown>(c, mp, 2 * l);
    HIPCHK(hipStreamSynchronize(c->stream)); // `small` is pageable host memory
  } catch (...) {
    evah_ct_free(c, o);
    throw;
  }
  *out = o;
  API_END
}

// SEAL Decryptor::decrypt + CKKSEncoder::decode: the first n_out slot values of the message of ct
int evah_decrypt_decode(evah_ctx *c, const evah_ct *ct, uint32_t n_out, double *out) {
  API_BEGIN
  use(c);
  if (c->capturing) throw std::logic_error("this call synchronises with the host and cannot be captured into a graph");
  if (!c->sh->sk.d) throw std::invalid_argument("secret key not present");
  if (ct->batch != 1) throw std::invalid_argument("decrypt takes a single ciphertext");
  acquire(c, ct->buf);
  const uint32_t l = ct->limbs, N = c->N, slots = N >> 1;
  if (n_out < 1 || n_out > slots) throw std::invalid_argument("slot count out of range");
  if (l > 61) throw std::invalid_argument("too many limbs");
  check_scale(c, ct->scale, l); // decode_internal: "scale out of bounds"
  enc_tables(c);
  if (!c->sh->dec_roots) { // forward roots zeta^br(j) (hostmath.h), once per context family
    const CkksRoots cr = ckks_roots(N);
    std::vector<double> roots(2 * (size_t)N);
    for (uint32_t j = 0; j < N; j++) { roots[2 * j] = cr.fwd[j].real(); roots[2 * j + 1] = cr.fwd[j].imag(); }
    HIPCHK(hipMalloc(&c->sh->dec_roots, sizeof(double2) * N));
    h2d_now(c, c->sh->dec_roots, roots.data(), sizeof(double2) * N);
  }
  // Garner tables of this level: [inv_prefix l][pre_mod l*l][prefix l*l][Q l][floor(Q/2) l]
  std::vector<u64> tab((size_t)2 * l * l + 3 * l, 0);
  {
    u64 *inv_prefix = tab.data(), *pre_mod = inv_prefix + l, *prefix = pre_mod + (size_t)l * l,
        *qwords = prefix + (size_t)l * l, *half = qwords + l;
    std::vector<u64> w{1}; // q_0..q_{i-1}, little-endian words
    for (uint32_t i = 0; i < l; i++) {
      const u64 qi = c->primes[i];
      u64 acc = 1 % qi;
      for (uint32_t j = 0; j < i; j++) {
        pre_mod[i * l + j] = acc;
        acc = mulmod(acc, c->primes[j] % qi, qi);
      }
      inv_prefix[i] = invmod(acc, qi);
      for (size_t t = 0; t < w.size() && t < l; t++) prefix[(size_t)i * l + t] = w[t];
      u64 carry = 0;
      for (auto &x : w) { u128 t = (u128)x * qi + carry; x = (u64)t; carry = (u64)(t >> 64); }
      if (carry) w.push_back(carry);
    }
    for (size_t t = 0; t < w.size() && t < l; t++) qwords[t] = w[t];
    for (size_t t = 0; t < w.size() && t < l; t++) half[t] = (w[t] >> 1) | (t + 1 < w.size() ? w[t + 1] << 63 : 0);
  }
  Scratch m(c, (size_t)l * N), tabd(c, tab.size()), cbuf(c, 2 * (size_t)N), outd(c, n_out);
  HIPCHK(hipMemcpyAsync(tabd.d, tab.data(), sizeof(u64) * tab.size(), hipMemcpyHostToDevice, c->stream));
  EW_LAUNCH(k_decrypt_dot, dim3(N / 256, l), dim3(256), 0, c->stream, c->dev, ct->d, ct->ps, ct->size, c->sh->sk.d, m.d);
  OpPlain::Params ip{m.d, m.d, 0, 0, l, 0, 0, {}};
  ntt_inverse<OpPlain>(c, ip, l);
  CrtTab t{tabd.d, tabd.d + l, tabd.d + l + (size_t)l * l, tabd.d + l + (size_t)2 * l * l, tabd.d + 2 * l + (size_t)2 * l * l};
  double2 *cd = reinterpret_cast<double2 *>(cbuf.d);
  EW_LAUNCH(k_crt_to_double, dim3(N / 256), dim3(256), 0, c->stream, c->dev, t, l, m.d, 1.0 / ct->scale, cd);
  // the decrypted message and its FP image do not stay behind in pool memory the next call reuses
  HIPCHK(hipMemsetAsync(m.d, 0, sizeof(u64) * (size_t)l * N, c->stream));
  for (uint32_t groups = 1, lg = c->logN - 1; groups < N; groups <<= 1, lg--)
    hipLaunchKernelGGL(k_dec_fft_stage, dim3((slots + 255) / 256), dim3(256), 0, c->stream, cd, c->sh->dec_roots, groups, lg, slots);
  hipLaunchKernelGGL(k_dec_gather, dim3((n_out + 255) / 256), dim3(256), 0, c->stream, cd, c->sh->enc_slot_map, n_out,
                     reinterpret_cast<double *>(outd.d));
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(out, outd.d, sizeof(double) * n_out, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipMemsetAsync(cbuf.d, 0, sizeof(double2) * (size_t)N, c->stream));
  HIPCHK(hipMemsetAsync(outd.d, 0, sizeof(double) * n_out, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  API_END
}

} // extern "C"
