// evaluator.hip — libeva_hip.so: HIP kernels + one C-ABI entry point per seal::Evaluator call made by
// SEALExecutor::operator() (/root/reference/eva/seal/seal_executor.h:279-404), the device CKKS
// encoder (:217-243), and the fused / batched forms the level scheduler uses.  include/eva_hip.h
// names the reference line each entry point replaces.
#include "internal.hip.h"
#include "ntt.hip.h"

namespace evah {

// ------------------------------------------------------------------------------- kernels

// 2 coefficients per thread (16-byte accesses); grid = (N/512, limbs, polys)
#define EW_SETUP                                                                                 \
  const uint32_t p = blockIdx.z, i = blockIdx.y;                                                 \
  const size_t off = (size_t)i * cx.N + 2 * ((size_t)blockIdx.x * blockDim.x + threadIdx.x);    \
  const DevPrime pm = cx.primes[cx.prime_of(i)];                                                 \
  (void)p;                                                                                        \
  (void)pm;

__device__ __forceinline__ ulonglong2 ld2(const u64 *p) { return *reinterpret_cast<const ulonglong2 *>(p); }
__device__ __forceinline__ void st2(u64 *p, ulonglong2 v) { *reinterpret_cast<ulonglong2 *>(p) = v; }

// K1/K2: add / sub / add_plain / sub_plain.  Common polys combined, extra
// polys of the longer operand copied (or negated when it is the subtrahend).
__global__ void __launch_bounds__(256)
k_addsub(DevCtx cx, const u64 *a, size_t a_ps, uint32_t sa, const u64 *b, size_t b_ps, uint32_t sb,
         u64 *out, size_t o_ps, int sub, uint32_t smax) {
  EW_SETUP
  // grid.z = instance * smax + poly (smax = max(sa, sb)); a plaintext operand has b_ps == 0
  const uint32_t inst = p / smax, pp = p % smax;
  const size_t ao = (size_t)(inst * sa + pp) * a_ps + off, bo = (size_t)(inst * sb + pp) * b_ps + off;
  ulonglong2 r;
  if (pp < sa && pp < sb) {
    ulonglong2 x = ld2(a + ao), y = ld2(b + bo);
    r.x = sub ? submod(x.x, y.x, pm.q) : addmod(x.x, y.x, pm.q);
    r.y = sub ? submod(x.y, y.y, pm.q) : addmod(x.y, y.y, pm.q);
  } else if (pp < sa) {
    r = ld2(a + ao);
  } else {
    r = ld2(b + bo);
    if (sub) { r.x = negmod(r.x, pm.q); r.y = negmod(r.y, pm.q); }
  }
  st2(out + p * o_ps + off, r);
}

// K3: negate
__global__ void __launch_bounds__(256)
k_negate(DevCtx cx, const u64 *a, size_t a_ps, u64 *out, size_t o_ps) {
  EW_SETUP
  ulonglong2 r = ld2(a + p * a_ps + off);
  r.x = negmod(r.x, pm.q);
  r.y = negmod(r.y, pm.q);
  st2(out + p * o_ps + off, r);
}

// K4: multiply 2x2 -> 3: (a0b0, a0b1 + a1b0, a1b1); grid.z = instance of a batched handle
__global__ void __launch_bounds__(256)
k_mul22(DevCtx cx, const u64 *a, size_t a_ps, const u64 *b, size_t b_ps, u64 *out, size_t o_ps) {
  EW_SETUP
  a += (size_t)p * 2 * a_ps;
  b += (size_t)p * 2 * b_ps;
  out += (size_t)p * 3 * o_ps;
  ulonglong2 a0 = ld2(a + off), a1 = ld2(a + a_ps + off);
  ulonglong2 b0 = ld2(b + off), b1 = ld2(b + b_ps + off);
  ulonglong2 d0, d1, d2;
  d0.x = mulmod(a0.x, b0.x, pm);
  d0.y = mulmod(a0.y, b0.y, pm);
  d2.x = mulmod(a1.x, b1.x, pm);
  d2.y = mulmod(a1.y, b1.y, pm);
  u128_t t = mul128(a0.x, b1.x);
  acc128(t, a1.x, b0.x);
  d1.x = barrett128(t, pm);
  t = mul128(a0.y, b1.y);
  acc128(t, a1.y, b0.y);
  d1.y = barrett128(t, pm);
  st2(out + off, d0);
  st2(out + o_ps + off, d1);
  st2(out + 2 * o_ps + off, d2);
}

// K4 batched: n independent 2x2 products in one launch; grid.z = instance
__global__ void __launch_bounds__(256)
k_mul22_many(DevCtx cx, MulTab tab, u64 *out_b, size_t o_ps) {
  EW_SETUP
  const u64 *a = tab.a[p], *b = tab.b[p];
  const size_t a_ps = (size_t)tab.a_ps[p] * cx.N, b_ps = (size_t)tab.b_ps[p] * cx.N;
  u64 *out = out_b + (size_t)p * 3 * o_ps;
  ulonglong2 a0 = ld2(a + off), a1 = ld2(a + a_ps + off);
  ulonglong2 b0 = ld2(b + off), b1 = ld2(b + b_ps + off);
  ulonglong2 d0, d1, d2;
  d0.x = mulmod(a0.x, b0.x, pm);
  d0.y = mulmod(a0.y, b0.y, pm);
  d2.x = mulmod(a1.x, b1.x, pm);
  d2.y = mulmod(a1.y, b1.y, pm);
  u128_t t = mul128(a0.x, b1.x);
  acc128(t, a1.x, b0.x);
  d1.x = barrett128(t, pm);
  t = mul128(a0.y, b1.y);
  acc128(t, a1.y, b0.y);
  d1.y = barrett128(t, pm);
  st2(out + off, d0);
  st2(out + o_ps + off, d1);
  st2(out + 2 * o_ps + off, d2);
}

// K6b: out = sum_j ct_j (*) pt_j  (pt_j == nullptr: ct_j itself) — a convolution / linear-layer
// row as ONE pass: every input word is read once, products accumulate unreduced in 128 bits
// (n <= 64 terms of < 2^122) and are reduced once.  Same canonical result as the
// multiply_plain / add sequence it stands for.
struct WsTab {
  const u64 *ct[KS_BATCH_MAX], *pt[KS_BATCH_MAX];
  uint32_t ct_ps[KS_BATCH_MAX]; // poly strides in units of N coefficients
};
__global__ void __launch_bounds__(256)
k_weighted_sum(DevCtx cx, WsTab tab, uint32_t n, u64 *out, size_t o_ps) {
  EW_SETUP
  u128_t a0 = {0, 0}, a1 = {0, 0};
  for (uint32_t j = 0; j < n; j++) {
    const ulonglong2 x = ld2(tab.ct[j] + (size_t)p * tab.ct_ps[j] * cx.N + off);
    if (tab.pt[j]) {
      const ulonglong2 w = ld2(tab.pt[j] + off);
      acc128(a0, x.x, w.x);
      acc128(a1, x.y, w.y);
    } else {
      acc128(a0, x.x, 1);
      acc128(a1, x.y, 1);
    }
  }
  ulonglong2 r;
  r.x = barrett128(a0, pm);
  r.y = barrett128(a1, pm);
  st2(out + p * o_ps + off, r);
}

// K5: square 2 -> 3: (a0^2, 2 a0 a1, a1^2)
__global__ void __launch_bounds__(256)
k_square(DevCtx cx, const u64 *a, size_t a_ps, u64 *out, size_t o_ps) {
  EW_SETUP
  a += (size_t)p * 2 * a_ps;
  out += (size_t)p * 3 * o_ps;
  ulonglong2 a0 = ld2(a + off), a1 = ld2(a + a_ps + off);
  ulonglong2 d0, d1, d2;
  d0.x = mulmod(a0.x, a0.x, pm);
  d0.y = mulmod(a0.y, a0.y, pm);
  d2.x = mulmod(a1.x, a1.x, pm);
  d2.y = mulmod(a1.y, a1.y, pm);
  u64 x = mulmod(a0.x, a1.x, pm), y = mulmod(a0.y, a1.y, pm);
  d1.x = addmod(x, x, pm.q);
  d1.y = addmod(y, y, pm.q);
  st2(out + off, d0);
  st2(out + o_ps + off, d1);
  st2(out + 2 * o_ps + off, d2);
}

// K6: multiply_plain, every poly x pt
__global__ void __launch_bounds__(256)
k_mul_plain(DevCtx cx, const u64 *a, size_t a_ps, const u64 *pt, u64 *out, size_t o_ps) {
  EW_SETUP
  ulonglong2 x = ld2(a + p * a_ps + off), y = ld2(pt + off), r;
  r.x = mulmod(x.x, y.x, pm);
  r.y = mulmod(x.y, y.y, pm);
  st2(out + p * o_ps + off, r);
}

// K6 batched: n independent multiply_plain of one shape in one launch; grid.z = instance * size + poly
struct MpTab {
  const u64 *ct[KS_BATCH_MAX], *pt[KS_BATCH_MAX];
  uint32_t ct_ps[KS_BATCH_MAX]; // poly strides in units of N coefficients
};
__global__ void __launch_bounds__(256)
k_mul_plain_many(DevCtx cx, MpTab tab, uint32_t size, u64 *out, size_t o_ps) {
  EW_SETUP
  const uint32_t inst = p / size, poly = p - inst * size;
  ulonglong2 x = ld2(tab.ct[inst] + (size_t)poly * tab.ct_ps[inst] * cx.N + off), y = ld2(tab.pt[inst] + off), r;
  r.x = mulmod(x.x, y.x, pm);
  r.y = mulmod(x.y, y.y, pm);
  st2(out + p * o_ps + off, r);
}

// K8: NTT-domain Galois automorphism out[p][i][n] = in[p][i][perm[n]]
__global__ void __launch_bounds__(256)
k_galois_perm(DevCtx cx, const u64 *a, size_t a_ps, const uint32_t *perm, u64 *out, size_t o_ps) {
  EW_SETUP
  const uint32_t n = 2 * (blockIdx.x * blockDim.x + threadIdx.x);
  const uint2 pi = *reinterpret_cast<const uint2 *>(perm + n);
  const u64 *src = a + p * a_ps + (size_t)i * cx.N;
  ulonglong2 r;
  r.x = src[pi.x];
  r.y = src[pi.y];
  st2(out + p * o_ps + off, r);
}

// ---- CKKS encoder on the device (SEAL 3.6 CKKSEncoder::encode_internal, reached from
// seal_executor.h:242): values -> conjugate-symmetric slot vector -> inverse special FFT in FP64
// (Gentleman-Sande, one launch per stage, roots in the order the stages consume them) with the
// factor scale/N folded into the LAST stage exactly as SEAL's DWTHandler::transform_from_rev
// does (sums scaled, differences times the pre-scaled root) -> round -> residues.  Each complex
// product is four rounded multiplies, a rounded difference and a rounded sum; FMA contraction is
// off, so the doubles — and the plaintext — are those of the host encoder and the CPU oracle.
__global__ void __launch_bounds__(256)
k_enc_scatter(const double *vals, uint32_t n_vals, const uint32_t *slot_map, double2 *c, uint32_t slots) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= slots) return;
  const double v = vals[i % n_vals]; // the vector is replicated over the N/2 slots (seal_executor.h:226-240)
  c[slot_map[i]] = make_double2(v, 0.0);
  c[slot_map[slots + i]] = make_double2(v, -0.0); // conjugate of a real value
}
// stage with gap 2^log_gap: group g uses roots[root0 + g] (already conjugated)
__global__ void __launch_bounds__(256)
k_enc_fft_stage(double2 *c, const double2 *roots, uint32_t root0, uint32_t log_gap, uint32_t half_n) {
#pragma clang fp contract(off)
  const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= half_n) return;
  const uint32_t gap = 1u << log_gap, g = idx >> log_gap, j = idx & (gap - 1);
  const uint32_t a = 2 * g * gap + j, b = a + gap;
  const double2 r = roots[root0 + g];
  const double2 u = c[a], v = c[b];
  c[a] = make_double2(u.x + v.x, u.y + v.y);
  const double dx = u.x - v.x, dy = u.y - v.y;
  const double p = dx * r.x, q = dy * r.y, s = dx * r.y, t = dy * r.x;
  c[b] = make_double2(p - q, s + t);
}
// last stage (one group, gap = N/2): x = (u + v) * fix, y = (u - v) * (root * fix)
__global__ void __launch_bounds__(256)
k_enc_fft_last(double2 *c, double2 scaled_root, double fix, uint32_t half_n) {
#pragma clang fp contract(off)
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= half_n) return;
  const double2 u = c[j], v = c[j + half_n];
  c[j] = make_double2((u.x + v.x) * fix, (u.y + v.y) * fix);
  const double dx = u.x - v.x, dy = u.y - v.y;
  const double p = dx * scaled_root.x, q = dy * scaled_root.y, s = dx * scaled_root.y, t = dy * scaled_root.x;
  c[j + half_n] = make_double2(p - q, s + t);
}
__global__ void __launch_bounds__(256)
k_enc_round(DevCtx cx, const double2 *c, uint32_t limbs, u64 *out) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= cx.N) return;
  const double t = c[j].x;
  const double x = fabs(t) < 4503599627370496.0 ? round(t) : t; // >= 2^52: already an integer
  const bool neg = signbit(x);
  const u64 mant = (u64)fabs(x); // |x| < 2^63 is guaranteed by the caller's bound
  for (uint32_t i = 0; i < limbs; i++) {
    const DevPrime pm = cx.primes[cx.prime_of(i)];
    const u64 r = barrett64(mant, pm.q, pm.brt);
    out[(size_t)i * cx.N + j] = (neg && r) ? pm.q - r : r;
  }
}

// K8 for (ciphertext, rotation) pairs: pair r reads its own source; grid.z = r * 2 + p
struct PermPairs {
  const uint32_t *perm[KS_BATCH_MAX];
  const u64 *src[KS_BATCH_MAX];
  uint32_t src_ps[KS_BATCH_MAX]; // poly strides in units of N coefficients
};
__global__ void __launch_bounds__(256)
k_galois_perm_pairs(DevCtx cx, PermPairs pt, u64 *out, size_t o_ps, uint32_t polys) {
  // polys == 2: z = 2 r + K, polynomial K of pair r; polys == 1: z = r and only c0 is permuted (the
  // hoisted form never needs the permuted c1).  Output slot 2 r + K either way.
  if (cx.skipped()) return;
  const uint32_t z = blockIdx.z, r = polys == 2 ? z >> 1 : z, p = polys == 2 ? z & 1 : 0, i = blockIdx.y;
  const uint32_t n = 2 * (blockIdx.x * blockDim.x + threadIdx.x);
  const uint2 pi = *reinterpret_cast<const uint2 *>(pt.perm[r] + n);
  const u64 *src = pt.src[r] + ((size_t)p * pt.src_ps[r] + i) * cx.N;
  ulonglong2 v;
  v.x = src[pi.x];
  v.y = src[pi.y];
  st2(out + (size_t)(2 * r + p) * o_ps + (size_t)i * cx.N + n, v);
}

// per-limb constant fill (uniform-constant plaintexts); the per-limb values travel as a
// kernel argument so the call needs no host->device copy and no synchronisation
struct LimbVals {
  u64 v[64];
};
__global__ void __launch_bounds__(256) k_fill_limbs(DevCtx cx, LimbVals vals, u64 *out) {
  EW_SETUP
  ulonglong2 r;
  r.x = r.y = vals.v[i];
  st2(out + off, r);
}

// K9 inner product: prod[K][I] = sum_J op(I,J) * key[J][K][kappa(I)],
// op(I,J) = target[J] when I == J, else scratch[I][J].  128-bit lazy accumulation, one Barrett
// reduction at the end.  grid = (N/512, l+1).
__global__ void __launch_bounds__(256)
k_ks_mac(DevCtx cx, const u64 *target, const u64 *scratch, const u64 *key, u64 *prod, uint32_t l) {
  if (cx.skipped()) return;
  const uint32_t I = blockIdx.y;
  const uint32_t kap = (I == l) ? cx.k - 1 : I;
  const size_t n = 2 * ((size_t)blockIdx.x * blockDim.x + threadIdx.x);
  const DevPrime pm = cx.primes[kap];
  const size_t N = cx.N, key_digit = (size_t)2 * cx.k * N;
  u128_t a0x = {0, 0}, a0y = {0, 0}, a1x = {0, 0}, a1y = {0, 0};
  for (uint32_t J = 0; J < l; J++) {
    const u64 *op = (I == J) ? target + (size_t)J * N : scratch + ((size_t)I * l + J) * N;
    const ulonglong2 o = ld2(op + n);
    const u64 *kp = key + J * key_digit + (size_t)kap * N + n;
    const ulonglong2 k0 = ld2(kp), k1 = ld2(kp + (size_t)cx.k * N);
    acc128(a0x, o.x, k0.x);
    acc128(a0y, o.y, k0.y);
    acc128(a1x, o.x, k1.x);
    acc128(a1y, o.y, k1.y);
  }
  ulonglong2 r0, r1;
  r0.x = barrett128(a0x, pm);
  r0.y = barrett128(a0y, pm);
  r1.x = barrett128(a1x, pm);
  r1.y = barrett128(a1y, pm);
  st2(prod + (size_t)I * N + n, r0);
  st2(prod + ((size_t)(l + 1) + I) * N + n, r1);
}

// ---- Hoisted rotations: n rotations of ONE ciphertext share the digit decomposition of c1.
// SEAL rotates first and decomposes sigma(c1) (Evaluator::rotate_internal -> apply_galois_ntt ->
// switch_key_inplace; seal_executor.h:181/:188): digit J of the rotated polynomial is
// sigma(t_J) with the sign flips taken modulo q_J, i.e. as an integer polynomial
//     t'_J = sigma_Z(t_J) + q_J * s,   s[k'] = 1 where sigma flips the sign at k' and t_J[k] != 0
// (sigma_Z = the automorphism with integer negation).  NTT_I is linear and commutes with sigma
// as the NTT-domain index permutation, so under every output prime q_I
//     NTT_I(t'_J) = perm(NTT_I(t_J)) + (q_J mod q_I) * NTT_I(s)
// and the key inner product of the rotated ciphertext is
//     sum_J perm(D[I][J]) * key[J][K][I]  +  NTT_I(s) * sum_J (q_J mod q_I) * key[J][K][I]
// with D = the transformed digits of the UNROTATED c1 (computed once) and a second term that is a
// constant of (Galois element, level): the same residues SEAL gets, 1/n of the transforms.
// The identity needs t_J[k] != 0 at the flipped positions: where t_J[k] = 0 the true digit is 0, not
// q_J, so the sum above is too large by (q_J mod q_I) * NTT_I(X^k') * key[J][K][I].  Zero coefficients
// are rare (N / q_J per limb: ~2^-44 at 60 bits, but 6 % of the ciphertexts for N = 2^16 and one of
// EVA's 20-bit primes), so the inverse transform records them and k_hoist_fix subtracts their terms
// one by one (NTT_I(X^k')[n] = psi_I^((2 brv(n) + 1) k')).  More than HOIST_ZERO_CAP zeros (a
// transparent ciphertext) make the guarded, unhoisted launch set recompute the outputs instead.
struct HoistTab { // per (source, rotation) pair of the launch
  const uint32_t *perm[KS_BATCH_MAX];
  const u64 *key[KS_BATCH_MAX];
  const u64 *corr[KS_BATCH_MAX]; // [2][l+1][N]
  const u64 *c1[KS_BATCH_MAX];   // the source's own c1 (NTT form): the digit used as is where I == J
  uint32_t elt[KS_BATCH_MAX];
  uint8_t src[KS_BATCH_MAX];     // index of the source among the set's transformed digits
};
// corr[K][I][n] = sign[kap][n] * sum_J (q_J mod q_kap) * key[J][K][kap][n]   (kap = prime of row I)
__global__ void __launch_bounds__(256)
k_hoist_corr(DevCtx cx, const u64 *sign, const u64 *key, u64 *corr, uint32_t l) {
  const uint32_t I = blockIdx.y, K = blockIdx.z, kap = (I == l) ? cx.k - 1 : I;
  const size_t n = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const DevPrime pm = cx.primes[kap];
  const size_t N = cx.N, key_digit = (size_t)2 * cx.k * N;
  u64 acc = 0;
  for (uint32_t J = 0; J < l; J++) {
    const u64 qj = cx.primes[J].q % pm.q; // 0 when J == I
    acc = addmod(acc, mulmod(qj, key[J * key_digit + ((size_t)K * cx.k + kap) * N + n], pm), pm.q);
  }
  corr[((size_t)K * (l + 1) + I) * N + n] = mulmod(sign[(size_t)kap * N + n], acc, pm);
}
// prod[z][K][I][n] = sum_J D_s[I][J][perm_z[n]] * key_z[J][K][I][n] + corr_z[K][I][n] for pair z with source s.
// D_s[I][J] is row (I * l + J) of source s's converted digits, or limb J of the source's own c1
// when I == J (SEAL's shortcut: the NTT-form limb is used as is).  grid = (N/512, l+1, pairs).
__global__ void __launch_bounds__(256)
k_hoist_mac(DevCtx cx, const u64 *digits, size_t dg_bs, HoistTab tab, u64 *prod, size_t prod_bs, uint32_t l) {
  const uint32_t I = blockIdx.y, kap = (I == l) ? cx.k - 1 : I;
  const uint32_t z = blockIdx.z;
  const size_t n = 2 * ((size_t)blockIdx.x * blockDim.x + threadIdx.x);
  const DevPrime pm = cx.primes[kap];
  const size_t N = cx.N, key_digit = (size_t)2 * cx.k * N;
  const uint2 pi = *reinterpret_cast<const uint2 *>(tab.perm[z] + n);
  const u64 *key = tab.key[z] + (size_t)kap * N + n;
  const u64 *dg = digits + tab.src[z] * dg_bs + (size_t)I * l * N, *own = tab.c1[z];
  u128_t a0x = {0, 0}, a0y = {0, 0}, a1x = {0, 0}, a1y = {0, 0};
  // operands are canonical (< q < 2^60): 256 products fit the 128-bit accumulators, l <= k - 1 < 64
  for (uint32_t J = 0; J < l; J++) {
    const u64 *op = (I == J) ? own + (size_t)J * N : dg + (size_t)J * N;
    const u64 ox = op[pi.x], oy = op[pi.y];
    const ulonglong2 k0 = ld2(key + J * key_digit), k1 = ld2(key + J * key_digit + (size_t)cx.k * N);
    acc128(a0x, ox, k0.x);
    acc128(a0y, oy, k0.y);
    acc128(a1x, ox, k1.x);
    acc128(a1y, oy, k1.y);
  }
  const u64 *cr = tab.corr[z] + (size_t)I * N + n;
  const ulonglong2 c0 = ld2(cr), c1c = ld2(cr + (size_t)(l + 1) * N);
  ulonglong2 r0, r1;
  r0.x = addmod(barrett128(a0x, pm), c0.x, pm.q);
  r0.y = addmod(barrett128(a0y, pm), c0.y, pm.q);
  r1.x = addmod(barrett128(a1x, pm), c1c.x, pm.q);
  r1.y = addmod(barrett128(a1y, pm), c1c.y, pm.q);
  u64 *pr = prod + z * prod_bs + (size_t)I * N + n;
  st2(pr, r0);
  st2(pr + (size_t)(l + 1) * N, r1);
}

// zeros[0] (low word) = number of zero digit coefficients seen, zeros[1 + e] = (source << 48 | J << 32 | k).
// Same grid as k_hoist_mac with one coefficient per thread; every test below is block-uniform.
__global__ void __launch_bounds__(256)
k_hoist_fix(DevCtx cx, const u64 *zeros, HoistTab tab, u64 *prod, size_t prod_bs, uint32_t l) {
  const uint32_t count = *reinterpret_cast<const uint32_t *>(zeros);
  if (count == 0 || count > HOIST_ZERO_CAP) return;
  const uint32_t I = blockIdx.y, kap = (I == l) ? cx.k - 1 : I;
  const uint32_t z = blockIdx.z, b = tab.src[z];
  const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
  const DevPrime pm = cx.primes[kap];
  const size_t N = cx.N, key_digit = (size_t)2 * cx.k * N;
  const uint32_t en = 2u * (__brev(n) >> (32 - cx.logN)) + 1u; // slot n holds the evaluation at psi^en
  const ulonglong2 *tw = cx.tw_fwd + (size_t)kap * N;
  const u64 *key = tab.key[z] + (size_t)kap * N + n;
  u64 acc0 = 0, acc1 = 0;
  bool any = false;
  for (uint32_t e = 0; e < count; e++) {
    const u64 ent = zeros[1 + e];
    const uint32_t J = (uint32_t)(ent >> 32) & 0xffffu, k = (uint32_t)ent;
    if ((uint32_t)(ent >> 48) != b || J >= l || J == kap) continue; // q_J mod q_J = 0
    const u64 raw = (u64)k * tab.elt[z];
    if (!((raw >> cx.logN) & 1)) continue; // the automorphism does not flip this coefficient
    const uint32_t kp = (uint32_t)raw & (uint32_t)(N - 1);
    const uint32_t m = (uint32_t)(((u64)en * kp) & (2 * N - 1));
    u64 w = tw[__brev(m & (uint32_t)(N - 1)) >> (32 - cx.logN)].x; // psi^(m mod N)
    if (m >= N) w = negmod(w, pm.q);
    const u64 t = mulmod(cx.primes[J].q % pm.q, w, pm);
    acc0 = addmod(acc0, mulmod(t, key[J * key_digit], pm), pm.q);
    acc1 = addmod(acc1, mulmod(t, key[J * key_digit + (size_t)cx.k * N], pm), pm.q);
    any = true;
  }
  if (!any) return;
  u64 *pr = prod + z * prod_bs + (size_t)I * N + n;
  pr[0] = submod(pr[0], acc0, pm.q);
  pr[(size_t)(l + 1) * N] = submod(pr[(size_t)(l + 1) * N], acc1, pm.q);
}

// ---- NTT launch plumbing
template <class Op> struct OpClass;
template <bool Z> struct OpClass<OpPlainT<Z>> { static constexpr int fwd_a = KC_NTT_A, fwd_b = KC_NTT_B; };
template <> struct OpClass<OpMulIntt> { static constexpr int fwd_a = KC_NTT_A, fwd_b = KC_NTT_B; };
template <> struct OpClass<OpKsDigit> { static constexpr int fwd_a = KC_KSDIGIT_A, fwd_b = KC_KSDIGIT_B; };
template <> struct OpClass<OpModDown> { static constexpr int fwd_a = KC_MODDOWN_A, fwd_b = KC_MODDOWN_B; };
template <bool M> struct OpClass<OpRRT<M>> { static constexpr int fwd_a = KC_MODDOWN_A, fwd_b = KC_MODDOWN_B; };
template <bool M> struct OpClass<OpRRLastT<M>> { static constexpr int fwd_a = KC_MODDOWN_A, fwd_b = KC_MODDOWN_B; };

template <int P, int LR, bool STRIDED, bool INVERSE, class Op>
static void launch_pass(evah_ctx *c, const typename Op::Params &prm, uint32_t jobs) {
  ProfScope ps(c, INVERSE ? (STRIDED ? KC_INTT_B : KC_INTT_A)
                          : (STRIDED ? OpClass<Op>::fwd_a : OpClass<Op>::fwd_b));
  const uint32_t max_tile = (uint32_t)NTT_THREADS << LR;
  const uint32_t tile = c->N < max_tile ? c->N : max_tile;
  const int logC = (int)ilog2(tile) - P;
  size_t lds = ((((size_t)1 << logC) * lds_sub_stride<P>() + 1) & ~(size_t)1) * sizeof(u64);
  if (STRIDED) lds += ((size_t)1 << P) * sizeof(ulonglong2); // staged twiddles
  const uint32_t n_tiles = c->N / tile;
  const int log_tiles = (int)ilog2(n_tiles);
  dim3 grid = Op::grid(prm, jobs), block(tile >> LR);
  grid.x *= n_tiles;
  if (tile == max_tile) {
    hipLaunchKernelGGL((ntt_pass_kernel<P, LR, STRIDED, INVERSE, Op, true>), grid, block, lds, c->stream, c->dev,
                       prm, logC, log_tiles);
  } else if constexpr (P == 5) { // N = 1024: one partial tile per polynomial
    hipLaunchKernelGGL((ntt_pass_kernel<P, LR, STRIDED, INVERSE, Op, false>), grid, block, lds, c->stream, c->dev,
                       prm, logC, log_tiles);
  } else {
    throw std::logic_error("partial NTT tile with P != 5");
  }
  HIPCHK(hipGetLastError());
}

template <int LR, bool STRIDED, bool INVERSE, class Op>
static void launch_pass_lr(evah_ctx *c, int P, const typename Op::Params &prm, uint32_t jobs) {
  switch (P) {
  case 5: launch_pass<5, LR, STRIDED, INVERSE, Op>(c, prm, jobs); break;
  case 6: launch_pass<6, LR, STRIDED, INVERSE, Op>(c, prm, jobs); break;
  case 7: launch_pass<7, LR, STRIDED, INVERSE, Op>(c, prm, jobs); break;
  case 8: launch_pass<8, LR, STRIDED, INVERSE, Op>(c, prm, jobs); break;
  case 9:
    if constexpr (STRIDED) { launch_pass<9, LR, STRIDED, INVERSE, Op>(c, prm, jobs); break; }
    [[fallthrough]];
  default: throw std::runtime_error("unsupported poly_modulus_degree for the NTT kernels");
  }
}
// 8 coefficients per thread measured best for the stand-alone passes on MI355X (vs 4: +12 %,
// vs 16: +10 %, profiles/r01_tuning_notes.md); the fused key-switch kernel uses 4.
template <bool STRIDED, bool INVERSE, class Op>
static void launch_pass_p(evah_ctx *c, int P, const typename Op::Params &prm, uint32_t jobs) {
  // a launch that cannot fill the chip is bound by ONE wave's instruction stream (a thread's 8
  // coefficients are ~600 integer instructions per pass): with 4 coefficients per thread the same
  // tile work is spread over twice the workgroups and the critical path of a workgroup shrinks
  if (c->small_lr == 2 && (uint64_t)jobs * (c->N >> 11) <= c->small_lr_blocks && c->N >= 2048) {
    launch_pass_lr<2, STRIDED, INVERSE, Op>(c, P, prm, jobs);
    return;
  }
  launch_pass_lr<3, STRIDED, INVERSE, Op>(c, P, prm, jobs);
}

struct KsBatch { // one launch worth of key-switches: regular strides, irregular keys
  uint32_t n = 1;
  uint32_t i0 = 0, ni = 0; // output-limb slice
  size_t target_bs = 0, scratch_bs = 0, prod_bs = 0;
  KsKeys keys{};
  PtrTab targets{}; // used when the targets are separate allocations (target == nullptr)
  const MulTab *mul = nullptr; // fused multiply: the target of instance b is d2 = a1 b1 of product b
  uint32_t istep = 1, nout = 0; // output limbs I = i0 + y * istep; nout = rows per polynomial of prod (0: l + 1)
  u64 *r_out = nullptr; // != nullptr: the special row leaves as the first inverse pass of the mod-down (INVSP)
};
template <int P, int LR>
static void launch_ks_inner_plr(evah_ctx *c, const u64 *target, const u64 *scratch, const KsBatch &kb, u64 *prod, uint32_t l) {
  ProfScope ps(c, KC_KSMAC);
  const uint32_t max_tile = (uint32_t)c->ks_threads << LR;
  const uint32_t tile = c->N < max_tile ? c->N : max_tile;
  const int logC = (int)ilog2(tile) - P;
  if (logC < 0) throw std::runtime_error("ks_inner tile smaller than one sub-transform");
  // coefficients tile + per-sub twiddle heaps (16 B per node)
  const size_t lds = ((((size_t)1 << logC) * lds_sub_stride<P>() + 1) & ~(size_t)1) * sizeof(u64) +
                     ((size_t)1 << (logC + P)) * sizeof(ulonglong2);
  const uint32_t n_tiles = c->N / tile;
  // the one-wave workgroup (the default) is compiled with its own launch bound: the register
  // allocator is not held to the 256-thread budget
  auto go = [&](auto kernel, const auto &mt) {
    hipLaunchKernelGGL(kernel, dim3(n_tiles * kb.n, kb.ni), dim3(tile >> LR), lds, c->stream, c->dev, target, kb.target_bs, scratch,
                       kb.scratch_bs, kb.keys, prod, kb.prod_bs, l, kb.i0, logC, n_tiles, kb.n, kb.targets, mt, kb.istep,
                       kb.nout ? kb.nout : l + 1, kb.r_out);
  };
  if (kb.r_out && ((tile >> LR) > 64 || kb.istep != 1)) throw std::logic_error("fused special-row inverse pass needs the one-wave key-switch kernel");
  if (kb.r_out) {
    if (kb.mul) go(ks_inner_kernel<P, LR, 64, true, true>, *kb.mul);
    else go(ks_inner_kernel<P, LR, 64, false, true>, NoMul{});
  } else if ((tile >> LR) <= 64) {
    if (kb.mul) go(ks_inner_kernel<P, LR, 64, true>, *kb.mul);
    else go(ks_inner_kernel<P, LR, 64, false>, NoMul{});
  } else {
    if (kb.mul) go(ks_inner_kernel<P, LR, NTT_THREADS, true>, *kb.mul);
    else go(ks_inner_kernel<P, LR, NTT_THREADS, false>, NoMul{});
  }
  HIPCHK(hipGetLastError());
}
template <int LR>
static void launch_ks_inner_lr(evah_ctx *c, int P, const u64 *target, const u64 *scratch, const KsBatch &key, u64 *prod, uint32_t l) {
  switch (P) {
  case 5: launch_ks_inner_plr<5, LR>(c, target, scratch, key, prod, l); break;
  case 6: launch_ks_inner_plr<6, LR>(c, target, scratch, key, prod, l); break;
  case 7: launch_ks_inner_plr<7, LR>(c, target, scratch, key, prod, l); break;
  case 8: launch_ks_inner_plr<8, LR>(c, target, scratch, key, prod, l); break;
  default: throw std::runtime_error("unsupported poly_modulus_degree for the key-switch kernel");
  }
}
static void launch_ks_inner(evah_ctx *c, int P, const u64 *target, const u64 *scratch, const KsBatch &key, u64 *prod, uint32_t l) {
  launch_ks_inner_lr<2>(c, P, target, scratch, key, prod, l); // 4 coefficients per thread: 32 accumulator VGPRs
}

template <class Op> static void ntt_forward(evah_ctx *c, const typename Op::Params &prm, uint32_t jobs) {
  const int a = (c->logN + 1) / 2, b = c->logN / 2;
  launch_pass_p<true, false, Op>(c, a, prm, jobs);
  launch_pass_p<false, false, Op>(c, b, prm, jobs);
}
template <class Op> static void ntt_inverse(evah_ctx *c, const typename Op::Params &prm, uint32_t jobs) {
  const int a = (c->logN + 1) / 2, b = c->logN / 2;
  launch_pass_p<false, true, Op>(c, b, prm, jobs);
  launch_pass_p<true, true, Op>(c, a, prm, jobs);
}

// ---- latency-bound launches: inverse strided pass + forward strided pass as one launch (ntt_inv_fwd_kernel)
static bool fuse_small_launch(evah_ctx *c, uint32_t fwd_jobs) {
  const uint32_t tile = (uint32_t)NTT_THREADS << 3;
  if (!c->fuse_small_blocks || c->N < tile) return false; // partial tiles (N = 1024) keep the two-launch form
  if (c->dev.guard) return true; // a guarded (normally skipped) launch set: the fewest launches, whatever the size
  return (uint64_t)fwd_jobs * (c->N / tile) <= c->fuse_small_blocks;
}
template <int P, class Op, int LR> static void launch_inv_fwd_plr(evah_ctx *c, const typename Op::Params &prm, uint32_t jobs) {
  ProfScope ps(c, OpClass<Op>::fwd_a);
  const uint32_t tile = (uint32_t)NTT_THREADS << LR, n_tiles = c->N / tile;
  const int logC = (int)ilog2(tile) - P;
  const size_t lds = ((((size_t)1 << logC) * lds_sub_stride<P>() + 1) & ~(size_t)1) * sizeof(u64) + 2 * ((size_t)1 << P) * sizeof(ulonglong2);
  dim3 grid = Op::grid(prm, jobs);
  grid.x *= n_tiles;
  hipLaunchKernelGGL((ntt_inv_fwd_kernel<P, LR, Op>), grid, dim3(NTT_THREADS), lds, c->stream, c->dev, prm, (int)ilog2(n_tiles));
  HIPCHK(hipGetLastError());
}
template <int P, class Op> static void launch_inv_fwd_p(evah_ctx *c, const typename Op::Params &prm, uint32_t jobs) {
  if (c->small_lr == 2) launch_inv_fwd_plr<P, Op, 2>(c, prm, jobs);
  else launch_inv_fwd_plr<P, Op, 3>(c, prm, jobs);
}
template <class Op> static void launch_inv_fwd(evah_ctx *c, const typename Op::Params &prm, uint32_t jobs) {
  switch ((c->logN + 1) / 2) {
  case 6: launch_inv_fwd_p<6, Op>(c, prm, jobs); break;
  case 7: launch_inv_fwd_p<7, Op>(c, prm, jobs); break;
  case 8: launch_inv_fwd_p<8, Op>(c, prm, jobs); break;
  case 9: launch_inv_fwd_p<9, Op>(c, prm, jobs); break;
  default: throw std::runtime_error("unsupported poly_modulus_degree for the fused inverse/forward pass");
  }
}
// inverse transform of the source limb(s) (InvOp jobs) followed by the forward transforms of Op:
// four launches, or three when the forward launch is too small to fill the chip
// inv_pass1_done: the contiguous inverse pass already left its intermediate in ip.dst (fused into the
// key-switch kernel); only ever set when fuse_small_launch(c, fwd_jobs) holds
template <class InvOp, class Op>
static void inverse_then_forward(evah_ctx *c, const typename InvOp::Params &ip, uint32_t inv_jobs, const typename Op::Params &fp,
                                 uint32_t fwd_jobs, bool inv_pass1_done = false) {
  if (inv_pass1_done && !fuse_small_launch(c, fwd_jobs)) throw std::logic_error("fused inverse pass outside the small-launch form");
  if (fuse_small_launch(c, fwd_jobs)) {
    if (!inv_pass1_done) launch_pass_p<false, true, InvOp>(c, c->logN / 2, ip, inv_jobs); // contiguous inverse pass: lazy intermediate in ip.dst
    launch_inv_fwd<Op>(c, fp, fwd_jobs);
    launch_pass_p<false, false, Op>(c, c->logN / 2, fp, fwd_jobs);
  } else {
    ntt_inverse<InvOp>(c, ip, inv_jobs);
    ntt_forward<Op>(c, fp, fwd_jobs);
  }
}

// SEAL Evaluator::switch_key_inplace, device version.
//   out[K] = (add && K < add_polys ? add[K] : 0) + keyswitch(target)[K],  K in {0,1}
// steps 1-2 of switch_key for a batch of n (target, key) pairs in one set of launches:
// prod[b][K][I] (I <= l, slot l = special prime) = sum_J op_b(I,J) * key_b[J][K].
// target_b = target + b * target_bs; prod_b = prod_d + b * 2 (l+1) N.
// r_small != nullptr: the caller will mod-down through the latency-bound launch form and offers
// r_small[2 n][N] for the special rows' first inverse pass; returns true when that pass was done
// here (fused into the key-switch kernel) — the special rows of prod are then NOT written.
static bool switch_key_products(evah_ctx *c, uint32_t l, const u64 *target, size_t target_bs, const KeyDev *const *keys,
                                uint32_t n, u64 *prod_d, const PtrTab *target_tab = nullptr, const MulTab *mul = nullptr,
                                u64 *r_small = nullptr) {
  const size_t N = c->N;
  if (n < 1 || n > (uint32_t)KS_BATCH_MAX) throw std::runtime_error("key-switch batch out of range");
  KsBatch kb;
  kb.n = n;
  kb.target_bs = target_bs;
  kb.scratch_bs = (size_t)(l + 1) * l * N;
  kb.prod_bs = (size_t)2 * (l + 1) * N;
  for (uint32_t b = 0; b < n; b++) {
    if (keys[b]->n_digits < l) throw std::runtime_error("key switching key has too few digits");
    kb.keys.key[b] = keys[b]->d;
  }
  if (target_tab) kb.targets = *target_tab; // target == nullptr: separately allocated targets
  kb.mul = mul;
  if (mul && !c->fuse_mac) throw std::logic_error("the fused multiply needs the fused key-switch kernel");
  Scratch t(c, (size_t)n * l * N);        // coefficient-form digits
  Scratch sc(c, n * kb.scratch_bs);       // converted digits, NTT form per output limb
  // 1. digits to coefficient form (job -> (b, J))
  OpKsDigit::Params dp{t.d, sc.d, l, (size_t)l * N, kb.scratch_bs, 0, l + 1};
  // a small key switch is latency-bound: the digits' strided inverse pass and the first pass of the
  // digit conversion then run as one launch
  const bool small = c->fuse_mac && std::max(1, c->ks_groups) == 1 && fuse_small_launch(c, n * (l + 1) * l);
  if (mul) { // the target is the product's d2, formed on load
    OpMulIntt::Params ip{*mul, t.d, (size_t)l * N, l};
    if (small) launch_pass_p<false, true, OpMulIntt>(c, c->logN / 2, ip, n * l);
    else ntt_inverse<OpMulIntt>(c, ip, n * l);
  } else {
    OpPlain::Params ip{target, t.d, target_bs, (size_t)l * N, l, 0, 0, {}};
    if (target_tab) ip.src_tab = *target_tab;
    if (small) launch_pass_p<false, true, OpPlain>(c, c->logN / 2, ip, n * l);
    else ntt_inverse<OpPlain>(c, ip, n * l);
  }
  if (small) {
    dp.i0 = kb.i0 = 0;
    dp.ni = kb.ni = l + 1;
    launch_inv_fwd<OpKsDigit>(c, dp, n * (l + 1) * l);
    const uint32_t max_tile = (uint32_t)c->ks_threads << 2;
    if (r_small && c->fuse_special_inv && (std::min<uint32_t>(c->N, max_tile) >> 2) <= 64) kb.r_out = r_small;
    launch_ks_inner(c, c->logN / 2, target, sc.d, kb, prod_d, l);
    return kb.r_out != nullptr;
  }
  if (c->fuse_mac) { // 128-bit accumulation of lazy (<16q) products, folded every 16 digits
    // Output limbs are processed in slices so that a slice's converted digits (ni * l * N words)
    // are still in L2 / Infinity Cache when the fused second pass consumes them.
    const int a = (c->logN + 1) / 2, b = c->logN / 2;
    const uint32_t groups = std::min<uint32_t>(std::max(1, c->ks_groups), l + 1);
    for (uint32_t g = 0; g < groups; g++) {
      const uint32_t i0 = (uint32_t)((uint64_t)(l + 1) * g / groups), i1 = (uint32_t)((uint64_t)(l + 1) * (g + 1) / groups);
      if (i1 == i0) continue;
      dp.i0 = kb.i0 = i0;
      dp.ni = kb.ni = i1 - i0;
      // 2a. base-convert + first (strided) NTT pass of every digit under the slice's output primes
      launch_pass_p<true, false, OpKsDigit>(c, a, dp, n * (i1 - i0) * l);
      // 2b. second (contiguous) pass fused with the inner product with the key
      launch_ks_inner(c, b, target, sc.d, kb, prod_d, l);
    }
  } else {
    // unfused reference path (EVAH_FUSE_MAC=0): full digit NTTs, then a separate MAC kernel
    ntt_forward<OpKsDigit>(c, dp, n * (l + 1) * l);
    for (uint32_t b = 0; b < n; b++) {
      ProfScope ps(c, KC_KSMAC);
      hipLaunchKernelGGL(k_ks_mac, dim3(c->N / 512, l + 1), dim3(256), 0, c->stream, c->dev,
                         target ? target + b * target_bs : target_tab->p[b], sc.d + b * kb.scratch_bs, keys[b]->d,
                         prod_d + b * kb.prod_bs, l);
      HIPCHK(hipGetLastError());
    }
  }
  return false;
}

static void switch_key(evah_ctx *c, uint32_t l, const u64 *target, const KeyDev &key, const u64 *add,
                       size_t add_ps, uint32_t add_polys, u64 *out, size_t out_ps) {
  const size_t N = c->N;
  Scratch prod(c, (size_t)2 * (l + 1) * N);    // [K][l+1][N]
  Scratch r(c, 2 * N);
  const KeyDev *kp = &key;
  const bool inv1 = switch_key_products(c, l, target, 0, &kp, 1, prod.d, nullptr, nullptr, fuse_small_launch(c, 2 * l) ? r.d : nullptr);
  // 3. mod-down by the special prime: INTT(special limb) + P/2, then per-limb NTT + combine
  OpPlain::Params sp{prod.d + (size_t)l * N, r.d, (size_t)(l + 1) * N, N, 1, c->k - 1, 1, {}};
    OpModDown::Params mp{r.d, N, prod.d, (size_t)(l + 1) * N, add, add_ps, add_polys, out, out_ps,
                       c->k - 1, l};
  inverse_then_forward<OpPlain, OpModDown>(c, sp, 2, mp, 2 * l, inv1);
}

bool hoist_wanted(const evah_ctx *c, uint32_t l, uint32_t n, uint32_t B) {
  // hoisting pays when the digit transforms it saves are throughput, not latency (the exact fallback
  // costs a set of empty launches); limb-sharded contexts go through the shard phases instead
  return c->hoist && n >= 2 && c->dev.pstep == 1 && c->N >= 2048 &&
         (uint64_t)n * B * l * (l + 1) * (c->N >> 11) >= c->hoist_min_tiles;
}

} // namespace evah

extern "C" {

int evah_pt_upload_coeff(evah_ctx *c, uint32_t limbs, double scale, const uint64_t *data, evah_pt **out) {
  API_BEGIN
  use(c);
  if (c->capturing) throw std::logic_error("this call synchronises with the host and cannot be captured into a graph");
  if (limbs < 1 || limbs > c->k - 1) throw std::invalid_argument("invalid limb count for this context");
  evah_pt *t = pt_new(c, limbs, scale);
  HIPCHK(hipMemcpyAsync(t->d, data, sizeof(u64) * (size_t)limbs * c->N, hipMemcpyHostToDevice, c->stream));
  OpPlain::Params p{t->d, t->d, 0, 0, limbs, 0, 0, {}};
  ntt_forward<OpPlain>(c, p, limbs);
  HIPCHK(hipStreamSynchronize(c->stream)); // the pageable host buffer may go away after return
  count_h2d(c, sizeof(u64) * (size_t)limbs * c->N, true);
  t->buf->ready_everywhere = true;
  *out = t;
  API_END
}

// encoder tables of a context family (built on first use, never inside a capture)
static void enc_tables(evah_ctx *c) {
  if (c->sh->enc_roots) return;
  if (c->capturing) throw std::logic_error("first use of the device encoder cannot be captured into a graph");
  const uint32_t N = c->N, slots = N >> 1, m = 2 * N;
  std::vector<uint32_t> map(N);
  u64 pos = 1;
  for (uint32_t i = 0; i < slots; i++) {
    map[i] = bitrev((uint32_t)((pos - 1) >> 1), c->logN);
    map[slots + i] = bitrev((uint32_t)((m - pos - 1) >> 1), c->logN);
    pos = (pos * 3) & (m - 1);
  }
  // inverse-transform roots in consumption order, the doubles SEAL's ComplexRoots holds (hostmath.h)
  const CkksRoots cr = ckks_roots(N);
  std::vector<double> roots(2 * (size_t)N);
  for (uint32_t j = 0; j < N; j++) {
    roots[2 * j] = cr.inv_seq[j].real();
    roots[2 * j + 1] = cr.inv_seq[j].imag();
  }
  c->sh->enc_last_root[0] = cr.inv_seq[N - 1].real();
  c->sh->enc_last_root[1] = cr.inv_seq[N - 1].imag();
  HIPCHK(hipMalloc(&c->sh->enc_slot_map, sizeof(uint32_t) * N));
  HIPCHK(hipMalloc(&c->sh->enc_roots, sizeof(double2) * N));
  HIPCHK(hipMemcpy(c->sh->enc_slot_map, map.data(), sizeof(uint32_t) * N, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(c->sh->enc_roots, roots.data(), sizeof(double2) * N, hipMemcpyHostToDevice));
}

// CKKSEncoder::encode of `n_values` reals replicated over the N/2 slots, at 2^scale_bits... (scale is
// passed as the double SEAL takes), to `limbs` primes, NTT form.  The caller guarantees that every
// coefficient round(x * scale / N) is below 2^62 in magnitude (the host checks a bound on
// sum |values|); larger encodings take the host's multi-precision path + evah_pt_upload_coeff.
int evah_pt_encode(evah_ctx *c, const double *values, uint32_t n_values, uint32_t limbs, double scale, evah_pt **out) {
  API_BEGIN
  use(c);
  if (c->capturing) throw std::logic_error("this call synchronises with the host and cannot be captured into a graph");
  if (limbs < 1 || limbs > c->k - 1) throw std::invalid_argument("invalid limb count for this context");
  const uint32_t N = c->N, slots = N >> 1;
  if (n_values < 1 || n_values > slots || slots % n_values) throw std::invalid_argument("value count must divide the slot count");
  enc_tables(c);
  evah_pt *t = pt_new(c, limbs, scale);
  try {
    Scratch vals(c, n_values), cbuf(c, 2 * (size_t)N); // doubles / double2 in u64-sized words
    HIPCHK(hipMemcpyAsync(vals.d, values, sizeof(double) * n_values, hipMemcpyHostToDevice, c->stream));
    double2 *cd = reinterpret_cast<double2 *>(cbuf.d);
    ProfScope ps(c, KC_EW);
    hipLaunchKernelGGL(k_enc_scatter, dim3((slots + 255) / 256), dim3(256), 0, c->stream, reinterpret_cast<const double *>(vals.d),
                       n_values, c->sh->enc_slot_map, cd, slots);
    const double fix = scale / (double)N;
    for (uint32_t mm = N >> 1, lg = 0; mm > 1; mm >>= 1, lg++) // stage with mm groups starts at root N - 2mm + 1
      hipLaunchKernelGGL(k_enc_fft_stage, dim3((slots + 255) / 256), dim3(256), 0, c->stream, cd, c->sh->enc_roots, N - 2 * mm + 1,
                         lg, slots);
    hipLaunchKernelGGL(k_enc_fft_last, dim3((slots + 255) / 256), dim3(256), 0, c->stream, cd,
                       make_double2(c->sh->enc_last_root[0] * fix, c->sh->enc_last_root[1] * fix), fix, slots);
    hipLaunchKernelGGL(k_enc_round, dim3((N + 255) / 256), dim3(256), 0, c->stream, c->dev, cd, limbs, t->d);
    HIPCHK(hipGetLastError());
    OpPlain::Params p{t->d, t->d, 0, 0, limbs, 0, 0, {}};
    ntt_forward<OpPlain>(c, p, limbs);
    HIPCHK(hipStreamSynchronize(c->stream)); // `values` is pageable host memory
  } catch (...) {
    evah_pt_free(c, t);
    throw;
  }
  t->buf->ready_everywhere = true;
  *out = t;
  API_END
}

int evah_pt_uniform(evah_ctx *c, uint32_t limbs, double scale, const uint64_t *value, evah_pt **out) {
  API_BEGIN
  use(c);
  if (limbs < 1 || limbs > c->k - 1) throw std::invalid_argument("invalid limb count for this context");
  if (limbs > 64) throw std::invalid_argument("too many limbs");
  evah_pt *t = pt_new(c, limbs, scale);
  LimbVals lv;
  for (uint32_t i = 0; i < limbs; i++) lv.v[i] = value[i];
  EW_LAUNCH(k_fill_limbs, ew_grid(c, limbs, 1), dim3(256), 0, c->stream, c->dev, lv, t->d);
  HIPCHK(hipGetLastError());
  *out = t;
  API_END
}

// ---- evaluator

static int addsub_impl(evah_ctx *c, const evah_ct *a, const evah_ct *b, evah_ct **out, int sub) {
  API_BEGIN
  use(c);
  acquire(c, a->buf);
  acquire(c, b->buf);
  if (a->limbs != b->limbs) throw std::invalid_argument("encrypted1 and encrypted2 parameter mismatch");
  if (!same_scale(a->scale, b->scale)) throw std::invalid_argument("scale mismatch");
  if (a->batch != b->batch) throw std::invalid_argument("batch size mismatch");
  const uint32_t s = std::max(a->size, b->size);
  evah_ct *o = ct_new(c, s, a->limbs, a->scale, a->batch);
  EW_LAUNCH(k_addsub, ew_grid(c, a->limbs, s * a->batch), dim3(256), 0, c->stream, c->dev, a->d, a->ps, a->size,
                     b->d, b->ps, b->size, o->d, o->ps, sub, s);
  HIPCHK(hipGetLastError());
  *out = o;
  API_END
}
int evah_add(evah_ctx *c, const evah_ct *a, const evah_ct *b, evah_ct **out) { return addsub_impl(c, a, b, out, 0); }
int evah_sub(evah_ctx *c, const evah_ct *a, const evah_ct *b, evah_ct **out) { return addsub_impl(c, a, b, out, 1); }

static int addsub_plain_impl(evah_ctx *c, const evah_ct *a, const evah_pt *b, evah_ct **out, int sub) {
  API_BEGIN
  use(c);
  acquire(c, a->buf);
  acquire(c, b->buf);
  if (a->limbs != b->limbs) throw std::invalid_argument("encrypted and plain parameter mismatch");
  if (!same_scale(a->scale, b->scale)) throw std::invalid_argument("scale mismatch");
  evah_ct *o = ct_new(c, a->size, a->limbs, a->scale, a->batch);
  EW_LAUNCH(k_addsub, ew_grid(c, a->limbs, a->size * a->batch), dim3(256), 0, c->stream, c->dev, a->d, a->ps,
                     a->size, b->d, (size_t)0, 1u, o->d, o->ps, sub, a->size);
  HIPCHK(hipGetLastError());
  *out = o;
  API_END
}
int evah_add_plain(evah_ctx *c, const evah_ct *a, const evah_pt *b, evah_ct **out) { return addsub_plain_impl(c, a, b, out, 0); }
int evah_sub_plain(evah_ctx *c, const evah_ct *a, const evah_pt *b, evah_ct **out) { return addsub_plain_impl(c, a, b, out, 1); }

int evah_negate(evah_ctx *c, const evah_ct *a, evah_ct **out) {
  API_BEGIN
  use(c);
  acquire(c, a->buf);
  evah_ct *o = ct_new(c, a->size, a->limbs, a->scale, a->batch);
  EW_LAUNCH(k_negate, ew_grid(c, a->limbs, a->size * a->batch), dim3(256), 0, c->stream, c->dev, a->d, a->ps, o->d, o->ps);
  HIPCHK(hipGetLastError());
  *out = o;
  API_END
}

int evah_multiply(evah_ctx *c, const evah_ct *a, const evah_ct *b, evah_ct **out) {
  API_BEGIN
  use(c);
  acquire(c, a->buf);
  acquire(c, b->buf);
  if (a->limbs != b->limbs) throw std::invalid_argument("encrypted1 and encrypted2 parameter mismatch");
  if (a->size != 2 || b->size != 2) throw std::invalid_argument("multiply supports size-2 operands only (relinearize first)");
  const double ns = a->scale * b->scale;
  check_scale(c, ns, a->limbs);
  if (a->batch != b->batch) throw std::invalid_argument("batch size mismatch");
  evah_ct *o = ct_new(c, 3, a->limbs, ns, a->batch);
  EW_LAUNCH(k_mul22, ew_grid(c, a->limbs, a->batch), dim3(256), 0, c->stream, c->dev, a->d, a->ps, b->d, b->ps, o->d, o->ps);
  HIPCHK(hipGetLastError());
  *out = o;
  API_END
}

// n (<= 64) independent products at one level as ONE launch (same ciphertexts as n evah_multiply
// calls); the outputs are views into one allocation.
int evah_multiply_many(evah_ctx *c, const evah_ct *const *as, const evah_ct *const *bs, uint32_t n, evah_ct **outs) {
  API_BEGIN
  use(c);
  if (n < 1 || n > (uint32_t)KS_BATCH_MAX) throw std::invalid_argument("multiply_many handles 1..64 products per call");
  const uint32_t l = as[0]->limbs;
  const size_t N = c->N, ops = (size_t)l * N;
  MulTab tab{};
  std::vector<double> scales(n);
  for (uint32_t i = 0; i < n; i++) {
    const evah_ct *a = as[i], *b = bs[i];
    if (a->size != 2 || b->size != 2) throw std::invalid_argument("multiply supports size-2 operands only (relinearize first)");
    if (a->batch != 1 || b->batch != 1) throw std::invalid_argument("multiply_many takes single ciphertexts (a batched handle already multiplies in one launch)");
    if (a->limbs != l || b->limbs != l) throw std::invalid_argument("encrypted parameter mismatch in batch");
    scales[i] = a->scale * b->scale;
    check_scale(c, scales[i], l);
    acquire(c, a->buf);
    acquire(c, b->buf);
    tab.a[i] = a->d;
    tab.b[i] = b->d;
    tab.a_ps[i] = (uint32_t)(a->ps / N);
    tab.b_ps[i] = (uint32_t)(b->ps / N);
  }
  Buffer *ob = buf_new(c, (size_t)n * 3 * ops);
  EW_LAUNCH(k_mul22_many, ew_grid(c, l, n), dim3(256), 0, c->stream, c->dev, tab, ob->d, ops);
  HIPCHK(hipGetLastError());
  ob->refs = (int)n;
  for (uint32_t i = 0; i < n; i++) {
    evah_ct *t = new evah_ct;
    t->buf = ob;
    t->d = ob->d + (size_t)i * 3 * ops;
    t->size = 3;
    t->limbs = l;
    t->ps = ops;
    t->scale = scales[i];
    outs[i] = t;
  }
  API_END
}

int evah_square(evah_ctx *c, const evah_ct *a, evah_ct **out) {
  API_BEGIN
  use(c);
  acquire(c, a->buf);
  if (a->size != 2) throw std::invalid_argument("square supports size-2 operands only (relinearize first)");
  const double ns = a->scale * a->scale;
  check_scale(c, ns, a->limbs);
  evah_ct *o = ct_new(c, 3, a->limbs, ns, a->batch);
  EW_LAUNCH(k_square, ew_grid(c, a->limbs, a->batch), dim3(256), 0, c->stream, c->dev, a->d, a->ps, o->d, o->ps);
  HIPCHK(hipGetLastError());
  *out = o;
  API_END
}

int evah_multiply_plain(evah_ctx *c, const evah_ct *a, const evah_pt *b, evah_ct **out) {
  API_BEGIN
  use(c);
  acquire(c, a->buf);
  acquire(c, b->buf);
  if (a->limbs != b->limbs) throw std::invalid_argument("encrypted and plain parameter mismatch");
  const double ns = a->scale * b->scale;
  check_scale(c, ns, a->limbs);
  evah_ct *o = ct_new(c, a->size, a->limbs, ns, a->batch);
  EW_LAUNCH(k_mul_plain, ew_grid(c, a->limbs, a->size * a->batch), dim3(256), 0, c->stream, c->dev, a->d, a->ps, b->d, o->d, o->ps);
  HIPCHK(hipGetLastError());
  *out = o;
  API_END
}

// n (<= 64) independent multiply_plain calls of one shape (size, limbs) as one launch
int evah_multiply_plain_many(evah_ctx *c, const evah_ct *const *cts, const evah_pt *const *pts, uint32_t n, evah_ct **outs) {
  API_BEGIN
  use(c);
  if (n < 1 || n > (uint32_t)KS_BATCH_MAX) throw std::invalid_argument("multiply_plain_many handles 1..64 products per call");
  const uint32_t size = cts[0]->size, l = cts[0]->limbs;
  const size_t N = c->N, ops = (size_t)l * N;
  MpTab tab{};
  std::vector<double> scales(n);
  for (uint32_t i = 0; i < n; i++) {
    const evah_ct *a = cts[i];
    const evah_pt *b = pts[i];
    if (a->batch != 1) throw std::invalid_argument("multiply_plain_many takes single ciphertexts");
    if (a->size != size || a->limbs != l) throw std::invalid_argument("encrypted parameter mismatch in batch");
    if (b->limbs != l) throw std::invalid_argument("encrypted and plain parameter mismatch");
    scales[i] = a->scale * b->scale;
    check_scale(c, scales[i], l);
    acquire(c, a->buf);
    acquire(c, b->buf);
    tab.ct[i] = a->d;
    tab.pt[i] = b->d;
    tab.ct_ps[i] = (uint32_t)(a->ps / N);
  }
  Buffer *ob = buf_new(c, (size_t)n * size * ops);
  EW_LAUNCH(k_mul_plain_many, ew_grid(c, l, n * size), dim3(256), 0, c->stream, c->dev, tab, size, ob->d, ops);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    buf_unref(c, ob);
    HIPCHK(e);
  }
  ob->refs = (int)n;
  for (uint32_t i = 0; i < n; i++) {
    evah_ct *t = new evah_ct;
    t->buf = ob;
    t->d = ob->d + (size_t)i * size * ops;
    t->size = size;
    t->limbs = l;
    t->ps = ops;
    t->scale = scales[i];
    outs[i] = t;
  }
  API_END
}

// sum_j cts[j] (*) pts[j], pts[j] == NULL standing for the ciphertext itself: the value of
// add(... add(multiply_plain(cts[0], pts[0]), multiply_plain(cts[1], pts[1])) ...) in one pass
int evah_weighted_sum(evah_ctx *c, const evah_ct *const *cts, const evah_pt *const *pts, uint32_t n, evah_ct **out) {
  API_BEGIN
  use(c);
  if (n < 1 || n > (uint32_t)KS_BATCH_MAX) throw std::invalid_argument("weighted_sum handles 1..64 terms per call");
  const evah_ct *f = cts[0];
  const double scale = f->scale * (pts[0] ? pts[0]->scale : 1.0);
  WsTab tab{};
  for (uint32_t j = 0; j < n; j++) {
    const evah_ct *a = cts[j];
    if (a->limbs != f->limbs) throw std::invalid_argument("encrypted1 and encrypted2 parameter mismatch");
    if (a->size != f->size) throw std::invalid_argument("weighted_sum terms must have one size");
    if (a->batch != f->batch) throw std::invalid_argument("batch size mismatch");
    if (pts[j] && pts[j]->limbs != a->limbs) throw std::invalid_argument("encrypted and plain parameter mismatch");
    const double sj = a->scale * (pts[j] ? pts[j]->scale : 1.0);
    if (pts[j]) check_scale(c, sj, a->limbs);
    if (!same_scale(sj, scale)) throw std::invalid_argument("scale mismatch");
    acquire(c, a->buf);
    if (pts[j]) acquire(c, pts[j]->buf);
    tab.ct[j] = a->d;
    tab.pt[j] = pts[j] ? pts[j]->d : nullptr;
    tab.ct_ps[j] = (uint32_t)(a->ps / c->N);
  }
  evah_ct *o = ct_new(c, f->size, f->limbs, scale, f->batch);
  EW_LAUNCH(k_weighted_sum, ew_grid(c, f->limbs, f->size * f->batch), dim3(256), 0, c->stream, c->dev, tab, n, o->d, o->ps);
  HIPCHK(hipGetLastError());
  *out = o;
  API_END
}

int evah_relinearize(evah_ctx *c, const evah_ct *a, evah_ct **out) {
  API_BEGIN
  use(c);
  acquire(c, a->buf);
  if (a->size != 3) throw std::invalid_argument("relinearize expects a size-3 ciphertext");
  if (!c->sh->relin.d) throw std::invalid_argument("relinearization key not present");
  evah_ct *o = ct_new(c, 2, a->limbs, a->scale, a->batch);
  try {
    if (a->batch == 1) {
      switch_key(c, a->limbs, a->d + 2 * a->ps, c->sh->relin, a->d, a->ps, 2, o->d, o->ps);
    } else { // all instances in one launch set (chunks of KS_BATCH_MAX)
      const uint32_t l = a->limbs;
      const size_t N = c->N, pps = (size_t)(l + 1) * N;
      for (uint32_t b0 = 0; b0 < a->batch; b0 += KS_BATCH_MAX) {
        const uint32_t n = std::min<uint32_t>(KS_BATCH_MAX, a->batch - b0);
        const u64 *a0 = a->d + (size_t)b0 * 3 * a->ps;
        Scratch prod(c, (size_t)n * 2 * pps);
        std::vector<const KeyDev *> keys(n, &c->sh->relin);
        switch_key_products(c, l, a0 + 2 * a->ps, 3 * a->ps, keys.data(), n, prod.d);
        Scratch r(c, (size_t)n * 2 * N);
        OpPlain::Params sp{prod.d + (size_t)l * N, r.d, pps, N, 1, c->k - 1, 1, {}};
        OpModDown::Params mp{r.d, N, prod.d, pps, a0, a->ps, 2, o->d + (size_t)b0 * 2 * o->ps, o->ps, c->k - 1, l};
        mp.add_bs = 3 * a->ps;
        inverse_then_forward<OpPlain, OpModDown>(c, sp, 2 * n, mp, 2 * n * l);
      }
    }
  } catch (...) {
    evah_ct_free(c, o);
    throw;
  }
  *out = o;
  API_END
}

static void relin_rescale_core(evah_ctx *c, const evah_ct *const *as, uint32_t n, u64 *out_d, const MulTab *mul = nullptr,
                               uint32_t mul_limbs = 0);

int evah_relinearize_rescale(evah_ctx *c, const evah_ct *a, uint32_t divisor_bits, evah_ct **out) {
  API_BEGIN
  use(c);
  acquire(c, a->buf);
  if (a->size != 3) throw std::invalid_argument("relinearize expects a size-3 ciphertext");
  if (!c->sh->relin.d) throw std::invalid_argument("relinearization key not present");
  if (a->limbs < 2) throw std::invalid_argument("end of modulus switching chain reached");
  const uint32_t l = a->limbs, last = l - 1, sp = c->k - 1;
  const size_t N = c->N, pps = (size_t)(l + 1) * N;
  evah_ct *o = ct_new(c, 2, l - 1, a->scale / std::pow(2.0, (double)divisor_bits), a->batch);
  if (a->batch > 1) { // every instance through the batched form, KS_BATCH_MAX at a time
    try {
      std::vector<evah_ct> views(a->batch, *a);
      std::vector<const evah_ct *> ptrs(a->batch);
      for (uint32_t b = 0; b < a->batch; b++) {
        views[b].d = a->d + (size_t)b * 3 * a->ps;
        views[b].batch = 1;
        ptrs[b] = &views[b];
      }
      for (uint32_t b0 = 0; b0 < a->batch; b0 += KS_BATCH_MAX)
        relin_rescale_core(c, ptrs.data() + b0, std::min<uint32_t>(KS_BATCH_MAX, a->batch - b0),
                           o->d + (size_t)b0 * 2 * o->ps);
    } catch (...) {
      evah_ct_free(c, o);
      throw;
    }
    *out = o;
    g_err.clear();
    return 0;
  }
  try {
    Scratch prod(c, 2 * pps);
    const KeyDev *kp = &c->sh->relin;
    switch_key_products(c, l, a->d + 2 * a->ps, 0, &kp, 1, prod.d);
    Scratch r(c, 2 * N), t(c, 2 * N);
    // r_K = INTT_P(prod[K][special]) + P/2
    OpPlain::Params spp{prod.d + (size_t)l * N, r.d, pps, N, 1, sp, 1, {}};
    ntt_inverse<OpPlain>(c, spp, 2);
    // t_K = INTT_last(a[K][last] + prod[K][last] P^-1) - u_K,last P^-1 + q_last/2
    OpRRLast::Params lp{a->d + (size_t)last * N, a->ps, prod.d + (size_t)last * N, pps, r.d, N, t.d, N, last, sp, {}};
    ntt_inverse<OpRRLast>(c, lp, 2);
    // out[K][i] = (a[K][i] + prod[K][i] P^-1 - NTT_i(u P^-1 + v)) q_last^-1
    OpRR::Params rp{r.d, N, t.d, N, a->d, a->ps, prod.d, pps, o->d, o->ps, sp, last, l - 1, {}};
    ntt_forward<OpRR>(c, rp, 2 * (l - 1));
  } catch (...) {
    evah_ct_free(c, o);
    throw;
  }
  *out = o;
  API_END
}

// n (<= 64) independent size-3 ciphertexts at the same level, all relinearized with the (shared)
// relinearization key and rescaled: one set of n-times-wider launches; instances are co-scheduled
// per XCD so the key tiles are read from HBM once per XCD, not once per instance.
// core of the batched form: n (<= KS_BATCH_MAX) size-3 ciphertexts at one level -> out_d[n][2][(l-1) N]
// mul != nullptr: instance b is the product a[b] x b[b] of mul (size-2 operands at mul_limbs limbs),
// its polynomials d0, d1, d2 evaluated where they are consumed (as == nullptr then)
static void relin_rescale_core(evah_ctx *c, const evah_ct *const *as, uint32_t n, u64 *out_d, const MulTab *mul, uint32_t mul_limbs) {
  const uint32_t l = mul ? mul_limbs : as[0]->limbs;
  const uint32_t last = l - 1, sp = c->k - 1;
  const size_t N = c->N, pps = (size_t)(l + 1) * N, ops = (size_t)(l - 1) * N;
  PtrTab c2{}, a_last{}, a_polys{};
  for (uint32_t b = 0; b < n && !mul; b++) {
    const evah_ct *a = as[b];
    c2.p[b] = a->d + 2 * a->ps;
    for (uint32_t K = 0; K < 2; K++) {
      a_last.p[2 * b + K] = a->d + K * a->ps + (size_t)last * N;
      a_polys.p[2 * b + K] = a->d + K * a->ps;
    }
  }
  Scratch prod(c, (size_t)n * 2 * pps);
  std::vector<const KeyDev *> keys(n, &c->sh->relin);
  switch_key_products(c, l, nullptr, 0, keys.data(), n, prod.d, &c2, mul);
  Scratch r(c, (size_t)n * 2 * N), t(c, (size_t)n * 2 * N);
  OpPlain::Params spp{prod.d + (size_t)l * N, r.d, pps, N, 1, sp, 1, {}};
  ntt_inverse<OpPlain>(c, spp, 2 * n);
  if (mul) {
    OpRRLastMul::Params lp{nullptr, 0, prod.d + (size_t)last * N, pps, r.d, N, t.d, N, last, sp, a_last, *mul};
    ntt_inverse<OpRRLastMul>(c, lp, 2 * n);
    OpRRMul::Params rp{r.d, N, t.d, N, nullptr, 0, prod.d, pps, out_d, ops, sp, last, l - 1, a_polys, *mul};
    ntt_forward<OpRRMul>(c, rp, 2 * n * (l - 1));
  } else {
    OpRRLast::Params lp{nullptr, 0, prod.d + (size_t)last * N, pps, r.d, N, t.d, N, last, sp, a_last};
    ntt_inverse<OpRRLast>(c, lp, 2 * n);
    OpRR::Params rp{r.d, N, t.d, N, nullptr, 0, prod.d, pps, out_d, ops, sp, last, l - 1, a_polys};
    ntt_forward<OpRR>(c, rp, 2 * n * (l - 1));
  }
}

// multiply (size 2 x size 2) -> relinearize -> rescale_to_next for n (<= 64) independent pairs at one
// level, the three SEAL calls of seal_executor.h:164, :200, :213-214 evaluated together: the size-3
// product is never materialised — d2 = a1 b1 is formed in the load of the digit inverse
// transform (and in the key-switch kernel where the NTT-form digit is used as is), d0 and d1 in
// the epilogue that combines them with the key-switch result.  Same ciphertext, bit for bit.
static void mul_relin_rescale(evah_ctx *c, const evah_ct *const *as, const evah_ct *const *bs, uint32_t n, uint32_t divisor_bits,
                              evah_ct **outs) {
  if (n < 1 || n > (uint32_t)KS_BATCH_MAX) throw std::invalid_argument("multiply_relinearize_rescale_many handles 1..64 products per call");
  if (!c->sh->relin.d) throw std::invalid_argument("relinearization key not present");
  const uint32_t l = as[0]->limbs;
  if (l < 2) throw std::invalid_argument("end of modulus switching chain reached");
  const size_t N = c->N, ops = (size_t)(l - 1) * N;
  MulTab tab{};
  std::vector<double> scales(n);
  for (uint32_t i = 0; i < n; i++) {
    const evah_ct *a = as[i], *b = bs[i];
    if (a->size != 2 || b->size != 2) throw std::invalid_argument("multiply supports size-2 operands only (relinearize first)");
    if (a->batch != 1 || b->batch != 1) throw std::invalid_argument("multiply_relinearize_rescale_many takes single ciphertexts");
    if (a->limbs != l || b->limbs != l) throw std::invalid_argument("encrypted parameter mismatch in batch");
    scales[i] = a->scale * b->scale;
    check_scale(c, scales[i], l);
    acquire(c, a->buf);
    acquire(c, b->buf);
    tab.a[i] = a->d;
    tab.b[i] = b->d;
    tab.a_ps[i] = (uint32_t)(a->ps / N);
    tab.b_ps[i] = (uint32_t)(b->ps / N);
  }
  Buffer *ob = buf_new(c, (size_t)n * 2 * ops);
  try {
    relin_rescale_core(c, nullptr, n, ob->d, &tab, l);
  } catch (...) {
    buf_unref(c, ob);
    throw;
  }
  ob->refs = (int)n;
  for (uint32_t b = 0; b < n; b++) {
    evah_ct *t = new evah_ct;
    t->buf = ob;
    t->d = ob->d + (size_t)b * 2 * ops;
    t->size = 2;
    t->limbs = l - 1;
    t->ps = ops;
    t->scale = scales[b] / std::pow(2.0, (double)divisor_bits);
    outs[b] = t;
  }
}

int evah_relinearize_rescale_many(evah_ctx *c, const evah_ct *const *as, uint32_t n, uint32_t divisor_bits, evah_ct **outs) {
  API_BEGIN
  use(c);
  if (n < 1 || n > (uint32_t)KS_BATCH_MAX) throw std::invalid_argument("relinearize_rescale_many handles 1..64 ciphertexts per call");
  if (!c->sh->relin.d) throw std::invalid_argument("relinearization key not present");
  const uint32_t l = as[0]->limbs;
  if (l < 2) throw std::invalid_argument("end of modulus switching chain reached");
  const size_t N = c->N, ops = (size_t)(l - 1) * N;
  for (uint32_t b = 0; b < n; b++) {
    const evah_ct *a = as[b];
    if (a->size != 3) throw std::invalid_argument("relinearize expects a size-3 ciphertext");
    if (a->batch != 1) throw std::invalid_argument("relinearize_rescale_many takes single ciphertexts (a batched handle goes through evah_relinearize_rescale)");
    if (a->limbs != l) throw std::invalid_argument("encrypted parameter mismatch in batch");
    acquire(c, a->buf);
  }
  Buffer *ob = buf_new(c, (size_t)n * 2 * ops);
  try {
    relin_rescale_core(c, as, n, ob->d);
  } catch (...) {
    buf_unref(c, ob);
    throw;
  }
  ob->refs = (int)n;
  for (uint32_t b = 0; b < n; b++) {
    evah_ct *t = new evah_ct;
    t->buf = ob;
    t->d = ob->d + (size_t)b * 2 * ops;
    t->size = 2;
    t->limbs = l - 1;
    t->ps = ops;
    t->scale = as[b]->scale / std::pow(2.0, (double)divisor_bits);
    outs[b] = t;
  }
  API_END
}

int evah_multiply_relinearize_rescale_many(evah_ctx *c, const evah_ct *const *as, const evah_ct *const *bs, uint32_t n,
                                           uint32_t divisor_bits, evah_ct **outs) {
  API_BEGIN
  use(c);
  if (!c->fuse_mac) { // EVAH_FUSE_MAC=0 (the unfused reference path): the three calls one after the other
    std::vector<evah_ct *> ms(n, nullptr);
    if (evah_multiply_many(c, as, bs, n, ms.data())) throw std::runtime_error(g_err);
    int rc = evah_relinearize_rescale_many(c, ms.data(), n, divisor_bits, outs);
    std::string err = g_err;
    for (evah_ct *m : ms) evah_ct_free(c, m);
    if (rc) throw std::runtime_error(err);
  } else {
    mul_relin_rescale(c, as, bs, n, divisor_bits, outs);
  }
  API_END
}

int evah_multiply_relinearize_rescale(evah_ctx *c, const evah_ct *a, const evah_ct *b, uint32_t divisor_bits, evah_ct **out) {
  API_BEGIN
  use(c);
  if (a->batch != 1 || b->batch != 1 || !c->fuse_mac) { // batched handles: the separate (already batched) calls
    evah_ct *m = nullptr;
    if (evah_multiply(c, a, b, &m)) throw std::runtime_error(g_err);
    int rc = evah_relinearize_rescale(c, m, divisor_bits, out);
    std::string err = g_err;
    evah_ct_free(c, m);
    if (rc) throw std::runtime_error(err);
  } else {
    mul_relin_rescale(c, &a, &b, 1, divisor_bits, out);
  }
  API_END
}

// NTT-domain permutation table of a Galois element (SEAL GaloisTool::generate_table_ntt), cached
static const uint32_t *perm_table(evah_ctx *c, uint32_t elt) {
  auto pit = c->sh->perms.find(elt);
  if (pit != c->sh->perms.end()) return pit->second;
  if (c->capturing) throw std::logic_error("first use of a Galois element cannot be captured into a graph");
  const size_t N = c->N;
  std::vector<uint32_t> tab(N);
  for (uint32_t i = 0; i < N; i++) {
    uint32_t reversed = bitrev((uint32_t)N + i, c->logN + 1);
    u64 raw = (((u64)elt * reversed) >> 1) & (u64)(N - 1);
    tab[i] = bitrev((uint32_t)raw, c->logN);
  }
  uint32_t *d = nullptr;
  HIPCHK(hipMalloc(&d, sizeof(uint32_t) * N));
  HIPCHK(hipMemcpy(d, tab.data(), sizeof(uint32_t) * N, hipMemcpyHostToDevice));
  c->sh->perms.emplace(elt, d);
  return d;
}

// Hoisted rotations, tables (first use of a Galois element / level: not capturable, like perm_table).
// sign[k][N]: NTT under every prime of the 0/1 polynomial marking the coefficients whose sign the
// automorphism flips (SEAL GaloisTool::apply_galois: index_raw = i * elt, bit logN of it set).
static const u64 *hoist_sign(evah_ctx *c, uint32_t elt) {
  auto it = c->sh->hoist_sign.find(elt);
  if (it != c->sh->hoist_sign.end()) return it->second;
  if (c->capturing) throw std::logic_error("first hoisted use of a Galois element cannot be captured into a graph");
  const size_t N = c->N;
  std::vector<u64> s(N * c->k, 0);
  for (uint32_t i = 0; i < N; i++) {
    const u64 raw = (u64)i * elt;
    if ((raw >> c->logN) & 1) s[raw & (N - 1)] = 1;
  }
  for (uint32_t p = 1; p < c->k; p++) std::copy_n(s.begin(), N, s.begin() + (size_t)p * N);
  u64 *d = nullptr;
  HIPCHK(hipMalloc(&d, sizeof(u64) * N * c->k));
  try {
    HIPCHK(hipMemcpyAsync(d, s.data(), sizeof(u64) * N * c->k, hipMemcpyHostToDevice, c->stream));
    OpPlain::Params p{d, d, 0, 0, c->k, 0, 0, {}};
    ntt_forward<OpPlain>(c, p, c->k);
    HIPCHK(hipStreamSynchronize(c->stream)); // `s` goes out of scope; other queues may use the table next
  } catch (...) {
    (void)hipFree(d);
    throw;
  }
  c->sh->hoist_sign.emplace(elt, d);
  return d;
}
static const u64 *hoist_corr(evah_ctx *c, uint32_t elt, uint32_t l, const KeyDev &key) {
  auto it = c->sh->hoist_corr.find({elt, l});
  if (it != c->sh->hoist_corr.end()) return it->second;
  if (c->capturing) throw std::logic_error("first hoisted use of a Galois element cannot be captured into a graph");
  const u64 *sign = hoist_sign(c, elt);
  u64 *d = nullptr;
  HIPCHK(hipMalloc(&d, sizeof(u64) * 2 * (l + 1) * c->N));
  hipLaunchKernelGGL(k_hoist_corr, dim3(c->N / 256, l + 1, 2), dim3(256), 0, c->stream, c->dev, sign, key.d, d, l);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  if (e != hipSuccess) {
    (void)hipFree(d);
    HIPCHK(e);
  }
  c->sh->hoist_corr.emplace(std::make_pair(elt, l), d);
  return d;
}

// ---- rotation sets.  A set is a list of (source ciphertext, Galois element) pairs at one level,
// issued KS_BATCH_MAX pairs at a time; pair r of a chunk writes out_d[r][2][l N].
struct RotPair {
  const u64 *src;   // c0 of the source ciphertext; c1 = src + src_ps
  size_t src_ps;
  uint32_t src_idx; // index into the set's distinct sources (hoisted digits)
  uint32_t elt;
  const KeyDev *key;
  const uint32_t *perm;
  const u64 *corr;  // hoisting constant of (elt, l); null when the set is not hoisted
};
struct RotChunk {
  uint32_t first, count; // pairs [first, first + count)
  u64 *out;
};
static RotPair rot_pair(evah_ctx *c, const u64 *src, size_t src_ps, uint32_t src_idx, int32_t step, uint32_t l, const char *who) {
  if (step == 0) throw std::invalid_argument(std::string(who) + ": zero steps are copies, not key switches");
  RotPair p{src, src_ps, src_idx, 0, nullptr, nullptr, nullptr};
  if (evah_galois_elt_from_step(c, step, &p.elt)) throw std::invalid_argument(g_err);
  auto kit = c->sh->galois.find(p.elt);
  if (kit == c->sh->galois.end()) throw std::invalid_argument("Galois key not present");
  if (kit->second.n_digits < l) throw std::runtime_error("key switching key has too few digits");
  p.key = &kit->second;
  p.perm = perm_table(c, p.elt);
  return p;
}
static void rot_perm_launch(evah_ctx *c, uint32_t l, const RotPair *pr, uint32_t np, u64 *perm_d, uint32_t polys) {
  PermPairs pt{};
  for (uint32_t r = 0; r < np; r++) {
    pt.perm[r] = pr[r].perm;
    pt.src[r] = pr[r].src;
    pt.src_ps[r] = (uint32_t)(pr[r].src_ps / c->N);
  }
  ProfScope ps(c, KC_EW);
  hipLaunchKernelGGL(k_galois_perm_pairs, dim3(c->N / 512, l, polys * np), dim3(256), 0, c->stream, c->dev, pt, perm_d, (size_t)l * c->N,
                     polys);
  HIPCHK(hipGetLastError());
}
// mod-down of a chunk's products (step 3 of switch_key); c0' = perm_d[2r] is added to the even polys
static void rot_mod_down(evah_ctx *c, uint32_t l, uint32_t np, u64 *prod_d, const u64 *perm_d, u64 *out_d, u64 *r_d, bool inv1) {
  const size_t N = c->N, pps = (size_t)l * N;
  // INTT of the special limbs, job = r*2 + K
  OpPlain::Params sp{prod_d + (size_t)l * N, r_d, (size_t)(l + 1) * N, N, 1, c->k - 1, 1, {}};
  // mod-down + combine, poly index pp = r*2 + K; c0' (even pp) is added, odd pp start from 0
  OpModDown::Params mp{r_d, N, prod_d, (size_t)(l + 1) * N, perm_d, pps, ~0u, out_d, pps, c->k - 1, l};
  inverse_then_forward<OpPlain, OpModDown>(c, sp, 2 * np, mp, 2 * np * l, inv1);
}
// SEAL's order — rotate, then decompose the rotated c1 (every launch honours c->dev.guard)
static void rot_chunk_plain(evah_ctx *c, uint32_t l, const RotPair *pr, uint32_t np, u64 *out_d) {
  const size_t N = c->N, pps = (size_t)l * N, prod_bs = (size_t)2 * (l + 1) * N;
  std::vector<const KeyDev *> keys(np);
  for (uint32_t r = 0; r < np; r++) keys[r] = pr[r].key;
  Scratch perm(c, (size_t)np * 2 * pps); // [r][c0 permuted | c1 permuted = key-switch target]
  rot_perm_launch(c, l, pr, np, perm.d, 2);
  Scratch prod(c, np * prod_bs), r(c, (size_t)np * 2 * N);
  const bool inv1 = switch_key_products(c, l, perm.d + pps, 2 * pps, keys.data(), np, prod.d, nullptr, nullptr,
                                        fuse_small_launch(c, 2 * np * l) ? r.d : nullptr);
  rot_mod_down(c, l, np, prod.d, perm.d, out_d, r.d, inv1);
}
// The whole set.  hoisted: the digits of every distinct source are transformed once and each pair's
// key inner product is formed from them (k_hoist_mac / k_hoist_fix), then the unhoisted launches
// follow under the device-side guard.  srcs[i] = c0 of distinct source i (poly stride src_ps[i]).
static void rotation_set(evah_ctx *c, uint32_t l, const std::vector<RotPair> &pairs, const std::vector<RotChunk> &chunks,
                         const std::vector<const u64 *> &srcs, const std::vector<size_t> &src_ps, bool hoisted) {
  const size_t N = c->N, pps = (size_t)l * N, prod_bs = (size_t)2 * (l + 1) * N;
  if (!hoisted) {
    for (const RotChunk &ch : chunks) rot_chunk_plain(c, l, pairs.data() + ch.first, ch.count, ch.out);
    return;
  }
  const uint32_t n_src = (uint32_t)srcs.size();
  if (n_src > (uint32_t)KS_BATCH_MAX) throw std::logic_error("hoisted rotation set with too many sources");
  Scratch flag(c, 1 + HOIST_ZERO_CAP); // [0]: zero-coefficient count, then the recorded positions
  HIPCHK(hipMemsetAsync(flag.d, 0, sizeof(u64), c->stream));
  const size_t dg_bs = (size_t)(l + 1) * l * N;
  {
    // digits of the unrotated c1 of every source, once: coefficient form (zeros recorded), then the
    // full transforms under every output prime
    Scratch t(c, (size_t)n_src * l * N), dg(c, n_src * dg_bs);
    PtrTab c1{};
    for (uint32_t i = 0; i < n_src; i++) c1.p[i] = srcs[i] + src_ps[i];
    OpPlainZ::Params ip{nullptr, t.d, 0, (size_t)l * N, l, 0, 0, c1};
    ip.zero_list = flag.d;
    ntt_inverse<OpPlainZ>(c, ip, n_src * l);
    OpKsDigit::Params dp{t.d, dg.d, l, (size_t)l * N, dg_bs, 0, l + 1};
    ntt_forward<OpKsDigit>(c, dp, n_src * (l + 1) * l);
    for (const RotChunk &ch : chunks) {
      const RotPair *pr = pairs.data() + ch.first;
      const uint32_t np = ch.count;
      HoistTab ht{};
      for (uint32_t r = 0; r < np; r++) {
        ht.perm[r] = pr[r].perm;
        ht.key[r] = pr[r].key->d;
        ht.corr[r] = pr[r].corr;
        ht.elt[r] = pr[r].elt;
        ht.c1[r] = pr[r].src + pr[r].src_ps;
        ht.src[r] = (uint8_t)pr[r].src_idx;
      }
      Scratch perm(c, (size_t)np * 2 * pps); // only the c0 slots (even polys) are filled and read
      rot_perm_launch(c, l, pr, np, perm.d, 1);
      Scratch prod(c, np * prod_bs), r(c, (size_t)np * 2 * N);
      {
        ProfScope ps(c, KC_KSMAC);
        hipLaunchKernelGGL(k_hoist_mac, dim3(c->N / 512, l + 1, np), dim3(256), 0, c->stream, c->dev, dg.d, dg_bs, ht, prod.d, prod_bs, l);
        HIPCHK(hipGetLastError());
        // the terms of recorded zero coefficients (returns at once when there are none)
        hipLaunchKernelGGL(k_hoist_fix, dim3(c->N / 256, l + 1, np), dim3(256), 0, c->stream, c->dev, flag.d, ht, prod.d, prod_bs, l);
        HIPCHK(hipGetLastError());
      }
      rot_mod_down(c, l, np, prod.d, perm.d, ch.out, r.d, false);
    }
  }
  // exact fallback: the same outputs through the unhoisted launches, each a no-op unless there
  // were more zero digit coefficients than k_hoist_fix handles
  struct GuardScope {
    evah_ctx *c;
    GuardScope(evah_ctx *c_, const uint32_t *g) : c(c_) { c->dev.guard = g; c->dev.guard_min = HOIST_ZERO_CAP; }
    ~GuardScope() { c->dev.guard = nullptr; }
  } gs(c, reinterpret_cast<const uint32_t *>(flag.d));
  for (const RotChunk &ch : chunks) rot_chunk_plain(c, l, pairs.data() + ch.first, ch.count, ch.out);
  if (!c->capturing && std::getenv("EVAH_HOIST_DEBUG")) { // diagnostics: how many zero coefficients did this set see?
    uint32_t f = 0;
    HIPCHK(hipMemcpyAsync(&f, flag.d, sizeof(f), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    std::fprintf(stderr, "rotation set: hoisted %zu rotations of %u sources, l = %u, zero coefficients = %u\n", pairs.size(), n_src, l, f);
  }
}

// Several rotations of ONE ciphertext (the convolution pattern: image << i*w+j for a 3x3
// window) issued as one set of wide launches: same results as n evah_rotate calls, 1/n of the
// kernel launches, each launch n times wider.  Large launch sets are hoisted (see k_hoist_mac).
int evah_rotate_many(evah_ctx *c, const evah_ct *a, const int32_t *steps, uint32_t n, evah_ct **outs) {
  API_BEGIN
  use(c);
  acquire(c, a->buf);
  if (a->size != 2) throw std::invalid_argument("rotate expects a size-2 ciphertext (relinearize first)");
  if (n < 1 || n > (uint32_t)KS_BATCH_MAX) throw std::invalid_argument("rotate_many handles 1..64 rotations per call");
  const uint32_t l = a->limbs, B = a->batch;
  const size_t pps = (size_t)l * c->N;
  const bool hoisted = hoist_wanted(c, l, n, B);
  // (rotation j, instance b) pairs go out KS_BATCH_MAX at a time: m rotations x B instances per
  // launch set; pair index r = j * B + b, so rotation j's B outputs are one batched handle.
  std::vector<RotPair> pairs;
  pairs.reserve((size_t)n * B);
  for (uint32_t j = 0; j < n; j++) {
    RotPair p = rot_pair(c, a->d, a->ps, 0, steps[j], l, "rotate_many");
    if (hoisted) p.corr = hoist_corr(c, p.elt, l, *p.key);
    for (uint32_t b = 0; b < B; b++) {
      p.src = a->d + (size_t)b * 2 * a->ps;
      p.src_idx = b;
      pairs.push_back(p);
    }
  }
  std::vector<const u64 *> srcs(B);
  std::vector<size_t> src_ps(B, a->ps);
  for (uint32_t b = 0; b < B; b++) srcs[b] = a->d + (size_t)b * 2 * a->ps;
  const uint32_t m_max = std::max<uint32_t>(1, KS_BATCH_MAX / B);
  std::vector<evah_ct *> made;
  std::vector<Buffer *> chunk_buf;
  std::vector<RotChunk> chunks;
  try {
    for (uint32_t j0 = 0; j0 < n; j0 += m_max) {
      const uint32_t m = std::min(m_max, n - j0), np = m * B;
      Buffer *ob = buf_new(c, (size_t)np * 2 * pps); // one buffer for the chunk; the m handles are views into it
      ob->refs = 0;
      chunk_buf.push_back(ob);
      chunks.push_back({j0 * B, np, ob->d});
      for (uint32_t j = 0; j < m; j++) {
        evah_ct *t = new evah_ct;
        t->buf = ob;
        ob->refs++;
        t->d = ob->d + (size_t)j * B * 2 * pps;
        t->size = 2;
        t->limbs = l;
        t->ps = pps;
        t->scale = a->scale;
        t->batch = B;
        made.push_back(t);
      }
    }
    rotation_set(c, l, pairs, chunks, srcs, src_ps, hoisted);
  } catch (...) {
    for (evah_ct *t : made) evah_ct_free(c, t);
    if (made.empty())
      for (Buffer *b : chunk_buf) { b->refs = 1; buf_unref(c, b); }
    throw;
  }
  for (uint32_t r = 0; r < n; r++) outs[r] = made[r];
  API_END
}

// n (<= 64) independent (ciphertext, step) rotations at one level as one launch set: the sibling
// rotations of SEVERAL ciphertexts (independent convolutions of one program level).  Sources that
// appear more than once share their digit decomposition when the set is large enough to hoist.
int evah_rotate_pairs(evah_ctx *c, const evah_ct *const *cts, const int32_t *steps, uint32_t n, evah_ct **outs) {
  API_BEGIN
  use(c);
  if (n < 1 || n > (uint32_t)KS_BATCH_MAX) throw std::invalid_argument("rotate_pairs handles 1..64 rotations per call");
  const uint32_t l = cts[0]->limbs;
  const size_t pps = (size_t)l * c->N;
  std::vector<RotPair> pairs;
  std::vector<const u64 *> srcs;
  std::vector<size_t> src_ps;
  for (uint32_t r = 0; r < n; r++) {
    const evah_ct *a = cts[r];
    if (a->size != 2) throw std::invalid_argument("rotate expects a size-2 ciphertext (relinearize first)");
    if (a->batch != 1) throw std::invalid_argument("rotate_pairs takes single ciphertexts");
    if (a->limbs != l) throw std::invalid_argument("encrypted parameter mismatch in batch");
    acquire(c, a->buf);
    uint32_t si = 0;
    while (si < srcs.size() && !(srcs[si] == a->d && src_ps[si] == a->ps)) si++;
    if (si == srcs.size()) {
      srcs.push_back(a->d);
      src_ps.push_back(a->ps);
    }
    pairs.push_back(rot_pair(c, a->d, a->ps, si, steps[r], l, "rotate_pairs"));
  }
  // worth hoisting when sources repeat and the digit transforms of the set are throughput-sized
  const bool hoisted = srcs.size() < n && hoist_wanted(c, l, n, 1);
  if (hoisted)
    for (RotPair &p : pairs) p.corr = hoist_corr(c, p.elt, l, *p.key);
  Buffer *ob = buf_new(c, (size_t)n * 2 * pps);
  try {
    rotation_set(c, l, pairs, {RotChunk{0, n, ob->d}}, srcs, src_ps, hoisted);
  } catch (...) {
    buf_unref(c, ob);
    throw;
  }
  ob->refs = (int)n;
  for (uint32_t r = 0; r < n; r++) {
    evah_ct *t = new evah_ct;
    t->buf = ob;
    t->d = ob->d + (size_t)r * 2 * pps;
    t->size = 2;
    t->limbs = l;
    t->ps = pps;
    t->scale = cts[r]->scale;
    outs[r] = t;
  }
  API_END
}

// n independent ciphertexts of one size and level rescaled in one launch set (n * size <= 128)
int evah_rescale_many(evah_ctx *c, const evah_ct *const *cts, uint32_t n, uint32_t divisor_bits, evah_ct **outs) {
  API_BEGIN
  use(c);
  const uint32_t size = cts[0]->size, l = cts[0]->limbs;
  if (n < 1 || (size_t)n * size > 2 * KS_BATCH_MAX) throw std::invalid_argument("rescale_many: too many polynomials for one call");
  if (l < 2) throw std::invalid_argument("end of modulus switching chain reached");
  const size_t N = c->N, ops = (size_t)(l - 1) * N;
  const uint32_t polys = n * size;
  PtrTab last{}, all{};
  for (uint32_t b = 0; b < n; b++) {
    const evah_ct *a = cts[b];
    if (a->size != size || a->limbs != l) throw std::invalid_argument("encrypted parameter mismatch in batch");
    if (a->batch != 1) throw std::invalid_argument("rescale_many takes single ciphertexts");
    acquire(c, a->buf);
    for (uint32_t p = 0; p < size; p++) {
      all.p[b * size + p] = a->d + (size_t)p * a->ps;
      last.p[b * size + p] = a->d + (size_t)p * a->ps + (size_t)(l - 1) * N;
    }
  }
  Buffer *ob = buf_new(c, (size_t)polys * ops);
  try {
    Scratch r(c, (size_t)polys * N);
    OpPlain::Params ip{nullptr, r.d, 0, N, 1, l - 1, 1, last};
        OpModDown::Params mp{r.d, N, nullptr, 0, nullptr, 0, 0, ob->d, ops, l - 1, l - 1};
    mp.c_tab = all;
    inverse_then_forward<OpPlain, OpModDown>(c, ip, polys, mp, polys * (l - 1));
  } catch (...) {
    buf_unref(c, ob);
    throw;
  }
  ob->refs = (int)n;
  for (uint32_t b = 0; b < n; b++) {
    evah_ct *t = new evah_ct;
    t->buf = ob;
    t->d = ob->d + (size_t)b * size * ops;
    t->size = size;
    t->limbs = l - 1;
    t->ps = ops;
    t->scale = cts[b]->scale / std::pow(2.0, (double)divisor_bits);
    outs[b] = t;
  }
  API_END
}

// n (<= 64) independent size-3 ciphertexts of one level relinearized in one launch set
int evah_relinearize_many(evah_ctx *c, const evah_ct *const *cts, uint32_t n, evah_ct **outs) {
  API_BEGIN
  use(c);
  if (n < 1 || n > (uint32_t)KS_BATCH_MAX) throw std::invalid_argument("relinearize_many handles 1..64 ciphertexts per call");
  if (!c->sh->relin.d) throw std::invalid_argument("relinearization key not present");
  const uint32_t l = cts[0]->limbs;
  const size_t N = c->N, pps = (size_t)(l + 1) * N, ops = (size_t)l * N;
  PtrTab c2{}, c01{};
  for (uint32_t b = 0; b < n; b++) {
    const evah_ct *a = cts[b];
    if (a->size != 3) throw std::invalid_argument("relinearize expects a size-3 ciphertext");
    if (a->batch != 1) throw std::invalid_argument("relinearize_many takes single ciphertexts");
    if (a->limbs != l) throw std::invalid_argument("encrypted parameter mismatch in batch");
    acquire(c, a->buf);
    c2.p[b] = a->d + 2 * a->ps;
    c01.p[2 * b] = a->d;
    c01.p[2 * b + 1] = a->d + a->ps;
  }
  Buffer *ob = buf_new(c, (size_t)n * 2 * ops);
  try {
    Scratch prod(c, (size_t)n * 2 * pps);
    std::vector<const KeyDev *> keys(n, &c->sh->relin);
    switch_key_products(c, l, nullptr, 0, keys.data(), n, prod.d, &c2);
    Scratch r(c, (size_t)n * 2 * N);
    OpPlain::Params sp{prod.d + (size_t)l * N, r.d, pps, N, 1, c->k - 1, 1, {}};
        OpModDown::Params mp{r.d, N, prod.d, pps, nullptr, 0, 0, ob->d, ops, c->k - 1, l};
    mp.use_add_tab = true;
    mp.add_tab = c01;
    inverse_then_forward<OpPlain, OpModDown>(c, sp, 2 * n, mp, 2 * n * l);
  } catch (...) {
    buf_unref(c, ob);
    throw;
  }
  ob->refs = (int)n;
  for (uint32_t b = 0; b < n; b++) {
    evah_ct *t = new evah_ct;
    t->buf = ob;
    t->d = ob->d + (size_t)b * 2 * ops;
    t->size = 2;
    t->limbs = l;
    t->ps = ops;
    t->scale = cts[b]->scale;
    outs[b] = t;
  }
  API_END
}

int evah_rotate(evah_ctx *c, const evah_ct *a, int32_t steps, evah_ct **out) {
  API_BEGIN
  use(c);
  acquire(c, a->buf);
  if (a->size != 2) throw std::invalid_argument("rotate expects a size-2 ciphertext (relinearize first)");
  const size_t N = c->N;
  if (steps == 0) { // SEAL rotate_internal: no-op — the result is the operand; handles are immutable, so share the buffer
    evah_ct *o = new evah_ct(*a);
    o->buf->refs++;
    *out = o;
  } else if (a->batch > 1) {
    if (evah_rotate_many(c, a, &steps, 1, out)) throw std::runtime_error(g_err);
  } else {
    uint32_t elt = 0;
    if (evah_galois_elt_from_step(c, steps, &elt)) throw std::invalid_argument(g_err);
    auto kit = c->sh->galois.find(elt);
    if (kit == c->sh->galois.end()) throw std::invalid_argument("Galois key not present");
    const uint32_t *ptab = perm_table(c, elt);
    Scratch perm(c, (size_t)2 * a->limbs * N); // [c0 permuted][c1 permuted = key-switch target]
    const size_t pps = (size_t)a->limbs * N;
    EW_LAUNCH(k_galois_perm, ew_grid(c, a->limbs, 2), dim3(256), 0, c->stream, c->dev, a->d, a->ps,
                       ptab, perm.d, pps);
    HIPCHK(hipGetLastError());
    evah_ct *o = ct_new(c, 2, a->limbs, a->scale);
    try {
      switch_key(c, a->limbs, perm.d + pps, kit->second, perm.d, pps, 1, o->d, o->ps);
    } catch (...) {
      evah_ct_free(c, o);
      throw;
    }
    *out = o;
  }
  API_END
}

int evah_rescale(evah_ctx *c, const evah_ct *a, uint32_t divisor_bits, evah_ct **out) {
  API_BEGIN
  use(c);
  acquire(c, a->buf);
  if (a->limbs < 2) throw std::invalid_argument("end of modulus switching chain reached");
  const uint32_t l = a->limbs;
  const size_t N = c->N;
  const uint32_t polys = a->size * a->batch; // a batched handle is batch * size polynomials at stride ps
  evah_ct *o = ct_new(c, a->size, l - 1, a->scale / std::pow(2.0, (double)divisor_bits), a->batch);
  Scratch r(c, (size_t)polys * N);
  OpPlain::Params ip{a->d + (size_t)(l - 1) * N, r.d, a->ps, N, 1, l - 1, 1, {}};
    OpModDown::Params mp{r.d, N, a->d, a->ps, nullptr, 0, 0, o->d, o->ps, l - 1, l - 1};
  inverse_then_forward<OpPlain, OpModDown>(c, ip, polys, mp, polys * (l - 1));
  *out = o;
  API_END
}

int evah_mod_switch(evah_ctx *c, const evah_ct *a, evah_ct **out) {
  API_BEGIN
  use(c);
  if (a->limbs < 2) throw std::invalid_argument("end of modulus switching chain reached");
  check_scale(c, a->scale, a->limbs - 1);
  // dropping the last limb is a view: same buffer, same poly stride, one limb fewer
  evah_ct *o = new evah_ct(*a);
  o->limbs = a->limbs - 1;
  o->buf->refs++;
  *out = o;
  API_END
}

int evah_test_ntt(evah_ctx *c, uint32_t prime_idx, int inverse, uint64_t *host) {
  API_BEGIN
  use(c);
  if (c->capturing) throw std::logic_error("this call synchronises with the host and cannot be captured into a graph");
  if (prime_idx >= c->k) throw std::invalid_argument("prime index out of range");
  Scratch s(c, c->N);
  HIPCHK(hipMemcpyAsync(s.d, host, sizeof(u64) * c->N, hipMemcpyHostToDevice, c->stream));
  OpPlain::Params p{s.d, s.d, 0, 0, 1, prime_idx, 0, {}};
  if (inverse) ntt_inverse<OpPlain>(c, p, 1);
  else ntt_forward<OpPlain>(c, p, 1);
  HIPCHK(hipMemcpyAsync(host, s.d, sizeof(u64) * c->N, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  API_END
}

} // extern "C"

#include "shard.hip.h"
#include "client.hip.h"
