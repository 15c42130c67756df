// rotation_sets.hip.h — the machinery of rotation sets, shared by rotate.hip (evah_rotate_many / evah_rotate_pairs) and
// windows.hip (evah_rotate_weighted_sums): the pair permutation kernel, the hoisted key inner products (k_hoist_mac,
// k_hoist_fix) with their tables (inverse permutations, permuted keys, the per-(element, level) constants, tiles), the
// digit decomposition shared by a set, the mod-down of a chunk, the un-hoisted chunk (SEAL's order) and the exact
// fallback's launch.  Everything here has internal linkage; the few functions other units call live in rotate.hip.
#pragma once
#include "launch.hip.h"
#include <array>
#include "rot_fallback.hip.h"

namespace evah {

bool hoist_wanted(const evah_ctx *c, uint32_t l, uint32_t n, uint32_t B); // rotate.hip
// K8 for (ciphertext, rotation) pairs: pair r reads its own source; grid.z = r * 2 + p
struct PermPairs {
  const uint32_t *perm[KS_BATCH_MAX];
  const u64 *src[KS_BATCH_MAX];
  uint32_t src_ps[KS_BATCH_MAX]; // poly strides in units of N coefficients
};
static __global__ void __launch_bounds__(256)
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
// Where the gathers go.  The first term above reads l (l+1) digit rows per pair THROUGH the permutation — a gather for every
// multiply.  Substituting m = perm(n):
//     sum_J D[I][J][m] * key[J][K][I][perm^-1(m)]                       — elementwise in the SOURCE's index space m
// so with a copy of the key whose rows are read through perm^-1 (KeyDev::d_perm, built at the first hoisted use) and the
// constant term stored the same way, the whole inner product is coalesced loads and the digits of a source are loaded
// once for all the rotations a workgroup serves.  The result E[z][K][I][m] stays in the source's index space; the
// rotated value is E[perm(n)], and its consumers — the mod-down's special-row inverse transform (OpPlainG) and its
// combine epilogue (OpModDownG / moddown_sum_kernel) — read it through the pair's table: 2 (l+1) gathered rows per pair
// instead of l (l+1), and none inside the multiply loop.
//
// E[z][K][I][m] = sum_J D_s[I][J][m] * keyp_z[J][K][I][m] + corrp_z[K][I][m]  (+ P * c0_s[I][m] for K = 0, I < l: the
//   rotated c0 the key-switch result is added to, carried through the mod-down by its factor P as KS_FOLDADD does)
// D_s[I][J] is row (I * l + J) of source s's converted digits, or limb J of the source's own c1 when I == J (SEAL's
// shortcut: the NTT-form limb is used as is).
// A workgroup serves a TILE of up to HT_S sources x HT_R Galois elements (every combination that is a pair of the chunk):
// per digit J it loads HT_S digit words and 2 HT_R key words and does 2 HT_S HT_R multiply-accumulates, so digits are
// shared by the elements of a tile and keys by its sources (the instances of a batched handle, the three convolutions
// of Harris).  grid = (N / 256, l + 1, tiles), one coefficient per thread.
// The tile shape (TS sources x TR elements, TS TR <= 8) is a template parameter chosen per launch from the chunk's
// shape, and every tile of a launch is full — short ones are padded with repeats whose results are not stored — so the
// loops below have no exits: all TS + 2 TR loads of a digit step are in flight together.
constexpr int HT_TILES = 32; // tiles per launch (the tables travel as kernel arguments)
struct HoistMacTab {
  const u64 *c1[KS_BATCH_MAX];    // per source of the chunk: its c1 (NTT form); c0 = c1 - c1_ps * N
  uint32_t c1_ps[KS_BATCH_MAX];   // poly stride in units of N coefficients
  uint32_t dg[KS_BATCH_MAX];      // index of the source among the set's transformed digits
  const u64 *keyp[KS_BATCH_MAX];  // per Galois element of the chunk: the permuted key, the permuted constant [2][l+1][N]
  const u64 *corrp[KS_BATCH_MAX];
  uint32_t keyp_nd[KS_BATCH_MAX]; // digits of that key: a block of keyp is keyp_nd * 512 words
  // per tile, one byte per entry: sources [0..1], elements [2..3], pair (output slot) of combination s * TR + r [4..5]
  // (0xff: not a pair of the chunk)
  uint32_t tile[HT_TILES][6];
};
struct HoistFixTab { // per pair (k_hoist_fix)
  const uint32_t *perm[KS_BATCH_MAX];
  const u64 *key[KS_BATCH_MAX]; // the key as uploaded
  uint32_t elt[KS_BATCH_MAX];
  uint8_t src[KS_BATCH_MAX];    // index of the source among the set's transformed digits
};
// corrp[K][I][m] = corr[K][I][pinv[m]],  corr[K][I][n] = sign[kap][n] * sum_J (q_J mod q_kap) * key[J][K][kap][n]   (kap = prime of row I)
static __global__ void __launch_bounds__(256)
k_hoist_corr(DevCtx cx, const u64 *sign, const u64 *key, const uint32_t *pinv, u64 *corr, uint32_t l) {
  const uint32_t I = blockIdx.y, K = blockIdx.z, kap = (I == l) ? cx.k - 1 : I;
  const size_t m = (size_t)blockIdx.x * blockDim.x + threadIdx.x, n = pinv[m];
  const DevPrime pm = cx.primes[kap];
  const size_t N = cx.N, key_digit = (size_t)2 * cx.k * N;
  u64 acc = 0;
  for (uint32_t J = 0; J < l; J++) {
    const u64 qj = cx.primes[J].q % pm.q; // 0 when J == I
    acc = addmod(acc, mulmod(qj, key[J * key_digit + ((size_t)K * cx.k + kap) * N + n], pm), pm.q);
  }
  corr[((size_t)K * (l + 1) + I) * N + m] = mulmod(sign[(size_t)kap * N + n], acc, pm);
}
// The permuted key copy, keyp[kap][x][J][K][256] = key[J][K][kap][pinv[256 x + .]]: every row of the key read through the
// inverse permutation AND (r6) regrouped so that the words one workgroup of k_hoist_mac needs — prime row kap, 256
// coefficients, every digit J, both polynomials K — are ONE contiguous block of n_digits * 4 KiB.  With the key's own
// layout [J][K][kap][N] a workgroup took 2 KiB from each of 2 l rows megabytes apart, and the launch as a whole walked
// several hundred DRAM streams at once (config 5's window: 1.3 GB of keys at 3.9 TB/s).  grid = (N / 256, rows of the key)
// (keyp_block_words, internal.hip.h: one 2 KiB pad per block)
static __global__ void __launch_bounds__(256)
k_key_perm(const u64 *key, const uint32_t *pinv, u64 *out, uint32_t N, uint32_t k, uint32_t nd) {
  const uint32_t row = blockIdx.y, J = row / (2 * k), K = (row / k) & 1u, kap = row % k;
  const uint32_t m = blockIdx.x * blockDim.x + threadIdx.x;
  out[((size_t)kap * gridDim.x + blockIdx.x) * keyp_block_words(nd) + (J * 2 + K) * 256 + threadIdx.x] = key[(size_t)row * N + pinv[m]];
}
// r6 — which workgroups run together.  Tiles of a launch share operands: tiles with the same sources read the same digit
// rows, tiles with the same elements the same key rows (8 instances x 8 rotations as 4 x 2 tiles: every key row is read by
// 2 tiles, every digit row by 4).  With grid = (N/256, l+1, tiles) the workgroups that share a row are (N/256)(l+1)
// dispatches apart — further than the L2s and the Infinity Cache reach at N = 2^15, l = 8 — so a 64-pair launch moved
// 1.5 GB where 0.75 GB are distinct.  MAP = 1: a 1-D grid whose linear id is cut as (xcd | tile | x_hi | I): the dispatcher
// places block b on XCD b mod 8 (MI355X_MICROARCH.md, workgroup dispatch), so the eight blocks of one round have eight
// different coefficient ranges, and an XCD sees ALL tiles of its (x, I) back to back — the second reader of a row finds it
// in that XCD's L2.  Placement only decides speed; the result does not depend on it.
// V = coefficients per thread (2: 16-byte accesses, for the shapes with few accumulators).
#ifndef EVAH_HOIST_NT
#define EVAH_HOIST_NT 1 // (r6 default: the key stream is read once — config 5 -3 %, Harris -4 %, the batches unchanged)
// build-time experiment: 1 = the key stream with non-temporal loads, 2 = the products with non-temporal stores, 3 = both
#endif
#ifndef EVAH_HOIST_UNROLL
#define EVAH_HOIST_UNROLL 2 // digit steps whose loads are in flight together
#endif
template <int V> struct HmVec { u64 v[V]; };
template <int V, bool NT = false> __device__ __forceinline__ HmVec<V> hm_ld(const u64 *p) {
  HmVec<V> o;
  if constexpr (NT) {
#pragma unroll
    for (int v = 0; v < V; v++) o.v[v] = __builtin_nontemporal_load(p + v);
  } else if constexpr (V == 2) { const ulonglong2 t = ld2(p); o.v[0] = t.x; o.v[1] = t.y; }
  else o.v[0] = p[0];
  return o;
}
template <int V, bool NT = false> __device__ __forceinline__ void hm_st(u64 *p, const HmVec<V> &x) {
  if constexpr (NT) {
#pragma unroll
    for (int v = 0; v < V; v++) __builtin_nontemporal_store(x.v[v], p + v);
  } else if constexpr (V == 2) { ulonglong2 t; t.x = x.v[0]; t.y = x.v[1]; st2(p, t); }
  else p[0] = x.v[0];
}
template <int TS, int TR, int V, bool MAP>
__global__ void __launch_bounds__(256)
k_hoist_mac(DevCtx cx, const u64 *digits, size_t dg_bs, HoistMacTab tab, uint32_t n_tiles, u64 *prod, size_t prod_bs, uint32_t l, bool fold_c0) {
  uint32_t xt, I, zt;
  if constexpr (MAP) { // (host: N / (256 V) is a multiple of 8)
    const uint32_t xh_log = cx.logN - (V == 2 ? 9 : 8) - 3;
    uint32_t r = blockIdx.x >> 3;
    zt = r % n_tiles;
    r /= n_tiles;
    xt = ((r & ((1u << xh_log) - 1u)) << 3) | (blockIdx.x & 7u);
    I = r >> xh_log;
  } else {
    xt = blockIdx.x, I = blockIdx.y, zt = blockIdx.z;
  }
  const uint32_t kap = (I == l) ? cx.k - 1 : I;
  const uint32_t *tw = tab.tile[zt]; // wave-uniform: scalar loads, the bytes are cut out with scalar shifts
  const u64 srcs = tw[0] | ((u64)tw[1] << 32), rots = tw[2] | ((u64)tw[3] << 32), outs = tw[4] | ((u64)tw[5] << 32);
  const size_t m = (size_t)V * ((size_t)xt * blockDim.x + threadIdx.x);
  const DevPrime pm = cx.primes[kap];
  const size_t N = cx.N;
  const u64 *dgp[TS], *own[TS], *kp[TR];
#pragma unroll
  for (int s = 0; s < TS; s++) {
    const uint32_t si = (uint32_t)(srcs >> (8 * s)) & 0xffu;
    dgp[s] = digits + tab.dg[si] * dg_bs + (size_t)I * l * N + m;
    own[s] = tab.c1[si] + m;
  }
  static_assert(V == 1, "the blocked key copy is addressed by 256-coefficient blocks");
#pragma unroll
  for (int r = 0; r < TR; r++) {
    const uint32_t ei = (uint32_t)(rots >> (8 * r)) & 0xffu;
    kp[r] = tab.keyp[ei] + ((size_t)kap * (N >> 8) + xt) * keyp_block_words(tab.keyp_nd[ei]) + threadIdx.x;
  }
  u128_t a0[TS][TR][V], a1[TS][TR][V];
#pragma unroll
  for (int s = 0; s < TS; s++)
#pragma unroll
    for (int r = 0; r < TR; r++)
#pragma unroll
      for (int v = 0; v < V; v++) a0[s][r][v] = a1[s][r][v] = {0, 0};
  // operands are canonical (< q < 2^60): 256 products fit the 128-bit accumulators, l <= k - 1 < 64
#pragma unroll EVAH_HOIST_UNROLL
  for (uint32_t J = 0; J < l; J++) {
    HmVec<V> d[TS], k0[TR], k1[TR];
#pragma unroll
    for (int s = 0; s < TS; s++) d[s] = hm_ld<V>((I == J) ? own[s] + (size_t)J * N : dgp[s] + (size_t)J * N);
#pragma unroll
    for (int r = 0; r < TR; r++) {
      k0[r] = hm_ld<V, (EVAH_HOIST_NT & 1) != 0>(kp[r] + J * 512);
      k1[r] = hm_ld<V, (EVAH_HOIST_NT & 1) != 0>(kp[r] + J * 512 + 256);
    }
    __builtin_amdgcn_sched_barrier(0); // all the loads of the step are issued before the first multiply waits for one
#pragma unroll
    for (int r = 0; r < TR; r++)
#pragma unroll
      for (int s = 0; s < TS; s++)
#pragma unroll
        for (int v = 0; v < V; v++) {
          acc128c(a0[s][r][v], d[s].v[v], k0[r].v[v]);
          acc128c(a1[s][r][v], d[s].v[v], k1[r].v[v]);
        }
  }
  if (fold_c0 && I < l) { // block-uniform
    const u64 pmod = cx.modq[(size_t)(cx.k - 1) * cx.k + kap].x; // P mod q_I
#pragma unroll
    for (int s = 0; s < TS; s++) {
      const uint32_t si = (uint32_t)(srcs >> (8 * s)) & 0xffu;
      const HmVec<V> c0v = hm_ld<V>((own[s] - (size_t)tab.c1_ps[si] * N) + (size_t)I * N);
#pragma unroll
      for (int r = 0; r < TR; r++)
#pragma unroll
        for (int v = 0; v < V; v++) acc128c(a0[s][r][v], c0v.v[v], pmod);
    }
  }
#pragma unroll
  for (int r = 0; r < TR; r++) {
    const u64 *cr = tab.corrp[(uint32_t)(rots >> (8 * r)) & 0xffu] + (size_t)I * N + m;
    const HmVec<V> c0 = hm_ld<V>(cr), c1c = hm_ld<V>(cr + (size_t)(l + 1) * N);
#pragma unroll
    for (int s = 0; s < TS; s++) {
      const uint32_t z = (uint32_t)(outs >> (8 * (s * TR + r))) & 0xffu;
      if (z == 0xffu) continue; // padding, or a (source, element) combination that is not a pair of the chunk
      u64 *pr = prod + z * prod_bs + (size_t)I * N + m;
      HmVec<V> o0, o1;
#pragma unroll
      for (int v = 0; v < V; v++) {
        o0.v[v] = addmod(barrett128(a0[s][r][v], pm), c0.v[v], pm.q);
        o1.v[v] = addmod(barrett128(a1[s][r][v], pm), c1c.v[v], pm.q);
      }
      hm_st<V, (EVAH_HOIST_NT & 2) != 0>(pr, o0);
      hm_st<V, (EVAH_HOIST_NT & 2) != 0>(pr + (size_t)(l + 1) * N, o1);
    }
  }
}

// zeros[0] (low word) = number of zero digit coefficients seen, zeros[1 + e] = (source << 48 | J << 32 | k).
// grid = (N / 256, l + 1, pairs), one coefficient n of the ROTATED polynomial per thread — its term is subtracted where
// the inner product keeps it, at perm[n]; every test below is block-uniform.
static __global__ void __launch_bounds__(256)
k_hoist_fix(DevCtx cx, const u64 *zeros, HoistFixTab tab, u64 *prod, size_t prod_bs, uint32_t l) {
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
  u64 *pr = prod + z * prod_bs + (size_t)I * N + tab.perm[z][n];
  pr[0] = submod(pr[0], acc0, pm.q);
  pr[(size_t)(l + 1) * N] = submod(pr[(size_t)(l + 1) * N], acc1, pm.q);
}

// The hoisting tables are an optimisation's working set (a permuted copy of every Galois key used in a hoisted set, a
// constant per (element, level)): when the device has no room for one, the set runs unhoisted — SEAL's order, the
// path that needs no tables — instead of failing.  null = no room (the runtime's error state is cleared).
template <class T> static T *hoist_table_alloc(size_t bytes) {
  void *d = nullptr;
  if (hipMalloc(&d, bytes) != hipSuccess) {
    (void)hipGetLastError();
    return nullptr;
  }
  return static_cast<T *>(d);
}
// the inverse table: perm_inv[perm[n]] = n
static const uint32_t *perm_inv_table(evah_ctx *c, uint32_t elt) {
  auto pit = c->sh->perms_inv.find(elt);
  if (pit != c->sh->perms_inv.end()) return pit->second;
  if (c->capturing) throw std::logic_error("first hoisted use of a Galois element cannot be captured into a graph");
  const size_t N = c->N;
  std::vector<uint32_t> inv(N);
  for (uint32_t i = 0; i < N; i++) {
    uint32_t reversed = bitrev((uint32_t)N + i, c->logN + 1);
    u64 raw = (((u64)elt * reversed) >> 1) & (u64)(N - 1);
    inv[bitrev((uint32_t)raw, c->logN)] = i;
  }
  uint32_t *d = hoist_table_alloc<uint32_t>(sizeof(uint32_t) * N);
  if (!d) return nullptr;
  h2d_now(c, d, inv.data(), sizeof(uint32_t) * N);
  c->sh->perms_inv.emplace(elt, d);
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
  u64 *d = hoist_table_alloc<u64>(sizeof(u64) * N * c->k);
  if (!d) return nullptr;
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
  const uint32_t *pinv = perm_inv_table(c, elt);
  if (!sign || !pinv) return nullptr;
  u64 *d = hoist_table_alloc<u64>(sizeof(u64) * 2 * (l + 1) * c->N);
  if (!d) return nullptr;
  hipLaunchKernelGGL(k_hoist_corr, dim3(c->N / 256, l + 1, 2), dim3(256), 0, c->stream, c->dev, sign, key.d, pinv, d, l);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  if (e != hipSuccess) {
    (void)hipFree(d);
    HIPCHK(e);
  }
  c->sh->hoist_corr.emplace(std::make_pair(elt, l), d);
  return d;
}
// the key of a Galois element with its rows read through the inverse permutation (KeyDev::d_perm), built once per key
static const u64 *hoist_key(evah_ctx *c, uint32_t elt, KeyDev &key) {
  if (key.d_perm) return key.d_perm;
  if (c->capturing) throw std::logic_error("first hoisted use of a Galois key cannot be captured into a graph");
  const uint32_t *pinv = perm_inv_table(c, elt);
  if (!pinv) return nullptr;
  u64 *d = hoist_table_alloc<u64>(sizeof(u64) * keyp_block_words(key.n_digits) * (c->N / 256) * c->k);
  if (!d) return nullptr;
  const uint32_t rows = (uint32_t)(key.bytes / (sizeof(u64) * c->N));
  if (key.rows != c->k || c->N < 256) { (void)hipFree(d); return nullptr; } // (a shard's partial key: never hoisted)
  hipLaunchKernelGGL(k_key_perm, dim3(c->N / 256, rows), dim3(256), 0, c->stream, key.d, pinv, d, (uint32_t)c->N, c->k, key.n_digits);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream); // other queues may use the copy next
  if (e != hipSuccess) {
    (void)hipFree(d);
    HIPCHK(e);
  }
  key.d_perm = d;
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
  const u64 *corr;  // hoisting constant of (elt, l) and the permuted key (hoist_prepare); null when the set is not hoisted
  const u64 *keyp;
};
// the tables a hoisted pair needs (first use of an element / level: not capturable, like perm_table).
// false: the device has no room for one of them — the caller runs its set unhoisted (rot_chunk_plain), as it would with
// EVAH_HOIST=0; the tables that did fit stay for later sets.  EVAH_HOIST_TABLE_FAIL=1 (tests) refuses every NEW table.
static bool hoist_prepare(evah_ctx *c, RotPair &p, uint32_t l) {
  KeyDev &key = c->sh->galois.at(p.elt);
  if (c->tun.hoist_table_fail && (!key.d_perm || !c->sh->hoist_corr.count({p.elt, l}))) return false;
  p.corr = hoist_corr(c, p.elt, l, key);
  p.keyp = p.corr ? hoist_key(c, p.elt, key) : nullptr;
  return p.corr && p.keyp;
}
struct RotChunk {
  uint32_t first, count; // pairs [first, first + count)
  u64 *out;
};
static RotPair rot_pair(evah_ctx *c, const u64 *src, size_t src_ps, uint32_t src_idx, int32_t step, uint32_t l, const char *who) {
  if (step == 0) throw std::invalid_argument(std::string(who) + ": zero steps are copies, not key switches");
  RotPair p{src, src_ps, src_idx, 0, nullptr, nullptr, nullptr, nullptr};
  if (evah_galois_elt_from_step(c, step, &p.elt)) throw std::invalid_argument(g_err);
  auto kit = c->sh->galois.find(p.elt);
  if (kit == c->sh->galois.end()) throw std::invalid_argument("Galois key not present");
  if (kit->second.n_digits < l) throw std::runtime_error("key switching key has too few digits");
  if (kit->second.rows != c->k) throw std::logic_error("this context holds a limb shard's key rows: use the evah_shard_* entry points");
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
// perm_d == nullptr: P c0' was added to the products already (KS_FOLDADD)
// gather != nullptr: prod is indexed in each pair's SOURCE space (k_hoist_mac), read through gather->p[pair]
static void rot_mod_down(evah_ctx *c, uint32_t l, uint32_t np, u64 *prod_d, const u64 *perm_d, u64 *out_d, u64 *r_d, bool inv1,
                         const PermTab *gather = nullptr) {
  const size_t N = c->N, pps = (size_t)l * N;
  if (gather) {
    OpPlainG::Params sp{prod_d + (size_t)l * N, r_d, (size_t)(l + 1) * N, N, 1, c->k - 1, 1, {}};
    sp.perm_tab = *gather;
    OpModDownG::Params mp{r_d, N, prod_d, (size_t)(l + 1) * N, perm_d, pps, ~0u, out_d, pps, c->k - 1, l};
    mp.perm_tab = *gather;
    inverse_then_forward<OpPlainG, OpModDownG>(c, sp, 2 * np, mp, 2 * np * l, inv1);
    return;
  }
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
  const bool fold = c->tun.fold_pa && c->tun.fuse_mac;
  PtrTab adds{}; // P * (permuted c0) joins the inner product of K = 0; nothing is added to K = 1
  for (uint32_t rr = 0; rr < np && fold; rr++) adds.p[2 * rr] = perm.d + (size_t)rr * 2 * pps;
  const bool inv1 = switch_key_products(c, l, perm.d + pps, 2 * pps, keys.data(), np, prod.d, nullptr, nullptr,
                                        fuse_small_launch(c, 2 * np * l) ? r.d : nullptr, fold, fold ? &adds : nullptr);
  rot_mod_down(c, l, np, prod.d, fold ? nullptr : perm.d, out_d, r.d, inv1);
}
// the zero-coefficient record of a hoisted set: d[0] = count, d[1..] = positions (OpPlainZ, k_hoist_fix), preceded in the
// same allocation by one word per chunk for the persistent fallback (its ticket and finished-chunk counters); everything
// that must start at zero is adjacent
// (r5: cleared by the first workgroup of the set's first kernel — the contiguous pass of hoist_digits' inverse transform,
// ntt.hip.h first_pass_clear — instead of a memset launch per set)
struct ZeroFlag {
  Scratch s;
  u64 *d;
  uint32_t clear_words;
  ZeroFlag(evah_ctx *c, size_t chunks) : s(c, chunks + 1 + HOIST_ZERO_CAP), d(s.d + chunks), clear_words((uint32_t)chunks + 1) {}
  u64 *bar(size_t chunk) const { return s.d + chunk; }
};
// launches issued while one of these lives return at once unless more than HOIST_ZERO_CAP zero digit coefficients were recorded
struct GuardScope {
  evah_ctx *c;
  GuardScope(evah_ctx *c_, const uint32_t *g) : c(c_) { c->dev.guard = g; c->dev.guard_min = HOIST_ZERO_CAP; }
  ~GuardScope() { c->dev.guard = nullptr; }
};
// the exact fallback of one chunk as one persistent launch (rot_fallback.hip.h); call inside a GuardScope.
// rot_out: the chunk's outputs [np][2][l N] (plain rotation sets) or null with the window tables of the chunk
static uint32_t cu_count(int device) {
  static std::mutex mu;
  static std::map<int, uint32_t> known;
  std::lock_guard<std::mutex> lk(mu);
  auto it = known.find(device);
  if (it != known.end()) return it->second;
  int n = 0;
  HIPCHK(hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, device));
  return known[device] = (uint32_t)std::max(n, 1);
}
static void rot_fallback_launch(evah_ctx *c, uint32_t l, const RotPair *pr, uint32_t np, u64 *rot_out, const WinSumTab *wt, uint32_t n_win, int F,
                                size_t out_ps, u64 *bar_word) {
  if (!c->dev.guard) throw std::logic_error("the rotation fallback runs under its guard only");
  const size_t N = c->N, lN = (size_t)l * N;
  FbPairs fp{};
  for (uint32_t r = 0; r < np; r++) {
    fp.perm[r] = pr[r].perm;
    fp.src[r] = pr[r].src;
    fp.key[r] = pr[r].key->d;
    fp.src_ps[r] = (uint32_t)(pr[r].src_ps / N);
  }
  Scratch rc1(c, np * lN), t(c, np * lN), dig(c, np * lN), prod(c, (size_t)np * 2 * (l + 1) * N), r(c, (size_t)np * 2 * N), u(c, (size_t)np * 2 * lN);
  FbBufs b{rc1.d, t.d, dig.d, prod.d, r.d, u.d, reinterpret_cast<unsigned *>(bar_word)};
  // One workgroup per CU: enough to make the (rare) active path a matter of milliseconds, few enough that the launch
  // costs what an empty kernel costs when the hoisted results stand.  Correctness does not depend on it — phases are
  // ordered by tickets (rot_fallback.hip.h), not by a barrier over resident workgroups; EVAH_FB_GRID overrides (tests).
  uint32_t grid = std::min<uint32_t>(cu_count(c->device), 256);
  if (c->tun.fb_grid) grid = c->tun.fb_grid;
  WinSumTab none{};
  ProfScope ps(c, KC_EW);
  if (F == 0) hipLaunchKernelGGL((k_rot_fallback<0>), dim3(grid), dim3(256), 0, c->stream, c->dev, fp, np, l, b, rot_out, none, 0u, (size_t)0);
  else if (F == 1) hipLaunchKernelGGL((k_rot_fallback<1>), dim3(grid), dim3(256), 0, c->stream, c->dev, fp, np, l, b, (u64 *)nullptr, *wt, n_win, out_ps);
  else hipLaunchKernelGGL((k_rot_fallback<2>), dim3(grid), dim3(256), 0, c->stream, c->dev, fp, np, l, b, (u64 *)nullptr, *wt, n_win, out_ps);
  HIPCHK(hipGetLastError());
}
// kernel-argument tables of one chunk of hoisted pairs: the tile shape for the chunk and its tiles
struct HoistTiles {
  int TS = 1, TR = 1;
  uint32_t n_tiles = 0;
  std::vector<std::array<uint32_t, 6>> tiles; // HT_TILES at a time go into HoistMacTab::tile
};
static HoistTiles hoist_tables(const RotPair *pr, uint32_t np, size_t N, HoistMacTab &mt, HoistFixTab &ft) {
  std::vector<uint32_t> srcs, elts; // the chunk's distinct sources (by digit index) and Galois elements
  std::vector<uint32_t> ps(np), pe(np);
  for (uint32_t r = 0; r < np; r++) {
    ft.perm[r] = pr[r].perm;
    ft.key[r] = pr[r].key->d;
    ft.elt[r] = pr[r].elt;
    ft.src[r] = (uint8_t)pr[r].src_idx;
    size_t si = std::find(srcs.begin(), srcs.end(), pr[r].src_idx) - srcs.begin();
    if (si == srcs.size()) {
      srcs.push_back(pr[r].src_idx);
      mt.c1[si] = pr[r].src + pr[r].src_ps;
      mt.c1_ps[si] = (uint32_t)(pr[r].src_ps / N);
      mt.dg[si] = pr[r].src_idx;
    }
    size_t ei = std::find(elts.begin(), elts.end(), pr[r].elt) - elts.begin();
    if (ei == elts.size()) {
      elts.push_back(pr[r].elt);
      mt.keyp[ei] = pr[r].keyp;
      mt.corrp[ei] = pr[r].corr;
      mt.keyp_nd[ei] = pr[r].key->n_digits;
    }
    ps[r] = (uint32_t)si;
    pe[r] = (uint32_t)ei;
  }
  // shape: as many sources as the chunk has (up to 4), then as many elements as 8 accumulator pairs allow
  HoistTiles ht;
  const size_t S = srcs.size(), R = elts.size();
  // (8 x 1 for the instances of a batched handle reads every key once but every digit row eight times: 14.3 k against
  // 14.5 k DAGs/s with 4 x 2 on config 4; 1 x 8 needs 159 VGPRs; a 64-VGPR ceiling for 8 waves per SIMD spills: 9.8 k)
  ht.TS = S >= 4 ? 4 : (int)S;
  if (const char *e = std::getenv("EVAH_HOIST_TS")) ht.TS = std::max(1, std::min(ht.TS, std::atoi(e))); // (experiments)
  int tr_max = ht.TS <= 2 ? 4 : 2; // (1 x 2 instead of 1 x 4: the same; 3 x 1 instead of 3 x 2: Harris +2.5 %)
  if (const char *e = std::getenv("EVAH_HOIST_TR")) tr_max = std::max(1, std::min(tr_max, std::atoi(e)));
  ht.TR = 1;
  while (ht.TR < tr_max && (size_t)ht.TR < R) ht.TR *= 2;
  // tiles: TR elements x TS sources, taken greedily in pair order (a rectangular set — every source with every element —
  // fills its tiles completely); short tiles repeat their last entry, the repeats' outputs are 0xff (not stored)
  std::vector<char> taken(np, 0);
  for (uint32_t r0 = 0; r0 < np; r0++) {
    if (taken[r0]) continue;
    std::vector<uint32_t> te, ts;
    for (uint32_t q = r0; q < np && te.size() < (size_t)ht.TR; q++)
      if (!taken[q] && ps[q] == ps[r0] && std::find(te.begin(), te.end(), pe[q]) == te.end()) te.push_back(pe[q]);
    for (uint32_t q = r0; q < np && ts.size() < (size_t)ht.TS; q++)
      if (!taken[q] && std::find(te.begin(), te.end(), pe[q]) != te.end() && std::find(ts.begin(), ts.end(), ps[q]) == ts.end()) ts.push_back(ps[q]);
    uint8_t b[24];
    for (int i = 0; i < 8; i++) {
      b[i] = (uint8_t)ts[std::min<size_t>(i, ts.size() - 1)];
      b[8 + i] = (uint8_t)te[std::min<size_t>(i, te.size() - 1)];
      b[16 + i] = 0xff;
    }
    for (size_t si = 0; si < ts.size(); si++)
      for (size_t ei = 0; ei < te.size(); ei++)
        for (uint32_t q = r0; q < np; q++)
          if (!taken[q] && ps[q] == ts[si] && pe[q] == te[ei]) {
            b[16 + si * ht.TR + ei] = (uint8_t)q;
            taken[q] = 1;
            break;
          }
    std::array<uint32_t, 6> w{};
    for (int i = 0; i < 6; i++) w[i] = b[4 * i] | (uint32_t)b[4 * i + 1] << 8 | (uint32_t)b[4 * i + 2] << 16 | (uint32_t)b[4 * i + 3] << 24;
    ht.tiles.push_back(w);
  }
  ht.n_tiles = (uint32_t)ht.tiles.size();
  return ht;
}
// the hoisted inner products of a chunk (k_hoist_mac), HT_TILES tiles per launch
static void hoist_mac_launch(evah_ctx *c, HoistMacTab &mt, const HoistTiles &ht, const u64 *dg, size_t dg_bs, u64 *prod, size_t prod_bs, uint32_t l,
                             bool fold) {
  ProfScope ps(c, KC_KSMAC);
  for (uint32_t t0 = 0; t0 < ht.n_tiles; t0 += HT_TILES) {
    const uint32_t n = std::min<uint32_t>(HT_TILES, ht.n_tiles - t0);
    for (uint32_t t = 0; t < n; t++)
      for (int i = 0; i < 6; i++) mt.tile[t][i] = ht.tiles[t0 + t][i];
    // V = 2 (16-byte accesses) for the shapes with at most four accumulator pairs; MAP needs N / (256 V) >= 8 tiles
    const int V = 1; // (r6: the 2-coefficient form was measured — config 5 1.67 -> 1.71 ms — and went with the blocked key copy)
    const uint32_t xt = c->N / (256 * V);
    const bool map = c->tun.hoist_map && xt >= 8;
#define HMV(S_, R_, V_)                                                                                                                \
  if (ht.TS == S_ && ht.TR == R_ && V == V_) {                                                                                         \
    if (map)                                                                                                                           \
      hipLaunchKernelGGL((k_hoist_mac<S_, R_, V_, true>), dim3(xt * (l + 1) * n), dim3(256), 0, c->stream, c->dev, dg, dg_bs, mt, n, prod, prod_bs, l, fold); \
    else                                                                                                                               \
      hipLaunchKernelGGL((k_hoist_mac<S_, R_, V_, false>), dim3(xt, l + 1, n), dim3(256), 0, c->stream, c->dev, dg, dg_bs, mt, n, prod, prod_bs, l, fold); \
    HIPCHK(hipGetLastError());                                                                                                         \
    continue;                                                                                                                          \
  }
#define HM(S_, R_) HMV(S_, R_, 1)
    HM(1, 1) HM(1, 2) HM(1, 4) HM(2, 1) HM(2, 2) HM(2, 4) HM(3, 1) HM(3, 2) HM(4, 1) HM(4, 2)
#undef HMV
#undef HM
    throw std::logic_error("hoisted inner product: no kernel for this tile shape");
  }
}
// digits of the unrotated c1 of every source, once: coefficient form (zeros recorded in flag_d), then the full
// transforms under every output prime into dg_d[source][(l+1) l N]; t_d: n_src * l * N words of scratch
static void hoist_digits(evah_ctx *c, uint32_t l, const std::vector<const u64 *> &srcs, const std::vector<size_t> &src_ps, const ZeroFlag &flag,
                         u64 *t_d, u64 *dg_d) {
  const size_t N = c->N, dg_bs = (size_t)(l + 1) * l * N;
  const uint32_t n_src = (uint32_t)srcs.size();
  PtrTab c1{};
  for (uint32_t i = 0; i < n_src; i++) c1.p[i] = srcs[i] + src_ps[i];
  OpPlainZ::Params ip{nullptr, t_d, 0, (size_t)l * N, l, 0, 0, c1};
  ip.zero_list = flag.d;
  ip.clear_base = flag.s.d; // the fallback's ticket words and the zero count: cleared by this transform's first pass
  ip.clear_words = flag.clear_words;
  ntt_inverse<OpPlainZ>(c, ip, n_src * l);
  OpKsDigit::Params dp{t_d, dg_d, l, (size_t)l * N, dg_bs, 0, l + 1};
  ntt_forward<OpKsDigit>(c, dp, n_src * (l + 1) * l);
}

} // namespace evah
