// ntt_window_sum.hip.h — moddown_sum_kernel: the mod-down of a set of rotations fused with the plaintext-weighted sums that consume them
// (part of ntt.hip.h until r5; included by it, in the order the definitions depend on each other)
#pragma once
#include "ntt.hip.h"

namespace evah {

// ---- Window sums: the mod-down of a set of rotations fused with the plaintext-weighted sums that consume them
// (the convolution pattern: out_f = sum_t w_ft (*) rotate(x, step_t); examples/image_processing.py convolutionXY).
// Unfused, every rotated ciphertext is written by the mod-down's second pass and read back once by the weighted sum;
// here the second pass of the mod-down walks the rotations of ONE window for its (limb, tile, polynomial), multiplies
// each finished value by the window's weights and keeps the 128-bit partial sums of up to two sums in registers: the
// rotated ciphertexts never exist in memory.  Same canonical residues as rotate -> multiply_plain -> add one by one.
//   pair t of the chunk: polynomial pp = 2 t + K; mid[pp][i] = first (strided) pass of NTT_i(u) (OpModDown, dst = mid),
//   prod[pp][i] = key inner product in the SOURCE's index space (k_hoist_mac), P * c0 already folded in: the pair's
//   Galois permutation is applied when it is read, nothing is added here.
//   value_t[n] = (prod[pp][i][perm_t[n]] - NTT_i(u)[n]) * P^-1 mod q_i;   out_f[K][i] = sum_t w_f[t][i] * value_t  (+ the unrotated term)
#ifndef EVAH_KS_BATCH_MAX_DEFINED
#define EVAH_KS_BATCH_MAX_DEFINED
constexpr int KS_BATCH_MAX = 64; // as internal.hip.h (this header is also used alone)
#endif
// index tables of a launch's pairs (two polynomials each): data kept in a pair's source index space is read through them
struct PermTab {
  const uint32_t *p[KS_BATCH_MAX];
};
constexpr int WIN_MAX = 16; // windows per launch (the tables below travel as kernel arguments)
struct WinSumTab {
  uint8_t first[WIN_MAX], count[WIN_MAX];          // window w = pairs [first, first + count) of the chunk
  const u64 *w0[KS_BATCH_MAX], *w1[KS_BATCH_MAX];  // per pair: weights (NTT form, [l][N]) in the window's sums; null = 1
  const u64 *id_src[WIN_MAX];                      // per window: c0 of its unrotated term (null: none), c1 = + id_ps * N
  const u64 *id_w0[WIN_MAX], *id_w1[WIN_MAX];
  uint32_t id_ps[WIN_MAX];
  u64 *out0[WIN_MAX], *out1[WIN_MAX];              // per window: c0 of the sums' outputs (c1 = + out_ps)
};
// grid = (N / 256, l, 2 * windows); one wave, 4 coefficients per thread (the shape of ntt_loop_kernel, whose
// twiddle staging and tile pipeline this repeats)
template <int P, int F>
__global__ void __launch_bounds__(64)
moddown_sum_kernel(DevCtx cx, WinSumTab ws, PermTab perms, const u64 *mid, size_t mid_ps, const u64 *prod, size_t prod_ps, size_t out_ps, int logC) {
  extern __shared__ __attribute__((aligned(16))) u64 lds[];
  if (cx.skipped()) return;
  constexpr int LR = 2, NTT_R = 1 << LR, NPAIR = NTT_R / 2, T = 64;
  constexpr int S = 1 << P, TPS = S / NTT_R, SP = lds_sub_stride<P>();
  const uint32_t i = blockIdx.y, w = blockIdx.z >> 1, K = blockIdx.z & 1u;
  const uint32_t first = ws.first[w], cnt = ws.count[w];
  const int C = 1 << logC;
  const uint32_t pre = cx.logN - P, sub0 = blockIdx.x << logC, gbase = sub0 << P;
  const int sub = threadIdx.x / TPS, tid = threadIdx.x % TPS;
  const uint32_t prime = cx.prime_of(i), a = cx.k - 1;
  const DevPrime pm = cx.primes[prime];
  const ulonglong2 inv = cx.invq[(size_t)a * cx.k + prime];
  const ulonglong2 *tw = cx.tw_fwd + (size_t)prime * cx.N;
  ulonglong2 *twl = reinterpret_cast<ulonglong2 *>(lds + ((C * SP + 1) & ~1));
  for (int idx = threadIdx.x; idx < (C << P); idx += T) {
    const int sb = idx >> P, n = idx & (S - 1);
    if (n) {
      const int d = 31 - __clz(n);
      twl[idx] = tw[((size_t)((1u << pre) + sub0 + sb) << d) + (n - (1 << d))];
    }
  }
  const size_t row = (size_t)i * cx.N + gbase + 2 * threadIdx.x; // + it * 2 T
  auto load_tile = [&](uint32_t t, ulonglong2 *d) {
    const u64 *src = mid + (size_t)(2 * t + K) * mid_ps + row;
#pragma unroll
    for (int it = 0; it < NPAIR; it++) d[it] = *reinterpret_cast<const ulonglong2 *>(src + it * 2 * T);
  };
  u128_t acc[F][NTT_R];
#pragma unroll
  for (int f = 0; f < F; f++)
#pragma unroll
    for (int e = 0; e < NTT_R; e++) acc[f][e] = {0, 0};
  auto mac = [&](int f, int it, const ulonglong2 &v, const u64 *wt) { // wt: block-uniform; null stands for the weight 1
    ulonglong2 x;
    x.x = x.y = 1;
    if (wt) x = *reinterpret_cast<const ulonglong2 *>(wt + row + it * 2 * T);
    acc128(acc[f][2 * it], v.x, x.x);
    acc128(acc[f][2 * it + 1], v.y, x.y);
  };
  ulonglong2 dreg[NPAIR];
  uint2 pnext[NPAIR]; // the NEXT pair's gather indices (prod is indexed in the source's space): a pair ahead, so that the
                      // gathers of a pair do not wait for an index load first
  auto load_perm = [&](uint32_t t) {
    const uint32_t *pi = perms.p[t] + gbase + 2 * threadIdx.x;
#pragma unroll
    for (int it = 0; it < NPAIR; it++) pnext[it] = *reinterpret_cast<const uint2 *>(pi + it * 2 * T);
  };
  if (cnt) {
    load_tile(first, dreg);
    load_perm(first);
  }
  for (uint32_t t = first; t < first + cnt; t++) {
    uint2 at[NPAIR];
#pragma unroll
    for (int it = 0; it < NPAIR; it++) at[it] = pnext[it];
    __syncthreads(); // the previous pair's LDS reads are done
#pragma unroll
    for (int it = 0; it < NPAIR; it++) {
      const int idx = 2 * (threadIdx.x + it * T);
      const int sb = idx >> P, e = idx & (S - 1);
      lds[sb * SP + lds_pad<P>(e)] = dreg[it].x;
      lds[sb * SP + lds_pad<P>(e + 1)] = dreg[it].y;
    }
    // the epilogue's operands, requested before the transform
    const u64 *pr = prod + (size_t)(2 * t + K) * prod_ps + (size_t)i * cx.N;
    const u64 *wt0 = ws.w0[t], *wt1 = F > 1 ? ws.w1[t] : nullptr;
    ulonglong2 cp[NPAIR], x0[NPAIR], x1[NPAIR];
#pragma unroll
    for (int it = 0; it < NPAIR; it++) {
      cp[it].x = pr[at[it].x];
      cp[it].y = pr[at[it].y];
      x0[it].x = x0[it].y = x1[it].x = x1[it].y = 1;
      if (wt0) x0[it] = *reinterpret_cast<const ulonglong2 *>(wt0 + row + it * 2 * T);
      if (F > 1 && wt1) x1[it] = *reinterpret_cast<const ulonglong2 *>(wt1 + row + it * 2 * T);
    }
    if (t + 1 < first + cnt) {
      load_tile(t + 1, dreg);
      load_perm(t + 1);
    }
    __syncthreads();
    forward_rounds<P, LR, true, true>(lds + sub * SP, tid, 0, 0, twl + (sub << P), pm);
    __syncthreads();
#pragma unroll
    for (int it = 0; it < NPAIR; it++) {
      const int idx = 2 * (threadIdx.x + it * T);
      const int sb = idx >> P, e = idx & (S - 1);
      u64 ux = lds[sb * SP + lds_pad<P>(e)], uy = lds[sb * SP + lds_pad<P>(e + 1)];
      ux += (ux >= pm.q8 ? pm.nq8 : 0); // [0,16q) -> [0,8q)
      uy += (uy >= pm.q8 ? pm.nq8 : 0);
      const u64 vx = mul_shoup(cp[it].x + pm.q8 - ux, inv.x, inv.y, pm.q), vy = mul_shoup(cp[it].y + pm.q8 - uy, inv.x, inv.y, pm.q);
      acc128(acc[0][2 * it], vx, x0[it].x);
      acc128(acc[0][2 * it + 1], vy, x0[it].y);
      if constexpr (F > 1) {
        acc128(acc[1][2 * it], vx, x1[it].x);
        acc128(acc[1][2 * it + 1], vy, x1[it].y);
      }
    }
  }
  if (ws.id_src[w]) { // the window's unrotated term: the source ciphertext itself
    const u64 *src = ws.id_src[w] + (size_t)K * ws.id_ps[w] * cx.N + row;
#pragma unroll
    for (int it = 0; it < NPAIR; it++) {
      const ulonglong2 v = *reinterpret_cast<const ulonglong2 *>(src + it * 2 * T);
      mac(0, it, v, ws.id_w0[w]);
      if constexpr (F > 1) mac(1, it, v, ws.id_w1[w]);
    }
  }
#pragma unroll
  for (int f = 0; f < F; f++) {
    u64 *o = (f ? ws.out1[w] : ws.out0[w]) + (size_t)K * out_ps + row;
#pragma unroll
    for (int it = 0; it < NPAIR; it++) {
      ulonglong2 r;
      r.x = barrett128(acc[f][2 * it], pm);
      r.y = barrett128(acc[f][2 * it + 1], pm);
      *reinterpret_cast<ulonglong2 *>(o + it * 2 * T) = r;
    }
  }
}

} // namespace evah
