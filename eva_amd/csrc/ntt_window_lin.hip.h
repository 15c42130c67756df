// ntt_window_lin.hip.h — the mod-down of a convolution window whose weights are UNIFORM plaintexts (r6): ONE forward
// transform per (sum, polynomial, limb) instead of one per rotation.
//
// A window sum is  out_f[K][i] = sum_t w_ft (*) value_t[K][i]  with  value_t = (prod_t[perm_t] - NTT_i(u_t)) * P^-1 mod q_i,
// u_t = (r_t mod q_i) - floor(P/2) mod q_i and r_t = INTT_P(special row of prod_t) + floor(P/2)  (one mod-down
// per rotation: seal_executor.h:181/:188 rotate_vector, :168 multiply_plain, :124 add).  EVA's filter taps are scalar
// constants (/root/reference/examples/image_processing.py:22-34 `rotated * filter[i][j]`; Program::makeUniformConstant,
// program.h:58-60), whose encoding is the SAME residue W_ft,i in every NTT slot of limb i (evah_pt_uniform).  Multiplying
// by such a plaintext is multiplying by a scalar of Z_q, and the NTT is linear over Z_q, so
//     sum_t W_t * (prod_t[perm_t[n]] - NTT_i(u_t)[n]) * P^-1  =  ( sum_t W_t * prod_t[perm_t[n]]  -  NTT_i( sum_t W_t * u_t )[n] ) * P^-1
// as residues: every value SEAL stores is canonical, so the canonical residue of the right-hand side IS the word SEAL
// ends up with.  The inverse transforms stay one per rotation (r_t is reduced from [0, P) to [0, q_i) as an integer: not
// linear), the forward transforms drop from 2 l per rotation to 2 l per sum: a 3x3 window with one sum (Harris' box
// filters, the config-5 convolution) runs 1/8 of them, one with two sums (Sobel's Ix / Iy) 1/4.
//   pass 1 (OpWinLin, strided, through ntt_pass_kernel):  x[n] = sum_t W_t (r_t[n] + q_i - half_i) mod q_i, first pass into mid
//   pass 2 (winlin_pass2_kernel, contiguous): U = second pass of mid;  out = (sum_t W_t prod_t[perm_t[n]] - U) P^-1 (+ unrotated term)
// Weights that are not uniform (a plaintext INPUT as a filter) keep moddown_sum_kernel.
#pragma once
#include "launch.hip.h"

namespace evah {

struct OpWinLin {
  struct Params {
    const u64 *r;   // [2 pair + K][N]: INTT_P(special row) + floor(P/2), coefficient form, canonical mod P
    u64 *mid;       // [(unit F + f) 2 + K][l][N]
    uint32_t a;     // the special prime
    uint32_t l, F;
    WinSumTab ws;   // first / count per unit, w0 / w1 per pair (null = the weight 1)
  };
  struct Job {
    uint32_t prime, first, cnt, f, K, i;
    const u64 *r;
    u64 *dst;
    u64 off; // q_i - (floor(P/2) mod q_i)
    bool lazy;
    const u64 *const *w; // the sum's weight pointers, indexed by pair
  };
  static dim3 grid(const Params &p, uint32_t jobs) { return dim3(1, p.l, jobs / p.l); }
  static constexpr int loop_axis = 2;
  static __device__ __forceinline__ bool setup(const DevCtx &cx, const Params &p, uint32_t, uint32_t i, uint32_t z, Job &j) {
    const uint32_t K = z & 1u, uf = z >> 1, f = p.F > 1 ? (uf & 1u) : 0u, u = p.F > 1 ? (uf >> 1) : uf;
    j.prime = cx.prime_of(i);
    j.first = p.ws.first[u];
    j.cnt = p.ws.count[u];
    j.f = f;
    j.K = K;
    j.i = i;
    j.r = p.r;
    j.dst = p.mid + ((size_t)z * p.l + i) * cx.N;
    j.off = cx.primes[j.prime].q - cx.halfmod[p.a * cx.k + j.prime];
    j.lazy = false; // the sum leaves barrett128 canonical
    j.w = f ? p.ws.w1 : p.ws.w0;
    return true;
  }
  template <bool LZ>
  static __device__ __forceinline__ u64 load(const DevCtx &cx, const Job &j, const DevPrime &pm, uint32_t n) {
    // r_t < P < 2^60 and off <= q_i < 2^60: every term is below 2^121, a window has at most 64 of them
    u128_t acc = {0, 0};
    for (uint32_t t = j.first; t < j.first + j.cnt; t++) {
      const u64 *wt = j.w[t];                                   // wave-uniform
      const u64 W = wt ? wt[(size_t)j.i * cx.N] : 1;            // a uniform plaintext: any word of the limb
      acc128(acc, j.r[(size_t)(2 * t + j.K) * cx.N + n] + j.off, W);
    }
    return barrett128(acc, pm);
  }
  // the thread's R elements n0 + e * nstep, term-outer: two terms' loads (2 R words) in flight per step
  static constexpr bool fills_all = true;
  template <int R>
  static __device__ __forceinline__ void fill_all(const DevCtx &cx, const Job &j, const DevPrime &pm, uint32_t n0, uint32_t nstep, u64 *out) {
    u128_t acc[R];
#pragma unroll
    for (int e = 0; e < R; e++) acc[e] = {0, 0};
    const u64 *rk = j.r + (size_t)j.K * cx.N + n0;
    const size_t wofs = (size_t)j.i * cx.N;
    uint32_t t = j.first;
    const uint32_t end = j.first + j.cnt;
    for (; t + 1 < end; t += 2) {
      const u64 *wa = j.w[t], *wb = j.w[t + 1];                 // wave-uniform
      const u64 Wa = wa ? wa[wofs] : 1, Wb = wb ? wb[wofs] : 1; // a uniform plaintext: any word of the limb
      const u64 *ra = rk + (size_t)(2 * t) * cx.N, *rb = ra + 2 * (size_t)cx.N;
      u64 va[R], vb[R];
#pragma unroll
      for (int e = 0; e < R; e++) {
        va[e] = ra[e * nstep];
        vb[e] = rb[e * nstep];
      }
#pragma unroll
      for (int e = 0; e < R; e++) {
        acc128(acc[e], va[e] + j.off, Wa);
        acc128(acc[e], vb[e] + j.off, Wb);
      }
    }
    if (t < end) {
      const u64 *wa = j.w[t];
      const u64 Wa = wa ? wa[wofs] : 1;
      const u64 *ra = rk + (size_t)(2 * t) * cx.N;
#pragma unroll
      for (int e = 0; e < R; e++) acc128(acc[e], ra[e * nstep] + j.off, Wa);
    }
#pragma unroll
    for (int e = 0; e < R; e++) out[e] = barrett128(acc[e], pm);
  }
  static __device__ __forceinline__ void store(const DevCtx &, const Job &, const DevPrime &, uint32_t, u64) {}
  static __device__ __forceinline__ void store_fwd(const DevCtx &, const Job &, const DevPrime &, uint32_t, u64) {}
};
template <> struct OpClass<OpWinLin> { static constexpr int fwd_a = KC_MODDOWN_A, fwd_b = KC_MODDOWN_B; };

// grid = (N / 256, l, units); one wave, 4 coefficients per thread; the workgroup stages the twiddle heaps of its tile once
// and walks the 2 F (polynomial, sum) tiles of its unit.
template <int P, int F>
__global__ void __launch_bounds__(64)
winlin_pass2_kernel(DevCtx cx, WinSumTab ws, PermTab perms, const u64 *mid, size_t mid_ps, const u64 *prod, size_t prod_ps, size_t out_ps, int logC) {
  extern __shared__ __attribute__((aligned(16))) u64 lds[];
  if (cx.skipped()) return;
  constexpr int LR = 2, NTT_R = 1 << LR, NPAIR = NTT_R / 2, T = 64;
  constexpr int S = 1 << P, TPS = S / NTT_R, SP = lds_sub_stride<P>();
  const uint32_t i = blockIdx.y, u = blockIdx.z;
  const uint32_t first = ws.first[u], cnt = ws.count[u];
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
  // sum_t W_ft * prod_t[K][i][perm_t[n]]: prod is kept in each pair's source index space.  G terms per step: their G NTT_R
  // gathers are in flight together, the next step's indices behind them.  (The Galois permutation maps an aligned
  // 256-coefficient tile ONTO an aligned tile — the low bits of 2 br(n) + 1 pick the low bits of its product with the
  // element — so a wave's gather stays inside one 2 KiB block of the row.)
  constexpr int G = F > 1 ? 2 : 4;
  const uint32_t end = first + cnt;
  for (uint32_t K = 0; K < 2; K++) {
    // the first sum's tile is requested before the gathers: it arrives while they are accumulated
    ulonglong2 dreg[NPAIR];
    {
      const u64 *src = mid + (size_t)((u * F + 0) * 2 + K) * mid_ps + row;
#pragma unroll
      for (int it = 0; it < NPAIR; it++) dreg[it] = *reinterpret_cast<const ulonglong2 *>(src + it * 2 * T);
    }
    u128_t acc[F][NTT_R];
#pragma unroll
    for (int f = 0; f < F; f++)
#pragma unroll
      for (int e = 0; e < NTT_R; e++) acc[f][e] = {0, 0};
    uint2 pnext[G][NPAIR];
    auto load_perms = [&](uint32_t t0) {
#pragma unroll
      for (int g = 0; g < G; g++) {
        const uint32_t t = t0 + g < end ? t0 + g : end - 1; // (a short last group repeats its last term: loaded, weight 0)
        const uint32_t *pi = perms.p[t] + gbase + 2 * threadIdx.x;
#pragma unroll
        for (int it = 0; it < NPAIR; it++) pnext[g][it] = *reinterpret_cast<const uint2 *>(pi + it * 2 * T);
      }
    };
    if (cnt) load_perms(first);
    for (uint32_t t0 = first; t0 < end; t0 += G) {
      u64 pv[G][NTT_R];
      u64 W[G][F];
#pragma unroll
      for (int g = 0; g < G; g++) {
        const bool live = t0 + g < end;
        const uint32_t t = live ? t0 + g : end - 1;
        const u64 *pr = prod + (size_t)(2 * t + K) * prod_ps + (size_t)i * cx.N;
#pragma unroll
        for (int it = 0; it < NPAIR; it++) {
          pv[g][2 * it] = pr[pnext[g][it].x];
          pv[g][2 * it + 1] = pr[pnext[g][it].y];
        }
        const u64 *wt0 = ws.w0[t], *wt1 = F > 1 ? ws.w1[t] : nullptr;
        W[g][0] = !live ? 0 : wt0 ? wt0[(size_t)i * cx.N] : 1; // (a repeated term counts with the weight 0)
        if constexpr (F > 1) W[g][1] = !live ? 0 : wt1 ? wt1[(size_t)i * cx.N] : 1;
      }
      if (t0 + G < end) load_perms(t0 + G);
#pragma unroll
      for (int g = 0; g < G; g++)
#pragma unroll
        for (int e = 0; e < NTT_R; e++) {
          acc128(acc[0][e], pv[g][e], W[g][0]);
          if constexpr (F > 1) acc128(acc[1][e], pv[g][e], W[g][1]);
        }
    }
#pragma unroll
    for (int f = 0; f < F; f++) {
      __syncthreads(); // the previous tile's LDS reads are done
#pragma unroll
      for (int it = 0; it < NPAIR; it++) {
        const int idx = 2 * (threadIdx.x + it * T);
        const int sb = idx >> P, e = idx & (S - 1);
        lds[sb * SP + lds_pad<P>(e)] = dreg[it].x;
        lds[sb * SP + lds_pad<P>(e + 1)] = dreg[it].y;
      }
      if (f + 1 < F) { // the next sum's tile, in flight during this transform
        const u64 *src = mid + (size_t)((u * F + f + 1) * 2 + K) * mid_ps + row;
#pragma unroll
        for (int it = 0; it < NPAIR; it++) dreg[it] = *reinterpret_cast<const ulonglong2 *>(src + it * 2 * T);
      }
      // the unrotated term of the window (the source ciphertext itself, any plaintext as its weight)
      ulonglong2 idv[NPAIR], idw[NPAIR];
      const u64 *id_wt = f ? ws.id_w1[u] : ws.id_w0[u];
      if (ws.id_src[u]) {
        const u64 *src = ws.id_src[u] + (size_t)K * ws.id_ps[u] * cx.N + row;
#pragma unroll
        for (int it = 0; it < NPAIR; it++) {
          idv[it] = *reinterpret_cast<const ulonglong2 *>(src + it * 2 * T);
          idw[it].x = idw[it].y = 1;
          if (id_wt) idw[it] = *reinterpret_cast<const ulonglong2 *>(id_wt + row + it * 2 * T);
        }
      }
      __syncthreads();
      forward_rounds<P, LR, true, true>(lds + sub * SP, tid, 0, 0, twl + (sub << P), pm);
      __syncthreads();
      u64 *o = (f ? ws.out1[u] : ws.out0[u]) + (size_t)K * out_ps + row;
#pragma unroll
      for (int it = 0; it < NPAIR; it++) {
        const int idx = 2 * (threadIdx.x + it * T);
        const int sb = idx >> P, e = idx & (S - 1);
        u64 ux = lds[sb * SP + lds_pad<P>(e)], uy = lds[sb * SP + lds_pad<P>(e + 1)];
        ux += (ux >= pm.q8 ? pm.nq8 : 0); // [0,16q) -> [0,8q)
        uy += (uy >= pm.q8 ? pm.nq8 : 0);
        ulonglong2 v;
        v.x = mul_shoup(barrett128(acc[f][2 * it], pm) + pm.q8 - ux, inv.x, inv.y, pm.q);
        v.y = mul_shoup(barrett128(acc[f][2 * it + 1], pm) + pm.q8 - uy, inv.x, inv.y, pm.q);
        if (ws.id_src[u]) { // block-uniform
          v.x = addmod(v.x, mulmod(idv[it].x, idw[it].x, pm), pm.q);
          v.y = addmod(v.y, mulmod(idv[it].y, idw[it].y, pm), pm.q);
        }
        *reinterpret_cast<ulonglong2 *>(o + it * 2 * T) = v;
      }
    }
  }
}

template <int P>
static void launch_winlin_pass2(evah_ctx *c, uint32_t l, uint32_t n_units, int F, const WinSumTab &wt, const PermTab &perms, const u64 *mid, size_t mid_ps,
                                const u64 *prod, size_t prod_ps, size_t out_ps) {
  ProfScope ps(c, KC_MODDOWN_B);
  const int logC = 8 - P;
  const size_t lds = ((((size_t)1 << logC) * lds_sub_stride<P>() + 1) & ~(size_t)1) * sizeof(u64) + ((size_t)1 << (logC + P)) * sizeof(ulonglong2);
  const dim3 grid(c->N / 256, l, n_units);
  if (F == 1) hipLaunchKernelGGL((winlin_pass2_kernel<P, 1>), grid, dim3(64), lds, c->stream, c->dev, wt, perms, mid, mid_ps, prod, prod_ps, out_ps, logC);
  else hipLaunchKernelGGL((winlin_pass2_kernel<P, 2>), grid, dim3(64), lds, c->stream, c->dev, wt, perms, mid, mid_ps, prod, prod_ps, out_ps, logC);
  HIPCHK(hipGetLastError());
}

} // namespace evah
