// rot_fallback.hip.h — the exact fallback of hoisted rotation sets as ONE launch (included by rotation_sets.hip.h).
//
// A hoisted set (k_hoist_mac, rotation_sets.hip.h) corrects up to HOIST_ZERO_CAP zero digit coefficients one by one; a source with
// more of them (a transparent ciphertext, a zero limb) needs SEAL's own order — rotate, then decompose
// (Evaluator::rotate_internal -> switch_key_inplace; /root/reference/eva/seal/seal_executor.h:181/188).  That path is
// taken once in a blue moon, but its launches used to be issued every time, each returning at once under the device-side
// guard: ~8 empty launches per chunk, 4-8 % of the DAG legs' time.  Here the whole unhoisted computation of a chunk is
// one persistent kernel: every workgroup reads the zero counter first and leaves if the hoisted results stand; otherwise
// the workgroups walk the phases below.  Nothing here is tuned — radix-2 stages on global memory — it only has to be
// exact and bounded (milliseconds).
//   A  rc1[p][J] = c1_p[J] o perm_p (the rotated c1, NTT form); t = rc1
//   B  t[p][J] <- INTT_J                                   (coefficient form of the digits, canonical)
//   C  for every output prime I (the data limbs, then the special prime):
//        dig[p][J] = NTT_I(t[p][J] mod q_I), J != I;  prod[p][K][I] = sum_J (J == I ? rc1 : dig)[p][J] * key_p[J][K][I]
//   D  mod-down by the special prime P: r = INTT_P(prod[p][K][l]) + P/2;  u[p][K][i] = NTT_i((r mod q_i) - (P/2 mod q_i));
//      rot[p][K][i] = (prod[p][K][i] - u[p][K][i]) * P^-1  (+ c0_p[i] o perm_p for K = 0)
//   E  the window sums of rot (evah_rotate_weighted_sums), when the set has any
// Every value is a canonical residue at every step, so the outputs are the bits of the ordinary kernels.
//
// Ordering between phases — correct by construction, whatever part of the grid is resident (r5; the r4 form was a
// grid-wide spin barrier that assumed every workgroup resident and gave up after ~1 s otherwise).  Every phase is cut
// into FB_NCH chunks; (phase, chunk) pairs are numbered phase-major and HANDED OUT IN THAT ORDER by one atomic ticket
// counter: a workgroup takes the next ticket, waits until `done` — the count of finished chunks — has reached
// phase * FB_NCH (all earlier phases complete), does its chunk, adds one to `done`, and takes the next ticket, until
// the tickets run out.  A ticket is only ever held by a workgroup that is running, and a chunk waits only for tickets
// with smaller numbers, so the unfinished chunk with the smallest ticket never waits for anything unfinished: there is
// always a workgroup that can run to the end of its chunk, with one resident workgroup as with a thousand, and with any
// number of other kernels (or other active fallbacks) on other queues.  It is the ordered-ticket argument of decoupled
// look-back scans (rocPRIM's lookback_scan takes its tile ids from an atomic counter for the same reason).  No timeout,
// no host-visible failure word, no assumption about the grid size.
#pragma once
#include "launch.hip.h"

namespace evah {

struct FbPairs { // one chunk of (source, Galois element) pairs, by value
  const uint32_t *perm[KS_BATCH_MAX];
  const u64 *src[KS_BATCH_MAX]; // c0 of the source; c1 = src + src_ps * N
  const u64 *key[KS_BATCH_MAX]; // [digit J][K][k primes][N]
  uint32_t src_ps[KS_BATCH_MAX];
};
struct FbBufs {
  u64 *rc1, *t, *dig; // [np][l][N]
  u64 *prod;          // [np][2][l + 1][N]
  u64 *r;             // [np][2][N]
  u64 *u;             // [np][2][l][N]; the rotated ciphertexts end up here unless rot_out is given
  unsigned *bar;      // [0]: tickets handed out, [1]: chunks finished — both zero at launch (ZeroFlag's memset)
};
constexpr unsigned FB_NCH = 256; // chunks per phase

// one radix-2 stage (s = 0 .. logN - 1) of the in-place forward / inverse negacyclic transforms of `npoly` polynomials of
// N words, chunk `ch` of FB_NCH; polynomial pl is modulo primes[fixed >= 0 ? fixed : pl % per]; the inverse leaves out the
// factor N^-1 (the consumer multiplies by ninv)
__device__ __forceinline__ void fb_forward_stage(const DevCtx &cx, u64 *buf, uint32_t npoly, int fixed, uint32_t per, uint32_t s, unsigned ch) {
  const uint32_t N = cx.N, logh = cx.logN - 1, m = 1u << s, logt = logh - s;
  const size_t total = (size_t)npoly << logh, stride = (size_t)FB_NCH * blockDim.x;
  for (size_t idx = (size_t)ch * blockDim.x + threadIdx.x; idx < total; idx += stride) {
    const uint32_t pl = (uint32_t)(idx >> logh), b = (uint32_t)idx & ((1u << logh) - 1u);
    const uint32_t i = b >> logt, j = b & ((1u << logt) - 1u), t = 1u << logt;
    const uint32_t pr = fixed >= 0 ? (uint32_t)fixed : pl % per;
    const DevPrime pm = cx.primes[pr];
    const u64 W = cx.tw_fwd[(size_t)pr * N + m + i].x;
    u64 *x = buf + (size_t)pl * N + 2 * (size_t)i * t + j;
    const u64 a = x[0], v = mulmod(x[t], W, pm);
    x[0] = addmod(a, v, pm.q);
    x[t] = submod(a, v, pm.q);
  }
}
__device__ __forceinline__ void fb_inverse_stage(const DevCtx &cx, u64 *buf, uint32_t npoly, int fixed, uint32_t per, uint32_t s, unsigned ch) {
  const uint32_t N = cx.N, logh = cx.logN - 1, m = N >> (s + 1), logt = s;
  const size_t total = (size_t)npoly << logh, stride = (size_t)FB_NCH * blockDim.x;
  for (size_t idx = (size_t)ch * blockDim.x + threadIdx.x; idx < total; idx += stride) {
    const uint32_t pl = (uint32_t)(idx >> logh), b = (uint32_t)idx & ((1u << logh) - 1u);
    const uint32_t i = b >> logt, j = b & ((1u << logt) - 1u), t = 1u << logt;
    const uint32_t pr = fixed >= 0 ? (uint32_t)fixed : pl % per;
    const DevPrime pm = cx.primes[pr];
    const u64 W = cx.tw_inv[(size_t)pr * N + m + i].x;
    u64 *x = buf + (size_t)pl * N + 2 * (size_t)i * t + j;
    const u64 a = x[0], v = x[t];
    x[0] = addmod(a, v, pm.q);
    x[t] = mulmod(submod(a, v, pm.q), W, pm);
  }
}

// F: sums per window (0: a plain rotation set, the rotated ciphertexts go to rot_out[np][2][l N])
template <int F>
__global__ void __launch_bounds__(256)
k_rot_fallback(DevCtx cx, FbPairs pr, uint32_t np, uint32_t l, FbBufs b, u64 *rot_out, WinSumTab ws, uint32_t n_win, size_t out_ps) {
  if (cx.skipped()) return; // the hoisted results stand: every workgroup leaves before it takes a ticket
  const uint32_t N = cx.N, logN = cx.logN, k = cx.k, sp = k - 1;
  const size_t stride = (size_t)FB_NCH * blockDim.x;
  const size_t lN = (size_t)l * N;
  // phases, in order: A | B: logN inverse stages, scale | C: per output prime (reduce, logN forward stages, inner product)
  // | D: special rows, logN inverse stages, u, logN forward stages, combine | E: window sums (F > 0)
  const uint32_t pB = 1, pC = pB + logN + 1, cper = logN + 2, pD = pC + (l + 1) * cper, pE = pD + 2 * logN + 3;
  const uint32_t n_phases = pE + (F > 0 ? 1u : 0u);
  u64 *rot = F == 0 ? rot_out : b.u;
  // The ticket is made wave-uniform explicitly (readfirstlane): the loop condition and the phase dispatch are then scalar
  // branches.  (As a per-lane value read from LDS, with "if (threadIdx.x == 0) atomic" at the loop's tail, the structurizer
  // turned the loop into nested exec-mask loops in which the lanes other than lane 0 went round again — through the
  // barrier and into the same chunk — before lane 0 had taken the next ticket.)
  __shared__ unsigned s_ticket;
  if (threadIdx.x == 0) s_ticket = __hip_atomic_fetch_add(b.bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  unsigned ticket = __builtin_amdgcn_readfirstlane(s_ticket);
  while (ticket / FB_NCH < n_phases) {
    const uint32_t ph = ticket / FB_NCH;
    const unsigned ch = ticket % FB_NCH;
    if (ph > 0) { // every chunk of the earlier phases is finished (they hold smaller tickets: see the header)
      if (threadIdx.x == 0)
        while (__hip_atomic_load(b.bar + 1, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < ph * FB_NCH) __builtin_amdgcn_s_sleep(4);
      __syncthreads();
      __threadfence();
    }
    const size_t gt = (size_t)ch * blockDim.x + threadIdx.x;
    if (ph == 0) {
      // A: the rotated c1
      for (size_t idx = gt; idx < (size_t)np * lN; idx += stride) {
        const uint32_t p = (uint32_t)(idx / lN), n = (uint32_t)idx & (N - 1);
        const size_t J = (idx - (size_t)p * lN) >> logN;
        const u64 v = pr.src[p][((size_t)pr.src_ps[p] + J) * N + pr.perm[p][n]];
        b.rc1[idx] = v;
        b.t[idx] = v;
      }
    } else if (ph < pB + logN) {
      // B: digits in coefficient form
      fb_inverse_stage(cx, b.t, np * l, -1, l, ph - pB, ch);
    } else if (ph == pB + logN) {
      for (size_t idx = gt; idx < (size_t)np * lN; idx += stride) {
        const DevPrime pm = cx.primes[(idx >> logN) % l];
        b.t[idx] = mulmod(b.t[idx], pm.ninv, pm);
      }
    } else if (ph < pD) {
      // C: per output prime, the converted digits and the key inner product
      const uint32_t I = (ph - pC) / cper, sub = (ph - pC) % cper;
      const uint32_t kap = I == l ? sp : I;
      const DevPrime pm = cx.primes[kap];
      if (sub == 0) {
        for (size_t idx = gt; idx < (size_t)np * lN; idx += stride) b.dig[idx] = barrett64(b.t[idx], pm.q, pm.brt);
      } else if (sub <= logN) {
        fb_forward_stage(cx, b.dig, np * l, (int)kap, 1, sub - 1, ch);
      } else {
        for (size_t idx = gt; idx < (size_t)np * 2 * N; idx += stride) {
          const uint32_t n = (uint32_t)idx & (N - 1), K = (uint32_t)(idx >> logN) & 1u, p = (uint32_t)(idx >> (logN + 1));
          const u64 *key = pr.key[p] + ((size_t)K * k + kap) * N + n;
          u64 acc = 0;
          for (uint32_t J = 0; J < l; J++) {
            const size_t at = (size_t)p * lN + (size_t)J * N + n;
            const u64 d = J == I ? b.rc1[at] : b.dig[at];
            acc = addmod(acc, mulmod(d, key[(size_t)J * 2 * k * N], pm), pm.q);
          }
          b.prod[((size_t)(2 * p + K) * (l + 1) + I) * N + n] = acc;
        }
      }
    } else if (ph < pE) {
      // D: mod-down by the special prime
      const uint32_t sub = ph - pD;
      if (sub == 0) {
        for (size_t idx = gt; idx < (size_t)np * 2 * N; idx += stride)
          b.r[idx] = b.prod[((idx >> logN) * (l + 1) + l) * N + (idx & (N - 1))];
      } else if (sub <= logN) {
        fb_inverse_stage(cx, b.r, np * 2, (int)sp, 1, sub - 1, ch);
      } else if (sub == logN + 1) {
        const DevPrime pa = cx.primes[sp];
        for (size_t idx = gt; idx < (size_t)np * 2 * lN; idx += stride) {
          const uint32_t n = (uint32_t)idx & (N - 1), i = (uint32_t)((idx >> logN) % l);
          const size_t pp = idx / lN;
          const DevPrime pm = cx.primes[i];
          const u64 v = addmod(mulmod(b.r[pp * N + n], pa.ninv, pa), pa.q >> 1, pa.q);
          b.u[idx] = submod(barrett64(v, pm.q, pm.brt), cx.halfmod[(size_t)sp * k + i], pm.q);
        }
      } else if (sub <= 2 * logN + 1) {
        fb_forward_stage(cx, b.u, np * 2 * l, -1, l, sub - logN - 2, ch);
      } else {
        for (size_t idx = gt; idx < (size_t)np * 2 * lN; idx += stride) {
          const uint32_t n = (uint32_t)idx & (N - 1), i = (uint32_t)((idx >> logN) % l);
          const size_t pp = idx / lN;
          const uint32_t p = (uint32_t)(pp >> 1);
          const DevPrime pm = cx.primes[i];
          const ulonglong2 inv = cx.invq[(size_t)sp * k + i];
          u64 v = mul_shoup(submod(b.prod[(pp * (l + 1) + i) * N + n], b.u[idx], pm.q), inv.x, inv.y, pm.q);
          if ((pp & 1) == 0) v = addmod(v, pr.src[p][(size_t)i * N + pr.perm[p][n]], pm.q);
          rot[idx] = v;
        }
      }
    } else {
      if constexpr (F > 0) {
        // E: the window sums (k_window_sums, one coefficient per thread)
        for (size_t idx = gt; idx < (size_t)n_win * 2 * lN; idx += stride) {
          const uint32_t n = (uint32_t)idx & (N - 1), i = (uint32_t)((idx >> logN) % l);
          const uint32_t wK = (uint32_t)(idx / lN), w = wK >> 1, K = wK & 1u;
          const DevPrime pm = cx.primes[i];
          const size_t off = (size_t)i * N + n;
          u128_t acc[F];
#pragma unroll
          for (int f = 0; f < F; f++) acc[f] = {0, 0};
          const uint32_t first = ws.first[w], cnt = ws.count[w];
          for (uint32_t t = first; t < first + cnt; t++) {
            const u64 v = rot[(size_t)(2 * t + K) * lN + off];
            acc128(acc[0], v, ws.w0[t] ? ws.w0[t][off] : 1);
            if constexpr (F > 1) acc128(acc[1], v, ws.w1[t] ? ws.w1[t][off] : 1);
          }
          if (ws.id_src[w]) {
            const u64 v = ws.id_src[w][(size_t)K * ws.id_ps[w] * N + off];
            acc128(acc[0], v, ws.id_w0[w] ? ws.id_w0[w][off] : 1);
            if constexpr (F > 1) acc128(acc[1], v, ws.id_w1[w] ? ws.id_w1[w][off] : 1);
          }
          ws.out0[w][(size_t)K * out_ps + off] = barrett128(acc[0], pm);
          if constexpr (F > 1) ws.out1[w][(size_t)K * out_ps + off] = barrett128(acc[1], pm);
        }
      }
    }
    // this chunk is finished: its writes first, then the count the later phases wait for; then the next ticket
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) {
      __hip_atomic_fetch_add(b.bar + 1, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      s_ticket = __hip_atomic_fetch_add(b.bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    ticket = __builtin_amdgcn_readfirstlane(s_ticket);
  }
}
static_assert(sizeof(DevCtx) + sizeof(FbPairs) + sizeof(FbBufs) + sizeof(WinSumTab) + 64 <= 4096, "kernel arguments of k_rot_fallback");

} // namespace evah
