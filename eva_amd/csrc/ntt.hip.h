// ntt.hip.h — two-pass negacyclic NTT / INTT over 64-bit RNS primes for gfx950.
//
// Replaces SEAL's ntt_negacyclic_harvey / inverse_ntt_negacyclic_harvey as reached from
// /root/reference/eva/seal/seal_executor.h:200 (relinearize), :181/:188 (rotate_vector) and
// :213 (rescale_to_next).  Same mathematical transform (psi = minimal primitive 2N-th root,
// natural -> bit-reversed order forward, Gentleman-Sande inverse scaled by N^-1), so outputs
// are the same canonical residues.
//
// Decomposition (N = 2^logN, logN = a + b, a = ceil(logN/2)):
//   forward  pass 1 "strided": stages 0..a-1   — N/2^a... independent 2^a-point column NTTs
//            pass 2 "contig" : stages a..logN-1 — 2^a independent contiguous 2^b-point NTTs
//   inverse  pass 1 "contig" : stages logN-1..a, pass 2 "strided": stages a-1..0 (+ N^-1)
// The twiddle table is a binary heap (stage m, group i -> index m+i), so a sub-transform
// rooted at heap node `node` uses tw[(node << s) + v] at its local stage s: both passes share
// one table and no re-indexing is needed.
//
// One workgroup (256 threads) owns a tile of 2048 coefficients in LDS (16 KiB + padding); each
// thread keeps 8 coefficients in registers per round and runs 3 butterfly stages on them per LDS
// round trip (measured best on MI355X: vs 4 per thread +12 %, vs 16 +10 %; the fused key-switch
// kernel below uses one wave and 4 per thread).  Lazy butterflies: forward values live in
// [0,16q) with a conditional subtraction every other stage, inverse in [0,5q) (q < 2^60); only
// the value finally stored is canonical.
//
// The first pass reads through Op::load and the second writes through Op::store / store_fwd,
// which is how the digit base-conversion, the rescale / mod-down combine and the +q/2 rounding
// offset are fused into the transforms instead of being separate HBM round trips; the ops work
// on lazy values where the moduli allow it (no Barrett reduction on the way in or out).
#pragma once
// cache policy of the converted-digit stream into ks_inner_kernel<MAC3> (aux bits of global_load_lds: 1 = sc0, 2 = nt,
// 16 = sc1).  0 measured best; 2 (non-temporal: the digits are read once and should not push the key out of the Infinity
// Cache) was the r4 verdict's item 9 — profiles/r05_tuning_notes.md
#ifndef EVAH_DIGIT_AUX
#define EVAH_DIGIT_AUX 0
#endif
#include "devmath.hip.h"
#include <type_traits>

namespace evah {


constexpr int NTT_THREADS = 256; // threads per workgroup (max); tile = NTT_THREADS << LR coefficients
constexpr uint32_t HOIST_ZERO_CAP = 64; // zero digit coefficients a hoisted rotation set corrects individually

// LR = log2(coefficients per thread): 4 -> 16 coefficients / up to 4 stages per LDS round trip,
// 3 -> 8 coefficients / 3 stages (half the registers, twice the waves in flight)
template <int P, int LR> struct Rounds {
  static constexpr int NR = (P + LR - 1) / LR;
  static constexpr int bits(int i) { return P / NR + (i < P % NR ? 1 : 0); }
  static constexpr int lo(int i) {
    int l = P;
    for (int j = 0; j <= i; j++) l -= bits(j);
    return l;
  }
};

// LDS layout of a sub-transform: element e at lds_pad<P>(e).  The pad is a sum of shifts, so it is additive over
// disjoint bit fields (ntt_round adds the per-element part as an immediate offset).  One word per 16 elements for the
// short sub-transforms, two for 2^7 points and more: the 64-bit accesses of a round are served in groups of 8 / 16
// lanes over 32 banks (MI355X_MICROARCH.md, LDS), and with one word the 8-coefficient rounds of a 256-point transform
// spend 30 % of their LDS cycles in 2-way conflicts (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.30 measured, the
// same from a simulation of the access pattern); two words bring that to 12 % (r04_tuning_notes.md)
#ifndef EVAH_LDS_PAD2
#define EVAH_LDS_PAD2 1 // 0: one word per 16 elements everywhere (the r03 layout; A/B switch of the build)
#endif
template <int P> __device__ __forceinline__ int lds_pad(int e) {
  return (EVAH_LDS_PAD2 && P >= 7) ? e + 2 * (e >> 4) : e + (e >> 4);
}
template <int P> constexpr int lds_sub_stride() {
  return (EVAH_LDS_PAD2 && P >= 7) ? (1 << P) + 2 * (((1 << P) - 1) >> 4) : (1 << P) + ((1 << P) >> 4) + 1;
}
#ifndef EVAH_ROUND_BARRIER
#define EVAH_ROUND_BARRIER 1 // 0 (experiment): rounds of a sub-transform worked on by one wave exchange through LDS in program order only
#endif

// mul_tw_lazy5 (devmath.hip.h): x*w - q~*q in [0, 4q) for any 64-bit x with a 3-multiply quotient estimate.
// Moduli are < 2^60, so every lazy value below stays < 16q <= 2^64.
// a + (x*w - q~*q): the mad chain of mul_tw_lazy5 started from `a` instead of 0 (the first
// v_mad_u64_u32 has a free 64-bit addend), so the butterfly's sum costs nothing extra
__device__ __forceinline__ u64 mul_tw_lazy5_add(u64 x, u64 w, u64 ws, u64 nq, u64 a) {
  const uint32_t x0 = (uint32_t)x, x1 = (uint32_t)(x >> 32), s0 = (uint32_t)ws, s1 = (uint32_t)(ws >> 32);
  const u64 qt = (u64)x1 * s1 + (u64)__umulhi(x1, s0) + (u64)__umulhi(x0, s1);
  return (a + x * w) + qt * nq;
}
// The same value with EVERY partial product going through v_mad_u64_u32, the low-word cross terms
// included: the compiler lowers `hi += lo32(a*b)` to v_mul_lo_u32 + v_add3_u32, and on gfx950 a chain of
// v_mad_u64_u32 whose upper result word is simply never read issues faster than that pair
// (scripts/microbench_bfly.hip: 60.0 against 68.4 SIMD cycles per wave-butterfly; in the library: the
// 8-coefficient strided forward passes -3..-6 %, but +1.5 % in the fused key-switch kernel and +8 % in
// the combine pass, whose register budgets it tips — profiles/r03_tuning_notes.md).  Inline asm, because
// the compiler narrows the C form back.  ntt_round selects it per pass.
__device__ __forceinline__ u64 mad64(uint32_t a, uint32_t b, u64 c) {
  u64 d;
  asm("v_mad_u64_u32 %0, vcc, %1, %2, %3" : "=&v"(d) : "v"(a), "v"(b), "v"(c) : "vcc");
  return d;
}
__device__ __forceinline__ u64 mul_tw_lazy5_add_mad(u64 x, u64 w, u64 ws, u64 nq, u64 a) {
  const uint32_t x0 = (uint32_t)x, x1 = (uint32_t)(x >> 32), s0 = (uint32_t)ws, s1 = (uint32_t)(ws >> 32);
  const u64 t = (u64)x1 * s1 + (u64)__umulhi(x1, s0) + (u64)__umulhi(x0, s1);
  const uint32_t w0 = (uint32_t)w, w1 = (uint32_t)(w >> 32);
  const uint32_t t0 = (uint32_t)t, t1 = (uint32_t)(t >> 32), n0 = (uint32_t)nq, n1 = (uint32_t)(nq >> 32);
  u64 r = mad64(x0, w0, a);
  r = mad64(t0, n0, r);
  u64 h = mad64(x0, w1, r >> 32); // from here on only the low word of h matters: its upper word is never read
  h = mad64(x1, w0, h);
  h = mad64(t0, n1, h);
  h = mad64(t1, n0, h);
  return (h << 32) | (uint32_t)r;
}
#ifndef EVAH_MADLO
#define EVAH_MADLO 1 // 0: the compiler's form everywhere (A/B switch of the build)
#endif
// The same products for q = 2^b - c (b = 32 + sh, c < 2^32: DevPrime::tb_c / tb_sh):  x*w - t*q = x*w + t*c - t*2^b, and
// t*2^b mod 2^64 is (t0 << sh) in the upper word — the product with the modulus costs TWO multiplies (t0*c, t1*c) plus a
// 32-bit shift and subtract instead of three.  8 multiplies per butterfly instead of 9; the value is the same 64-bit
// word, so every lazy bound above holds unchanged.  Measured and NOT kept (r04_tuning_notes.md 15): 14.67-14.77 k against
// 14.85-15.01 k op-triples/s — the key-switch kernel got slower (841 -> 876 us), the digit pass did not get faster.
#ifndef EVAH_TBMUL
#define EVAH_TBMUL 0 // 1: the experiment (A/B switch of the build)
#endif
__device__ __forceinline__ u64 mul_tw_tb_add(u64 x, u64 w, u64 ws, uint32_t c, uint32_t sh, u64 a) {
  const uint32_t x0 = (uint32_t)x, x1 = (uint32_t)(x >> 32), s0 = (uint32_t)ws, s1 = (uint32_t)(ws >> 32);
  const u64 t = (u64)x1 * s1 + (u64)__umulhi(x1, s0) + (u64)__umulhi(x0, s1);
  return ((a + x * w) + t * (u64)c) - ((u64)((uint32_t)t << sh) << 32);
}
__device__ __forceinline__ u64 mul_tw_tb_add_mad(u64 x, u64 w, u64 ws, uint32_t c, uint32_t sh, u64 a) {
  const uint32_t x0 = (uint32_t)x, x1 = (uint32_t)(x >> 32), s0 = (uint32_t)ws, s1 = (uint32_t)(ws >> 32);
  const u64 t = (u64)x1 * s1 + (u64)__umulhi(x1, s0) + (u64)__umulhi(x0, s1);
  const uint32_t w0 = (uint32_t)w, w1 = (uint32_t)(w >> 32), t0 = (uint32_t)t, t1 = (uint32_t)(t >> 32);
  u64 r = mad64(x0, w0, a);
  r = mad64(t0, c, r);
  u64 h = mad64(x0, w1, r >> 32); // from here on only the low word of h matters
  h = mad64(x1, w0, h);
  h = mad64(t1, c, h);
  return ((u64)((uint32_t)h - (t0 << sh)) << 32) | (uint32_t)r;
}
// forward Cooley-Tukey butterfly.  The twiddle product is in [0,4q), so each stage grows the
// bound by 4q; moduli are < 2^60 (16q < 2^64), which leaves room to reduce only every other stage:
//   REDUCE stage : X < 16q -> x < 8q  -> outputs < 12q
//   plain stage  : X < 12q            -> outputs < 16q
// (Y only feeds the multiply, which accepts any 64-bit value.)  X' = x + t comes out of the mad
// chain; Y' = x + 4q - t = (2x + 4q) - X' (mod 2^64; the true value is < 16q).
#ifndef EVAH_TOPBIT
#define EVAH_TOPBIT 1 // 0: compare-and-subtract reductions for every prime (the r03 butterflies; A/B switch of the build)
#endif
// TB ("top bits"): for q = 2^b - c with b > 32 and c < 2^32 — every CoeffModulus::Create prime of 33..60 bits, the
// search walks down from 2^b in steps of 2N — the reduction is  x = (X mod 2^b) + (X >> b) c :  congruent to X and
// < q + 16c for any X < 16q, in a shift, a mask and one v_mad_u64_u32 where the compare-and-subtract form takes four
// instructions to get below 8q.  Starting from ~q instead of 8q, THREE stages fit before the next reduction (5q, 9q,
// 13q) instead of two, so a pass reduces on every third stage (ntt_round).  Primes of another shape (DevPrime::tb_c
// == 0) take the compare-and-subtract butterflies; the choice is block-uniform, made once per transform.
template <bool REDUCE, bool MAD = false, bool TB = false>
__device__ __forceinline__ void bfly_fwd(u64 &X, u64 &Y, ulonglong2 w, u64 nq, u64 q4, u64 q8, u64 nq8, uint32_t tbc = 0,
                                         uint32_t tbs = 0, uint32_t tbm = 0) {
  u64 x = X;
  if constexpr (REDUCE && TB) {
    const uint32_t hi = (uint32_t)(X >> 32);
    x = mad64(hi >> tbs, tbc, ((u64)(hi & tbm) << 32) | (uint32_t)X); // < q + 16c
  } else if constexpr (REDUCE) {
    x = X + (X >= q8 ? nq8 : 0);
  }
  if constexpr (TB && EVAH_TBMUL) X = MAD ? mul_tw_tb_add_mad(Y, w.x, w.y, tbc, tbs, x) : mul_tw_tb_add(Y, w.x, w.y, tbc, tbs, x);
  else X = MAD ? mul_tw_lazy5_add_mad(Y, w.x, w.y, nq, x) : mul_tw_lazy5_add(Y, w.x, w.y, nq, x);
  Y = ((x << 1) + q4) - X;
}
// inverse Gentleman-Sande butterfly, X,Y in [0,5q) -> [0,5q)
__device__ __forceinline__ void bfly_inv(u64 &X, u64 &Y, ulonglong2 w, u64 nq, u64 q5, u64 nq5) {
  u64 s = X + Y;
  u64 d = X + q5 - Y;
  X = s + (s >= q5 ? nq5 : 0);
  Y = mul_tw_lazy5(d, w.x, w.y, nq);
}

// One register round: RB stages over bit range [LO, LO+RB) of the P-bit local index.
// RED_EVEN: forward passes reduce on even (true) or odd (false) local stage indices — the first
// (strided) pass starts from canonical input and reduces on odd stages, so it always exits < 16q;
// the second pass therefore reduces on even stages.
// LASTFOLD: this pass ends the inverse transform (its very last stage multiplies by N^-1) — the strided pass,
// unless a contiguous pass borrows the local-heap indexing (ntt_loop_kernel)
// TB forward passes reduce on every third stage.  First pass (RED_EVEN false; input canonical or lazy < 12q, output
// < 9q): stages s = P - 2 (mod 3), and stage 0 as well when that leaves stages 0 and 1 unreduced; second pass (input
// < 9q): stages s = 1 (mod 3), output < 13q.  Every value stays below 16q < 2^64.
template <int P, bool RED_EVEN> constexpr bool tb_reduce_stage(int s) {
  if (RED_EVEN) return s % 3 == 1;
  return s % 3 == (P - 2) % 3 || (s == 0 && (P - 2) % 3 == 2);
}
// lin_in != nullptr: this round's inputs are read from an UNPADDED copy of the sub-transform (element e at lin_in[e] —
// where an LDS-DMA load put the tile, ks_inner_kernel<MAC3>), its outputs go to the padded tile as usual
struct NoHook { __device__ __forceinline__ void operator()() const {} };
template <int P, int LR, int RB, int LO, bool INVERSE, bool STRIDED, bool RED_EVEN, bool LASTFOLD = STRIDED, bool TB = false, class AfterLoad = NoHook>
__device__ __forceinline__ void ntt_round(u64 *sub_lds, int tid, uint32_t h, uint32_t pre,
                                          const ulonglong2 *__restrict__ tw, const DevPrime &pm, const u64 *lin_in = nullptr,
                                          AfterLoad after_load = AfterLoad()) {
  constexpr int NTT_R = 1 << LR;
  constexpr int S = 1 << P, TPS = S / NTT_R, G = NTT_R >> RB, NU = 1 << RB;
  constexpr int S0 = P - LO - RB; // local stages above this round
  const u64 q = pm.q, nq = pm.nq, q5 = pm.q5, q4 = pm.q4, q8 = pm.q8, nq5 = pm.nq5, nq8 = pm.nq8;
  (void)q4; (void)q8; (void)q5; (void)nq5; (void)nq8;
  // the all-mad twiddle product pays in the 8-coefficient strided forward passes (digit conversion
  // -4.4 %, mod-down pass 1 -2.5 %); the contiguous passes (combine epilogue: +8 %), the one-wave
  // key-switch kernel (LR = 2: +1.5 %) and the inverse butterflies keep the compiler's form
  constexpr bool MAD = EVAH_MADLO && !INVERSE && STRIDED && LR == 3;
  u64 x[NTT_R];
#pragma unroll
  for (int g = 0; g < G; g++) {
    const int o = g * TPS + tid;
    const int o_lo = o & ((1 << LO) - 1), o_hi = o >> LO;
    const int ebase = (o_hi << (LO + RB)) | o_lo;
    // ebase and (u << LO) occupy disjoint bit fields, so pad(ebase | u << LO) = pad(ebase) + pad(u << LO):
    // one padded base per group, the per-element part is an immediate offset of the LDS access
    u64 *grp = sub_lds + lds_pad<P>(ebase);
    if (lin_in) { // compile-time known at every call site (inlined)
#pragma unroll
      for (int u = 0; u < NU; u++) x[g * NU + u] = lin_in[ebase | (u << LO)];
      if (g == G - 1) after_load(); // every input of the round is on its way to registers: `lin_in` may be refilled
    } else {
#pragma unroll
      for (int u = 0; u < NU; u++) x[g * NU + u] = grp[lds_pad<P>(u << LO)];
    }
    const uint32_t node = STRIDED ? ((1u << S0) | (uint32_t)o_hi)
                                  : ((1u << (pre + S0)) | (h << S0) | (uint32_t)o_hi);
    if (!INVERSE) {
#pragma unroll
      for (int s = 0; s < RB; s++) {
        const int half = 1 << (RB - 1 - s);
#pragma unroll
        for (int u = 0; u < NU; u++) {
          if (u & half) continue;
          const int v = u >> (RB - s);
          const ulonglong2 w = tw[((size_t)node << s) + v];
          const bool red = TB ? tb_reduce_stage<P, RED_EVEN>(S0 + s) : ((((S0 + s) & 1) == 0) == RED_EVEN);
          if (red) bfly_fwd<true, MAD, TB>(x[g * NU + u], x[g * NU + u + half], w, nq, q4, q8, nq8, pm.tb_c, pm.tb_sh, pm.tb_mask);
          else bfly_fwd<false, MAD, TB>(x[g * NU + u], x[g * NU + u + half], w, nq, q4, q8, nq8, pm.tb_c, pm.tb_sh, pm.tb_mask);
        }
      }
    } else {
#pragma unroll
      for (int s = RB - 1; s >= 0; s--) {
        const int half = 1 << (RB - 1 - s);
        // the very last stage of the whole inverse transform folds in N^-1
        const bool last = LASTFOLD && (S0 == 0) && (s == 0);
#pragma unroll
        for (int u = 0; u < NU; u++) {
          if (u & half) continue;
          u64 &X = x[g * NU + u], &Y = x[g * NU + u + half];
          if (last) {
            u64 sum = X + Y, d = X + q5 - Y; // any 64-bit input is fine for the exact Shoup product
            X = mul_shoup(sum, pm.ninv, pm.ninv_s, q);
            Y = mul_shoup(d, pm.w0ninv, pm.w0ninv_s, q);
          } else {
            const int v = u >> (RB - s);
            const ulonglong2 w = tw[((size_t)node << s) + v];
            bfly_inv(X, Y, w, nq, q5, nq5);
          }
        }
      }
    }
#pragma unroll
    for (int u = 0; u < NU; u++) grp[lds_pad<P>(u << LO)] = x[g * NU + u];
  }
}

// WAVESYNC: the workgroup is ONE wave (ks_inner_kernel<MAC3>), whose DS operations execute in issue order: the rounds
// need no s_barrier, and must not use __syncthreads() — with an LDS-DMA load in flight its fence drains vmcnt, i.e. waits
// for the very loads that are meant to overlap the transform
template <int P, int LR, int I, bool INVERSE, bool STRIDED, bool RED_EVEN, bool LASTFOLD = STRIDED, bool TB = false, bool WAVESYNC = false> struct RoundSeq {
  // forward: rounds 0..NR-1 (top bits first); inverse: NR-1..0 (low bits first)
  static __device__ __forceinline__ void run(u64 *sub_lds, int tid, uint32_t h, uint32_t pre,
                                             const ulonglong2 *tw, const DevPrime &pm) {
    using RS = Rounds<P, LR>;
    constexpr int idx = INVERSE ? (RS::NR - 1 - I) : I;
    ntt_round<P, LR, RS::bits(idx), RS::lo(idx), INVERSE, STRIDED, RED_EVEN, LASTFOLD, TB>(sub_lds, tid, h, pre, tw, pm);
    if constexpr (I + 1 < RS::NR) {
      // a sub-transform of <= 64 threads lives in one wave: its LDS exchange is ordered by the wave's own DS queue
      if constexpr (WAVESYNC || (!EVAH_ROUND_BARRIER && ((1 << P) >> LR) <= 64)) __builtin_amdgcn_wave_barrier();
      else __syncthreads();
      RoundSeq<P, LR, I + 1, INVERSE, STRIDED, RED_EVEN, LASTFOLD, TB, WAVESYNC>::run(sub_lds, tid, h, pre, tw, pm);
    }
  }
};
// forward rounds of one pass: the top-bit butterflies when the prime has the shape, else compare-and-subtract
// (block-uniform: pm is the workgroup's prime)
template <int P, int LR, bool STRIDED, bool RED_EVEN>
__device__ __forceinline__ void forward_rounds(u64 *sub_lds, int tid, uint32_t h, uint32_t pre, const ulonglong2 *tw, const DevPrime &pm) {
  if (EVAH_TOPBIT && pm.tb_c) RoundSeq<P, LR, 0, false, STRIDED, RED_EVEN, STRIDED, true>::run(sub_lds, tid, h, pre, tw, pm);
  else RoundSeq<P, LR, 0, false, STRIDED, RED_EVEN, STRIDED, false>::run(sub_lds, tid, h, pre, tw, pm);
}

// One pass.  grid.x = (N / tile) * jx-count (tile index in the low log_tiles bits), grid.y / grid.z
// = the op's job coordinates (no integer division in the kernel), block = tile >> LR threads.
// FULL: the tile is the full NTT_THREADS << LR coefficients (every N >= 2048), so the thread
// count and the tile shape are compile-time constants and the per-element global / LDS addresses
// of the load and store loops are one base plus immediate steps.
#ifndef EVAH_PREFETCH
#define EVAH_PREFETCH 1
#endif
template <class Op, class = void> struct HasPre : std::false_type {};
template <class Op> struct HasPre<Op, std::void_t<typename Op::Pre>> : std::true_type {};
struct NoPre {};
template <class Op, bool ON> struct PreOf { using type = NoPre; };
template <class Op> struct PreOf<Op, true> { using type = typename Op::Pre; };
// hooks that only the looped contiguous pass uses (ntt_loop_kernel: one wave walks several jobs, so whatever it waits
// for at a job's start or end is exposed once per job):
//   LoopPre / loop_prefetch / store_fwd_loop   the forward epilogue's operands, requested before the job's transform
//   Raw / raw_load / finish_load               the inverse pass's input transform split into its loads (issued before the
//                                              PREVIOUS job's transform) and its arithmetic (after it)
template <class Op, class = void> struct HasLoopPre : std::false_type {};
template <class Op> struct HasLoopPre<Op, std::void_t<typename Op::LoopPre>> : std::true_type {};
template <class Op, bool ON> struct LoopPreOf { using type = NoPre; };
template <class Op> struct LoopPreOf<Op, true> { using type = typename Op::LoopPre; };
template <class Op, class = void> struct HasRaw : std::false_type {};
template <class Op> struct HasRaw<Op, std::void_t<typename Op::Raw>> : std::true_type {};
template <class Op, bool ON> struct RawOf { using type = NoPre; };
template <class Op> struct RawOf<Op, true> { using type = typename Op::Raw; };
// an op whose input transform reads SEVERAL words per element (OpWinLin: one per rotation of a window) fills the thread's
// NTT_R elements itself — term-outer, so that the NTT_R loads of a term are in flight together instead of one element's
// terms one after the other:  Op::fill_all<NTT_R>(cx, jb, pm, n0, nstep, out)
template <class Op, class = void> struct HasFillAll : std::false_type {};
template <class Op> struct HasFillAll<Op, std::void_t<decltype(Op::fills_all)>> : std::bool_constant<Op::fills_all> {};

// An op whose transform starts a launch set may have words to clear before the set's later kernels count into them (the
// zero-coefficient record and the fallback's ticket words of a hoisted rotation set, OpPlainT<ZEROS>): the first workgroup
// of the inverse transform's FIRST pass clears them — the recording happens in its second pass, a kernel boundary later —
// so the set needs no memset launch of its own.
template <class Op, class = void> struct ClearsWords : std::false_type {};
template <class Op> struct ClearsWords<Op, std::void_t<decltype(Op::clears_words)>> : std::bool_constant<Op::clears_words> {};
template <class Op> __device__ __forceinline__ void first_pass_clear(const typename Op::Params &prm) {
  if constexpr (ClearsWords<Op>::value) {
    if (prm.clear_words && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0)
      for (uint32_t w = threadIdx.x; w < prm.clear_words; w += blockDim.x) prm.clear_base[w] = 0;
  }
}

template <int P, int LR, bool STRIDED, bool INVERSE, class Op, bool FULL>
__global__ void __launch_bounds__(NTT_THREADS)
ntt_pass_kernel(DevCtx cx, typename Op::Params prm, int logC_rt, int log_tiles) {
  extern __shared__ __attribute__((aligned(16))) u64 lds[];
  constexpr int NTT_R = 1 << LR;
  constexpr int S = 1 << P, TPS = S / NTT_R, SP = lds_sub_stride<P>();
  constexpr bool FIRST = (STRIDED != INVERSE);
  if (cx.skipped()) return;
  if constexpr (INVERSE && !STRIDED) first_pass_clear<Op>(prm);
  const uint32_t tile_idx = blockIdx.x & ((1u << log_tiles) - 1u);
  typename Op::Job jb;
  if (!Op::setup(cx, prm, blockIdx.x >> log_tiles, blockIdx.y, blockIdx.z, jb)) return; // block-uniform
  const DevPrime pm = cx.primes[jb.prime];
  const ulonglong2 *tw = (INVERSE ? cx.tw_inv : cx.tw_fwd) + (size_t)jb.prime * cx.N;
  const int logC = FULL ? (8 + LR - P) : logC_rt;
  const int C = 1 << logC, T = FULL ? NTT_THREADS : (int)blockDim.x;
  const uint32_t pre = STRIDED ? 0u : (cx.logN - P);
  const uint32_t stride_log = cx.logN - P; // strided pass: distance between local elements

  // ---- element it of this thread: global index n0 + it * nstep, LDS slot lds_at(it)
  uint32_t gbase, sub0 = 0;
  if (STRIDED) gbase = tile_idx << logC; // column c0 (pre = 0 => single prefix)
  else { sub0 = tile_idx << logC; gbase = sub0 << P; }
  // strided: idx = tid + it*T -> column c = idx mod C (fixed, T is a multiple of C), row e = e0 + it*(T/C)
  // contig : idx = tid + it*T -> sub-transform (tid >> P) + it*(T >> P), element tid mod S (fixed, P <= 8)
  constexpr int ES = 1 << (P - LR); // FULL: rows per step of the strided pass = NTT_THREADS >> logC
  constexpr bool LINEAR = FULL && (STRIDED ? (ES % 16 == 0) : (P <= 8));
  uint32_t n0, nstep;
  int l0;
  if (STRIDED) {
    const int c = threadIdx.x & (C - 1), e0 = threadIdx.x >> logC;
    n0 = gbase + ((uint32_t)e0 << stride_log) + c;
    nstep = (uint32_t)(T >> logC) << stride_log;
    l0 = c * SP + lds_pad<P>(e0);
  } else {
    n0 = gbase + threadIdx.x;
    nstep = T;
    l0 = (threadIdx.x >> P) * SP + lds_pad<P>(threadIdx.x & (S - 1));
  }
  auto lds_at = [&](int it) -> int {
    if constexpr (LINEAR) {
      return l0 + it * (STRIDED ? lds_pad<P>(ES) : (NTT_THREADS >> P) * SP);
    } else {
      const int idx = threadIdx.x + it * T;
      if (STRIDED) return (idx & (C - 1)) * SP + lds_pad<P>(idx >> logC);
      return (idx >> P) * SP + lds_pad<P>(idx & (S - 1));
    }
  };

  // ---- tile -> LDS.  First passes fuse the op's input transform; when the moduli involved are
  // of similar size (block-uniform test in Op::setup) the cheap form is used: the butterflies
  // accept lazy values (< 12q forward), so no Barrett reduction is needed on the way in
  auto fill = [&](auto lazy_tag) {
    constexpr bool LZ = decltype(lazy_tag)::value;
    if constexpr (FIRST && HasFillAll<Op>::value) {
      u64 v[NTT_R];
      Op::template fill_all<NTT_R>(cx, jb, pm, n0, nstep, v);
#pragma unroll
      for (int it = 0; it < NTT_R; it++) lds[lds_at(it)] = v[it];
    } else {
#pragma unroll
      for (int it = 0; it < NTT_R; it++) {
        const uint32_t n = n0 + it * nstep;
        lds[lds_at(it)] = FIRST ? Op::template load<LZ>(cx, jb, pm, n) : jb.dst[n];
      }
    }
  };
  // second forward pass: the epilogue's operands are requested before the tile, so they are in flight
  // while the butterflies run (ops that opt in with `Pre` / `prefetch` / `store_fwd_pre`: the mod-down
  // combine, whose launches wait for bytes — Harris 1.155 -> 1.135 ms host valuations, 1.015 -> 0.998 resident)
  constexpr bool PREFETCH = EVAH_PREFETCH && !FIRST && !INVERSE && HasPre<Op>::value;
  typename PreOf<Op, PREFETCH>::type epi[NTT_R];
  if constexpr (PREFETCH) {
#pragma unroll
    for (int it = 0; it < NTT_R; it++) epi[it] = Op::prefetch(cx, jb, pm, n0 + it * nstep);
  }
  if (FIRST && !INVERSE && jb.lazy) fill(std::true_type{});
  else fill(std::false_type{});
  // strided pass: every column transform of the tile uses the same 2^P twiddles (heap nodes
  // 1..2^P-1) — stage them in LDS once per workgroup instead of per-thread global loads
  ulonglong2 *twl = reinterpret_cast<ulonglong2 *>(lds + ((C * SP + 1) & ~1));
  if (STRIDED)
    for (int idx = threadIdx.x; idx < S; idx += T) twl[idx] = tw[idx];
  __syncthreads();

  // ---- register rounds
  {
    const int sub = threadIdx.x / TPS, tid = threadIdx.x % TPS;
    if constexpr (INVERSE) RoundSeq<P, LR, 0, true, STRIDED, !STRIDED>::run(lds + sub * SP, tid, sub0 + sub, pre, STRIDED ? twl : tw, pm);
    else forward_rounds<P, LR, STRIDED, !STRIDED>(lds + sub * SP, tid, sub0 + sub, pre, STRIDED ? twl : tw, pm);
  }
  __syncthreads();

  // ---- LDS -> global
#pragma unroll
  for (int it = 0; it < NTT_R; it++) {
    const uint32_t n = n0 + it * nstep;
    u64 v = lds[lds_at(it)];
    if (FIRST) {
      jb.dst[n] = v; // lazy intermediate
    } else {
      if constexpr (INVERSE) Op::store(cx, jb, pm, n, v);   // canonical
      else if constexpr (PREFETCH) Op::store_fwd_pre(cx, jb, pm, n, v, epi[it]);
      else Op::store_fwd(cx, jb, pm, n, v);                 // lazy [0,16q): the op reduces as it needs
    }
  }
}

// Contiguous pass, one wave per workgroup, several jobs per workgroup ("looped").  A contiguous pass runs the 2^P-point
// sub-transforms rooted at heap nodes 2^pre + h: every sub-transform has its OWN 2^P - 1 twiddles, so ntt_pass_kernel
// fetches 16 bytes of twiddle for every 8-byte coefficient it transforms — per-thread global loads in front of every
// register round, which is what these passes waited for (VALU busy 0.56-0.66 where the strided passes and the
// key-switch kernel, whose twiddles sit in LDS, reach 0.76-0.98).  The twiddles depend on (prime, tile) only: all
// polynomials of a launch that share the prime — both polynomials of a ciphertext, every instance of a batched call —
// can use one copy.  So a workgroup (64 threads, 4 coefficients per thread, a tile of 256 coefficients: the shape of
// ks_inner_kernel) stages the local twiddle heaps of its tile in LDS once and walks up to `nloop` jobs along the grid
// axis Op::loop_axis, the next job's tile in flight while the current one is transformed.
//   forward (second pass): tile from jb.dst (the strided pass's intermediate), out through Op::store_fwd(_pre)
//   inverse (first pass) : tile through Op::load, lazy intermediate to jb.dst
// grid.x = n_tiles * (x-extent), grid.y / grid.z = the op's job coordinates with the loop axis divided by nloop.
template <int P, int LR, bool INVERSE, class Op>
__global__ void __launch_bounds__(64)
ntt_loop_kernel(DevCtx cx, typename Op::Params prm, int logC, int log_tiles, uint32_t nloop, uint32_t loop_count) {
  extern __shared__ __attribute__((aligned(16))) u64 lds[];
  if (cx.skipped()) return;
  if constexpr (INVERSE) first_pass_clear<Op>(prm);
  constexpr int NTT_R = 1 << LR, NPAIR = NTT_R / 2;
  constexpr int S = 1 << P, TPS = S / NTT_R, SP = lds_sub_stride<P>();
  constexpr int AX = Op::loop_axis;
  const uint32_t tile_idx = blockIdx.x & ((1u << log_tiles) - 1u);
  uint32_t co[3] = {blockIdx.x >> log_tiles, blockIdx.y, blockIdx.z};
  const uint32_t j0 = co[AX] * nloop, j1 = (j0 + nloop < loop_count) ? j0 + nloop : loop_count;
  const int T = blockDim.x, C = 1 << logC;
  const uint32_t pre = cx.logN - P, sub0 = tile_idx << logC, gbase = sub0 << P;
  const int sub = threadIdx.x / TPS, tid = threadIdx.x % TPS;
  typename Op::Job jb;
  // first job that exists (an op may skip coordinates: OpKsDigit's I == J); every job of the walk has the same prime
  uint32_t j = j0;
  auto setup = [&](uint32_t jj, typename Op::Job &out) {
    uint32_t c3[3] = {co[0], co[1], co[2]};
    c3[AX] = jj;
    return Op::setup(cx, prm, c3[0], c3[1], c3[2], out);
  };
  while (j < j1 && !setup(j, jb)) j++;
  if (j >= j1) return; // block-uniform
  const DevPrime pm = cx.primes[jb.prime];
  const ulonglong2 *tw = (INVERSE ? cx.tw_inv : cx.tw_fwd) + (size_t)jb.prime * cx.N;
  ulonglong2 *twl = reinterpret_cast<ulonglong2 *>(lds + ((C * SP + 1) & ~1));
  for (int idx = threadIdx.x; idx < (C << P); idx += T) {
    const int sb = idx >> P, n = idx & (S - 1);
    if (n) {
      const int d = 31 - __clz(n);
      twl[idx] = tw[((size_t)((1u << pre) + sub0 + sb) << d) + (n - (1 << d))];
    }
  }
  auto load_tile = [&](const typename Op::Job &jj, ulonglong2 *d) {
#pragma unroll
    for (int it = 0; it < NPAIR; it++) {
      const uint32_t n = gbase + 2 * (threadIdx.x + it * T);
      if constexpr (INVERSE) {
        d[it].x = Op::template load<false>(cx, jj, pm, n);
        d[it].y = Op::template load<false>(cx, jj, pm, n + 1);
      } else {
        d[it] = *reinterpret_cast<const ulonglong2 *>(jj.dst + n);
      }
    }
  };
  constexpr bool PREFETCH = EVAH_PREFETCH && !INVERSE && HasPre<Op>::value;
  constexpr bool LOOPPRE = EVAH_PREFETCH && !INVERSE && !PREFETCH && HasLoopPre<Op>::value;
  constexpr bool RAW = INVERSE && HasRaw<Op>::value;
  ulonglong2 dreg[NPAIR];
  typename RawOf<Op, RAW>::type raw[NTT_R];
  load_tile(jb, dreg);
  while (true) {
    // next existing job of the walk (block-uniform)
    typename Op::Job jn;
    uint32_t jnext = j + 1;
    while (jnext < j1 && !setup(jnext, jn)) jnext++;
    const bool more = jnext < j1;
    __syncthreads(); // the previous job's LDS reads are done
#pragma unroll
    for (int it = 0; it < NPAIR; it++) {
      const int idx = 2 * (threadIdx.x + it * T);
      const int sb = idx >> P, e = idx & (S - 1);
      lds[sb * SP + lds_pad<P>(e)] = dreg[it].x;
      lds[sb * SP + lds_pad<P>(e + 1)] = dreg[it].y;
    }
    typename PreOf<Op, PREFETCH>::type epi[NTT_R];
    if constexpr (PREFETCH) {
#pragma unroll
      for (int it = 0; it < NTT_R; it++) epi[it] = Op::prefetch(cx, jb, pm, gbase + 2 * (threadIdx.x + (it >> 1) * T) + (it & 1));
    }
    typename LoopPreOf<Op, LOOPPRE>::type lepi[NTT_R];
    if constexpr (LOOPPRE) {
#pragma unroll
      for (int it = 0; it < NTT_R; it++) lepi[it] = Op::loop_prefetch(cx, jb, pm, gbase + 2 * (threadIdx.x + (it >> 1) * T) + (it & 1));
    }
    if constexpr (RAW) {
      if (more) {
#pragma unroll
        for (int it = 0; it < NTT_R; it++) raw[it] = Op::raw_load(cx, jn, pm, gbase + 2 * (threadIdx.x + (it >> 1) * T) + (it & 1));
      }
    } else {
      if (more) load_tile(jn, dreg);
    }
    __syncthreads();
    // STRIDED = true selects local-heap node indexing (the LDS copy); LASTFOLD = false: N^-1 belongs to the strided pass
    if constexpr (INVERSE) RoundSeq<P, LR, 0, true, true, true, false>::run(lds + sub * SP, tid, 0, 0, twl + (sub << P), pm);
    else forward_rounds<P, LR, true, true>(lds + sub * SP, tid, 0, 0, twl + (sub << P), pm);
    __syncthreads();
#pragma unroll
    for (int it = 0; it < NPAIR; it++) {
      const int idx = 2 * (threadIdx.x + it * T);
      const int sb = idx >> P, e = idx & (S - 1);
      const uint32_t n = gbase + idx;
      const u64 vx = lds[sb * SP + lds_pad<P>(e)], vy = lds[sb * SP + lds_pad<P>(e + 1)];
      if constexpr (INVERSE) {
        ulonglong2 v;
        v.x = vx;
        v.y = vy;
        *reinterpret_cast<ulonglong2 *>(jb.dst + n) = v; // lazy intermediate of the inverse transform
      } else if constexpr (PREFETCH) {
        Op::store_fwd_pre(cx, jb, pm, n, vx, epi[2 * it]);
        Op::store_fwd_pre(cx, jb, pm, n + 1, vy, epi[2 * it + 1]);
      } else if constexpr (LOOPPRE) {
        Op::store_fwd_loop(cx, jb, pm, n, vx, lepi[2 * it]);
        Op::store_fwd_loop(cx, jb, pm, n + 1, vy, lepi[2 * it + 1]);
      } else {
        Op::store_fwd(cx, jb, pm, n, vx);
        Op::store_fwd(cx, jb, pm, n + 1, vy);
      }
    }
    if (!more) break;
    if constexpr (RAW) { // the next job's input transform, on operands that arrived during this job's butterflies
#pragma unroll
      for (int it = 0; it < NPAIR; it++) {
        const uint32_t n = gbase + 2 * (threadIdx.x + it * T);
        dreg[it].x = Op::finish_load(cx, jn, pm, n, raw[2 * it]);
        dreg[it].y = Op::finish_load(cx, jn, pm, n + 1, raw[2 * it + 1]);
      }
    }
    jb = jn;
    j = jnext;
  }
}

// Inverse strided pass + forward strided pass in one launch, for the latency-bound (small) launches:
// a mod-down / rescale / digit conversion starts from the inverse transform of ONE source limb and
// continues with forward transforms of that polynomial under other primes.  The second (strided)
// pass of the inverse transform and the first (strided) pass of the forward transforms work on the
// same column tiles, so a workgroup finishes the inverse transform of its tile (from the
// intermediate the contiguous inverse pass left in Op::pre_src), applies the op's conversion in LDS
// and goes straight on with the forward stages under its own prime: one launch and one HBM round
// trip fewer per operation.  The inverse tile is recomputed by every job that shares the source
// limb, so the host uses this form only while the launch is far from filling the chip
// (fuse_small_launch in launch.hip.h); results are the same canonical residues either way.
template <int P, int LR, class Op>
__global__ void __launch_bounds__(NTT_THREADS)
ntt_inv_fwd_kernel(DevCtx cx, typename Op::Params prm, int log_tiles) {
  extern __shared__ __attribute__((aligned(16))) u64 lds[];
  constexpr int NTT_R = 1 << LR;
  constexpr int S = 1 << P, TPS = S / NTT_R, SP = lds_sub_stride<P>();
  if (cx.skipped()) return;
  const uint32_t tile_idx = blockIdx.x & ((1u << log_tiles) - 1u);
  typename Op::Job jb;
  if (!Op::setup(cx, prm, blockIdx.x >> log_tiles, blockIdx.y, blockIdx.z, jb)) return; // block-uniform
  const uint32_t pa = Op::pre_prime(prm, jb);
  const DevPrime pmA = cx.primes[pa], pm = cx.primes[jb.prime];
  const ulonglong2 *twA = cx.tw_inv + (size_t)pa * cx.N, *tw = cx.tw_fwd + (size_t)jb.prime * cx.N;
  constexpr int logC = 8 + LR - P, C = 1 << logC, T = NTT_THREADS;
  const uint32_t stride_log = cx.logN - P;
  constexpr int ES = 1 << (P - LR);
  constexpr bool LINEAR = (ES % 16 == 0);
  const int c = threadIdx.x & (C - 1), e0 = threadIdx.x >> logC;
  const uint32_t n0 = (tile_idx << logC) + ((uint32_t)e0 << stride_log) + c, nstep = (uint32_t)(T >> logC) << stride_log;
  const int l0 = c * SP + lds_pad<P>(e0);
  auto lds_at = [&](int it) -> int {
    if constexpr (LINEAR) return l0 + it * lds_pad<P>(ES);
    const int idx = threadIdx.x + it * T;
    return (idx & (C - 1)) * SP + lds_pad<P>(idx >> logC);
  };
  const u64 *src = Op::pre_src(jb);
#pragma unroll
  for (int it = 0; it < NTT_R; it++) lds[lds_at(it)] = src[n0 + it * nstep];
  ulonglong2 *twlA = reinterpret_cast<ulonglong2 *>(lds + ((C * SP + 1) & ~1)), *twl = twlA + S;
  for (int idx = threadIdx.x; idx < S; idx += T) { twlA[idx] = twA[idx]; twl[idx] = tw[idx]; }
  __syncthreads();
  const int sub = threadIdx.x / TPS, tid = threadIdx.x % TPS;
  RoundSeq<P, LR, 0, true, true, false>::run(lds + sub * SP, tid, 0, 0, twlA, pmA); // canonical mod q_a (N^-1 folded in)
  __syncthreads();
  auto convert = [&](auto lazy_tag) {
    constexpr bool LZ = decltype(lazy_tag)::value;
#pragma unroll
    for (int it = 0; it < NTT_R; it++) {
      u64 v = lds[lds_at(it)];
      if (Op::pre_addhalf) v = addmod(v, pmA.q >> 1, pmA.q);
      lds[lds_at(it)] = Op::template conv<LZ>(jb, pm, v);
    }
  };
  if (jb.lazy) convert(std::true_type{});
  else convert(std::false_type{});
  __syncthreads();
  forward_rounds<P, LR, true, false>(lds + sub * SP, tid, 0, 0, twl, pm);
  __syncthreads();
#pragma unroll
  for (int it = 0; it < NTT_R; it++) jb.dst[n0 + it * nstep] = lds[lds_at(it)]; // lazy intermediate of the forward transform
}

} // namespace evah

// the rest of what used to be this one file (r5: split by subject, same text, same order of definitions)
#include "ntt_window_sum.hip.h" // moddown_sum_kernel: the mod-down's second pass with a convolution window's sums as its epilogue
#include "ntt_ks_inner.hip.h"   // ks_inner_kernel: second pass of the digit transforms fused with the key inner product; operand tables
#include "ntt_ops.hip.h"        // the fused load / store ops of the passes (OpPlainT, OpMulIntt, OpKsDigit, OpModDownT, OpRR*)
#include "ntt_chain.hip.h"      // the chain step Mul -> Rescale -> Relinearize in six launches: ntt_inv2_kernel and its ops
