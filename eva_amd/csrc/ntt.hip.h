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
#pragma unroll
    for (int it = 0; it < NTT_R; it++) {
      const uint32_t n = n0 + it * nstep;
      lds[lds_at(it)] = FIRST ? Op::template load<LZ>(cx, jb, pm, n) : jb.dst[n];
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

// Key-switch inner product fused with the second (contiguous) pass of the digit NTTs
// (SURVEY.md A.6 step 2).  One workgroup owns output limb I = blockIdx.y and one tile of
// coefficient positions; it walks the digits J, finishing NTT_kappa(t_J) for its tile in LDS
// (or taking target[J] as is when I == J) and multiply-accumulating with key[J][0/1][kappa] into
// 128-bit register accumulators.  The l^2 N converted digits are therefore never written back:
// HBM sees the pass-1 intermediates once, the key once and prod[2][l+1][N] once.
// Keys of a batch of key-switches issued as one launch (sibling rotations of one ciphertext).
// KS_BATCH_MAX (instances per batched launch): internal.hip.h / defined below when this header is used alone
#ifndef EVAH_KS_BATCH_MAX_DEFINED
#define EVAH_KS_BATCH_MAX_DEFINED
constexpr int KS_BATCH_MAX = 64;
#endif
struct KsKeys {
  const u64 *key[KS_BATCH_MAX];
  uint32_t rows; // 0: whole keys (k prime rows); else a limb shard's rows (KeyDev::rows), indexed by local limb
};
// Base pointers of a batch of separately allocated polynomials (2 per instance), passed by value.
struct PtrTab {
  const u64 *p[2 * KS_BATCH_MAX];
};

// Operands of a batch of ciphertext products (instance b = a[b] x b[b], both size 2), passed by
// value.  With it the consumers of a product's polynomials d0 = a0 b0, d1 = a0 b1 + a1 b0,
// d2 = a1 b1 evaluate the one they need from the operands where they would have
// loaded it, so multiply -> relinearize -> rescale runs without the size-3 product ever being
// written to or read back from HBM.  Every d_K is a canonical residue, so the result is the one
// the separate evah_multiply call stores.
struct MulTab {
  const u64 *a[KS_BATCH_MAX], *b[KS_BATCH_MAX];
  uint32_t a_ps[KS_BATCH_MAX], b_ps[KS_BATCH_MAX]; // poly strides in units of N coefficients
};
// One product's operands, resolved from a MulTab entry (block-uniform: scalar loads of the kernel argument)
struct MulSrc {
  const u64 *a, *b; // polynomial 0 of each operand
  size_t sa, sb;    // poly strides in words
};
__device__ __forceinline__ MulSrc mul_src(const MulTab &t, uint32_t N, uint32_t inst) {
  return MulSrc{t.a[inst], t.b[inst], (size_t)t.a_ps[inst] * N, (size_t)t.b_ps[inst] * N};
}
// d_K of a product at word `off` (= limb * N + n) of a polynomial, K in {0, 1, 2}
__device__ __forceinline__ u64 product_poly(const MulSrc &m, uint32_t K, size_t off, const DevPrime &pm) {
  const u64 *a0 = m.a + off, *b0 = m.b + off;
  const size_t sa = m.sa, sb = m.sb;
  if (K == 0) return mulmod(a0[0], b0[0], pm);
  if (K == 2) return mulmod(a0[sa], b0[sb], pm);
  u128_t s = mul128(a0[0], b0[sb]);
  acc128(s, a0[sa], b0[0]);
  return barrett128(s, pm);
}

struct NoMul {}; // placeholder for the operand table in the variants that read a stored product

// Where the key-switch target and the polynomials the result is added to come from (MODE):
//   KS_PLAIN   target from memory, nothing added
//   KS_MUL     fused multiply, r03 form: the target d2 = a1 b1 of product `inst` is evaluated where the NTT-form
//              digit is used as is (I == J); d0, d1 are left to the combine epilogue of the mod-down
//   KS_FOLDMUL fused multiply, r04 form: the target is read from memory (OpMulIntt stores d2 next to its inverse
//              transform) and P * d0, P * d1 (P = the special prime) are added to the inner products of the data
//              limbs right here:  prod'[K][I] = prod[K][I] + P d_K[I]  mod q_I.  The mod-down computes
//              (prod' - U) P^-1 = d_K + (prod - U) P^-1, the same canonical residue, without reading the operands
//              again — its combine pass was waiting for those bytes (6 words per output word) while this
//              kernel, which is bound by integer issue, has the memory slack to fetch them.  The products are
//              128-bit MACs of a canonical operand with a lazy Shoup product (< 4q) of the other operand and P.
//   KS_FOLDADD the same for stored polynomials (relinearize, relinearize + rescale of a size-3 ciphertext; a
//              rotation's permuted c0): P * c_K is added, c_K = adds.p[2 inst + K] (limb 0; null = nothing to add),
//              so the mod-down's combine pass no longer reads c_K.
// (plain ints, not an unnamed enum: the enum's type would be mangled into the kernel's name as a local type and the
// runtime could not find the symbol)
constexpr int KS_PLAIN = 0, KS_MUL = 1, KS_FOLDMUL = 2, KS_FOLDADD = 3;
// kernel-argument types per mode (a traits struct, so the kernel's mangled name carries no constant expression)
template <int MODE> struct KsArgs { using Mul = NoMul; using Add = NoMul; };
template <> struct KsArgs<KS_MUL> { using Mul = MulTab; using Add = NoMul; };
template <> struct KsArgs<KS_FOLDMUL> { using Mul = MulTab; using Add = NoMul; };
template <> struct KsArgs<KS_FOLDADD> { using Mul = NoMul; using Add = PtrTab; };
template <int MODE> using KsMulArg = typename KsArgs<MODE>::Mul;
template <int MODE> using KsAddArg = typename KsArgs<MODE>::Add;

// INVSP (latency-bound launches): the workgroups of the special-prime row (I == l) go straight on
// with the contiguous pass of that row's inverse transform — the first step of the mod-down that
// always follows — on the tile they hold, and store its lazy intermediate to r_out[2 inst + K]
// instead of the row itself: one launch fewer per key switch, same residues.
// MAC3 (contexts whose primes all have the top-bit shape, whole keys in the split layout KeyDev::d_split): the inner
// product accumulates in radix 2^30.  The transformed digit is brought to < 2^60 + 2^36 with the top-bit reduction and
// cut at bit 30 (v0, v1), the key word arrives as (k0 | k1 << 32) with k0, k1 < 2^30, and the four partial products
// — each < 2^60.1 — go into three 64-bit sums A0 += v0 k0, A1 += v0 k1 + v1 k0, A2 += v1 k1 with ONE v_mad_u64_u32
// each and no carry handling: 13 VALU per coefficient and digit for both key polynomials (3 reduce + 2 split + 8 mad)
// where the 128-bit accumulation takes 28.  A1 holds 7 digits (14 products < 16 x 2^60), so the sums are normalised
// (carry words moved up) every 7 digits; after the loop they are recombined into the 128-bit accumulators the
// epilogue works on.  6 registers per accumulator instead of 4.
template <int P, int LR, int MAXT, int MODE, bool INVSP = false, bool MAC3 = false>
__global__ void __launch_bounds__(MAXT)
ks_inner_kernel(DevCtx cx, const u64 *__restrict__ target_b, size_t target_bs, const u64 *__restrict__ scratch_b,
                size_t scratch_bs, KsKeys keys, u64 *__restrict__ prod_b, size_t prod_bs, uint32_t l, uint32_t i0,
                int logC, uint32_t n_tiles, uint32_t n_inst, PtrTab targets, KsMulArg<MODE> mul, uint32_t istep,
                uint32_t nout, u64 *__restrict__ r_out, KsAddArg<MODE> adds, int lazy_out) {
  constexpr bool MUL = MODE == KS_MUL;
  extern __shared__ __attribute__((aligned(16))) u64 lds[];
  if (cx.skipped()) return;
  // grid.x carries (tile, instance): instances of one tile are placed 8 block ids apart, i.e. on
  // the same XCD (blocks are dealt round-robin over the 8 XCDs) and close in dispatch order, so
  // instances that share a key find its tile in that XCD's L2.  Speed only, never correctness.
  uint32_t tile_idx, inst;
  if ((n_tiles & 7u) == 0) {
    const uint32_t x = blockIdx.x, lo = x & 7u, rest = x >> 3;
    inst = rest % n_inst;
    tile_idx = (rest / n_inst) * 8u + lo;
  } else {
    inst = blockIdx.x / n_tiles;
    tile_idx = blockIdx.x % n_tiles;
  }
  // the divisions above go through the vector unit: make the results wave-uniform for the compiler, so that what is
  // indexed by them (key, operand and product base pointers) lives in SGPRs
  inst = __builtin_amdgcn_readfirstlane(inst);
  tile_idx = __builtin_amdgcn_readfirstlane(tile_idx);
  // MUL: the key-switch target is d2 = a1 b1 of product `inst`, evaluated where it is needed (I == J)
  const u64 *__restrict__ target = MUL ? nullptr : (target_b ? target_b + inst * target_bs : targets.p[inst]);
  const u64 *__restrict__ scratch = scratch_b + inst * scratch_bs;
  const u64 *__restrict__ key = keys.key[inst];
  u64 *__restrict__ prod = prod_b + inst * prod_bs;
  constexpr int NTT_R = 1 << LR, NPAIR = NTT_R / 2;
  constexpr int S = 1 << P, TPS = S / NTT_R, SP = lds_sub_stride<P>();
  // output limb of this workgroup: I = i0 + blockIdx.y * istep.  istep == 1: the rows of scratch /
  // prod / target are indexed by I itself (nout = l + 1); istep == G (a shard's limbs): by blockIdx.y
  const uint32_t I = i0 + blockIdx.y * istep;
  const uint32_t Irow = istep > 1 ? blockIdx.y : I;
  const uint32_t kap = (I == l) ? cx.k - 1 : I;
  const DevPrime pm = cx.primes[kap];
  const ulonglong2 *tw = cx.tw_fwd + (size_t)kap * cx.N;
  const int T = blockDim.x;
  const uint32_t pre = cx.logN - P;
  const uint32_t sub0 = tile_idx << logC, gbase = sub0 << P;
  // key rows: by prime for a whole key; by local limb (last row: the special prime) for a shard's rows
  const uint32_t krows = keys.rows ? keys.rows : cx.k;
  const uint32_t krow = keys.rows ? (kap == cx.k - 1 ? krows - 1 : blockIdx.y) : kap;
  const size_t N = cx.N, key_digit = (size_t)2 * krows * N;
  const int sub = threadIdx.x / TPS, tid = threadIdx.x % TPS;

  // The twiddles of this tile's sub-transforms are the same for every digit J: stage them in LDS
  // once as per-sub local heaps (node n of sub s = global node ((2^pre + h_s) << depth(n)) + pos(n)),
  // so the J loop touches global memory only for coefficients and key.
  const int C = 1 << logC;
  ulonglong2 *twl = reinterpret_cast<ulonglong2 *>(lds + ((C * SP + 1) & ~1));
  for (int idx = threadIdx.x; idx < (C << P); idx += T) {
    const int sb = idx >> P, n = idx & (S - 1);
    if (n) {
      const int d = 31 - __clz(n);
      twl[idx] = tw[((size_t)((1u << pre) + sub0 + sb) << d) + (n - (1 << d))];
    }
  }

  u128_t acc0[NTT_R], acc1[NTT_R];
#pragma unroll
  for (int i = 0; i < NTT_R; i++) { acc0[i] = {0, 0}; acc1[i] = {0, 0}; }
  constexpr int NA = MAC3 ? NTT_R : 1;
  u64 s0[2][NA], s1[2][NA], s2[2][NA]; // MAC3: radix-2^30 partial sums per key polynomial K
#pragma unroll
  for (int i = 0; i < NA; i++) { s0[0][i] = s1[0][i] = s2[0][i] = s0[1][i] = s1[1][i] = s2[1][i] = 0; }

  // Software pipeline over the digits: the key words of digit J are requested before its
  // transform starts and the coefficients of digit J+1 as soon as those of J sit in LDS, so both
  // streams are in flight during the register rounds instead of being waited for at their use.
  MulSrc msrc{nullptr, nullptr, 0, 0};
  if constexpr (MUL || MODE == KS_FOLDMUL) msrc = mul_src(mul, cx.N, inst);
  auto load_digits = [&](uint32_t J, ulonglong2 *d) {
    if (MUL && I == J) { // block-uniform
#pragma unroll
      for (int it = 0; it < NPAIR; it++) {
        const size_t off = (size_t)J * N + gbase + 2 * (threadIdx.x + it * T);
        d[it].x = product_poly(msrc, 2, off, pm);
        d[it].y = product_poly(msrc, 2, off + 1, pm);
      }
    } else {
      const u64 *src = (I == J ? target + (size_t)Irow * N : scratch + ((size_t)Irow * l + J) * N) + gbase;
#pragma unroll
      for (int it = 0; it < NPAIR; it++) d[it] = *reinterpret_cast<const ulonglong2 *>(src + 2 * (threadIdx.x + it * T));
    }
  };
  // MAC3: the digit tiles do not pass through registers.  An LDS-DMA load (global_load_lds_dwordx4: 16 bytes per lane,
  // contiguous) puts tile J + 1 into an unpadded buffer `lin` while tile J is being transformed; the first register
  // round reads its inputs from `lin`, and the load of the next tile is issued right after that round.  This frees
  // the 8 prefetch registers (and the tile's ds_write) for the radix-2^30 sums.
  u64 *lin = reinterpret_cast<u64 *>(twl + (C << P)); // [256] after the twiddle heaps
  auto dma_digits = [&](uint32_t J) {
    const u64 *src = (I == J ? target + (size_t)Irow * N : scratch + ((size_t)Irow * l + J) * N) + gbase;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // reads of `lin` issued so far have returned
#pragma unroll
    for (int it = 0; it < NPAIR; it++)
      __builtin_amdgcn_global_load_lds(src + 2 * (threadIdx.x + it * T), lin + 2 * it * T, 16, 0, EVAH_DIGIT_AUX);
  };
  ulonglong2 dreg[MAC3 ? 1 : NPAIR];
  if constexpr (MAC3) dma_digits(0);
  else load_digits(0, dreg);
  uint32_t since_fold = 0;
  for (uint32_t J = 0; J < l; J++) {
    ulonglong2 k0r[NPAIR], k1r[NPAIR];
    if constexpr (!MAC3) {
      const u64 *kp = key + J * key_digit + (size_t)krow * N + gbase;
#pragma unroll
      for (int it = 0; it < NPAIR; it++) {
        const int idx = 2 * (threadIdx.x + it * T);
        k0r[it] = *reinterpret_cast<const ulonglong2 *>(kp + idx);
        k1r[it] = *reinterpret_cast<const ulonglong2 *>(kp + (size_t)krows * N + idx);
      }
    }
    u64 val[NTT_R];
    const uint32_t Jn = J + 1 < l ? J + 1 : J;
    if constexpr (MAC3) {
      using RS = Rounds<P, LR>;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // tile J has landed in `lin` (its load was issued a digit ago)
      // the key words of this digit: requested now, used after the transform
      const __amdgpu_buffer_rsrc_t krs = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<u64 *>(key + (size_t)krow * N + gbase), 0, 0x7fffffff, 0x00020000);
      const uint32_t voff = 16u * threadIdx.x;
      const uint32_t soff0 = (uint32_t)(J * key_digit * sizeof(u64)), soff1 = soff0 + (uint32_t)((size_t)krows * N * sizeof(u64));
#pragma unroll
      for (int it = 0; it < NPAIR; it++) {
        typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
        const u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(krs, voff + 16u * (uint32_t)(it * T), soff0, 0);
        const u32x4 b = __builtin_amdgcn_raw_buffer_load_b128(krs, voff + 16u * (uint32_t)(it * T), soff1, 0);
        k0r[it].x = ((u64)a.y << 32) | a.x;
        k0r[it].y = ((u64)a.w << 32) | a.z;
        k1r[it].x = ((u64)b.y << 32) | b.x;
        k1r[it].y = ((u64)b.w << 32) | b.z;
      }
      if (I == J) { // NTT form already: the tile as it lies in `lin`
#pragma unroll
        for (int it = 0; it < NPAIR; it++) {
          const ulonglong2 v = *reinterpret_cast<const ulonglong2 *>(lin + 2 * (threadIdx.x + it * T));
          val[2 * it] = v.x;
          val[2 * it + 1] = v.y;
        }
        if (J + 1 < l) dma_digits(J + 1);
      } else {
        __builtin_amdgcn_wave_barrier(); // (one wave: DS operations are in order; no s_barrier, no vmcnt drain)
        // first round: from `lin` into the padded tile; then the next tile's load; then the remaining rounds
        auto next_tile = [&]() { if (J + 1 < l) dma_digits(J + 1); }; // (waits for the round's reads of `lin` first)
        ntt_round<P, LR, RS::bits(0), RS::lo(0), false, true, true, true, true>(lds + sub * SP, tid, 0, 0, twl + (sub << P), pm,
                                                                              lin + sub * S, next_tile);
        if constexpr (RS::NR > 1) {
          __builtin_amdgcn_wave_barrier();
          RoundSeq<P, LR, 1, false, true, true, true, true, true>::run(lds + sub * SP, tid, 0, 0, twl + (sub << P), pm);
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int it = 0; it < NPAIR; it++) {
          const int idx = 2 * (threadIdx.x + it * T);
          const int sb = idx >> P, e = idx & (S - 1);
          val[2 * it] = lds[sb * SP + lds_pad<P>(e)];
          val[2 * it + 1] = lds[sb * SP + lds_pad<P>(e + 1)];
        }
      }
    } else if (I == J) { // already in NTT form mod q_J: use the key-switch target directly
#pragma unroll
      for (int it = 0; it < NPAIR; it++) {
        val[2 * it] = dreg[it].x;
        val[2 * it + 1] = dreg[it].y;
      }
      load_digits(Jn, dreg);
    } else {
      __syncthreads(); // previous iteration's LDS reads are done
#pragma unroll
      for (int it = 0; it < NPAIR; it++) {
        const int idx = 2 * (threadIdx.x + it * T);
        const int sb = idx >> P, e = idx & (S - 1);
        lds[sb * SP + lds_pad<P>(e)] = dreg[it].x;
        lds[sb * SP + lds_pad<P>(e + 1)] = dreg[it].y;
      }
      load_digits(Jn, dreg);
      __syncthreads();
      // STRIDED=true selects local-heap node indexing, which is what the LDS copy uses
      forward_rounds<P, LR, true, true>(lds + sub * SP, tid, 0, 0, twl + (sub << P), pm);
      __syncthreads();
#pragma unroll
      for (int it = 0; it < NPAIR; it++) {
        const int idx = 2 * (threadIdx.x + it * T);
        const int sb = idx >> P, e = idx & (S - 1);
        val[2 * it] = lds[sb * SP + lds_pad<P>(e)];       // lazy [0,16q): fine for the 128-bit MAC (folded every 16 digits)
        val[2 * it + 1] = lds[sb * SP + lds_pad<P>(e + 1)];
      }
    }
    if constexpr (MAC3) {
#pragma unroll
      for (int i = 0; i < NTT_R; i++) {
        const uint32_t hi = (uint32_t)(val[i] >> 32);
        const u64 v = mad64(hi >> pm.tb_sh, pm.tb_c, ((u64)(hi & pm.tb_mask) << 32) | (uint32_t)val[i]); // < 2^60 + 2^36
        const uint32_t v0 = (uint32_t)v & 0x3fffffffu, v1 = (uint32_t)(v >> 30);
        const u64 kw0 = (i & 1) ? k0r[i >> 1].y : k0r[i >> 1].x, kw1 = (i & 1) ? k1r[i >> 1].y : k1r[i >> 1].x;
        s0[0][i] = mad64(v0, (uint32_t)kw0, s0[0][i]);
        s1[0][i] = mad64(v0, (uint32_t)(kw0 >> 32), s1[0][i]);
        s1[0][i] = mad64(v1, (uint32_t)kw0, s1[0][i]);
        s2[0][i] = mad64(v1, (uint32_t)(kw0 >> 32), s2[0][i]);
        s0[1][i] = mad64(v0, (uint32_t)kw1, s0[1][i]);
        s1[1][i] = mad64(v0, (uint32_t)(kw1 >> 32), s1[1][i]);
        s1[1][i] = mad64(v1, (uint32_t)kw1, s1[1][i]);
        s2[1][i] = mad64(v1, (uint32_t)(kw1 >> 32), s2[1][i]);
      }
      if (++since_fold == 7u && J + 1 < l) { // block-uniform: carry words up, 7 more digits fit
        since_fold = 0;
#pragma unroll
        for (int K = 0; K < 2; K++)
#pragma unroll
          for (int i = 0; i < NTT_R; i++) {
            s1[K][i] += s0[K][i] >> 30;
            s0[K][i] &= 0x3fffffffull;
            s2[K][i] += s1[K][i] >> 30;
            s1[K][i] &= 0x3fffffffull;
          }
      }
      continue;
    }
#pragma unroll
    for (int it = 0; it < NPAIR; it++) {
      acc128(acc0[2 * it], val[2 * it], k0r[it].x);
      acc128(acc0[2 * it + 1], val[2 * it + 1], k0r[it].y);
      acc128(acc1[2 * it], val[2 * it], k1r[it].x);
      acc128(acc1[2 * it + 1], val[2 * it + 1], k1r[it].y);
    }
    // 16 lazy products (each < 16q * q < 2^124) fill the 128-bit accumulators, 15 of them and a folded word leave
    // room for the P * d_K terms added after the loop (< 12 q^2 together): with more digits than that, fold the
    // accumulators back to one word every 15 (block-uniform, only ever taken when l > 15)
    if (++since_fold == 15u && J + 1 < l) {
      since_fold = 0;
#pragma unroll
      for (int i = 0; i < NTT_R; i++) {
        acc0[i] = {barrett128(acc0[i], pm), 0};
        acc1[i] = {barrett128(acc1[i], pm), 0};
      }
    }
  }
  if constexpr (MAC3) { // S = s0 + s1 2^30 + s2 2^60 < 2^125: the 128-bit accumulators of the epilogue
#pragma unroll
    for (int i = 0; i < NTT_R; i++) {
      unsigned __int128 a = s0[0][i], b = s0[1][i];
      a += (unsigned __int128)s1[0][i] << 30;
      a += (unsigned __int128)s2[0][i] << 60;
      b += (unsigned __int128)s1[1][i] << 30;
      b += (unsigned __int128)s2[1][i] << 60;
      acc0[i] = {(u64)a, (u64)(a >> 64)};
      acc1[i] = {(u64)b, (u64)(b >> 64)};
    }
  }
  if constexpr (MODE == KS_FOLDMUL || MODE == KS_FOLDADD) {
    { // after the digit loop, where its prefetch registers are free (as a prologue the block cost 32 VGPRs: 147, 3 waves
      // per SIMD).  No branch: the special row multiplies by modq[P][P] = (0, 0) — P = 0 mod P — and reads a row that exists
      const ulonglong2 Pm = cx.modq[(size_t)(cx.k - 1) * cx.k + kap]; // (P mod q_I, Shoup quotient)
      const size_t off = (size_t)(Irow < l ? Irow : l - 1) * N + gbase;
#pragma unroll
      for (int it = 0; it < NPAIR; it++) {
        const size_t o = off + 2 * (threadIdx.x + it * T);
        if constexpr (MODE == KS_FOLDMUL) {
          const ulonglong2 a0 = *reinterpret_cast<const ulonglong2 *>(msrc.a + o);
          const ulonglong2 a1 = *reinterpret_cast<const ulonglong2 *>(msrc.a + msrc.sa + o);
          const ulonglong2 b0 = *reinterpret_cast<const ulonglong2 *>(msrc.b + o);
          const ulonglong2 b1 = *reinterpret_cast<const ulonglong2 *>(msrc.b + msrc.sb + o);
          // a < q, lazy(b P) < 4q: each term < 2^122, three of them on top of 15 digit products still fit 128 bits
          const u64 u0x = mul_tw_lazy5(b0.x, Pm.x, Pm.y, pm.nq), u0y = mul_tw_lazy5(b0.y, Pm.x, Pm.y, pm.nq);
          const u64 u1x = mul_tw_lazy5(b1.x, Pm.x, Pm.y, pm.nq), u1y = mul_tw_lazy5(b1.y, Pm.x, Pm.y, pm.nq);
          acc128(acc0[2 * it], a0.x, u0x);
          acc128(acc0[2 * it + 1], a0.y, u0y);
          acc128(acc1[2 * it], a0.x, u1x);
          acc128(acc1[2 * it + 1], a0.y, u1y);
          acc128(acc1[2 * it], a1.x, u0x);
          acc128(acc1[2 * it + 1], a1.y, u0y);
        } else {
          // a null entry: nothing is added to that polynomial (a rotation adds the permuted c0 to K = 0 only)
          const u64 *p0 = adds.p[2 * inst], *p1 = adds.p[2 * inst + 1]; // block-uniform
          if (p0) {
            const ulonglong2 c0 = *reinterpret_cast<const ulonglong2 *>(p0 + o);
            acc128(acc0[2 * it], c0.x, Pm.x);
            acc128(acc0[2 * it + 1], c0.y, Pm.x);
          }
          if (p1) {
            const ulonglong2 c1 = *reinterpret_cast<const ulonglong2 *>(p1 + o);
            acc128(acc1[2 * it], c1.x, Pm.x);
            acc128(acc1[2 * it + 1], c1.y, Pm.x);
          }
        }
      }
    }
  }
  if constexpr (INVSP) {
    if (I == l) { // block-uniform
      const ulonglong2 *twi = cx.tw_inv + (size_t)kap * cx.N;
#pragma unroll
      for (int K = 0; K < 2; K++) {
        __syncthreads(); // the tile in LDS has been consumed
#pragma unroll
        for (int it = 0; it < NPAIR; it++) {
          const int idx = 2 * (threadIdx.x + it * T);
          const int sb = idx >> P, e = idx & (S - 1);
          lds[sb * SP + lds_pad<P>(e)] = barrett128(K ? acc1[2 * it] : acc0[2 * it], pm);
          lds[sb * SP + lds_pad<P>(e + 1)] = barrett128(K ? acc1[2 * it + 1] : acc0[2 * it + 1], pm);
        }
        __syncthreads();
        // as ntt_pass_kernel<P, LR, contiguous, inverse>: global twiddle heap of the row's prime
        RoundSeq<P, LR, 0, true, false, true>::run(lds + sub * SP, tid, sub0 + sub, pre, twi, pm);
        __syncthreads();
        u64 *r = r_out + ((size_t)2 * inst + K) * N + gbase;
#pragma unroll
        for (int it = 0; it < NPAIR; it++) {
          const int idx = 2 * (threadIdx.x + it * T);
          const int sb = idx >> P, e = idx & (S - 1);
          ulonglong2 v;
          v.x = lds[sb * SP + lds_pad<P>(e)];
          v.y = lds[sb * SP + lds_pad<P>(e + 1)];
          *reinterpret_cast<ulonglong2 *>(r + idx) = v; // lazy intermediate of the inverse transform
        }
      }
      return;
    }
  }
  u64 *p0 = prod + (size_t)Irow * N + gbase, *p1 = prod + ((size_t)nout + Irow) * N + gbase;
  // lazy_out: the data rows go to combine passes that take any 64-bit representative (OpRRT / OpRRLastT multiply prod by
  // P^-1 first), so their last Barrett step is skipped; the special row feeds an inverse transform and stays canonical
  if (lazy_out && I != l) { // block-uniform
#pragma unroll
    for (int it = 0; it < NPAIR; it++) {
      const int idx = 2 * (threadIdx.x + it * T);
      ulonglong2 r0, r1;
      r0.x = reduce128_lazy(acc0[2 * it], pm);
      r0.y = reduce128_lazy(acc0[2 * it + 1], pm);
      r1.x = reduce128_lazy(acc1[2 * it], pm);
      r1.y = reduce128_lazy(acc1[2 * it + 1], pm);
      *reinterpret_cast<ulonglong2 *>(p0 + idx) = r0;
      *reinterpret_cast<ulonglong2 *>(p1 + idx) = r1;
    }
    return;
  }
#pragma unroll
  for (int it = 0; it < NPAIR; it++) {
    const int idx = 2 * (threadIdx.x + it * T);
    ulonglong2 r0, r1;
    r0.x = barrett128(acc0[2 * it], pm);
    r0.y = barrett128(acc0[2 * it + 1], pm);
    r1.x = barrett128(acc1[2 * it], pm);
    r1.y = barrett128(acc1[2 * it + 1], pm);
    *reinterpret_cast<ulonglong2 *>(p0 + idx) = r0;
    *reinterpret_cast<ulonglong2 *>(p1 + idx) = r1;
  }
}

// ------------------------------------------------------------------ fused load/store ops

// Plain batched transform over limbs.  job -> (poly p = job / jl, limb i = job % jl),
// prime = prime0 + i.  addhalf: x <- x + floor(q/2) mod q on store (rounding offset of
// rescale / key-switch mod-down).
// GATHER: polynomial pp is read through an index table, x[n] = src[perm_tab.p[pp >> 1][n]] (the special rows of hoisted
// key inner products, which rotation_sets.hip.h keeps in the source's index space: the Galois permutation is applied here)
struct NoGather {};
template <bool ZEROS, bool GATHER = false> struct OpPlainT { // ZEROS: the inverse transform also records zero coefficients
  struct Params {
    const u64 *src;
    u64 *dst;
    size_t src_ps, dst_ps; // poly strides (elements)
    uint32_t jl, prime0;
    int addhalf;
    PtrTab src_tab; // used when src == nullptr: polynomial pp starts at src_tab.p[pp]
    uint32_t pstep = 1; // limb i is modulo primes[prime0 + i * pstep] (limb-sharded values: the shard count)
    // inverse transforms of hoisted rotations: coefficients that come out 0 are counted in the low
    // word of zero_list[0] and the first HOIST_ZERO_CAP of them recorded as (poly << 48 | limb << 32 | index)
    u64 *zero_list = nullptr;
    std::conditional_t<GATHER, PermTab, NoGather> perm_tab{};
    // ZEROS: words the first pass clears before the second pass counts into zero_list (first_pass_clear)
    u64 *clear_base = nullptr;
    uint32_t clear_words = 0;
  };
  static constexpr bool clears_words = ZEROS;
  struct Job {
    uint32_t prime;
    const u64 *src;
    u64 *dst;
    int addhalf;
    bool lazy;
    u64 *zero_list;
    uint32_t pp;
    const uint32_t *perm;
  };
  // jobs = polys * jl: grid.y = limb i, grid.z = poly
  static dim3 grid(const Params &p, uint32_t jobs) { return dim3(1, p.jl, jobs / p.jl); }
  static constexpr int loop_axis = 2; // jobs that share a prime lie along grid.z (ntt_loop_kernel)
  static __device__ __forceinline__ bool setup(const DevCtx &cx, const Params &p, uint32_t, uint32_t i,
                                               uint32_t pp, Job &j) {
    j.prime = p.prime0 + i * p.pstep;
    j.src = (p.src ? p.src + pp * p.src_ps : p.src_tab.p[pp]) + (size_t)i * cx.N;
    j.dst = p.dst + pp * p.dst_ps + (size_t)i * cx.N;
    j.addhalf = p.addhalf;
    j.lazy = false;
    j.zero_list = p.zero_list;
    j.pp = pp;
    if constexpr (GATHER) j.perm = p.perm_tab.p[pp >> 1];
    else j.perm = nullptr;
    return true;
  }
  template <bool LZ>
  static __device__ __forceinline__ u64 load(const DevCtx &, const Job &j, const DevPrime &, uint32_t n) {
    if constexpr (GATHER) return j.src[j.perm[n]];
    return j.src[n];
  }
  static __device__ __forceinline__ void store(const DevCtx &, const Job &j, const DevPrime &pm,
                                               uint32_t n, u64 v) {
    if constexpr (ZEROS) {
      if (v == 0) {
        const uint32_t at = atomicAdd(reinterpret_cast<uint32_t *>(j.zero_list), 1u);
        if (at < HOIST_ZERO_CAP) j.zero_list[1 + at] = ((u64)j.pp << 48) | ((u64)j.prime << 32) | n;
      }
    }
    if (j.addhalf) v = addmod(v, pm.q >> 1, pm.q);
    j.dst[n] = v;
  }
  static __device__ __forceinline__ void store_fwd(const DevCtx &cx, const Job &j, const DevPrime &pm,
                                                   uint32_t n, u64 v) {
    store(cx, j, pm, n, barrett64(v, pm.q, pm.brt));
  }
};

using OpPlain = OpPlainT<false>;
using OpPlainZ = OpPlainT<true>;
using OpPlainG = OpPlainT<false, true>;

// Inverse transform of d2 = a1 b1 of a batch of products (the key-switch target of a fused
// multiply -> relinearize): job -> (instance b = job / jl, limb i = job % jl); the product is
// formed on load, so d2 itself never exists in memory.
struct OpMulIntt {
  struct Params {
    MulTab mul;
    u64 *dst;      // [batch][jl][N] coefficient-form digits
    size_t dst_ps; // batch stride
    uint32_t jl;
    u64 *d2 = nullptr; // != nullptr: d2 itself (NTT form) is stored too, [batch][jl][N] at the same stride — the
                       // key-switch kernel reads it where the digit is used as is (I == J) instead of forming it again
  };
  struct Job {
    uint32_t prime;
    size_t off;
    MulSrc mul;
    u64 *dst, *d2;
    bool lazy;
  };
  static dim3 grid(const Params &p, uint32_t jobs) { return dim3(1, p.jl, jobs / p.jl); }
  static constexpr int loop_axis = 2; // jobs that share a prime lie along grid.z (ntt_loop_kernel)
  static __device__ __forceinline__ bool setup(const DevCtx &cx, const Params &p, uint32_t, uint32_t i, uint32_t b, Job &j) {
    j.prime = cx.prime_of(i);
    j.off = (size_t)i * cx.N;
    j.mul = mul_src(p.mul, cx.N, b);
    j.dst = p.dst + b * p.dst_ps + (size_t)i * cx.N;
    j.d2 = p.d2 ? p.d2 + b * p.dst_ps + (size_t)i * cx.N : nullptr;
    j.lazy = false;
    return true;
  }
  template <bool LZ>
  static __device__ __forceinline__ u64 load(const DevCtx &cx, const Job &j, const DevPrime &pm, uint32_t n) {
    const u64 v = product_poly(j.mul, 2, j.off + n, pm);
    if (j.d2) j.d2[n] = v; // block-uniform
    return v;
  }
  static __device__ __forceinline__ void store(const DevCtx &, const Job &j, const DevPrime &, uint32_t n, u64 v) {
    j.dst[n] = v;
  }
  static __device__ __forceinline__ void store_fwd(const DevCtx &, const Job &, const DevPrime &, uint32_t, u64) {}
  // ntt_loop_kernel: operands of the NEXT product requested before the current job's transform, multiplied after it
  struct Raw { u64 a, b; };
  static __device__ __forceinline__ Raw raw_load(const DevCtx &, const Job &j, const DevPrime &, uint32_t n) {
    return Raw{j.mul.a[j.off + n + j.mul.sa], j.mul.b[j.off + n + j.mul.sb]};
  }
  static __device__ __forceinline__ u64 finish_load(const DevCtx &, const Job &j, const DevPrime &pm, uint32_t n, const Raw &r) {
    const u64 v = mulmod(r.a, r.b, pm);
    if (j.d2) j.d2[n] = v; // block-uniform
    return v;
  }
};

// Key-switch digit conversion: job -> (I = job / l, J = job % l);
// scratch[I][J] = NTT_{kappa(I)}( t[J] mod q_kappa(I) ), I == J skipped (NTT form reused).
struct OpKsDigit {
  struct Params {
    const u64 *t;   // [batch][l][N] coefficient-form digits
    u64 *scratch;   // [batch][l+1][l][N]
    uint32_t l;
    size_t t_bs, scratch_bs; // batch strides
    uint32_t i0, ni;         // output limbs handled by this launch: I = i0 + iy * istep, iy < ni (I == l: special prime)
    uint32_t istep = 1;      // 1: a slice of all limbs; G: the limbs a shard of G owns (scratch rows are then local: iy)
    uint32_t t_split = 1, t_rows = 0; // digit J sits at row (J % t_split) * t_rows + J / t_split of t (an all-gathered
                                      // buffer is shard-major); t_split == 1: row J
  };
  struct Job {
    uint32_t prime, digit;
    const u64 *src;
    u64 *dst;
    bool lazy;
  };
  // jobs = batch * ni * l: grid.x carries the digit J, grid.y the output limb, grid.z the batch
  static dim3 grid(const Params &p, uint32_t jobs) { return dim3(p.l, p.ni, jobs / (p.ni * p.l)); }
  static constexpr int loop_axis = 0; // the digits J of one output limb share its prime (ntt_loop_kernel)
  static __device__ __forceinline__ bool setup(const DevCtx &cx, const Params &p, uint32_t J, uint32_t iy,
                                               uint32_t b, Job &j) {
    const uint32_t I = p.i0 + iy * p.istep;
    if (I == J) return false;
    j.digit = J;
    j.prime = (I == p.l) ? cx.k - 1 : I;
    // t_J < q_J: when q_J <= 8 q_kappa the digit is already a valid lazy input (< 12 q_kappa)
    j.lazy = cx.primes[J].q <= cx.primes[j.prime].q8;
    const uint32_t row = p.t_split > 1 ? (J % p.t_split) * p.t_rows + J / p.t_split : J;
    j.src = p.t + b * p.t_bs + (size_t)row * cx.N;
    const uint32_t Irow = p.istep > 1 ? iy : I;
    j.dst = p.scratch + b * p.scratch_bs + ((size_t)Irow * p.l + J) * cx.N;
    return true;
  }
  template <bool LZ>
  static __device__ __forceinline__ u64 load(const DevCtx &, const Job &j, const DevPrime &pm, uint32_t n) {
    return conv<LZ>(j, pm, j.src[n]);
  }
  // hooks of ntt_inv_fwd_kernel: the source is digit J's contiguous-inverse-pass intermediate
  static constexpr bool pre_addhalf = false;
  static __device__ __forceinline__ uint32_t pre_prime(const Params &, const Job &j) { return j.digit; }
  static __device__ __forceinline__ const u64 *pre_src(const Job &j) { return j.src; }
  template <bool LZ> static __device__ __forceinline__ u64 conv(const Job &, const DevPrime &pm, u64 v) {
    return LZ ? v : barrett64(v, pm.q, pm.brt);
  }
  static __device__ __forceinline__ void store_fwd(const DevCtx &, const Job &j, const DevPrime &pm,
                                                   uint32_t n, u64 v) {
    j.dst[n] = barrett64(v, pm.q, pm.brt);
  }
};

// Divide-and-round by prime a (rescale: a = last data prime; key-switch: a = special prime).
// job -> (p = job / jl, i = job % jl).  r[p] is INTT(limb a) + floor(q_a/2) in coefficient form.
//   load : u = (r mod q_i) - (floor(q_a/2) mod q_i)
//   store: v = (c[p][i] - NTT(u)) * q_a^-1 mod q_i ;  dst = add ? add + v : v
// GATHER: c (the key inner products) is read through the pair's index table, c[perm_tab.p[pp >> 1][n]] (hoisted sets)
template <bool GATHER> struct OpModDownT {
  struct Params {
    const u64 *r;
    size_t r_ps;
    const u64 *c;
    size_t c_ps;
    const u64 *add; // nullable; applies to polys p < add_polys (add_polys == ~0u: even p only)
    size_t add_ps;
    uint32_t add_polys;
    u64 *dst;
    size_t dst_ps;
    uint32_t a, jl;
    size_t add_bs = 0; // != 0: poly pp = 2b + K adds add[b * add_bs + K * add_ps] (batched relinearize)
    // separately allocated operands (the *_many entry points): used when c == nullptr /
    // use_add_tab; entry pp is limb 0 of polynomial pp, a null add entry means "nothing to add"
    bool use_add_tab = false;
    PtrTab c_tab{}, add_tab{};
    std::conditional_t<GATHER, PermTab, NoGather> perm_tab{};
  };
  struct Job {
    uint32_t prime;
    const u64 *src, *c, *add;
    u64 *dst;
    u64 halfm;
    ulonglong2 inv;
    bool lazy;
    const uint32_t *perm;
  };
  static dim3 grid(const Params &p, uint32_t jobs) { return dim3(1, p.jl, jobs / p.jl); }
  static constexpr int loop_axis = 2; // jobs that share a prime lie along grid.z (ntt_loop_kernel)
  static __device__ __forceinline__ bool setup(const DevCtx &cx, const Params &p, uint32_t, uint32_t i,
                                               uint32_t pp, Job &j) {
    j.prime = cx.prime_of(i); // limb i of the values (c, add, dst) — the prime itself on an ordinary context
    j.src = p.r + pp * p.r_ps;
    j.c = (p.c ? p.c + pp * p.c_ps : p.c_tab.p[pp]) + (size_t)i * cx.N;
    if (p.use_add_tab) {
      j.add = p.add_tab.p[pp] ? p.add_tab.p[pp] + (size_t)i * cx.N : nullptr;
    } else {
      const bool use_add = p.add && (p.add_bs ? true : p.add_polys == ~0u ? (pp & 1u) == 0 : pp < p.add_polys);
      const size_t add_off = p.add_bs ? (pp >> 1) * p.add_bs + (pp & 1u) * p.add_ps : pp * p.add_ps;
      j.add = use_add ? p.add + add_off + (size_t)i * cx.N : nullptr;
    }
    j.dst = p.dst + pp * p.dst_ps + (size_t)i * cx.N;
    j.halfm = cx.halfmod[p.a * cx.k + j.prime];
    j.inv = cx.invq[p.a * cx.k + j.prime];
    j.lazy = cx.primes[p.a].q <= cx.primes[j.prime].q8; // r < q_a: r + (q_i - halfm) < 9 q_i
    if constexpr (GATHER) j.perm = p.perm_tab.p[pp >> 1];
    else j.perm = nullptr;
    return true;
  }
  static __device__ __forceinline__ u64 c_at(const Job &j, uint32_t n) {
    if constexpr (GATHER) return j.c[j.perm[n]];
    return j.c[n];
  }
  template <bool LZ>
  static __device__ __forceinline__ u64 load(const DevCtx &, const Job &j, const DevPrime &pm, uint32_t n) {
    return conv<LZ>(j, pm, j.src[n]);
  }
  // hooks of ntt_inv_fwd_kernel: r holds the contiguous-inverse-pass intermediate of limb a
  static constexpr bool pre_addhalf = true;
  static __device__ __forceinline__ uint32_t pre_prime(const Params &p, const Job &) { return p.a; }
  static __device__ __forceinline__ const u64 *pre_src(const Job &j) { return j.src; }
  template <bool LZ> static __device__ __forceinline__ u64 conv(const Job &j, const DevPrime &pm, u64 v) {
    if (LZ) return v + (pm.q - j.halfm);
    return submod(barrett64(v, pm.q, pm.brt), j.halfm, pm.q);
  }
  static __device__ __forceinline__ void store_fwd(const DevCtx &, const Job &j, const DevPrime &pm,
                                                   uint32_t n, u64 U) {
    U += (U >= pm.q8 ? pm.nq8 : 0);                       // [0,16q) -> [0,8q)
    u64 v = mul_shoup(c_at(j, n) + pm.q8 - U, j.inv.x, j.inv.y, pm.q); // exact for any 64-bit operand
    if (j.add) v = addmod(j.add[n], v, pm.q);
    j.dst[n] = v;
  }
  struct Pre { u64 c, add; };
  static __device__ __forceinline__ Pre prefetch(const DevCtx &, const Job &j, const DevPrime &, uint32_t n) {
    return Pre{c_at(j, n), j.add ? j.add[n] : 0};
  }
  static __device__ __forceinline__ void store_fwd_pre(const DevCtx &, const Job &j, const DevPrime &pm, uint32_t n, u64 U, const Pre &p) {
    U += (U >= pm.q8 ? pm.nq8 : 0);
    u64 v = mul_shoup(p.c + pm.q8 - U, j.inv.x, j.inv.y, pm.q);
    if (j.add) v = addmod(p.add, v, pm.q);
    j.dst[n] = v;
  }
};

using OpModDown = OpModDownT<false>;
using OpModDownG = OpModDownT<true>;

// ---- relinearize followed by rescale, evaluated together (same canonical result as the two
// SEAL calls in sequence, seal_executor.h:200 then :213).  With ct' = relinearize(a):
//   ct'[K][i] = a[K][i] + (prod[K][i] - NTT_i(u_Ki)) * P^-1,  u_Ki = (r_K mod q_i) - floor(P/2) mod q_i
//   out[K][i] = (ct'[K][i] - NTT_i(v_Ki)) * q_last^-1,        v_Ki = (t_K mod q_i) - floor(q_last/2) mod q_i
// NTT is linear, so NTT_i(u)*P^-1 + NTT_i(v) = NTT_i(u*P^-1 + v): one forward transform per
// (K,i) instead of two, and t_K = INTT(ct'[K][last]) + q_last/2 needs no NTT of u at all:
//   t_K = INTT_last(a[K][last] + prod[K][last]*P^-1) - u_K,last*P^-1 + floor(q_last/2).

// Where a[K] (the polynomials the key-switch result is added to) comes from (AM):
//   RR_MEM    read from memory
//   RR_MUL    d_K of a fused product, evaluated on load / in the epilogue (r03 form)
//   RR_FOLDED nowhere: the key-switch kernel already added P * a[K] to prod (KS_FOLDMUL / KS_FOLDADD), so
//             prod * P^-1 carries it
constexpr int RR_MEM = 0, RR_MUL = 1, RR_FOLDED = 2;

// inverse transform producing t_K; job = K, prime = last data prime
template <int AM> struct OpRRLastT {
  static constexpr bool MUL = AM == RR_MUL;
  struct Params {
    const u64 *a;     // a[0][last]
    size_t a_ps;
    const u64 *prod;  // prod[0][last]
    size_t prod_ps;
    const u64 *r;     // r_0 (INTT of the special limb + P/2)
    size_t r_ps;
    u64 *t;
    size_t t_ps;
    uint32_t last, sp;
    PtrTab a_tab; // used when a == nullptr: a_tab.p[job] = poly K of instance b at limb `last` (job = 2b+K)
    std::conditional_t<MUL, MulTab, NoMul> mul{}; // MUL: a[K] = d_K of product b = job / 2 (fused multiply), evaluated on load
  };
  struct Job {
    uint32_t prime;
    const u64 *a, *prod, *r;
    u64 *dst;
    u64 halfP;
    ulonglong2 pinv;
    bool lazy;
    MulSrc mul;
    uint32_t K;
    size_t off;
  };
  static dim3 grid(const Params &, uint32_t jobs) { return dim3(1, jobs, 1); }
  static constexpr int loop_axis = 1; // every job is modulo the last data prime (ntt_loop_kernel)
  static __device__ __forceinline__ bool setup(const DevCtx &cx, const Params &p, uint32_t, uint32_t job, uint32_t,
                                               Job &j) {
    j.prime = p.last;
    if constexpr (MUL) j.mul = mul_src(p.mul, cx.N, job >> 1);
    j.K = job & 1u;
    j.off = (size_t)p.last * cx.N;
    j.a = AM != RR_MEM ? nullptr : (p.a ? p.a + job * p.a_ps : p.a_tab.p[job]);
    j.prod = p.prod + job * p.prod_ps;
    j.r = p.r + job * p.r_ps;
    j.dst = p.t + job * p.t_ps;
    j.halfP = cx.halfmod[p.sp * cx.k + p.last];
    j.pinv = cx.invq[p.sp * cx.k + p.last];
    j.lazy = false;
    return true;
  }
  template <bool LZ>
  static __device__ __forceinline__ u64 load(const DevCtx &cx, const Job &j, const DevPrime &pm, uint32_t n) {
    const u64 pv = mul_shoup(j.prod[n], j.pinv.x, j.pinv.y, pm.q);
    if constexpr (AM == RR_FOLDED) return pv;
    const u64 av = MUL ? product_poly(j.mul, j.K, j.off + n, pm) : j.a[n];
    return addmod(av, pv, pm.q);
  }
  static __device__ __forceinline__ void store(const DevCtx &, const Job &j, const DevPrime &pm, uint32_t n, u64 x) {
    const u64 u = submod(barrett64(j.r[n], pm.q, pm.brt), j.halfP, pm.q);
    x = submod(x, mul_shoup(u, j.pinv.x, j.pinv.y, pm.q), pm.q);
    j.dst[n] = addmod(x, pm.q >> 1, pm.q);
  }
};

// forward transform of u*P^-1 + v with the combined epilogue; job -> (K = job / jl, i = job % jl)
template <int AM> struct OpRRT {
  static constexpr bool MUL = AM == RR_MUL;
  struct Params {
    const u64 *r;
    size_t r_ps;
    const u64 *t;
    size_t t_ps;
    const u64 *a;
    size_t a_ps;
    const u64 *prod;
    size_t prod_ps;
    u64 *dst;
    size_t dst_ps;
    uint32_t sp, last, jl;
    PtrTab a_tab; // used when a == nullptr: a_tab.p[K] = poly K (limb 0), K = 2b + {0,1}
    std::conditional_t<MUL, MulTab, NoMul> mul{}; // MUL: a[K] = d_(K&1) of product b = K / 2 (fused multiply), evaluated in the epilogue
  };
  struct Job {
    uint32_t prime;
    const u64 *r, *t, *a, *prod;
    u64 *dst;
    u64 halfP, halfL;
    ulonglong2 pinv, linv;
    bool lazy;
    MulSrc mul;
    uint32_t K;
    size_t off;
  };
  static dim3 grid(const Params &p, uint32_t jobs) { return dim3(1, p.jl, jobs / p.jl); }
  static constexpr int loop_axis = 2; // jobs that share a prime lie along grid.z (ntt_loop_kernel)
  static __device__ __forceinline__ bool setup(const DevCtx &cx, const Params &p, uint32_t, uint32_t i, uint32_t K,
                                               Job &j) {
    j.prime = i;
    j.r = p.r + K * p.r_ps;
    j.t = p.t + K * p.t_ps;
    if constexpr (MUL) j.mul = mul_src(p.mul, cx.N, K >> 1);
    j.K = K & 1u;
    j.off = (size_t)i * cx.N;
    j.a = AM != RR_MEM ? nullptr : (p.a ? p.a + K * p.a_ps : p.a_tab.p[K]) + (size_t)i * cx.N;
    j.prod = p.prod + K * p.prod_ps + (size_t)i * cx.N;
    j.dst = p.dst + K * p.dst_ps + (size_t)i * cx.N;
    j.halfP = cx.halfmod[p.sp * cx.k + i];
    j.halfL = cx.halfmod[p.last * cx.k + i];
    j.pinv = cx.invq[p.sp * cx.k + i];
    j.linv = cx.invq[p.last * cx.k + i];
    // lazy input: lazy5(r + q_i - halfP) + t + q_i - halfL < 5q_i + q_last + q_i <= 10 q_i
    j.lazy = cx.primes[p.last].q <= cx.primes[i].q4 && cx.primes[p.sp].q <= cx.primes[i].q8;
    return true;
  }
  template <bool LZ>
  static __device__ __forceinline__ u64 load(const DevCtx &, const Job &j, const DevPrime &pm, uint32_t n) {
    if (LZ)
      return mul_tw_lazy5(j.r[n] + (pm.q - j.halfP), j.pinv.x, j.pinv.y, pm.nq) + (j.t[n] + (pm.q - j.halfL));
    const u64 u = submod(barrett64(j.r[n], pm.q, pm.brt), j.halfP, pm.q);
    const u64 v = submod(barrett64(j.t[n], pm.q, pm.brt), j.halfL, pm.q);
    return addmod(mul_shoup(u, j.pinv.x, j.pinv.y, pm.q), v, pm.q);
  }
  static __device__ __forceinline__ void store_fwd(const DevCtx &cx, const Job &j, const DevPrime &pm, uint32_t n, u64 W) {
    W += (W >= pm.q8 ? pm.nq8 : 0);                                                  // [0,16q) -> [0,8q)
    u64 av = 0;
    if constexpr (AM != RR_FOLDED) av = MUL ? product_poly(j.mul, j.K, j.off + n, pm) : j.a[n];
    const u64 x = av + mul_tw_lazy5(j.prod[n], j.pinv.x, j.pinv.y, pm.nq) + pm.q8 - W; // < 14q < 2^64
    j.dst[n] = mul_shoup(x, j.linv.x, j.linv.y, pm.q);                               // exact for any 64-bit operand
  }
  // (no Pre here: requesting prod ahead of the tile measured 2 % slower on this pass — 372 against 364 us per
  // 32-triple launch — its on-the-fly products already keep four loads per word in flight)
  // ntt_loop_kernel (one wave walks the jobs: the epilogue's loads would be waited for once per job): prod, and a when
  // it comes from memory, requested before the job's transform
  struct LoopPre { u64 prod, a; };
  static __device__ __forceinline__ LoopPre loop_prefetch(const DevCtx &, const Job &j, const DevPrime &, uint32_t n) {
    return LoopPre{j.prod[n], AM == RR_MEM ? j.a[n] : 0};
  }
  static __device__ __forceinline__ void store_fwd_loop(const DevCtx &cx, const Job &j, const DevPrime &pm, uint32_t n, u64 W, const LoopPre &p) {
    if constexpr (AM == RR_MUL) {
      store_fwd(cx, j, pm, n, W);
    } else {
      W += (W >= pm.q8 ? pm.nq8 : 0);
      const u64 x = p.a + mul_tw_lazy5(p.prod, j.pinv.x, j.pinv.y, pm.nq) + pm.q8 - W;
      j.dst[n] = mul_shoup(x, j.linv.x, j.linv.y, pm.q);
    }
  }
};

using OpRRLast = OpRRLastT<RR_MEM>;
using OpRRLastMul = OpRRLastT<RR_MUL>;
using OpRRLastFolded = OpRRLastT<RR_FOLDED>;
using OpRR = OpRRT<RR_MEM>;
using OpRRMul = OpRRT<RR_MUL>;
using OpRRFolded = OpRRT<RR_FOLDED>;

} // namespace evah
