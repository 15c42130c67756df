// ntt_chain.hip.h — the chain step Mul -> Rescale -> Relinearize (lazy relinearization's order, seal_executor.h:164 / :162,
// :213-214, :200) for latency-bound launches: six dependent launches instead of nine or ten, and 3 l' forward transforms fewer.
//
// With d_0, d_1, d_2 the polynomials of the product at l limbs (L = q_{l-1} the prime the rescale divides by, l' = l - 1,
// P the special prime) the three SEAL calls compute, per data limb i < l':
//   d'_K,i  = (d_K,i - NTT_i(u_K,i)) L^-1              u_K,i = [INTT_L(d_K,L) + L/2] mod q_i - (L/2 mod q_i)     (rescale)
//   prod_K  = sum_J NTT(t_J mod q) key_J,K             t_J   = INTT_J(d'_2,J)                                    (digits)
//   out_K,i = d'_K,i + (prod_K,i - NTT_i(v_K,i)) P^-1  v_K,i = [INTT_P(prod_K,P) + P/2] mod q_i - (P/2 mod q_i)  (mod-down)
// The transforms are linear over Z_{q_i} and every quantity above is a canonical residue, so the same words come out of:
//   (1) t_J = (INTT_J(d_2,J) - u_2,J) L^-1 — the rescaled d_2 formed in COEFFICIENT form, where the digit decomposition wants
//       it: l' inverse transforms instead of l' forward + l' inverse ones (the diagonal term NTT_J(t_J) that the key-switch
//       kernel used to read from the stored d'_2 joins the digit transforms it already runs);
//   (2) out_K,i = (prod'_K,i - NTT_i(u_K,i L^-1 P + v_K,i)) P^-1 with prod' = prod + (P L^-1) d_K formed inside the key inner
//       product (KS_FOLDMUL with DevCtx::plinv in place of P): the rescale of d_0, d_1 and the mod-down share ONE forward
//       transform per (K, i) and one launch pair, and d'_0, d'_1 are never formed.
// Launches: [products + contiguous inverse pass: d_2 all limbs, d_0 / d_1 limb L] -> [strided inverse passes of limb J and
// limb L, (1), digit conversion + strided forward pass] -> [ks_inner_kernel: contiguous forward pass + inner product + fold +
// the special limb's contiguous inverse pass] -> [strided inverse passes of d_K,L and prod_K,P, (2)'s input, strided forward
// pass] -> [contiguous forward pass + OpModDown's combine].  Larger launch sets split the second and fourth of these.
#pragma once
#include "ntt.hip.h"
#include "ntt_ops.hip.h"

namespace evah {

// TWO strided inverse passes (source limbs A and B of the op, from the contiguous passes' intermediates) on one column tile,
// the op's combine of the two canonical values, then either the store of the result (FWD = false) or the strided forward
// pass under the job's own prime (FWD = true; lazy intermediate to jb.dst) — ntt_inv_fwd_kernel with a second source.
template <int P, int LR, class Op, bool FWD>
__global__ void __launch_bounds__(NTT_THREADS)
ntt_inv2_kernel(DevCtx cx, typename Op::Params prm, int log_tiles) {
  extern __shared__ __attribute__((aligned(16))) u64 lds[];
  constexpr int NTT_R = 1 << LR;
  constexpr int S = 1 << P, TPS = S / NTT_R, SP = lds_sub_stride<P>();
  if (cx.skipped()) return;
  const uint32_t tile_idx = blockIdx.x & ((1u << log_tiles) - 1u);
  typename Op::Job jb;
  if (!Op::setup(cx, prm, blockIdx.x >> log_tiles, blockIdx.y, blockIdx.z, jb)) return; // block-uniform
  const uint32_t pa = Op::prime_a(prm, jb), pb = Op::prime_b(prm, jb);
  const DevPrime pmA = cx.primes[pa], pmB = cx.primes[pb], pm = cx.primes[jb.prime];
  const ulonglong2 *twA = cx.tw_inv + (size_t)pa * cx.N, *twB = cx.tw_inv + (size_t)pb * cx.N, *tw = cx.tw_fwd + (size_t)jb.prime * cx.N;
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
  const u64 *srcA = Op::src_a(jb), *srcB = Op::src_b(jb);
  // both tiles' words are requested up front: B's wait in registers while A is transformed
  u64 vb[NTT_R];
#pragma unroll
  for (int it = 0; it < NTT_R; it++) lds[lds_at(it)] = srcA[n0 + it * nstep];
#pragma unroll
  for (int it = 0; it < NTT_R; it++) vb[it] = srcB[n0 + it * nstep];
  ulonglong2 *twlI = reinterpret_cast<ulonglong2 *>(lds + ((C * SP + 1) & ~1)), *twl = twlI + S;
  for (int idx = threadIdx.x; idx < S; idx += T) {
    twlI[idx] = twA[idx];
    if constexpr (FWD) twl[idx] = tw[idx];
  }
  __syncthreads();
  const int sub = threadIdx.x / TPS, tid = threadIdx.x % TPS;
  RoundSeq<P, LR, 0, true, true, false>::run(lds + sub * SP, tid, 0, 0, twlI, pmA); // canonical mod q_a (N^-1 folded in)
  __syncthreads();
  u64 va[NTT_R];
#pragma unroll
  for (int it = 0; it < NTT_R; it++) {
    u64 v = lds[lds_at(it)];
    if (Op::a_addhalf) v = addmod(v, pmA.q >> 1, pmA.q);
    va[it] = v;
    lds[lds_at(it)] = vb[it];
  }
  for (int idx = threadIdx.x; idx < S; idx += T) twlI[idx] = twB[idx];
  __syncthreads();
  RoundSeq<P, LR, 0, true, true, false>::run(lds + sub * SP, tid, 0, 0, twlI, pmB);
  __syncthreads();
  auto combine = [&](auto lazy_tag) {
    constexpr bool LZ = decltype(lazy_tag)::value;
#pragma unroll
    for (int it = 0; it < NTT_R; it++) {
      u64 v = lds[lds_at(it)];
      if (Op::b_addhalf) v = addmod(v, pmB.q >> 1, pmB.q);
      const u64 w = Op::template combine<LZ>(jb, pm, pmB, va[it], v);
      if constexpr (FWD) lds[lds_at(it)] = w;
      else jb.dst[n0 + it * nstep] = w;
    }
  };
  if (jb.lazy) combine(std::true_type{});
  else combine(std::false_type{});
  if constexpr (FWD) {
    __syncthreads();
    forward_rounds<P, LR, true, false>(lds + sub * SP, tid, 0, 0, twl, pm);
    __syncthreads();
#pragma unroll
    for (int it = 0; it < NTT_R; it++) jb.dst[n0 + it * nstep] = lds[lds_at(it)]; // lazy intermediate of the forward transform
  }
}

// Contiguous inverse pass of the chain step's sources, products formed on load: job (y, b) of grid (1, l + 2, n) is limb y
// of d_2 of product b for y < l (intermediate to t2[b][y]), limb l - 1 of d_0 / d_1 for y = l, l + 1 (to r01[2 b + K]).
struct OpChainIntt {
  struct Params {
    MulTab mul;
    u64 *t2;      // [n][l][N]
    u64 *r01;     // [2 n][N]
    uint32_t l;
  };
  struct Job {
    uint32_t prime, K;
    size_t off;
    MulSrc mul;
    u64 *dst;
    bool lazy;
  };
  static dim3 grid(const Params &p, uint32_t jobs) { return dim3(1, p.l + 2, jobs / (p.l + 2)); }
  static constexpr int loop_axis = 2; // the products of a launch share every limb's prime
  static __device__ __forceinline__ bool setup(const DevCtx &cx, const Params &p, uint32_t, uint32_t y, uint32_t b, Job &j) {
    const uint32_t row = y < p.l ? y : p.l - 1;
    j.prime = cx.prime_of(row);
    j.K = y < p.l ? 2u : y - p.l;
    j.off = (size_t)row * cx.N;
    j.mul = mul_src(p.mul, cx.N, b);
    j.dst = y < p.l ? p.t2 + ((size_t)b * p.l + y) * cx.N : p.r01 + ((size_t)2 * b + (y - p.l)) * cx.N;
    j.lazy = false;
    return true;
  }
  // (mul.b == nullptr, block-uniform: a STORED size-3 ciphertext — evah_rescale_relinearize — whose polynomial K lies
  // at mul.a + K * mul.sa)
  template <bool LZ>
  static __device__ __forceinline__ u64 load(const DevCtx &, const Job &j, const DevPrime &pm, uint32_t n) {
    if (!j.mul.b) return j.mul.a[j.K * j.mul.sa + j.off + n];
    return product_poly(j.mul, j.K, j.off + n, pm);
  }
  static __device__ __forceinline__ void store(const DevCtx &, const Job &j, const DevPrime &, uint32_t n, u64 v) { j.dst[n] = v; }
  static __device__ __forceinline__ void store_fwd(const DevCtx &, const Job &, const DevPrime &, uint32_t, u64) {}
};

// (1) of the header: t_J = (INTT_J(d_2,J) - u_J) L^-1 mod q_J, canonical, from the intermediates OpChainIntt left.
// A = limb L = l - 1 (with the rounding offset L/2), B = limb J.
struct ChainDigitCore {
  // x = INTT_L(d_2,L) + L/2 in [0, L), y = INTT_J(d_2,J) in [0, q_J)
  static __device__ __forceinline__ u64 digit(const DevPrime &pmJ, u64 halfm, ulonglong2 linv, bool lazy_a, u64 x, u64 y) {
    // lazy_a (L <= 8 q_J): x + q_J - halfm < 9 q_J stands in for u; y + 9 q_J - that stays positive and below 2^64
    const u64 d = lazy_a ? y + (pmJ.q8 + pmJ.q) - (x + (pmJ.q - halfm))
                         : y + pmJ.q - submod(barrett64(x, pmJ.q, pmJ.brt), halfm, pmJ.q);
    return mul_shoup(d, linv.x, linv.y, pmJ.q); // exact for any 64-bit operand
  }
};
// stored form: job (J, b) of grid (1, l', n); t[b][J][N] feeds OpKsDigit (diag) — launch sets too large to recompute it per output limb
struct OpChainT {
  struct Params {
    const u64 *t2; // [n][l][N] intermediates (OpChainIntt)
    u64 *t;        // [n][l'][N]
    uint32_t l;    // limbs of the product (l' = l - 1)
  };
  struct Job {
    uint32_t prime;
    const u64 *a, *b;
    u64 *dst;
    u64 halfm;
    ulonglong2 linv;
    bool lazy, lazy_a;
  };
  static dim3 grid(const Params &p, uint32_t jobs) { return dim3(1, p.l - 1, jobs / (p.l - 1)); }
  static __device__ __forceinline__ bool setup(const DevCtx &cx, const Params &p, uint32_t, uint32_t J, uint32_t b, Job &j) {
    const uint32_t last = p.l - 1;
    j.prime = J;
    j.a = p.t2 + ((size_t)b * p.l + last) * cx.N;
    j.b = p.t2 + ((size_t)b * p.l + J) * cx.N;
    j.dst = p.t + ((size_t)b * (p.l - 1) + J) * cx.N;
    j.halfm = cx.halfmod[last * cx.k + J];
    j.linv = cx.invq[last * cx.k + J];
    j.lazy_a = cx.primes[last].q <= cx.primes[J].q8;
    j.lazy = false;
    return true;
  }
  static constexpr bool a_addhalf = true, b_addhalf = false;
  static __device__ __forceinline__ uint32_t prime_a(const Params &p, const Job &) { return p.l - 1; }
  static __device__ __forceinline__ uint32_t prime_b(const Params &, const Job &j) { return j.prime; }
  static __device__ __forceinline__ const u64 *src_a(const Job &j) { return j.a; }
  static __device__ __forceinline__ const u64 *src_b(const Job &j) { return j.b; }
  template <bool LZ>
  static __device__ __forceinline__ u64 combine(const Job &j, const DevPrime &, const DevPrime &pmJ, u64 x, u64 y) {
    return ChainDigitCore::digit(pmJ, j.halfm, j.linv, j.lazy_a, x, y);
  }
};
// fused form: job (J, I, b) of grid (l', l' + 1, n) recomputes t_J on its tile and goes on with the digit conversion's strided
// forward pass under prime kappa(I) — scratch[b][I][J], the diagonal I == J included (as OpKsDigit with diag)
struct OpChainDigit {
  struct Params {
    const u64 *t2; // [n][l][N]
    u64 *scratch;  // [n][l' + 1][l'][N]
    uint32_t l;
    size_t scratch_bs;
  };
  struct Job {
    uint32_t prime, digit;
    const u64 *a, *b;
    u64 *dst;
    u64 halfm;
    ulonglong2 linv;
    bool lazy, lazy_a;
  };
  static dim3 grid(const Params &p, uint32_t jobs) { return dim3(p.l - 1, p.l, jobs / ((p.l - 1) * p.l)); }
  static __device__ __forceinline__ bool setup(const DevCtx &cx, const Params &p, uint32_t J, uint32_t I, uint32_t b, Job &j) {
    const uint32_t last = p.l - 1, lp = p.l - 1;
    j.digit = J;
    j.prime = (I == lp) ? cx.k - 1 : I;
    j.a = p.t2 + ((size_t)b * p.l + last) * cx.N;
    j.b = p.t2 + ((size_t)b * p.l + J) * cx.N;
    j.dst = p.scratch + b * p.scratch_bs + ((size_t)I * lp + J) * cx.N;
    j.halfm = cx.halfmod[last * cx.k + J];
    j.linv = cx.invq[last * cx.k + J];
    j.lazy_a = cx.primes[last].q <= cx.primes[J].q8;
    j.lazy = cx.primes[J].q <= cx.primes[j.prime].q8; // t_J < q_J: a valid lazy input (< 12 q_kappa) as it is
    return true;
  }
  static constexpr bool a_addhalf = true, b_addhalf = false;
  static __device__ __forceinline__ uint32_t prime_a(const Params &p, const Job &) { return p.l - 1; }
  static __device__ __forceinline__ uint32_t prime_b(const Params &, const Job &j) { return j.digit; }
  static __device__ __forceinline__ const u64 *src_a(const Job &j) { return j.a; }
  static __device__ __forceinline__ const u64 *src_b(const Job &j) { return j.b; }
  template <bool LZ>
  static __device__ __forceinline__ u64 combine(const Job &j, const DevPrime &pm, const DevPrime &pmJ, u64 x, u64 y) {
    const u64 t = ChainDigitCore::digit(pmJ, j.halfm, j.linv, j.lazy_a, x, y);
    return LZ ? t : barrett64(t, pm.q, pm.brt);
  }
};

// (2) of the header: the input of the forward transform shared by the rescale of d_K and the mod-down of prod_K,
//   w = u L^-1 P + v  (mod q_i),  u = [x] - (L/2 mod q_i), v = [y] - (P/2 mod q_i),
// x = INTT_L(d_K,L) + L/2 (A: r01[pp]), y = INTT_P(prod_K,P) + P/2 (B: r[pp]); job (i, pp) of grid (1, l', 2 n), pp = 2 b + K.
// The second (contiguous) forward pass and the combine are OpModDown's, on the same intermediate (dst).
struct OpRsMd {
  struct Params {
    const u64 *r01, *r; // [2 n][N] each
    u64 *dst;
    size_t dst_ps;
    uint32_t last, sp, jl;
  };
  struct Job {
    uint32_t prime;
    const u64 *a, *b;
    u64 *dst;
    u64 halfL, halfP;
    ulonglong2 cl;
    bool lazy;
  };
  static dim3 grid(const Params &p, uint32_t jobs) { return dim3(1, p.jl, jobs / p.jl); }
  static constexpr int loop_axis = 2;
  static __device__ __forceinline__ bool setup(const DevCtx &cx, const Params &p, uint32_t, uint32_t i, uint32_t pp, Job &j) {
    j.prime = cx.prime_of(i);
    j.a = p.r01 + (size_t)pp * cx.N;
    j.b = p.r + (size_t)pp * cx.N;
    j.dst = p.dst + pp * p.dst_ps + (size_t)i * cx.N;
    j.halfL = cx.halfmod[p.last * cx.k + j.prime];
    j.halfP = cx.halfmod[p.sp * cx.k + j.prime];
    j.cl = cx.plinv[p.last * cx.k + j.prime];
    j.lazy = cx.primes[p.sp].q <= cx.primes[j.prime].q8; // y + q_i - halfP < 9 q_i, and the Shoup product below is < q_i
    return true;
  }
  static constexpr bool a_addhalf = true, b_addhalf = true;
  static __device__ __forceinline__ uint32_t prime_a(const Params &p, const Job &) { return p.last; }
  static __device__ __forceinline__ uint32_t prime_b(const Params &p, const Job &) { return p.sp; }
  static __device__ __forceinline__ const u64 *src_a(const Job &j) { return j.a; }
  static __device__ __forceinline__ const u64 *src_b(const Job &j) { return j.b; }
  template <bool LZ>
  static __device__ __forceinline__ u64 combine(const Job &j, const DevPrime &pm, const DevPrime &, u64 x, u64 y) {
    const u64 ul = mul_shoup(x + (pm.q - j.halfL), j.cl.x, j.cl.y, pm.q); // exact for any 64-bit operand
    if (LZ) return ul + (y + (pm.q - j.halfP));
    return addmod(ul, submod(barrett64(y, pm.q, pm.brt), j.halfP, pm.q), pm.q);
  }
  // larger launch sets: both sources fully inverse-transformed in place (rounding offsets added), this is the strided
  // forward pass's load
  template <bool LZ>
  static __device__ __forceinline__ u64 load(const DevCtx &, const Job &j, const DevPrime &pm, uint32_t n) {
    return combine<LZ>(j, pm, pm, j.a[n], j.b[n]);
  }
  static __device__ __forceinline__ void store_fwd(const DevCtx &, const Job &, const DevPrime &, uint32_t, u64) {}
};

} // namespace evah
