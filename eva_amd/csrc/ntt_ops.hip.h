// ntt_ops.hip.h — the fused load / store ops of the transform passes: base conversion, rounding offsets, mod-down / rescale combines, on-the-fly products
// (part of ntt.hip.h until r5; included by it, in the order the definitions depend on each other)
#pragma once
#include "ntt.hip.h"
#include "ntt_window_sum.hip.h"
#include "ntt_ks_inner.hip.h"

namespace evah {

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
    bool diag = false; // the diagonal I == J is transformed as well (no NTT-form target exists: the chain step, ntt_chain.hip.h)
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
    if (I == J && !p.diag) return false;
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

// ---- a ciphertext product that is rescaled straight away (Mul -> Rescale, r6): the size-3 product is never written —
// its polynomials d_K are formed where the rescale reads them (the last limb in the inverse transform's load, every other
// limb in the combine epilogue).  Same canonical residues as evaluator.multiply / square then rescale_to_next.
// Inverse transform of limb `last` of d_K; job = b * nK + (K - K0) for the polynomials K0 .. K0 + nK - 1 of product b
struct OpMulPolyIntt {
  struct Params {
    MulTab mul;
    u64 *dst;          // r[job][N]
    uint32_t K0, nK, last;
    int addhalf;
  };
  struct Job {
    uint32_t prime, K;
    size_t off;
    MulSrc mul;
    u64 *dst;
    int addhalf;
    bool lazy;
  };
  static dim3 grid(const Params &, uint32_t jobs) { return dim3(1, jobs, 1); }
  static constexpr int loop_axis = 1; // every job is modulo the last data prime
  static __device__ __forceinline__ bool setup(const DevCtx &cx, const Params &p, uint32_t, uint32_t job, uint32_t, Job &j) {
    j.prime = cx.prime_of(p.last);
    j.K = p.K0 + job % p.nK;
    j.off = (size_t)p.last * cx.N;
    j.mul = mul_src(p.mul, cx.N, job / p.nK);
    j.dst = p.dst + (size_t)job * cx.N;
    j.addhalf = p.addhalf;
    j.lazy = false;
    return true;
  }
  template <bool LZ>
  static __device__ __forceinline__ u64 load(const DevCtx &, const Job &j, const DevPrime &pm, uint32_t n) {
    return product_poly(j.mul, j.K, j.off + n, pm);
  }
  static __device__ __forceinline__ void store(const DevCtx &, const Job &j, const DevPrime &pm, uint32_t n, u64 v) {
    if (j.addhalf) v = addmod(v, pm.q >> 1, pm.q);
    j.dst[n] = v;
  }
  static __device__ __forceinline__ void store_fwd(const DevCtx &, const Job &, const DevPrime &, uint32_t, u64) {}
};
// Divide-and-round by the last data prime with c = d_K of a product formed in the epilogue; job -> (pp = job / jl, i = job % jl),
// pp = b * nK + (K - K0); r[pp] = INTT(limb a of d_K) + floor(q_a/2) (OpMulPolyIntt)
struct OpModDownMul {
  struct Params {
    const u64 *r;      // [pp][N]
    MulTab mul;
    u64 *dst;          // polynomial pp at dst + pp * dst_ps
    size_t dst_ps;
    uint32_t K0, nK, a, jl;
  };
  struct Job {
    uint32_t prime, K;
    const u64 *src;
    size_t off;
    MulSrc mul;
    u64 *dst;
    u64 halfm;
    ulonglong2 inv;
    bool lazy;
  };
  static dim3 grid(const Params &p, uint32_t jobs) { return dim3(1, p.jl, jobs / p.jl); }
  static constexpr int loop_axis = 2;
  static __device__ __forceinline__ bool setup(const DevCtx &cx, const Params &p, uint32_t, uint32_t i, uint32_t pp, Job &j) {
    j.prime = cx.prime_of(i);
    j.K = p.K0 + pp % p.nK;
    j.src = p.r + (size_t)pp * cx.N;
    j.off = (size_t)i * cx.N;
    j.mul = mul_src(p.mul, cx.N, pp / p.nK);
    j.dst = p.dst + pp * p.dst_ps + (size_t)i * cx.N;
    j.halfm = cx.halfmod[p.a * cx.k + j.prime];
    j.inv = cx.invq[p.a * cx.k + j.prime];
    j.lazy = cx.primes[p.a].q <= cx.primes[j.prime].q8;
    return true;
  }
  template <bool LZ>
  static __device__ __forceinline__ u64 load(const DevCtx &, const Job &j, const DevPrime &pm, uint32_t n) {
    return conv<LZ>(j, pm, j.src[n]);
  }
  static constexpr bool pre_addhalf = true;
  static __device__ __forceinline__ uint32_t pre_prime(const Params &p, const Job &) { return p.a; }
  static __device__ __forceinline__ const u64 *pre_src(const Job &j) { return j.src; }
  template <bool LZ> static __device__ __forceinline__ u64 conv(const Job &j, const DevPrime &pm, u64 v) {
    if (LZ) return v + (pm.q - j.halfm);
    return submod(barrett64(v, pm.q, pm.brt), j.halfm, pm.q);
  }
  static __device__ __forceinline__ void store_fwd(const DevCtx &, const Job &j, const DevPrime &pm, uint32_t n, u64 U) {
    U += (U >= pm.q8 ? pm.nq8 : 0); // [0,16q) -> [0,8q)
    j.dst[n] = mul_shoup(product_poly(j.mul, j.K, j.off + n, pm) + pm.q8 - U, j.inv.x, j.inv.y, pm.q);
  }
};

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
