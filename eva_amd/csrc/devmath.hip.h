// devmath.hip.h — 64-bit modular arithmetic for gfx950 (CDNA4).
//
// All ciphertext data are residues < q < 2^61.  CDNA4 has no 64x64->128 multiply; every
// product below lowers to v_mad_u64_u32 / v_mul_hi_u32 chains, so the instruction budget is
// counted in 32-bit multiplies: Shoup mulmod = 10, lazy 128-bit MAC term = 4, Barrett-128 = 18.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace evah {

typedef unsigned long long u64;

struct u128_t {
  u64 lo, hi;
};

// Per-prime constants, device resident (one record per RNS prime).
struct DevPrime {
  u64 q;        // modulus
  u64 brt;      // floor(2^64 / q)            (64-bit Barrett)
  u64 r0, r1;   // floor(2^128 / q) lo, hi    (128-bit Barrett)
  u64 ninv;     // N^-1 mod q
  u64 ninv_s;   // Shoup quotient of ninv
  u64 w0ninv;   // irp[1] * N^-1 mod q  (last inverse stage twiddle with the scaling folded in)
  u64 w0ninv_s;
  u64 nq;       // 2^64 - q   (kept as data: the compiler must not turn x + t*nq back into x - t*q)
  u64 q5;       // 5q, the inverse lazy-butterfly threshold
  u64 q4, q8;   // 4q (forward difference offset), 8q (forward reduction threshold)
  u64 nq5, nq8; // 2^64 - 5q, 2^64 - 8q: conditional subtraction as select + one 64-bit add (data, as nq)
  u64 c64, c64s; // 2^64 mod q and its Shoup quotient: folds the high word of a 128-bit value (barrett128)
  // q = 2^b - c (b = bit length): x = (X mod 2^b) + (X >> b) c is congruent to X and < q + 16c for X < 16q.
  // tb_c = c when it fits 32 bits and b > 32 (every CoeffModulus::Create prime of 33..60 bits), else 0
  uint32_t tb_c, tb_sh;   // c, b - 32
  uint32_t tb_mask, tb_pad; // 2^(b-32) - 1
};

// Device-side view of a context (passed by value to kernels).
struct DevCtx {
  const DevPrime *primes;      // [k]
  const ulonglong2 *tw_fwd;    // [k][N]  (w, floor(w*2^64/q)), heap order: stage m group i -> [m+i]
  const ulonglong2 *tw_inv;    // [k][N]  inverses, same indexing
  const ulonglong2 *invq;      // [k][k]  invq[a*k+b] = (q_a^-1 mod q_b, Shoup quotient)
  const u64 *halfmod;          // [k][k]  (q_a >> 1) mod q_b
  const ulonglong2 *modq;      // [k][k]  modq[a*k+b] = (q_a mod q_b, Shoup quotient); (0, 0) on the diagonal
  const ulonglong2 *plinv;     // [k][k]  plinv[a*k+b] = (P q_a^-1 mod q_b, Shoup quotient), P = primes[k-1]; (0, 0) for b == a, b == k-1
  uint32_t N, logN, k;
  // Limb -> prime map of the context's values: limb j of a ciphertext / plaintext is modulo
  // primes[p0 + j * pstep].  An ordinary context has (0, 1).  A limb-sharded context (shard s of G,
  // "limb i of every poly lives on GPU i mod G") has (s, G): its values hold
  // only the limbs i = s, s + G, ... and every per-limb kernel works on them unchanged.
  uint32_t p0, pstep;
  // Guarded launches (hoisted rotations' exact fallback, rotate.hip): when set, a kernel does
  // nothing unless the counter it points to exceeds guard_min.  Null for every ordinary launch.
  const uint32_t *guard;
  uint32_t guard_min;
  __host__ __device__ uint32_t prime_of(uint32_t limb) const { return p0 + limb * pstep; }
  __device__ __forceinline__ bool skipped() const { return guard && *guard <= guard_min; }
};

__device__ __forceinline__ u128_t mul128(u64 a, u64 b) {
  const unsigned __int128 p = (unsigned __int128)a * b; // 4 v_mad_u64_u32 (a*b and __umul64hi separately: 7 multiplies)
  u128_t r;
  r.lo = (u64)p;
  r.hi = (u64)(p >> 64);
  return r;
}
// acc += a*b (128-bit).  Written on unsigned __int128 so the backend keeps the sum in a
// v_add_co / v_addc carry chain (14 VALU instructions instead of 19 with explicit carry tests).
__device__ __forceinline__ void acc128(u128_t &acc, u64 a, u64 b) {
  unsigned __int128 s = ((unsigned __int128)acc.hi << 64) | acc.lo;
  s += (unsigned __int128)a * b;
  acc.lo = (u64)s;
  acc.hi = (u64)(s >> 64);
}

// acc += a*b for CANONICAL operands (a, b < q <= 2^60: runtime.hip refuses larger primes) and sums of fewer than 2^7 terms —
// the shape of the hoisted key inner products.  With both high words below 2^28 the cross terms a0 b1 + a1 b0 (+ the high
// word of a0 b0 + lo) fit ONE 64-bit multiply-add chain, and a1 b1 + hi (< 2^56 + 2^63) cannot carry out, so the only carry
// of the whole update is the one out of the first v_mad_u64_u32 (its scalar destination), added where the cross terms' high
// word is added: 4 v_mad_u64_u32 + 2 v_addc_co_u32 + 2 moves, against the ~15 instructions the unsigned __int128 form
// compiles to (r6: k_hoist_mac<4,2> 2 567 -> 1 700 VALU instructions per wave).  The result is the same 128-bit sum.
#ifndef EVAH_MAC_ASM
#define EVAH_MAC_ASM 1
#endif
__device__ __forceinline__ void acc128c(u128_t &acc, u64 a, u64 b) {
#if EVAH_MAC_ASM
  const uint32_t a0 = (uint32_t)a, a1 = (uint32_t)(a >> 32), b0 = (uint32_t)b, b1 = (uint32_t)(b >> 32);
  u64 t, c1, c2, c3;
  asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(t), "=s"(c1) : "v"(a0), "v"(b0), "v"(acc.lo));
  u64 m = (u64)a0 * b1 + (t >> 32);       // < 2^60 + 2^32
  m = (u64)a1 * b0 + m;                   // < 2^61 + 2^32
  const u64 x = (u64)a1 * b1 + acc.hi;    // < 2^56 + acc.hi: no carry while the sum stays below 2^127
  uint32_t x0, x1;
  asm("v_addc_co_u32 %0, %1, %2, %3, %4" : "=v"(x0), "=s"(c2) : "v"((uint32_t)x), "v"((uint32_t)(m >> 32)), "s"(c1));
  asm("v_addc_co_u32 %0, %1, %2, 0, %3" : "=v"(x1), "=s"(c3) : "v"((uint32_t)(x >> 32)), "s"(c2));
  acc.lo = (m << 32) | (uint32_t)t;
  acc.hi = ((u64)x1 << 32) | x0;
#else
  acc128(acc, a, b);
#endif
}

__device__ __forceinline__ u64 addmod(u64 a, u64 b, u64 q) {
  u64 s = a + b;
  return s >= q ? s - q : s;
}
__device__ __forceinline__ u64 submod(u64 a, u64 b, u64 q) { return a >= b ? a - b : a + q - b; }
__device__ __forceinline__ u64 negmod(u64 a, u64 q) { return a ? q - a : 0; }

// x*w mod q in [0, 2q), any 64-bit x, ws = floor(w*2^64/q)   (Shoup / Harvey)
__device__ __forceinline__ u64 mul_shoup_lazy(u64 x, u64 w, u64 ws, u64 q) {
  return x * w - __umul64hi(x, ws) * q;
}
__device__ __forceinline__ u64 mul_shoup(u64 x, u64 w, u64 ws, u64 q) {
  u64 r = mul_shoup_lazy(x, w, ws, q);
  return r >= q ? r - q : r;
}

// x mod q for any 64-bit x; brt = floor(2^64/q)
__device__ __forceinline__ u64 barrett64(u64 x, u64 q, u64 brt) {
  u64 r = x - __umul64hi(x, brt) * q;
  return r >= q ? r - q : r;
}

// Twiddle product with a cheap quotient estimate.  Shoup's q^ = floor(x*ws / 2^64) needs the
// full 64x64 high product (4 multiplies + carries); dropping the partial products that only
// feed carries gives q~ with q^ - 2 <= q~ <= q^, i.e. x*w - q~*q in [0, 4q) for ANY 64-bit x,
// with 3 multiplies.  x*w - q~*q is evaluated as x*w + q~*(2^64 - q) so the second product
// accumulates onto the first (one mad chain, no 64-bit subtract).
__device__ __forceinline__ u64 mul_tw_lazy5(u64 x, u64 w, u64 ws, u64 nq) {
  const uint32_t x0 = (uint32_t)x, x1 = (uint32_t)(x >> 32), s0 = (uint32_t)ws, s1 = (uint32_t)(ws >> 32);
  const u64 qt = (u64)x1 * s1 + (u64)__umulhi(x1, s0) + (u64)__umulhi(x0, s1);
  return x * w + qt * nq;
}

// (hi:lo) mod q for any 128-bit input, canonical.  The high word is folded with the constant
// 2^64 mod q as a lazy Shoup product (< 4q, 11 instructions), the sum with the low word is brought
// back below 2^64 with the same constant when it carries, and one 64-bit Barrett step finishes:
// ~40 VALU instructions where the textbook 128-bit Barrett (two 64x64 high products against
// floor(2^128/q)) compiles to ~95 on gfx950.  Every ciphertext product, weighted sum and key inner
// product ends in this reduction.
__device__ __forceinline__ u64 barrett128(u128_t x, const DevPrime &m) {
  const u64 t = mul_tw_lazy5(x.hi, m.c64, m.c64s, m.nq); // hi * 2^64 mod q, lazy in [0, 4q)
  u64 v = t + x.lo;
  v += (v < t) ? m.c64 : 0; // carried: v = t + lo - 2^64 < t < 4q, and 2^64 = c64 (mod q); no second carry
  return barrett64(v, m.q, m.brt);
}
// (hi:lo) -> a 64-bit value congruent to it mod q, NOT canonical: barrett128 without its last Barrett step (~21 VALU
// instead of ~40).  For consumers that multiply the value by a constant with an exact / lazy Shoup product (those
// accept any 64-bit operand) — the key inner products on their way into the relinearize + rescale combine passes.
__device__ __forceinline__ u64 reduce128_lazy(u128_t x, const DevPrime &m) {
  const u64 t = mul_tw_lazy5(x.hi, m.c64, m.c64s, m.nq);
  u64 v = t + x.lo;
  v += (v < t) ? m.c64 : 0;
  return v;
}
__device__ __forceinline__ u64 mulmod(u64 a, u64 b, const DevPrime &m) {
  return barrett128(mul128(a, b), m);
}

}  // namespace evah
