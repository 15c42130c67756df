/*
 * eva_oracle.c — see eva_oracle.h.  TEST INFRASTRUCTURE ONLY; PARITY UNPINNED vs SEAL bits.
 *
 * Plain C restatement of the SEAL 3.6 arithmetic that EVA's SEALExecutor calls
 * (/root/reference/eva/seal/seal_executor.h:114-243,279-404).  SEAL itself is an
 * un-vendored dependency (microsoft/SEAL v3.6.4), so each routine names the SEAL routine
 * whose published algorithm it restates and the EVA call site that reaches it.
 */
#include "eva_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;
typedef uint64_t u64;

/* ------------------------------------------------------------------ modular arithmetic */

uint64_t evo_mulmod(uint64_t a, uint64_t b, uint64_t q) { return (u64)(((u128)a * b) % q); }

uint64_t evo_powmod(uint64_t a, uint64_t e, uint64_t q) {
  u64 r = 1 % q;
  a %= q;
  while (e) {
    if (e & 1) r = evo_mulmod(r, a, q);
    a = evo_mulmod(a, a, q);
    e >>= 1;
  }
  return r;
}

uint64_t evo_invmod(uint64_t a, uint64_t q) { return evo_powmod(a, q - 2, q); /* q prime */ }

/* deterministic Miller-Rabin for 64-bit (SEAL Modulus::is_prime is probabilistic MR; the set
 * of primes is the same) */
int evo_is_prime(uint64_t n) {
  static const u64 bases[] = {2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37};
  if (n < 2) return 0;
  for (size_t i = 0; i < 12; i++) {
    if (n == bases[i]) return 1;
    if (n % bases[i] == 0) return 0;
  }
  u64 d = n - 1;
  int r = 0;
  while ((d & 1) == 0) { d >>= 1; r++; }
  for (size_t i = 0; i < 12; i++) {
    u64 x = evo_powmod(bases[i], d, n);
    if (x == 1 || x == n - 1) continue;
    int comp = 1;
    for (int j = 1; j < r; j++) {
      x = evo_mulmod(x, x, n);
      if (x == n - 1) { comp = 0; break; }
    }
    if (comp) return 0;
  }
  return 1;
}

/* SEAL util::get_primes(ntt_size=N, bit_size, count): candidates 2^b - 2N + 1, step -2N,
 * while > 2^(b-1); kept in descending order. */
static int get_primes(uint32_t N, int bit_size, uint32_t count, u64 *out) {
  u64 factor = 2ull * N;
  u64 value = (((u64)1 << bit_size) - 1) / factor * factor + 1;
  u64 lower = (u64)1 << (bit_size - 1);
  uint32_t found = 0;
  while (found < count && value > lower) {
    if (evo_is_prime(value)) out[found++] = value;
    value -= factor;
  }
  return found == count ? 0 : -1;
}

/* SEAL CoeffModulus::Create: per distinct bit size the primes are found descending; the
 * result is filled in bit_sizes order popping from the BACK of each size's list, so the first
 * occurrence of a size gets the smallest of its primes. */
int evo_coeff_modulus_create(uint32_t N, const int *bit_sizes, uint32_t n_bits, uint64_t *out) {
  uint32_t count[64] = {0}, used[64] = {0};
  u64 *table[64] = {0};
  int rc = 0;
  for (uint32_t i = 0; i < n_bits; i++) {
    if (bit_sizes[i] < 2 || bit_sizes[i] > 61) return -2;
    count[bit_sizes[i]]++;
  }
  for (int b = 2; b <= 61 && rc == 0; b++) {
    if (!count[b]) continue;
    table[b] = (u64 *)malloc(sizeof(u64) * count[b]);
    rc = get_primes(N, b, count[b], table[b]);
  }
  if (rc == 0)
    for (uint32_t i = 0; i < n_bits; i++) {
      int b = bit_sizes[i];
      out[i] = table[b][count[b] - 1 - used[b]];
      used[b]++;
    }
  for (int b = 0; b < 64; b++) free(table[b]);
  return rc;
}

/* SEAL util::try_minimal_primitive_root(degree=2N, modulus): any primitive 2N-th root, then
 * the minimum over its odd powers (= all primitive 2N-th roots). */
uint64_t evo_minimal_primitive_root(uint32_t N, uint64_t q) {
  u64 degree = 2ull * N;
  if ((q - 1) % degree) return 0;
  u64 e = (q - 1) / degree, root = 0;
  for (u64 g = 2; g < q; g++) {
    u64 r = evo_powmod(g, e, q);
    if (evo_powmod(r, N, q) == q - 1) { root = r; break; } /* r^N = -1 <=> order exactly 2N */
  }
  u64 sq = evo_mulmod(root, root, q), cur = root, best = root;
  for (u64 i = 0; i < N; i++) {
    if (cur < best) best = cur;
    cur = evo_mulmod(cur, sq, q);
  }
  return best;
}

/* ------------------------------------------------------------------ context */

typedef struct {
  u64 q;
  u64 ratio0, ratio1; /* floor(2^128 / q), SEAL Modulus::const_ratio */
  u64 psi;
  u64 *rp, *rps;   /* root powers (bit-reversed index) + Shoup quotients */
  u64 *irp, *irps; /* inverse root powers, same indexing */
  u64 ninv, ninvs; /* N^-1 mod q */
} evo_mod;

struct evo_ctx {
  uint32_t N, logN, k;
  evo_mod *m;
};

static inline uint32_t bitrev(uint32_t x, uint32_t bits) {
  uint32_t r = 0;
  for (uint32_t i = 0; i < bits; i++) r |= ((x >> i) & 1u) << (bits - 1 - i);
  return r;
}
static inline u64 shoup(u64 w, u64 q) { return (u64)(((u128)w << 64) / q); }

evo_ctx *evo_ctx_create(uint32_t N, uint32_t k, const uint64_t *primes) {
  evo_ctx *c = (evo_ctx *)calloc(1, sizeof(evo_ctx));
  c->N = N;
  c->k = k;
  c->logN = 0;
  while ((1u << c->logN) < N) c->logN++;
  c->m = (evo_mod *)calloc(k, sizeof(evo_mod));
  for (uint32_t i = 0; i < k; i++) {
    evo_mod *m = &c->m[i];
    u64 q = primes[i];
    m->q = q;
    /* floor(2^128/q): 2^128 = q*Q + R */
    u128 hi = (~(u128)0) / q; /* floor((2^128-1)/q); equals floor(2^128/q) unless q | 2^128 */
    m->ratio0 = (u64)hi;
    m->ratio1 = (u64)(hi >> 64);
    m->psi = evo_minimal_primitive_root(N, q);
    if (!m->psi) { evo_ctx_destroy(c); return NULL; }
    m->rp = (u64 *)malloc(sizeof(u64) * N * 4);
    m->rps = m->rp + N;
    m->irp = m->rp + 2 * N;
    m->irps = m->rp + 3 * N;
    u64 p = 1;
    for (uint32_t j = 0; j < N; j++) {
      uint32_t r = bitrev(j, c->logN);
      m->rp[r] = p;
      p = evo_mulmod(p, m->psi, q);
    }
    for (uint32_t j = 0; j < N; j++) {
      m->rps[j] = shoup(m->rp[j], q);
      m->irp[j] = evo_invmod(m->rp[j], q);
      m->irps[j] = shoup(m->irp[j], q);
    }
    m->ninv = evo_invmod(N % q, q);
    m->ninvs = shoup(m->ninv, q);
  }
  return c;
}

void evo_ctx_destroy(evo_ctx *c) {
  if (!c) return;
  for (uint32_t i = 0; i < c->k; i++) free(c->m[i].rp);
  free(c->m);
  free(c);
}
uint32_t evo_ctx_n(const evo_ctx *c) { return c->N; }
uint32_t evo_ctx_k(const evo_ctx *c) { return c->k; }
uint64_t evo_ctx_prime(const evo_ctx *c, uint32_t i) { return c->m[i].q; }
uint64_t evo_ctx_psi(const evo_ctx *c, uint32_t i) { return c->m[i].psi; }
const uint64_t *evo_ctx_root_powers(const evo_ctx *c, uint32_t i) { return c->m[i].rp; }
const uint64_t *evo_ctx_inv_root_powers(const evo_ctx *c, uint32_t i) { return c->m[i].irp; }

/* SEAL barrett_reduce_128 (util/uintarithsmallmod.h) */
static inline u64 barrett128(u128 x, const evo_mod *m) {
  u64 x0 = (u64)x, x1 = (u64)(x >> 64);
  u64 carry = (u64)(((u128)x0 * m->ratio0) >> 64);
  u128 t = (u128)x0 * m->ratio1;
  u64 tmp1 = (u64)t + carry;
  u64 tmp3 = (u64)(t >> 64) + (tmp1 < carry);
  t = (u128)x1 * m->ratio0;
  u64 tmp1b = tmp1 + (u64)t;
  carry = (u64)(t >> 64) + (tmp1b < tmp1);
  u64 qhat = x1 * m->ratio1 + tmp3 + carry;
  u64 r = x0 - qhat * m->q;
  while (r >= m->q) r -= m->q;
  return r;
}
static inline u64 mulm(u64 a, u64 b, const evo_mod *m) { return barrett128((u128)a * b, m); }
static inline u64 addm(u64 a, u64 b, u64 q) { u64 s = a + b; return s >= q ? s - q : s; }
static inline u64 subm(u64 a, u64 b, u64 q) { return a >= b ? a - b : a + q - b; }
static inline u64 negm(u64 a, u64 q) { return a ? q - a : 0; }
/* x*w mod q in [0,2q) given ws = floor(w*2^64/q) (Shoup / Harvey) */
static inline u64 mul_shoup_lazy(u64 x, u64 w, u64 ws, u64 q) {
  u64 h = (u64)(((u128)x * ws) >> 64);
  return x * w - h * q;
}

/* SEAL ntt_negacyclic_harvey: Cooley-Tukey, natural in, bit-reversed out, twiddle of stage m,
 * group i is root_powers[m+i]; Harvey lazy butterflies in [0,4q), canonical at the end. */
static void ntt_fwd(const evo_ctx *c, const evo_mod *m, u64 *x) {
  const u64 q = m->q, q2 = 2 * q;
  uint32_t N = c->N;
  for (uint32_t mm = 1, gap = N >> 1; mm < N; mm <<= 1, gap >>= 1) {
    for (uint32_t i = 0; i < mm; i++) {
      const u64 w = m->rp[mm + i], ws = m->rps[mm + i];
      u64 *a = x + 2 * (size_t)i * gap, *b = a + gap;
      for (uint32_t j = 0; j < gap; j++) {
        u64 X = a[j];
        X -= (X >= q2) ? q2 : 0;
        u64 T = mul_shoup_lazy(b[j], w, ws, q);
        a[j] = X + T;
        b[j] = X + q2 - T;
      }
    }
  }
  for (uint32_t j = 0; j < N; j++) {
    u64 v = x[j];
    v -= (v >= q2) ? q2 : 0;
    v -= (v >= q) ? q : 0;
    x[j] = v;
  }
}

/* SEAL inverse_ntt_negacyclic_harvey: Gentleman-Sande, bit-reversed in, natural out, N^-1. */
static void ntt_inv(const evo_ctx *c, const evo_mod *m, u64 *x) {
  const u64 q = m->q, q2 = 2 * q;
  uint32_t N = c->N;
  for (uint32_t mm = N >> 1, gap = 1; mm >= 1; mm >>= 1, gap <<= 1) {
    for (uint32_t i = 0; i < mm; i++) {
      const u64 w = m->irp[mm + i], ws = m->irps[mm + i];
      u64 *a = x + 2 * (size_t)i * gap, *b = a + gap;
      for (uint32_t j = 0; j < gap; j++) {
        u64 X = a[j], Y = b[j]; /* both in [0,2q) */
        u64 S = X + Y;
        S -= (S >= q2) ? q2 : 0;
        a[j] = S;
        b[j] = mul_shoup_lazy(X + q2 - Y, w, ws, q);
      }
    }
  }
  for (uint32_t j = 0; j < N; j++) {
    u64 v = mul_shoup_lazy(x[j], m->ninv, m->ninvs, q);
    x[j] = v >= q ? v - q : v;
  }
}

void evo_ntt(const evo_ctx *c, uint32_t pi, uint64_t *x) { ntt_fwd(c, &c->m[pi], x); }
void evo_intt(const evo_ctx *c, uint32_t pi, uint64_t *x) { ntt_inv(c, &c->m[pi], x); }

/* ------------------------------------------------------------------ evaluator */

#define POLY(p, l, N) ((size_t)(p) * (l) * (N))

/* SEAL Evaluator::add_inplace: common polys added; extra polys of the longer one copied. */
void evo_add(const evo_ctx *c, uint32_t l, const uint64_t *a, uint32_t sa, const uint64_t *b,
             uint32_t sb, uint64_t *out) {
  uint32_t N = c->N, smin = sa < sb ? sa : sb, smax = sa < sb ? sb : sa;
  for (uint32_t p = 0; p < smax; p++)
    for (uint32_t i = 0; i < l; i++) {
      u64 q = c->m[i].q;
      size_t o = POLY(p, l, N) + (size_t)i * N;
      for (uint32_t j = 0; j < N; j++)
        out[o + j] = p < smin ? addm(a[o + j], b[o + j], q) : (sa > sb ? a[o + j] : b[o + j]);
    }
}

/* SEAL Evaluator::sub_inplace: extra polys of b are negated, of a copied. */
void evo_sub(const evo_ctx *c, uint32_t l, const uint64_t *a, uint32_t sa, const uint64_t *b,
             uint32_t sb, uint64_t *out) {
  uint32_t N = c->N, smin = sa < sb ? sa : sb, smax = sa < sb ? sb : sa;
  for (uint32_t p = 0; p < smax; p++)
    for (uint32_t i = 0; i < l; i++) {
      u64 q = c->m[i].q;
      size_t o = POLY(p, l, N) + (size_t)i * N;
      for (uint32_t j = 0; j < N; j++)
        out[o + j] = p < smin ? subm(a[o + j], b[o + j], q)
                              : (sa > sb ? a[o + j] : negm(b[o + j], q));
    }
}

void evo_add_plain(const evo_ctx *c, uint32_t l, const uint64_t *a, uint32_t sa,
                   const uint64_t *pt, uint64_t *out) {
  uint32_t N = c->N;
  memmove(out, a, sizeof(u64) * POLY(sa, l, N));
  for (uint32_t i = 0; i < l; i++)
    for (uint32_t j = 0; j < N; j++) {
      size_t o = (size_t)i * N + j;
      out[o] = addm(a[o], pt[o], c->m[i].q);
    }
}

void evo_sub_plain(const evo_ctx *c, uint32_t l, const uint64_t *a, uint32_t sa,
                   const uint64_t *pt, uint64_t *out) {
  uint32_t N = c->N;
  memmove(out, a, sizeof(u64) * POLY(sa, l, N));
  for (uint32_t i = 0; i < l; i++)
    for (uint32_t j = 0; j < N; j++) {
      size_t o = (size_t)i * N + j;
      out[o] = subm(a[o], pt[o], c->m[i].q);
    }
}

void evo_negate(const evo_ctx *c, uint32_t l, const uint64_t *a, uint32_t sa, uint64_t *out) {
  uint32_t N = c->N;
  for (uint32_t p = 0; p < sa; p++)
    for (uint32_t i = 0; i < l; i++)
      for (uint32_t j = 0; j < N; j++) {
        size_t o = POLY(p, l, N) + (size_t)i * N + j;
        out[o] = negm(a[o], c->m[i].q);
      }
}

/* SEAL Evaluator::ckks_multiply for 2x2: (a0b0, a0b1+a1b0, a1b1), dyadic in the NTT domain */
void evo_multiply(const evo_ctx *c, uint32_t l, const uint64_t *a, const uint64_t *b,
                  uint64_t *out3) {
  uint32_t N = c->N;
  size_t P = (size_t)l * N;
  for (uint32_t i = 0; i < l; i++) {
    const evo_mod *m = &c->m[i];
    for (uint32_t j = 0; j < N; j++) {
      size_t o = (size_t)i * N + j;
      u64 a0 = a[o], a1 = a[P + o], b0 = b[o], b1 = b[P + o];
      u64 d0 = mulm(a0, b0, m);
      u64 d1 = addm(mulm(a0, b1, m), mulm(a1, b0, m), m->q);
      u64 d2 = mulm(a1, b1, m);
      out3[o] = d0;
      out3[P + o] = d1;
      out3[2 * P + o] = d2;
    }
  }
}

/* SEAL Evaluator::ckks_square: (a0^2, 2 a0 a1, a1^2) */
void evo_square(const evo_ctx *c, uint32_t l, const uint64_t *a, uint64_t *out3) {
  uint32_t N = c->N;
  size_t P = (size_t)l * N;
  for (uint32_t i = 0; i < l; i++) {
    const evo_mod *m = &c->m[i];
    for (uint32_t j = 0; j < N; j++) {
      size_t o = (size_t)i * N + j;
      u64 a0 = a[o], a1 = a[P + o];
      u64 x = mulm(a0, a1, m);
      u64 d0 = mulm(a0, a0, m), d2 = mulm(a1, a1, m);
      out3[o] = d0;
      out3[P + o] = addm(x, x, m->q);
      out3[2 * P + o] = d2;
    }
  }
}

void evo_multiply_plain(const evo_ctx *c, uint32_t l, const uint64_t *a, uint32_t sa,
                        const uint64_t *pt, uint64_t *out) {
  uint32_t N = c->N;
  for (uint32_t p = 0; p < sa; p++)
    for (uint32_t i = 0; i < l; i++)
      for (uint32_t j = 0; j < N; j++) {
        size_t o = (size_t)i * N + j;
        out[POLY(p, l, N) + o] = mulm(a[POLY(p, l, N) + o], pt[o], &c->m[i]);
      }
}

/* SEAL RNSTool::divide_and_round_q_last_ntt_inplace: per poly
 * t = INTT(c[last]); t += q_last/2 mod q_last; per remaining limb i:
 * u = (t mod q_i) - (q_last/2 mod q_i); c'[i] = (c[i] - NTT(u)) * q_last^-1 mod q_i. */
void evo_rescale(const evo_ctx *c, uint32_t l, const uint64_t *a, uint32_t sa, uint64_t *out) {
  uint32_t N = c->N;
  const evo_mod *ml = &c->m[l - 1];
  u64 half = ml->q >> 1;
  u64 *t = (u64 *)malloc(sizeof(u64) * N * 2), *u = t + N;
  for (uint32_t p = 0; p < sa; p++) {
    memcpy(t, a + POLY(p, l, N) + (size_t)(l - 1) * N, sizeof(u64) * N);
    ntt_inv(c, ml, t);
    for (uint32_t j = 0; j < N; j++) t[j] = addm(t[j], half, ml->q);
    for (uint32_t i = 0; i + 1 < l; i++) {
      const evo_mod *m = &c->m[i];
      u64 halfi = half % m->q, inv = evo_invmod(ml->q % m->q, m->q);
      for (uint32_t j = 0; j < N; j++) u[j] = subm(t[j] % m->q, halfi, m->q);
      ntt_fwd(c, m, u);
      const u64 *src = a + POLY(p, l, N) + (size_t)i * N;
      u64 *dst = out + POLY(p, l - 1, N) + (size_t)i * N;
      for (uint32_t j = 0; j < N; j++) dst[j] = mulm(subm(src[j], u[j], m->q), inv, m);
    }
  }
  free(t);
}

void evo_mod_switch(const evo_ctx *c, uint32_t l, const uint64_t *a, uint32_t sa, uint64_t *out) {
  uint32_t N = c->N;
  for (uint32_t p = 0; p < sa; p++)
    memmove(out + POLY(p, l - 1, N), a + POLY(p, l, N), sizeof(u64) * (size_t)(l - 1) * N);
}

/* SEAL Evaluator::switch_key_inplace.  key[J][K][i][N], J < k-1
 * digits, K in {0,1}, i over all k key primes. */
void evo_switch_key(const evo_ctx *c, uint32_t l, uint64_t *ct2, const uint64_t *target,
                    const uint64_t *key) {
  uint32_t N = c->N, k = c->k;
  const evo_mod *mp = &c->m[k - 1]; /* special prime P */
  size_t keyJ = (size_t)2 * k * N;  /* stride between digits */
  u64 *t = (u64 *)malloc(sizeof(u64) * (size_t)N * (l + 1));
  u64 *tmp = t + (size_t)l * N;
  u64 *prod = (u64 *)malloc(sizeof(u64) * (size_t)2 * (l + 1) * N); /* [K][I<=l][N] */
  u128 *acc = (u128 *)malloc(sizeof(u128) * (size_t)2 * N);
  /* 1. digits back to coefficient form */
  memcpy(t, target, sizeof(u64) * (size_t)l * N);
  for (uint32_t J = 0; J < l; J++) ntt_inv(c, &c->m[J], t + (size_t)J * N);
  /* 2. inner product with the key over every output limb I (data limbs + special) */
  for (uint32_t I = 0; I <= l; I++) {
    uint32_t ki = (I == l) ? k - 1 : I;
    const evo_mod *m = &c->m[ki];
    memset(acc, 0, sizeof(u128) * (size_t)2 * N);
    for (uint32_t J = 0; J < l; J++) {
      const u64 *op;
      if (I == J) {
        op = target + (size_t)J * N; /* already NTT form mod q_J */
      } else {
        const u64 *src = t + (size_t)J * N;
        if (c->m[J].q <= m->q) memcpy(tmp, src, sizeof(u64) * N);
        else for (uint32_t j = 0; j < N; j++) tmp[j] = src[j] % m->q;
        ntt_fwd(c, m, tmp);
        op = tmp;
      }
      const u64 *k0 = key + J * keyJ + (size_t)ki * N;
      const u64 *k1 = k0 + (size_t)k * N;
      for (uint32_t j = 0; j < N; j++) {
        acc[j] += (u128)op[j] * k0[j];
        acc[N + j] += (u128)op[j] * k1[j];
      }
    }
    for (uint32_t K = 0; K < 2; K++)
      for (uint32_t j = 0; j < N; j++)
        prod[((size_t)K * (l + 1) + I) * N + j] = barrett128(acc[(size_t)K * N + j], m);
  }
  /* 3. mod-down by P with rounding, add into ct */
  u64 half = mp->q >> 1;
  for (uint32_t K = 0; K < 2; K++) {
    u64 *r = prod + ((size_t)K * (l + 1) + l) * N;
    ntt_inv(c, mp, r);
    for (uint32_t j = 0; j < N; j++) r[j] = addm(r[j], half, mp->q);
    for (uint32_t J = 0; J < l; J++) {
      const evo_mod *m = &c->m[J];
      u64 halfJ = half % m->q, inv = evo_invmod(mp->q % m->q, m->q);
      for (uint32_t j = 0; j < N; j++) tmp[j] = subm(r[j] % m->q, halfJ, m->q);
      ntt_fwd(c, m, tmp);
      const u64 *pj = prod + ((size_t)K * (l + 1) + J) * N;
      u64 *dst = ct2 + POLY(K, l, N) + (size_t)J * N;
      for (uint32_t j = 0; j < N; j++)
        dst[j] = addm(dst[j], mulm(subm(pj[j], tmp[j], m->q), inv, m), m->q);
    }
  }
  free(acc);
  free(prod);
  free(t);
}

/* SEAL Evaluator::relinearize_internal for size 3 -> 2 */
void evo_relinearize(const evo_ctx *c, uint32_t l, const uint64_t *a3, const uint64_t *relin_key,
                     uint64_t *out2) {
  size_t P = (size_t)l * c->N;
  memmove(out2, a3, sizeof(u64) * 2 * P);
  evo_switch_key(c, l, out2, a3 + 2 * P, relin_key);
}

/* SEAL GaloisTool::get_elt_from_step (steps != 0) */
uint32_t evo_galois_elt_from_step(uint32_t N, int32_t steps) {
  uint32_t m = 2 * N;
  if (steps == 0) return m - 1;
  uint32_t pos = steps < 0 ? (uint32_t)(-steps) : (uint32_t)steps;
  if (pos >= (N >> 1)) return 0;
  uint32_t s = steps < 0 ? (N >> 1) - pos : pos;
  uint32_t elt = 1;
  for (uint32_t i = 0; i < s; i++) elt = (elt * 3u) & (m - 1);
  return elt;
}

/* SEAL GaloisTool::generate_table_ntt */
void evo_galois_table(uint32_t N, uint32_t elt, uint32_t *table) {
  uint32_t logN = 0;
  while ((1u << logN) < N) logN++;
  for (uint32_t i = 0; i < N; i++) {
    uint32_t reversed = bitrev(N + i, logN + 1); /* = 2*br(i)+1 */
    u64 raw = ((u64)elt * reversed) >> 1;
    raw &= (u64)(N - 1);
    table[i] = bitrev((uint32_t)raw, logN);
  }
}

/* SEAL Evaluator::rotate_internal + apply_galois_inplace (CKKS branch) */
void evo_rotate(const evo_ctx *c, uint32_t l, const uint64_t *a2, int32_t steps,
                const uint64_t *galois_key, uint64_t *out2) {
  uint32_t N = c->N;
  size_t P = (size_t)l * N;
  if (steps == 0) { memmove(out2, a2, sizeof(u64) * 2 * P); return; }
  uint32_t elt = evo_galois_elt_from_step(N, steps);
  uint32_t *tab = (uint32_t *)malloc(sizeof(uint32_t) * N);
  u64 *tgt = (u64 *)malloc(sizeof(u64) * P);
  evo_galois_table(N, elt, tab);
  for (uint32_t i = 0; i < l; i++)
    for (uint32_t j = 0; j < N; j++) {
      tgt[(size_t)i * N + j] = a2[P + (size_t)i * N + tab[j]];
    }
  /* out may alias a2: permute c0 through a temporary */
  u64 *c0 = (u64 *)malloc(sizeof(u64) * P);
  for (uint32_t i = 0; i < l; i++)
    for (uint32_t j = 0; j < N; j++) c0[(size_t)i * N + j] = a2[(size_t)i * N + tab[j]];
  memcpy(out2, c0, sizeof(u64) * P);
  memset(out2 + P, 0, sizeof(u64) * P);
  evo_switch_key(c, l, out2, tgt, galois_key);
  free(c0);
  free(tgt);
  free(tab);
}

/* ------------------------------------------------------------------ CKKS encoder */

/* The encoder is FP64 code, so bits depend on the ORDER of floating-point operations.  What
 * follows restates, routine by routine, Microsoft SEAL v3.6.x (absent from /root/reference; the
 * call site is /root/reference/eva/seal/seal_executor.h:242 `encoder.encode(vec, parms_id,
 * pow(2.0, scale), plaintext)`):
 *   - util::ComplexRoots (native/src/seal/util/croots.cpp): one octant of the 2N-th roots from
 *     std::polar(1.0, 2*PI_*i/2N), the rest by the 8-fold symmetry (get_root);
 *   - CKKSEncoder::CKKSEncoder (ckks.cpp): matrix_reps_index_map_, and
 *     inv_root_powers_[i] = conj(get_root(reverse_bits(i-1, logn) + 1)), i = 1..n-1, which
 *     DWTHandler consumes SEQUENTIALLY (`*++roots`);
 *   - util::DWTHandler::transform_from_rev (util/dwthandler.h) with the Arithmetic<complex<double>,
 *     complex<double>, double> of ckks.h: Gentleman-Sande stages x = u+v, y = (u-v)*r, and the
 *     LAST stage with the scalar fix = scale/n folded in: x = (u+v)*fix, y = (u-v)*(r*fix);
 *   - CKKSEncoder::encode_internal (ckks.h): round (ties away), sign, residues, per-limb NTT.
 * A complex product is evaluated as libgcc's __muldc3 does for finite operands — (a.re*b.re -
 * a.im*b.im, a.re*b.im + a.im*b.re), each product rounded (no FMA: SEAL's stock build targets
 * baseline x86-64; this file is compiled with -ffp-contract=off) — and complex*double scales both
 * parts.  cos/sin come from the platform libm exactly as std::polar takes them. */
typedef struct { double re, im; } cplx;
static const double SEAL_PI = 3.1415926535897932384626433832795028842; /* ComplexRoots::PI_ */

static inline cplx c_add(cplx a, cplx b) { return (cplx){a.re + b.re, a.im + b.im}; }
static inline cplx c_sub(cplx a, cplx b) { return (cplx){a.re - b.re, a.im - b.im}; }
static inline cplx c_mul(cplx a, cplx b) {
  double ac = a.re * b.re, bd = a.im * b.im, ad = a.re * b.im, bc = a.im * b.re;
  return (cplx){ac - bd, ad + bc};
}
static inline cplx c_scale(cplx a, double s) { return (cplx){a.re * s, a.im * s}; }

/* ComplexRoots::get_root over the stored octant oct[0 .. degree/8] */
static cplx croots_get(const cplx *oct, size_t degree, size_t index) {
  index &= degree - 1;
  if (index <= degree / 8) return oct[index];
  if (index <= degree / 4) { /* mirror: swap real and imaginary parts */
    cplx t = oct[degree / 4 - index];
    return (cplx){t.im, t.re};
  }
  if (index <= degree / 2) { /* -conj */
    cplx t = croots_get(oct, degree, degree / 2 - index);
    return (cplx){-t.re, t.im};
  }
  if (index <= 3 * degree / 4) { /* negation */
    cplx t = croots_get(oct, degree, index - degree / 2);
    return (cplx){-t.re, -t.im};
  }
  cplx t = croots_get(oct, degree, degree - index); /* conj */
  return (cplx){t.re, -t.im};
}

/* std::polar(1.0, theta) is (cos(theta), sin(theta)) from libm.  GCC fuses such a pair into ONE
 * sincos() call, and glibc 2.35's sincos() is not bit-identical to sin(): at 2N = 8192 the octant
 * angle i = 487 differs in the last place.  The reference recommends building SEAL with clang
 * (/root/reference/README.md:21-25), which keeps the two separate calls; calling through volatile
 * pointers keeps them separate under any compiler, so oracle, host and device tables agree. */
static double (*volatile libm_cos)(double) = cos;
static double (*volatile libm_sin)(double) = sin;

/* inv_root_powers_ of CKKSEncoder for degree N (entry 0 unused), caller frees */
static cplx *ckks_inv_root_powers(uint32_t N, uint32_t logN) {
  size_t degree = (size_t)2 * N;
  cplx *oct = (cplx *)malloc(sizeof(cplx) * (degree / 8 + 1));
  for (size_t i = 0; i <= degree / 8; i++) {
    double theta = 2 * SEAL_PI * (double)i / (double)degree;
    oct[i] = (cplx){libm_cos(theta), libm_sin(theta)}; /* std::polar(1.0, theta) */
  }
  cplx *inv = (cplx *)malloc(sizeof(cplx) * N);
  inv[0] = (cplx){0, 0};
  for (uint32_t i = 1; i < N; i++) {
    cplx r = croots_get(oct, degree, (size_t)bitrev(i - 1, logN) + 1);
    inv[i] = (cplx){r.re, -r.im};
  }
  free(oct);
  return inv;
}

/* CKKSEncoder::encode_internal up to the rounded real coefficients.
 * values: N/2 slots (EVA replicates the vector first, seal_executor.h:226-240). */
static void encode_reals(uint32_t N, const double *values, double scale, double *coeffs, int rounded) {
  uint32_t logN = 0, slots = N >> 1, m = 2 * N;
  while ((1u << logN) < N) logN++;
  cplx *v = (cplx *)calloc(N, sizeof(cplx));
  u64 pos = 1;
  for (uint32_t i = 0; i < slots; i++) { /* matrix_reps_index_map_ */
    uint32_t i1 = bitrev((uint32_t)((pos - 1) >> 1), logN);
    uint32_t i2 = bitrev((uint32_t)((m - pos - 1) >> 1), logN);
    v[i1] = (cplx){values[i], 0.0};
    v[i2] = (cplx){values[i], -0.0}; /* std::conj of a real value */
    pos = (pos * 3) & (m - 1);
  }
  cplx *inv = ckks_inv_root_powers(N, logN);
  const cplx *roots = inv;
  const double fix = scale / (double)N;
  /* transform_from_rev(values, log_n, roots, &fix) */
  size_t gap = 1, mm = N >> 1;
  for (; mm > 1; mm >>= 1) {
    size_t offset = 0;
    for (size_t i = 0; i < mm; i++) {
      cplx r = *++roots;
      cplx *x = v + offset, *y = x + gap;
      for (size_t j = 0; j < gap; j++) {
        cplx u = *x, w = *y;
        *x++ = c_add(u, w);
        *y++ = c_mul(c_sub(u, w), r);
      }
      offset += gap << 1;
    }
    gap <<= 1;
  }
  {
    cplx r = *++roots;
    cplx scaled_r = c_scale(r, fix); /* mul_root_scalar */
    cplx *x = v, *y = v + gap;
    for (size_t j = 0; j < gap; j++) {
      cplx u = *x, w = *y;
      *x++ = c_scale(c_add(u, w), fix);
      *y++ = c_mul(c_sub(u, w), scaled_r);
    }
  }
  for (uint32_t j = 0; j < N; j++) coeffs[j] = rounded ? round(v[j].re) : v[j].re;
  free(inv);
  free(v);
}
void evo_encode_coeffs(uint32_t N, const double *values, double scale, double *coeffs) {
  encode_reals(N, values, scale, coeffs, 1);
}

/* rest of encode_internal: |coefficient| -> base-2^64 words (exact: the double is an integer)
 * -> residue per prime, negated for negative coefficients; then ntt_negacyclic_harvey per limb.
 * returns -1 for "encoded values are too large" (max_coeff_bit_count >= total modulus bits). */
int evo_encode(const evo_ctx *c, uint32_t l, const double *values, double scale, uint64_t *pt) {
  uint32_t N = c->N;
  double *co = (double *)malloc(sizeof(double) * N);
  encode_reals(N, values, scale, co, 0);
  double max_coeff = 0; /* over the unrounded real parts, as encode_internal does */
  for (uint32_t j = 0; j < N; j++) max_coeff = fmax(max_coeff, fabs(co[j]));
  for (uint32_t j = 0; j < N; j++) co[j] = round(co[j]);
  /* total_coeff_modulus_bit_count at this level */
  int total_bits = 0;
  {
    u64 w[64] = {1};
    int nw = 1;
    for (uint32_t i = 0; i < l; i++) {
      u64 carry = 0;
      for (int t = 0; t < nw; t++) {
        u128 p = (u128)w[t] * c->m[i].q + carry;
        w[t] = (u64)p;
        carry = (u64)(p >> 64);
      }
      if (carry) w[nw++] = carry;
    }
    total_bits = (nw - 1) * 64;
    for (u64 top = w[nw - 1]; top; top >>= 1) total_bits++;
  }
  int bitcount = (int)ceil(log2(fmax(max_coeff, 1.0))) + 1;
  int rc = bitcount >= total_bits ? -1 : 0;
  for (uint32_t j = 0; j < N && rc == 0; j++) {
    int neg = signbit(co[j]);
    double a = fabs(co[j]);
    /* words of the integer a, least significant first (fmod / divide by 2^64 are exact) */
    u64 words[20];
    int nw = 0;
    const double two64 = 18446744073709551616.0;
    while (a >= 1 && nw < 20) {
      words[nw++] = (u64)fmod(a, two64);
      a /= two64;
    }
    for (uint32_t i = 0; i < l; i++) {
      u64 q = c->m[i].q, r = 0;
      for (int t = nw - 1; t >= 0; t--) r = (u64)((((u128)r << 64) | words[t]) % q);
      pt[(size_t)i * N + j] = neg ? negm(r, q) : r;
    }
  }
  if (rc == 0)
    for (uint32_t i = 0; i < l; i++) ntt_fwd(c, &c->m[i], pt + (size_t)i * N);
  free(co);
  return rc;
}

/* ------------------------------------------------------------------ decrypt + CKKS decoder */

/* SEAL 3.6 Decryptor::decrypt for CKKS (decryptor.cpp: ckks_decrypt -> dot_product_ct_sk_array), the
 * call at /root/reference/eva/seal/seal.cpp:132-133: pt = c0 + c1 s (+ c2 s^2) per limb, NTT form
 * in and out, scale unchanged.  Canonical residues, so the evaluation order is free.
 * ct: [size][l][N], sk_ntt: [k][N] (the secret key under every key prime, NTT form), pt: [l][N]. */
void evo_decrypt(const evo_ctx *c, uint32_t l, uint32_t size, const uint64_t *ct, const uint64_t *sk_ntt, uint64_t *pt) {
  const uint32_t N = c->N;
  for (uint32_t i = 0; i < l; i++) {
    const evo_mod *m = &c->m[i];
    const u64 *s = sk_ntt + (size_t)i * N;
    for (uint32_t j = 0; j < N; j++) {
      u64 acc = ct[(size_t)i * N + j], sp = s[j];
      for (uint32_t p = 1; p < size; p++) {
        acc = addm(acc, mulm(ct[((size_t)p * l + i) * N + j], sp, m), m->q);
        sp = mulm(sp, s[j], m);
      }
      pt[(size_t)i * N + j] = acc;
    }
  }
}

/* little-endian multi-word helpers for the CRT composition (n <= 64 words) */
static void mp_mul_word(const u64 *a, int n, u64 w, u64 *out /* n + 1 words */) {
  u64 carry = 0;
  for (int t = 0; t < n; t++) {
    u128 p = (u128)a[t] * w + carry;
    out[t] = (u64)p;
    carry = (u64)(p >> 64);
  }
  out[n] = carry;
}
static int mp_cmp(const u64 *a, const u64 *b, int n) {
  for (int t = n - 1; t >= 0; t--)
    if (a[t] != b[t]) return a[t] > b[t] ? 1 : -1;
  return 0;
}
static void mp_sub(u64 *a, const u64 *b, int n) { /* a -= b */
  u64 borrow = 0;
  for (int t = 0; t < n; t++) {
    u128 d = (u128)a[t] - b[t] - borrow;
    a[t] = (u64)d;
    borrow = (u64)(d >> 64) & 1;
  }
}
static void mp_add(u64 *a, const u64 *b, int n) { /* a += b, carry out dropped by the caller's sizing */
  u64 carry = 0;
  for (int t = 0; t < n; t++) {
    u128 d = (u128)a[t] + b[t] + carry;
    a[t] = (u64)d;
    carry = (u64)(d >> 64);
  }
}

/* root_powers_ of CKKSEncoder (ckks.cpp): root_powers_[i] = get_root(reverse_bits(i, logn)), i = 1..n-1 */
static cplx *ckks_root_powers(uint32_t N, uint32_t logN) {
  size_t degree = (size_t)2 * N;
  cplx *oct = (cplx *)malloc(sizeof(cplx) * (degree / 8 + 1));
  for (size_t i = 0; i <= degree / 8; i++) {
    double theta = 2 * SEAL_PI * (double)i / (double)degree;
    oct[i] = (cplx){libm_cos(theta), libm_sin(theta)};
  }
  cplx *rp = (cplx *)malloc(sizeof(cplx) * N);
  rp[0] = (cplx){0, 0};
  for (uint32_t i = 1; i < N; i++) rp[i] = croots_get(oct, degree, (size_t)bitrev(i, logN));
  free(oct);
  return rp;
}

/* SEAL 3.6 CKKSEncoder::decode_internal (ckks.h), the call at /root/reference/eva/seal/seal.cpp:134-135
 * (`encoder.decode(plain, vec)`), routine by routine:
 *   - per-limb inverse_ntt_negacyclic_harvey;
 *   - RNSBase::compose_array: every coefficient as a base-2^64 integer x in [0, Q), l words;
 *   - the words to ONE double, least significant first, with inv_scale = 1.0 / scale folded into the
 *     running power of 2^64 (res += (double)word * scaled; scaled *= 2^64), zero words skipped; a
 *     coefficient >= upper_half_threshold = (Q + 1) / 2 is negative and is accumulated as the signed
 *     per-word differences against the words of Q (word > Q_j: += (word - Q_j) * scaled, else
 *     -= (Q_j - word) * scaled) — exactly SEAL's loop, which is NOT a multi-word subtraction;
 *   - util::DWTHandler::transform_to_rev with root_powers_ (Cooley-Tukey: x = u + v r, y = u - v r,
 *     roots consumed sequentially), complex product as in the encoder above (no FMA);
 *   - slot i = real part of res[matrix_reps_index_map_[i]].
 * pt: [l][N] NTT form; out: N/2 slot values.  returns 0, or -1 for "scale out of bounds". */
int evo_decode(const evo_ctx *c, uint32_t l, const uint64_t *pt, double scale, double *out) {
  const uint32_t N = c->N, slots = N >> 1, m2 = 2 * N;
  uint32_t logN = 0;
  while ((1u << logN) < N) logN++;
  if (l < 1 || l > 62) return -1;
  /* Q and the punctured products Q / q_i (l words each), (Q / q_i)^-1 mod q_i */
  u64 Q[64] = {1}, tmp[65];
  int nw = 1;
  for (uint32_t i = 0; i < l; i++) {
    mp_mul_word(Q, nw, c->m[i].q, tmp);
    nw++;
    memcpy(Q, tmp, sizeof(u64) * nw);
  }
  while (nw > 1 && Q[nw - 1] == 0) nw--;
  int total_bits = (nw - 1) * 64;
  for (u64 top = Q[nw - 1]; top; top >>= 1) total_bits++;
  if (!(scale > 0) || (int)log2(scale) >= total_bits) return -1;
  const int L = (int)l; /* coeff_modulus_size words per composed coefficient, as SEAL lays them out */
  u64 *punct = (u64 *)calloc((size_t)l * (L + 1), sizeof(u64)), *inv_punct = (u64 *)malloc(sizeof(u64) * l);
  for (uint32_t i = 0; i < l; i++) {
    u64 *pp = punct + (size_t)i * (L + 1);
    pp[0] = 1;
    int n = 1;
    for (uint32_t j = 0; j < l; j++) {
      if (j == i) continue;
      mp_mul_word(pp, n, c->m[j].q, tmp);
      n++;
      memcpy(pp, tmp, sizeof(u64) * (n < L + 1 ? n : L + 1));
    }
    u64 r = 0; /* (Q / q_i) mod q_i */
    for (int t = L - 1; t >= 0; t--) r = (u64)((((u128)r << 64) | pp[t]) % c->m[i].q);
    inv_punct[i] = evo_invmod(r, c->m[i].q);
  }
  u64 Qw[64] = {0}, half[64] = {0}; /* Q and (Q + 1) >> 1 on L words */
  memcpy(Qw, Q, sizeof(u64) * (size_t)(nw < L ? nw : L));
  {
    u64 t1[64];
    memcpy(t1, Qw, sizeof(t1));
    u64 one[64] = {1};
    mp_add(t1, one, L);
    for (int t = 0; t < L; t++) half[t] = (t1[t] >> 1) | (t + 1 < L ? t1[t + 1] << 63 : 0);
  }
  u64 *co = (u64 *)malloc(sizeof(u64) * (size_t)l * N);
  memcpy(co, pt, sizeof(u64) * (size_t)l * N);
  for (uint32_t i = 0; i < l; i++) ntt_inv(c, &c->m[i], co + (size_t)i * N);
  cplx *res = (cplx *)malloc(sizeof(cplx) * N);
  const double inv_scale = 1.0 / scale, two_pow_64 = pow(2.0, 64);
  for (uint32_t n = 0; n < N; n++) {
    u64 x[65] = {0}, term[66];
    for (uint32_t i = 0; i < l; i++) { /* x = sum_i [r_i (Q/q_i)^-1 mod q_i] (Q/q_i) mod Q */
      const u64 tp = mulm(co[(size_t)i * N + n], inv_punct[i], &c->m[i]);
      mp_mul_word(punct + (size_t)i * (L + 1), L, tp, term); /* < Q, fits L words */
      mp_add(x, term, L + 1);
      if (x[L] || mp_cmp(x, Qw, L) >= 0) { mp_sub(x, Qw, L); x[L] = 0; } /* add_uint_uint_mod */
    }
    double acc = 0.0, scaled = inv_scale;
    if (mp_cmp(x, half, L) >= 0) {
      for (int j = 0; j < L; j++, scaled *= two_pow_64) {
        if (x[j] > Qw[j]) {
          const u64 diff = x[j] - Qw[j];
          acc += diff ? (double)diff * scaled : 0.0;
        } else {
          const u64 diff = Qw[j] - x[j];
          acc -= diff ? (double)diff * scaled : 0.0;
        }
      }
    } else {
      for (int j = 0; j < L; j++, scaled *= two_pow_64) acc += x[j] ? (double)x[j] * scaled : 0.0;
    }
    res[n] = (cplx){acc, 0.0};
  }
  /* transform_to_rev(res, logn, root_powers_) */
  cplx *rp = ckks_root_powers(N, logN);
  const cplx *roots = rp;
  size_t gap = N >> 1, mm = 1;
  for (; mm < N; mm <<= 1) { /* the unrolled last stage (gap = 1) of SEAL is the same butterfly */
    size_t offset = 0;
    for (size_t i = 0; i < mm; i++) {
      cplx r = *++roots;
      cplx *x = res + offset, *y = x + gap;
      for (size_t j = 0; j < gap; j++) {
        cplx u = *x, v = c_mul(*y, r);
        *x++ = c_add(u, v);
        *y++ = c_sub(u, v);
      }
      offset += gap << 1;
    }
    gap >>= 1;
  }
  u64 pos = 1;
  for (uint32_t i = 0; i < slots; i++) { /* matrix_reps_index_map_[i], i < slots */
    out[i] = res[bitrev((uint32_t)((pos - 1) >> 1), logN)].re;
    pos = (pos * 3) & (m2 - 1);
  }
  free(rp);
  free(res);
  free(co);
  free(punct);
  free(inv_punct);
  return 0;
}

/* ------------------------------------------------------------------ bench helper */

void evo_op_triple(const evo_ctx *c, uint32_t l, const uint64_t *a2, const uint64_t *b2,
                   const uint64_t *relin_key, uint64_t *out2) {
  size_t P = (size_t)l * c->N;
  u64 *t3 = (u64 *)malloc(sizeof(u64) * 3 * P), *t2 = (u64 *)malloc(sizeof(u64) * 2 * P);
  evo_multiply(c, l, a2, b2, t3);
  evo_relinearize(c, l, t3, relin_key, t2);
  evo_rescale(c, l, t2, 2, out2);
  free(t3);
  free(t2);
}
