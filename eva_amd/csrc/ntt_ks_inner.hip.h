// ntt_ks_inner.hip.h — ks_inner_kernel: the key-switch inner product fused with the second pass of the digit transforms, and the operand tables of batched launches (KsKeys, PtrTab, MulTab)
// (part of ntt.hip.h until r5; included by it, in the order the definitions depend on each other)
#pragma once
#include "ntt.hip.h"

namespace evah {

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
                uint32_t nout, u64 *__restrict__ r_out, KsAddArg<MODE> adds, int flags, uint32_t fold_row) {
  // flags: bit 0 = lazy_out; bit 1 = diag — the diagonal digit I == J comes from scratch like every other (first-pass
  // intermediate of its forward transform) instead of an NTT-form target (the chain step, ntt_chain.hip.h).
  // fold_row (KS_FOLDMUL / KS_FOLDADD): ~0u adds P d_K; a = fold_row adds (P q_a^-1) d_K (DevCtx::plinv — the chain step's folded rescale)
  const bool lazy_out = flags & 1, diag = flags & 2;
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
      const u64 *src = (I == J && !diag ? target + (size_t)Irow * N : scratch + ((size_t)Irow * l + J) * N) + gbase;
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
    const u64 *src = (I == J && !diag ? target + (size_t)Irow * N : scratch + ((size_t)Irow * l + J) * N) + gbase;
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
      if (I == J && !diag) { // NTT form already: the tile as it lies in `lin`
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
    } else if (I == J && !diag) { // already in NTT form mod q_J: use the key-switch target directly
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
      const ulonglong2 Pm = (fold_row != ~0u) ? cx.plinv[(size_t)fold_row * cx.k + kap]
                                                                     : cx.modq[(size_t)(cx.k - 1) * cx.k + kap]; // (P mod q_I, Shoup quotient)
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

} // namespace evah
