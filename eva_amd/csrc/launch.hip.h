// launch.hip.h — launch plumbing of the transform kernels (ntt.hip.h), shared by the translation
// units of libeva_hip.so that issue transforms: pass selection by size (8 / 4 coefficients per thread,
// the fused inverse + forward form for latency-bound launches), profiling classes, and the entry
// points of the key-switch core that keyswitch.hip defines for rotate.hip and shard.hip.
#pragma once
#include "internal.hip.h"
#include "ntt.hip.h"

namespace evah {

// 2 coefficients per thread (16-byte accesses); grid = (N/512, limbs, polys)
#define EW_SETUP                                                                                 \
  const uint32_t p = blockIdx.z, i = blockIdx.y;                                                 \
  const size_t off = (size_t)i * cx.N + 2 * ((size_t)blockIdx.x * blockDim.x + threadIdx.x);    \
  const DevPrime pm = cx.primes[cx.prime_of(i)];                                                 \
  (void)p;                                                                                        \
  (void)pm;

__device__ __forceinline__ ulonglong2 ld2(const u64 *p) { return *reinterpret_cast<const ulonglong2 *>(p); }
__device__ __forceinline__ void st2(u64 *p, ulonglong2 v) { *reinterpret_cast<ulonglong2 *>(p) = v; }

// ---- NTT launch plumbing
template <class Op> struct OpClass;
template <bool Z, bool G> struct OpClass<OpPlainT<Z, G>> { static constexpr int fwd_a = KC_NTT_A, fwd_b = KC_NTT_B; };
template <> struct OpClass<OpMulIntt> { static constexpr int fwd_a = KC_NTT_A, fwd_b = KC_NTT_B; };
template <> struct OpClass<OpKsDigit> { static constexpr int fwd_a = KC_KSDIGIT_A, fwd_b = KC_KSDIGIT_B; };
template <bool G> struct OpClass<OpModDownT<G>> { static constexpr int fwd_a = KC_MODDOWN_A, fwd_b = KC_MODDOWN_B; };
template <> struct OpClass<OpMulPolyIntt> { static constexpr int fwd_a = KC_NTT_A, fwd_b = KC_NTT_B; };
template <> struct OpClass<OpModDownMul> { static constexpr int fwd_a = KC_MODDOWN_A, fwd_b = KC_MODDOWN_B; };
template <int M> struct OpClass<OpRRT<M>> { static constexpr int fwd_a = KC_MODDOWN_A, fwd_b = KC_MODDOWN_B; };
template <int M> struct OpClass<OpRRLastT<M>> { static constexpr int fwd_a = KC_MODDOWN_A, fwd_b = KC_MODDOWN_B; };
template <> struct OpClass<OpChainIntt> { static constexpr int fwd_a = KC_NTT_A, fwd_b = KC_NTT_B; };
template <> struct OpClass<OpChainT> { static constexpr int fwd_a = KC_INTT_B, fwd_b = KC_INTT_B; };
template <> struct OpClass<OpChainDigit> { static constexpr int fwd_a = KC_KSDIGIT_A, fwd_b = KC_KSDIGIT_B; };
template <> struct OpClass<OpRsMd> { static constexpr int fwd_a = KC_MODDOWN_A, fwd_b = KC_MODDOWN_B; };

template <int P, int LR, bool STRIDED, bool INVERSE, class Op>
static void launch_pass(evah_ctx *c, const typename Op::Params &prm, uint32_t jobs) {
  ProfScope ps(c, INVERSE ? (STRIDED ? KC_INTT_B : KC_INTT_A)
                          : (STRIDED ? OpClass<Op>::fwd_a : OpClass<Op>::fwd_b));
  const uint32_t max_tile = (uint32_t)NTT_THREADS << LR;
  const uint32_t tile = c->N < max_tile ? c->N : max_tile;
  const int logC = (int)ilog2(tile) - P;
  size_t lds = ((((size_t)1 << logC) * lds_sub_stride<P>() + 1) & ~(size_t)1) * sizeof(u64);
  if (STRIDED) lds += ((size_t)1 << P) * sizeof(ulonglong2); // staged twiddles
  lds += c->tun.lds_extra; // occupancy probe (EVAH_LDS_EXTRA), 0 ordinarily
  const uint32_t n_tiles = c->N / tile;
  const int log_tiles = (int)ilog2(n_tiles);
  dim3 grid = Op::grid(prm, jobs), block(tile >> LR);
  grid.x *= n_tiles;
  if (tile == max_tile) {
    hipLaunchKernelGGL((ntt_pass_kernel<P, LR, STRIDED, INVERSE, Op, true>), grid, block, lds, c->stream, c->dev,
                       prm, logC, log_tiles);
  } else if constexpr (P == 5) { // N = 1024: one partial tile per polynomial
    hipLaunchKernelGGL((ntt_pass_kernel<P, LR, STRIDED, INVERSE, Op, false>), grid, block, lds, c->stream, c->dev,
                       prm, logC, log_tiles);
  } else {
    throw std::logic_error("partial NTT tile with P != 5");
  }
  HIPCHK(hipGetLastError());
}

template <int LR, bool STRIDED, bool INVERSE, class Op>
static void launch_pass_lr(evah_ctx *c, int P, const typename Op::Params &prm, uint32_t jobs) {
  switch (P) {
  case 5: launch_pass<5, LR, STRIDED, INVERSE, Op>(c, prm, jobs); break;
  case 6: launch_pass<6, LR, STRIDED, INVERSE, Op>(c, prm, jobs); break;
  case 7: launch_pass<7, LR, STRIDED, INVERSE, Op>(c, prm, jobs); break;
  case 8: launch_pass<8, LR, STRIDED, INVERSE, Op>(c, prm, jobs); break;
  case 9:
    if constexpr (STRIDED) { launch_pass<9, LR, STRIDED, INVERSE, Op>(c, prm, jobs); break; }
    [[fallthrough]];
  default: throw std::runtime_error("unsupported poly_modulus_degree for the NTT kernels");
  }
}
// Contiguous pass as ntt_loop_kernel: one wave per 256-coefficient tile, the tile's twiddle heaps staged in LDS once
// and reused by up to loop_n jobs that share the prime (Tunables::loop_n; 0 = off).  Returns false when the launch
// has fewer than loop_min jobs along the op's loop axis (nothing to share) and the caller takes ntt_pass_kernel.
template <int P, bool INVERSE, class Op>
static bool launch_loop_pass(evah_ctx *c, const typename Op::Params &prm, uint32_t jobs) {
  constexpr int LR = 2;
  dim3 grid = Op::grid(prm, jobs);
  uint32_t ext[3] = {grid.x, grid.y, grid.z};
  const uint32_t count = ext[Op::loop_axis];
  if (!c->tun.loop_n || count < c->tun.loop_min || c->N < 256) return false;
  // a launch that cannot fill the chip is latency-bound: walking jobs one after the other only lengthens it
  if ((uint64_t)ext[0] * ext[1] * ext[2] / count * (c->N / 256) * ((count + 1) / 2) < c->tun.loop_min_wgs) return false;
  ProfScope ps(c, INVERSE ? KC_INTT_A : OpClass<Op>::fwd_b);
  const uint32_t tile = 256, n_tiles = c->N / tile;
  const int logC = 8 - P, log_tiles = (int)ilog2(n_tiles);
  // enough workgroups to fill the chip several times over before twiddle sharing is taken further
  uint32_t nloop = std::min<uint32_t>(c->tun.loop_n, count);
  const uint64_t others = (uint64_t)ext[0] * ext[1] * ext[2] / count * n_tiles;
  while (nloop > 2 && others * ((count + nloop - 1) / nloop) < c->tun.loop_target_wgs) nloop = (nloop + 1) / 2;
  ext[Op::loop_axis] = (count + nloop - 1) / nloop;
  const size_t lds = ((((size_t)1 << logC) * lds_sub_stride<P>() + 1) & ~(size_t)1) * sizeof(u64) +
                     ((size_t)1 << (logC + P)) * sizeof(ulonglong2);
  hipLaunchKernelGGL((ntt_loop_kernel<P, LR, INVERSE, Op>), dim3(ext[0] * n_tiles, ext[1], ext[2]), dim3(tile >> LR), lds, c->stream,
                     c->dev, prm, logC, log_tiles, nloop, count);
  HIPCHK(hipGetLastError());
  return true;
}
template <bool INVERSE, class Op>
static bool launch_loop_pass_p(evah_ctx *c, int P, const typename Op::Params &prm, uint32_t jobs) {
  switch (P) {
  case 5: return launch_loop_pass<5, INVERSE, Op>(c, prm, jobs);
  case 6: return launch_loop_pass<6, INVERSE, Op>(c, prm, jobs);
  case 7: return launch_loop_pass<7, INVERSE, Op>(c, prm, jobs);
  case 8: return launch_loop_pass<8, INVERSE, Op>(c, prm, jobs);
  default: return false;
  }
}

// 8 coefficients per thread measured best for the stand-alone passes on MI355X (vs 4: +12 %,
// vs 16: +10 %, profiles/r01_tuning_notes.md); the fused key-switch kernel uses 4.
template <bool STRIDED, bool INVERSE, class Op>
static void launch_pass_p(evah_ctx *c, int P, const typename Op::Params &prm, uint32_t jobs) {
  if constexpr (!STRIDED) {
    if (launch_loop_pass_p<INVERSE, Op>(c, P, prm, jobs)) return;
  }
  // a launch that cannot fill the chip is bound by ONE wave's instruction stream (a thread's 8
  // coefficients are ~600 integer instructions per pass): with 4 coefficients per thread the same
  // tile work is spread over twice the workgroups and the critical path of a workgroup shrinks
  if (c->tun.small_lr == 2 && (uint64_t)jobs * (c->N >> 11) <= c->tun.small_lr_blocks && c->N >= 2048) {
    launch_pass_lr<2, STRIDED, INVERSE, Op>(c, P, prm, jobs);
    return;
  }
  launch_pass_lr<3, STRIDED, INVERSE, Op>(c, P, prm, jobs);
}

struct KsBatch { // one launch worth of key-switches: regular strides, irregular keys
  uint32_t n = 1;
  uint32_t i0 = 0, ni = 0; // output-limb slice
  size_t target_bs = 0, scratch_bs = 0, prod_bs = 0;
  KsKeys keys{};
  PtrTab targets{}; // used when the targets are separate allocations (target == nullptr)
  const MulTab *mul = nullptr; // fused multiply: the target of instance b is d2 = a1 b1 of product b
  bool fold = false;           // fused multiply: P * d0, P * d1 are added to the products here (KS_FOLDMUL; the target is then
                               // read from memory); without mul: P * adds.p[2b + K] is (KS_FOLDADD)
  const PtrTab *adds = nullptr;
  bool mac3 = false;           // keys point at the split layout: ks_inner_kernel<MAC3> (radix-2^30 accumulation)
  bool lazy_out = false;       // the data rows of prod may be any 64-bit representative (consumer: the relinearize + rescale combine)
  uint32_t istep = 1, nout = 0; // output limbs I = i0 + y * istep; nout = rows per polynomial of prod (0: l + 1)
  u64 *r_out = nullptr; // != nullptr: the special row leaves as the first inverse pass of the mod-down (INVSP)
  bool diag = false;    // the diagonal digit comes from scratch too (no NTT-form target: the chain step, ntt_chain.hip.h)
  uint32_t fold_row = ~0u; // KS_FOLDMUL: != ~0u adds (P q_a^-1) d_K, a = fold_row, instead of P d_K
};
// second (contiguous) pass of the digit transforms fused with the key inner product (keyswitch.hip)
void launch_ks_inner(evah_ctx *c, int P, const u64 *target, const u64 *scratch, const KsBatch &key, u64 *prod, uint32_t l);

template <class Op> static void ntt_forward(evah_ctx *c, const typename Op::Params &prm, uint32_t jobs) {
  const int a = (c->logN + 1) / 2, b = c->logN / 2;
  launch_pass_p<true, false, Op>(c, a, prm, jobs);
  launch_pass_p<false, false, Op>(c, b, prm, jobs);
}
template <class Op> static void ntt_inverse(evah_ctx *c, const typename Op::Params &prm, uint32_t jobs) {
  const int a = (c->logN + 1) / 2, b = c->logN / 2;
  launch_pass_p<false, true, Op>(c, b, prm, jobs);
  launch_pass_p<true, true, Op>(c, a, prm, jobs);
}

// ---- latency-bound launches: inverse strided pass + forward strided pass as one launch (ntt_inv_fwd_kernel)
static inline bool fuse_small_launch(evah_ctx *c, uint32_t fwd_jobs) {
  const uint32_t tile = (uint32_t)NTT_THREADS << 3;
  if (!c->tun.fuse_small_blocks || c->N < tile) return false; // partial tiles (N = 1024) keep the two-launch form
  if (c->dev.guard) return true; // a guarded (normally skipped) launch set: the fewest launches, whatever the size
  return (uint64_t)fwd_jobs * (c->N / tile) <= c->tun.fuse_small_blocks;
}
template <int P, class Op, int LR> static void launch_inv_fwd_plr(evah_ctx *c, const typename Op::Params &prm, uint32_t jobs) {
  ProfScope ps(c, OpClass<Op>::fwd_a);
  const uint32_t tile = (uint32_t)NTT_THREADS << LR, n_tiles = c->N / tile;
  const int logC = (int)ilog2(tile) - P;
  const size_t lds = ((((size_t)1 << logC) * lds_sub_stride<P>() + 1) & ~(size_t)1) * sizeof(u64) + 2 * ((size_t)1 << P) * sizeof(ulonglong2);
  dim3 grid = Op::grid(prm, jobs);
  grid.x *= n_tiles;
  hipLaunchKernelGGL((ntt_inv_fwd_kernel<P, LR, Op>), grid, dim3(NTT_THREADS), lds, c->stream, c->dev, prm, (int)ilog2(n_tiles));
  HIPCHK(hipGetLastError());
}
template <int P, class Op> static void launch_inv_fwd_p(evah_ctx *c, const typename Op::Params &prm, uint32_t jobs) {
  if (c->tun.small_lr == 2) launch_inv_fwd_plr<P, Op, 2>(c, prm, jobs);
  else launch_inv_fwd_plr<P, Op, 3>(c, prm, jobs);
}
template <class Op> static void launch_inv_fwd(evah_ctx *c, const typename Op::Params &prm, uint32_t jobs) {
  switch ((c->logN + 1) / 2) {
  case 6: launch_inv_fwd_p<6, Op>(c, prm, jobs); break;
  case 7: launch_inv_fwd_p<7, Op>(c, prm, jobs); break;
  case 8: launch_inv_fwd_p<8, Op>(c, prm, jobs); break;
  case 9: launch_inv_fwd_p<9, Op>(c, prm, jobs); break;
  default: throw std::runtime_error("unsupported poly_modulus_degree for the fused inverse/forward pass");
  }
}
// two strided inverse passes + combine (+ strided forward pass): ntt_inv2_kernel (ntt_chain.hip.h)
template <int P, class Op, bool FWD, int LR> static void launch_inv2_plr(evah_ctx *c, const typename Op::Params &prm, uint32_t jobs) {
  ProfScope ps(c, OpClass<Op>::fwd_a);
  const uint32_t tile = (uint32_t)NTT_THREADS << LR, n_tiles = c->N / tile;
  const int logC = (int)ilog2(tile) - P;
  const size_t lds = ((((size_t)1 << logC) * lds_sub_stride<P>() + 1) & ~(size_t)1) * sizeof(u64) + 2 * ((size_t)1 << P) * sizeof(ulonglong2);
  dim3 grid = Op::grid(prm, jobs);
  grid.x *= n_tiles;
  hipLaunchKernelGGL((ntt_inv2_kernel<P, LR, Op, FWD>), grid, dim3(NTT_THREADS), lds, c->stream, c->dev, prm, (int)ilog2(n_tiles));
  HIPCHK(hipGetLastError());
}
template <class Op, bool FWD> static void launch_inv2(evah_ctx *c, const typename Op::Params &prm, uint32_t jobs) {
  auto go = [&](auto ptag) {
    constexpr int P = decltype(ptag)::value;
    if (c->tun.small_lr == 2) launch_inv2_plr<P, Op, FWD, 2>(c, prm, jobs);
    else launch_inv2_plr<P, Op, FWD, 3>(c, prm, jobs);
  };
  switch ((c->logN + 1) / 2) {
  case 6: go(std::integral_constant<int, 6>{}); break;
  case 7: go(std::integral_constant<int, 7>{}); break;
  case 8: go(std::integral_constant<int, 8>{}); break;
  case 9: go(std::integral_constant<int, 9>{}); break;
  default: throw std::runtime_error("unsupported poly_modulus_degree for the fused inverse/forward pass");
  }
}
// inverse transform of the source limb(s) (InvOp jobs) followed by the forward transforms of Op:
// four launches, or three when the forward launch is too small to fill the chip
// inv_pass1_done: the contiguous inverse pass already left its intermediate in ip.dst (fused into the
// key-switch kernel); only ever set when fuse_small_launch(c, fwd_jobs) holds
template <class InvOp, class Op>
static void inverse_then_forward(evah_ctx *c, const typename InvOp::Params &ip, uint32_t inv_jobs, const typename Op::Params &fp,
                                 uint32_t fwd_jobs, bool inv_pass1_done = false) {
  if (inv_pass1_done && !fuse_small_launch(c, fwd_jobs)) throw std::logic_error("fused inverse pass outside the small-launch form");
  if (fuse_small_launch(c, fwd_jobs)) {
    if (!inv_pass1_done) launch_pass_p<false, true, InvOp>(c, c->logN / 2, ip, inv_jobs); // contiguous inverse pass: lazy intermediate in ip.dst
    launch_inv_fwd<Op>(c, fp, fwd_jobs);
    launch_pass_p<false, false, Op>(c, c->logN / 2, fp, fwd_jobs);
  } else {
    ntt_inverse<InvOp>(c, ip, inv_jobs);
    ntt_forward<Op>(c, fp, fwd_jobs);
  }
}

// ---- defined in keyswitch.hip
// steps 1-2 of SEAL's switch_key_inplace for a batch of n (target, key) pairs (see the definition)
bool switch_key_products(evah_ctx *c, uint32_t l, const u64 *target, size_t target_bs, const KeyDev *const *keys, uint32_t n,
                         u64 *prod_d, const PtrTab *target_tab = nullptr, const MulTab *mul = nullptr, u64 *r_small = nullptr,
                         bool fold = false, const PtrTab *adds = nullptr, bool lazy_out = false);
void switch_key(evah_ctx *c, uint32_t l, const u64 *target, const KeyDev &key, const u64 *add, size_t add_ps, uint32_t add_polys,
                u64 *out, size_t out_ps);
// ---- defined in elementwise.hip: dst[i] = (src[i] mod 2^30) | (src[i] >> 30) << 32 (KeyDev::d_split)
void key_split_launch(evah_ctx *c, const u64 *src, u64 *dst, size_t words);
// ---- defined in rotate.hip: NTT-domain permutation table of a Galois element, cached per device state
const uint32_t *perm_table(evah_ctx *c, uint32_t elt);
// out[p][i][n] = a[p][i][perm[n]] for p < polys over `limbs` limbs (the NTT-domain Galois automorphism)
void galois_perm_launch(evah_ctx *c, const u64 *a, size_t a_ps, uint32_t limbs, uint32_t polys, const uint32_t *perm, u64 *out, size_t o_ps);
// ---- defined in elementwise.hip: FP64 root / slot tables of the CKKS encoder, built on first use
void enc_tables(evah_ctx *c);

} // namespace evah
