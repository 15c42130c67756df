// keyswitch.hip — libeva_hip.so: the key-switch core (SEAL Evaluator::switch_key_inplace, SURVEY.md A.6: digit decomposition, the fused
// second pass + key inner product, mod-down) and the evaluator calls built on it or on the same transforms:
// relinearize, rescale_to_next and their fused / batched forms (/root/reference/eva/seal/seal_executor.h:197-215).
#include "launch.hip.h"

namespace evah {

// K9 inner product: prod[K][I] = sum_J op(I,J) * key[J][K][kappa(I)],
// op(I,J) = target[J] when I == J, else scratch[I][J].  128-bit lazy accumulation, one Barrett
// reduction at the end.  grid = (N/512, l+1).
__global__ void __launch_bounds__(256)
k_ks_mac(DevCtx cx, const u64 *target, const u64 *scratch, const u64 *key, u64 *prod, uint32_t l) {
  if (cx.skipped()) return;
  const uint32_t I = blockIdx.y;
  const uint32_t kap = (I == l) ? cx.k - 1 : I;
  const size_t n = 2 * ((size_t)blockIdx.x * blockDim.x + threadIdx.x);
  const DevPrime pm = cx.primes[kap];
  const size_t N = cx.N, key_digit = (size_t)2 * cx.k * N;
  u128_t a0x = {0, 0}, a0y = {0, 0}, a1x = {0, 0}, a1y = {0, 0};
  for (uint32_t J = 0; J < l; J++) {
    const u64 *op = (I == J) ? target + (size_t)J * N : scratch + ((size_t)I * l + J) * N;
    const ulonglong2 o = ld2(op + n);
    const u64 *kp = key + J * key_digit + (size_t)kap * N + n;
    const ulonglong2 k0 = ld2(kp), k1 = ld2(kp + (size_t)cx.k * N);
    acc128(a0x, o.x, k0.x);
    acc128(a0y, o.y, k0.y);
    acc128(a1x, o.x, k1.x);
    acc128(a1y, o.y, k1.y);
  }
  ulonglong2 r0, r1;
  r0.x = barrett128(a0x, pm);
  r0.y = barrett128(a0y, pm);
  r1.x = barrett128(a1x, pm);
  r1.y = barrett128(a1y, pm);
  st2(prod + (size_t)I * N + n, r0);
  st2(prod + ((size_t)(l + 1) + I) * N + n, r1);
}

template <int P, int LR>
static void launch_ks_inner_plr(evah_ctx *c, const u64 *target, const u64 *scratch, const KsBatch &kb, u64 *prod, uint32_t l) {
  ProfScope ps(c, KC_KSMAC);
  const uint32_t max_tile = (uint32_t)c->tun.ks_threads << LR;
  const uint32_t tile = c->N < max_tile ? c->N : max_tile;
  const int logC = (int)ilog2(tile) - P;
  if (logC < 0) throw std::runtime_error("ks_inner tile smaller than one sub-transform");
  // coefficients tile + per-sub twiddle heaps (16 B per node)
  const size_t lds = ((((size_t)1 << logC) * lds_sub_stride<P>() + 1) & ~(size_t)1) * sizeof(u64) +
                     ((size_t)1 << (logC + P)) * sizeof(ulonglong2) +
                     (kb.mac3 ? (size_t)tile * sizeof(u64) : 0); // MAC3: the unpadded tile the LDS-DMA loads fill
  const uint32_t n_tiles = c->N / tile;
  // the one-wave workgroup (the default) is compiled with its own launch bound: the register
  // allocator is not held to the 256-thread budget
  auto go = [&](auto kernel, const auto &mt, const auto &at) {
    hipLaunchKernelGGL(kernel, dim3(n_tiles * kb.n, kb.ni), dim3(tile >> LR), lds, c->stream, c->dev, target, kb.target_bs, scratch,
                       kb.scratch_bs, kb.keys, prod, kb.prod_bs, l, kb.i0, logC, n_tiles, kb.n, kb.targets, mt, kb.istep,
                       kb.nout ? kb.nout : l + 1, kb.r_out, at, (kb.lazy_out ? 1 : 0) | (kb.diag ? 2 : 0), kb.fold_row);
  };
  if (kb.r_out && ((tile >> LR) > 64 || kb.istep != 1)) throw std::logic_error("fused special-row inverse pass needs the one-wave key-switch kernel");
  if (kb.fold && kb.istep != 1) throw std::logic_error("the folded key-switch forms are not used on limb shards");
  if (kb.fold && !kb.mul && !kb.adds) throw std::logic_error("folded key switch without polynomials to fold");
  const int mode = kb.mul ? (kb.fold ? KS_FOLDMUL : KS_MUL) : (kb.fold ? KS_FOLDADD : KS_PLAIN);
  auto with_mode = [&](auto mode_tag) {
    constexpr int M = decltype(mode_tag)::value;
    const KsMulArg<M> mt = [&] { if constexpr (M == KS_MUL || M == KS_FOLDMUL) return *kb.mul; else return NoMul{}; }();
    const KsAddArg<M> at = [&] { if constexpr (M == KS_FOLDADD) return *kb.adds; else return NoMul{}; }();
    if constexpr (M != KS_MUL) { // radix-2^30 accumulation: the one-wave kernels of the current forms
      if (kb.mac3 && (tile >> LR) <= 64) {
        if (kb.r_out) go(ks_inner_kernel<P, LR, 64, M, true, true>, mt, at);
        else go(ks_inner_kernel<P, LR, 64, M, false, true>, mt, at);
        return;
      }
    }
    if (kb.mac3) throw std::logic_error("split keys handed to a key-switch kernel that accumulates in 128 bits");
    if (kb.r_out) go(ks_inner_kernel<P, LR, 64, M, true>, mt, at);
    else if ((tile >> LR) <= 64) go(ks_inner_kernel<P, LR, 64, M>, mt, at);
    else go(ks_inner_kernel<P, LR, NTT_THREADS, M>, mt, at);
  };
  switch (mode) {
  case KS_PLAIN: with_mode(std::integral_constant<int, KS_PLAIN>{}); break;
  case KS_MUL: with_mode(std::integral_constant<int, KS_MUL>{}); break;
  case KS_FOLDMUL: with_mode(std::integral_constant<int, KS_FOLDMUL>{}); break;
  default: with_mode(std::integral_constant<int, KS_FOLDADD>{}); break;
  }
  HIPCHK(hipGetLastError());
}
template <int LR>
static void launch_ks_inner_lr(evah_ctx *c, int P, const u64 *target, const u64 *scratch, const KsBatch &key, u64 *prod, uint32_t l) {
  switch (P) {
  case 5: launch_ks_inner_plr<5, LR>(c, target, scratch, key, prod, l); break;
  case 6: launch_ks_inner_plr<6, LR>(c, target, scratch, key, prod, l); break;
  case 7: launch_ks_inner_plr<7, LR>(c, target, scratch, key, prod, l); break;
  case 8: launch_ks_inner_plr<8, LR>(c, target, scratch, key, prod, l); break;
  default: throw std::runtime_error("unsupported poly_modulus_degree for the key-switch kernel");
  }
}
void launch_ks_inner(evah_ctx *c, int P, const u64 *target, const u64 *scratch, const KsBatch &key, u64 *prod, uint32_t l) {
  launch_ks_inner_lr<2>(c, P, target, scratch, key, prod, l); // 4 coefficients per thread: 32 accumulator VGPRs
}

// SEAL Evaluator::switch_key_inplace, device version.
//   out[K] = (add && K < add_polys ? add[K] : 0) + keyswitch(target)[K],  K in {0,1}
// steps 1-2 of switch_key for a batch of n (target, key) pairs in one set of launches:
// prod[b][K][I] (I <= l, slot l = special prime) = sum_J op_b(I,J) * key_b[J][K].
// target_b = target + b * target_bs; prod_b = prod_d + b * 2 (l+1) N.
// r_small != nullptr: the caller will mod-down through the latency-bound launch form and offers
// r_small[2 n][N] for the special rows' first inverse pass; returns true when that pass was done
// here (fused into the key-switch kernel) — the special rows of prod are then NOT written.
bool switch_key_products(evah_ctx *c, uint32_t l, const u64 *target, size_t target_bs, const KeyDev *const *keys,
                         uint32_t n, u64 *prod_d, const PtrTab *target_tab, const MulTab *mul, u64 *r_small, bool fold,
                         const PtrTab *adds, bool lazy_out) {
  const size_t N = c->N;
  if (n < 1 || n > (uint32_t)KS_BATCH_MAX) throw std::runtime_error("key-switch batch out of range");
  KsBatch kb;
  kb.n = n;
  kb.target_bs = target_bs;
  kb.scratch_bs = (size_t)(l + 1) * l * N;
  kb.prod_bs = (size_t)2 * (l + 1) * N;
  // radix-2^30 accumulation (ks_inner_kernel<MAC3>): every prime of the top-bit shape, every key with its split copy,
  // at most 15 limbs (the top sum holds 15 digits), the one-wave kernel, not the r03 fused-multiply form
  const uint32_t ks_tile = std::min<uint32_t>(c->N, (uint32_t)c->tun.ks_threads << 2);
  bool mac3 = c->tun.mac3 && c->tun.fuse_mac && c->all_tb && l <= 15 && (ks_tile >> 2) <= 64 && !(mul && !fold);
  for (uint32_t b = 0; b < n; b++) {
    if (keys[b]->n_digits < l) throw std::runtime_error("key switching key has too few digits");
    if (keys[b]->rows != c->k) throw std::logic_error("this context holds a limb shard's key rows: use the evah_shard_* entry points");
    mac3 = mac3 && keys[b]->d_split;
  }
  for (uint32_t b = 0; b < n; b++) kb.keys.key[b] = mac3 ? keys[b]->d_split : keys[b]->d;
  kb.mac3 = mac3;
  if (target_tab) kb.targets = *target_tab; // target == nullptr: separately allocated targets
  kb.mul = mul;
  kb.fold = fold;
  kb.adds = adds;
  kb.lazy_out = lazy_out;
  if ((mul || fold) && !c->tun.fuse_mac) throw std::logic_error("the fused multiply / folded forms need the fused key-switch kernel");
  Scratch t(c, (size_t)n * l * N);        // coefficient-form digits
  Scratch sc(c, n * kb.scratch_bs);       // converted digits, NTT form per output limb
  // folded fused multiply: d2 is stored by the inverse transform that forms it and is the target from memory
  Scratch d2(c, mul && fold ? (size_t)n * l * N : 1);
  if (mul && fold) {
    target = d2.d;
    kb.target_bs = target_bs = (size_t)l * N;
  }
  // 1. digits to coefficient form (job -> (b, J))
  OpKsDigit::Params dp{t.d, sc.d, l, (size_t)l * N, kb.scratch_bs, 0, l + 1};
  // a small key switch is latency-bound: the digits' strided inverse pass and the first pass of the
  // digit conversion then run as one launch
  const bool small = c->tun.fuse_mac && std::max(1, c->tun.ks_groups) == 1 && fuse_small_launch(c, n * (l + 1) * l);
  if (mul) { // the target is the product's d2, formed on load
    OpMulIntt::Params ip{*mul, t.d, (size_t)l * N, l, fold ? d2.d : nullptr};
    if (small) launch_pass_p<false, true, OpMulIntt>(c, c->logN / 2, ip, n * l);
    else ntt_inverse<OpMulIntt>(c, ip, n * l);
  } else {
    OpPlain::Params ip{target, t.d, target_bs, (size_t)l * N, l, 0, 0, {}};
    if (target_tab) ip.src_tab = *target_tab;
    if (small) launch_pass_p<false, true, OpPlain>(c, c->logN / 2, ip, n * l);
    else ntt_inverse<OpPlain>(c, ip, n * l);
  }
  if (small) {
    dp.i0 = kb.i0 = 0;
    dp.ni = kb.ni = l + 1;
    launch_inv_fwd<OpKsDigit>(c, dp, n * (l + 1) * l);
    const uint32_t max_tile = (uint32_t)c->tun.ks_threads << 2;
    if (r_small && c->tun.fuse_special_inv && (std::min<uint32_t>(c->N, max_tile) >> 2) <= 64) kb.r_out = r_small;
    launch_ks_inner(c, c->logN / 2, target, sc.d, kb, prod_d, l);
    return kb.r_out != nullptr;
  }
  if (c->tun.fuse_mac) { // 128-bit accumulation of lazy (<16q) products, folded every 16 digits
    // Output limbs are processed in slices so that a slice's converted digits (ni * l * N words)
    // are still in L2 / Infinity Cache when the fused second pass consumes them.
    const int a = (c->logN + 1) / 2, b = c->logN / 2;
    const uint32_t groups = std::min<uint32_t>(std::max(1, c->tun.ks_groups), l + 1);
    // r5: the digit transforms may be throughput-sized while the mod-down that follows is still a small launch
    // (relinearize_many of Harris' three products: 3456 digit tiles, 768 mod-down tiles): the special rows' first inverse
    // pass then rides the key-switch kernel here as well (one launch of ~6 us less)
    const uint32_t max_tile = (uint32_t)c->tun.ks_threads << 2;
    if (r_small && groups == 1 && c->tun.fuse_special_inv && (std::min<uint32_t>(c->N, max_tile) >> 2) <= 64) kb.r_out = r_small;
    for (uint32_t g = 0; g < groups; g++) {
      const uint32_t i0 = (uint32_t)((uint64_t)(l + 1) * g / groups), i1 = (uint32_t)((uint64_t)(l + 1) * (g + 1) / groups);
      if (i1 == i0) continue;
      dp.i0 = kb.i0 = i0;
      dp.ni = kb.ni = i1 - i0;
      // 2a. base-convert + first (strided) NTT pass of every digit under the slice's output primes
      launch_pass_p<true, false, OpKsDigit>(c, a, dp, n * (i1 - i0) * l);
      // 2b. second (contiguous) pass fused with the inner product with the key
      launch_ks_inner(c, b, target, sc.d, kb, prod_d, l);
    }
    return kb.r_out != nullptr;
  } else {
    // unfused reference path (EVAH_FUSE_MAC=0): full digit NTTs, then a separate MAC kernel
    ntt_forward<OpKsDigit>(c, dp, n * (l + 1) * l);
    for (uint32_t b = 0; b < n; b++) {
      ProfScope ps(c, KC_KSMAC);
      hipLaunchKernelGGL(k_ks_mac, dim3(c->N / 512, l + 1), dim3(256), 0, c->stream, c->dev,
                         target ? target + b * target_bs : target_tab->p[b], sc.d + b * kb.scratch_bs, keys[b]->d,
                         prod_d + b * kb.prod_bs, l);
      HIPCHK(hipGetLastError());
    }
  }
  return false;
}

void switch_key(evah_ctx *c, uint32_t l, const u64 *target, const KeyDev &key, const u64 *add,
                       size_t add_ps, uint32_t add_polys, u64 *out, size_t out_ps) {
  const size_t N = c->N;
  Scratch prod(c, (size_t)2 * (l + 1) * N);    // [K][l+1][N]
  Scratch r(c, 2 * N);
  const KeyDev *kp = &key;
  // fold_pa: P * add[K] goes into the inner products (KS_FOLDADD), the combine pass then adds nothing
  const bool fold = c->tun.fold_pa && c->tun.fuse_mac && add && add_polys >= 1 && add_polys <= 2;
  PtrTab adds{};
  if (fold)
    for (uint32_t K = 0; K < add_polys; K++) adds.p[K] = add + K * add_ps;
  const bool inv1 = switch_key_products(c, l, target, 0, &kp, 1, prod.d, nullptr, nullptr, fuse_small_launch(c, 2 * l) ? r.d : nullptr,
                                        fold, fold ? &adds : nullptr);
  if (fold) add = nullptr;
  // 3. mod-down by the special prime: INTT(special limb) + P/2, then per-limb NTT + combine
  OpPlain::Params sp{prod.d + (size_t)l * N, r.d, (size_t)(l + 1) * N, N, 1, c->k - 1, 1, {}};
    OpModDown::Params mp{r.d, N, prod.d, (size_t)(l + 1) * N, add, add_ps, add_polys, out, out_ps,
                       c->k - 1, l};
  inverse_then_forward<OpPlain, OpModDown>(c, sp, 2, mp, 2 * l, inv1);
}

} // namespace evah

extern "C" {

int evah_relinearize(evah_ctx *c, const evah_ct *a, evah_ct **out) {
  API_BEGIN
  use(c);
  acquire(c, a->buf);
  if (a->size != 3) throw std::invalid_argument("relinearize expects a size-3 ciphertext");
  if (!c->sh->relin.d) throw std::invalid_argument("relinearization key not present");
  evah_ct *o = ct_new(c, 2, a->limbs, a->scale, a->batch);
  try {
    if (a->batch == 1) {
      switch_key(c, a->limbs, a->d + 2 * a->ps, c->sh->relin, a->d, a->ps, 2, o->d, o->ps);
    } else { // all instances in one launch set (chunks of KS_BATCH_MAX)
      const uint32_t l = a->limbs;
      const size_t N = c->N, pps = (size_t)(l + 1) * N;
      for (uint32_t b0 = 0; b0 < a->batch; b0 += KS_BATCH_MAX) {
        const uint32_t n = std::min<uint32_t>(KS_BATCH_MAX, a->batch - b0);
        const u64 *a0 = a->d + (size_t)b0 * 3 * a->ps;
        Scratch prod(c, (size_t)n * 2 * pps);
        std::vector<const KeyDev *> keys(n, &c->sh->relin);
        const bool fold = c->tun.fold_pa && c->tun.fuse_mac;
        PtrTab adds{};
        for (uint32_t b = 0; b < n && fold; b++)
          for (uint32_t K = 0; K < 2; K++) adds.p[2 * b + K] = a0 + (size_t)b * 3 * a->ps + K * a->ps;
        switch_key_products(c, l, a0 + 2 * a->ps, 3 * a->ps, keys.data(), n, prod.d, nullptr, nullptr, nullptr, fold, fold ? &adds : nullptr);
        Scratch r(c, (size_t)n * 2 * N);
        OpPlain::Params sp{prod.d + (size_t)l * N, r.d, pps, N, 1, c->k - 1, 1, {}};
        OpModDown::Params mp{r.d, N, prod.d, pps, fold ? nullptr : a0, a->ps, 2, o->d + (size_t)b0 * 2 * o->ps, o->ps, c->k - 1, l};
        mp.add_bs = 3 * a->ps;
        inverse_then_forward<OpPlain, OpModDown>(c, sp, 2 * n, mp, 2 * n * l);
      }
    }
  } catch (...) {
    evah_ct_free(c, o);
    throw;
  }
  *out = o;
  API_END
}

static void relin_rescale_core(evah_ctx *c, const evah_ct *const *as, uint32_t n, u64 *out_d, const MulTab *mul = nullptr,
                               uint32_t mul_limbs = 0);

int evah_relinearize_rescale(evah_ctx *c, const evah_ct *a, uint32_t divisor_bits, evah_ct **out) {
  API_BEGIN
  use(c);
  acquire(c, a->buf);
  if (a->size != 3) throw std::invalid_argument("relinearize expects a size-3 ciphertext");
  if (!c->sh->relin.d) throw std::invalid_argument("relinearization key not present");
  if (a->limbs < 2) throw std::invalid_argument("end of modulus switching chain reached");
  const uint32_t l = a->limbs, last = l - 1, sp = c->k - 1;
  const size_t N = c->N, pps = (size_t)(l + 1) * N;
  evah_ct *o = ct_new(c, 2, l - 1, a->scale / std::pow(2.0, (double)divisor_bits), a->batch);
  if (a->batch > 1) { // every instance through the batched form, KS_BATCH_MAX at a time
    try {
      std::vector<evah_ct> views(a->batch, *a);
      std::vector<const evah_ct *> ptrs(a->batch);
      for (uint32_t b = 0; b < a->batch; b++) {
        views[b].d = a->d + (size_t)b * 3 * a->ps;
        views[b].batch = 1;
        ptrs[b] = &views[b];
      }
      for (uint32_t b0 = 0; b0 < a->batch; b0 += KS_BATCH_MAX)
        relin_rescale_core(c, ptrs.data() + b0, std::min<uint32_t>(KS_BATCH_MAX, a->batch - b0),
                           o->d + (size_t)b0 * 2 * o->ps);
    } catch (...) {
      evah_ct_free(c, o);
      throw;
    }
    *out = o;
    g_err.clear();
    return 0;
  }
  try {
    Scratch prod(c, 2 * pps);
    const KeyDev *kp = &c->sh->relin;
    const bool fold = c->tun.fold_pa && c->tun.fuse_mac;
    PtrTab adds{};
    adds.p[0] = a->d;
    adds.p[1] = a->d + a->ps;
    switch_key_products(c, l, a->d + 2 * a->ps, 0, &kp, 1, prod.d, nullptr, nullptr, nullptr, fold, fold ? &adds : nullptr, /*lazy_out=*/fold);
    Scratch r(c, 2 * N), t(c, 2 * N);
    // r_K = INTT_P(prod[K][special]) + P/2
    OpPlain::Params spp{prod.d + (size_t)l * N, r.d, pps, N, 1, sp, 1, {}};
    ntt_inverse<OpPlain>(c, spp, 2);
    if (fold) { // prod already carries P a[K]
      OpRRLastFolded::Params lp{nullptr, 0, prod.d + (size_t)last * N, pps, r.d, N, t.d, N, last, sp, {}};
      ntt_inverse<OpRRLastFolded>(c, lp, 2);
      OpRRFolded::Params rp{r.d, N, t.d, N, nullptr, 0, prod.d, pps, o->d, o->ps, sp, last, l - 1, {}};
      ntt_forward<OpRRFolded>(c, rp, 2 * (l - 1));
    } else {
      // t_K = INTT_last(a[K][last] + prod[K][last] P^-1) - u_K,last P^-1 + q_last/2
      OpRRLast::Params lp{a->d + (size_t)last * N, a->ps, prod.d + (size_t)last * N, pps, r.d, N, t.d, N, last, sp, {}};
      ntt_inverse<OpRRLast>(c, lp, 2);
      // out[K][i] = (a[K][i] + prod[K][i] P^-1 - NTT_i(u P^-1 + v)) q_last^-1
      OpRR::Params rp{r.d, N, t.d, N, a->d, a->ps, prod.d, pps, o->d, o->ps, sp, last, l - 1, {}};
      ntt_forward<OpRR>(c, rp, 2 * (l - 1));
    }
  } catch (...) {
    evah_ct_free(c, o);
    throw;
  }
  *out = o;
  API_END
}

// n (<= 64) independent size-3 ciphertexts at the same level, all relinearized with the (shared)
// relinearization key and rescaled: one set of n-times-wider launches; instances are co-scheduled
// per XCD so the key tiles are read from HBM once per XCD, not once per instance.
// core of the batched form: n (<= KS_BATCH_MAX) size-3 ciphertexts at one level -> out_d[n][2][(l-1) N]
// mul != nullptr: instance b is the product a[b] x b[b] of mul (size-2 operands at mul_limbs limbs),
// its polynomials d0, d1, d2 evaluated where they are consumed (as == nullptr then)
static void relin_rescale_core(evah_ctx *c, const evah_ct *const *as, uint32_t n, u64 *out_d, const MulTab *mul, uint32_t mul_limbs) {
  const uint32_t l = mul ? mul_limbs : as[0]->limbs;
  const uint32_t last = l - 1, sp = c->k - 1;
  const size_t N = c->N, pps = (size_t)(l + 1) * N, ops = (size_t)(l - 1) * N;
  PtrTab c2{}, a_last{}, a_polys{};
  for (uint32_t b = 0; b < n && !mul; b++) {
    const evah_ct *a = as[b];
    c2.p[b] = a->d + 2 * a->ps;
    for (uint32_t K = 0; K < 2; K++) {
      a_last.p[2 * b + K] = a->d + K * a->ps + (size_t)last * N;
      a_polys.p[2 * b + K] = a->d + K * a->ps;
    }
  }
  Scratch prod(c, (size_t)n * 2 * pps);
  std::vector<const KeyDev *> keys(n, &c->sh->relin);
  // r04: P * (the polynomials the key-switch result is added to) goes into the inner products themselves, so the
  // combine passes below read prod only (Tunables::fold_pa; the r03 forms stay for A/B runs)
  const bool fold = c->tun.fold_pa && c->tun.fuse_mac; // EVAH_FUSE_MAC=0 (the unfused reference path): the OpRR / OpRRLast forms below
  switch_key_products(c, l, nullptr, 0, keys.data(), n, prod.d, &c2, mul, nullptr, fold, mul ? nullptr : &a_polys, /*lazy_out=*/fold);
  Scratch r(c, (size_t)n * 2 * N), t(c, (size_t)n * 2 * N);
  OpPlain::Params spp{prod.d + (size_t)l * N, r.d, pps, N, 1, sp, 1, {}};
  ntt_inverse<OpPlain>(c, spp, 2 * n);
  if (fold) {
    OpRRLastFolded::Params lp{nullptr, 0, prod.d + (size_t)last * N, pps, r.d, N, t.d, N, last, sp, {}};
    ntt_inverse<OpRRLastFolded>(c, lp, 2 * n);
    OpRRFolded::Params rp{r.d, N, t.d, N, nullptr, 0, prod.d, pps, out_d, ops, sp, last, l - 1, {}};
    ntt_forward<OpRRFolded>(c, rp, 2 * n * (l - 1));
  } else if (mul) {
    OpRRLastMul::Params lp{nullptr, 0, prod.d + (size_t)last * N, pps, r.d, N, t.d, N, last, sp, a_last, *mul};
    ntt_inverse<OpRRLastMul>(c, lp, 2 * n);
    OpRRMul::Params rp{r.d, N, t.d, N, nullptr, 0, prod.d, pps, out_d, ops, sp, last, l - 1, a_polys, *mul};
    ntt_forward<OpRRMul>(c, rp, 2 * n * (l - 1));
  } else {
    OpRRLast::Params lp{nullptr, 0, prod.d + (size_t)last * N, pps, r.d, N, t.d, N, last, sp, a_last};
    ntt_inverse<OpRRLast>(c, lp, 2 * n);
    OpRR::Params rp{r.d, N, t.d, N, nullptr, 0, prod.d, pps, out_d, ops, sp, last, l - 1, a_polys};
    ntt_forward<OpRR>(c, rp, 2 * n * (l - 1));
  }
}

// multiply (size 2 x size 2) -> relinearize -> rescale_to_next for n (<= 64) independent pairs at one
// level, the three SEAL calls of seal_executor.h:164, :200, :213-214 evaluated together: the size-3
// product is never materialised — d2 = a1 b1 is formed in the load of the digit inverse
// transform (and in the key-switch kernel where the NTT-form digit is used as is), d0 and d1 in
// the epilogue that combines them with the key-switch result.  Same ciphertext, bit for bit.
static void mul_relin_rescale(evah_ctx *c, const evah_ct *const *as, const evah_ct *const *bs, uint32_t n, uint32_t divisor_bits,
                              evah_ct **outs) {
  if (n < 1 || n > (uint32_t)KS_BATCH_MAX) throw std::invalid_argument("multiply_relinearize_rescale_many handles 1..64 products per call");
  if (!c->sh->relin.d) throw std::invalid_argument("relinearization key not present");
  const uint32_t l = as[0]->limbs;
  if (l < 2) throw std::invalid_argument("end of modulus switching chain reached");
  const size_t N = c->N, ops = (size_t)(l - 1) * N;
  if (!c->tun.fuse_mac) { // the unfused reference path has no kernel that forms d2 on the fly: the three calls in turn
    std::vector<evah_ct *> prods(n, nullptr);
    if (evah_multiply_many(c, as, bs, n, prods.data())) throw std::runtime_error(g_err);
    const int rc = evah_relinearize_rescale_many(c, prods.data(), n, divisor_bits, outs);
    const std::string msg = g_err;
    for (evah_ct *p : prods) evah_ct_free(c, p);
    if (rc) throw std::runtime_error(msg);
    return;
  }
  MulTab tab{};
  std::vector<double> scales(n);
  for (uint32_t i = 0; i < n; i++) {
    const evah_ct *a = as[i], *b = bs[i];
    if (a->size != 2 || b->size != 2) throw std::invalid_argument("multiply supports size-2 operands only (relinearize first)");
    if (a->batch != 1 || b->batch != 1) throw std::invalid_argument("multiply_relinearize_rescale_many takes single ciphertexts");
    if (a->limbs != l || b->limbs != l) throw std::invalid_argument("encrypted parameter mismatch in batch");
    scales[i] = a->scale * b->scale;
    check_scale(c, scales[i], l);
    acquire(c, a->buf);
    acquire(c, b->buf);
    tab.a[i] = a->d;
    tab.b[i] = b->d;
    tab.a_ps[i] = (uint32_t)(a->ps / N);
    tab.b_ps[i] = (uint32_t)(b->ps / N);
  }
  Buffer *ob = buf_new(c, (size_t)n * 2 * ops);
  try {
    relin_rescale_core(c, nullptr, n, ob->d, &tab, l);
  } catch (...) {
    buf_unref(c, ob);
    throw;
  }
  ob->refs = (int)n;
  for (uint32_t b = 0; b < n; b++) {
    evah_ct *t = new evah_ct;
    t->buf = ob;
    t->d = ob->d + (size_t)b * 2 * ops;
    t->size = 2;
    t->limbs = l - 1;
    t->ps = ops;
    t->scale = scales[b] / std::pow(2.0, (double)divisor_bits);
    outs[b] = t;
  }
}

int evah_relinearize_rescale_many(evah_ctx *c, const evah_ct *const *as, uint32_t n, uint32_t divisor_bits, evah_ct **outs) {
  API_BEGIN
  use(c);
  if (n < 1 || n > (uint32_t)KS_BATCH_MAX) throw std::invalid_argument("relinearize_rescale_many handles 1..64 ciphertexts per call");
  if (!c->sh->relin.d) throw std::invalid_argument("relinearization key not present");
  const uint32_t l = as[0]->limbs;
  if (l < 2) throw std::invalid_argument("end of modulus switching chain reached");
  const size_t N = c->N, ops = (size_t)(l - 1) * N;
  for (uint32_t b = 0; b < n; b++) {
    const evah_ct *a = as[b];
    if (a->size != 3) throw std::invalid_argument("relinearize expects a size-3 ciphertext");
    if (a->batch != 1) throw std::invalid_argument("relinearize_rescale_many takes single ciphertexts (a batched handle goes through evah_relinearize_rescale)");
    if (a->limbs != l) throw std::invalid_argument("encrypted parameter mismatch in batch");
    acquire(c, a->buf);
  }
  Buffer *ob = buf_new(c, (size_t)n * 2 * ops);
  try {
    relin_rescale_core(c, as, n, ob->d);
  } catch (...) {
    buf_unref(c, ob);
    throw;
  }
  ob->refs = (int)n;
  for (uint32_t b = 0; b < n; b++) {
    evah_ct *t = new evah_ct;
    t->buf = ob;
    t->d = ob->d + (size_t)b * 2 * ops;
    t->size = 2;
    t->limbs = l - 1;
    t->ps = ops;
    t->scale = as[b]->scale / std::pow(2.0, (double)divisor_bits);
    outs[b] = t;
  }
  API_END
}

int evah_multiply_relinearize_rescale_many(evah_ctx *c, const evah_ct *const *as, const evah_ct *const *bs, uint32_t n,
                                           uint32_t divisor_bits, evah_ct **outs) {
  API_BEGIN
  use(c);
  if (!c->tun.fuse_mac) { // EVAH_FUSE_MAC=0 (the unfused reference path): the three calls one after the other
    std::vector<evah_ct *> ms(n, nullptr);
    if (evah_multiply_many(c, as, bs, n, ms.data())) throw std::runtime_error(g_err);
    int rc = evah_relinearize_rescale_many(c, ms.data(), n, divisor_bits, outs);
    std::string err = g_err;
    for (evah_ct *m : ms) evah_ct_free(c, m);
    if (rc) throw std::runtime_error(err);
  } else {
    mul_relin_rescale(c, as, bs, n, divisor_bits, outs);
  }
  API_END
}

int evah_multiply_relinearize_rescale(evah_ctx *c, const evah_ct *a, const evah_ct *b, uint32_t divisor_bits, evah_ct **out) {
  API_BEGIN
  use(c);
  if (a->batch != 1 || b->batch != 1 || !c->tun.fuse_mac) { // batched handles: the separate (already batched) calls
    evah_ct *m = nullptr;
    if (evah_multiply(c, a, b, &m)) throw std::runtime_error(g_err);
    int rc = evah_relinearize_rescale(c, m, divisor_bits, out);
    std::string err = g_err;
    evah_ct_free(c, m);
    if (rc) throw std::runtime_error(err);
  } else {
    mul_relin_rescale(c, &a, &b, 1, divisor_bits, out);
  }
  API_END
}

// multiply (size 2 x size 2; a == b: square) -> rescale_to_next -> relinearize for n (<= 64) independent pairs at one level:
// the order lazy relinearization gives a product under the waterline rescalers (seal_executor.h:164 / :162, :213-214, :200).
// The size-3 product is never written — its polynomials are formed where the rescale reads them (OpMulPolyIntt,
// OpModDownMul).  Same ciphertext, bit for bit, as the three calls.
// The chain step in six launches (ntt_chain.hip.h: the rescaled d2 formed in coefficient form where the digit decomposition
// wants it; the rescale of d0 / d1 and the mod-down sharing one forward transform through (P L^-1) d_K folded into the inner
// products).  For ordinary contexts with whole keys and full transform tiles; anything else keeps the forms below.
static bool chain_step_applies(evah_ctx *c, uint32_t lp, uint32_t n) {
  (void)lp;
  (void)n;
  return c->tun.chain_step && c->tun.fuse_mac && c->tun.fold_pa && std::max(1, c->tun.ks_groups) == 1 && c->dev.pstep == 1 &&
         c->dev.p0 == 0 && !c->dev.guard && c->N >= ((uint32_t)NTT_THREADS << 3) && c->sh->relin.rows == c->k;
}
// stored: instance b is a STORED size-3 ciphertext at tab.a[b] (polynomial stride tab.a_ps[b], tab.b[b] == nullptr) instead of
// the product tab.a[b] x tab.b[b] — Rescale -> Relinearize of a sum of products (evah_rescale_relinearize)
static void chain_step(evah_ctx *c, const MulTab &tab, uint32_t n, uint32_t l, u64 *out_d, bool stored = false) {
  const uint32_t lp = l - 1, last = l - 1, sp = c->k - 1;
  const size_t N = c->N, ops = (size_t)lp * N, pps = (size_t)(lp + 1) * N;
  const KeyDev &key = c->sh->relin;
  if (key.n_digits < lp) throw std::runtime_error("key switching key has too few digits");
  Scratch t2(c, (size_t)n * l * N), r01(c, (size_t)n * 2 * N);
  // 1. products + contiguous inverse pass: every limb of d2, limb L of d0 and d1
  OpChainIntt::Params ip{tab, t2.d, r01.d, l};
  launch_pass_p<false, true, OpChainIntt>(c, c->logN / 2, ip, n * (l + 2));
  // 2. t_J = rescaled d2 in coefficient form, digit conversion, strided forward pass
  KsBatch kb;
  kb.n = n;
  kb.scratch_bs = (size_t)(lp + 1) * lp * N;
  kb.prod_bs = (size_t)2 * (lp + 1) * N;
  const uint32_t ks_tile = std::min<uint32_t>(c->N, (uint32_t)c->tun.ks_threads << 2);
  kb.mac3 = c->tun.mac3 && c->all_tb && lp <= 15 && (ks_tile >> 2) <= 64 && key.d_split;
  for (uint32_t b = 0; b < n; b++) kb.keys.key[b] = kb.mac3 ? key.d_split : key.d;
  PtrTab adds{};
  if (stored) { // (P L^-1) c_K from memory: KS_FOLDADD
    for (uint32_t b = 0; b < n; b++)
      for (uint32_t K = 0; K < 2; K++) adds.p[2 * b + K] = tab.a[b] + (size_t)K * tab.a_ps[b] * N;
    kb.adds = &adds;
  } else {
    kb.mul = &tab;
  }
  kb.fold = true;
  kb.fold_row = last;
  kb.diag = true;
  kb.i0 = 0;
  kb.ni = lp + 1;
  Scratch sc(c, n * kb.scratch_bs);
  const uint32_t dig_jobs = n * (lp + 1) * lp;
  if (fuse_small_launch(c, dig_jobs) && (uint64_t)dig_jobs * (c->N >> 11) <= c->tun.chain_fuse_blocks) {
    OpChainDigit::Params dp{t2.d, sc.d, l, kb.scratch_bs};
    launch_inv2<OpChainDigit, true>(c, dp, dig_jobs);
  } else {
    Scratch t(c, (size_t)n * lp * N);
    OpChainT::Params tp{t2.d, t.d, l};
    launch_inv2<OpChainT, false>(c, tp, n * lp);
    OpKsDigit::Params dp{t.d, sc.d, lp, (size_t)lp * N, kb.scratch_bs, 0, lp + 1};
    dp.diag = true;
    launch_pass_p<true, false, OpKsDigit>(c, (c->logN + 1) / 2, dp, dig_jobs);
    // (t is released in stream order: the digit pass above is enqueued before anything that could reuse it)
  }
  // 3. contiguous forward pass + key inner product + (P L^-1) d_K; the special limb's contiguous inverse pass when the
  //    mod-down is a small launch
  Scratch prod(c, (size_t)n * 2 * pps), r(c, (size_t)n * 2 * N);
  const bool small_md = fuse_small_launch(c, 2 * n * lp);
  if (small_md && c->tun.fuse_special_inv && (ks_tile >> 2) <= 64) kb.r_out = r.d;
  launch_ks_inner(c, c->logN / 2, nullptr, sc.d, kb, prod.d, lp);
  // 4. / 5. one forward transform per (K, i) for the rescale of d_K and the mod-down of prod_K
  OpRsMd::Params fp{r01.d, r.d, out_d, ops, last, sp, lp};
  OpModDown::Params mp{r.d, N, prod.d, pps, nullptr, 0, 0, out_d, ops, sp, lp};
  if (small_md) {
    if (!kb.r_out) {
      OpPlain::Params spp{prod.d + (size_t)lp * N, r.d, pps, N, 1, sp, 1, {}};
      launch_pass_p<false, true, OpPlain>(c, c->logN / 2, spp, 2 * n);
    }
    launch_inv2<OpRsMd, true>(c, fp, 2 * n * lp);
  } else {
    OpPlain::Params spp{prod.d + (size_t)lp * N, r.d, pps, N, 1, sp, 1, {}};
    ntt_inverse<OpPlain>(c, spp, 2 * n);
    OpPlain::Params lpp{r01.d, r01.d, N, N, 1, last, 1, {}}; // second (strided) pass of limb L of d0 / d1, in place
    launch_pass_p<true, true, OpPlain>(c, (c->logN + 1) / 2, lpp, 2 * n);
    launch_pass_p<true, false, OpRsMd>(c, (c->logN + 1) / 2, fp, 2 * n * lp);
  }
  launch_pass_p<false, false, OpModDown>(c, c->logN / 2, mp, 2 * n * lp);
}

static void mul_rescale_relin(evah_ctx *c, const evah_ct *const *as, const evah_ct *const *bs, uint32_t n, uint32_t divisor_bits, evah_ct **outs) {
  // batched handles (r6): every operand holds B instances; product i of instance b is entry i * B + b of the launch set, so the
  // outputs of one pair are contiguous — a batched handle again.  pairs * B <= 64 per call.
  const uint32_t pairs = n, B = pairs ? as[0]->batch : 0;
  if (pairs < 1 || B < 1 || (uint64_t)pairs * B > (uint64_t)KS_BATCH_MAX)
    throw std::invalid_argument("multiply_rescale_relinearize_many handles 1..64 products per call (pairs x instances of a batched handle)");
  n = pairs * B;
  if (!c->sh->relin.d) throw std::invalid_argument("relinearization key not present");
  const uint32_t l = as[0]->limbs;
  if (l < 2) throw std::invalid_argument("end of modulus switching chain reached");
  const uint32_t lp = l - 1, last = l - 1, sp = c->k - 1;
  const size_t N = c->N, ops = (size_t)lp * N, pps = (size_t)(lp + 1) * N;
  MulTab tab{};
  std::vector<double> scales(pairs);
  for (uint32_t i = 0; i < pairs; i++) {
    const evah_ct *a = as[i], *b = bs[i];
    if (a->size != 2 || b->size != 2) throw std::invalid_argument("multiply supports size-2 operands only (relinearize first)");
    if (a->batch != B || b->batch != B) throw std::invalid_argument("multiply_rescale_relinearize_many: operands of one call hold the same number of instances");
    if (a->limbs != l || b->limbs != l) throw std::invalid_argument("encrypted parameter mismatch in batch");
    scales[i] = a->scale * b->scale;
    check_scale(c, scales[i], l);
    acquire(c, a->buf);
    acquire(c, b->buf);
    for (uint32_t x = 0; x < B; x++) {
      tab.a[i * B + x] = a->d + (size_t)x * 2 * a->ps;
      tab.b[i * B + x] = b->d + (size_t)x * 2 * b->ps;
      tab.a_ps[i * B + x] = (uint32_t)(a->ps / N);
      tab.b_ps[i * B + x] = (uint32_t)(b->ps / N);
    }
  }
  Buffer *ob = buf_new(c, (size_t)n * 2 * ops);
  try {
    std::vector<const KeyDev *> keys(n, &c->sh->relin);
    if (chain_step_applies(c, lp, n)) {
      chain_step(c, tab, n, l, ob->d);
    } else if (!c->tun.side_stream) {
      // one stream (the default: inside a replayed hipGraph a fork / join costs more than the overlap returns — config 5
      // 1.56 ms either way, r06_tuning_notes.md): the rescale of the three polynomials as one launch set, then the key
      // switch of d2' with P * d0', P * d1' folded into its inner products, as evah_relinearize does on a stored ciphertext
      Scratch dp(c, (size_t)n * 3 * ops), r3(c, (size_t)n * 3 * N);
      OpMulPolyIntt::Params ip{tab, r3.d, 0, 3, last, 1};
      OpModDownMul::Params mpr{r3.d, tab, dp.d, ops, 0, 3, last, lp};
      inverse_then_forward<OpMulPolyIntt, OpModDownMul>(c, ip, 3 * n, mpr, 3 * n * lp);
      Scratch prod(c, (size_t)n * 2 * pps), r(c, (size_t)n * 2 * N);
      const bool fold = c->tun.fold_pa && c->tun.fuse_mac;
      PtrTab adds{};
      for (uint32_t b = 0; b < n && fold; b++)
        for (uint32_t K = 0; K < 2; K++) adds.p[2 * b + K] = dp.d + (size_t)(3 * b + K) * ops;
      const bool inv1 = switch_key_products(c, lp, dp.d + 2 * ops, 3 * ops, keys.data(), n, prod.d, nullptr, nullptr,
                                            fuse_small_launch(c, 2 * n * lp) ? r.d : nullptr, fold, fold ? &adds : nullptr);
      OpPlain::Params spp{prod.d + (size_t)lp * N, r.d, pps, N, 1, sp, 1, {}};
      OpModDown::Params mp{r.d, N, prod.d, pps, fold ? nullptr : dp.d, ops, 2, ob->d, ops, sp, lp};
      mp.add_bs = 3 * ops;
      inverse_then_forward<OpPlain, OpModDown>(c, spp, 2 * n, mp, 2 * n * lp, inv1);
    } else {
    // two streams (EVAH_SIDE_STREAM=1; pays on eager walks: config 5 1.77 -> 1.70 ms): the rescale of d2 followed by its
    // key switch on the queue's stream, the rescale of d0 and d1 beside them on the queue's side stream; the halves meet in
    // the mod-down's combine, which adds d0' and d1'
    Scratch dp2(c, (size_t)n * ops), dp01(c, (size_t)n * 2 * ops), r3(c, (size_t)n * 3 * N);
    u64 *r01 = r3.d, *r2 = r3.d + (size_t)2 * n * N;
    SideStream side(c);
    // d2' = rescale(d2): what the key switch starts from
    OpMulPolyIntt::Params ip2{tab, r2, 2, 1, last, 1};
    OpModDownMul::Params mp2{r2, tab, dp2.d, ops, 2, 1, last, lp};
    inverse_then_forward<OpMulPolyIntt, OpModDownMul>(c, ip2, n, mp2, n * lp);
    // d0', d1' beside it
    side.fork();
    OpMulPolyIntt::Params ip01{tab, r01, 0, 2, last, 1};
    OpModDownMul::Params mp01{r01, tab, dp01.d, ops, 0, 2, last, lp};
    inverse_then_forward<OpMulPolyIntt, OpModDownMul>(c, ip01, 2 * n, mp01, 2 * n * lp);
    side.back();
    // key switch of d2' (the products only; d0', d1' join in the combine)
    Scratch prod(c, (size_t)n * 2 * pps), r(c, (size_t)n * 2 * N);
    const bool inv1 = switch_key_products(c, lp, dp2.d, ops, keys.data(), n, prod.d, nullptr, nullptr, fuse_small_launch(c, 2 * n * lp) ? r.d : nullptr);
    side.join();
    OpPlain::Params spp{prod.d + (size_t)lp * N, r.d, pps, N, 1, sp, 1, {}};
    OpModDown::Params mp{r.d, N, prod.d, pps, dp01.d, ops, 2, ob->d, ops, sp, lp};
    mp.add_bs = 2 * ops;
    inverse_then_forward<OpPlain, OpModDown>(c, spp, 2 * n, mp, 2 * n * lp, inv1);
    }
  } catch (...) {
    buf_unref(c, ob);
    throw;
  }
  ob->refs = (int)pairs;
  for (uint32_t b = 0; b < pairs; b++) {
    evah_ct *t = new evah_ct;
    t->buf = ob;
    t->d = ob->d + (size_t)b * B * 2 * ops;
    t->size = 2;
    t->limbs = lp;
    t->ps = ops;
    t->scale = scales[b] / std::pow(2.0, (double)divisor_bits);
    t->batch = B;
    outs[b] = t;
  }
}

int evah_multiply_rescale_relinearize_many(evah_ctx *c, const evah_ct *const *as, const evah_ct *const *bs, uint32_t n, uint32_t divisor_bits,
                                           evah_ct **outs) {
  API_BEGIN
  use(c);
  mul_rescale_relin(c, as, bs, n, divisor_bits, outs);
  API_END
}
int evah_multiply_rescale_relinearize(evah_ctx *c, const evah_ct *a, const evah_ct *b, uint32_t divisor_bits, evah_ct **out) {
  API_BEGIN
  use(c);
  mul_rescale_relin(c, &a, &b, 1, divisor_bits, out);
  API_END
}

// rescale_to_next of a size-3 ciphertext followed by relinearize (seal_executor.h:213-214, :200) — what lazy relinearization
// leaves after a SUM of products (Sobel's Ix^2 + Iy^2, Harris' response) — as the chain step on stored polynomials: n handles
// of B instances each, n * B <= 64.  Same ciphertext as the two calls.
static void rescale_relin(evah_ctx *c, const evah_ct *const *as, uint32_t pairs, uint32_t divisor_bits, evah_ct **outs) {
  const uint32_t B = pairs ? as[0]->batch : 0;
  if (pairs < 1 || B < 1 || (uint64_t)pairs * B > (uint64_t)KS_BATCH_MAX)
    throw std::invalid_argument("rescale_relinearize_many handles 1..64 ciphertexts per call (handles x instances of a batched handle)");
  if (!c->sh->relin.d) throw std::invalid_argument("relinearization key not present");
  const uint32_t l = as[0]->limbs, n = pairs * B;
  if (l < 2) throw std::invalid_argument("end of modulus switching chain reached");
  const uint32_t lp = l - 1;
  const size_t N = c->N, ops = (size_t)lp * N;
  for (uint32_t i = 0; i < pairs; i++) {
    if (as[i]->size != 3) throw std::invalid_argument("relinearize expects a size-3 ciphertext");
    if (as[i]->batch != B) throw std::invalid_argument("rescale_relinearize_many: the handles of one call hold the same number of instances");
    if (as[i]->limbs != l) throw std::invalid_argument("encrypted parameter mismatch in batch");
  }
  if (!chain_step_applies(c, lp, n)) { // the two calls
    for (uint32_t i = 0; i < pairs; i++) {
      evah_ct *r = nullptr;
      if (evah_rescale(c, as[i], divisor_bits, &r)) throw std::runtime_error(g_err);
      const int rc = evah_relinearize(c, r, &outs[i]);
      const std::string msg = g_err;
      evah_ct_free(c, r);
      if (rc) {
        for (uint32_t j = 0; j < i; j++) evah_ct_free(c, outs[j]);
        throw std::runtime_error(msg);
      }
    }
    return;
  }
  MulTab tab{};
  for (uint32_t i = 0; i < pairs; i++) {
    acquire(c, as[i]->buf);
    for (uint32_t x = 0; x < B; x++) {
      tab.a[i * B + x] = as[i]->d + (size_t)x * 3 * as[i]->ps;
      tab.a_ps[i * B + x] = (uint32_t)(as[i]->ps / N);
    }
  }
  Buffer *ob = buf_new(c, (size_t)n * 2 * ops);
  try {
    chain_step(c, tab, n, l, ob->d, /*stored=*/true);
  } catch (...) {
    buf_unref(c, ob);
    throw;
  }
  ob->refs = (int)pairs;
  for (uint32_t b = 0; b < pairs; b++) {
    evah_ct *t = new evah_ct;
    t->buf = ob;
    t->d = ob->d + (size_t)b * B * 2 * ops;
    t->size = 2;
    t->limbs = lp;
    t->ps = ops;
    t->scale = as[b]->scale / std::pow(2.0, (double)divisor_bits);
    t->batch = B;
    outs[b] = t;
  }
}
int evah_rescale_relinearize_many(evah_ctx *c, const evah_ct *const *as, uint32_t n, uint32_t divisor_bits, evah_ct **outs) {
  API_BEGIN
  use(c);
  rescale_relin(c, as, n, divisor_bits, outs);
  API_END
}
int evah_rescale_relinearize(evah_ctx *c, const evah_ct *a, uint32_t divisor_bits, evah_ct **out) {
  API_BEGIN
  use(c);
  rescale_relin(c, &a, 1, divisor_bits, out);
  API_END
}

// n independent ciphertexts of one size and level rescaled in one launch set (n * size <= 128)
int evah_rescale_many(evah_ctx *c, const evah_ct *const *cts, uint32_t n, uint32_t divisor_bits, evah_ct **outs) {
  API_BEGIN
  use(c);
  const uint32_t size = cts[0]->size, l = cts[0]->limbs;
  if (n < 1 || (size_t)n * size > 2 * KS_BATCH_MAX) throw std::invalid_argument("rescale_many: too many polynomials for one call");
  if (l < 2) throw std::invalid_argument("end of modulus switching chain reached");
  const size_t N = c->N, ops = (size_t)(l - 1) * N;
  const uint32_t polys = n * size;
  PtrTab last{}, all{};
  for (uint32_t b = 0; b < n; b++) {
    const evah_ct *a = cts[b];
    if (a->size != size || a->limbs != l) throw std::invalid_argument("encrypted parameter mismatch in batch");
    if (a->batch != 1) throw std::invalid_argument("rescale_many takes single ciphertexts");
    acquire(c, a->buf);
    for (uint32_t p = 0; p < size; p++) {
      all.p[b * size + p] = a->d + (size_t)p * a->ps;
      last.p[b * size + p] = a->d + (size_t)p * a->ps + (size_t)(l - 1) * N;
    }
  }
  Buffer *ob = buf_new(c, (size_t)polys * ops);
  try {
    Scratch r(c, (size_t)polys * N);
    OpPlain::Params ip{nullptr, r.d, 0, N, 1, l - 1, 1, last};
        OpModDown::Params mp{r.d, N, nullptr, 0, nullptr, 0, 0, ob->d, ops, l - 1, l - 1};
    mp.c_tab = all;
    inverse_then_forward<OpPlain, OpModDown>(c, ip, polys, mp, polys * (l - 1));
  } catch (...) {
    buf_unref(c, ob);
    throw;
  }
  ob->refs = (int)n;
  for (uint32_t b = 0; b < n; b++) {
    evah_ct *t = new evah_ct;
    t->buf = ob;
    t->d = ob->d + (size_t)b * size * ops;
    t->size = size;
    t->limbs = l - 1;
    t->ps = ops;
    t->scale = cts[b]->scale / std::pow(2.0, (double)divisor_bits);
    outs[b] = t;
  }
  API_END
}

// n (<= 64) independent size-3 ciphertexts of one level relinearized in one launch set
int evah_relinearize_many(evah_ctx *c, const evah_ct *const *cts, uint32_t n, evah_ct **outs) {
  API_BEGIN
  use(c);
  if (n < 1 || n > (uint32_t)KS_BATCH_MAX) throw std::invalid_argument("relinearize_many handles 1..64 ciphertexts per call");
  if (!c->sh->relin.d) throw std::invalid_argument("relinearization key not present");
  const uint32_t l = cts[0]->limbs;
  const size_t N = c->N, pps = (size_t)(l + 1) * N, ops = (size_t)l * N;
  PtrTab c2{}, c01{};
  for (uint32_t b = 0; b < n; b++) {
    const evah_ct *a = cts[b];
    if (a->size != 3) throw std::invalid_argument("relinearize expects a size-3 ciphertext");
    if (a->batch != 1) throw std::invalid_argument("relinearize_many takes single ciphertexts");
    if (a->limbs != l) throw std::invalid_argument("encrypted parameter mismatch in batch");
    acquire(c, a->buf);
    c2.p[b] = a->d + 2 * a->ps;
    c01.p[2 * b] = a->d;
    c01.p[2 * b + 1] = a->d + a->ps;
  }
  Buffer *ob = buf_new(c, (size_t)n * 2 * ops);
  try {
    Scratch prod(c, (size_t)n * 2 * pps);
    std::vector<const KeyDev *> keys(n, &c->sh->relin);
    const bool fold = c->tun.fold_pa && c->tun.fuse_mac;
    Scratch r(c, (size_t)n * 2 * N);
    const bool inv1 = switch_key_products(c, l, nullptr, 0, keys.data(), n, prod.d, &c2, nullptr, fuse_small_launch(c, 2 * n * l) ? r.d : nullptr,
                                          fold, fold ? &c01 : nullptr);
    OpPlain::Params sp{prod.d + (size_t)l * N, r.d, pps, N, 1, c->k - 1, 1, {}};
    OpModDown::Params mp{r.d, N, prod.d, pps, nullptr, 0, 0, ob->d, ops, c->k - 1, l};
    mp.use_add_tab = !fold; // folded: P c_K is in prod already
    mp.add_tab = c01;
    inverse_then_forward<OpPlain, OpModDown>(c, sp, 2 * n, mp, 2 * n * l, inv1);
  } catch (...) {
    buf_unref(c, ob);
    throw;
  }
  ob->refs = (int)n;
  for (uint32_t b = 0; b < n; b++) {
    evah_ct *t = new evah_ct;
    t->buf = ob;
    t->d = ob->d + (size_t)b * 2 * ops;
    t->size = 2;
    t->limbs = l;
    t->ps = ops;
    t->scale = cts[b]->scale;
    outs[b] = t;
  }
  API_END
}

int evah_rescale(evah_ctx *c, const evah_ct *a, uint32_t divisor_bits, evah_ct **out) {
  API_BEGIN
  use(c);
  acquire(c, a->buf);
  if (a->limbs < 2) throw std::invalid_argument("end of modulus switching chain reached");
  const uint32_t l = a->limbs;
  const size_t N = c->N;
  const uint32_t polys = a->size * a->batch; // a batched handle is batch * size polynomials at stride ps
  evah_ct *o = ct_new(c, a->size, l - 1, a->scale / std::pow(2.0, (double)divisor_bits), a->batch);
  Scratch r(c, (size_t)polys * N);
  OpPlain::Params ip{a->d + (size_t)(l - 1) * N, r.d, a->ps, N, 1, l - 1, 1, {}};
    OpModDown::Params mp{r.d, N, a->d, a->ps, nullptr, 0, 0, o->d, o->ps, l - 1, l - 1};
  inverse_then_forward<OpPlain, OpModDown>(c, ip, polys, mp, polys * (l - 1));
  *out = o;
  API_END
}

} // extern "C"
