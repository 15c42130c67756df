// rotate.hip — libeva_hip.so: rotate_vector (/root/reference/eva/seal/seal_executor.h:177-189): the NTT-domain Galois permutation, single
// rotations, and rotation sets — sibling rotations as one launch set, hoisted when throughput-sized (one digit
// decomposition per source) with the exact fallback.  The machinery of the sets is rotation_sets.hip.h
// (shared with windows.hip, where the sets feed weighted sums).
#include "rotation_sets.hip.h"

namespace evah {


// K8: NTT-domain Galois automorphism out[p][i][n] = in[p][i][perm[n]]
__global__ void __launch_bounds__(256)
k_galois_perm(DevCtx cx, const u64 *a, size_t a_ps, const uint32_t *perm, u64 *out, size_t o_ps) {
  EW_SETUP
  const uint32_t n = 2 * (blockIdx.x * blockDim.x + threadIdx.x);
  const uint2 pi = *reinterpret_cast<const uint2 *>(perm + n);
  const u64 *src = a + p * a_ps + (size_t)i * cx.N;
  ulonglong2 r;
  r.x = src[pi.x];
  r.y = src[pi.y];
  st2(out + p * o_ps + off, r);
}

void galois_perm_launch(evah_ctx *c, const u64 *a, size_t a_ps, uint32_t limbs, uint32_t polys, const uint32_t *perm, u64 *out, size_t o_ps) {
  EW_LAUNCH(k_galois_perm, ew_grid(c, limbs, polys), dim3(256), 0, c->stream, c->dev, a, a_ps, perm, out, o_ps);
  HIPCHK(hipGetLastError());
}

bool hoist_wanted(const evah_ctx *c, uint32_t l, uint32_t n, uint32_t B) {
  // hoisting pays when the digit transforms it saves are throughput, not latency (the exact fallback
  // costs a set of empty launches); limb-sharded contexts go through the shard phases instead
  return c->tun.hoist && n >= 2 && c->dev.pstep == 1 && c->N >= 2048 &&
         (uint64_t)n * B * l * (l + 1) * (c->N >> 11) >= c->tun.hoist_min_tiles;
}


// NTT-domain permutation table of a Galois element (SEAL GaloisTool::generate_table_ntt), cached
const uint32_t *perm_table(evah_ctx *c, uint32_t elt) {
  auto pit = c->sh->perms.find(elt);
  if (pit != c->sh->perms.end()) return pit->second;
  if (c->capturing) throw std::logic_error("first use of a Galois element cannot be captured into a graph");
  const size_t N = c->N;
  std::vector<uint32_t> tab(N);
  for (uint32_t i = 0; i < N; i++) {
    uint32_t reversed = bitrev((uint32_t)N + i, c->logN + 1);
    u64 raw = (((u64)elt * reversed) >> 1) & (u64)(N - 1);
    tab[i] = bitrev((uint32_t)raw, c->logN);
  }
  uint32_t *d = nullptr;
  HIPCHK(hipMalloc(&d, sizeof(uint32_t) * N));
  h2d_now(c, d, tab.data(), sizeof(uint32_t) * N);
  c->sh->perms.emplace(elt, d);
  return d;
}
} // namespace evah

extern "C" {

// The whole set.  hoisted: the digits of every distinct source are transformed once and each pair's
// key inner product is formed from them (k_hoist_mac / k_hoist_fix), then the unhoisted launches
// follow under the device-side guard.  srcs[i] = c0 of distinct source i (poly stride src_ps[i]).
static void rotation_set(evah_ctx *c, uint32_t l, const std::vector<RotPair> &pairs, const std::vector<RotChunk> &chunks,
                         const std::vector<const u64 *> &srcs, const std::vector<size_t> &src_ps, bool hoisted) {
  const size_t N = c->N, pps = (size_t)l * N, prod_bs = (size_t)2 * (l + 1) * N;
  if (!hoisted) {
    for (const RotChunk &ch : chunks) rot_chunk_plain(c, l, pairs.data() + ch.first, ch.count, ch.out);
    return;
  }
  const uint32_t n_src = (uint32_t)srcs.size();
  if (n_src > (uint32_t)KS_BATCH_MAX) throw std::logic_error("hoisted rotation set with too many sources");
  // one barrier word per chunk (k_rot_fallback), then the zero-coefficient count and the recorded positions: the words
  // that must start at zero are adjacent (cleared by the first pass of hoist_digits)
  ZeroFlag flag(c, chunks.size());
  const size_t dg_bs = (size_t)(l + 1) * l * N;
  {
    Scratch t(c, (size_t)n_src * l * N), dg(c, n_src * dg_bs);
    hoist_digits(c, l, srcs, src_ps, flag, t.d, dg.d);
    for (const RotChunk &ch : chunks) {
      const RotPair *pr = pairs.data() + ch.first;
      const uint32_t np = ch.count;
      HoistMacTab mt{};
      HoistFixTab ft{};
      const HoistTiles tiles = hoist_tables(pr, np, N, mt, ft);
      // the rotated c0: folded into the inner product (fold_pa), or a permuted copy the mod-down adds (even polys only)
      const bool fold = c->tun.fold_pa;
      Scratch perm(c, fold ? 1 : (size_t)np * 2 * pps);
      if (!fold) rot_perm_launch(c, l, pr, np, perm.d, 1);
      Scratch prod(c, np * prod_bs), r(c, (size_t)np * 2 * N);
      hoist_mac_launch(c, mt, tiles, dg.d, dg_bs, prod.d, prod_bs, l, fold);
      {
        ProfScope ps(c, KC_KSMAC);
        // the terms of recorded zero coefficients (returns at once when there are none)
        hipLaunchKernelGGL(k_hoist_fix, dim3(c->N / 256, l + 1, np), dim3(256), 0, c->stream, c->dev, flag.d, ft, prod.d, prod_bs, l);
        HIPCHK(hipGetLastError());
      }
      PermTab gt{};
      for (uint32_t q = 0; q < np; q++) gt.p[q] = pr[q].perm;
      rot_mod_down(c, l, np, prod.d, fold ? nullptr : perm.d, ch.out, r.d, false, &gt);
    }
  }
  // exact fallback: the same outputs through the unhoisted launches, each a no-op unless there
  // were more zero digit coefficients than k_hoist_fix handles
  GuardScope gs(c, reinterpret_cast<const uint32_t *>(flag.d));
  for (size_t ci = 0; ci < chunks.size(); ci++) {
    const RotChunk &ch = chunks[ci];
    if (c->tun.fb_persist) rot_fallback_launch(c, l, pairs.data() + ch.first, ch.count, ch.out, nullptr, 0, 0, 0, flag.bar(ci));
    else rot_chunk_plain(c, l, pairs.data() + ch.first, ch.count, ch.out);
  }
  if (!c->capturing && c->tun.hoist_debug) { // diagnostics: how many zero coefficients did this set see?
    uint32_t f = 0;
    HIPCHK(hipMemcpyAsync(&f, flag.d, sizeof(f), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    std::fprintf(stderr, "rotation set: hoisted %zu rotations of %u sources, l = %u, zero coefficients = %u\n", pairs.size(), n_src, l, f);
  }
}

// Several rotations of ONE ciphertext (the convolution pattern: image << i*w+j for a 3x3
// window) issued as one set of wide launches: same results as n evah_rotate calls, 1/n of the
// kernel launches, each launch n times wider.  Large launch sets are hoisted (see k_hoist_mac).
int evah_rotate_many(evah_ctx *c, const evah_ct *a, const int32_t *steps, uint32_t n, evah_ct **outs) {
  API_BEGIN
  use(c);
  acquire(c, a->buf);
  if (a->size != 2) throw std::invalid_argument("rotate expects a size-2 ciphertext (relinearize first)");
  if (n < 1 || n > (uint32_t)KS_BATCH_MAX) throw std::invalid_argument("rotate_many handles 1..64 rotations per call");
  const uint32_t l = a->limbs, B = a->batch;
  const size_t pps = (size_t)l * c->N;
  bool hoisted = hoist_wanted(c, l, n, B);
  // the hoisting tables of every step first: if one does not fit the device, the whole set runs unhoisted
  std::vector<RotPair> step_pair(n);
  for (uint32_t j = 0; j < n; j++) step_pair[j] = rot_pair(c, a->d, a->ps, 0, steps[j], l, "rotate_many");
  for (uint32_t j = 0; j < n && hoisted; j++) hoisted = hoist_prepare(c, step_pair[j], l);
  // (rotation j, instance b) pairs go out KS_BATCH_MAX at a time: m rotations x B instances per
  // launch set; pair index r = j * B + b, so rotation j's B outputs are one batched handle.
  std::vector<RotPair> pairs;
  pairs.reserve((size_t)n * B);
  for (uint32_t j = 0; j < n; j++) {
    RotPair p = step_pair[j];
    for (uint32_t b = 0; b < B; b++) {
      p.src = a->d + (size_t)b * 2 * a->ps;
      p.src_idx = b;
      pairs.push_back(p);
    }
  }
  std::vector<const u64 *> srcs(B);
  std::vector<size_t> src_ps(B, a->ps);
  for (uint32_t b = 0; b < B; b++) srcs[b] = a->d + (size_t)b * 2 * a->ps;
  const uint32_t m_max = std::max<uint32_t>(1, KS_BATCH_MAX / B);
  std::vector<evah_ct *> made;
  std::vector<Buffer *> chunk_buf;
  std::vector<RotChunk> chunks;
  try {
    for (uint32_t j0 = 0; j0 < n; j0 += m_max) {
      const uint32_t m = std::min(m_max, n - j0), np = m * B;
      Buffer *ob = buf_new(c, (size_t)np * 2 * pps); // one buffer for the chunk; the m handles are views into it
      ob->refs = 0;
      chunk_buf.push_back(ob);
      chunks.push_back({j0 * B, np, ob->d});
      for (uint32_t j = 0; j < m; j++) {
        evah_ct *t = new evah_ct;
        t->buf = ob;
        ob->refs++;
        t->d = ob->d + (size_t)j * B * 2 * pps;
        t->size = 2;
        t->limbs = l;
        t->ps = pps;
        t->scale = a->scale;
        t->batch = B;
        made.push_back(t);
      }
    }
    rotation_set(c, l, pairs, chunks, srcs, src_ps, hoisted);
  } catch (...) {
    for (evah_ct *t : made) evah_ct_free(c, t);
    if (made.empty())
      for (Buffer *b : chunk_buf) { b->refs = 1; buf_unref(c, b); }
    throw;
  }
  for (uint32_t r = 0; r < n; r++) outs[r] = made[r];
  API_END
}

// n (<= 64) independent (ciphertext, step) rotations at one level as one launch set: the sibling
// rotations of SEVERAL ciphertexts (independent convolutions of one program level).  Sources that
// appear more than once share their digit decomposition when the set is large enough to hoist.
int evah_rotate_pairs(evah_ctx *c, const evah_ct *const *cts, const int32_t *steps, uint32_t n, evah_ct **outs) {
  API_BEGIN
  use(c);
  if (n < 1 || n > (uint32_t)KS_BATCH_MAX) throw std::invalid_argument("rotate_pairs handles 1..64 rotations per call");
  const uint32_t l = cts[0]->limbs;
  const size_t pps = (size_t)l * c->N;
  std::vector<RotPair> pairs;
  std::vector<const u64 *> srcs;
  std::vector<size_t> src_ps;
  for (uint32_t r = 0; r < n; r++) {
    const evah_ct *a = cts[r];
    if (a->size != 2) throw std::invalid_argument("rotate expects a size-2 ciphertext (relinearize first)");
    if (a->batch != 1) throw std::invalid_argument("rotate_pairs takes single ciphertexts");
    if (a->limbs != l) throw std::invalid_argument("encrypted parameter mismatch in batch");
    acquire(c, a->buf);
    uint32_t si = 0;
    while (si < srcs.size() && !(srcs[si] == a->d && src_ps[si] == a->ps)) si++;
    if (si == srcs.size()) {
      srcs.push_back(a->d);
      src_ps.push_back(a->ps);
    }
    pairs.push_back(rot_pair(c, a->d, a->ps, si, steps[r], l, "rotate_pairs"));
  }
  // worth hoisting when sources repeat and the digit transforms of the set are throughput-sized
  bool hoisted = srcs.size() < n && hoist_wanted(c, l, n, 1);
  for (size_t i = 0; i < pairs.size() && hoisted; i++) hoisted = hoist_prepare(c, pairs[i], l); // no room for a table: unhoisted
  Buffer *ob = buf_new(c, (size_t)n * 2 * pps);
  try {
    rotation_set(c, l, pairs, {RotChunk{0, n, ob->d}}, srcs, src_ps, hoisted);
  } catch (...) {
    buf_unref(c, ob);
    throw;
  }
  ob->refs = (int)n;
  for (uint32_t r = 0; r < n; r++) {
    evah_ct *t = new evah_ct;
    t->buf = ob;
    t->d = ob->d + (size_t)r * 2 * pps;
    t->size = 2;
    t->limbs = l;
    t->ps = pps;
    t->scale = cts[r]->scale;
    outs[r] = t;
  }
  API_END
}

int evah_rotate(evah_ctx *c, const evah_ct *a, int32_t steps, evah_ct **out) {
  API_BEGIN
  use(c);
  acquire(c, a->buf);
  if (a->size != 2) throw std::invalid_argument("rotate expects a size-2 ciphertext (relinearize first)");
  const size_t N = c->N;
  if (steps == 0) { // SEAL rotate_internal: no-op — the result is the operand; handles are immutable, so share the buffer
    evah_ct *o = new evah_ct(*a);
    o->buf->refs++;
    *out = o;
  } else if (a->batch > 1) {
    if (evah_rotate_many(c, a, &steps, 1, out)) throw std::runtime_error(g_err);
  } else {
    uint32_t elt = 0;
    if (evah_galois_elt_from_step(c, steps, &elt)) throw std::invalid_argument(g_err);
    auto kit = c->sh->galois.find(elt);
    if (kit == c->sh->galois.end()) throw std::invalid_argument("Galois key not present");
    const uint32_t *ptab = perm_table(c, elt);
    Scratch perm(c, (size_t)2 * a->limbs * N); // [c0 permuted][c1 permuted = key-switch target]
    const size_t pps = (size_t)a->limbs * N;
    EW_LAUNCH(k_galois_perm, ew_grid(c, a->limbs, 2), dim3(256), 0, c->stream, c->dev, a->d, a->ps,
                       ptab, perm.d, pps);
    HIPCHK(hipGetLastError());
    evah_ct *o = ct_new(c, 2, a->limbs, a->scale);
    try {
      switch_key(c, a->limbs, perm.d + pps, kit->second, perm.d, pps, 1, o->d, o->ps);
    } catch (...) {
      evah_ct_free(c, o);
      throw;
    }
    *out = o;
  }
  API_END
}

} // extern "C"
