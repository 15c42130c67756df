// shard.hip — limb-sharded execution (BASELINE config 5): the RNS limbs
// of every ciphertext and plaintext are dealt over G shards, limb i to shard i mod G, the special
// prime's limb of the key-switch products to shard l mod G.  A translation unit of libeva_hip.so (launch plumbing:
// launch.hip.h, key-switch core: keyswitch.hip).
//
// A shard is an evah_ctx with a limb -> prime map (evah_ctx_set_shard): its values hold the local
// limbs only, and every per-limb entry point of the evaluator (add, sub, negate, multiply, square,
// multiply_plain, add_plain, sub_plain, weighted_sum, uploads, downloads) works on them unchanged.
// The three operations that mix limbs run in phases with one exchange step between phases — the
// caller (eva_amd/shard.py) does the exchange: device / peer copies when the shards live in one
// process, RCCL collectives (all-gather, broadcast) when every shard is a process on its own GPU:
//   key switch (relinearize, rotate; SEAL switch_key_inplace)
//     1. evah_shard_ks_digits    INTT of the local digits t_J, J = s (mod G), into the gather buffer
//        -- all-gather of the l coefficient-form digits (l N 8 bytes in total) --
//     2. evah_shard_ks_products  for the local output limbs I: sum_J NTT_I(t_J mod q_I) (*) key[J][K][I];
//                                the owner of the special limb also produces r_K = INTT_P(.) + P/2
//        -- broadcast of r (2 N 8 bytes) from the owner of the special limb --
//     3. evah_shard_ks_finish    mod-down and combine on the local data limbs
//   rescale_to_next
//     1. evah_shard_rescale_last    owner of the last limb: r_p = INTT(c[p][last]) + q_last/2
//        -- broadcast of r (size N 8 bytes) --
//     2. evah_shard_rescale_finish  divide-and-round on the local limbs of the next level
// Every stored word is the canonical residue the unsharded path stores for that limb.

#include "launch.hip.h"

struct evah_buf {
  Buffer *buf;
  size_t words;
};

namespace {

uint32_t shard_of(evah_ctx *c) { return c->dev.p0; }
uint32_t shards_of(evah_ctx *c) { return c->dev.pstep; }
// limbs i < l owned by this shard
uint32_t nloc(evah_ctx *c, uint32_t l) {
  const uint32_t s = shard_of(c), G = shards_of(c);
  return l > s ? (l - s + G - 1) / G : 0;
}
void need_shard(evah_ctx *c) {
  if (c->dev.pstep < 1 || (c->dev.pstep == 1 && c->dev.p0 != 0)) throw std::logic_error("context has no shard map");
}
const KeyDev &shard_key(evah_ctx *c, int kind, uint32_t elt) {
  if (c->sh->key_rows == 2 && c->sh->key_shard != c->dev.p0) throw std::logic_error("the keys of this device state hold another shard's prime rows");
  if (kind == EVAH_KEY_RELIN) {
    if (!c->sh->relin.d) throw std::invalid_argument("relinearization key not present");
    return c->sh->relin;
  }
  auto it = c->sh->galois.find(elt);
  if (it == c->sh->galois.end()) throw std::invalid_argument("Galois key not present");
  return it->second;
}

} // namespace

extern "C" {

int evah_ctx_set_shard(evah_ctx *c, uint32_t shard, uint32_t n_shards) {
  API_BEGIN
  if (n_shards < 1 || n_shards > 64 || shard >= n_shards) throw std::invalid_argument("invalid shard index / count");
  if (c->sh->key_rows == 2 && (shard != c->dev.p0 || n_shards != c->dev.pstep))
    throw std::logic_error("the keys of this device state hold the prime rows of its current shard map");
  c->dev.p0 = shard;
  c->dev.pstep = n_shards;
  // scale checks see local limb counts: allow what the largest level with that many local limbs allows
  // (the caller, who knows the level, makes the exact check)
  for (uint32_t nl = 0; nl <= c->k; nl++) {
    const uint32_t lmax = std::min<uint32_t>(c->k - 1, shard + nl * n_shards);
    c->total_bits[nl] = bitlen_of_product(c->primes, std::max<uint32_t>(lmax, nl ? 1u : 0u));
  }
  API_END
}

int evah_ctx_shard_info(evah_ctx *c, uint32_t *shard, uint32_t *n_shards) {
  API_BEGIN
  *shard = c->dev.p0;
  *n_shards = c->dev.pstep;
  API_END
}

int evah_buf_alloc(evah_ctx *c, size_t words, evah_buf **out) {
  API_BEGIN
  use(c);
  if (!words) throw std::invalid_argument("empty buffer");
  auto *b = new evah_buf;
  b->buf = buf_new(c, words);
  b->words = words;
  *out = b;
  API_END
}
void evah_buf_free(evah_ctx *c, evah_buf *b) {
  if (!b) return;
  buf_unref(c, b->buf);
  delete b;
}
void *evah_buf_ptr(evah_buf *b) { return b ? b->buf->d : nullptr; }
size_t evah_buf_words(const evah_buf *b) { return b ? b->words : 0; }

// dst[dst_off ..) = src[src_off ..): device copy on `c`'s stream (peer copy across devices),
// ordered after the producer of src; c must be the context dst was allocated from
int evah_buf_copy(evah_ctx *c, evah_buf *dst, size_t dst_off, const evah_buf *src, size_t src_off, size_t words) {
  API_BEGIN
  use(c);
  if (dst_off + words > dst->words || src_off + words > src->words) throw std::invalid_argument("buffer copy out of range");
  acquire(c, src->buf);
  acquire(c, dst->buf);
  HIPCHK(hipMemcpyAsync(dst->buf->d + dst_off, src->buf->d + src_off, sizeof(u64) * words, hipMemcpyDefault, c->stream));
  API_END
}
// ONE launch that pulls n chunks into dst: chunk j = srcs[j][src_offs[j] .. + words) -> dst[dst_offs[j] ..).  The sources
// may live on other devices (peer reads over xGMI; evah_ctx_enable_peer must have succeeded for the pair): the all-gather
// of a key switch is one such launch per shard instead of G - 1 copies with an event wait each.
struct GatherTab {
  const u64 *src[KS_BATCH_MAX];
  uint32_t dst_tile[KS_BATCH_MAX]; // dst offsets in units of 2 words (16-byte accesses)
};
__global__ void __launch_bounds__(256) k_buf_gather(GatherTab t, u64 *dst, size_t pairs) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= pairs) return;
  const ulonglong2 v = reinterpret_cast<const ulonglong2 *>(t.src[blockIdx.y])[i];
  reinterpret_cast<ulonglong2 *>(dst)[(size_t)t.dst_tile[blockIdx.y] + i] = v;
}
int evah_buf_gather(evah_ctx *c, evah_buf *dst, uint32_t n, const evah_buf *const *srcs, const size_t *src_offs,
                    const size_t *dst_offs, size_t words) {
  API_BEGIN
  use(c);
  if (n < 1 || n > (uint32_t)KS_BATCH_MAX) throw std::invalid_argument("buffer gather takes 1..64 sources");
  if (!words || (words & 1)) throw std::invalid_argument("buffer gather moves an even, non-zero number of words");
  if (dst->buf->owner != c) throw std::invalid_argument("buffer gather runs on the queue that owns the destination");
  GatherTab t{};
  for (uint32_t j = 0; j < n; j++) {
    if (src_offs[j] + words > srcs[j]->words || dst_offs[j] + words > dst->words || (src_offs[j] & 1) || (dst_offs[j] & 1))
      throw std::invalid_argument("buffer gather out of range / misaligned");
    if ((dst_offs[j] >> 1) > 0xffffffffull) throw std::invalid_argument("buffer gather: destination offset too large");
    acquire(c, srcs[j]->buf); // ordered after the producer of every source, whatever device it is on
    t.src[j] = srcs[j]->buf->d + src_offs[j];
    t.dst_tile[j] = (uint32_t)(dst_offs[j] >> 1);
  }
  const size_t pairs = words / 2;
  ProfScope ps(c, KC_EW);
  hipLaunchKernelGGL(k_buf_gather, dim3((unsigned)((pairs + 255) / 256), n), dim3(256), 0, c->stream, t, dst->buf->d, pairs);
  HIPCHK(hipGetLastError());
  API_END
}

// Peer access between the devices of two contexts, both directions.  Without it a device-to-device copy may be staged
// through host memory and a kernel cannot read the other device's buffers at all — on the paths whose whole point is
// xGMI — so a refusal is an error here, not a silent fallback.  Contexts of one device: nothing to do.
int evah_ctx_enable_peer(evah_ctx *a, evah_ctx *b) {
  API_BEGIN
  if (a->device == b->device && !a->tun.peer_self_check) return 0;
  for (int dir = 0; dir < 2; dir++) {
    const int from = dir ? b->device : a->device, to = dir ? a->device : b->device;
    int can = 0;
    HIPCHK(hipDeviceCanAccessPeer(&can, from, to));
    if (!can)
      throw std::runtime_error("device " + std::to_string(from) + " cannot access device " + std::to_string(to) +
                               " as a peer (hipDeviceCanAccessPeer): the multi-GPU modes need peer access over xGMI / PCIe");
    HIPCHK(hipSetDevice(from));
    const hipError_t e = hipDeviceEnablePeerAccess(to, 0);
    if (e == hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError();
    else if (e != hipSuccess)
      throw std::runtime_error(std::string("hipDeviceEnablePeerAccess(") + std::to_string(to) + ") on device " + std::to_string(from) +
                               " failed: " + hipGetErrorString(e));
  }
  API_END
}

int evah_buf_download(evah_ctx *c, const evah_buf *b, size_t off, size_t words, uint64_t *host) {
  API_BEGIN
  use(c);
  if (off + words > b->words) throw std::invalid_argument("buffer range out of bounds");
  acquire(c, b->buf);
  HIPCHK(hipMemcpyAsync(host, b->buf->d + off, sizeof(u64) * words, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  API_END
}
int evah_buf_upload(evah_ctx *c, evah_buf *b, size_t off, size_t words, const uint64_t *host) {
  API_BEGIN
  use(c);
  if (off + words > b->words) throw std::invalid_argument("buffer range out of bounds");
  acquire(c, b->buf);
  HIPCHK(hipMemcpyAsync(b->buf->d + off, host, sizeof(u64) * words, hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  API_END
}

// local limbs of both polynomials of a size-2 ciphertext under the Galois automorphism (the
// permutation acts inside each limb, so it needs no exchange)
int evah_shard_galois_perm(evah_ctx *c, const evah_ct *a, uint32_t galois_elt, evah_ct **out) {
  API_BEGIN
  use(c);
  acquire(c, a->buf);
  if (a->size != 2 || a->batch != 1) throw std::invalid_argument("rotate expects a single size-2 ciphertext (relinearize first)");
  if (!(galois_elt & 1) || galois_elt >= 2 * c->N) throw std::invalid_argument("Galois element is not valid");
  const uint32_t *ptab = perm_table(c, galois_elt);
  evah_ct *o = ct_new(c, 2, a->limbs, a->scale);
  if (a->limbs) {
    galois_perm_launch(c, a->d, a->ps, a->limbs, 2, ptab, o->d, o->ps);
    HIPCHK(hipGetLastError());
  }
  *out = o;
  API_END
}

// phase 1 of a key switch at global level l: digits[shard][j] = INTT(a[poly][limb j]), j < nloc(l)
int evah_shard_ks_digits(evah_ctx *c, const evah_ct *a, uint32_t poly, uint32_t l, evah_buf *digits, uint32_t rows) {
  API_BEGIN
  use(c);
  need_shard(c);
  acquire(c, a->buf);
  acquire(c, digits->buf);
  const uint32_t nl = nloc(c, l), G = shards_of(c);
  if (poly >= a->size || a->limbs != nl || a->batch != 1) throw std::invalid_argument("key-switch target does not match the level");
  if (rows * G < l || digits->words < (size_t)G * rows * c->N) throw std::invalid_argument("digit buffer too small");
  if (nl) {
    OpPlain::Params ip{a->d + (size_t)poly * a->ps, digits->buf->d + (size_t)shard_of(c) * rows * c->N, 0, 0, nl, shard_of(c), 0, {}};
    ip.pstep = G;
    ntt_inverse<OpPlain>(c, ip, nl);
  }
  API_END
}

// phase 2: prod[K][iy] for the local output limbs I = shard + iy G (the special limb when l = shard
// mod G is the last row); the owner of the special limb also writes r[K] = INTT_P(prod[K][special]) + P/2
int evah_shard_ks_products(evah_ctx *c, const evah_ct *a, uint32_t poly, uint32_t l, const evah_buf *digits, uint32_t rows,
                           int key_kind, uint32_t galois_elt, evah_buf *prod, evah_buf *r) {
  API_BEGIN
  use(c);
  need_shard(c);
  if (!c->tun.fuse_mac) throw std::logic_error("limb-sharded key switching needs the fused key-switch kernel (EVAH_FUSE_MAC=1)");
  if (a) acquire(c, a->buf); // a == NULL: a shard that owns only the special limb at this level
  acquire(c, digits->buf);
  acquire(c, prod->buf);
  const uint32_t s = shard_of(c), G = shards_of(c), nl = nloc(c, l);
  const bool owner = (l % G) == s;
  const uint32_t ni = nl + (owner ? 1u : 0u);
  const size_t N = c->N;
  if (nl ? (!a || poly >= a->size || a->limbs != nl) : a != nullptr) throw std::invalid_argument("key-switch target does not match the level");
  if (prod->words < (size_t)2 * std::max(ni, 1u) * N) throw std::invalid_argument("product buffer too small");
  const KeyDev &key = shard_key(c, key_kind, galois_elt);
  if (key.n_digits < l) throw std::runtime_error("key switching key has too few digits");
  if (ni) {
    Scratch sc(c, (size_t)ni * l * N);
    OpKsDigit::Params dp{digits->buf->d, sc.d, l, 0, 0, s, ni};
    dp.istep = G;
    dp.t_split = G;
    dp.t_rows = rows;
    launch_pass_p<true, false, OpKsDigit>(c, (c->logN + 1) / 2, dp, ni * l);
    KsBatch kb;
    kb.n = 1;
    kb.i0 = s;
    kb.ni = ni;
    kb.istep = G;
    kb.nout = ni;
    kb.keys.key[0] = key.d;
    kb.keys.rows = key.rows == c->k ? 0 : key.rows;
    launch_ks_inner(c, c->logN / 2, a ? a->d + (size_t)poly * a->ps : nullptr, sc.d, kb, prod->buf->d, l);
  }
  if (owner) {
    acquire(c, r->buf);
    if (r->words < 2 * N) throw std::invalid_argument("r buffer too small");
    OpPlain::Params sp{prod->buf->d + (size_t)(ni - 1) * N, r->buf->d, (size_t)ni * N, N, 1, c->k - 1, 1, {}};
    ntt_inverse<OpPlain>(c, sp, 2);
  }
  API_END
}

// phase 3: out[K][j] = (add ? add[K][j] : 0) + (prod[K][j] - NTT_j((r_K mod q_j) - P/2 mod q_j)) P^-1, local data limbs
int evah_shard_ks_finish(evah_ctx *c, uint32_t l, const evah_buf *prod, const evah_buf *r, const evah_ct *add, uint32_t add_polys,
                         double scale, evah_ct **out) {
  API_BEGIN
  use(c);
  need_shard(c);
  acquire(c, prod->buf);
  acquire(c, r->buf);
  if (add) acquire(c, add->buf);
  const uint32_t s = shard_of(c), G = shards_of(c), nl = nloc(c, l);
  const uint32_t ni = nl + (((l % G) == s) ? 1u : 0u);
  const size_t N = c->N;
  if (add && (add->limbs != nl || add->size < add_polys)) throw std::invalid_argument("ciphertext to add does not match the level");
  evah_ct *o = ct_new(c, 2, nl, scale);
  if (nl) {
    OpModDown::Params mp{r->buf->d, N, prod->buf->d, (size_t)ni * N, add ? add->d : nullptr, add ? add->ps : 0, add_polys, o->d, o->ps,
                         c->k - 1, nl};
    ntt_forward<OpModDown>(c, mp, 2 * nl);
  }
  *out = o;
  API_END
}

// rescale phase 1, on the owner of limb l - 1 only: r[p] = INTT(a[p][last]) + q_last / 2
int evah_shard_rescale_last(evah_ctx *c, const evah_ct *a, uint32_t l, evah_buf *r) {
  API_BEGIN
  use(c);
  need_shard(c);
  acquire(c, a->buf);
  acquire(c, r->buf);
  const uint32_t s = shard_of(c), G = shards_of(c), nl = nloc(c, l);
  if (l < 2) throw std::invalid_argument("end of modulus switching chain reached");
  if (((l - 1) % G) != s) throw std::logic_error("only the owner of the last limb runs this phase");
  if (a->limbs != nl || a->batch != 1) throw std::invalid_argument("ciphertext does not match the level");
  if (r->words < (size_t)a->size * c->N) throw std::invalid_argument("r buffer too small");
  OpPlain::Params ip{a->d + (size_t)(nl - 1) * c->N, r->buf->d, a->ps, c->N, 1, l - 1, 1, {}};
  ntt_inverse<OpPlain>(c, ip, a->size);
  API_END
}

// rescale phase 2: out[p][j] = (a[p][j] - NTT_j((r_p mod q_j) - q_last/2 mod q_j)) q_last^-1 on the local limbs of level l - 1
int evah_shard_rescale_finish(evah_ctx *c, const evah_ct *a, uint32_t l, const evah_buf *r, uint32_t divisor_bits, evah_ct **out) {
  API_BEGIN
  use(c);
  need_shard(c);
  acquire(c, a->buf);
  acquire(c, r->buf);
  if (l < 2) throw std::invalid_argument("end of modulus switching chain reached");
  const uint32_t nl = nloc(c, l), nn = nloc(c, l - 1);
  if (a->limbs != nl || a->batch != 1) throw std::invalid_argument("ciphertext does not match the level");
  evah_ct *o = ct_new(c, a->size, nn, a->scale / std::pow(2.0, (double)divisor_bits));
  if (nn) {
    OpModDown::Params mp{r->buf->d, c->N, a->d, a->ps, nullptr, 0, 0, o->d, o->ps, l - 1, nn};
    ntt_forward<OpModDown>(c, mp, a->size * nn);
  }
  *out = o;
  API_END
}

} // extern "C"
