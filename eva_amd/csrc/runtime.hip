// runtime.hip — libeva_hip.so: contexts, issue queues, keys, value transfer, graph capture,
// profiling hooks.  Replaces, for EVA's execute() hot path, the SEALContext / key objects and the
// setInputs / getOutputs copies of SEALExecutor (/root/reference/eva/seal/seal.h:52-66,
// seal_executor.h:264-277, 420-435) — see include/eva_hip.h for the per-entry-point mapping.
// gfx950 only; no CPU fallback: every entry point needs a HIP device and fails otherwise.
#include "internal.hip.h"
namespace evah {
void key_split_launch(evah_ctx *c, const u64 *src, u64 *dst, size_t words); // elementwise.hip
void ct_stack_launch(evah_ctx *c, const evah_ct *const *cts, uint32_t n, evah_ct *o); // elementwise.hip
}

namespace evah {

thread_local std::string g_err;

static std::mutex g_ctx_mu;
static std::vector<const void *> g_live_ctx;
void ctx_register(const void *c) {
  std::lock_guard<std::mutex> lk(g_ctx_mu);
  g_live_ctx.push_back(c);
}
void ctx_unregister(const void *c) {
  std::lock_guard<std::mutex> lk(g_ctx_mu);
  g_live_ctx.erase(std::remove(g_live_ctx.begin(), g_live_ctx.end(), c), g_live_ctx.end());
}
bool ctx_alive(const void *c) {
  std::lock_guard<std::mutex> lk(g_ctx_mu);
  return std::find(g_live_ctx.begin(), g_live_ctx.end(), c) != g_live_ctx.end();
}

} // namespace evah

extern "C" {

const char *evah_last_error(void) { return g_err.c_str(); }
int evah_abi_version(void) { return 1; }

int evah_device_count(int *count) {
  API_BEGIN
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    n = 0;
  }
  *count = n;
  API_END
}

int evah_ctx_create(uint32_t N, uint32_t k, const uint64_t *primes, int device, evah_ctx **out) {
  API_BEGIN
  if (N < 1024 || N > 131072 || (N & (N - 1))) throw std::invalid_argument("poly_modulus_degree must be a power of two in [1024, 131072]");
  if (k < 2 || k > 62) throw std::invalid_argument("need at least one data prime and one special prime");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
    throw std::runtime_error("libeva_hip: no HIP device available (this backend has no CPU fallback)");
  if (device < 0 || device >= ndev) throw std::invalid_argument("invalid device index");
  auto *c = new evah_ctx;
  try {
    c->device = device;
    c->N = N;
    c->logN = ilog2(N);
    c->k = k;
    c->primes.assign(primes, primes + k);
    c->tun = Tunables::from_env(N);
    for (u64 q : c->primes)
      if (q >= ((u64)1 << 60) || (q - 1) % (2ull * N) || !is_prime(q)) // SEAL_USER_MOD_BIT_COUNT_MAX = 60
        throw std::invalid_argument("coeff modulus primes must be at most 60 bits, prime and 1 mod 2N");
    for (uint32_t l = 0; l <= k; l++) c->total_bits.push_back(l ? bitlen_of_product(c->primes, l) : 0);
    use(c);
    HIPCHK(hipStreamCreateWithFlags(&c->own, hipStreamNonBlocking));
    c->stream = c->own;
    HIPCHK(hipEventCreate(&c->ev0));
    HIPCHK(hipEventCreate(&c->ev1));
    // ---- tables: [primes k][tw_fwd k*N][tw_inv k*N][invq k*k][modq k*k][plinv k*k][halfmod k*k]
    const size_t sz_pr = sizeof(DevPrime) * k, sz_tw = sizeof(ulonglong2) * (size_t)k * N,
                 sz_iq = sizeof(ulonglong2) * (size_t)k * k, sz_hm = sizeof(u64) * (size_t)k * k;
    const size_t total = sz_pr + 2 * sz_tw + 3 * sz_iq + sz_hm;
    std::vector<unsigned char> host(total);
    auto *hp = reinterpret_cast<DevPrime *>(host.data());
    auto *hf = reinterpret_cast<ulonglong2 *>(host.data() + sz_pr);
    auto *hi = reinterpret_cast<ulonglong2 *>(host.data() + sz_pr + sz_tw);
    auto *hq = reinterpret_cast<ulonglong2 *>(host.data() + sz_pr + 2 * sz_tw);
    auto *hm = reinterpret_cast<ulonglong2 *>(host.data() + sz_pr + 2 * sz_tw + sz_iq);
    auto *hl = reinterpret_cast<ulonglong2 *>(host.data() + sz_pr + 2 * sz_tw + 2 * sz_iq);
    auto *hh = reinterpret_cast<u64 *>(host.data() + sz_pr + 2 * sz_tw + 3 * sz_iq);
    for (uint32_t i = 0; i < k; i++) {
      const u64 q = c->primes[i];
      const u64 psi = minimal_primitive_root(N, q), psi_inv = invmod(psi, q);
      std::vector<u64> rp = root_power_table(N, q, psi), irp = root_power_table(N, q, psi_inv);
      for (uint32_t j = 0; j < N; j++) {
        hf[(size_t)i * N + j] = make_ulonglong2(rp[j], shoup(rp[j], q));
        hi[(size_t)i * N + j] = make_ulonglong2(irp[j], shoup(irp[j], q));
      }
      DevPrime &d = hp[i];
      d.q = q;
      d.brt = (u64)((((u128)1) << 64) / q);
      u128 ratio = (~(u128)0) / q;
      d.r0 = (u64)ratio;
      d.r1 = (u64)(ratio >> 64);
      d.ninv = invmod(N % q, q);
      d.ninv_s = shoup(d.ninv, q);
      d.w0ninv = mulmod(irp[1], d.ninv, q);
      d.w0ninv_s = shoup(d.w0ninv, q);
      d.nq = 0ull - q;
      d.q5 = 5 * q;
      d.q4 = 4 * q;
      d.q8 = 8 * q;
      d.nq5 = 0ull - 5 * q;
      d.nq8 = 0ull - 8 * q;
      d.c64 = (u64)((((u128)1) << 64) % q);
      d.c64s = shoup(d.c64, q);
      {
        uint32_t b = 0;
        while ((q >> b) != 0) b++;
        const u64 cc = (b < 64 ? ((u64)1 << b) : 0) - q;
        // top-bit reduction x -> (x mod 2^b) + (x >> b) c leaves x < q + 16 c for x < 16 q, and a pass then runs THREE
        // lazy stages (+ 4 q each) before the next reduction: q + 16 c + 12 q < 16 q needs 16 c < 3 q.  Every
        // CoeffModulus::Create prime has c < 2^(b-10); a caller's or a file's 33/34-bit prime far below 2^b may not
        // (b = 33, q = 1.2 * 2^32: c = 0.8 * 2^32 passes "c < 2^32" and breaks the bound), so the shape also needs
        // c < q / 16 — other primes take the compare-and-subtract butterflies
        const bool ok = b > 32 && cc < ((u64)1 << 32) && cc < (q >> 4);
        d.tb_c = ok ? (uint32_t)cc : 0;
        d.tb_sh = ok ? b - 32 : 0;
        // a prime of another shape: (c, shift, mask) = (0, 0, ~0) makes the top-bit reduction the identity.  The ordinary
        // passes never use it (tb_c == 0 selects the compare-and-subtract butterflies); ks_inner_kernel<MAC3>, whose rounds
        // are compiled with the top-bit form only, then runs such a row without any reduction — correct while the 8-stage
        // lazy growth (9 q in, + 4 q per stage: < 41 q) stays below the 2^60 the radix-2^30 split needs, i.e. q < 2^54
        // (mac3_ok below): the 20..50-bit scale primes of EVA's chains
        d.tb_mask = ok ? (uint32_t)(((u64)1 << (b - 32)) - 1) : 0xffffffffu;
        d.tb_pad = 0;
      }
      for (uint32_t a = 0; a < k; a++) {
        const u64 qa = c->primes[a];
        if (a == i) {
          hq[a * k + i] = make_ulonglong2(0, 0);
          hm[a * k + i] = make_ulonglong2(0, 0);
          hl[a * k + i] = make_ulonglong2(0, 0);
          hh[a * k + i] = 0;
        } else {
          u64 inv = invmod(qa % q, q);
          // P q_a^-1 mod q_i (P = the special prime; 0 under P itself): the chain step's folded rescale (DevCtx::plinv)
          const u64 pl = mulmod(c->primes[k - 1] % q, inv, q);
          hl[a * k + i] = make_ulonglong2(pl, shoup(pl, q));
          hq[a * k + i] = make_ulonglong2(inv, shoup(inv, q));
          hm[a * k + i] = make_ulonglong2(qa % q, shoup(qa % q, q));
          hh[a * k + i] = (qa >> 1) % q;
        }
      }
    }
    c->all_tb = true;
#if defined(EVAH_TBMUL) && EVAH_TBMUL // (the build-time experiment multiplies by c = 2^b - q in every top-bit butterfly: strict shape)
    for (uint32_t i = 0; i < k; i++) c->all_tb = c->all_tb && hp[i].tb_c != 0;
#else
    for (uint32_t i = 0; i < k; i++) c->all_tb = c->all_tb && (hp[i].tb_c != 0 || hp[i].q < ((u64)1 << 54));
#endif
    c->sh = std::make_shared<SharedDev>();
    c->sh->device = device;
    HIPCHK(hipMalloc(&c->sh->d_tables, total));
    h2d_now(c, c->sh->d_tables, host.data(), total);
    auto *base = reinterpret_cast<unsigned char *>(c->sh->d_tables);
    c->dev.primes = reinterpret_cast<const DevPrime *>(base);
    c->dev.tw_fwd = reinterpret_cast<const ulonglong2 *>(base + sz_pr);
    c->dev.tw_inv = reinterpret_cast<const ulonglong2 *>(base + sz_pr + sz_tw);
    c->dev.invq = reinterpret_cast<const ulonglong2 *>(base + sz_pr + 2 * sz_tw);
    c->dev.modq = reinterpret_cast<const ulonglong2 *>(base + sz_pr + 2 * sz_tw + sz_iq);
    c->dev.plinv = reinterpret_cast<const ulonglong2 *>(base + sz_pr + 2 * sz_tw + 2 * sz_iq);
    c->dev.halfmod = reinterpret_cast<const u64 *>(base + sz_pr + 2 * sz_tw + 3 * sz_iq);
    c->dev.N = N;
    c->dev.logN = c->logN;
    c->dev.k = k;
    c->dev.p0 = 0;
    c->dev.pstep = 1;
  } catch (...) {
    evah_ctx_destroy(c);
    throw;
  }
  ctx_register(c);
  *out = c;
  API_END
}

int evah_ctx_fork(evah_ctx *parent, evah_ctx **out) {
  API_BEGIN
  use(parent);
  auto *c = new evah_ctx;
  try {
    c->sh = parent->sh;
    c->device = parent->device;
    c->N = parent->N;
    c->logN = parent->logN;
    c->k = parent->k;
    c->primes = parent->primes;
    c->total_bits = parent->total_bits;
    c->dev = parent->dev;
    c->tun = parent->tun;
    c->all_tb = parent->all_tb;
    HIPCHK(hipStreamCreateWithFlags(&c->own, hipStreamNonBlocking));
    c->stream = c->own;
    HIPCHK(hipEventCreate(&c->ev0));
    HIPCHK(hipEventCreate(&c->ev1));
  } catch (...) {
    evah_ctx_destroy(c);
    throw;
  }
  ctx_register(c);
  *out = c;
  API_END
}

void evah_ctx_destroy(evah_ctx *c) {
  if (!c) return;
  ctx_unregister(c);
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  if (c->side) { (void)hipStreamSynchronize(c->side); (void)hipStreamDestroy(c->side); c->side = nullptr; }
  c->pool.release_cached();
  c->sh.reset(); // tables and keys go when the last fork goes
  for (auto &r : c->prof_recs) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
  for (auto e : c->prof_free) (void)hipEventDestroy(e);
  for (auto e : c->sync_events) (void)hipEventDestroy(e);
  for (auto e : c->capture_events) (void)hipEventDestroy(e);
  if (c->ev0) (void)hipEventDestroy(c->ev0);
  if (c->ev1) (void)hipEventDestroy(c->ev1);
  if (c->own) (void)hipStreamDestroy(c->own);
  delete c;
}

int evah_ctx_set_stream(evah_ctx *c, void *s) {
  API_BEGIN
  use(c);
  HIPCHK(hipStreamSynchronize(c->stream)); // pool reuse is ordered per stream
  c->stream = s ? (hipStream_t)s : c->own;
  API_END
}

int evah_ctx_busy(evah_ctx *c, int *busy) {
  API_BEGIN
  use(c);
  const hipError_t e = hipStreamQuery(c->stream);
  if (e != hipSuccess && e != hipErrorNotReady) HIPCHK(e);
  if (e == hipErrorNotReady) (void)hipGetLastError(); // (not an error: clear the runtime's sticky state)
  *busy = e == hipErrorNotReady ? 1 : 0;
  API_END
}

int evah_ctx_sync(evah_ctx *c) {
  API_BEGIN
  use(c);
  HIPCHK(hipStreamSynchronize(c->stream));
  API_END
}

int evah_ctx_mem_info(evah_ctx *c, size_t *in_use, size_t *cached) {
  API_BEGIN
  *in_use = c->pool.in_use;
  *cached = c->pool.cached;
  API_END
}

int evah_galois_elt_from_step(evah_ctx *c, int32_t steps, uint32_t *elt) {
  API_BEGIN
  const uint32_t N = c->N, m = 2 * N;
  if (steps == 0) {
    *elt = m - 1;
  } else {
    uint32_t pos = steps < 0 ? (uint32_t)(-(int64_t)steps) : (uint32_t)steps;
    if (pos >= (N >> 1)) throw std::invalid_argument("step count too large");
    uint32_t s = steps < 0 ? (N >> 1) - pos : pos, e = 1;
    for (uint32_t i = 0; i < s; i++) e = (e * 3u) & (m - 1);
    *elt = e;
  }
  API_END
}

int evah_key_upload(evah_ctx *c, int kind, uint32_t galois_elt, uint32_t n_digits, const uint64_t *data) {
  API_BEGIN
  use(c);
  if (c->capturing) throw std::logic_error("this call synchronises with the host and cannot be captured into a graph");
  if (n_digits == 0 || n_digits > c->k - 1) throw std::invalid_argument("invalid key digit count");
  KeyDev kd;
  kd.n_digits = n_digits;
  // A limb shard (evah_ctx_set_shard before the upload) multiplies digits only into its own limbs and,
  // when it owns it at that level, the special prime: it keeps those prime rows of the key and nothing else
  // — (ceil((k-1)/G) + 1) / k of the key per shard.
  const uint32_t s = c->dev.p0, G = c->dev.pstep;
  const bool local_rows = G > 1;
  if (c->sh->key_rows && c->sh->key_rows != (local_rows ? 1u : 0u) + 1u)
    throw std::logic_error("the keys of this device state were uploaded under a different shard map");
  kd.rows = local_rows ? (c->k - 1 > s ? (c->k - 1 - s + G - 1) / G : 0) + 1 : c->k;
  kd.bytes = sizeof(u64) * (size_t)n_digits * 2 * kd.rows * c->N;
  HIPCHK(hipMalloc(&kd.d, kd.bytes));
  if (!local_rows) {
    h2d_now(c, kd.d, data, kd.bytes);
  } else {
    const size_t row = sizeof(u64) * c->N;
    hipError_t e = hipSuccess;
    for (uint32_t dk = 0; dk < 2 * n_digits && e == hipSuccess; dk++) {
      // rows s, s + G, ... of this (digit, polynomial): one strided copy; then the special prime's row
      const u64 *src = (const u64 *)data + (size_t)dk * c->k * c->N;
      u64 *dst = kd.d + (size_t)dk * kd.rows * c->N;
      for (uint32_t r = 0; r + 1 < kd.rows && e == hipSuccess; r++) // (linear copies: see evah_ct_download on 2-D copies and pageable memory)
        e = hipMemcpyAsync(dst + (size_t)r * c->N, src + (size_t)(s + r * G) * c->N, row, hipMemcpyHostToDevice, c->stream);
      if (e == hipSuccess) e = hipMemcpyAsync(dst + (size_t)(kd.rows - 1) * c->N, src + (size_t)(c->k - 1) * c->N, row, hipMemcpyHostToDevice, c->stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream); // complete before any queue reads the rows (h2d_now)
    if (e != hipSuccess) {
      (void)hipFree(kd.d);
      HIPCHK(e);
    }
  }
  if (!local_rows && c->all_tb && c->tun.mac3) { // the split copy ks_inner_kernel<MAC3> multiplies with
    if (hipMalloc(&kd.d_split, kd.bytes) != hipSuccess) {
      (void)hipGetLastError();
      kd.d_split = nullptr; // no memory for the second copy: the 128-bit accumulation is used
    } else {
      key_split_launch(c, kd.d, kd.d_split, kd.bytes / sizeof(u64));
      HIPCHK(hipStreamSynchronize(c->stream));
    }
  }
  c->sh->key_rows = local_rows ? 2 : 1;
  c->sh->key_shard = s;
  if (kind == EVAH_KEY_RELIN) {
    if (c->sh->relin.d) (void)hipFree(c->sh->relin.d);
    if (c->sh->relin.d_split) (void)hipFree(c->sh->relin.d_split);
    c->sh->relin = kd;
  } else if (kind == EVAH_KEY_GALOIS) {
    if (!(galois_elt & 1) || galois_elt >= 2 * c->N) {
      (void)hipFree(kd.d);
      if (kd.d_split) (void)hipFree(kd.d_split);
      throw std::invalid_argument("Galois element is not valid");
    }
    auto it = c->sh->galois.find(galois_elt);
    if (it != c->sh->galois.end()) {
      (void)hipFree(it->second.d);
      if (it->second.d_split) (void)hipFree(it->second.d_split);
      if (it->second.d_perm) (void)hipFree(it->second.d_perm);
    }
    c->sh->galois[galois_elt] = kd;
    // the hoisting constants of this element were derived from the key it replaces
    for (auto hc = c->sh->hoist_corr.begin(); hc != c->sh->hoist_corr.end();) {
      if (hc->first.first == galois_elt) {
        (void)hipFree(hc->second);
        hc = c->sh->hoist_corr.erase(hc);
      } else {
        ++hc;
      }
    }
  } else {
    (void)hipFree(kd.d);
    if (kd.d_split) (void)hipFree(kd.d_split);
    throw std::invalid_argument("unknown key kind");
  }
  API_END
}

// Pinned (page-locked) host memory for the values that cross the boundary: copies from it are
// DMA transfers at PCIe rate instead of the runtime's staged pageable path (~10 GB/s on one
// core).  Blocks are recycled by size — pinning is far too slow to do per value.
namespace {
std::mutex g_host_mu;
std::unordered_multimap<size_t, void *> g_host_free; // size -> idle pinned block
std::unordered_map<void *, size_t> g_host_live;       // block handed out -> size
size_t g_host_cached = 0;
} // namespace
void *evah_host_alloc(size_t bytes) {
  if (!bytes) return nullptr;
  std::lock_guard<std::mutex> lk(g_host_mu);
  auto it = g_host_free.find(bytes);
  void *p = nullptr;
  if (it != g_host_free.end()) {
    p = it->second;
    g_host_free.erase(it);
    g_host_cached -= bytes;
  } else if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) {
    (void)hipGetLastError();
    return nullptr; // no device / no memory: the caller falls back to ordinary memory
  }
  g_host_live.emplace(p, bytes);
  return p;
}
void evah_host_free(void *p) {
  if (!p) return;
  std::lock_guard<std::mutex> lk(g_host_mu);
  auto it = g_host_live.find(p);
  if (it == g_host_live.end()) return;
  const size_t bytes = it->second;
  g_host_live.erase(it);
  if (g_host_cached + bytes > ((size_t)4 << 30)) { // keep at most 4 GiB idle
    (void)hipHostFree(p);
    return;
  }
  g_host_free.emplace(bytes, p);
  g_host_cached += bytes;
}

// Host <-> device copies of the instances of a batched handle, back to back on the context's
// stream.  (Measured on MI355X: splitting them over 4 host threads with a copy stream each is
// slower — 5.1 k vs 6.9 k Sobel DAGs/s — the pageable staging path of the runtime serialises.)
static void io_copy(evah_ctx *c, uint32_t n, const std::function<hipError_t(uint32_t, hipStream_t)> &copy_one, bool wait = true) {
  for (uint32_t b = 0; b < n; b++) HIPCHK(copy_one(b, c->stream));
  if (wait) {
    HIPCHK(hipStreamSynchronize(c->stream));
  }
}

int evah_ct_upload(evah_ctx *c, uint32_t size, uint32_t limbs, double scale, const uint64_t *data, evah_ct **out) {
  API_BEGIN
  use(c);
  if (c->capturing) throw std::logic_error("this call synchronises with the host and cannot be captured into a graph");
  if (size < 1 || size > 3) throw std::invalid_argument("ciphertext size must be 1..3");
  if (limbs < 1 || limbs > c->k - 1) throw std::invalid_argument("invalid limb count for this context");
  evah_ct *t = ct_new(c, size, limbs, scale);
  HIPCHK(hipMemcpyAsync(t->d, data, sizeof(u64) * (size_t)size * limbs * c->N, hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  count_h2d(c, sizeof(u64) * (size_t)size * limbs * c->N);
  t->buf->ready_everywhere = true;
  *out = t;
  API_END
}

// `batch` ciphertexts of one shape as ONE handle; data = [batch][size][limbs][N]
int evah_ct_upload_batch(evah_ctx *c, uint32_t batch, uint32_t size, uint32_t limbs, double scale, const uint64_t *data,
                         evah_ct **out) {
  API_BEGIN
  use(c);
  if (c->capturing) throw std::logic_error("this call synchronises with the host and cannot be captured into a graph");
  if (batch < 1 || batch > (uint32_t)KS_BATCH_MAX) throw std::invalid_argument("batch must be 1..64");
  if (size < 1 || size > 3) throw std::invalid_argument("ciphertext size must be 1..3");
  if (limbs < 1 || limbs > c->k - 1) throw std::invalid_argument("invalid limb count for this context");
  evah_ct *t = ct_new(c, size, limbs, scale, batch);
  HIPCHK(hipMemcpyAsync(t->d, data, sizeof(u64) * (size_t)batch * size * limbs * c->N, hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  count_h2d(c, sizeof(u64) * (size_t)batch * size * limbs * c->N);
  t->buf->ready_everywhere = true;
  *out = t;
  API_END
}

// the same from `batch` separate host arrays (each [size][limbs][N]): no host-side concatenation
static void ct_upload_instances(evah_ctx *c, uint32_t batch, uint32_t size, uint32_t limbs, double scale,
                                const uint64_t *const *data, evah_ct **out, bool wait) {
  use(c);
  if (c->capturing) throw std::logic_error("host transfers cannot be captured into a graph");
  if (batch < 1 || batch > (uint32_t)KS_BATCH_MAX) throw std::invalid_argument("batch must be 1..64");
  if (size < 1 || size > 3) throw std::invalid_argument("ciphertext size must be 1..3");
  if (limbs < 1 || limbs > c->k - 1) throw std::invalid_argument("invalid limb count for this context");
  evah_ct *t = ct_new(c, size, limbs, scale, batch);
  const size_t each = (size_t)size * limbs * c->N;
  try {
    io_copy(c, batch, [&](uint32_t b, hipStream_t st) {
      return hipMemcpyAsync(t->d + each * b, data[b], sizeof(u64) * each, hipMemcpyHostToDevice, st);
    }, wait);
  } catch (...) {
    evah_ct_free(c, t);
    throw;
  }
  count_h2d(c, sizeof(u64) * each * batch);
  *out = t;
}
int evah_ct_upload_instances(evah_ctx *c, uint32_t batch, uint32_t size, uint32_t limbs, double scale,
                             const uint64_t *const *data, evah_ct **out) {
  API_BEGIN
  ct_upload_instances(c, batch, size, limbs, scale, data, out, true);
  API_END
}
// stream-ordered form: the copies are enqueued on the context's queue and the call returns; the
// host arrays must stay untouched until evah_ctx_sync(ctx) (pinned memory from evah_host_alloc
// makes the copies overlap other queues' kernels; pageable memory is staged by the runtime)
int evah_ct_upload_instances_async(evah_ctx *c, uint32_t batch, uint32_t size, uint32_t limbs, double scale,
                                   const uint64_t *const *data, evah_ct **out) {
  API_BEGIN
  ct_upload_instances(c, batch, size, limbs, scale, data, out, false);
  API_END
}

// instance b of a batched handle -> out[b] ([size][limbs][N] each), all instances in one call
static void ct_download_instances(evah_ctx *c, const evah_ct *ct, uint64_t *const *out, bool wait) {
  use(c);
  if (c->capturing) throw std::logic_error("host transfers cannot be captured into a graph");
  acquire(c, ct->buf);
  const size_t row = sizeof(u64) * (size_t)ct->limbs * c->N;
  const bool dense = ct->ps == (size_t)ct->limbs * c->N; // not a mod-switched view: one linear copy per instance
  io_copy(c, ct->batch, [&](uint32_t b, hipStream_t st) {
    const u64 *src = ct->d + (size_t)b * ct->size * ct->ps;
    if (dense) return hipMemcpyAsync(out[b], src, row * ct->size, hipMemcpyDeviceToHost, st);
    // a mod-switched view: one linear copy per polynomial (not hipMemcpy2DAsync — see evah_ct_download)
    hipError_t e = hipSuccess;
    for (uint32_t p = 0; p < ct->size && e == hipSuccess; p++)
      e = hipMemcpyAsync(reinterpret_cast<char *>(out[b]) + p * row, src + (size_t)p * ct->ps, row, hipMemcpyDeviceToHost, st);
    return e;
  }, wait);
  count_d2h(c, row * ct->size * ct->batch);
}
int evah_ct_download_instances(evah_ctx *c, const evah_ct *ct, uint64_t *const *out) {
  API_BEGIN
  ct_download_instances(c, ct, out, true);
  API_END
}
// stream-ordered form: out[b] holds the data after evah_ctx_sync(ctx); the handle may be freed right
// after this call (the pool recycles in queue order)
int evah_ct_download_instances_async(evah_ctx *c, const evah_ct *ct, uint64_t *const *out) {
  API_BEGIN
  ct_download_instances(c, ct, out, false);
  API_END
}

int evah_ct_batch(const evah_ct *ct, uint32_t *batch) {
  API_BEGIN
  *batch = ct->batch;
  API_END
}

// n single ciphertexts of one shape and scale -> one batched handle (device copies)
int evah_ct_stack(evah_ctx *c, const evah_ct *const *cts, uint32_t n, evah_ct **out) {
  API_BEGIN
  use(c);
  if (n < 1 || n > (uint32_t)KS_BATCH_MAX) throw std::invalid_argument("batch must be 1..64");
  const evah_ct *f = cts[0];
  for (uint32_t i = 0; i < n; i++) {
    if (cts[i]->batch != 1) throw std::invalid_argument("stack takes single ciphertexts");
    if (cts[i]->size != f->size || cts[i]->limbs != f->limbs) throw std::invalid_argument("encrypted parameter mismatch in batch");
    if (!same_scale(cts[i]->scale, f->scale)) throw std::invalid_argument("scale mismatch");
    acquire(c, cts[i]->buf);
  }
  evah_ct *o = ct_new(c, f->size, f->limbs, f->scale, n);
  try {
    ct_stack_launch(c, cts, n, o); // one gather launch (r6; a 2-D copy per instance before)
  } catch (...) {
    evah_ct_free(c, o);
    throw;
  }
  *out = o;
  API_END
}

// instance b of a batched handle as a single-ciphertext view (shares the buffer)
int evah_ct_unstack(evah_ctx *c, const evah_ct *ct, uint32_t b, evah_ct **out) {
  API_BEGIN
  (void)c;
  if (b >= ct->batch) throw std::invalid_argument("instance index out of range");
  evah_ct *o = new evah_ct(*ct);
  o->d = ct->d + (size_t)b * ct->size * ct->ps;
  o->batch = 1;
  o->buf->refs++;
  *out = o;
  API_END
}

// a copy of `src` owned by `c` — another device (peer copy over xGMI) or another queue of the same
// device; ordered after the producer of src, asynchronous on c's stream
int evah_ct_copy(evah_ctx *c, const evah_ct *src, evah_ct **out) {
  API_BEGIN
  use(c);
  acquire(c, src->buf);
  evah_ct *o = ct_new(c, src->size, src->limbs, src->scale, src->batch);
  const size_t row = sizeof(u64) * (size_t)src->limbs * c->N, polys = (size_t)src->size * src->batch;
  if (src->ps == (size_t)src->limbs * c->N)
    HIPCHK(hipMemcpyAsync(o->d, src->d, row * polys, hipMemcpyDefault, c->stream));
  else // a mod-switched view: gather its rows
    HIPCHK(hipMemcpy2DAsync(o->d, row, src->d, sizeof(u64) * src->ps, row, polys, hipMemcpyDefault, c->stream));
  *out = o;
  API_END
}
int evah_pt_copy(evah_ctx *c, const evah_pt *src, evah_pt **out) {
  API_BEGIN
  use(c);
  acquire(c, src->buf);
  evah_pt *o = pt_new(c, src->limbs, src->scale);
  HIPCHK(hipMemcpyAsync(o->d, src->d, sizeof(u64) * (size_t)src->limbs * c->N, hipMemcpyDefault, c->stream));
  o->uniform = src->uniform;
  *out = o;
  API_END
}

// refill of an existing handle from another handle of the same shape, device to device
int evah_ct_assign(evah_ctx *c, evah_ct *dst, const evah_ct *src) {
  API_BEGIN
  use(c);
  if (c->capturing) throw std::logic_error("evah_ct_assign cannot be captured into a graph");
  if (dst->size != src->size || dst->limbs != src->limbs || dst->batch != src->batch)
    throw std::invalid_argument("evah_ct_assign: shapes differ");
  if (dst->ps != (size_t)dst->limbs * c->N) throw std::invalid_argument("cannot write into a mod-switched view");
  // the per-buffer ordering treats values as immutable after their producer: a rewrite is only ordered on the
  // queue that owns the buffer (its later reads follow on the same stream)
  if (dst->buf->owner != c) throw std::invalid_argument("evah_ct_assign: a value is rewritten on the queue that owns it");
  acquire(c, dst->buf);
  acquire(c, src->buf);
  const size_t row = sizeof(u64) * (size_t)src->limbs * c->N, polys = (size_t)src->size * src->batch;
  if (src->ps == (size_t)src->limbs * c->N)
    HIPCHK(hipMemcpyAsync(dst->d, src->d, row * polys, hipMemcpyDefault, c->stream));
  else
    HIPCHK(hipMemcpy2DAsync(dst->d, row, src->d, sizeof(u64) * src->ps, row, polys, hipMemcpyDefault, c->stream));
  dst->scale = src->scale;
  API_END
}

int evah_ctx_wait(evah_ctx *waiter, evah_ctx *signaller) {
  API_BEGIN
  use(waiter);
  if (waiter->capturing || signaller->capturing) throw std::logic_error("evah_ctx_wait cannot be captured into a graph");
  stream_wait(waiter, signaller);
  API_END
}

int evah_ctx_transfer_stats(evah_ctx *c, uint64_t out[6]) {
  API_BEGIN
  for (int i = 0; i < 6; i++) out[i] = c->sh->xfer[i];
  API_END
}

int evah_ctx_key_bytes(evah_ctx *c, uint64_t *bytes) {
  API_BEGIN
  uint64_t b = c->sh->relin.d ? c->sh->relin.bytes : 0;
  for (auto &kv : c->sh->galois) b += kv.second.bytes;
  *bytes = b;
  API_END
}

int evah_ctx_key_bytes_detail(evah_ctx *c, uint64_t out[3]) {
  API_BEGIN
  out[0] = out[1] = out[2] = 0;
  auto count = [&](const KeyDev &kd) {
    if (!kd.d) return;
    out[0] += kd.bytes;
    if (kd.d_split) out[1] += kd.bytes;
    if (kd.d_perm) out[2] += sizeof(u64) * keyp_block_words(kd.n_digits) * (c->N / 256) * kd.rows;
  };
  count(c->sh->relin);
  for (auto &kv : c->sh->galois) count(kv.second);
  API_END
}

int evah_ct_write(evah_ctx *c, evah_ct *ct, const uint64_t *data) {
  API_BEGIN
  use(c);
  if (c->capturing) throw std::logic_error("evah_ct_write cannot be captured into a graph");
  if (ct->ps != (size_t)ct->limbs * c->N) throw std::invalid_argument("cannot write into a mod-switched view");
  acquire(c, ct->buf);
  HIPCHK(hipMemcpyAsync(ct->d, data, sizeof(u64) * (size_t)ct->batch * ct->size * ct->limbs * c->N, hipMemcpyHostToDevice,
                        c->stream));
  HIPCHK(hipStreamSynchronize(c->stream)); // pageable source: the caller may reuse it after return
  count_h2d(c, sizeof(u64) * (size_t)ct->batch * ct->size * ct->limbs * c->N);
  API_END
}

int evah_pt_write(evah_ctx *c, evah_pt *pt, const uint64_t *data) {
  API_BEGIN
  use(c);
  if (c->capturing) throw std::logic_error("evah_pt_write cannot be captured into a graph");
  acquire(c, pt->buf);
  pt->uniform = false;
  HIPCHK(hipMemcpyAsync(pt->d, data, sizeof(u64) * (size_t)pt->limbs * c->N, hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  count_h2d(c, sizeof(u64) * (size_t)pt->limbs * c->N, true);
  API_END
}

int evah_capture_begin(evah_ctx *q0, evah_ctx **others, uint32_t n_others) {
  API_BEGIN
  use(q0);
  HIPCHK(hipStreamSynchronize(q0->stream));
  for (uint32_t i = 0; i < n_others; i++) HIPCHK(hipStreamSynchronize(others[i]->stream));
  HIPCHK(hipStreamBeginCapture(q0->stream, hipStreamCaptureModeRelaxed));
  q0->capturing = true;
  for (uint32_t i = 0; i < n_others; i++) { // fork: every queue joins the capture
    stream_wait(others[i], q0);
    others[i]->capturing = true;
  }
  API_END
}

int evah_capture_end(evah_ctx *q0, evah_ctx **others, uint32_t n_others, evah_graph **out) {
  API_BEGIN
  use(q0);
  for (uint32_t i = 0; i < n_others; i++) { // join
    stream_wait(q0, others[i]);
    others[i]->capturing = false;
  }
  q0->capturing = false;
  auto *g = new evah_graph;
  hipError_t e = hipStreamEndCapture(q0->stream, &g->graph);
  if (e != hipSuccess) {
    delete g;
    throw std::runtime_error(std::string("hipStreamEndCapture failed: ") + hipGetErrorString(e));
  }
  for (uint32_t i = 0; i <= n_others; i++) { // events of this capture may be recycled now
    evah_ctx *q = i ? others[i - 1] : q0;
    q->sync_events.insert(q->sync_events.end(), q->capture_events.begin(), q->capture_events.end());
    q->capture_events.clear();
  }
  e = hipGraphInstantiate(&g->exec, g->graph, nullptr, nullptr, 0);
  if (e != hipSuccess) {
    (void)hipGraphDestroy(g->graph);
    delete g;
    throw std::runtime_error(std::string("hipGraphInstantiate failed: ") + hipGetErrorString(e));
  }
  for (uint32_t i = 0; i <= n_others; i++) { // the walk's temporaries stay the graph's (see evah_graph)
    evah_ctx *q = i ? others[i - 1] : q0;
    for (auto &kv : q->pool.free_)
      for (void *p : kv.second) g->reserved.push_back({q, p, kv.first});
    q->pool.free_.clear();
    q->pool.cached = 0;
  }
  *out = g;
  API_END
}

int evah_graph_launch(evah_ctx *q0, evah_graph *g) {
  API_BEGIN
  use(q0);
  HIPCHK(hipGraphLaunch(g->exec, q0->stream));
  API_END
}

void evah_graph_free(evah_graph *g) {
  if (!g) return;
  if (g->exec) (void)hipGraphExecDestroy(g->exec);
  if (g->graph) (void)hipGraphDestroy(g->graph);
  for (auto &r : g->reserved) { // back to the pool it came from, or to the device when that queue is gone
    if (ctx_alive(r.ctx)) {
      r.ctx->pool.free_[r.bytes].push_back(r.p);
      r.ctx->pool.cached += r.bytes;
    } else {
      (void)hipFree(r.p);
    }
  }
  delete g;
}

int evah_ct_info(const evah_ct *ct, uint32_t *size, uint32_t *limbs, double *scale) {
  API_BEGIN
  if (size) *size = ct->size;
  if (limbs) *limbs = ct->limbs;
  if (scale) *scale = ct->scale;
  API_END
}

int evah_ct_download(evah_ctx *c, const evah_ct *ct, uint64_t *out) {
  API_BEGIN
  use(c);
  if (c->capturing) throw std::logic_error("this call synchronises with the host and cannot be captured into a graph");
  acquire(c, ct->buf);
  const size_t row = sizeof(u64) * (size_t)ct->limbs * c->N;
  // a batched handle downloads as [batch][size][limbs][N]; a dense handle (not a mod-switched view) is one linear copy,
  // a view one linear copy per polynomial.  NOT hipMemcpy2DAsync: into pageable memory it is several times slower and —
  // r6 fuzz soak, 32 processes sharing the GPU — hipStreamSynchronize returned before the LAST row of such a copy was in the
  // caller's memory (4 of ~50 000 downloads of a view: the words of polynomial 1 arrived after the call had returned; the
  // linear copies, tens of millions of them in the same runs, never did).
  if (ct->ps == (size_t)ct->limbs * c->N) {
    HIPCHK(hipMemcpyAsync(out, ct->d, row * ct->size * ct->batch, hipMemcpyDeviceToHost, c->stream));
  } else {
    for (size_t p = 0; p < (size_t)ct->size * ct->batch; p++)
      HIPCHK(hipMemcpyAsync(reinterpret_cast<char *>(out) + p * row, ct->d + p * ct->ps, row, hipMemcpyDeviceToHost, c->stream));
  }
  HIPCHK(hipStreamSynchronize(c->stream));
  count_d2h(c, row * ct->size * ct->batch);
  API_END
}

void evah_ct_free(evah_ctx *c, evah_ct *ct) {
  if (!ct) return;
  buf_unref(c, ct->buf);
  delete ct;
}

int evah_pt_upload(evah_ctx *c, uint32_t limbs, double scale, const uint64_t *data, evah_pt **out) {
  API_BEGIN
  use(c);
  if (c->capturing) throw std::logic_error("this call synchronises with the host and cannot be captured into a graph");
  if (limbs < 1 || limbs > c->k - 1) throw std::invalid_argument("invalid limb count for this context");
  evah_pt *t = pt_new(c, limbs, scale);
  HIPCHK(hipMemcpyAsync(t->d, data, sizeof(u64) * (size_t)limbs * c->N, hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  count_h2d(c, sizeof(u64) * (size_t)limbs * c->N, true);
  t->buf->ready_everywhere = true;
  *out = t;
  API_END
}

int evah_pt_info(const evah_pt *pt, uint32_t *limbs, double *scale) {
  API_BEGIN
  if (limbs) *limbs = pt->limbs;
  if (scale) *scale = pt->scale;
  API_END
}

int evah_pt_download(evah_ctx *c, const evah_pt *pt, uint64_t *out) {
  API_BEGIN
  use(c);
  if (c->capturing) throw std::logic_error("this call synchronises with the host and cannot be captured into a graph");
  acquire(c, pt->buf);
  HIPCHK(hipMemcpyAsync(out, pt->d, sizeof(u64) * (size_t)pt->limbs * c->N, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  count_d2h(c, sizeof(u64) * (size_t)pt->limbs * c->N, true);
  API_END
}

void evah_pt_free(evah_ctx *c, evah_pt *pt) {
  if (!pt) return;
  buf_unref(c, pt->buf);
  delete pt;
}

int evah_profile_enable(evah_ctx *c, int on) {
  API_BEGIN
  use(c);
  c->prof_on = on != 0; // no host wait here: the records are resolved by evah_profile_get/_reset
  API_END
}
int evah_profile_reset(evah_ctx *c) {
  API_BEGIN
  use(c);
  prof_drain(c);
  for (int i = 0; i < KC_COUNT; i++) { c->prof_ms[i] = 0; c->prof_n[i] = 0; }
  API_END
}
int evah_profile_classes(void) { return KC_COUNT; }
const char *evah_profile_class_name(int cls) { return (cls >= 0 && cls < KC_COUNT) ? kclass_names[cls] : ""; }
int evah_profile_get(evah_ctx *c, int cls, uint64_t *launches, double *total_ms) {
  API_BEGIN
  use(c);
  if (cls < 0 || cls >= KC_COUNT) throw std::invalid_argument("kernel class out of range");
  prof_drain(c);
  *launches = c->prof_n[cls];
  *total_ms = c->prof_ms[cls];
  API_END
}

int evah_timer_start(evah_ctx *c) {
  API_BEGIN
  use(c);
  HIPCHK(hipEventRecord(c->ev0, c->stream));
  API_END
}
int evah_timer_stop(evah_ctx *c, float *ms) {
  API_BEGIN
  use(c);
  HIPCHK(hipEventRecord(c->ev1, c->stream));
  HIPCHK(hipEventSynchronize(c->ev1));
  HIPCHK(hipEventElapsedTime(ms, c->ev0, c->ev1));
  API_END
}

} // extern "C"
