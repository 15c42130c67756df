// internal.hip.h — state shared by the translation units of libeva_hip.so (not part of the ABI):
// device pool, buffers with cross-queue ordering, handle and context structs, profiling scopes.
//   runtime.hip    contexts, forks (issue queues), keys, pinned host memory, uploads / downloads,
//                  graph capture, profiling hooks                       (no kernels)
//   elementwise.hip  elementwise kernels + entry points (add .. multiply_plain, weighted sums), the device
//                    CKKS encoder, plaintext uploads, mod_switch
//   keyswitch.hip    the key-switch core (digit decomposition, fused second pass + key inner product,
//                    mod-down) and relinearize / rescale with their fused and batched forms
//   rotate.hip       Galois permutation, rotations, rotation sets (machinery: rotation_sets.hip.h, rot_fallback.hip.h)
//   windows.hip      evah_rotate_weighted_sums: convolution windows over the same machinery
//   shard.hip        limb-sharded phases (evah_shard_*), exchange buffers
//   client.hip       encrypt, decrypt + decode
//   launch.hip.h     launch plumbing of the transform kernels shared by the units above
//   scheduler.hip  evah_execute: the whole-DAG submit (level scheduler, peepholes)   (no kernels)
#pragma once
#include "../../include/eva_hip.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <tuple>
#include <unordered_map>
#include <vector>

#include "devmath.hip.h"
#include "hostmath.h"

namespace evah {

extern thread_local std::string g_err; // last error of the calling thread (runtime.hip)

// Contexts that exist: a buffer remembers the foreign queues that read it, and such a queue may
// have been destroyed (after a sync) before the buffer is released.  (runtime.hip)
void ctx_register(const void *c);
void ctx_unregister(const void *c);
bool ctx_alive(const void *c);

#define HIPCHK(x)                                                                                \
  do {                                                                                           \
    hipError_t e_ = (x);                                                                         \
    if (e_ != hipSuccess)                                                                        \
      throw std::runtime_error(std::string(#x) + " failed: " + hipGetErrorString(e_));          \
  } while (0)

#ifndef EVAH_KS_BATCH_MAX_DEFINED
#define EVAH_KS_BATCH_MAX_DEFINED
constexpr int KS_BATCH_MAX = 64; // instances per batched launch set / batched handle
#endif


struct Pool {
  std::map<size_t, std::vector<void *>> free_;
  size_t in_use = 0, cached = 0;
  void *alloc(size_t bytes) {
    bytes = (bytes + 255) & ~(size_t)255;
    auto it = free_.find(bytes);
    void *p = nullptr;
    if (it != free_.end() && !it->second.empty()) {
      p = it->second.back();
      it->second.pop_back();
      cached -= bytes;
    } else {
      hipError_t e = hipMalloc(&p, bytes);
      if (e != hipSuccess) {
        release_cached();
        HIPCHK(hipMalloc(&p, bytes));
      }
    }
    in_use += bytes;
    return p;
  }
  void free(void *p, size_t bytes) {
    bytes = (bytes + 255) & ~(size_t)255;
    free_[bytes].push_back(p);
    in_use -= bytes;
    cached += bytes;
  }
  void release_cached() {
    for (auto &kv : free_)
      for (void *p : kv.second) (void)hipFree(p);
    free_.clear();
    cached = 0;
  }
};

struct Buffer {
  u64 *d;
  size_t bytes;
  int refs;
  Pool *pool;                       // owning queue's pool (handles may be freed through any fork)
  evah_ctx *owner;                  // queue whose stream produced / will recycle this buffer
  bool ready_everywhere;            // contents were synchronised with the host (uploads)
  std::vector<evah_ctx *> synced;   // foreign queues that already wait for the producer
  std::vector<evah_ctx *> readers;  // foreign queues that have enqueued reads
};

// Kernel classes for the optional per-launch HIP-event profile (bench.py's roofline leg).
enum KClass {
  KC_EW = 0,        // elementwise add/sub/negate/mul/square/mul_plain/perm/fill
  KC_INTT_A,        // inverse pass 1 (contig)
  KC_INTT_B,        // inverse pass 2 (strided)
  KC_KSDIGIT_A,     // key-switch digit conversion, forward pass 1 (strided, fused base conversion)
  KC_KSDIGIT_B,     // key-switch digit conversion, forward pass 2 (contig)
  KC_KSMAC,         // key-switch inner product
  KC_MODDOWN_A,     // rescale / mod-down forward pass 1 (fused reduce - half)
  KC_MODDOWN_B,     // rescale / mod-down forward pass 2 (fused combine)
  KC_NTT_A,         // plain forward pass 1
  KC_NTT_B,         // plain forward pass 2
  KC_COUNT
};
static const char *const kclass_names[KC_COUNT] = {
    "elementwise", "intt_pass1", "intt_pass2", "ksdigit_pass1", "ksdigit_pass2", "ks_mac",
    "moddown_pass1", "moddown_pass2", "ntt_pass1", "ntt_pass2"};

struct ProfRec {
  hipEvent_t e0, e1;
  int cls;
};

// words of one block of a permuted key copy (KeyDev::d_perm): every digit and both polynomials of one prime row and 256
// coefficients, plus one 2 KiB pad — with 8 digits a block would be exactly 32 KiB and the blocks of the eight workgroups
// the dispatcher starts together would begin on the same memory channel (Harris' 3 x 2 tiles: 78 -> 94 us without the pad)
__host__ __device__ inline size_t keyp_block_words(uint32_t nd) { return ((size_t)2 * nd + 1) * 256; }
struct KeyDev {
  u64 *d = nullptr;
  uint32_t n_digits = 0;
  size_t bytes = 0;
  // prime rows stored per (digit, polynomial): k for a whole key [digits][2][k][N]; a limb shard that
  // set its map before the upload keeps only its own data limbs and the special prime
  // ([digits][2][rows][N], row y = prime shard + y * shards, last row = the special prime)
  uint32_t rows = 0;
  // the same key words cut at bit 30 — (k mod 2^30) | (k >> 30) << 32 — for the radix-2^30 inner product of
  // ks_inner_kernel<MAC3>; built at upload for whole keys of contexts whose primes all have the top-bit shape
  u64 *d_split = nullptr;
  // Galois keys used by hoisted rotation sets: the same words with every row read through the inverse of the element's
  // NTT-domain permutation — d_perm[..][m] = d[..][pi^-1(m)] — so that the hoisted inner product is elementwise in the
  // source's own index space (rotation_sets.hip.h, k_hoist_mac); built at the first hoisted use of the key
  u64 *d_perm = nullptr; // r6 layout: [prime row][N / 256][keyp_block_words(n_digits)], rotation_sets.hip.h k_key_perm
};

} // namespace evah

using namespace evah;

struct evah_ct {
  Buffer *buf;
  u64 *d;
  uint32_t size, limbs;
  size_t ps; // poly stride in elements
  double scale;
  // `batch` independent ciphertexts of identical shape in one handle: instance b starts at
  // d + b * size * ps, i.e. the handle is batch * size polynomials at a uniform stride.  Every
  // evaluator entry point applies to all instances in one launch set (plaintext operands and keys
  // are shared); this is how a batch of independent DAG instances is run (BASELINE config 4).
  uint32_t batch = 1;
};
struct evah_pt {
  Buffer *buf;
  u64 *d;
  uint32_t limbs;
  double scale;
  // every word of a limb's row is the same residue (evah_pt_uniform: the encoding of a uniform constant,
  // Program::makeUniformConstant).  Multiplying by such a plaintext is multiplying by a scalar of Z_q, which commutes with
  // the NTT: the convolution windows use it (ntt_window_lin.hip.h).  Cleared by anything that rewrites the words.
  bool uniform = false;
};
struct evah_graph {
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
  // The temporaries of the captured walk were returned to their queues' pools, but every replay writes
  // them again: the blocks are taken out of the pools for the graph's lifetime, so the queues can go on
  // allocating (copies of the outputs, on the replay's own stream) without handing that memory out.
  struct Reserved { evah_ctx *ctx; void *p; size_t bytes; };
  std::vector<Reserved> reserved;
};

// Device state shared by a context and its forks: tables, keys, permutation tables.
struct SharedDev {
  int device = 0;
  void *d_tables = nullptr;
  KeyDev relin;
  KeyDev pk, sk; // client side (client.hip): public key [2][k][N], secret key in NTT form [k][N]
  double2 *dec_roots = nullptr; // CKKS decoder: zeta^br(j) (forward special FFT, heap order)
  std::map<uint32_t, KeyDev> galois;
  std::map<uint32_t, uint32_t *> perms;
  std::map<uint32_t, uint32_t *> perms_inv; // the inverse tables (hoisted rotation sets)
  // hoisted rotations (rotation_sets.hip.h): NTT of the sign pattern of a Galois element under every prime
  // ([k][N]) and, per (element, level), the constant it contributes to the key inner product ([2][l+1][N])
  std::map<uint32_t, u64 *> hoist_sign;
  std::map<std::pair<uint32_t, uint32_t>, u64 *> hoist_corr;
  double2 *enc_roots = nullptr;     // CKKS encoder: inverse-FFT roots in the order the stages consume them
  double enc_last_root[2] = {0, 0}; // the single root of the last stage (scaled by fix on the host per call)
  uint32_t *enc_slot_map = nullptr; // slot i (and its conjugate, at slots + i) -> FFT input index
  uint32_t key_rows = 0, key_shard = 0; // evaluation keys: 0 none yet, 1 whole, 2 the prime rows of limb shard `key_shard`
  uint64_t xfer[6] = {0, 0, 0, 0, 0, 0}; // evah_ctx_transfer_stats: ct up / down, pt up / down, bytes up / down
  ~SharedDev() {
    (void)hipSetDevice(device);
    if (enc_roots) (void)hipFree(enc_roots);
    if (enc_slot_map) (void)hipFree(enc_slot_map);
    if (relin.d) (void)hipFree(relin.d);
    if (relin.d_split) (void)hipFree(relin.d_split);
    if (pk.d) (void)hipFree(pk.d);
    if (sk.d) { (void)hipMemset(sk.d, 0, sk.bytes); (void)hipFree(sk.d); } // key material does not stay behind in freed HBM
    if (dec_roots) (void)hipFree(dec_roots);
    for (auto &kv : galois) {
      (void)hipFree(kv.second.d);
      if (kv.second.d_split) (void)hipFree(kv.second.d_split);
      if (kv.second.d_perm) (void)hipFree(kv.second.d_perm);
    }
    for (auto &kv : perms) (void)hipFree(kv.second);
    for (auto &kv : perms_inv) (void)hipFree(kv.second);
    for (auto &kv : hoist_sign) (void)hipFree(kv.second);
    for (auto &kv : hoist_corr) (void)hipFree(kv.second);
    if (d_tables) (void)hipFree(d_tables);
  }
};

// Every knob that decides HOW a call is launched (never what it computes: all settings give the same
// residues).  Defaults are the measured optima on MI355X (profiles/r0*_tuning_notes.md); each can be
// overridden by the environment variable named beside it, read ONCE in evah_ctx_create — forks copy
// their parent's values.
struct Tunables {
  // EVAH_FUSE_MAC (1): key switch — fuse the key inner product into the digit transforms' second pass
  // (ks_inner_kernel); 0 = separate digit transforms + k_ks_mac, the unfused reference path
  bool fuse_mac = true;
  // EVAH_FUSE_MUL (1): evah_execute runs Mul -> Relinearize -> Rescale chains as ONE
  // evah_multiply_relinearize_rescale_many (the size-3 product never reaches HBM).  r02 measured it 1.6 %
  // slower at N = 2^16 and kept it for N <= 8192; with the r03 128-bit reduction its on-the-fly products
  // are cheap enough that it wins at every size (op-triple 13 383 -> 13 635 /s)
  bool fuse_mul = true;
  // EVAH_FUSE_MUL2 (1): evah_execute runs Mul -> Rescale -> Relinearize chains (lazy relinearization's order) as ONE
  // evah_multiply_rescale_relinearize_many (r6); 0 = the three calls
  bool fuse_mul2 = true;
  // EVAH_EW_FUSE (1): evah_execute defers elementwise ops (negate / add / sub / multiply, ciphertexts and plaintexts) on
  // values nobody needs stored and runs each connected run of them as ONE evah_elementwise_program; 0 = one launch per op
  bool ew_fuse = true;
  // EVAH_EW_UNIFORM (1): an elementwise program multiplies by a uniform plaintext (evah_pt_uniform) as by a scalar per limb —
  // Shoup products, the plaintext's polynomial never loaded (EW_MULU); 0 = the general product
  bool ew_uniform = true;
  // EVAH_FUSE_SMALL (2048): launches of at most this many 2048-coefficient tiles are latency-bound — an
  // inverse transform followed by forward transforms of the result runs its two strided passes as one
  // launch (ntt_inv_fwd_kernel); 0 disables.  r03 sweep at the BASELINE sizes: Harris L=8 1.18 -> 1.15 ms,
  // 256 Sobel DAGs (l=5) 9.6 k -> 10.3 k/s against the r02 value 8192
  uint32_t fuse_small_blocks = 2048;
  // EVAH_SMALL_LR (2) / EVAH_SMALL_LR_BLOCKS (1024): log2 coefficients per thread of the transform passes
  // in launches of at most that many tiles (2 = 4 per thread: twice the workgroups, half a wave's critical path)
  int small_lr = 2;
  uint32_t small_lr_blocks = 1024;
  // EVAH_HOIST (1) / EVAH_HOIST_MIN_TILES (2048): several rotations of one ciphertext decompose it once
  // (hoisting) when the set is at least this many tiles of digit transforms
  bool hoist = true;
  uint32_t hoist_min_tiles = 2048;
  // EVAH_HOIST_DEBUG (0): print the zero-coefficient count of every hoisted set (synchronises)
  bool hoist_debug = false;
  // EVAH_PEER_SELF_CHECK (0): tests only — evah_ctx_enable_peer asks the runtime about the pair even when both contexts
  // are on one device (where it answers "no"), so the refusal path can run on a 1-GPU box
  bool peer_self_check = false;
  // EVAH_HOIST_TABLE_FAIL (0): tests only — behave as if the device had no room for any NEW hoisting table (permuted
  // key copies, per-(element, level) constants): the affected sets must run unhoisted with the same results
  bool hoist_table_fail = false;
  // EVAH_FUSE_SPECIAL_INV (1): latency-bound key switches run the special row's first inverse pass
  // inside the key-switch kernel (ks_inner_kernel INVSP)
  bool fuse_special_inv = true;
  // EVAH_KS_GROUPS (1): output-limb slices per key switch; EVAH_KS_THREADS (64): threads per workgroup of the
  // fused key-switch kernel — one wave = one 2^P-point sub-transform measured best (barriers are intra-wave)
  int ks_groups = 1;
  int ks_threads = 64;
  // EVAH_FOLD_PA (1): relinearize (+ rescale) and the fused multiply add P * (the polynomials the key-switch result
  // is added to) to the key inner products inside ks_inner_kernel (KS_FOLDMUL / KS_FOLDADD), so the mod-down's
  // combine pass — which waits for bytes — reads the products only; 0 = the r03 forms (operands read in the epilogue)
  bool fold_pa = true;
  // EVAH_WIN_FUSE (1): evah_rotate_weighted_sums and the scheduler's convolution windows run the mod-down of the
  // window's rotations fused with the weighted sums (moddown_sum_kernel); 0 = rotation set, then evah_weighted_sum
  bool win_fuse = true;
  // EVAH_WIN_LINEAR (1): a fused window all of whose rotated terms have uniform plaintexts (or 1) as weights runs the
  // mod-down's forward transforms once per sum instead of once per rotation (ntt_window_lin.hip.h); 0 = moddown_sum_kernel
  bool win_linear = true;
  // EVAH_FB_PERSIST (1): the exact fallback of a hoisted rotation set is one persistent launch per chunk (k_rot_fallback,
  // rot_fallback.hip.h) that leaves at once unless the zero counter overflowed; 0 = the ordinary unhoisted launches, each
  // guarded (about eight launches that return at once per chunk)
  // EVAH_FB_GRID (0 = one workgroup per CU): workgroups of that launch; any value >= 1 gives the same bits (its phases
  // are ordered by tickets, not by residency) — the tests run it with 1, 3 and 2000
  bool fb_persist = true;
  uint32_t fb_grid = 0;
  // EVAH_MAC3 (1): key inner products accumulate in radix 2^30 (ks_inner_kernel<MAC3>) when every prime of the context
  // has the top-bit shape and the level has at most 15 limbs; the keys are then kept in the split layout as well
  bool mac3 = true;
  // EVAH_LOOP_N (8) / EVAH_LOOP_MIN (2): contiguous transform passes whose launch holds at least loop_min jobs modulo
  // one prime (both polynomials of a ciphertext, the instances of a batched call, the digits of a hoisted set) run as
  // ntt_loop_kernel: the twiddle heaps of a tile staged in LDS once per workgroup and reused by up to loop_n jobs;
  // loop_n = 0: ntt_pass_kernel everywhere (per-thread global twiddle loads)
  // EVAH_LOOP_MIN_WGS (2048): below this many workgroups (at two jobs per walk) the launch keeps ntt_pass_kernel
  // EVAH_LOOP_TARGET_WGS: the walk is shortened (down to 2 jobs) until the launch has this many one-wave workgroups
  uint32_t loop_n = 8, loop_min = 2, loop_min_wgs = 2048, loop_target_wgs = 8192;
  // EVAH_HOIST_MAP (1): k_hoist_mac's workgroups in the order (xcd | tile | x_hi | I) — the tiles of one coefficient range
  // run back to back on one XCD, so operand rows shared between tiles are read from HBM once; 0 = the r4 grid (x, I, tile)
  // EVAH_HOIST_V (1): coefficients per thread of k_hoist_mac for tile shapes with <= 4 accumulator pairs (2 = 16-byte accesses)
  bool hoist_map = true;
  uint32_t hoist_v = 1;
  // EVAH_SIDE_STREAM (0): evah_multiply_rescale_relinearize runs the rescale of d0 / d1 on the queue's side stream beside
  // the key switch of d2.  Measured (r06_tuning_notes.md): eager walks gain (config 5 1.77 -> 1.70 ms), replayed hipGraphs
  // do not — a fork / join inside a graph costs what the overlap returns — so the default is one stream
  bool side_stream = false;
  // EVAH_CHAIN_STEP (1): evah_multiply_rescale_relinearize(_many) as the six-launch chain step of ntt_chain.hip.h (the rescaled d2
  // formed in coefficient form, the rescale of d0 / d1 sharing the mod-down's forward transform); 0 = the r6 launch set
  // (rescale of the three polynomials, then the key switch: nine or ten launches)
  // EVAH_CHAIN_FUSE_BLOCKS (256): digit launches of at most this many 2048-coefficient tiles recompute t_J per output limb
  // inside the digit conversion's launch (one launch less); above it t is stored once and OpKsDigit reads it
  // EVAH_CHAIN_BATCHED (1): evah_execute defers Mul -> Rescale -> Relinearize chains on batched handles as well (the instances of
  // a handle become entries of the fused call's launch set); 0 = the three batched calls
  bool chain_step = true, chain_batched = true;
  uint32_t chain_fuse_blocks = 256;
  // EVAH_LDS_EXTRA (0): bytes of dynamic LDS added to every ntt_pass_kernel launch — an occupancy probe for the
  // tuning notes (fewer workgroups per CU), never set in production
  uint32_t lds_extra = 0;

  static Tunables from_env(uint32_t N) {
    Tunables t;
    auto flag = [](const char *name, bool &v) { if (const char *e = std::getenv(name)) v = std::atoi(e) != 0; };
    auto count = [](const char *name, uint32_t &v) { if (const char *e = std::getenv(name)) v = (uint32_t)std::max(0, std::atoi(e)); };
    (void)N;
    flag("EVAH_FUSE_MAC", t.fuse_mac);
    flag("EVAH_FUSE_MUL", t.fuse_mul);
    flag("EVAH_FUSE_MUL2", t.fuse_mul2);
    flag("EVAH_EW_FUSE", t.ew_fuse);
    flag("EVAH_EW_UNIFORM", t.ew_uniform);
    count("EVAH_FUSE_SMALL", t.fuse_small_blocks);
    if (const char *e = std::getenv("EVAH_SMALL_LR")) t.small_lr = std::atoi(e) == 3 ? 3 : 2;
    count("EVAH_SMALL_LR_BLOCKS", t.small_lr_blocks);
    flag("EVAH_HOIST", t.hoist);
    count("EVAH_HOIST_MIN_TILES", t.hoist_min_tiles);
    flag("EVAH_HOIST_DEBUG", t.hoist_debug);
    flag("EVAH_HOIST_TABLE_FAIL", t.hoist_table_fail);
    flag("EVAH_PEER_SELF_CHECK", t.peer_self_check);
    flag("EVAH_FUSE_SPECIAL_INV", t.fuse_special_inv);
    flag("EVAH_FOLD_PA", t.fold_pa);
    flag("EVAH_WIN_FUSE", t.win_fuse);
    flag("EVAH_WIN_LINEAR", t.win_linear);
    flag("EVAH_FB_PERSIST", t.fb_persist);
    count("EVAH_FB_GRID", t.fb_grid);
    flag("EVAH_MAC3", t.mac3);
    count("EVAH_LOOP_N", t.loop_n);
    count("EVAH_LOOP_MIN", t.loop_min);
    count("EVAH_LOOP_MIN_WGS", t.loop_min_wgs);
    count("EVAH_LOOP_TARGET_WGS", t.loop_target_wgs);
    count("EVAH_LDS_EXTRA", t.lds_extra);
    flag("EVAH_HOIST_MAP", t.hoist_map);
    flag("EVAH_SIDE_STREAM", t.side_stream);
    flag("EVAH_CHAIN_STEP", t.chain_step);
    flag("EVAH_CHAIN_BATCHED", t.chain_batched);
    count("EVAH_CHAIN_FUSE_BLOCKS", t.chain_fuse_blocks);
    count("EVAH_HOIST_V", t.hoist_v);
    if (const char *e = std::getenv("EVAH_KS_GROUPS")) t.ks_groups = std::max(1, std::atoi(e));
    if (const char *e = std::getenv("EVAH_KS_THREADS")) {
      const int n = std::atoi(e);
      if (n == 64 || n == 128 || n == 256) t.ks_threads = n;
    }
    return t;
  }
};

struct evah_ctx {
  std::shared_ptr<SharedDev> sh;
  int device = 0;
  uint32_t N = 0, logN = 0, k = 0;
  std::vector<u64> primes;
  std::vector<int> total_bits; // total_bits[l] = bit length of prod primes[0..l)
  DevCtx dev{};
  hipStream_t own = nullptr, stream = nullptr;
  hipStream_t side = nullptr; // second stream of this queue, created on first use (SideStream: independent halves of one fused call)
  Pool pool;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  bool capturing = false;              // inside evah_capture_begin/end: no syncs, no profiling events
  std::vector<hipEvent_t> capture_events; // events consumed by the capture in progress
  std::vector<hipEvent_t> sync_events; // recycled events for cross-queue ordering
  Tunables tun; // launch-shape decisions, read from the environment once when the context is created
  bool all_tb = false; // every prime is 2^b - c with b > 32, c < 2^32 (DevPrime::tb_c != 0) or below 2^54: ks_inner_kernel<MAC3> applies
  // per-launch profile
  bool prof_on = false;
  std::vector<ProfRec> prof_recs;
  std::vector<hipEvent_t> prof_free;
  double prof_ms[KC_COUNT] = {0};
  uint64_t prof_n[KC_COUNT] = {0};
};

namespace evah {

inline void use(evah_ctx *c) { HIPCHK(hipSetDevice(c->device)); }
inline void count_h2d(evah_ctx *c, size_t bytes, bool plain = false) { c->sh->xfer[plain ? 2 : 0]++; c->sh->xfer[4] += bytes; }
inline void count_d2h(evah_ctx *c, size_t bytes, bool plain = false) { c->sh->xfer[plain ? 3 : 1]++; c->sh->xfer[5] += bytes; }

inline hipEvent_t prof_event(evah_ctx *c) {
  if (!c->prof_free.empty()) {
    hipEvent_t e = c->prof_free.back();
    c->prof_free.pop_back();
    return e;
  }
  hipEvent_t e;
  HIPCHK(hipEventCreate(&e));
  return e;
}
inline void prof_drain(evah_ctx *c) {
  for (auto &r : c->prof_recs) {
    float ms = 0;
    HIPCHK(hipEventSynchronize(r.e1));
    HIPCHK(hipEventElapsedTime(&ms, r.e0, r.e1));
    c->prof_ms[r.cls] += ms;
    c->prof_n[r.cls]++;
    c->prof_free.push_back(r.e0);
    c->prof_free.push_back(r.e1);
  }
  c->prof_recs.clear();
}
struct ProfScope { // brackets one kernel launch with HIP events on the launch stream
  evah_ctx *c;
  int cls;
  hipEvent_t e0 = nullptr;
  ProfScope(evah_ctx *c_, int cls_) : c(c_), cls(cls_) {
    if (c->prof_on && !c->capturing) {
      if (c->prof_recs.size() >= 8192) prof_drain(c);
      e0 = prof_event(c);
      HIPCHK(hipEventRecord(e0, c->stream));
    }
  }
  ~ProfScope() {
    if (e0) {
      hipEvent_t e1 = prof_event(c);
      (void)hipEventRecord(e1, c->stream);
      c->prof_recs.push_back({e0, e1, cls});
    }
  }
};

inline Buffer *buf_new(evah_ctx *c, size_t elems) {
  Buffer *b = new Buffer;
  b->bytes = elems * sizeof(u64);
  b->d = (u64 *)c->pool.alloc(b->bytes);
  b->refs = 1;
  b->pool = &c->pool;
  b->owner = c;
  b->ready_everywhere = false;
  return b;
}
// Host -> device copies of set-up data (keys, tables) that are COMPLETE when they return, whatever queue reads the words next.
// hipMemcpy from pageable memory returns once the words are staged — the DMA into HBM may still be in flight — and only the
// NULL stream is ordered behind it; this library's queues are non-blocking streams, so a kernel launched right after such
// a copy (the split copy of a key, the first rotation through a new permutation table) could read the destination before
// the words had landed.  Found by the r6 fuzz soak: ~3 of 10 000 parameter sets with 32 processes sharing the GPU, never
// with the GPU to itself.  The copy goes on the queue's own stream and the queue is drained before returning.
inline void h2d_now(evah_ctx *c, void *dst, const void *src, size_t bytes) {
  HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
}
inline hipEvent_t sync_event(evah_ctx *c) {
  if (!c->capturing && !c->sync_events.empty()) {
    hipEvent_t e = c->sync_events.back();
    c->sync_events.pop_back();
    return e;
  }
  hipEvent_t e;
  HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  return e;
}
// make `waiter`'s stream wait for everything enqueued so far on `signaller`'s stream
inline void stream_wait(evah_ctx *waiter, evah_ctx *signaller) {
  if (waiter == signaller || waiter->stream == signaller->stream) return;
  if (waiter->sh.get() != signaller->sh.get()) {
    // Queues of two device states — two GPUs (sub-DAG members, limb shards), or two states on one GPU (how a 1-GPU box runs
    // this branch: limb shards, "virtual device" members).  An event is recorded on a stream of the device it was created
    // on, so it comes from the SIGNALLER's pool under the signaller's device; the wait is issued under the waiter's
    // (hipStreamWaitEvent takes events of other devices).  r5: until now the event came from the waiter — on a real
    // two-GPU box an event of device A recorded on a stream of device B, which no box had ever executed.
    int cur = 0;
    HIPCHK(hipGetDevice(&cur));
    HIPCHK(hipSetDevice(signaller->device));
    hipEvent_t e = sync_event(signaller);
    hipError_t rc = hipEventRecord(e, signaller->stream);
    (void)hipSetDevice(waiter->device);
    if (rc == hipSuccess) rc = hipStreamWaitEvent(waiter->stream, e, 0);
    (void)hipSetDevice(cur);
    if (waiter->capturing || signaller->capturing) signaller->capture_events.push_back(e);
    else signaller->sync_events.push_back(e);
    HIPCHK(rc);
    return;
  }
  hipEvent_t e = sync_event(waiter);
  HIPCHK(hipEventRecord(e, signaller->stream));
  HIPCHK(hipStreamWaitEvent(waiter->stream, e, 0));
  // eager mode: the wait captured this record, so the event can be re-recorded right away.
  // While capturing, an event is used for exactly one record/wait pair of the graph (re-recording
  // one inside a capture has produced cyclic graphs with the ROCm 7.2 runtime).
  if (waiter->capturing || signaller->capturing) waiter->capture_events.push_back(e);
  else waiter->sync_events.push_back(e);
}
// Two independent halves of ONE call on two streams of the same queue (r6; the rescale of a product's d0 / d1 beside the
// key switch of its d2): fork() makes the side stream wait for everything enqueued on the queue so far and redirects the
// queue's launches to it, back() returns to the queue's own stream, join() makes the queue wait for the side work.
// Temporaries come from the queue's pool as always (they are returned after join(), in stream order).  Inside a graph
// capture the side stream joins the capture through the fork's event and must be joined before the call returns.
struct SideStream {
  evah_ctx *c;
  hipStream_t main;
  bool forked = false, on_side = false;
  explicit SideStream(evah_ctx *c_) : c(c_), main(c_->stream) {}
  void link(hipStream_t from, hipStream_t to) {
    hipEvent_t e = sync_event(c);
    HIPCHK(hipEventRecord(e, from));
    HIPCHK(hipStreamWaitEvent(to, e, 0));
    if (c->capturing) c->capture_events.push_back(e);
    else c->sync_events.push_back(e);
  }
  void fork() {
    if (!c->side) HIPCHK(hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking));
    link(main, c->side);
    c->stream = c->side;
    forked = on_side = true;
  }
  void back() { c->stream = main; on_side = false; }
  void join() {
    back();
    if (forked) link(c->side, main);
    forked = false;
  }
  ~SideStream() { // (an exception between fork and join: the side work is still ordered before whatever follows)
    c->stream = main;
    if (forked) {
      hipEvent_t e = sync_event(c);
      if (hipEventRecord(e, c->side) == hipSuccess) (void)hipStreamWaitEvent(main, e, 0);
      (c->capturing ? c->capture_events : c->sync_events).push_back(e);
    }
  }
};
// Called before queue `c` enqueues a read of `b`: orders the read after the producer and
// remembers the reader so the buffer is not recycled under it.
inline void acquire(evah_ctx *c, Buffer *b) {
  if (!b || b->owner == c) return;
  if (!b->ready_everywhere && std::find(b->synced.begin(), b->synced.end(), c) == b->synced.end()) {
    stream_wait(c, b->owner);
    b->synced.push_back(c);
  }
  if (std::find(b->readers.begin(), b->readers.end(), c) == b->readers.end()) b->readers.push_back(c);
}
inline void buf_unref(evah_ctx *c, Buffer *b) {
  (void)c;
  if (b && --b->refs == 0) {
    if (!ctx_alive(b->owner)) {
      // the queue that produced the buffer is gone (it was synchronised when destroyed, and its pool
      // with it): nothing to order against and no pool to return to
      (void)hipFree(b->d);
      delete b;
      return;
    }
    for (evah_ctx *r : b->readers)
      if (ctx_alive(r)) stream_wait(b->owner, r); // recycle only after foreign reads (a destroyed queue was synchronised)
    b->pool->free(b->d, b->bytes);
    delete b;
  }
}
inline evah_ct *ct_new(evah_ctx *c, uint32_t size, uint32_t limbs, double scale, uint32_t batch = 1) {
  evah_ct *t = new evah_ct;
  t->batch = batch;
  t->buf = buf_new(c, (size_t)batch * size * limbs * c->N);
  t->d = t->buf->d;
  t->size = size;
  t->limbs = limbs;
  t->ps = (size_t)limbs * c->N;
  t->scale = scale;
  return t;
}
inline evah_pt *pt_new(evah_ctx *c, uint32_t limbs, double scale) {
  evah_pt *t = new evah_pt;
  t->buf = buf_new(c, (size_t)limbs * c->N);
  t->d = t->buf->d;
  t->limbs = limbs;
  t->scale = scale;
  return t;
}

// rotate.hip: would evah_rotate_many hoist n rotations of one l-limb ciphertext of B instances?
// (the scheduler groups sibling rotations for it only when it does)
bool hoist_wanted(const evah_ctx *c, uint32_t l, uint32_t n, uint32_t B);

struct Scratch { // pool-backed temporary, returned on scope exit (stream-ordered reuse)
  evah_ctx *c;
  u64 *d;
  size_t bytes;
  Scratch(evah_ctx *c_, size_t elems) : c(c_), bytes(elems * sizeof(u64)) {
    d = (u64 *)c->pool.alloc(bytes);
  }
  ~Scratch() { c->pool.free(d, bytes); }
};

inline dim3 ew_grid(evah_ctx *c, uint32_t limbs, uint32_t polys) {
  return dim3(c->N / 512, limbs, polys);
}

inline int bitlen_of_product(const std::vector<u64> &primes, uint32_t count) {
  std::vector<u64> w{1};
  for (uint32_t i = 0; i < count; i++) {
    u64 carry = 0;
    for (auto &x : w) {
      u128 t = (u128)x * primes[i] + carry;
      x = (u64)t;
      carry = (u64)(t >> 64);
    }
    if (carry) w.push_back(carry);
  }
  int bits = (int)(w.size() - 1) * 64;
  u64 top = w.back();
  while (top) { bits++; top >>= 1; }
  return bits;
}

inline void check_scale(evah_ctx *c, double scale, uint32_t limbs) {
  // SEAL is_scale_within_bounds: 0 < scale, log2(scale) < total coeff modulus bits at the level
  if (!(scale > 0) || (int)std::log2(scale) >= c->total_bits[limbs])
    throw std::invalid_argument("scale out of bounds");
}
inline bool same_scale(double a, double b) {
  // SEAL util::are_close<double>
  double scale_factor = std::max({std::fabs(a), std::fabs(b), 1.0});
  return std::fabs(a - b) < 2.220446049250313e-16 * scale_factor;
}

} // namespace evah

#define EW_LAUNCH(...) do { ProfScope ps_(c, KC_EW); hipLaunchKernelGGL(__VA_ARGS__); } while (0)
#define API_BEGIN try {
#define API_END                                                                                  \
  g_err.clear();                                                                                 \
  return 0;                                                                                      \
  }                                                                                              \
  catch (const std::exception &e) {                                                              \
    g_err = e.what();                                                                            \
    return 1;                                                                                    \
  }                                                                                              \
  catch (...) {                                                                                  \
    g_err = "unknown error";                                                                     \
    return 1;                                                                                    \
  }
