// multi_device.h — execute() of ONE program on several GPUs, chosen
// inside HipPublic::execute the way the reference chooses its parallel traversal inside
// SEALPublic::execute (/root/reference/eva/seal/seal.cpp:105-113, sized by set_num_threads,
// /root/reference/python/eva/wrapper.cpp:128-137):
//
//   sub-DAG split   independent sub-DAGs of the program (Harris: the three 3x3 convolutions,
//                   /root/reference/examples/image_processing.py:65-100) run on different devices, one
//                   evah_execute submit per piece, a peer copy (evah_ct_copy, xGMI) for every
//                   ciphertext that crosses a cut.  The GPU counterpart of the node-level parallelism
//                   of MulticoreProgramTraversal (multicore_program_traversal.h:55-78).
//   limb sharding   limb i of every value on shard i mod G; per key switch one all-gather of the
//                   coefficient-form digits and one broadcast, per rescale one broadcast
//                   (evah_shard_* phases, include/eva_hip.h).
//
// Devices are a list of device indices; a repeated index gives several contexts on one GPU (forks of
// one device state), which is how a single-GPU box validates both paths.  Results are the
// ciphertexts of the single-device run bit for bit: the partition only decides where work runs.
#pragma once
#include <algorithm>
#include <functional>
#include <numeric>

namespace evahost {

// One member per entry of `ids`; members that share a device share its tables and keys.
struct DeviceGroup {
  std::vector<int> ids;
  std::vector<std::shared_ptr<DeviceCtx>> roots; // per member: the device state it belongs to
  std::vector<std::shared_ptr<Fork>> forks;      // per member: its own queue when it is not a root itself (else null)
  std::vector<evah_ctx *> ctx;                   // per member: the queue to issue on
  size_t size() const { return ctx.size(); }
};

// A member id names a DEVICE STATE: ids below 256 are the HIP devices themselves; id = d + 256 v (v >= 1) is a further,
// separate state — own tables, own keys, own queues — on device d ("virtual device").  With them a 1-GPU box runs every
// branch that two GPUs take except the peer hardware itself: separate roots per member, key uploads per root, copies and
// ordering between the queues of different states (evah_ct_copy / evah_pt_copy, stream_wait across states).
inline int physical_device(int id) { return id & 0xff; }
// peer access for every pair of members on different devices; a refusal throws (evah_last_error names the pair)
inline void enable_peers(const DeviceGroup &g) {
  for (size_t a = 0; a < g.ctx.size(); a++)
    for (size_t b = a + 1; b < g.ctx.size(); b++)
      if (g.ids[a] != g.ids[b]) chk(evah_ctx_enable_peer(g.ctx[a], g.ctx[b]));
}

// member 0 is `first` (the context the public half already works on); every other distinct device
// gets its own DeviceCtx with the evaluation keys uploaded by `upload_keys`
inline DeviceGroup make_device_group(const std::vector<int> &ids, const std::shared_ptr<DeviceCtx> &first, int first_device,
                                     const HostContext &host, const std::function<void(evah_ctx *)> &upload_keys, bool fork_first) {
  DeviceGroup g;
  g.ids = ids;
  std::map<int, std::shared_ptr<DeviceCtx>> by_device;
  by_device[first_device] = first;
  for (size_t m = 0; m < ids.size(); m++) {
    auto it = by_device.find(ids[m]);
    const bool is_new = it == by_device.end();
    std::shared_ptr<DeviceCtx> root;
    if (is_new) {
      root = std::make_shared<DeviceCtx>(host.N, host.primes, physical_device(ids[m]));
      upload_keys(root->h);
      by_device[ids[m]] = root;
    } else {
      root = it->second;
    }
    g.roots.push_back(root);
    // a member is its device's root context itself only when it is the first user of a NEW state, or
    // member 0 of a group that may issue on `first` directly
    const bool own_root = is_new || (m == 0 && !fork_first);
    g.forks.push_back(own_root ? nullptr : std::make_shared<Fork>(root));
    g.ctx.push_back(own_root ? root->h : g.forks.back()->h);
  }
  enable_peers(g);
  return g;
}

// Limb sharding: member s is shard s of G = ids.size(), each with a device state of its own whose shard
// map is set BEFORE the keys go up — the library then keeps only that shard's prime rows of every key
// (evah_key_upload: its data limbs + the special prime, (ceil((k-1)/G) + 1)/k of the key), so a key set
// that does not fit one device's memory is spread over the group.
inline DeviceGroup make_limb_group(const std::vector<int> &ids, const HostContext &host, const std::function<void(evah_ctx *)> &upload_keys) {
  DeviceGroup g;
  g.ids = ids;
  const uint32_t G = (uint32_t)ids.size();
  for (uint32_t s = 0; s < G; s++) {
    auto root = std::make_shared<DeviceCtx>(host.N, host.primes, physical_device(ids[s]));
    chk(evah_ctx_set_shard(root->h, s, G));
    upload_keys(root->h);
    g.roots.push_back(root);
    g.forks.push_back(nullptr);
    g.ctx.push_back(root->h);
  }
  enable_peers(g);
  return g;
}

// One process per GPU (torchrun): this process is shard `rank` of `world`; the other members of the group are
// other processes (ctx[s] == nullptr) and the exchange steps go through LimbHooks (RCCL collectives).
inline DeviceGroup make_limb_group_rank(int device, uint32_t rank, uint32_t world, const HostContext &host,
                                        const std::function<void(evah_ctx *)> &upload_keys) {
  DeviceGroup g;
  g.ids.assign(world, -1);
  g.roots.assign(world, nullptr);
  g.forks.assign(world, nullptr);
  g.ctx.assign(world, nullptr);
  auto root = std::make_shared<DeviceCtx>(host.N, host.primes, device);
  chk(evah_ctx_set_shard(root->h, rank, world));
  upload_keys(root->h);
  g.ids[rank] = device;
  g.roots[rank] = root;
  g.ctx[rank] = root->h;
  return g;
}

// Exchange steps of a limb group whose shards live in DIFFERENT processes: collectives on the library's device
// buffers (RCCL through torch.distributed under torchrun; the bindings call back into Python).  Unset: all shards are
// here and the exchanges are evah_buf_gather launches.
struct LimbHooks {
  std::function<void(void *dev_ptr, size_t chunk_words)> all_gather;                 // in place: chunk r of the buffer from rank r
  std::function<void(void *dev_ptr, size_t words, uint32_t owner)> broadcast;        // in place, from rank `owner`
  std::function<void(u64 *host_words, size_t words)> sum_host;                       // in place all-reduce (sum) of host words
  explicit operator bool() const { return (bool)all_gather; }
};

// ------------------------------------------------------------------------------------ sub-DAG split

inline uint32_t op_arity(uint32_t op) { return (op == (uint32_t)Op::Add || op == (uint32_t)Op::Sub || op == (uint32_t)Op::Mul) ? 2u : 1u; }

struct SubDagPlan {
  std::vector<uint32_t> prefix, suffix;                             // op indices, on member 0
  std::vector<std::pair<uint32_t, std::vector<uint32_t>>> components; // (member, op indices)
};

// The cut is the pair of levels between which the op DAG falls into the most evenly loaded
// independent components (longest first over the members); no worthwhile cut: everything in the prefix.
inline SubDagPlan plan_subdag(const std::vector<evah_op> &ops, const std::function<bool(uint32_t)> &placed_ct, uint32_t n_dev) {
  const size_t n = ops.size();
  std::unordered_map<uint32_t, size_t> producer;
  for (size_t i = 0; i < n; i++) producer[ops[i].dst] = i;
  auto src = [&](size_t i, uint32_t k) { return k == 0 ? ops[i].src0 : ops[i].src1; };
  std::vector<int> level(n, 0);
  for (size_t i = 0; i < n; i++)
    for (uint32_t k = 0; k < op_arity(ops[i].op); k++) {
      auto it = producer.find(src(i, k));
      if (it != producer.end()) level[i] = std::max(level[i], level[it->second] + 1);
    }
  auto is_ct = [&](uint32_t slot) { return producer.count(slot) || placed_ct(slot); };
  std::vector<long> cost(n, 1);
  for (size_t i = 0; i < n; i++) {
    const uint32_t o = ops[i].op;
    const bool heavy = o == (uint32_t)Op::RotateLeftConst || o == (uint32_t)Op::RotateRightConst || o == (uint32_t)Op::Relinearize ||
                       o == (uint32_t)Op::Rescale || (o == (uint32_t)Op::Mul && is_ct(ops[i].src0) && is_ct(ops[i].src1));
    if (heavy) cost[i] = 10;
  }
  SubDagPlan all;
  all.prefix.resize(n);
  std::iota(all.prefix.begin(), all.prefix.end(), 0u);
  if (n_dev < 2 || n == 0) return all;
  const int depth = *std::max_element(level.begin(), level.end()) + 1;
  const long serial = std::accumulate(cost.begin(), cost.end(), 0L);
  long best = serial;
  int best_lo = -1, best_hi = -1;
  std::vector<std::pair<uint32_t, std::vector<uint32_t>>> best_assign;
  for (int lo = 0; lo < depth; lo++)
    for (int hi = lo + 1; hi <= depth; hi++) {
      std::vector<uint32_t> region;
      for (size_t i = 0; i < n; i++)
        if (level[i] >= lo && level[i] < hi) region.push_back((uint32_t)i);
      if (region.size() < 2) continue;
      std::unordered_map<uint32_t, uint32_t> parent;
      for (uint32_t i : region) parent[i] = i;
      std::function<uint32_t(uint32_t)> find = [&](uint32_t x) {
        while (parent[x] != x) { parent[x] = parent[parent[x]]; x = parent[x]; }
        return x;
      };
      for (uint32_t i : region)
        for (uint32_t k = 0; k < op_arity(ops[i].op); k++) {
          auto it = producer.find(src(i, k));
          if (it != producer.end() && parent.count((uint32_t)it->second)) parent[find(i)] = find((uint32_t)it->second);
        }
      std::map<uint32_t, std::vector<uint32_t>> comps;
      for (uint32_t i : region) comps[find(i)].push_back(i);
      if (comps.size() < 2) continue;
      std::vector<std::vector<uint32_t>> list;
      for (auto &kv : comps) list.push_back(kv.second);
      auto load_of = [&](const std::vector<uint32_t> &c) { long s = 0; for (uint32_t i : c) s += cost[i]; return s; };
      std::stable_sort(list.begin(), list.end(), [&](const auto &a, const auto &b) { return load_of(a) > load_of(b); });
      std::vector<long> loads(n_dev, 0);
      std::vector<std::pair<uint32_t, std::vector<uint32_t>>> assign;
      for (auto &c : list) {
        const uint32_t d = (uint32_t)(std::min_element(loads.begin(), loads.end()) - loads.begin());
        loads[d] += load_of(c);
        std::sort(c.begin(), c.end());
        assign.emplace_back(d, c);
      }
      long crossing = 0;
      for (auto &dc : assign) {
        if (dc.first == 0) continue;
        std::set<uint32_t> made;
        for (uint32_t i : dc.second) made.insert(ops[i].dst);
        for (uint32_t i : dc.second)
          for (uint32_t k = 0; k < op_arity(ops[i].op); k++)
            if (!made.count(src(i, k))) crossing++;
      }
      long outside = 0;
      for (size_t i = 0; i < n; i++)
        if (level[i] < lo || level[i] >= hi) outside += cost[i];
      const long est = outside + *std::max_element(loads.begin(), loads.end()) + 2 * crossing;
      if (est < best) { best = est; best_lo = lo; best_hi = hi; best_assign = assign; }
    }
  if (best_lo < 0 || best > serial * 9 / 10) return all;
  SubDagPlan p;
  for (size_t i = 0; i < n; i++) {
    if (level[i] < best_lo) p.prefix.push_back((uint32_t)i);
    else if (level[i] >= best_hi) p.suffix.push_back((uint32_t)i);
  }
  p.components = std::move(best_assign);
  return p;
}

// Runs an evah_execute op list over the members of a group: prefix and suffix on member 0, the
// components on their members; every piece is one asynchronous submit.  `table` is the caller's
// value table on member 0 (inputs and plaintexts placed, as for evah_execute); on return the slots the
// ops wrote hold handles the caller owns (program outputs and whatever was not released), all of them
// on member 0's device.  Returns the plan that was used.
inline SubDagPlan run_subdag(const DeviceGroup &g, const std::vector<evah_op> &ops, std::vector<evah_val> &table,
                             const std::set<uint32_t> &keep /* slots the caller reads afterwards (outputs) */) {
  const uint32_t n_vals = (uint32_t)table.size();
  auto src = [&](const evah_op &o, uint32_t k) { return k == 0 ? o.src0 : o.src1; };
  SubDagPlan plan = plan_subdag(ops, [&](uint32_t s) { return s < n_vals && table[s].kind == EVAH_VAL_CT; }, (uint32_t)g.size());
  std::unordered_map<uint32_t, uint32_t> consumers;
  for (auto &o : ops)
    for (uint32_t k = 0; k < op_arity(o.op); k++) consumers[src(o, k)]++;
  const size_t G = g.size();
  std::vector<std::vector<evah_val>> tab(G, std::vector<evah_val>(n_vals, evah_val{EVAH_VAL_NONE, nullptr}));
  tab[0] = table;
  // handles made here that the caller does not get: copies on other members, intermediates
  std::vector<std::pair<evah_ctx *, evah_val>> owned;
  std::set<std::pair<uint32_t, uint32_t>> is_copy; // (member, slot) filled by a copy, not by an op
  auto fetch = [&](uint32_t d, uint32_t s) {
    if (tab[d][s].kind != EVAH_VAL_NONE) return;
    uint32_t e = 0;
    while (e < G && tab[e][s].kind == EVAH_VAL_NONE) e++;
    if (e == G) throw std::runtime_error("sub-DAG split: operand was never produced");
    if (tab[e][s].kind == EVAH_VAL_CT) {
      evah_ct *h = nullptr;
      chk(evah_ct_copy(g.ctx[d], static_cast<evah_ct *>(tab[e][s].h), &h)); // peer copy when the members sit on different GPUs
      tab[d][s] = evah_val{EVAH_VAL_CT, h};
    } else {
      evah_pt *h = nullptr;
      chk(evah_pt_copy(g.ctx[d], static_cast<evah_pt *>(tab[e][s].h), &h));
      tab[d][s] = evah_val{EVAH_VAL_PT, h};
    }
    owned.emplace_back(g.ctx[d], tab[d][s]);
    is_copy.insert({d, s});
  };
  std::vector<std::pair<uint32_t, uint32_t>> produced; // (member, slot) written by an op and not released by a flag
  auto run = [&](uint32_t d, const std::vector<uint32_t> &idx) {
    if (idx.empty()) return;
    std::set<uint32_t> inside;
    for (uint32_t i : idx) inside.insert(ops[i].dst);
    std::unordered_map<uint32_t, uint32_t> reads, seen;
    for (uint32_t i : idx)
      for (uint32_t k = 0; k < op_arity(ops[i].op); k++) {
        const uint32_t s = src(ops[i], k);
        reads[s]++;
        if (!inside.count(s)) fetch(d, s);
      }
    std::vector<evah_op> sub;
    std::set<uint32_t> released;
    for (uint32_t i : idx) {
      evah_op o = ops[i];
      o.flags = 0;
      const uint32_t ar = op_arity(o.op);
      for (uint32_t k = 0; k < ar; k++) {
        if (k == 1 && o.src0 == o.src1) continue;
        const uint32_t s = src(o, k);
        seen[s] += (ar == 2 && o.src0 == o.src1) ? 2 : 1;
        // an intermediate of this piece whose every reader is in this piece: released at its last use
        if (inside.count(s) && !keep.count(s) && consumers[s] == reads[s] && seen[s] == reads[s]) {
          o.flags |= k == 0 ? EVAH_OPF_FREE_SRC0 : EVAH_OPF_FREE_SRC1;
          released.insert(s);
        }
      }
      sub.push_back(o);
    }
    const int rc = evah_execute(g.ctx[d], sub.data(), (uint32_t)sub.size(), tab[d].data(), n_vals);
    for (uint32_t s : inside)
      if (tab[d][s].kind != EVAH_VAL_NONE && !released.count(s)) produced.emplace_back(d, s);
    chk(rc);
  };
  auto cleanup = [&](bool failed) {
    // what the caller gets: every slot an op produced and did not release, brought to member 0
    for (auto &ds : produced) {
      const uint32_t d = ds.first, s = ds.second;
      if (tab[d][s].kind == EVAH_VAL_NONE) continue;
      if (d == 0) { table[s] = tab[0][s]; continue; }
      if (!failed && keep.count(s)) {
        evah_ct *h = nullptr;
        if (evah_ct_copy(g.ctx[0], static_cast<evah_ct *>(tab[d][s].h), &h) == 0) table[s] = evah_val{EVAH_VAL_CT, h};
      }
      owned.emplace_back(g.ctx[d], tab[d][s]); // the remote original is ours to release
    }
    for (auto &ov : owned) {
      if (ov.second.kind == EVAH_VAL_CT) evah_ct_free(ov.first, static_cast<evah_ct *>(ov.second.h));
      else if (ov.second.kind == EVAH_VAL_PT) evah_pt_free(ov.first, static_cast<evah_pt *>(ov.second.h));
    }
  };
  try {
    run(0, plan.prefix);
    for (auto &dc : plan.components) run(dc.first, dc.second);
    run(0, plan.suffix);
  } catch (...) {
    cleanup(true);
    throw;
  }
  cleanup(false);
  return plan;
}

// ------------------------------------------------------------------------------------ limb sharding

// A value dealt over the shards: parts[s] is shard s's handle over its local limbs (null when the
// shard owns no limb at this level).  size 0: a plaintext.
struct ShardedValue {
  std::vector<std::shared_ptr<CtHandle>> ct;
  std::vector<std::shared_ptr<PtHandle>> pt;
  uint32_t size = 0, limbs = 0;
  double scale = 1.0;
  bool is_ct() const { return size > 0; }
};

// RAII exchange buffer of one operation (returned to the shard's pool; the pool recycles it only
// after the other queues' reads of it)
struct ShardBuf {
  evah_ctx *c = nullptr;
  evah_buf *b = nullptr;
  ShardBuf() {}
  ShardBuf(evah_ctx *ctx, size_t words) : c(ctx) { chk(evah_buf_alloc(ctx, words, &b)); }
  ShardBuf(ShardBuf &&o) noexcept : c(o.c), b(o.b) { o.b = nullptr; }
  ShardBuf &operator=(ShardBuf &&o) noexcept { reset(); c = o.c; b = o.b; o.b = nullptr; return *this; }
  ShardBuf(const ShardBuf &) = delete;
  ShardBuf &operator=(const ShardBuf &) = delete;
  void reset() { if (b) evah_buf_free(c, b); b = nullptr; }
  ~ShardBuf() { reset(); }
};

// The evaluator calls of SEALExecutor (seal_executor.h:114-243) over limb-sharded values, all G
// shards in this process: the exchange steps are device / peer copies enqueued on the receiving
// shard's queue (evah_buf_copy), ordered after the producing shard's work by the library.
class LimbShardEvaluator {
public:
  LimbShardEvaluator(const HostContext &hc, DeviceGroup grp, LimbHooks hk = LimbHooks())
      : host(hc), g(std::move(grp)), G((uint32_t)g.size()), hooks(std::move(hk)) {
    for (uint32_t s = 0; s < G; s++)
      if (local(s)) chk(evah_ctx_set_shard(g.ctx[s], s, G));
    for (uint32_t s = 0; s < G; s++)
      if (!local(s) && !hooks) throw std::logic_error("limb group with remote shards needs exchange hooks");
  }
  uint32_t shards() const { return G; }
  const DeviceGroup &group() const { return g; }
  // shard s is a context of this process (all of them unless the group spans processes)
  bool local(uint32_t s) const { return g.ctx[s] != nullptr; }
  evah_ctx *any_ctx() const { for (uint32_t s = 0; s < G; s++) if (local(s)) return g.ctx[s]; return nullptr; }

  // data: all limbs, [size][l][N] (plaintext: [l][N], size 0)
  ShardedValue upload(const u64 *data, uint32_t size, uint32_t l, double scale) {
    ShardedValue v;
    v.size = size;
    v.limbs = l;
    v.scale = scale;
    const size_t N = host.N;
    const uint32_t polys = size ? size : 1;
    (size ? (void)v.ct.resize(G) : (void)v.pt.resize(G));
    std::vector<u64> local;
    for (uint32_t s = 0; s < G; s++) {
      const uint32_t nl = s < l ? (l - s + G - 1) / G : 0;
      if (!nl || !this->local(s)) continue;
      local.resize((size_t)polys * nl * N);
      for (uint32_t p = 0; p < polys; p++)
        for (uint32_t j = 0; j < nl; j++)
          std::memcpy(local.data() + ((size_t)p * nl + j) * N, data + ((size_t)p * l + s + (size_t)j * G) * N, sizeof(u64) * N);
      if (size) {
        evah_ct *h = nullptr;
        chk(evah_ct_upload(g.ctx[s], size, nl, scale, (const uint64_t *)local.data(), &h));
        v.ct[s] = std::make_shared<CtHandle>(g.ctx[s], h);
      } else {
        evah_pt *h = nullptr;
        chk(evah_pt_upload(g.ctx[s], nl, scale, (const uint64_t *)local.data(), &h));
        v.pt[s] = std::make_shared<PtHandle>(g.ctx[s], h);
      }
    }
    return v;
  }
  // -> all limbs on the host, [size][l][N]
  HostCipher download(const ShardedValue &v) {
    HostCipher out;
    out.size = v.size;
    out.limbs = v.limbs;
    out.scale = v.scale;
    const size_t N = host.N;
    out.data.resize((size_t)v.size * v.limbs * N);
    out.words_checked = true;
    if (hooks) std::memset(out.data.data(), 0, sizeof(u64) * out.data.size()); // remote limbs arrive through sum_host
    std::vector<u64> local;
    for (uint32_t s = 0; s < G; s++) {
      if (!v.ct[s]) continue;
      const uint32_t nl = (v.limbs - s + G - 1) / G;
      local.resize((size_t)v.size * nl * N);
      chk(evah_ct_download(g.ctx[s], v.ct[s]->h, (uint64_t *)local.data()));
      for (uint32_t p = 0; p < v.size; p++)
        for (uint32_t j = 0; j < nl; j++)
          std::memcpy(out.data.data() + ((size_t)p * v.limbs + s + (size_t)j * G) * N, local.data() + ((size_t)p * nl + j) * N, sizeof(u64) * N);
    }
    // every limb is owned by exactly one rank and zero elsewhere: the sum over ranks is the whole ciphertext
    if (hooks) hooks.sum_host(out.data.data(), out.data.size());
    return out;
  }

  std::vector<u64> download_plain(const ShardedValue &v) {
    const size_t N = host.N;
    std::vector<u64> out((size_t)v.limbs * N), local;
    for (uint32_t s = 0; s < G; s++) {
      if (!v.pt[s]) continue;
      const uint32_t nl = (v.limbs - s + G - 1) / G;
      local.resize((size_t)nl * N);
      chk(evah_pt_download(g.ctx[s], v.pt[s]->h, (uint64_t *)local.data()));
      for (uint32_t j = 0; j < nl; j++) std::memcpy(out.data() + ((size_t)s + (size_t)j * G) * N, local.data() + (size_t)j * N, sizeof(u64) * N);
    }
    if (hooks) hooks.sum_host(out.data(), out.size()); // the other ranks' limbs (zero here)
    return out;
  }

  // ---- per-limb operations (no exchange)
  ShardedValue add(const ShardedValue &a, const ShardedValue &b) { same_level(a, b, true); return each2(a, b, std::max(a.size, b.size), a.scale, evah_add); }
  ShardedValue sub(const ShardedValue &a, const ShardedValue &b) { same_level(a, b, true); return each2(a, b, std::max(a.size, b.size), a.scale, evah_sub); }
  ShardedValue add_plain(const ShardedValue &a, const ShardedValue &p) { same_level(a, p, true); return each_plain(a, p, a.scale, evah_add_plain); }
  ShardedValue sub_plain(const ShardedValue &a, const ShardedValue &p) { same_level(a, p, true); return each_plain(a, p, a.scale, evah_sub_plain); }
  ShardedValue multiply_plain(const ShardedValue &a, const ShardedValue &p) {
    same_level(a, p, false);
    check_scale(a.scale * p.scale, a.limbs);
    return each_plain(a, p, a.scale * p.scale, evah_multiply_plain);
  }
  ShardedValue negate(const ShardedValue &a) { return each1(a, a.size, a.limbs, a.scale, [](evah_ctx *c, const evah_ct *x, evah_ct **o) { return evah_negate(c, x, o); }); }
  ShardedValue multiply(const ShardedValue &a, const ShardedValue &b) {
    same_level(a, b, false);
    if (a.size != 2 || b.size != 2) throw std::runtime_error("multiply supports size-2 operands only (relinearize first)");
    check_scale(a.scale * b.scale, a.limbs);
    return each2(a, b, 3, a.scale * b.scale, evah_multiply);
  }
  ShardedValue square(const ShardedValue &a) {
    if (a.size != 2) throw std::runtime_error("square supports size-2 operands only (relinearize first)");
    check_scale(a.scale * a.scale, a.limbs);
    return each1(a, 3, a.limbs, a.scale * a.scale, [](evah_ctx *c, const evah_ct *x, evah_ct **o) { return evah_square(c, x, o); });
  }
  // drop the last limb: a view on its owner, nothing to do on the other shards
  ShardedValue mod_switch(const ShardedValue &a) {
    if (a.limbs < 2) throw std::runtime_error("end of modulus switching chain reached");
    const uint32_t owner = (a.limbs - 1) % G;
    ShardedValue o = a;
    o.limbs = a.limbs - 1;
    if (a.ct[owner]) {
      if (a.limbs - 1 > owner) {
        evah_ct *h = nullptr;
        chk(evah_mod_switch(g.ctx[owner], a.ct[owner]->h, &h));
        o.ct[owner] = std::make_shared<CtHandle>(g.ctx[owner], h);
      } else {
        o.ct[owner] = nullptr;
      }
    }
    return o;
  }

  // ---- operations with an exchange step
  ShardedValue relinearize(const ShardedValue &a) {
    if (a.size != 3) throw std::runtime_error("relinearize expects a size-3 ciphertext");
    return key_switch(a, 2, a.limbs, EVAH_KEY_RELIN, 0, &a, 2, a.scale);
  }
  ShardedValue rotate(const ShardedValue &a, int32_t steps) {
    if (a.size != 2) throw std::runtime_error("rotate expects a size-2 ciphertext (relinearize first)");
    if (steps == 0) return a;
    uint32_t elt = 0;
    chk(evah_galois_elt_from_step(any_ctx(), steps, &elt));
    ShardedValue perm = each1(a, 2, a.limbs, a.scale, [elt](evah_ctx *c, const evah_ct *x, evah_ct **o) { return evah_shard_galois_perm(c, x, elt, o); });
    return key_switch(perm, 1, a.limbs, EVAH_KEY_GALOIS, elt, &perm, 1, a.scale);
  }
  ShardedValue rescale(const ShardedValue &a, uint32_t divisor_bits) {
    const uint32_t l = a.limbs;
    if (l < 2) throw std::runtime_error("end of modulus switching chain reached");
    const uint32_t owner = (l - 1) % G;
    const size_t N = host.N;
    std::vector<ShardBuf> rbuf;
    for (uint32_t s = 0; s < G; s++) {
      if (local(s)) rbuf.emplace_back(g.ctx[s], 3 * N);
      else rbuf.emplace_back();
    }
    if (local(owner)) chk(evah_shard_rescale_last(g.ctx[owner], a.ct[owner]->h, l, rbuf[owner].b));
    broadcast(rbuf, owner, (size_t)a.size * N); // ---- exchange: INTT of the last limb
    ShardedValue o;
    o.size = a.size;
    o.limbs = l - 1;
    o.scale = a.scale / std::pow(2.0, (double)divisor_bits);
    o.ct.resize(G);
    for (uint32_t s = 0; s < G; s++) {
      if (s >= l - 1 || !local(s)) continue;
      evah_ct *h = nullptr;
      chk(evah_shard_rescale_finish(g.ctx[s], a.ct[s]->h, l, rbuf[s].b, divisor_bits, &h));
      o.ct[s] = std::make_shared<CtHandle>(g.ctx[s], h);
    }
    return o;
  }
  void sync() { for (uint32_t s = 0; s < G; s++) if (local(s)) chk(evah_ctx_sync(g.ctx[s])); }

  // exchange traffic of this evaluator so far (words moved between shards), for bench / tests
  uint64_t exchanged_words = 0;
  // launches those words took: per key switch G (all-gather, one per receiving shard) + G - 1 (broadcast), per rescale G - 1
  uint64_t exchange_launches = 0;

private:
  const HostContext &host;
  DeviceGroup g;
  uint32_t G;
  LimbHooks hooks; // set: the group spans processes, the exchange steps are collectives

  void same_level(const ShardedValue &a, const ShardedValue &b, bool scales) const {
    if (a.limbs != b.limbs) throw std::runtime_error("encrypted1 and encrypted2 parameter mismatch");
    if (scales && a.scale != b.scale) throw std::runtime_error("scale mismatch");
  }
  void check_scale(double scale, uint32_t limbs) const {
    if (!(scale > 0) || (int)std::log2(scale) >= host.total_bits[limbs]) throw std::runtime_error("scale out of bounds");
  }
  template <class F> ShardedValue each1(const ShardedValue &a, uint32_t size, uint32_t limbs, double scale, F fn) {
    ShardedValue o;
    o.size = size;
    o.limbs = limbs;
    o.scale = scale;
    o.ct.resize(G);
    for (uint32_t s = 0; s < G; s++) {
      if (!a.ct[s] || !local(s)) continue;
      evah_ct *h = nullptr;
      chk(fn(g.ctx[s], a.ct[s]->h, &h));
      o.ct[s] = std::make_shared<CtHandle>(g.ctx[s], h);
    }
    return o;
  }
  template <class F> ShardedValue each2(const ShardedValue &a, const ShardedValue &b, uint32_t size, double scale, F fn) {
    ShardedValue o;
    o.size = size;
    o.limbs = a.limbs;
    o.scale = scale;
    o.ct.resize(G);
    for (uint32_t s = 0; s < G; s++) {
      if (!a.ct[s] || !b.ct[s] || !local(s)) continue;
      evah_ct *h = nullptr;
      chk(fn(g.ctx[s], a.ct[s]->h, b.ct[s]->h, &h));
      o.ct[s] = std::make_shared<CtHandle>(g.ctx[s], h);
    }
    return o;
  }
  template <class F> ShardedValue each_plain(const ShardedValue &a, const ShardedValue &p, double scale, F fn) {
    ShardedValue o;
    o.size = a.size;
    o.limbs = a.limbs;
    o.scale = scale;
    o.ct.resize(G);
    for (uint32_t s = 0; s < G; s++) {
      if (!a.ct[s] || !p.pt[s] || !local(s)) continue;
      evah_ct *h = nullptr;
      chk(fn(g.ctx[s], a.ct[s]->h, p.pt[s]->h, &h));
      o.ct[s] = std::make_shared<CtHandle>(g.ctx[s], h);
    }
    return o;
  }

  // bufs[s]: buffer of G chunks, chunk s filled by shard s -> every buffer complete.  ONE launch per receiving shard
  // (evah_buf_gather: a kernel on the receiver's queue that reads the other shards' chunks as peers), not G - 1 copies
  void all_gather(std::vector<ShardBuf> &bufs, size_t chunk) {
    if (G < 2 && !hooks) return;
    if (hooks) { // one collective on this rank's buffer (chunk r from rank r), on the stream its kernels run on
      for (uint32_t s = 0; s < G; s++)
        if (local(s)) hooks.all_gather(evah_buf_ptr(bufs[s].b), chunk);
      exchanged_words += chunk * (G - 1);
      exchange_launches++;
      return;
    }
    for (uint32_t d = 0; d < G; d++) {
      std::vector<const evah_buf *> srcs;
      std::vector<size_t> offs;
      for (uint32_t s = 0; s < G; s++)
        if (s != d) {
          srcs.push_back(bufs[s].b);
          offs.push_back(s * chunk);
        }
      chk(evah_buf_gather(g.ctx[d], bufs[d].b, (uint32_t)srcs.size(), srcs.data(), offs.data(), offs.data(), chunk));
      exchanged_words += chunk * srcs.size();
      exchange_launches++;
    }
  }
  void broadcast(std::vector<ShardBuf> &bufs, uint32_t owner, size_t words) {
    if (hooks) {
      for (uint32_t s = 0; s < G; s++)
        if (local(s)) hooks.broadcast(evah_buf_ptr(bufs[s].b), words, owner);
      exchanged_words += words;
      exchange_launches++;
      return;
    }
    const size_t zero = 0;
    for (uint32_t d = 0; d < G; d++)
      if (d != owner) {
        const evah_buf *src = bufs[owner].b;
        chk(evah_buf_gather(g.ctx[d], bufs[d].b, 1, &src, &zero, &zero, words));
        exchanged_words += words;
        exchange_launches++;
      }
  }
  // target.ct[s] holds the key-switch target as polynomial `poly`; returns the size-2 result
  ShardedValue key_switch(const ShardedValue &target, uint32_t poly, uint32_t l, int kind, uint32_t elt, const ShardedValue *add,
                          uint32_t add_polys, double scale) {
    const size_t N = host.N;
    const uint32_t rows = (l + G - 1) / G;
    const size_t chunk = (size_t)rows * N;
    std::vector<ShardBuf> dig, prod, rbuf;
    for (uint32_t s = 0; s < G; s++) {
      if (local(s)) dig.emplace_back(g.ctx[s], G * chunk);
      else dig.emplace_back();
    }
    for (uint32_t s = 0; s < G; s++)
      if (target.ct[s] && local(s)) chk(evah_shard_ks_digits(g.ctx[s], target.ct[s]->h, poly, l, dig[s].b, rows));
    all_gather(dig, chunk); // ---- exchange 1: the l coefficient-form digits
    const uint32_t owner = l % G;
    for (uint32_t s = 0; s < G; s++) {
      const uint32_t nl = s < l ? (l - s + G - 1) / G : 0;
      if (local(s)) {
        prod.emplace_back(g.ctx[s], (size_t)2 * (nl + 1) * N);
        rbuf.emplace_back(g.ctx[s], 3 * N);
      } else {
        prod.emplace_back();
        rbuf.emplace_back();
      }
    }
    for (uint32_t s = 0; s < G; s++)
      if (local(s) && (target.ct[s] || s == owner))
        chk(evah_shard_ks_products(g.ctx[s], target.ct[s] ? target.ct[s]->h : nullptr, poly, l, dig[s].b, rows, kind, elt, prod[s].b, rbuf[s].b));
    broadcast(rbuf, owner, 2 * N); // ---- exchange 2: INTT of the special limb
    ShardedValue o;
    o.size = 2;
    o.limbs = l;
    o.scale = scale;
    o.ct.resize(G);
    for (uint32_t s = 0; s < G && s < l; s++) {
      if (!local(s)) continue;
      evah_ct *h = nullptr;
      chk(evah_shard_ks_finish(g.ctx[s], l, prod[s].b, rbuf[s].b, add && add->ct[s] ? add->ct[s]->h : nullptr, add_polys, scale, &h));
      o.ct[s] = std::make_shared<CtHandle>(g.ctx[s], h);
    }
    return o;
  }
};

} // namespace evahost
