// public_ctx.h — HipPublic, the public half of a key pair (SEALPublic, /root/reference/eva/seal/seal.h:45-97,
// seal.cpp:24-122): execute() with its eager walk -> hipGraph capture -> replay, valuations resident in HBM, the
// issue queues, and the choice of a multi-device mode inside execute() as the reference chooses its parallel
// traversal inside SEALPublic::execute (seal.cpp:105-113).  The larger member functions are defined in plans.h,
// batch.h, limb_exec.h and client.h (included at the end).  Included by executor.h; not a stand-alone header.
#pragma once

namespace evahost {

// ---- contexts (seal.h:45-97)
// The device state generate_keys() hands to BOTH halves of a key pair: a valuation produced by the
// public context can then be decrypted by the secret context without leaving the device.  Contexts
// loaded from files get a holder of their own.
struct DeviceHolder {
  std::shared_ptr<DeviceCtx> dev;
};

class HipPublic {
public:
  std::shared_ptr<HostContext> host;
  PublicKey pk;
  SwitchKey relin;
  std::map<uint32_t, SwitchKey> galois; // by Galois element
  int device = 0;
  bool free_eagerly = true;
  std::array<double, 3> last_timing{0, 0, 0}; // ms: input upload, DAG enqueue (host), drain + output download
  // Valuations stay on the device (the valuation "may hold device handles"): encrypt()
  // leaves its ciphertexts in HBM, execute() takes and returns handles and does NOT wait for the GPU,
  // decrypt() reads handles; host words appear when somebody asks for them (get(), save(), a context on
  // another device).  EVA_RESIDENT=0 restores host valuations (every call copies in and out and waits).
  bool resident = std::getenv("EVA_RESIDENT") ? std::atoi(std::getenv("EVA_RESIDENT")) != 0 : true;
  // Device-resident inputs above this many bytes are walked eagerly instead of replaying the captured
  // graph: a replay would first copy them into the graph's fixed input slots (and its outputs out
  // again), and launches of that size gain nothing from a graph.
  size_t graph_copy_limit = (size_t)32 << 20;
  std::shared_ptr<DeviceHolder> holder = std::make_shared<DeviceHolder>();
  // Several GPUs behind ONE execute() — the counterpart of the reference choosing its parallel
  // traversal inside SEALPublic::execute (seal.cpp:105-113).  `devices`: device index per member (a
  // repeated index = several contexts on one GPU, how a 1-GPU box validates the paths); `shard_mode`:
  //   "subdag"  independent sub-DAGs of the program on different members (multi_device.h)
  //   "limb"    RNS limbs dealt over the members, all-gather + broadcast per key switch
  //   "dag"     execute_batch deals the groups of a batch over the members (instances are independent)
  // Environment: EVA_NUM_GPUS=n (devices 0..n-1) or EVA_DEVICES=0,1,... and EVA_SHARD=subdag|limb|dag.
  std::vector<int> devices = devices_from_env();
  std::string shard_mode = std::getenv("EVA_SHARD") ? std::getenv("EVA_SHARD") : "";
  static std::vector<int> devices_from_env() {
    std::vector<int> d;
    if (const char *e = std::getenv("EVA_DEVICES")) {
      for (const char *p = e; *p;) {
        d.push_back(std::atoi(p));
        while (*p && *p != ',') p++;
        if (*p == ',') p++;
      }
    } else if (const char *n = std::getenv("EVA_NUM_GPUS")) {
      for (int i = 0; i < std::atoi(n); i++) d.push_back(i);
    }
    return d;
  }
  // what the last multi-device execute() did: pieces per member (sub-DAG) / words exchanged (limb)
  std::vector<std::pair<uint32_t, uint32_t>> last_subdag_plan; // (member, ops) with member 0 first = prefix, last = suffix
  uint64_t last_exchanged_words = 0, last_exchange_launches = 0;
  // Limb sharding across PROCESSES (one process per GPU under torchrun): this context is shard limb_rank of
  // limb_world, the exchange steps of a key switch / rescale are collectives on the library's device buffers
  // (limb_hooks: RCCL through torch.distributed, eva_amd/dist.py attach_limb_dist), issued on limb_stream — the stream
  // the shard's kernels run on as well, so nothing synchronises with the host between phases.  limb_world <= 1: all
  // shards are contexts of this process (`devices`).
  uint32_t limb_rank = 0, limb_world = 1;
  LimbHooks limb_hooks;
  uintptr_t limb_stream = 0;
  // HIP streams independent DAG nodes are spread over (EVA_NUM_STREAMS).  Default 1: at these
  // kernel sizes a single in-order queue keeps the GPU as busy as the host can feed it; more
  // queues are correct (ordering is enforced per buffer inside libeva_hip.so) and pay off when
  // nodes are large enough to be GPU-bound.
  int num_queues = 1;

  // SEALPublic::encrypt (seal.cpp:24-102)
  HipValuation encrypt(const Valuation &inputs, const CKKSSignature &sig);

  // SEALPublic::execute (seal.cpp:104-122) — THE hot path.  First call for a program: upload
  // inputs, walk the DAG issuing HIP work over the queues, download outputs.  From the second call
  // on (same program object, same input shapes, no Raw inputs) the whole walk is replayed from a
  // captured hipGraph: per call the host refills the input slots, launches one graph, downloads.
  bool use_graphs = true; // EVA_GRAPH=0 disables
  HipValuation execute(Program &program, const HipValuation &inputs) {
    const bool multi = devices.size() > 1;
    if ((multi || limb_hooks) && shard_mode == "limb") {
      ensure_device(false);
      return execute_limb(program, inputs);
    }
    ensure_device();
    const bool subdag = multi && shard_mode == "subdag";
    if (!subdag && graphs_enabled() && graphable(program, inputs) && resident_bytes(inputs) <= graph_copy_limit) {
      auto it = plans.find(&program);
      if (it == plans.end() && !no_graph.count(&program)) {
        seen[&program]++;
        if (seen[&program] >= 2) {
          // capture can fail (out of memory for the second buffer set, a runtime refusing the
          // capture or the instantiation, a first-use table build inside it): the eager walk that
          // served the first call still works, so remember the program as not graphable and go on
          try {
            it = plans.emplace(&program, build_plan(program, inputs)).first;
          } catch (const std::exception &e) {
            no_graph.insert(&program);
            if (std::getenv("EVA_VERBOSE")) std::fprintf(stderr, "EVA: graph capture disabled for this program: %s\n", e.what());
          }
        }
      }
      if (it != plans.end()) {
        if (it->second->matches(program, inputs)) {
          // r6: a replay is one in-order chain of mostly latency-bound launches; two chains side by side fill the chip
          // far better (config 5: 1.49 -> 1.1 ms per call, Harris 0.70 -> 0.53, as the eager walks on two queues already
          // showed).  A caller that issues the next execute() while the previous replay is still running gets a TWIN
          // plan — the same walk captured once more, with its own slots and buffers, on a queue of its own — and the
          // calls go to whichever plan is idle (else to the one not used last).  A caller that waits for every result
          // never has a busy plan and never pays for the twin.  EVA_GRAPH_TWIN=0 disables.
          GraphPlan *use = it->second.get();
          int busy = 0;
          if (resident && twin_plans && evah_ctx_busy(use->queues[0]->h, &busy) == 0 && busy) {
            auto it2 = plans2.find(&program);
            if (it2 != plans2.end() && !it2->second->matches(program, inputs)) { plans2.erase(it2); it2 = plans2.end(); }
            if (it2 == plans2.end() && !no_twin.count(&program)) {
              try {
                it2 = plans2.emplace(&program, build_plan(program, inputs)).first;
              } catch (const std::exception &e) {
                no_twin.insert(&program);
                if (std::getenv("EVA_VERBOSE")) std::fprintf(stderr, "EVA: no twin graph for this program: %s\n", e.what());
              }
            }
            if (it2 != plans2.end()) {
              int busy2 = 0;
              (void)evah_ctx_busy(it2->second->queues[0]->h, &busy2);
              if (!busy2 || last_plan[&program] == use) use = it2->second.get();
            }
          }
          last_plan[&program] = use;
          return run_plan(*use, inputs);
        }
        plans.erase(it); // same address, different program or shapes: forget the stale plan
        plans2.erase(&program);
        last_plan.erase(&program);
        seen[&program] = 1;
      }
    }
    using clk = std::chrono::steady_clock;
    auto t0 = clk::now();
    // Resident outputs: nothing below waits for the GPU, so consecutive calls queue up behind each
    // other.  Calls alternate between two issue queues; a call with host inputs blocks only in its own
    // uploads, which therefore overlap the previous call's kernels on the other queue (the
    // double-buffering of setInputs, seal_executor.h:264-277, against compute).
    std::shared_ptr<Fork> rq;
    std::vector<evah_ctx *> qh = queue_handles();
    if (subdag) { // member 0 of the device group is the queue this walk issues on
      if (devices.empty() || devices[0] != device)
        throw std::runtime_error("sub-DAG mode: devices[0] must be the context's own device " + std::to_string(device) +
                                 " (inputs, constants and outputs live there)");
      ensure_group(false);
      rq = group->forks[0];
      qh = {group->ctx[0]};
    } else if (resident && library_scheduler && num_queues <= 1 && qh.size() == 1) {
      if (!exec_q[0]) { exec_q[0] = std::make_shared<Fork>(dev); exec_q[1] = std::make_shared<Fork>(dev); }
      rq = exec_q[exec_turn++ & 1];
      qh = {rq->h};
    }
    HipExecutor ex(program, *host, qh, dev.get());
    if (subdag)
      ex.submit = [this](std::vector<evah_op> &ops, std::vector<evah_val> &table, const std::set<uint32_t> &keep) {
        SubDagPlan plan = run_subdag(*group, ops, table, keep);
        last_subdag_plan.clear();
        last_subdag_plan.emplace_back(0u, (uint32_t)plan.prefix.size());
        for (auto &dc : plan.components) last_subdag_plan.emplace_back(dc.first, (uint32_t)dc.second.size());
        last_subdag_plan.emplace_back(0u, (uint32_t)plan.suffix.size());
      };
    // constants (Constant / Encode nodes and arithmetic on them) are evaluated by the first walk
    // of a program and stay resident: later walks only look them up
    ConstCache &cc = const_cache[&program];
    const uint64_t h = program_hash(program);
    if (cc.values.size() != program.size() || cc.hash != h) {
      cc.done = ex.prepare_constants();
      cc.values.assign(program.size(), HipExecutor::RuntimeValue{});
      for (TermId t = 0; t < program.size(); t++)
        if (cc.done[t]) cc.values[t] = ex.value(t);
      cc.hash = h;
    } else {
      for (TermId t = 0; t < program.size(); t++)
        if (cc.done[t]) ex.set_value(t, cc.values[t]);
    }
    ex.set_inputs(inputs);
    auto t1 = clk::now();
    if (subdag || (library_scheduler && num_queues <= 1)) ex.run_library(&cc.done, free_eagerly);
    else run_counted(program, ex, &cc.done);
    auto t2 = clk::now();
    HipValuation out;
    if (resident) {
      const DeviceResident where{dev, rq, nullptr, host->N};
      ex.get_outputs(out, &where);
    } else {
      ex.get_outputs(out);
    }
    auto t3 = clk::now();
    last_timing = {std::chrono::duration<double, std::milli>(t1 - t0).count(), std::chrono::duration<double, std::milli>(t2 - t1).count(),
                   std::chrono::duration<double, std::milli>(t3 - t2).count()};
    return out;
  }

  // wait until everything execute() / encrypt() have enqueued on this context's queues is done
  void synchronize() {
    if (!dev) return;
    chk(evah_ctx_sync(dev->h));
    for (auto &f : forks) chk(evah_ctx_sync(f->h));
    for (auto &f : exec_q) if (f) chk(evah_ctx_sync(f->h));
    if (group) for (evah_ctx *c : group->ctx) chk(evah_ctx_sync(c));
    for (auto *m : {&plans, &plans2}) for (auto &kv : *m) for (auto &f : kv.second->queues) chk(evah_ctx_sync(f->h));
  }
  // Per-launch HIP-event profile by kernel class (evah_profile_*), over every issue queue this context owns: what
  // bench.py's roofline reads when the timed region is execute() itself rather than raw C-ABI calls.
  std::vector<evah_ctx *> all_queues() {
    std::vector<evah_ctx *> q;
    if (!dev) return q;
    q.push_back(dev->h);
    for (auto &f : forks) q.push_back(f->h);
    for (auto &f : exec_q) if (f) q.push_back(f->h);
    for (auto &f : batch_forks) q.push_back(f->h);
    for (auto &f : batch_queues) q.push_back(f->h);
    for (auto *m : {&plans, &plans2}) for (auto &kv : *m) for (auto &f : kv.second->queues) q.push_back(f->h);
    return q;
  }
  void profile(bool on) {
    ensure_device();
    if (on && !exec_q[0]) { exec_q[0] = std::make_shared<Fork>(dev); exec_q[1] = std::make_shared<Fork>(dev); }
    for (evah_ctx *q : all_queues()) chk(evah_profile_enable(q, on ? 1 : 0));
  }
  void profile_reset() { for (evah_ctx *q : all_queues()) chk(evah_profile_reset(q)); }
  std::map<std::string, std::pair<uint64_t, double>> profile_get() {
    std::map<std::string, std::pair<uint64_t, double>> out;
    for (evah_ctx *q : all_queues())
      for (int cls = 0; cls < evah_profile_classes(); cls++) {
        uint64_t n = 0;
        double ms = 0;
        chk(evah_profile_get(q, cls, &n, &ms));
        auto &e = out[evah_profile_class_name(cls)];
        e.first += n;
        e.second += ms;
      }
    return out;
  }
  // ciphertexts up / down, plaintexts up / down, bytes up / down across the host boundary (evah_ctx_transfer_stats)
  std::array<uint64_t, 6> transfer_stats() {
    std::array<uint64_t, 6> st{0, 0, 0, 0, 0, 0};
    if (dev) chk(evah_ctx_transfer_stats(dev->h, st.data()));
    return st;
  }

  // HBM bytes of evaluation keys per limb shard (after a limb-sharded execute()), then of this device's whole keys
  std::vector<uint64_t> key_bytes() {
    std::vector<uint64_t> out;
    if (limb)
      for (size_t s = 0; s < limb->group().size(); s++) {
        uint64_t b = 0;
        if (limb->group().ctx[s]) chk(evah_ctx_key_bytes(limb->group().ctx[s], &b)); // a shard of another process: 0 here
        out.push_back(b);
      }
    uint64_t b = 0;
    if (dev) chk(evah_ctx_key_bytes(dev->h, &b));
    out.push_back(b);
    return out;
  }

  // A batch of independent executions of one program (BASELINE config 4): instances are grouped
  // `batch_chunk` at a time into batched device handles, so each DAG node is one backend call —
  // one launch set — per group instead of per instance.  Results are those of execute() on each
  // valuation, bit for bit.  The reference has no counterpart: it loops SEALPublic::execute.
  // the encrypted part of a program as one evah_execute (EVA_LIBRARY_SCHEDULER=0: the host-side walks)
  bool library_scheduler = std::getenv("EVA_LIBRARY_SCHEDULER") ? std::atoi(std::getenv("EVA_LIBRARY_SCHEDULER")) != 0 : true;
  // r5: 24 (EVA_BATCH_CHUNK).  Once a program's constants stayed resident, config 4 (256 Sobel DAGs, N = 2^14) measured
  // 15.3 ms per call with groups of 32, 14.0 with 24, 14.6 with 16, 14.7 with 12, 17–20 with 8, 16.2 with 48, 18.3 with
  // 64 (profiles/r05_tuning_notes.md section 9): smaller groups shorten the pipeline's fill and drain (the first group's
  // uploads and the last group's downloads overlap nothing), larger ones launch wider kernels
  uint32_t batch_chunk = std::getenv("EVA_BATCH_CHUNK") ? (uint32_t)std::atoi(std::getenv("EVA_BATCH_CHUNK")) : 24;
  // groups in flight in execute_batch: group g is enqueued on queue g mod batch_depth, so the copies of one group
  // overlap the kernels of the others; device memory = batch_depth groups' working sets
  // (r6 sweep, profiles/r06_tuning_notes.md: three queues — config 4 22.0 k -> 24.1 k DAGs/s against four, Harris batch +1 %;
  // two and four to seven queues all measured lower)
  // 0 (the default) = by the valuations of the call: three queues when they are resident, four when the call uploads and
  // downloads host words (config 4: 25.8 k against 23.8 k DAGs/s resident, 17.5 k against 20.0 k with host valuations)
  uint32_t batch_depth = std::getenv("EVA_BATCH_DEPTH") ? (uint32_t)std::atoi(std::getenv("EVA_BATCH_DEPTH")) : 0;
  // groups of (nearly) equal size instead of full groups and a remainder (batch.h; EVA_BATCH_BALANCE=1).  Off by default: r6,
  // seven A/B pairs over two calls — resident valuations Harris batch 2 971 against 2 953 DAGs/s, config 4 24 730 against 24 462
  // (+0.6 % / +1.1 %, inside the noise), but host valuations of config 4 18 943 against 20 171 (-6 %, every pair: two group sizes
  // are two sets of pool blocks and staging copies), and 10.2 k in one full bench run
  bool batch_balance = std::getenv("EVA_BATCH_BALANCE") ? std::atoi(std::getenv("EVA_BATCH_BALANCE")) != 0 : false;
  // smaller groups at both ends of a batch (batch.h; EVA_BATCH_RAMP=1).  Off by default: measured on config 4
  // (profiles/r05_tuning_notes.md) the shorter fill / drain is real — the best calls are the same 16.8 ms — but the odd
  // group sizes make some calls 3-6 ms longer (pool misses), so the median is no better
  bool batch_ramp = std::getenv("EVA_BATCH_RAMP") ? std::atoi(std::getenv("EVA_BATCH_RAMP")) != 0 : false;
  std::vector<HipValuation> execute_batch(Program &program, const std::vector<const HipValuation *> &inputs);

  // "dag" mode (BASELINE config 4): the groups of a batch are dealt over the members
  // of `devices` — group g on member g mod G, batch_depth issue queues per member so a member's copies overlap its
  // kernels — with no data-path exchange: instances are independent.  Same results as execute_batch on one
  // device.  (The driver's scaling curve uses one process per GPU, eva_amd/dist.py; this is the same
  // partition inside one execute_batch call.)
  std::vector<HipValuation> execute_batch_multi(Program &program, const std::vector<const HipValuation *> &inputs);

  evah_ctx *device_ctx() {
    ensure_device();
    return dev->h;
  }

  ~HipPublic() {
    const_cache.clear();
    multi_const_cache.clear();
    plans2.clear();
    last_plan.clear();
    plans.clear();
    batch_forks.clear();
    batch_queues.clear();
    limb.reset();
    limb_const.clear();
    group.reset();
    exec_q[0].reset();
    exec_q[1].reset();
    forks.clear(); // queues go before the root context (each fork also holds it)
    dev.reset();
  }
  size_t graph_plan_count() const { return plans.size() + plans2.size(); }
  void drop_graphs() { plans.clear(); plans2.clear(); last_plan.clear(); no_twin.clear(); seen.clear(); no_graph.clear(); const_cache.clear(); multi_const_cache.clear(); }

private:
  std::shared_ptr<DeviceCtx> dev; // == holder->dev once a device is in use
  std::vector<std::shared_ptr<Fork>> forks;
  std::vector<std::shared_ptr<Fork>> batch_forks; // the further issue queues of execute_batch

  std::shared_ptr<Fork> exec_q[2];  // the two issue queues resident execute() calls alternate between
  unsigned exec_turn = 0;
  std::vector<std::shared_ptr<Fork>> batch_queues; // "dag" mode: batch_depth issue queues per member
  std::unique_ptr<DeviceGroup> group;        // sub-DAG split: members of `devices`
  std::vector<int> group_ids;
  std::unique_ptr<LimbShardEvaluator> limb;  // limb sharding: one shard context per member
  std::vector<int> limb_ids;
  struct LimbConst { uint64_t hash = 0; std::unordered_map<TermId, ShardedValue> plain; };
  std::unordered_map<const Program *, LimbConst> limb_const; // encoded plaintexts of a program, dealt over the shards
  void upload_eval_keys(evah_ctx *c) {
    chk(evah_key_upload(c, EVAH_KEY_RELIN, 0, relin.n_digits, (const uint64_t *)relin.data.data()));
    for (auto &kv : galois)
      chk(evah_key_upload(c, EVAH_KEY_GALOIS, kv.first, kv.second.n_digits, (const uint64_t *)kv.second.data.data()));
  }
  void check_devices() const {
    int n = 0;
    chk(evah_device_count(&n));
    for (int d : devices)
      if (d < 0 || physical_device(d) >= n)
        throw std::runtime_error("device " + std::to_string(physical_device(d)) + " requested, " + std::to_string(n) + " visible");
  }
  void ensure_group(bool) {
    if (group && group_ids == devices) return;
    check_devices();
    multi_const_cache.clear(); // constants of the old group's members
    batch_queues.clear();
    group.reset();
    group = std::make_unique<DeviceGroup>(make_device_group(devices, dev, device, *host, [this](evah_ctx *c) { upload_eval_keys(c); }, true));
    group_ids = devices;
  }

  // SEALPublic::execute over limb-sharded values: serial forwardPass, SEALExecutor's dispatch per node
  // (seal_executor.h:279-404) on a LimbShardEvaluator.  Values come in and go out as host words (a
  // sharded value has no single device handle); constants are encoded on the host once per program.
  HipValuation execute_limb(Program &program, const HipValuation &inputs);
  // does term `t` depend on term `src`?
  static bool depends_on(const Program &p, TermId t, TermId src) {
    if (t == src) return true;
    for (TermId o : p.at(t).operands)
      if (depends_on(p, o, src)) return true;
    return false;
  }

  // bytes of the inputs that are resident on this context's device (and nowhere on the host)
  size_t resident_bytes(const HipValuation &inputs) const {
    size_t b = 0;
    for (auto &kv : inputs.values)
      if (auto *c = std::get_if<HostCipher>(&kv.second))
        if (c->dev && c->dev->root == dev) b += sizeof(u64) * (size_t)c->size * c->limbs * host->N;
    return b;
  }

  // A captured execute(): its own queues (pools are exclusive to the graph), persistent input
  // slots and constant plaintexts, the outputs' handles, the instantiated hipGraph.
  static uint64_t program_hash(const Program &p) {
    uint64_t h = 1469598103934665603ull;
    auto mix = [&](uint64_t v) { h = (h ^ v) * 1099511628211ull; };
    for (TermId t : p.topo_order()) {
      const Term &x = p.at(t);
      mix(t); mix((uint64_t)x.op); mix((uint64_t)(uint32_t)x.rotation); mix(x.rescale_divisor); mix(x.encode_scale); mix(x.encode_level);
      for (TermId o : x.operands) mix(o);
      if (x.constant) for (double v : x.constant->values) { uint64_t b; std::memcpy(&b, &v, 8); mix(b); }
    }
    return h;
  }
  struct GraphPlan {
    size_t program_size = 0;
    uint64_t hash = 0;
    std::vector<std::shared_ptr<Fork>> queues;
    std::unordered_map<std::string, std::shared_ptr<CtHandle>> in_ct;
    std::unordered_map<std::string, std::shared_ptr<PtHandle>> in_pt;
    std::vector<HipExecutor::RuntimeValue> persistent; // constants
    std::unordered_map<std::string, HipExecutor::RuntimeValue> outputs;
    evah_graph *graph = nullptr;
    ~GraphPlan() {
      outputs.clear();
      persistent.clear();
      in_ct.clear();
      in_pt.clear();
      evah_graph_free(graph);
      queues.clear();
    }
    bool matches(const Program &p, const HipValuation &inputs) const {
      if (p.size() != program_size || program_hash(p) != hash || inputs.values.size() != in_ct.size() + in_pt.size()) return false;
      for (auto &kv : inputs.values) {
        if (auto *c = std::get_if<HostCipher>(&kv.second)) {
          auto it = in_ct.find(kv.first);
          if (it == in_ct.end()) return false;
          uint32_t s, l;
          double sc;
          if (evah_ct_info(it->second->h, &s, &l, &sc) || s != c->size || l != c->limbs || sc != c->scale) return false;
        } else if (auto *pl = std::get_if<HostPlain>(&kv.second)) {
          auto it = in_pt.find(kv.first);
          if (it == in_pt.end()) return false;
          uint32_t l;
          double sc;
          if (evah_pt_info(it->second->h, &l, &sc) || l != pl->limbs || sc != pl->scale) return false;
        } else return false;
      }
      return true;
    }
  };
  struct ConstCache {
    uint64_t hash = 0;
    std::vector<char> done;
    std::vector<HipExecutor::RuntimeValue> values;
  };
  std::unordered_map<const Program *, ConstCache> const_cache;
  std::unordered_map<const Program *, std::vector<ConstCache>> multi_const_cache; // "dag" mode: per member of the device group
  std::unordered_map<const Program *, std::unique_ptr<GraphPlan>> plans, plans2; // plans2: the twin of a plan found busy (execute())
  std::unordered_map<const Program *, GraphPlan *> last_plan;
  std::set<const Program *> no_twin;
 public:
  bool twin_plans = std::getenv("EVA_GRAPH_TWIN") ? std::atoi(std::getenv("EVA_GRAPH_TWIN")) != 0 : true;
 private:
  std::unordered_map<const Program *, int> seen;
  std::set<const Program *> no_graph; // programs whose capture failed once: always walked eagerly

  bool graphs_enabled() const {
    if (const char *e = std::getenv("EVA_GRAPH")) return std::atoi(e) != 0;
    return use_graphs;
  }
  static bool graphable(const Program &p, const HipValuation &inputs) {
    for (auto &kv : inputs.values)
      if (std::holds_alternative<std::vector<double>>(kv.second)) return false; // Raw inputs feed host-side encodes
    for (auto &kv : p.inputs())
      if (p.at(kv.second).type_attr == Type::Raw) return false;
    return true;
  }

  std::unique_ptr<GraphPlan> build_plan(Program &program, const HipValuation &inputs);

  HipValuation run_plan(GraphPlan &plan, const HipValuation &inputs);

  std::vector<evah_ctx *> queue_handles() {
    int want = num_queues;
    if (const char *e = std::getenv("EVA_NUM_STREAMS")) want = std::atoi(e);
    if (want < 1) want = 1;
    while ((int)forks.size() + 1 < want) forks.push_back(std::make_shared<Fork>(dev));
    std::vector<evah_ctx *> q{dev->h};
    for (int i = 0; i + 1 < want; i++) q.push_back(forks[i]->h);
    return q;
  }
  // EVA_DEVICE_CLIENT=0 keeps encrypt on the host; without a HIP device the host path is the only one
  // (encrypt, unlike execute(), is client-side work the reference also does on the CPU)
  bool client_on_device();
  int client_device = -1;
  bool pk_uploaded = false;
  // same bound as HipExecutor::device_encodable: every rounded coefficient below 2^62 and inside the modulus
  bool device_encodable(const std::vector<double> &in, double scale, uint32_t limbs) const;
  // coeff_pt: the host encoder's coefficient-form plaintext, or (null) values: the slot values for the device encoder
  HostCipher encrypt_on_device(const HostPlain *coeff_pt, const std::vector<double> *values, double scale, uint32_t limbs, SecureRng &rng);

  // eval_keys = false: encryption and limb-sharded execution (whose shards hold their own rows of the
  // keys) do not need the whole evaluation keys in this device's memory
  bool eval_keys_uploaded = false;
  void ensure_device(bool eval_keys = true) {
    if (!dev) {
      // a one-member `devices` list names the device of this context (several members: member 0 is checked by the modes)
      if (!holder->dev && devices.size() == 1) device = physical_device(devices[0]);
      if (!holder->dev) holder->dev = std::make_shared<DeviceCtx>(host->N, host->primes, device);
      dev = holder->dev; // may have been created by the secret half of the key pair (decrypt first)
    }
    if (eval_keys && !eval_keys_uploaded) {
      upload_eval_keys(dev->h);
      eval_keys_uploaded = true;
    }
  }
};

} // namespace evahost
#include "plans.h"
#include "batch.h"
#include "limb_exec.h"
#include "client.h"
