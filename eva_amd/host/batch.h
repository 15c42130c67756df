// batch.h — HipPublic::execute_batch: many input valuations of ONE program (BASELINE config 4) as batched device
// handles, groups of batch_chunk instances pipelined over batch_depth issue queues (uploads and downloads of one group
// overlap the kernels of the others); execute_batch_multi deals the groups over the members of a device group
// (shard_mode = "dag").  Included by public_ctx.h.
#pragma once

namespace evahost {

inline std::vector<HipValuation> HipPublic::execute_batch(Program &program, const std::vector<const HipValuation *> &inputs) {
  ensure_device();
  if (batch_chunk < 1 || batch_chunk > 64) throw std::runtime_error("batch_chunk must be 1..64");
  std::vector<HipValuation> all(inputs.size());
  // Groups rotate over batch_depth issue queues (r6 default: three on resident valuations — +9 % over four and +11 % over two on
  // config 4 — four when the call moves host words over PCIe) and nothing waits in between: each group's uploads,
  // launches and downloads are enqueued in queue order (evah_ct_*_instances_async), so the copies
  // of one group overlap the kernels of the other and the host never idles the device.  Device
  // memory stays at two groups' working sets (the pools recycle in queue order); the inputs belong
  // to the caller and the outputs are allocated up front, so both outlive the final synchronisation.
  if (devices.size() > 1 && shard_mode == "dag") return execute_batch_multi(program, inputs);
  uint32_t depth = batch_depth;
  if (depth == 0) { // auto: do the call's ciphertexts live in HBM already?
    bool in_hbm = resident && !inputs.empty();
    if (in_hbm) {
      in_hbm = false;
      for (auto &kv : inputs[0]->values)
        if (auto *c0 = std::get_if<HostCipher>(&kv.second)) { in_hbm = (bool)c0->dev; break; }
    }
    depth = in_hbm ? 3 : 4;
  }
  if (depth < 2 || depth > 8) throw std::runtime_error("batch_depth must be 2..8 (0: chosen by the call's valuations)");
  while (batch_forks.size() + 1 < depth) batch_forks.push_back(std::make_shared<Fork>(dev));
  std::vector<evah_ctx *> qs{dev->h};
  for (uint32_t i = 0; i + 1 < depth; i++) qs.push_back(batch_forks[i]->h);
  const size_t Q = qs.size();
  // constants (Constant / Encode nodes and raw arithmetic on them) are evaluated once per PROGRAM — by the first group
  // of the first call, or by an earlier execute() of the same program (const_cache) — and shared by all groups of all
  // calls.  (r5: they used to be encoded by the first group of every call and released at its end: ~20 encodes in front
  // of the first group's kernels and, at the release, an event wait per plaintext per reading queue — 0.5 ms of a 17 ms call)
  // (the cached plaintexts were encoded on the queue of whichever group ran first and are read from every batch queue in
  // later calls: nothing waits explicitly at the lookup — each library call acquires the buffers it reads, Buffer::owner /
  // readers in csrc/internal.hip.h, which is what orders a reading queue behind the encoding one)
  ConstCache &cc = const_cache[&program];
  const uint64_t prog_hash = program_hash(program);
  bool fresh = cc.values.size() != program.size() || cc.hash != prog_hash;
  std::vector<char> &done = cc.done;
  std::vector<HipExecutor::RuntimeValue> &consts = cc.values;
  auto finish = [&]() {
    int rc = 0;
    for (evah_ctx *q : qs) rc |= evah_ctx_sync(q);
    if (rc) throw_backend();
  };
  size_t g = 0;
  const bool bounded = std::getenv("EVA_BATCH_BOUNDED") ? std::atoi(std::getenv("EVA_BATCH_BOUNDED")) != 0 : false;
  // Group sizes.  The call includes the uploads of its first group and the downloads of its last, which nothing overlaps:
  // with batch_ramp the first and the last batch_chunk instances go as a quarter-sized and a three-quarter-sized group
  // (8, 24, 32, ..., 32, 24, 8 for 256 instances), so the pipeline fills and drains on a quarter of a group's copies.
  std::vector<size_t> sizes;
  {
    const size_t n = inputs.size(), c = batch_chunk, q = c / 4;
    if (batch_ramp && q >= 1 && n >= 4 * c) {
      sizes = {q, c - q};
      for (size_t left = n - 2 * c; left > 0; left -= std::min(left, c)) sizes.push_back(std::min(left, c));
      sizes.push_back(c - q);
      sizes.push_back(q);
    } else if (batch_balance) {
      // the same number of groups, of (nearly) equal size: 64 instances in groups of at most 12 go as 11 11 11 11 10 10 instead
      // of 12 12 12 12 12 4 — the queues the groups rotate over then finish together instead of one running dry early
      const size_t G = (n + c - 1) / c, base = n / G, extra = n % G;
      for (size_t i = 0; i < G; i++) sizes.push_back(base + (i < extra ? 1 : 0));
    } else {
      for (size_t left = n; left > 0; left -= std::min(left, c)) sizes.push_back(std::min(left, c));
    }
  }
  const auto t_begin = std::chrono::steady_clock::now();
  try {
    for (size_t i0 = 0; g < sizes.size(); i0 += sizes[g], g++) {
      const size_t n = sizes[g];
      std::vector<const HipValuation *> chunk(inputs.begin() + i0, inputs.begin() + i0 + n);
      if (bounded && g >= Q) chk(evah_ctx_sync(qs[g % Q])); // group g-Q (same queue) has left the device
      HipExecutor ex(program, *host, std::vector<evah_ctx *>{qs[g % Q]}, dev.get());
      if (fresh) {
        done = ex.prepare_constants();
        consts.assign(program.size(), HipExecutor::RuntimeValue{});
        for (TermId t = 0; t < program.size(); t++)
          if (done[t]) consts[t] = ex.value(t);
        cc.hash = prog_hash;
        fresh = false;
      } else {
        for (TermId t = 0; t < program.size(); t++)
          if (done[t]) ex.set_value(t, consts[t]);
      }
      ex.set_inputs_batch(chunk, true);
      if (library_scheduler) ex.run_library(&done, true);
      else run_counted(program, ex, &done);
      // r6: resident valuations (`resident`, the default) — inputs that encrypt() / execute() left in HBM were stacked device
      // to device, and the outputs leave as handles (views of the group's batched output): no ciphertext crosses PCIe in
      // the call.  Host valuations (resident = false, or inputs that hold host words only): uploads and downloads as before.
      if (resident && ex.batch_inputs_resident) {
        const DeviceResident res{dev, g % Q ? batch_forks[g % Q - 1] : nullptr, nullptr, host->N};
        ex.get_outputs_batch(all.data() + i0, n, true, &res);
      } else {
        ex.get_outputs_batch(all.data() + i0, n, true);
      }
    }
  } catch (...) {
    for (evah_ctx *q : qs) (void)evah_ctx_sync(q); // copies in flight still target `all` and the caller's inputs
    throw;
  }
  const auto t_issued = std::chrono::steady_clock::now();
  finish();
  if (std::getenv("EVA_BATCH_TIMING")) // how much of the call the host spends issuing (the rest it waits for the device)
    std::fprintf(stderr, "EVA: execute_batch issued %zu groups in %.3f ms, done after %.3f ms\n", g,
                 std::chrono::duration<double, std::milli>(t_issued - t_begin).count(),
                 std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count());
  return all;
}

// "dag" mode (BASELINE config 4): the groups of a batch are dealt over the members
// of `devices` — group g on member g mod G, batch_depth issue queues per member so a member's copies overlap its
// kernels — with no data-path exchange: instances are independent.  Same results as execute_batch on one
// device.  (The driver's scaling curve uses one process per GPU, eva_amd/dist.py; this is the same
// partition inside one execute_batch call.)
inline std::vector<HipValuation> HipPublic::execute_batch_multi(Program &program, const std::vector<const HipValuation *> &inputs) {
  if (batch_chunk < 1 || batch_chunk > 64) throw std::runtime_error("batch_chunk must be 1..64");
  ensure_group(false);
  const size_t G = group->size();
  if (batch_depth != 0 && (batch_depth < 2 || batch_depth > 8)) throw std::runtime_error("batch_depth must be 2..8 (0: chosen by the call's valuations)");
  const size_t D = batch_depth ? batch_depth : 4; // (dag mode deals host valuations over its members)
  if (batch_queues.size() != D * G) {
    batch_queues.clear();
    for (size_t m = 0; m < G; m++)
      for (size_t k = 0; k < D; k++) batch_queues.push_back(std::make_shared<Fork>(group->roots[m]));
  }
  std::vector<HipValuation> all(inputs.size());
  // per member: the program's constants, encoded once per program on that member's device state (as execute_batch does)
  std::vector<ConstCache> &mc = multi_const_cache[&program];
  const uint64_t prog_hash = program_hash(program);
  if (mc.size() != G) mc.assign(G, ConstCache{});
  for (ConstCache &c1 : mc)
    if (c1.hash != prog_hash || c1.values.size() != program.size()) c1 = ConstCache{};
  std::vector<size_t> turn(G, 0);
  auto sync_all = [&]() {
    int rc = 0;
    for (auto &f : batch_queues) rc |= evah_ctx_sync(f->h);
    return rc;
  };
  try {
    size_t g = 0;
    for (size_t i0 = 0; i0 < inputs.size(); i0 += batch_chunk, g++) {
      const size_t n = std::min<size_t>(batch_chunk, inputs.size() - i0), m = g % G;
      std::vector<const HipValuation *> chunk(inputs.begin() + i0, inputs.begin() + i0 + n);
      evah_ctx *q = batch_queues[D * m + (turn[m]++ % D)]->h;
      HipExecutor ex(program, *host, std::vector<evah_ctx *>{q}, group->roots[m].get());
      ConstCache &cm = mc[m];
      if (cm.done.empty()) { // the member's constants: encoded once per program, by its first group
        cm.done = ex.prepare_constants();
        cm.values.assign(program.size(), HipExecutor::RuntimeValue{});
        for (TermId t = 0; t < program.size(); t++)
          if (cm.done[t]) cm.values[t] = ex.value(t);
        cm.hash = prog_hash;
      } else {
        for (TermId t = 0; t < program.size(); t++)
          if (cm.done[t]) ex.set_value(t, cm.values[t]);
      }
      ex.set_inputs_batch(chunk, true);
      ex.run_library(&cm.done, true);
      ex.get_outputs_batch(all.data() + i0, n, true);
    }
  } catch (...) {
    (void)sync_all(); // copies in flight still target `all` and the caller's inputs
    throw;
  }
  if (sync_all()) throw_backend();
  return all;
}

} // namespace evahost
