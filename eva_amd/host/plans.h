// plans.h — hipGraph plans of repeated execute() calls (HipPublic::build_plan / run_plan): the second call of a program
// with the same input shapes is captured, later calls replay the graph (slot refill -> replay -> output copy on one queue).
// Replaces the per-call DAG walk of SEALPublic::execute (seal.cpp:104-122).  Included by public_ctx.h.
#pragma once

namespace evahost {

inline std::unique_ptr<HipPublic::GraphPlan> HipPublic::build_plan(Program &program, const HipValuation &inputs) {
  auto plan = std::make_unique<GraphPlan>();
  plan->program_size = program.size();
  plan->hash = program_hash(program);
  // One queue: multi-branch captures are both slow to launch and unstable to instantiate on
  // the ROCm 7.2 runtime (recursion blow-up in hipStreamEndCapture on reconvergent DAGs); a
  // linear graph replays with ~10 us of host time.
  const int want = 1;
  for (int i = 0; i < want; i++) plan->queues.push_back(std::make_shared<Fork>(dev));
  std::vector<evah_ctx *> q;
  for (auto &f : plan->queues) q.push_back(f->h);
  evah_ctx *q0 = q[0];
  HipExecutor ex(program, *host, q);
  // persistent input slots
  for (auto &kv : inputs.values) {
    TermId t = program.input(kv.first);
    if (auto *c = std::get_if<HostCipher>(&kv.second)) {
      ex.check_shape(kv.first, *c);
      evah_ct *h = nullptr;
      if (c->dev && c->dev->root == dev) chk(evah_ct_copy(q0, c->dev->h->h, &h)); // the slot is the graph's own buffer
      else chk(evah_ct_upload(q0, c->size, c->limbs, c->scale, (const uint64_t *)words(*c).data(), &h));
      auto sp = std::make_shared<CtHandle>(q0, h);
      plan->in_ct[kv.first] = sp;
      ex.set_value(t, sp);
    } else {
      auto &pl = std::get<HostPlain>(kv.second);
      ex.check_shape(kv.first, pl);
      evah_pt *h = nullptr;
      chk(evah_pt_upload(q0, pl.limbs, pl.scale, (const uint64_t *)pl.data.data(), &h));
      auto sp = std::make_shared<PtHandle>(q0, h);
      plan->in_pt[kv.first] = sp;
      ex.set_value(t, sp);
    }
  }
  // constants: encoded once, resident for the life of the plan
  std::vector<char> done = ex.prepare_constants();
  for (TermId t = 0; t < program.size(); t++)
    if (done[t]) plan->persistent.push_back(ex.value(t));
  chk(evah_ctx_sync(q0));
  // capture the walk
  chk(evah_capture_begin(q0, q.data() + 1, (uint32_t)q.size() - 1));
  try {
    if (library_scheduler) ex.run_library(&done, true);
    else run_counted(program, ex, &done);
    for (auto &kv : program.outputs()) plan->outputs[kv.first] = ex.value(kv.second);
    // every other value of the walk goes back to the queues' pools BEFORE the capture ends: the graph takes
    // the pools' free blocks with it (evah_capture_end), so that nothing allocated later aliases a temporary
    ex.drop_values();
  } catch (...) {
    evah_graph *g = nullptr;
    (void)evah_capture_end(q0, q.data() + 1, (uint32_t)q.size() - 1, &g);
    evah_graph_free(g);
    throw;
  }
  chk(evah_capture_end(q0, q.data() + 1, (uint32_t)q.size() - 1, &plan->graph));
  return plan;
}

inline HipValuation HipPublic::run_plan(HipPublic::GraphPlan &plan, const HipValuation &inputs) {
  using clk = std::chrono::steady_clock;
  auto t0 = clk::now();
  evah_ctx *q0 = plan.queues[0]->h;
  // Slot refills, the replay and the copies of its outputs are all enqueued on the plan's own queue: one
  // in-order stream, no cross-queue waits (those cost 10-20 us each against a 5 us kernel at N = 2^13).
  for (auto &kv : inputs.values) {
    // matches() compared the declared shapes with the slots; the data length must agree as well
    if (auto *c = std::get_if<HostCipher>(&kv.second)) {
      if (c->dev && c->dev->root == dev) { // resident: refill the slot device to device
        chk(evah_ct_assign(q0, plan.in_ct.at(kv.first)->h, c->dev->h->h));
        continue;
      }
      const CipherWords &w = words(*c);
      if (w.size() != (size_t)c->size * c->limbs * host->N) throw std::runtime_error("input " + kv.first + ": ciphertext shape does not match its data");
      chk(evah_ct_write(q0, plan.in_ct.at(kv.first)->h, (const uint64_t *)w.data()));
    } else {
      auto &pl = std::get<HostPlain>(kv.second);
      if (pl.data.size() != (size_t)pl.limbs * host->N) throw std::runtime_error("input " + kv.first + ": plaintext shape does not match its data");
      chk(evah_pt_write(q0, plan.in_pt.at(kv.first)->h, (const uint64_t *)pl.data.data()));
    }
  }
  auto t1 = clk::now();
  chk(evah_graph_launch(q0, plan.graph));
  auto t2 = clk::now();
  HipValuation out;
  for (auto &kv : plan.outputs) {
    if (auto *c = std::get_if<std::shared_ptr<CtHandle>>(&kv.second)) {
      HostCipher hc;
      chk(evah_ct_info((*c)->h, &hc.size, &hc.limbs, &hc.scale));
      if (resident) { // the graph owns its output buffers: hand out a device copy, made right behind the replay
        evah_ct *copy = nullptr;
        chk(evah_ct_copy(q0, (*c)->h, &copy));
        hc.dev = std::make_shared<DeviceResident>(DeviceResident{dev, plan.queues[0], std::make_shared<CtHandle>(q0, copy), host->N});
        out.values[kv.first] = std::move(hc);
        continue;
      }
      hc.data.resize((size_t)hc.size * hc.limbs * host->N);
      hc.words_checked = true;
      chk(evah_ct_download(q0, (*c)->h, (uint64_t *)hc.data.data()));
      out.values[kv.first] = std::move(hc);
    } else if (auto *p = std::get_if<std::shared_ptr<PtHandle>>(&kv.second)) {
      HostPlain hp;
      chk(evah_pt_info((*p)->h, &hp.limbs, &hp.scale));
      hp.data.resize((size_t)hp.limbs * host->N);
      chk(evah_pt_download(q0, (*p)->h, (uint64_t *)hp.data.data()));
      out.values[kv.first] = std::move(hp);
    } else if (auto *r = std::get_if<std::vector<double>>(&kv.second)) {
      out.values[kv.first] = *r;
    } else {
      throw std::runtime_error("Output " + kv.first + " was not computed");
    }
  }
  auto t3 = clk::now();
  last_timing = {std::chrono::duration<double, std::milli>(t1 - t0).count(), std::chrono::duration<double, std::milli>(t2 - t1).count(),
                 std::chrono::duration<double, std::milli>(t3 - t2).count()};
  return out;
}

} // namespace evahost
