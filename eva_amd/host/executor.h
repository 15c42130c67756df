// executor.h — the host half of execute(): walks a compiled DAG and issues one backend call per
// node through the C-ABI of libeva_hip.so (include/eva_hip.h).
//
//   HipExecutor        ~ SEALExecutor          (/root/reference/eva/seal/seal_executor.h:31-438)
//   run_serial / run_counted ~ ProgramTraversal / MulticoreProgramTraversal::forwardPass
//                        (/root/reference/eva/common/program_traversal.h:36-93,
//                         multicore_program_traversal.h:24-83)
//   HipPublic / HipSecret / HipValuation / generate_keys ~ SEALPublic / SEALSecret /
//                        SEALValuation / generateKeys (/root/reference/eva/seal/seal.{h,cpp})
//   ReferenceExecutor / evaluate ~ /root/reference/eva/common/reference_executor.cpp, eva.cpp:11-21
//
// There is no CPU evaluator behind execute(): every Cipher/Plain node goes to the GPU library and
// a missing device is an error.
//
// r04: this file is the DISPATCHER only.  The pieces that used to follow it live in
//   values.h       valuations, handles, device contexts (included first)
//   multi_device.h device groups, sub-DAG planner / runner, LimbShardEvaluator
//   public_ctx.h   HipPublic (the public half of a key pair: execute(), residency, queues, multi-device selection)
//   plans.h        hipGraph plans of repeated execute() calls         (HipPublic::build_plan / run_plan)
//   batch.h        execute_batch: the pipeline over batched handles    (HipPublic::execute_batch[_multi])
//   limb_exec.h    execute() over limb-sharded values                  (HipPublic::execute_limb)
//   client.h       encrypt (host / device), HipSecret, generate_keys
// and are pulled in at the end, so `#include "executor.h"` still gives the whole host side.
#pragma once
#include "values.h"

namespace evahost {

// Per-term dispatcher: Term -> one libeva_hip call (SEALExecutor::operator(), :279-404)
class HipExecutor {
public:
  // An Encode node is materialised lazily on the queue of its first consumer.
  struct LazyPlain {
    TermId src;
    uint32_t scale_bits, level;
  };
  // A Relinearize whose only consumer is a Rescale is evaluated together with it
  // (evah_relinearize_rescale: identical result, fewer transforms).
  struct LazyRelin {
    TermId src;
  };
  using RuntimeValue = std::variant<std::monostate, std::shared_ptr<CtHandle>, std::shared_ptr<PtHandle>, std::vector<double>, LazyPlain, LazyRelin>;

  // queues: issue queues (HIP streams) of one device — queues[0] is the root context, the rest
  // are its forks.  Independent DAG nodes are spread over them (the GPU counterpart of the
  // reference's Galois worker threads, multicore_program_traversal.h:55-78); ordering between
  // queues is enforced inside libeva_hip.so per buffer.
  // root: the device state the queues belong to — a resident input of the same state is used by handle
  HipExecutor(Program &g, const HostContext &hc, std::vector<evah_ctx *> qs, const DeviceCtx *root_ = nullptr)
      : program(g), host(hc), queues(std::move(qs)), ctx(queues.at(0)), objects(g.size()), queue_of(g.size(), 0), root(root_) {
    if (program.vec_size() > host.N / 2) throw std::runtime_error("Vector size cannot be larger than slot count");
  }

  // values may come from files or from Python (_set_cipher): before any upload the declared shape
  // has to agree with the data length and the context, or the copy would read past the host buffer
  void check_shape(const std::string &name, const HostCipher &c) const {
    if (c.size < 1 || c.size > 3 || c.limbs < 1 || c.limbs > host.k - 1 ||
        (!resident_only(c) && c.data.size() != (size_t)c.size * c.limbs * host.N) ||
        (c.dev && c.dev->N != host.N)) // a resident value of another key pair: its download would have the wrong length
      throw std::runtime_error("input " + name + ": ciphertext shape does not match its data or the encryption parameters");
    if (!c.words_checked && !c.data.empty()) { // once per value: files and Python arrays are untrusted
      check_words(name, c.data.data(), c.size, c.limbs);
      c.words_checked = true;
    }
  }
  // the kernels' lazy-reduction bounds assume canonical residues: a word >= its prime would give
  // silently wrong results, so it is an error at the boundary
  void check_words(const std::string &name, const u64 *w, uint32_t polys, uint32_t limbs) const {
    for (uint32_t p = 0; p < polys; p++)
      for (uint32_t i = 0; i < limbs; i++) {
        const u64 q = host.primes[i];
        const u64 *row = w + ((size_t)p * limbs + i) * host.N;
        for (uint32_t j = 0; j < host.N; j++)
          if (row[j] >= q) throw std::runtime_error("input " + name + ": a word is not reduced modulo its prime");
      }
  }
  // release every runtime value this executor still holds (handles shared with a plan / a valuation stay alive there)
  void drop_values() {
    deferred_free.clear();
    objects.assign(objects.size(), RuntimeValue{});
  }
  // a value resident on this executor's device state: its handle, else null
  std::shared_ptr<CtHandle> resident_handle(const HostCipher &c) const {
    if (!c.dev || !root || c.dev->root.get() != root) return nullptr;
    uint32_t s = 0, l = 0;
    if (evah_ct_info(c.dev->h->h, &s, &l, nullptr) || s != c.size || l != c.limbs) return nullptr;
    return c.dev->h;
  }
  void check_shape(const std::string &name, const HostPlain &p) const {
    if (p.limbs < 1 || p.limbs > host.k - 1 || p.data.size() != (size_t)p.limbs * host.N)
      throw std::runtime_error("input " + name + ": plaintext shape does not match its data or the encryption parameters");
    if (!p.words_checked) {
      check_words(name, p.data.data(), 1, p.limbs);
      p.words_checked = true;
    }
  }

  // seal_executor.h:264-277 (the reference deep-copies; here inputs are uploaded to HBM)
  void set_inputs(const HipValuation &inputs) {
    for (auto &kv : inputs.values) {
      TermId t = program.input(kv.first);
      if (auto *c = std::get_if<HostCipher>(&kv.second)) {
        check_shape(kv.first, *c);
        if (auto rh = resident_handle(*c)) { // already in HBM: passed by handle (ordering per buffer is the library's)
          objects[t] = std::move(rh);
          resident_inputs.push_back(c->dev);
          continue;
        }
        const CipherWords &w = words(*c); // a value of another device state comes through the host
        evah_ct *h = nullptr;
        chk(evah_ct_upload(ctx, c->size, c->limbs, c->scale, (const uint64_t *)w.data(), &h));
        objects[t] = std::make_shared<CtHandle>(ctx, h);
      } else if (auto *p = std::get_if<HostPlain>(&kv.second)) {
        check_shape(kv.first, *p);
        evah_pt *h = nullptr;
        chk(evah_pt_upload(ctx, p->limbs, p->scale, (const uint64_t *)p->data.data(), &h));
        objects[t] = std::make_shared<PtHandle>(ctx, h);
      } else {
        const auto &raw = std::get<std::vector<double>>(kv.second);
        std::vector<double> v;
        ConstantValue{raw}.expand_to(v, program.vec_size());
        objects[t] = std::move(v);
      }
    }
  }

  // A batch of independent input valuations for ONE program (BASELINE config 4): every encrypted
  // input becomes one batched device handle ([B][size][limbs][N]); from there each node is a
  // single backend call that covers all B instances.  Plaintext / raw inputs are shared by the
  // batch, so they have to be identical across the instances.
  // async: the uploads are only enqueued (evah_ct_upload_instances_async); the caller keeps the
  // valuations alive and synchronises the queue before touching them
  void set_inputs_batch(const std::vector<const HipValuation *> &batch, bool async = false) {
    const uint32_t B = (uint32_t)batch.size();
    for (auto &kv : batch[0]->values) {
      TermId t = program.input(kv.first);
      for (const HipValuation *v : batch)
        if (!v->values.count(kv.first) || v->values.at(kv.first).index() != kv.second.index())
          throw std::runtime_error("execute_batch: input " + kv.first + " is not present with one type in every valuation");
      if (auto *c0 = std::get_if<HostCipher>(&kv.second)) {
        std::vector<const uint64_t *> ptrs(B);
        for (uint32_t b = 0; b < B; b++) {
          const auto &c = std::get<HostCipher>(batch[b]->values.at(kv.first));
          check_shape(kv.first, c);
          if (c.size != c0->size || c.limbs != c0->limbs || c.scale != c0->scale)
            throw std::runtime_error("execute_batch: input " + kv.first + " differs in shape or scale across the batch");
        }
        // r6: instances that are already in HBM (valuations left there by encrypt() or an earlier execute()) are stacked
        // device to device (evah_ct_stack: one strided copy per instance on this queue, ordered per buffer by the
        // library) — no PCIe in the call; otherwise the batched handle is assembled from host words
        std::vector<std::shared_ptr<CtHandle>> rh(B);
        bool all_resident = true;
        for (uint32_t b = 0; b < B && all_resident; b++) {
          rh[b] = resident_handle(std::get<HostCipher>(batch[b]->values.at(kv.first)));
          all_resident = rh[b] != nullptr;
        }
        evah_ct *h = nullptr;
        if (all_resident) {
          std::vector<const evah_ct *> hs(B);
          for (uint32_t b = 0; b < B; b++) hs[b] = rh[b]->h;
          chk(evah_ct_stack(ctx, hs.data(), B, &h));
          batch_inputs_resident = true;
        } else {
          for (uint32_t b = 0; b < B; b++) ptrs[b] = (const uint64_t *)words(std::get<HostCipher>(batch[b]->values.at(kv.first))).data();
          chk((async ? evah_ct_upload_instances_async : evah_ct_upload_instances)(ctx, B, c0->size, c0->limbs, c0->scale, ptrs.data(), &h));
        }
        objects[t] = std::make_shared<CtHandle>(ctx, h);
      } else if (auto *p = std::get_if<HostPlain>(&kv.second)) {
        check_shape(kv.first, *p);
        for (const HipValuation *v : batch) {
          const auto &q = std::get<HostPlain>(v->values.at(kv.first));
          if (q.limbs != p->limbs || q.scale != p->scale || q.data != p->data)
            throw std::runtime_error("execute_batch: plaintext input " + kv.first + " must be the same for every instance");
        }
        evah_pt *h = nullptr;
        chk(evah_pt_upload(ctx, p->limbs, p->scale, (const uint64_t *)p->data.data(), &h));
        objects[t] = std::make_shared<PtHandle>(ctx, h);
      } else {
        const auto &raw = std::get<std::vector<double>>(kv.second);
        for (const HipValuation *v : batch)
          if (std::get<std::vector<double>>(v->values.at(kv.first)) != raw)
            throw std::runtime_error("execute_batch: unencrypted input " + kv.first + " must be the same for every instance");
        std::vector<double> v;
        ConstantValue{raw}.expand_to(v, program.vec_size());
        objects[t] = std::move(v);
      }
    }
  }
  // outputs of a batched run, split back into one valuation per instance (outs[0..n)).  async: the
  // downloads are only enqueued; the words are valid after the queue is synchronised
  // res != nullptr (resident valuations): instance b of an output leaves as a handle — a view of the batched output
  // (evah_ct_unstack: the instances share one allocation, freed with the last of them) — and nothing is downloaded
  void get_outputs_batch(HipValuation *outs, size_t n_outs, bool async = false, const DeviceResident *res = nullptr) {
    for (auto &kv : program.outputs()) {
      auto &o = objects[kv.second];
      if (auto *c = std::get_if<std::shared_ptr<CtHandle>>(&o)) {
        HostCipher hc;
        uint32_t B = 1;
        chk(evah_ct_info((*c)->h, &hc.size, &hc.limbs, &hc.scale));
        chk(evah_ct_batch((*c)->h, &B));
        if (B != n_outs) throw std::runtime_error("Output " + kv.first + " does not depend on an encrypted input of the batch");
        if (res) {
          for (uint32_t b = 0; b < B; b++) {
            evah_ct *view = nullptr;
            chk(evah_ct_unstack(ctx, (*c)->h, b, &view));
            HostCipher one = hc;
            one.dev = std::make_shared<DeviceResident>(DeviceResident{res->root, res->queue, std::make_shared<CtHandle>(ctx, view), host.N});
            outs[b].values[kv.first] = std::move(one);
          }
          continue;
        }
        const size_t each = (size_t)hc.size * hc.limbs * host.N;
        std::vector<uint64_t *> ptrs(B);
        for (uint32_t b = 0; b < B; b++) {
          HostCipher one = hc;
          one.data.resize(each);
          outs[b].values[kv.first] = std::move(one);
          ptrs[b] = (uint64_t *)std::get<HostCipher>(outs[b].values[kv.first]).data.data();
        }
        chk((async ? evah_ct_download_instances_async : evah_ct_download_instances)(ctx, (*c)->h, ptrs.data()));
      } else {
        if (auto *p = std::get_if<std::shared_ptr<PtHandle>>(&o)) {
          HostPlain hp;
          chk(evah_pt_info((*p)->h, &hp.limbs, &hp.scale));
          hp.data.resize((size_t)hp.limbs * host.N);
          chk(evah_pt_download(ctx, (*p)->h, (uint64_t *)hp.data.data()));
          for (size_t b = 0; b < n_outs; b++) outs[b].values[kv.first] = hp;
        } else if (auto *r = std::get_if<std::vector<double>>(&o)) {
          for (size_t b = 0; b < n_outs; b++) outs[b].values[kv.first] = *r;
        } else {
          throw std::runtime_error("Output " + kv.first + " was not computed");
        }
      }
    }
  }

  void operator()(TermId t) {
    const Term &x = program.at(t);
    if (verbosity() >= 2) {
      std::printf("EVA: Execute t%u = %s(", t, op_name(x.op));
      for (size_t i = 0; i < x.operands.size(); i++) std::printf(i ? ",t%u" : "t%u", x.operands[i]);
      std::printf(")\n");
      std::fflush(stdout);
    }
    if (x.op == Op::Input) {
      if (std::holds_alternative<std::monostate>(objects[t])) throw std::runtime_error("Input value missing for an Input term");
      return;
    }
    const auto &a = x.operands;
    ctx = queues[choose_queue(t)];
    switch (x.op) {
    case Op::Constant: {
      std::vector<double> v;
      x.constant->expand_to(v, program.vec_size());
      objects[t] = std::move(v);
    } break;
    case Op::Encode:
      if (!is_raw(a[0])) throw std::runtime_error("Encode expects a raw operand");
      objects[t] = LazyPlain{a[0], x.encode_scale, x.encode_level};
      break;
    case Op::Add:
    case Op::Sub:
    case Op::Mul:
      if (is_raw(a[0]) && is_raw(a[1])) {
        const auto &u = raw(a[0]), &v = raw(a[1]);
        std::vector<double> o(u.size());
        for (size_t i = 0; i < u.size(); i++) o[i] = x.op == Op::Add ? u[i] + v[i] : x.op == Op::Sub ? u[i] - v[i] : u[i] * v[i];
        objects[t] = std::move(o);
      } else if (x.op == Op::Add) objects[t] = add(a[0], a[1]);
      else if (x.op == Op::Sub) objects[t] = sub(a[0], a[1]);
      else objects[t] = mul(a[0], a[1]);
      break;
    case Op::RotateLeftConst:
    case Op::RotateRightConst:
      if (is_raw(a[0])) {
        std::vector<double> o;
        if (x.op == Op::RotateLeftConst) rotate_left(raw(a[0]), x.rotation, o);
        else rotate_right(raw(a[0]), x.rotation, o);
        objects[t] = std::move(o);
      } else {
        if (has_value(t)) break; // already produced together with its sibling rotations
        // rightRotate passes the negated step (seal_executor.h:188)
        auto step_of = [&](TermId r) { const Term &y = program.at(r); return y.op == Op::RotateLeftConst ? y.rotation : -y.rotation; };
        // sibling rotations of the same ciphertext (convolution windows) go out as one wide call
        std::vector<TermId> group;
        if (batch_rotations)
          for (TermId u : program.at(a[0]).uses) {
            const Term &y = program.at(u);
            if ((y.op == Op::RotateLeftConst || y.op == Op::RotateRightConst) && step_of(u) != 0 && !has_value(u) &&
                std::find(group.begin(), group.end(), u) == group.end())
              group.push_back(u);
          }
        if (group.size() >= 2 && step_of(t) != 0) {
          for (size_t i = 0; i < group.size(); i += 64) {
            const uint32_t n = (uint32_t)std::min<size_t>(64, group.size() - i);
            std::vector<int32_t> steps(n);
            std::vector<evah_ct *> outs(n, nullptr);
            for (uint32_t r = 0; r < n; r++) steps[r] = step_of(group[i + r]);
            chk(evah_rotate_many(ctx, ct(a[0]), steps.data(), n, outs.data()));
            for (uint32_t r = 0; r < n; r++) {
              objects[group[i + r]] = std::make_shared<CtHandle>(ctx, outs[r]);
              queue_of[group[i + r]] = queue_of[t];
            }
          }
        } else {
          evah_ct *h = nullptr;
          chk(evah_rotate(ctx, ct(a[0]), step_of(t), &h));
          objects[t] = std::make_shared<CtHandle>(ctx, h);
        }
      }
      break;
    case Op::Negate:
      if (is_raw(a[0])) {
        auto o = raw(a[0]);
        for (auto &v : o) v = -v;
        objects[t] = std::move(o);
      } else {
        evah_ct *h = nullptr;
        chk(evah_negate(ctx, ct(a[0]), &h));
        objects[t] = std::make_shared<CtHandle>(ctx, h);
      }
      break;
    case Op::Relinearize:
    case Op::ModSwitch:
    case Op::Rescale:
      if (is_raw(a[0])) {
        // scale management of an unencrypted value: the reduction balancer can pair constants, and
        // the rescaler then treats raw x raw like any product.  The reference's SEALExecutor would
        // throw std::bad_variant_access here (seal_executor.h:197-215 take a Ciphertext); its
        // semantic executor copies (reference_executor.cpp) — which is what the value means.
        objects[t] = raw(a[0]);
        break;
      }
      if (x.op == Op::Relinearize) {
        const auto &uses = x.uses;
        if (fuse_relin_rescale && uses.size() == 1 && program.at(uses[0]).op == Op::Rescale) {
          objects[t] = LazyRelin{a[0]};
          queue_of[t] = queue_of[a[0]];
          break;
        }
        evah_ct *h = nullptr;
        chk(evah_relinearize(ctx, ct(a[0]), &h));
        objects[t] = std::make_shared<CtHandle>(ctx, h);
      } else if (x.op == Op::ModSwitch) {
        evah_ct *h = nullptr;
        chk(evah_mod_switch(ctx, ct(a[0]), &h));
        objects[t] = std::make_shared<CtHandle>(ctx, h);
      } else {
        evah_ct *h = nullptr;
        if (auto *lz = std::get_if<LazyRelin>(&objects[a[0]])) chk(evah_relinearize_rescale(ctx, ct(lz->src), x.rescale_divisor, &h));
        else chk(evah_rescale(ctx, ct(a[0]), x.rescale_divisor, &h));
        objects[t] = std::make_shared<CtHandle>(ctx, h);
      }
      break;
    case Op::Output:
      if (std::holds_alternative<LazyPlain>(objects[a[0]])) (void)pt(a[0]);
      objects[t] = objects[a[0]];
      break;
    default: throw std::runtime_error(std::string("Unhandled op ") + op_name(x.op));
    }
  }

  // The default walk: host-side nodes (constants, arithmetic on unencrypted values, Encode) are
  // evaluated here, every node that produces a ciphertext is lowered to an evah_op, and the whole
  // encrypted part of the program goes to the library as ONE evah_execute — the level scheduler,
  // the batched levels, the fused relinearize+rescale and the weighted sums live there
  // (include/eva_hip.h).  `skip` marks nodes already evaluated (resident constants);
  // free_values lets the library release intermediates after their last reader.
  void run_library(const std::vector<char> *skip, bool free_values) {
    const auto order = program.topo_order();
    ctx = queues[0];
    std::vector<char> produced(program.size(), 0); // ciphertext nodes computed by the submit
    std::vector<evah_op> ops;
    std::vector<evah_val> table(program.size(), evah_val{EVAH_VAL_NONE, nullptr});
    auto cipher = [&](TermId t) { return produced[t] || is_device_ct(t); };
    auto place = [&](TermId t) { // an operand that already exists on the device
      if (produced[t] || table[t].kind != EVAH_VAL_NONE) return;
      if (is_device_ct(t)) table[t] = evah_val{EVAH_VAL_CT, ct(t)};
      else if (is_plain(t)) table[t] = evah_val{EVAH_VAL_PT, pt(t)};
      else throw std::runtime_error("Unsupported operation encountered");
    };
    for (TermId t : order) {
      if (skip && (*skip)[t]) continue;
      const Term &x = program.at(t);
      bool any_cipher = false;
      for (TermId o : x.operands) any_cipher = any_cipher || cipher(o);
      if (!any_cipher) { // Input, Constant, Encode, arithmetic on raw values, outputs of unencrypted values
        (*this)(t);
        continue;
      }
      evah_op op{};
      op.op = (uint32_t)x.op;
      op.dst = t;
      op.src0 = x.operands[0];
      op.src1 = x.operands.size() > 1 ? x.operands[1] : 0;
      op.imm = (x.op == Op::RotateLeftConst || x.op == Op::RotateRightConst) ? (int32_t)x.rotation
               : x.op == Op::Rescale                                         ? (int32_t)x.rescale_divisor
                                                                             : 0;
      for (size_t k = 0; k < x.operands.size() && k < 2; k++) {
        place(x.operands[k]);
        if (free_values && produced[x.operands[k]]) op.flags |= (k == 0 ? EVAH_OPF_FREE_SRC0 : EVAH_OPF_FREE_SRC1);
      }
      produced[t] = 1;
      ops.push_back(op);
    }
    if (ops.empty()) return;
    int rc = 0;
    if (submit) { // several devices: the submit is split over them (multi_device.h); the table comes back on this queue's device
      std::set<uint32_t> keep;
      for (auto &kv : program.outputs()) keep.insert(program.at(kv.second).operands.empty() ? kv.second : program.at(kv.second).operands[0]);
      for (auto &kv : program.outputs()) keep.insert(kv.second);
      try {
        submit(ops, table, keep);
      } catch (const std::exception &e) {
        for (TermId t = 0; t < program.size(); t++)
          if (produced[t] && table[t].kind == EVAH_VAL_CT) objects[t] = std::make_shared<CtHandle>(ctx, static_cast<evah_ct *>(table[t].h));
        throw;
      }
    } else {
      rc = evah_execute(ctx, ops.data(), (uint32_t)ops.size(), table.data(), (uint32_t)table.size());
    }
    // whatever the submit produced and did not release is owned here now (also after an error)
    for (TermId t = 0; t < program.size(); t++)
      if (produced[t] && table[t].kind == EVAH_VAL_CT) objects[t] = std::make_shared<CtHandle>(ctx, static_cast<evah_ct *>(table[t].h));
    chk(rc);
  }

  // when set, the encrypted part's op list is handed to this instead of one evah_execute (sub-DAG split)
  std::function<void(std::vector<evah_op> &, std::vector<evah_val> &, const std::set<uint32_t> &)> submit;

  // ---- hooks used when an execution is captured into a graph
  const RuntimeValue &value(TermId t) const { return objects[t]; }
  void set_value(TermId t, RuntimeValue v) { objects[t] = std::move(v); }
  bool has_value(TermId t) const { return !std::holds_alternative<std::monostate>(objects[t]); }
  // Evaluate, eagerly and on queue 0, every node that does not depend on a ciphertext or plaintext
  // input (constants, raw arithmetic, Encode): these become persistent device plaintexts.
  std::vector<char> prepare_constants() {
    std::vector<char> done(program.size(), 0);
    for (TermId t : program.topo_order()) {
      const Term &x = program.at(t);
      if (x.op == Op::Input || x.op == Op::Output) continue;
      bool ok = true;
      for (TermId o : x.operands)
        if (!done[o]) ok = false;
      if (x.operands.empty() && x.op != Op::Constant) ok = false;
      if (!ok) continue;
      (*this)(t);
      if (std::holds_alternative<LazyPlain>(objects[t])) (void)pt(t);
      done[t] = 1;
    }
    return done;
  }

  // seal_executor.h:406-418 — release a value whose last consumer has run.  Raw vectors that
  // feed a not-yet-materialised Encode stay until that Encode is dropped.
  void free(TermId t) {
    if (program.at(t).op == Op::Output) return;
    if (is_raw(t))
      for (TermId u : program.at(t).uses)
        if (std::holds_alternative<LazyPlain>(objects[u])) return;
    if (is_cipher(t))
      for (TermId u : program.at(t).uses)
        if (std::holds_alternative<LazyRelin>(objects[u])) { deferred_free.emplace_back(u, t); return; }
    objects[t] = std::monostate{};
    // a consumed LazyRelin releases the size-3 value it was holding on to
    for (size_t i = 0; i < deferred_free.size(); i++)
      if (deferred_free[i].first == t) {
        objects[deferred_free[i].second] = std::monostate{};
        deferred_free.erase(deferred_free.begin() + i);
        break;
      }
  }

  // seal_executor.h:420-435 — outputs are downloaded into host values, or (res != null) handed over
  // as handles: `res` is the template (root, queue) of the resident value, the handle is filled in
  void get_outputs(HipValuation &out, const DeviceResident *res = nullptr) {
    for (auto &kv : program.outputs()) {
      auto &o = objects[kv.second];
      if (auto *c = std::get_if<std::shared_ptr<CtHandle>>(&o)) {
        HostCipher hc;
        chk(evah_ct_info((*c)->h, &hc.size, &hc.limbs, &hc.scale));
        if (res) {
          // an output that IS an input (Output(Input)) keeps the queue it came with
          bool passed_through = false;
          for (auto &in : resident_inputs)
            if (in->h == *c) { hc.dev = in; passed_through = true; break; }
          if (!passed_through) hc.dev = std::make_shared<DeviceResident>(DeviceResident{res->root, res->queue, *c, host.N});
          out.values[kv.first] = std::move(hc);
          continue;
        }
        hc.data.resize((size_t)hc.size * hc.limbs * host.N);
        hc.words_checked = true;
        chk(evah_ct_download(ctx, (*c)->h, (uint64_t *)hc.data.data()));
        out.values[kv.first] = std::move(hc);
      } else if (auto *p = std::get_if<std::shared_ptr<PtHandle>>(&o)) {
        HostPlain hp;
        chk(evah_pt_info((*p)->h, &hp.limbs, &hp.scale));
        hp.data.resize((size_t)hp.limbs * host.N);
        chk(evah_pt_download(ctx, (*p)->h, (uint64_t *)hp.data.data()));
        out.values[kv.first] = std::move(hp);
      } else if (auto *r = std::get_if<std::vector<double>>(&o)) {
        out.values[kv.first] = *r;
      } else {
        throw std::runtime_error("Output " + kv.first + " was not computed");
      }
    }
  }

private:
  Program &program;
  const HostContext &host;
  std::vector<evah_ctx *> queues;
  evah_ctx *ctx; // queue the current node is issued on
  std::vector<RuntimeValue> objects;
  std::vector<uint32_t> queue_of;
  uint32_t next_queue = 0;
  std::vector<double> scratch;
  std::vector<std::pair<TermId, TermId>> deferred_free; // (lazy relin term, its source)
  const DeviceCtx *root = nullptr;
  std::vector<std::shared_ptr<DeviceResident>> resident_inputs;
public:
  bool batch_inputs_resident = false; // set_inputs_batch stacked at least one input from device handles
private:
  bool batch_rotations = std::getenv("EVA_BATCH_ROTATIONS") ? std::atoi(std::getenv("EVA_BATCH_ROTATIONS")) != 0 : true;
  bool fuse_relin_rescale = std::getenv("EVA_FUSE_RELIN_RESCALE") ? std::atoi(std::getenv("EVA_FUSE_RELIN_RESCALE")) != 0 : true;
  bool device_encode = std::getenv("EVA_DEVICE_ENCODE") ? std::atoi(std::getenv("EVA_DEVICE_ENCODE")) != 0 : true;
  // every coefficient of the encoding is bounded by 2 * sum|slot values| * scale / N; the device
  // path needs that below 2^62 (and within the modulus, so that the host's exact range check —
  // which throws "encoded values are too large" — is not needed)
  bool device_encodable(const std::vector<double> &in, double scale, uint32_t limbs) const {
    const size_t slots = host.N / 2;
    if (in.empty() || in.size() > slots || slots % in.size()) return false;
    double sum = 0;
    for (double v : in) {
      if (!std::isfinite(v)) return false;
      sum += std::fabs(v);
    }
    const double bound = 2.0 * sum * (double)(slots / in.size()) * scale / (double)host.N;
    const int bits = (int)std::ceil(std::log2(std::max(bound, 1.0))) + 1;
    return bits < 62 && bits < host.total_bits[limbs];
  }

  // Queue for node t: key-switching / rescaling consumers of a fanned-out value are spread
  // round-robin (they are independent and heavy); everything else follows its first
  // device-resident operand so chains stay on one stream and need no cross-queue ordering.
  uint32_t choose_queue(TermId t) {
    const Term &x = program.at(t);
    uint32_t base = 0;
    bool have = false;
    for (TermId o : x.operands)
      if (is_cipher(o)) { base = queue_of[o]; have = true; break; }
    uint32_t q = base;
    if (queues.size() > 1 && have) {
      const bool heavy = x.op == Op::RotateLeftConst || x.op == Op::RotateRightConst || x.op == Op::Relinearize ||
                         x.op == Op::Rescale || (x.op == Op::Mul && x.operands.size() == 2 && is_cipher(x.operands[0]) && is_cipher(x.operands[1]));
      bool fanout = false;
      for (TermId o : x.operands)
        if (is_cipher(o) && program.at(o).uses.size() > 1) fanout = true;
      if (heavy && fanout) q = next_queue++ % (uint32_t)queues.size();
    }
    queue_of[t] = q;
    return q;
  }

  bool is_cipher(TermId t) const { return std::holds_alternative<std::shared_ptr<CtHandle>>(objects[t]); }
  bool is_device_ct(TermId t) const { return std::holds_alternative<std::shared_ptr<CtHandle>>(objects[t]); }
  bool is_plain(TermId t) const {
    return std::holds_alternative<std::shared_ptr<PtHandle>>(objects[t]) || std::holds_alternative<LazyPlain>(objects[t]);
  }
  bool is_raw(TermId t) const { return std::holds_alternative<std::vector<double>>(objects[t]); }
  const std::vector<double> &raw(TermId t) const { return std::get<std::vector<double>>(objects[t]); }
  evah_ct *ct(TermId t) {
    auto *p = std::get_if<std::shared_ptr<CtHandle>>(&objects[t]);
    if (!p) throw std::runtime_error("Unsupported operation encountered");
    return (*p)->h;
  }
  evah_pt *pt(TermId t) {
    if (auto *lz = std::get_if<LazyPlain>(&objects[t])) objects[t] = encode_raw(lz->src, lz->scale_bits, lz->level);
    return std::get<std::shared_ptr<PtHandle>>(objects[t])->h;
  }

  RuntimeValue wrap(evah_ct *h) { return std::make_shared<CtHandle>(ctx, h); }

  // seal_executor.h:114-135: cipher first; cipher+cipher or cipher+plain
  RuntimeValue add(TermId a, TermId b) {
    if (!is_cipher(a)) {
      if (!is_cipher(b)) throw std::runtime_error("Unsupported operation encountered");
      return add(b, a);
    }
    evah_ct *h = nullptr;
    if (is_cipher(b)) chk(evah_add(ctx, ct(a), ct(b), &h));
    else if (is_plain(b)) chk(evah_add_plain(ctx, ct(a), pt(b), &h));
    else throw std::runtime_error("Unsupported operation encountered");
    return wrap(h);
  }
  // seal_executor.h:137-150
  RuntimeValue sub(TermId a, TermId b) {
    evah_ct *h = nullptr;
    if (is_cipher(b)) chk(evah_sub(ctx, ct(a), ct(b), &h));
    else if (is_plain(b)) chk(evah_sub_plain(ctx, ct(a), pt(b), &h));
    else throw std::runtime_error("Unsupported operation encountered");
    return wrap(h);
  }
  // seal_executor.h:152-175: square when both operands are the same term
  RuntimeValue mul(TermId a, TermId b) {
    if (!is_cipher(a) && is_cipher(b)) return mul(b, a);
    evah_ct *h = nullptr;
    if (is_cipher(b)) {
      if (a == b) chk(evah_square(ctx, ct(a), &h));
      else chk(evah_multiply(ctx, ct(a), ct(b), &h));
    } else if (is_plain(b)) chk(evah_multiply_plain(ctx, ct(a), pt(b), &h));
    else throw std::runtime_error("Unsupported operation encountered");
    return wrap(h);
  }

  // seal_executor.h:217-243: replicate vec_size -> N/2 slots, encode at 2^scale, level -> limbs.
  // FP64 special FFT on the host, per-limb NTT on the device; a uniform vector needs neither.
  RuntimeValue encode_raw(TermId src, uint32_t scale_bits, uint32_t level) {
    const auto &in = raw(src);
    const uint32_t limbs = host.k - 1 - level;
    if (level >= host.k - 1) throw std::runtime_error("Encode level exceeds the modulus chain");
    const double scale = std::pow(2.0, (double)scale_bits);
    bool uniform = true;
    for (double v : in)
      if (v != in[0]) { uniform = false; break; }
    evah_pt *h = nullptr;
    if (uniform) {
      std::vector<u64> vals(limbs);
      host.encode_uniform(in[0], scale, limbs, vals.data());
      chk(evah_pt_uniform(ctx, limbs, scale, (const uint64_t *)vals.data(), &h));
    } else if (device_encode && device_encodable(in, scale, limbs)) {
      // FP64 special FFT, rounding, residues and NTT all on the device (same plaintext, bit for bit)
      chk(evah_pt_encode(ctx, in.data(), (uint32_t)in.size(), limbs, scale, &h));
    } else {
      const size_t slots = host.N / 2;
      scratch.clear();
      scratch.reserve(slots);
      for (size_t r = slots / in.size(); r > 0; --r) scratch.insert(scratch.end(), in.begin(), in.end());
      std::vector<u64> coeff((size_t)limbs * host.N);
      host.encode_coeff(scratch.data(), scale, limbs, coeff.data());
      chk(evah_pt_upload_coeff(ctx, limbs, scale, (const uint64_t *)coeff.data(), &h));
    }
    return std::make_shared<PtHandle>(ctx, h);
  }
};

// Serial topological walk (ProgramTraversal::forwardPass; the reference never frees here).
template <class Exec> void run_serial(Program &p, Exec &ex) {
  for (TermId t : p.topo_order()) ex(t);
}
// Dependency-counting walk that releases operands when their last consumer has run — the
// single-queue form of MulticoreProgramTraversal::forwardPass (:55-78): device work is
// stream-ordered, so "evaluated" means "enqueued".
template <class Exec> void run_counted(Program &p, Exec &ex, const std::vector<char> *skip = nullptr) {
  auto order = p.topo_order();
  std::vector<uint32_t> succ(p.size(), 0);
  for (TermId t : order)
    for (TermId o : p.at(t).operands) succ[o]++;
  for (TermId t : order) {
    if (skip && (*skip)[t]) continue;
    ex(t);
    for (TermId o : p.at(t).operands)
      if (--succ[o] == 0) ex.free(o);
  }
}

} // namespace evahost
#include "multi_device.h"
#include "public_ctx.h"
