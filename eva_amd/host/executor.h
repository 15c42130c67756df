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
#pragma once
#include <array>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <functional>
#include <map>
#include <set>
#include <memory>
#include <tuple>
#include <random>
#include <string>
#include <unordered_map>
#include <variant>

#include "ckks_host.h"
#include "eva_hip.h"
#include "passes.h"

namespace evahost {

using Valuation = std::unordered_map<std::string, std::vector<double>>;

[[noreturn]] inline void throw_backend() { throw std::runtime_error(std::string("eva_hip: ") + evah_last_error()); }
inline void chk(int rc) { if (rc) throw_backend(); }

// ---- plain-double reference semantics (reference_executor.cpp:14-115)
inline void rotate_left(const std::vector<double> &in, int32_t shift, std::vector<double> &out) {
  int64_t n = (int64_t)in.size(), s = shift;
  while (s > 0 && s >= n) s -= n;
  while (s < 0) s += n;
  out.resize(in.size());
  for (int64_t i = 0; i < n; i++) out[i] = in[(i + s) % n];
}
inline void rotate_right(const std::vector<double> &in, int32_t shift, std::vector<double> &out) {
  int64_t n = (int64_t)in.size(), s = shift;
  while (s > 0 && s >= n) s -= n;
  while (s < 0) s += n;
  out.resize(in.size());
  for (int64_t i = 0; i < n; i++) out[(i + s) % n] = in[i];
}

inline Valuation evaluate(Program &p, const Valuation &inputs) {
  std::vector<std::vector<double>> vals(p.size());
  const size_t n = p.vec_size();
  for (auto &kv : inputs) {
    TermId t = p.input(kv.first);
    vals[t] = kv.second;
    if (vals[t].size() != n)
      throw std::runtime_error("The length of all inputs must be the same as program's vector size. Input " + kv.first +
                               " has length " + std::to_string(vals[t].size()) + ", but vector size is " + std::to_string(n));
  }
  for (TermId t : p.topo_order()) {
    const Term &x = p.at(t);
    auto &out = vals[t];
    auto bin = [&](auto f) {
      const auto &a = vals[x.operands[0]], &b = vals[x.operands[1]];
      out.resize(a.size());
      for (size_t i = 0; i < a.size(); i++) out[i] = f(a[i], b[i]);
    };
    switch (x.op) {
    case Op::Input: break;
    case Op::Constant: x.constant->expand_to(out, n); break;
    case Op::Add: bin([](double a, double b) { return a + b; }); break;
    case Op::Sub: bin([](double a, double b) { return a - b; }); break;
    case Op::Mul: bin([](double a, double b) { return a * b; }); break;
    case Op::RotateLeftConst: rotate_left(vals[x.operands[0]], x.rotation, out); break;
    case Op::RotateRightConst: rotate_right(vals[x.operands[0]], x.rotation, out); break;
    case Op::Negate: {
      const auto &a = vals[x.operands[0]];
      out.resize(a.size());
      for (size_t i = 0; i < a.size(); i++) out[i] = -a[i];
    } break;
    case Op::Encode:
    case Op::Output:
    case Op::Relinearize:
    case Op::ModSwitch:
    case Op::Rescale: out = vals[x.operands[0]]; break;
    default: throw std::runtime_error(std::string("Unhandled op ") + op_name(x.op));
    }
  }
  Valuation outv;
  for (auto &kv : p.outputs()) outv[kv.first] = vals[kv.second];
  return outv;
}

// ---- values crossing the execute() boundary (seal.h:21-41)
using SchemeValue = std::variant<HostCipher, HostPlain, std::vector<double>>;
struct HipValuation {
  std::unordered_map<std::string, SchemeValue> values;
  // the encryption parameters the values belong to (SEALValuation::params, seal.h:23-27): set by encrypt(),
  // execute() and load(); needed to write the valuation in the reference's SEAL wire format
  std::shared_ptr<const HostContext> params;
};

// RAII device handles
struct CtHandle {
  evah_ctx *ctx = nullptr;
  evah_ct *h = nullptr;
  CtHandle() {}
  CtHandle(evah_ctx *c, evah_ct *p) : ctx(c), h(p) {}
  CtHandle(CtHandle &&o) noexcept : ctx(o.ctx), h(o.h) { o.h = nullptr; }
  CtHandle &operator=(CtHandle &&o) noexcept { reset(); ctx = o.ctx; h = o.h; o.h = nullptr; return *this; }
  CtHandle(const CtHandle &) = delete;
  CtHandle &operator=(const CtHandle &) = delete;
  void reset() { if (h) evah_ct_free(ctx, h); h = nullptr; }
  ~CtHandle() { reset(); }
};
struct PtHandle {
  evah_ctx *ctx = nullptr;
  evah_pt *h = nullptr;
  PtHandle() {}
  PtHandle(evah_ctx *c, evah_pt *p) : ctx(c), h(p) {}
  PtHandle(PtHandle &&o) noexcept : ctx(o.ctx), h(o.h) { o.h = nullptr; }
  PtHandle &operator=(PtHandle &&o) noexcept { reset(); ctx = o.ctx; h = o.h; o.h = nullptr; return *this; }
  PtHandle(const PtHandle &) = delete;
  PtHandle &operator=(const PtHandle &) = delete;
  void reset() { if (h) evah_pt_free(ctx, h); h = nullptr; }
  ~PtHandle() { reset(); }
};

// ---- device contexts (seal.h:45-97 keeps a SEALContext per key set; here: tables + keys in HBM)
struct DeviceCtx {
  evah_ctx *h = nullptr;
  DeviceCtx(uint32_t N, const std::vector<u64> &primes, int device) {
    chk(evah_ctx_create(N, (uint32_t)primes.size(), (const uint64_t *)primes.data(), device, &h));
  }
  ~DeviceCtx() { evah_ctx_destroy(h); }
  DeviceCtx(const DeviceCtx &) = delete;
  DeviceCtx &operator=(const DeviceCtx &) = delete;
};
// A second issue queue of a device context (evah_ctx_fork).  It keeps its parent alive, so a value
// that was produced through it can outlive the HipPublic that created the queue.
struct Fork {
  std::shared_ptr<DeviceCtx> parent;
  evah_ctx *h = nullptr;
  explicit Fork(std::shared_ptr<DeviceCtx> p) : parent(std::move(p)) { chk(evah_ctx_fork(parent->h, &h)); }
  ~Fork() { evah_ctx_destroy(h); }
  Fork(const Fork &) = delete;
  Fork &operator=(const Fork &) = delete;
};
// The device half of a ciphertext value (ckks_host.h HostCipher::dev): a handle of `root`'s device
// state.  seal_executor.h:264-277 / :420-435 copy values in and out of the executor; a resident value
// is passed by handle instead — no copy, no PCIe.
struct DeviceResident {
  std::shared_ptr<DeviceCtx> root; // tables and keys the handle belongs to
  std::shared_ptr<Fork> queue;     // the issue queue whose pool holds the buffer (null: the root's own)
  std::shared_ptr<CtHandle> h;
  uint32_t N = 0;                  // poly_modulus_degree: words per limb
  evah_ctx *ctx() const { return queue ? queue->h : root->h; }
};
// host words of a ciphertext value, downloaded on first use (waits for the value to be computed)
inline const CipherWords &words(const HostCipher &c) {
  if (c.data.empty() && c.dev) {
    CipherWords w((size_t)c.size * c.limbs * c.dev->N);
    chk(evah_ct_download(c.dev->ctx(), c.dev->h->h, (uint64_t *)w.data()));
    c.data = std::move(w);
    c.words_checked = true; // the device's own residues
  }
  return c.data;
}
inline bool resident_only(const HostCipher &c) { return c.data.empty() && c.dev; }

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
          ptrs[b] = (const uint64_t *)words(c).data(); // the batched handle is assembled from host words
        }
        evah_ct *h = nullptr;
        chk((async ? evah_ct_upload_instances_async : evah_ct_upload_instances)(ctx, B, c0->size, c0->limbs, c0->scale, ptrs.data(), &h));
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
  void get_outputs_batch(HipValuation *outs, size_t n_outs, bool async = false) {
    for (auto &kv : program.outputs()) {
      auto &o = objects[kv.second];
      if (auto *c = std::get_if<std::shared_ptr<CtHandle>>(&o)) {
        HostCipher hc;
        uint32_t B = 1;
        chk(evah_ct_info((*c)->h, &hc.size, &hc.limbs, &hc.scale));
        chk(evah_ct_batch((*c)->h, &B));
        if (B != n_outs) throw std::runtime_error("Output " + kv.first + " does not depend on an encrypted input of the batch");
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
  // HIP streams independent DAG nodes are spread over (EVA_NUM_STREAMS).  Default 1: at these
  // kernel sizes a single in-order queue keeps the GPU as busy as the host can feed it; more
  // queues are correct (ordering is enforced per buffer inside libeva_hip.so) and pay off when
  // nodes are large enough to be GPU-bound.
  int num_queues = 1;

  // SEALPublic::encrypt (seal.cpp:24-102)
  HipValuation encrypt(const Valuation &inputs, const CKKSSignature &sig) {
    const size_t slots = host->N / 2;
    if (sig.vec_size <= 0) throw std::runtime_error("Signature vector size must be positive");
    if (slots < (size_t)sig.vec_size) throw std::runtime_error("Vector size cannot be larger than slot count");
    if (slots % sig.vec_size) throw std::runtime_error("Vector size must exactly divide the slot count");
    HipValuation out;
    SecureRng rng; // a fresh ChaCha20 stream keyed with 256 bits from the OS for this call (csprng.h)
    for (auto &kv : inputs) {
      const auto &v = kv.second;
      if (v.size() != (size_t)sig.vec_size) throw std::runtime_error("Input size does not match program vector size");
      auto it = sig.inputs.find(kv.first);
      if (it == sig.inputs.end()) throw std::out_of_range("No input named " + kv.first + " in the signature");
      const CKKSEncodingInfo &info = it->second;
      if (info.input_type == Type::Cipher || info.input_type == Type::Plain) {
        if ((uint32_t)info.level >= host->k - 1) throw std::runtime_error("Input level exceeds the modulus chain");
        HostPlain pt;
        pt.limbs = host->k - 1 - (uint32_t)info.level;
        pt.scale = std::pow(2.0, (double)info.scale);
        if (info.input_type == Type::Cipher && client_on_device() && device_encodable(v, pt.scale, pt.limbs)) {
          // encoder and encryptor both on the GPU (evah_pt_encode -> evah_encrypt): the plaintext never
          // exists on the host.  Same plaintext as the host encoder bit for bit (tests/test_encode_parity.py)
          // and the same sampler calls in the same order, hence the same ciphertext as every other path
          out.values[kv.first] = encrypt_on_device(nullptr, &v, pt.scale, pt.limbs, rng);
          continue;
        }
        pt.data.resize((size_t)pt.limbs * host->N);
        std::vector<double> vec(slots);
        for (size_t r = 0; r < slots / v.size(); r++) std::copy(v.begin(), v.end(), vec.begin() + r * v.size());
        host->encode_coeff(vec.data(), pt.scale, pt.limbs, pt.data.data());
        if (info.input_type == Type::Cipher && client_on_device()) {
          // device path: the per-limb transforms, the public-key products and the mod-down run on the
          // GPU (evah_encrypt); the host keeps the FP64 encoder and the sampling (same sampler calls,
          // in the same order, as evahost::encrypt — so both paths give the same ciphertext for the
          // same random stream)
          out.values[kv.first] = encrypt_on_device(&pt, nullptr, pt.scale, pt.limbs, rng);
          continue;
        }
        for (uint32_t i = 0; i < pt.limbs; i++) host->ntt(i, pt.data.data() + (size_t)i * host->N);
        if (info.input_type == Type::Cipher) out.values[kv.first] = evahost::encrypt(*host, pk, pt, rng);
        else out.values[kv.first] = std::move(pt);
      } else {
        out.values[kv.first] = v;
      }
    }
    return out;
  }

  // SEALPublic::execute (seal.cpp:104-122) — THE hot path.  First call for a program: upload
  // inputs, walk the DAG issuing HIP work over the queues, download outputs.  From the second call
  // on (same program object, same input shapes, no Raw inputs) the whole walk is replayed from a
  // captured hipGraph: per call the host refills the input slots, launches one graph, downloads.
  bool use_graphs = true; // EVA_GRAPH=0 disables
  HipValuation execute(Program &program, const HipValuation &inputs) {
    const bool multi = devices.size() > 1;
    if (multi && shard_mode == "limb") {
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
        if (it->second->matches(program, inputs)) return run_plan(*it->second, inputs);
        plans.erase(it); // same address, different program or shapes: forget the stale plan
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
    for (auto &kv : plans) for (auto &f : kv.second->queues) chk(evah_ctx_sync(f->h));
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
        chk(evah_ctx_key_bytes(limb->group().ctx[s], &b));
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
  uint32_t batch_chunk = 32;
  // groups in flight in execute_batch: group g is enqueued on queue g mod batch_depth, so the copies of one group
  // overlap the kernels of the others; device memory = batch_depth groups' working sets
  uint32_t batch_depth = std::getenv("EVA_BATCH_DEPTH") ? (uint32_t)std::atoi(std::getenv("EVA_BATCH_DEPTH")) : 4;
  std::vector<HipValuation> execute_batch(Program &program, const std::vector<const HipValuation *> &inputs) {
    ensure_device();
    if (batch_chunk < 1 || batch_chunk > 64) throw std::runtime_error("batch_chunk must be 1..64");
    std::vector<HipValuation> all(inputs.size());
    // Groups rotate over batch_depth issue queues (default four: +6 % over two on config 4) and nothing waits in between: each group's uploads,
    // launches and downloads are enqueued in queue order (evah_ct_*_instances_async), so the copies
    // of one group overlap the kernels of the other and the host never idles the device.  Device
    // memory stays at two groups' working sets (the pools recycle in queue order); the inputs belong
    // to the caller and the outputs are allocated up front, so both outlive the final synchronisation.
    if (devices.size() > 1 && shard_mode == "dag") return execute_batch_multi(program, inputs);
    if (batch_depth < 2 || batch_depth > 8) throw std::runtime_error("batch_depth must be 2..8");
    while (batch_forks.size() + 1 < batch_depth) batch_forks.push_back(std::make_shared<Fork>(dev));
    std::vector<evah_ctx *> qs{dev->h};
    for (uint32_t i = 0; i + 1 < batch_depth; i++) qs.push_back(batch_forks[i]->h);
    const size_t Q = qs.size();
    // constants (Constant / Encode nodes and raw arithmetic on them) are evaluated once, by the
    // first group, and shared by all groups: their plaintexts stay resident for the whole call
    std::vector<char> done;
    std::vector<HipExecutor::RuntimeValue> consts;
    auto finish = [&]() {
      int rc = 0;
      for (evah_ctx *q : qs) rc |= evah_ctx_sync(q);
      if (rc) throw_backend();
    };
    size_t g = 0;
    const bool bounded = std::getenv("EVA_BATCH_BOUNDED") ? std::atoi(std::getenv("EVA_BATCH_BOUNDED")) != 0 : false;
    try {
      for (size_t i0 = 0; i0 < inputs.size(); i0 += batch_chunk, g++) {
        const size_t n = std::min<size_t>(batch_chunk, inputs.size() - i0);
        std::vector<const HipValuation *> chunk(inputs.begin() + i0, inputs.begin() + i0 + n);
        if (bounded && g >= Q) chk(evah_ctx_sync(qs[g % Q])); // group g-Q (same queue) has left the device
        HipExecutor ex(program, *host, std::vector<evah_ctx *>{qs[g % Q]}, dev.get());
        if (g == 0) {
          done = ex.prepare_constants();
          consts.resize(program.size());
          for (TermId t = 0; t < program.size(); t++)
            if (done[t]) consts[t] = ex.value(t);
        } else {
          for (TermId t = 0; t < program.size(); t++)
            if (done[t]) ex.set_value(t, consts[t]);
        }
        ex.set_inputs_batch(chunk, true);
        if (library_scheduler) ex.run_library(&done, true);
        else run_counted(program, ex, &done);
        ex.get_outputs_batch(all.data() + i0, n, true);
      }
    } catch (...) {
      for (evah_ctx *q : qs) (void)evah_ctx_sync(q); // copies in flight still target `all` and the caller's inputs
      throw;
    }
    finish();
    return all;
  }

  // "dag" mode (BASELINE config 4): the groups of a batch are dealt over the members
  // of `devices` — group g on member g mod G, batch_depth issue queues per member so a member's copies overlap its
  // kernels — with no data-path exchange: instances are independent.  Same results as execute_batch on one
  // device.  (The driver's scaling curve uses one process per GPU, eva_amd/dist.py; this is the same
  // partition inside one execute_batch call.)
  std::vector<HipValuation> execute_batch_multi(Program &program, const std::vector<const HipValuation *> &inputs) {
    if (batch_chunk < 1 || batch_chunk > 64) throw std::runtime_error("batch_chunk must be 1..64");
    ensure_group(false);
    const size_t G = group->size();
    if (batch_depth < 2 || batch_depth > 8) throw std::runtime_error("batch_depth must be 2..8");
    const size_t D = batch_depth;
    if (batch_queues.size() != D * G) {
      batch_queues.clear();
      for (size_t m = 0; m < G; m++)
        for (size_t k = 0; k < D; k++) batch_queues.push_back(std::make_shared<Fork>(group->roots[m]));
    }
    std::vector<HipValuation> all(inputs.size());
    std::vector<std::vector<char>> done(G);
    std::vector<std::vector<HipExecutor::RuntimeValue>> consts(G);
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
        if (done[m].empty()) { // the member's constants: encoded once, by its first group
          done[m] = ex.prepare_constants();
          consts[m].resize(program.size());
          for (TermId t = 0; t < program.size(); t++)
            if (done[m][t]) consts[m][t] = ex.value(t);
        } else {
          for (TermId t = 0; t < program.size(); t++)
            if (done[m][t]) ex.set_value(t, consts[m][t]);
        }
        ex.set_inputs_batch(chunk, true);
        ex.run_library(&done[m], true);
        ex.get_outputs_batch(all.data() + i0, n, true);
      }
    } catch (...) {
      (void)sync_all(); // copies in flight still target `all` and the caller's inputs
      throw;
    }
    if (sync_all()) throw_backend();
    return all;
  }

  evah_ctx *device_ctx() {
    ensure_device();
    return dev->h;
  }

  ~HipPublic() {
    const_cache.clear();
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
  void drop_graphs() { plans.clear(); seen.clear(); no_graph.clear(); const_cache.clear(); }

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
      if (d < 0 || d >= n) throw std::runtime_error("device " + std::to_string(d) + " requested, " + std::to_string(n) + " visible");
  }
  void ensure_group(bool) {
    if (group && group_ids == devices) return;
    check_devices();
    group.reset();
    group = std::make_unique<DeviceGroup>(make_device_group(devices, dev, device, *host, [this](evah_ctx *c) { upload_eval_keys(c); }, true));
    group_ids = devices;
  }

  // SEALPublic::execute over limb-sharded values: serial forwardPass, SEALExecutor's dispatch per node
  // (seal_executor.h:279-404) on a LimbShardEvaluator.  Values come in and go out as host words (a
  // sharded value has no single device handle); constants are encoded on the host once per program.
  HipValuation execute_limb(Program &program, const HipValuation &inputs) {
    if (!limb || limb_ids != devices) {
      check_devices();
      limb.reset();
      limb_const.clear();
      limb = std::make_unique<LimbShardEvaluator>(*host, make_limb_group(devices, *host, [this](evah_ctx *c) { upload_eval_keys(c); }));
      limb_ids = devices;
    }
    LimbShardEvaluator &ev = *limb;
    const uint64_t words0 = ev.exchanged_words, launches0 = ev.exchange_launches;
    using Val = std::variant<std::monostate, ShardedValue, std::vector<double>>;
    std::vector<Val> vals(program.size());
    const size_t n_vec = program.vec_size();
    if (n_vec > host->N / 2) throw std::runtime_error("Vector size cannot be larger than slot count");
    HipExecutor shapes(program, *host, std::vector<evah_ctx *>{dev->h}); // for its shape / range checks of untrusted values
    for (auto &kv : inputs.values) {
      TermId t = program.input(kv.first);
      if (auto *c = std::get_if<HostCipher>(&kv.second)) {
        shapes.check_shape(kv.first, *c);
        vals[t] = ev.upload((const u64 *)words(*c).data(), c->size, c->limbs, c->scale);
      } else if (auto *p = std::get_if<HostPlain>(&kv.second)) {
        shapes.check_shape(kv.first, *p);
        vals[t] = ev.upload(p->data.data(), 0, p->limbs, p->scale);
      } else {
        std::vector<double> v;
        ConstantValue{std::get<std::vector<double>>(kv.second)}.expand_to(v, n_vec);
        vals[t] = std::move(v);
      }
    }
    LimbConst &lc = limb_const[&program];
    const uint64_t h = program_hash(program);
    if (lc.hash != h) { lc.plain.clear(); lc.hash = h; }
    auto is_raw = [&](TermId t) { return std::holds_alternative<std::vector<double>>(vals[t]); };
    auto raw = [&](TermId t) -> const std::vector<double> & { return std::get<std::vector<double>>(vals[t]); };
    auto sv = [&](TermId t) -> const ShardedValue & {
      auto *p = std::get_if<ShardedValue>(&vals[t]);
      if (!p) throw std::runtime_error("Unsupported operation encountered");
      return *p;
    };
    auto is_ct = [&](TermId t) { auto *p = std::get_if<ShardedValue>(&vals[t]); return p && p->is_ct(); };
    for (TermId t : program.topo_order()) {
      const Term &x = program.at(t);
      const auto &a = x.operands;
      switch (x.op) {
      case Op::Input:
        if (std::holds_alternative<std::monostate>(vals[t])) throw std::runtime_error("Input value missing for an Input term");
        break;
      case Op::Constant: {
        std::vector<double> v;
        x.constant->expand_to(v, n_vec);
        vals[t] = std::move(v);
      } break;
      case Op::Encode: {
        if (!is_raw(a[0])) throw std::runtime_error("Encode expects a raw operand");
        auto it = lc.plain.find(t);
        bool from_input = false; // an Encode fed by a Raw INPUT changes from call to call: never cached
        for (auto &kv : program.inputs()) from_input = from_input || depends_on(program, a[0], kv.second);
        if (it == lc.plain.end() || from_input) {
          if (x.encode_level >= host->k - 1) throw std::runtime_error("Encode level exceeds the modulus chain");
          const uint32_t limbs = host->k - 1 - x.encode_level;
          const double scale = std::pow(2.0, (double)x.encode_scale);
          const auto &in = raw(a[0]);
          const size_t slots = host->N / 2;
          std::vector<double> rep;
          rep.reserve(slots);
          for (size_t r = slots / in.size(); r > 0; --r) rep.insert(rep.end(), in.begin(), in.end());
          std::vector<u64> pt((size_t)limbs * host->N);
          host->encode_coeff(rep.data(), scale, limbs, pt.data());
          for (uint32_t i = 0; i < limbs; i++) host->ntt(i, pt.data() + (size_t)i * host->N);
          ShardedValue v = ev.upload(pt.data(), 0, limbs, scale);
          if (from_input) { vals[t] = std::move(v); break; }
          it = lc.plain.emplace(t, std::move(v)).first;
        }
        vals[t] = it->second;
      } break;
      case Op::Add:
      case Op::Sub:
      case Op::Mul:
        if (is_raw(a[0]) && is_raw(a[1])) {
          const auto &u = raw(a[0]), &v = raw(a[1]);
          std::vector<double> o(u.size());
          for (size_t i = 0; i < u.size(); i++) o[i] = x.op == Op::Add ? u[i] + v[i] : x.op == Op::Sub ? u[i] - v[i] : u[i] * v[i];
          vals[t] = std::move(o);
        } else if (x.op == Op::Sub) {
          if (!is_ct(a[0])) throw std::runtime_error("Unsupported operation encountered");
          vals[t] = is_ct(a[1]) ? ev.sub(sv(a[0]), sv(a[1])) : ev.sub_plain(sv(a[0]), sv(a[1]));
        } else {
          TermId c = a[0], o = a[1]; // the ciphertext first (seal_executor.h:116-119, :155-158)
          if (!is_ct(c)) std::swap(c, o);
          if (!is_ct(c)) throw std::runtime_error("Unsupported operation encountered");
          if (x.op == Op::Add) vals[t] = is_ct(o) ? ev.add(sv(c), sv(o)) : ev.add_plain(sv(c), sv(o));
          else vals[t] = is_ct(o) ? (a[0] == a[1] ? ev.square(sv(c)) : ev.multiply(sv(c), sv(o))) : ev.multiply_plain(sv(c), sv(o));
        }
        break;
      case Op::RotateLeftConst:
      case Op::RotateRightConst:
        if (is_raw(a[0])) {
          std::vector<double> o;
          if (x.op == Op::RotateLeftConst) rotate_left(raw(a[0]), x.rotation, o);
          else rotate_right(raw(a[0]), x.rotation, o);
          vals[t] = std::move(o);
        } else {
          vals[t] = ev.rotate(sv(a[0]), x.op == Op::RotateLeftConst ? x.rotation : -x.rotation);
        }
        break;
      case Op::Negate:
        if (is_raw(a[0])) {
          auto o = raw(a[0]);
          for (auto &v : o) v = -v;
          vals[t] = std::move(o);
        } else {
          vals[t] = ev.negate(sv(a[0]));
        }
        break;
      case Op::Relinearize:
      case Op::ModSwitch:
      case Op::Rescale:
        if (is_raw(a[0])) vals[t] = raw(a[0]);
        else if (x.op == Op::Relinearize) vals[t] = ev.relinearize(sv(a[0]));
        else if (x.op == Op::ModSwitch) vals[t] = ev.mod_switch(sv(a[0]));
        else vals[t] = ev.rescale(sv(a[0]), x.rescale_divisor);
        break;
      case Op::Output: vals[t] = vals[a[0]]; break;
      default: throw std::runtime_error(std::string("Unhandled op ") + op_name(x.op));
      }
    }
    HipValuation out;
    for (auto &kv : program.outputs()) {
      auto &o = vals[kv.second];
      if (auto *v = std::get_if<ShardedValue>(&o)) {
        if (v->is_ct()) {
          out.values[kv.first] = ev.download(*v);
        } else { // a plaintext output: assemble through a size-1 view of the same words
          ShardedValue as_ct = *v; // plaintext parts cannot be downloaded as ciphertexts: re-upload is not needed, use pt download
          HostPlain hp;
          hp.limbs = v->limbs;
          hp.scale = v->scale;
          hp.data = ev.download_plain(*v);
          hp.words_checked = true;
          out.values[kv.first] = std::move(hp);
        }
      } else if (auto *r = std::get_if<std::vector<double>>(&o)) {
        out.values[kv.first] = *r;
      } else {
        throw std::runtime_error("Output " + kv.first + " was not computed");
      }
    }
    last_exchanged_words = ev.exchanged_words - words0;
    last_exchange_launches = ev.exchange_launches - launches0;
    return out;
  }
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
  std::unordered_map<const Program *, std::unique_ptr<GraphPlan>> plans;
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

  std::unique_ptr<GraphPlan> build_plan(Program &program, const HipValuation &inputs) {
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

  HipValuation run_plan(GraphPlan &plan, const HipValuation &inputs) {
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
  bool client_on_device() {
    if (client_device < 0) {
      const char *e = std::getenv("EVA_DEVICE_CLIENT");
      int n = 0;
      client_device = (!e || std::atoi(e) != 0) && evah_device_count(&n) == 0 && n > 0 ? 1 : 0;
    }
    return client_device == 1;
  }
  int client_device = -1;
  bool pk_uploaded = false;
  // same bound as HipExecutor::device_encodable: every rounded coefficient below 2^62 and inside the modulus
  bool device_encodable(const std::vector<double> &in, double scale, uint32_t limbs) const {
    const size_t slots = host->N / 2;
    if (std::getenv("EVA_DEVICE_ENCODE") && !std::atoi(std::getenv("EVA_DEVICE_ENCODE"))) return false;
    if (in.empty() || in.size() > slots || slots % in.size()) return false;
    double sum = 0;
    for (double x : in) {
      if (!std::isfinite(x)) return false;
      sum += std::fabs(x);
    }
    const double bound = 2.0 * sum * (double)(slots / in.size()) * scale / (double)host->N;
    const int bits = (int)std::ceil(std::log2(std::max(bound, 1.0))) + 1;
    return bits < 62 && bits < host->total_bits[limbs];
  }
  // coeff_pt: the host encoder's coefficient-form plaintext, or (null) values: the slot values for the device encoder
  HostCipher encrypt_on_device(const HostPlain *coeff_pt, const std::vector<double> *values, double scale, uint32_t limbs, SecureRng &rng) {
    ensure_device(false);
    if (!pk_uploaded) {
      chk(evah_client_key_upload(dev->h, EVAH_KEY_PUBLIC, (const uint64_t *)pk.data.data()));
      pk_uploaded = true;
    }
    const uint32_t N = host->N;
    std::vector<int8_t> u, e0, e1, small((size_t)3 * N);
    host->sample_ternary(rng, u);
    host->sample_error(rng, e0);
    host->sample_error(rng, e1);
    std::copy(u.begin(), u.end(), small.begin());
    std::copy(e0.begin(), e0.end(), small.begin() + N);
    std::copy(e1.begin(), e1.end(), small.begin() + 2 * (size_t)N);
    evah_pt *p = nullptr;
    if (coeff_pt) chk(evah_pt_upload_coeff(dev->h, limbs, scale, (const uint64_t *)coeff_pt->data.data(), &p));
    else chk(evah_pt_encode(dev->h, values->data(), (uint32_t)values->size(), limbs, scale, &p));
    evah_ct *c = nullptr;
    int rc = evah_encrypt(dev->h, p, small.data(), &c);
    evah_pt_free(dev->h, p);
    chk(rc);
    HostCipher out;
    out.size = 2;
    out.limbs = limbs;
    out.scale = scale;
    auto handle = std::make_shared<CtHandle>(dev->h, c);
    if (resident) { // stays in HBM; host words on demand
      out.dev = std::make_shared<DeviceResident>(DeviceResident{dev, nullptr, handle, host->N});
      return out;
    }
    out.data.resize((size_t)2 * out.limbs * N);
    out.words_checked = true;
    chk(evah_ct_download(dev->h, c, (uint64_t *)out.data.data()));
    return out;
  }

  // eval_keys = false: encryption and limb-sharded execution (whose shards hold their own rows of the
  // keys) do not need the whole evaluation keys in this device's memory
  bool eval_keys_uploaded = false;
  void ensure_device(bool eval_keys = true) {
    if (!dev) {
      if (!holder->dev) holder->dev = std::make_shared<DeviceCtx>(host->N, host->primes, device);
      dev = holder->dev; // may have been created by the secret half of the key pair (decrypt first)
    }
    if (eval_keys && !eval_keys_uploaded) {
      upload_eval_keys(dev->h);
      eval_keys_uploaded = true;
    }
  }
};

class HipSecret {
public:
  std::shared_ptr<HostContext> host;
  SecretKey sk;
  int device = 0;
  // decrypt + decode on the GPU when one is present (EVA_DEVICE_CLIENT=0: host); the secret key is
  // uploaded once, in NTT form, to a context of its own
  bool on_device() {
    if (state < 0) {
      const char *e = std::getenv("EVA_DEVICE_CLIENT");
      int n = 0;
      state = (!e || std::atoi(e) != 0) && evah_device_count(&n) == 0 && n > 0 ? 1 : 0;
      if (state == 1) {
        // the device state of the key pair (generate_keys shares one holder between both halves), so
        // that the public context's resident results are read in place
        if (!holder->dev) holder->dev = std::make_shared<DeviceCtx>(host->N, host->primes, device);
        dev = holder->dev;
        chk(evah_client_key_upload(dev->h, EVAH_KEY_SECRET, (const uint64_t *)sk.s_ntt.data()));
      }
    }
    return state == 1;
  }
  int state = -1;
  std::shared_ptr<DeviceHolder> holder = std::make_shared<DeviceHolder>();
  std::shared_ptr<DeviceCtx> dev;
  // SEALSecret::decrypt (seal.cpp:124-146)
  Valuation decrypt(const HipValuation &enc, const CKKSSignature &sig) {
    Valuation out;
    for (auto &kv : enc.values) {
      std::vector<double> v;
      if (auto *c = std::get_if<HostCipher>(&kv.second)) {
        if (on_device()) { // dot product with s, inverse transforms, recomposition and the special FFT on the GPU
          if (c->size < 1 || c->size > 3 || c->limbs < 1 || c->limbs > host->k - 1 ||
              (!resident_only(*c) && c->data.size() != (size_t)c->size * c->limbs * host->N) || (c->dev && c->dev->N != host->N))
            throw std::runtime_error("output " + kv.first + ": ciphertext shape does not match its data or the encryption parameters");
          v.resize((size_t)sig.vec_size);
          if (c->dev && c->dev->root == dev) { // resident on this key pair's device state: read in place
            chk(evah_decrypt_decode(dev->h, c->dev->h->h, (uint32_t)sig.vec_size, v.data()));
          } else {
            evah_ct *h = nullptr;
            chk(evah_ct_upload(dev->h, c->size, c->limbs, c->scale, (const uint64_t *)words(*c).data(), &h));
            int rc = evah_decrypt_decode(dev->h, h, (uint32_t)sig.vec_size, v.data());
            evah_ct_free(dev->h, h);
            chk(rc);
          }
          out[kv.first] = std::move(v);
          continue;
        }
        (void)words(*c);
        auto m = decrypt_to_coeff(*host, sk, *c);
        host->decode_coeff(m.data(), c->limbs, c->scale, v);
      } else if (auto *p = std::get_if<HostPlain>(&kv.second)) {
        std::vector<u64> m = p->data;
        for (uint32_t i = 0; i < p->limbs; i++) host->intt(i, m.data() + (size_t)i * host->N);
        host->decode_coeff(m.data(), p->limbs, p->scale, v);
      } else {
        ConstantValue{std::get<std::vector<double>>(kv.second)}.expand_to(v, (size_t)sig.vec_size);
      }
      v.resize((size_t)sig.vec_size);
      out[kv.first] = std::move(v);
    }
    return out;
  }
};

// generateKeys (seal.cpp:174-203): prime chain from bit sizes, secret/public key, one Galois key
// per exact rotation step, relinearization key.
inline std::pair<std::shared_ptr<HipPublic>, std::shared_ptr<HipSecret>>
generate_keys(const CKKSParameters &params, uint64_t seed = 0) {
  std::vector<int> bits(params.prime_bits.begin(), params.prime_bits.end());
  if (bits.size() < 2) throw std::invalid_argument("need at least two primes (data + special)");
  auto primes = evah::coeff_modulus_create(params.poly_modulus_degree, bits);
  auto host = std::make_shared<HostContext>(params.poly_modulus_degree, primes);
  KeyGenerator kg(*host, seed); // seed == 0: keyed from the OS; otherwise the reproducible test hook
  auto pub = std::make_shared<HipPublic>();
  auto sec = std::make_shared<HipSecret>();
  sec->holder = pub->holder; // one device state for the pair: results stay resident from encrypt to decrypt
  pub->host = host;
  pub->pk = kg.public_key();
  pub->relin = kg.relin_key();
  const uint32_t N = host->N, m = 2 * N;
  for (int step : params.rotations) {
    uint32_t elt;
    if (step == 0) elt = m - 1;
    else {
      uint32_t pos = step < 0 ? (uint32_t)(-(int64_t)step) : (uint32_t)step;
      if (pos >= (N >> 1)) throw std::invalid_argument("step count too large");
      uint32_t s = step < 0 ? (N >> 1) - pos : pos;
      elt = 1;
      for (uint32_t i = 0; i < s; i++) elt = (elt * 3u) & (m - 1);
    }
    if (!pub->galois.count(elt)) pub->galois.emplace(elt, kg.galois_key(elt));
  }
  sec->host = host;
  sec->sk = kg.sk;
  return {pub, sec};
}

} // namespace evahost
