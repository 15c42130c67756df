// values.h — what crosses the execute() boundary and where it lives: the plain-double reference semantics
// (reference_executor.cpp:14-115), the valuation (seal.h:21-41: name -> ciphertext / plaintext / constant; a
// ciphertext may be host words, a device handle, or both), device handles and the device contexts / issue queues
// they belong to.  Split out of executor.h (r04); included by it.
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

} // namespace evahost
