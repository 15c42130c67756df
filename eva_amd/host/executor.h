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
#include <cstdio>
#include <functional>
#include <memory>
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

// Per-term dispatcher: Term -> one libeva_hip call (SEALExecutor::operator(), :279-404)
class HipExecutor {
public:
  using RuntimeValue = std::variant<std::monostate, std::shared_ptr<CtHandle>, std::shared_ptr<PtHandle>, std::vector<double>>;

  HipExecutor(Program &g, const HostContext &hc, evah_ctx *dev) : program(g), host(hc), ctx(dev), objects(g.size()) {
    if (program.vec_size() > host.N / 2) throw std::runtime_error("Vector size cannot be larger than slot count");
  }

  // seal_executor.h:264-277 (the reference deep-copies; here inputs are uploaded to HBM)
  void set_inputs(const HipValuation &inputs) {
    for (auto &kv : inputs.values) {
      TermId t = program.input(kv.first);
      if (auto *c = std::get_if<HostCipher>(&kv.second)) {
        evah_ct *h = nullptr;
        chk(evah_ct_upload(ctx, c->size, c->limbs, c->scale, (const uint64_t *)c->data.data(), &h));
        objects[t] = std::make_shared<CtHandle>(ctx, h);
      } else if (auto *p = std::get_if<HostPlain>(&kv.second)) {
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
    switch (x.op) {
    case Op::Constant: {
      std::vector<double> v;
      x.constant->expand_to(v, program.vec_size());
      objects[t] = std::move(v);
    } break;
    case Op::Encode: objects[t] = encode_raw(a[0], x.encode_scale, x.encode_level); break;
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
        // rightRotate passes the negated step (seal_executor.h:188)
        evah_ct *h = nullptr;
        chk(evah_rotate(ctx, ct(a[0]), x.op == Op::RotateLeftConst ? x.rotation : -x.rotation, &h));
        objects[t] = std::make_shared<CtHandle>(ctx, h);
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
    case Op::Relinearize: {
      evah_ct *h = nullptr;
      chk(evah_relinearize(ctx, ct(a[0]), &h));
      objects[t] = std::make_shared<CtHandle>(ctx, h);
    } break;
    case Op::ModSwitch: {
      evah_ct *h = nullptr;
      chk(evah_mod_switch(ctx, ct(a[0]), &h));
      objects[t] = std::make_shared<CtHandle>(ctx, h);
    } break;
    case Op::Rescale: {
      evah_ct *h = nullptr;
      chk(evah_rescale(ctx, ct(a[0]), x.rescale_divisor, &h));
      objects[t] = std::make_shared<CtHandle>(ctx, h);
    } break;
    case Op::Output: objects[t] = objects[a[0]]; break;
    default: throw std::runtime_error(std::string("Unhandled op ") + op_name(x.op));
    }
  }

  // seal_executor.h:406-418 — release a value whose last consumer has run
  void free(TermId t) {
    if (program.at(t).op == Op::Output) return;
    objects[t] = std::monostate{};
  }

  // seal_executor.h:420-435 — outputs are downloaded into host values
  void get_outputs(HipValuation &out) {
    for (auto &kv : program.outputs()) {
      auto &o = objects[kv.second];
      if (auto *c = std::get_if<std::shared_ptr<CtHandle>>(&o)) {
        HostCipher hc;
        chk(evah_ct_info((*c)->h, &hc.size, &hc.limbs, &hc.scale));
        hc.data.resize((size_t)hc.size * hc.limbs * host.N);
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
  evah_ctx *ctx;
  std::vector<RuntimeValue> objects;
  std::vector<double> scratch;

  bool is_cipher(TermId t) const { return std::holds_alternative<std::shared_ptr<CtHandle>>(objects[t]); }
  bool is_plain(TermId t) const { return std::holds_alternative<std::shared_ptr<PtHandle>>(objects[t]); }
  bool is_raw(TermId t) const { return std::holds_alternative<std::vector<double>>(objects[t]); }
  const std::vector<double> &raw(TermId t) const { return std::get<std::vector<double>>(objects[t]); }
  evah_ct *ct(TermId t) const {
    auto *p = std::get_if<std::shared_ptr<CtHandle>>(&objects[t]);
    if (!p) throw std::runtime_error("Unsupported operation encountered");
    return (*p)->h;
  }
  evah_pt *pt(TermId t) const { return std::get<std::shared_ptr<PtHandle>>(objects[t])->h; }

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
template <class Exec> void run_counted(Program &p, Exec &ex) {
  auto order = p.topo_order();
  std::vector<uint32_t> succ(p.size(), 0);
  for (TermId t : order)
    for (TermId o : p.at(t).operands) succ[o]++;
  for (TermId t : order) {
    ex(t);
    for (TermId o : p.at(t).operands)
      if (--succ[o] == 0) ex.free(o);
  }
}

// ---- contexts (seal.h:45-97)
struct DeviceCtx {
  evah_ctx *h = nullptr;
  DeviceCtx(uint32_t N, const std::vector<u64> &primes, int device) {
    chk(evah_ctx_create(N, (uint32_t)primes.size(), (const uint64_t *)primes.data(), device, &h));
  }
  ~DeviceCtx() { evah_ctx_destroy(h); }
  DeviceCtx(const DeviceCtx &) = delete;
  DeviceCtx &operator=(const DeviceCtx &) = delete;
};

class HipPublic {
public:
  std::shared_ptr<HostContext> host;
  PublicKey pk;
  SwitchKey relin;
  std::map<uint32_t, SwitchKey> galois; // by Galois element
  int device = 0;
  bool free_eagerly = true;

  // SEALPublic::encrypt (seal.cpp:24-102)
  HipValuation encrypt(const Valuation &inputs, const CKKSSignature &sig) {
    const size_t slots = host->N / 2;
    if (slots < (size_t)sig.vec_size) throw std::runtime_error("Vector size cannot be larger than slot count");
    if (slots % sig.vec_size) throw std::runtime_error("Vector size must exactly divide the slot count");
    HipValuation out;
    std::mt19937_64 rng(std::random_device{}());
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
        pt.data.resize((size_t)pt.limbs * host->N);
        std::vector<double> vec(slots);
        for (size_t r = 0; r < slots / v.size(); r++) std::copy(v.begin(), v.end(), vec.begin() + r * v.size());
        host->encode_coeff(vec.data(), pt.scale, pt.limbs, pt.data.data());
        for (uint32_t i = 0; i < pt.limbs; i++) host->ntt(i, pt.data.data() + (size_t)i * host->N);
        if (info.input_type == Type::Cipher) out.values[kv.first] = evahost::encrypt(*host, pk, pt, rng);
        else out.values[kv.first] = std::move(pt);
      } else {
        out.values[kv.first] = v;
      }
    }
    return out;
  }

  // SEALPublic::execute (seal.cpp:104-122) — THE hot path: upload inputs, walk the DAG issuing
  // HIP work, download outputs.
  HipValuation execute(Program &program, const HipValuation &inputs) {
    ensure_device();
    HipExecutor ex(program, *host, dev->h);
    ex.set_inputs(inputs);
    if (free_eagerly) run_counted(program, ex);
    else run_serial(program, ex);
    HipValuation out;
    ex.get_outputs(out);
    return out;
  }

  evah_ctx *device_ctx() {
    ensure_device();
    return dev->h;
  }

private:
  std::shared_ptr<DeviceCtx> dev;
  void ensure_device() {
    if (dev) return;
    dev = std::make_shared<DeviceCtx>(host->N, host->primes, device);
    chk(evah_key_upload(dev->h, EVAH_KEY_RELIN, 0, relin.n_digits, (const uint64_t *)relin.data.data()));
    for (auto &kv : galois)
      chk(evah_key_upload(dev->h, EVAH_KEY_GALOIS, kv.first, kv.second.n_digits, (const uint64_t *)kv.second.data.data()));
  }
};

class HipSecret {
public:
  std::shared_ptr<HostContext> host;
  SecretKey sk;
  // SEALSecret::decrypt (seal.cpp:124-146)
  Valuation decrypt(const HipValuation &enc, const CKKSSignature &sig) {
    Valuation out;
    for (auto &kv : enc.values) {
      std::vector<double> v;
      if (auto *c = std::get_if<HostCipher>(&kv.second)) {
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
  if (!seed) seed = ((uint64_t)std::random_device{}() << 32) ^ std::random_device{}();
  KeyGenerator kg(*host, seed);
  auto pub = std::make_shared<HipPublic>();
  auto sec = std::make_shared<HipSecret>();
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
