// limb_exec.h — HipPublic::execute_limb: SEALPublic::execute over limb-sharded values: serial
// forwardPass, SEALExecutor's dispatch per node (seal_executor.h:279-404) on a LimbShardEvaluator (multi_device.h).
// Included by public_ctx.h.
#pragma once

namespace evahost {

// SEALPublic::execute over limb-sharded values: serial forwardPass, SEALExecutor's dispatch per node
// (seal_executor.h:279-404) on a LimbShardEvaluator.  Values come in and go out as host words (a
// sharded value has no single device handle); constants are encoded on the host once per program.
inline HipValuation HipPublic::execute_limb(Program &program, const HipValuation &inputs) {
  if (!limb || limb_ids != devices || limb->shards() != (limb_hooks ? limb_world : (uint32_t)devices.size())) {
    limb.reset();
    limb_const.clear();
    if (limb_hooks) { // this process is one shard of a group that spans processes: collectives at the exchange steps
      if (limb_world < 1 || limb_rank >= limb_world) throw std::runtime_error("limb_rank out of range");
      DeviceGroup g = make_limb_group_rank(device, limb_rank, limb_world, *host, [this](evah_ctx *c) { upload_eval_keys(c); });
      if (limb_stream) chk(evah_ctx_set_stream(g.ctx[limb_rank], (void *)limb_stream));
      limb = std::make_unique<LimbShardEvaluator>(*host, std::move(g), limb_hooks);
    } else {
      check_devices();
      limb = std::make_unique<LimbShardEvaluator>(*host, make_limb_group(devices, *host, [this](evah_ctx *c) { upload_eval_keys(c); }));
    }
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

} // namespace evahost
