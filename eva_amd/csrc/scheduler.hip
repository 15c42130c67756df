// scheduler.hip — libeva_hip.so: evah_execute, the whole-DAG submit.  Replaces the per-node loop
// ProgramTraversal::forwardPass / MulticoreProgramTraversal::forwardPass + SEALExecutor::operator()
// (/root/reference/eva/common/program_traversal.h:36-93, multicore_program_traversal.h:24-83,
// /root/reference/eva/seal/seal_executor.h:279-404) for the encrypted part of a program.  Host code
// only: every device action goes through the entry points of the evaluator units / runtime.hip.
#include "internal.hip.h"

extern "C" {

// Whole-DAG submit over a value table (include/eva_hip.h).  Dispatch rules follow
// seal_executor.h:114-215.  The op list is scheduled level by level (depth = longest path from the
// caller-placed values): the ops of one level are independent, so its rotations, rescales,
// relinearizations and ciphertext products go out through the batched entry points; a
// Relinearize consumed only by a Rescale is evaluated with it; multiply_plain / add chains whose
// partial sums have no other reader collapse into evah_weighted_sum.  Same ciphertexts as calling
// the entry points one op at a time in list order.
int evah_execute(evah_ctx *c, const evah_op *ops, uint32_t n_ops, evah_val *tab, uint32_t n_vals) {
  struct LazySum { // unevaluated sum of products; the handles are aliases owned here
    std::vector<evah_ct *> cts;
    std::vector<evah_pt *> pts;
    uint32_t size = 0, limbs = 0;
    double scale = 0;
  };
  struct State {
    evah_ctx *c;
    std::map<uint32_t, LazySum> sums;
    std::map<uint32_t, evah_ct *> relins; // value -> alias of the size-3 operand of a deferred Relinearize
    // value -> aliases of the two operands of a deferred Mul (prods) / of a Relinearize of one (prodrel):
    // Mul -> Relinearize -> Rescale chains without other readers run as one fused call at the Rescale
    std::map<uint32_t, std::pair<evah_ct *, evah_ct *>> prods, prodrel;
    ~State() {
      for (auto &kv : sums) {
        for (evah_ct *h : kv.second.cts) evah_ct_free(c, h);
        for (evah_pt *h : kv.second.pts) evah_pt_free(c, h);
      }
      for (auto &kv : relins) evah_ct_free(c, kv.second);
      for (auto *m : {&prods, &prodrel})
        for (auto &kv : *m) { evah_ct_free(c, kv.second.first); evah_ct_free(c, kv.second.second); }
    }
  } st{c, {}, {}, {}, {}};
  auto chk = [&](int rc) {
    if (rc) throw std::runtime_error(g_err);
  };
  auto slot = [&](uint32_t i) -> evah_val & {
    if (i >= n_vals) throw std::invalid_argument("value index out of range");
    return tab[i];
  };
  auto alias_ct = [](evah_ct *a) { evah_ct *o = new evah_ct(*a); o->buf->refs++; return o; };
  auto alias_pt = [](evah_pt *a) { evah_pt *o = new evah_pt(*a); o->buf->refs++; return o; };
  auto release = [&](uint32_t i) {
    evah_val &v = tab[i];
    if (v.kind == EVAH_VAL_CT) evah_ct_free(c, static_cast<evah_ct *>(v.h));
    else if (v.kind == EVAH_VAL_PT) evah_pt_free(c, static_cast<evah_pt *>(v.h));
    v.kind = EVAH_VAL_NONE;
    v.h = nullptr;
  };
  auto put = [&](uint32_t dst, evah_ct *o) {
    tab[dst].kind = EVAH_VAL_CT;
    tab[dst].h = o;
  };
  auto drop_sum = [&](uint32_t v) {
    auto it = st.sums.find(v);
    if (it == st.sums.end()) return;
    for (evah_ct *h : it->second.cts) evah_ct_free(c, h);
    for (evah_pt *h : it->second.pts) evah_pt_free(c, h);
    st.sums.erase(it);
  };
  // a value as a device ciphertext: deferred forms are evaluated on first demand
  auto ct_of = [&](uint32_t v) -> evah_ct * {
    auto ls = st.sums.find(v);
    if (ls != st.sums.end()) {
      evah_ct *o = nullptr;
      std::vector<const evah_ct *> cc(ls->second.cts.begin(), ls->second.cts.end());
      std::vector<const evah_pt *> pp(ls->second.pts.begin(), ls->second.pts.end());
      chk(evah_weighted_sum(c, cc.data(), pp.data(), (uint32_t)cc.size(), &o));
      drop_sum(v);
      put(v, o);
    }
    auto lr = st.relins.find(v);
    if (lr != st.relins.end()) {
      evah_ct *o = nullptr;
      chk(evah_relinearize(c, lr->second, &o));
      evah_ct_free(c, lr->second);
      st.relins.erase(lr);
      put(v, o);
    }
    evah_val &x = slot(v);
    if (x.kind != EVAH_VAL_CT || !x.h) throw std::invalid_argument("operand is not a ciphertext");
    return static_cast<evah_ct *>(x.h);
  };
  auto is_ct = [&](uint32_t v) { return slot(v).kind == EVAH_VAL_CT || st.sums.count(v) || st.relins.count(v); };
  auto is_plain_ct = [&](uint32_t v) { return tab[v].kind == EVAH_VAL_CT && !st.sums.count(v) && !st.relins.count(v); };

  API_BEGIN
  use(c);
  // ---- analysis: producers, readers, levels (the list is in topological order, single assignment)
  const int NONE = -1;
  std::vector<int> producer(n_vals, NONE), only_reader(n_vals, NONE);
  std::vector<uint32_t> reads(n_vals, 0), level(n_ops, 0);
  std::vector<char> freeable(n_vals, 0);
  auto arity = [](uint32_t op) { return (op == 11 || op == 12 || op == 13) ? 2 : (op == 1 || op == 3 || op == 23) ? 0 : 1; };
  uint32_t depth = 0;
  for (uint32_t i = 0; i < n_ops; i++) {
    const evah_op &o = ops[i];
    const int na = arity(o.op);
    if (na == 0) {
      if (slot(o.dst).kind == EVAH_VAL_NONE) throw std::invalid_argument("input / plaintext slot is empty");
      continue;
    }
    if (slot(o.dst).kind != EVAH_VAL_NONE || producer[o.dst] != NONE)
      throw std::invalid_argument("every value slot is written by exactly one op (dst slots start empty)");
    const uint32_t srcs[2] = {o.src0, o.src1};
    for (int k = 0; k < na; k++) {
      const uint32_t v = srcs[k];
      if (v >= n_vals) throw std::invalid_argument("value index out of range");
      if (producer[v] == NONE && tab[v].kind == EVAH_VAL_NONE) throw std::invalid_argument("operand is used before it is produced");
      if (producer[v] != NONE) level[i] = std::max(level[i], level[producer[v]] + 1);
      only_reader[v] = reads[v] == 0 ? (int)i : -2;
      reads[v]++;
      if (o.flags & (k == 0 ? EVAH_OPF_FREE_SRC0 : EVAH_OPF_FREE_SRC1)) freeable[v] = 1;
    }
    producer[o.dst] = (int)i;
    depth = std::max(depth, level[i]);
  }
  std::vector<std::vector<uint32_t>> buckets(depth + 1);
  for (uint32_t i = 0; i < n_ops; i++)
    if (arity(ops[i].op)) buckets[level[i]].push_back(i);
  // dst is an intermediate nobody else sees and its one reader is an op of kind `by`
  auto feeds_only = [&](uint32_t dst, uint32_t by) {
    return reads[dst] == 1 && only_reader[dst] >= 0 && ops[only_reader[dst]].op == by && freeable[dst];
  };
  auto shape = [&](uint32_t v, uint32_t &size, uint32_t &limbs, double &scale) {
    auto ls = st.sums.find(v);
    if (ls != st.sums.end()) { size = ls->second.size; limbs = ls->second.limbs; scale = ls->second.scale; return; }
    chk(evah_ct_info(ct_of(v), &size, &limbs, &scale));
  };

  for (auto &lvl : buckets) {
    std::map<uint32_t, std::vector<uint32_t>> rots, relins, muls, batched_rots;
    std::map<std::pair<uint32_t, uint32_t>, std::vector<uint32_t>> mulps; // ct x pt products by (size, limbs)
    std::map<std::tuple<uint32_t, uint32_t, uint32_t>, std::vector<uint32_t>> rescales;
    std::map<std::pair<uint32_t, uint32_t>, std::vector<uint32_t>> fused, fused3;
    // ---- one op through the ordinary entry points (seal_executor.h:114-215)
    auto single = [&](const evah_op &o) {
      evah_ct *out = nullptr;
      switch (o.op) {
      case 2: { // Output: dst names the same ciphertext
        evah_val &s0 = slot(o.src0);
        if (s0.kind == EVAH_VAL_PT) { tab[o.dst].kind = EVAH_VAL_PT; tab[o.dst].h = alias_pt(static_cast<evah_pt *>(s0.h)); return; }
        out = alias_ct(ct_of(o.src0));
        break;
      }
      case 10: chk(evah_negate(c, ct_of(o.src0), &out)); break;
      case 11: case 13: { // Add / Mul: a plaintext first operand goes behind the ciphertext
        uint32_t a = o.src0, b = o.src1;
        if (!is_ct(a)) std::swap(a, b);
        if (!is_ct(a)) throw std::runtime_error("Unsupported operation encountered");
        if (is_ct(b)) {
          if (o.op == 11) chk(evah_add(c, ct_of(a), ct_of(b), &out));
          else if (a == b) chk(evah_square(c, ct_of(a), &out));
          else chk(evah_multiply(c, ct_of(a), ct_of(b), &out));
        } else if (slot(b).kind == EVAH_VAL_PT) {
          if (o.op == 11) chk(evah_add_plain(c, ct_of(a), static_cast<evah_pt *>(tab[b].h), &out));
          else chk(evah_multiply_plain(c, ct_of(a), static_cast<evah_pt *>(tab[b].h), &out));
        } else {
          throw std::runtime_error("Unsupported operation encountered");
        }
        break;
      }
      case 12:
        if (is_ct(o.src1)) chk(evah_sub(c, ct_of(o.src0), ct_of(o.src1), &out));
        else if (slot(o.src1).kind == EVAH_VAL_PT) chk(evah_sub_plain(c, ct_of(o.src0), static_cast<evah_pt *>(tab[o.src1].h), &out));
        else throw std::runtime_error("Unsupported operation encountered");
        break;
      case 14: chk(evah_rotate(c, ct_of(o.src0), o.imm, &out)); break;
      case 15: chk(evah_rotate(c, ct_of(o.src0), -o.imm, &out)); break; // seal_executor.h:188
      case 20: chk(evah_relinearize(c, ct_of(o.src0), &out)); break;
      case 21: chk(evah_mod_switch(c, ct_of(o.src0), &out)); break;
      case 22: chk(evah_rescale(c, ct_of(o.src0), (uint32_t)o.imm, &out)); break;
      default: throw std::runtime_error("Unhandled op " + std::to_string(o.op));
      }
      put(o.dst, out);
    };
    // ---- classify
    for (uint32_t i : lvl) {
      const evah_op &o = ops[i];
      uint32_t size = 0, limbs = 0;
      double scale = 0;
      // a batched handle already covers its instances in one launch set: the *_many forms take
      // single ciphertexts, so on batched operands only sibling rotations are grouped (rotate_many
      // accepts them) and the deferred forms below still apply
      auto batched_val = [&](uint32_t v) {
        return tab[v].kind == EVAH_VAL_CT && static_cast<evah_ct *>(tab[v].h)->batch > 1;
      };
      const bool batched = batched_val(o.src0) || ((o.op == 11 || o.op == 12 || o.op == 13) && batched_val(o.src1)) ||
                           (st.relins.count(o.src0) && st.relins[o.src0]->batch > 1);
      if (batched && (o.op == 14 || o.op == 15) && o.imm != 0) {
        batched_rots[o.src0].push_back(i);
        continue;
      }
      if (batched && (o.op == 22 || (o.op == 20 && !feeds_only(o.dst, 22)) || (o.op == 13 && o.src0 != o.src1 && is_ct(o.src0) && is_ct(o.src1)))) {
        if (o.op == 22 && st.relins.count(o.src0)) { // deferred relinearize + this rescale, on the batched handle
          evah_ct *out = nullptr;
          chk(evah_relinearize_rescale(c, st.relins[o.src0], (uint32_t)o.imm, &out));
          evah_ct_free(c, st.relins[o.src0]);
          st.relins.erase(o.src0);
          put(o.dst, out);
        } else {
          single(o);
        }
        continue;
      }
      if (o.op == 20 && st.prods.count(o.src0)) { // Relinearize of a deferred product: still deferred
        st.prodrel[o.dst] = st.prods[o.src0];
        st.prods.erase(o.src0);
      } else if (o.op == 22 && st.prodrel.count(o.src0)) {
        fused3[{st.prodrel[o.src0].first->limbs, (uint32_t)o.imm}].push_back(i);
      } else if ((o.op == 14 || o.op == 15) && o.imm != 0 && is_ct(o.src0)) {
        shape(o.src0, size, limbs, scale);
        rots[limbs].push_back(i);
      } else if (o.op == 22 && st.relins.count(o.src0)) {
        chk(evah_ct_info(st.relins[o.src0], &size, &limbs, &scale));
        fused[{limbs, (uint32_t)o.imm}].push_back(i);
      } else if (o.op == 22 && is_ct(o.src0)) {
        shape(o.src0, size, limbs, scale);
        rescales[{size, limbs, (uint32_t)o.imm}].push_back(i);
      } else if (o.op == 20 && is_ct(o.src0)) {
        if (feeds_only(o.dst, 22)) {
          st.relins[o.dst] = alias_ct(ct_of(o.src0)); // evaluated together with its Rescale
        } else {
          shape(o.src0, size, limbs, scale);
          relins[limbs].push_back(i);
        }
      } else if (o.op == 13 && is_ct(o.src0) && is_ct(o.src1) &&
                 (o.src0 != o.src1 || (!batched && ct_of(o.src0)->size == 2))) { // a square is the product (a, a): same residues
        // Mul read only by a Relinearize that is read only by a Rescale (the commonest CKKS
        // pattern): nothing is computed here, the three run as one fused call at the Rescale
        evah_ct *x = ct_of(o.src0), *y = ct_of(o.src1);
        const bool chain = o.src0 != o.src1 && c->tun.fuse_mac && c->tun.fuse_mul && feeds_only(o.dst, 20) && feeds_only(ops[only_reader[o.dst]].dst, 22);
        if (chain && x->size == 2 && y->size == 2 && x->limbs == y->limbs && x->limbs >= 2 && x->batch == 1 && y->batch == 1) {
          check_scale(c, x->scale * y->scale, x->limbs);
          st.prods[o.dst] = {alias_ct(x), alias_ct(y)};
        } else {
          shape(o.src0, size, limbs, scale);
          muls[limbs].push_back(i);
        }
      } else if (o.op == 13 && feeds_only(o.dst, 11) &&
                 ((is_plain_ct(o.src0) && slot(o.src1).kind == EVAH_VAL_PT) || (is_plain_ct(o.src1) && slot(o.src0).kind == EVAH_VAL_PT))) {
        const uint32_t a = is_plain_ct(o.src0) ? o.src0 : o.src1, b = a == o.src0 ? o.src1 : o.src0;
        evah_ct *x = static_cast<evah_ct *>(tab[a].h);
        evah_pt *w = static_cast<evah_pt *>(tab[b].h);
        if (w->limbs != x->limbs) { single(o); continue; } // multiply_plain reports the mismatch
        LazySum ls;
        ls.size = x->size; ls.limbs = x->limbs; ls.scale = x->scale * w->scale;
        ls.cts.push_back(alias_ct(x));
        ls.pts.push_back(alias_pt(w));
        st.sums[o.dst] = std::move(ls);
      } else if (o.op == 13 && !batched &&
                 ((is_plain_ct(o.src0) && slot(o.src1).kind == EVAH_VAL_PT) || (is_plain_ct(o.src1) && slot(o.src0).kind == EVAH_VAL_PT))) {
        // independent ciphertext x plaintext products of one shape at this level: one launch
        const uint32_t a = is_plain_ct(o.src0) ? o.src0 : o.src1, b = a == o.src0 ? o.src1 : o.src0;
        evah_ct *x = static_cast<evah_ct *>(tab[a].h);
        if (static_cast<evah_pt *>(tab[b].h)->limbs != x->limbs) { single(o); continue; } // multiply_plain reports the mismatch
        mulps[{x->size, x->limbs}].push_back(i);
      } else if (o.op == 11 && is_ct(o.src0) && is_ct(o.src1) && !st.relins.count(o.src0) && !st.relins.count(o.src1) &&
                 (st.sums.count(o.src0) || st.sums.count(o.src1) || feeds_only(o.dst, 11))) {
        uint32_t s0, l0, s1, l1;
        double c0, c1;
        shape(o.src0, s0, l0, c0);
        shape(o.src1, s1, l1, c1);
        size_t nterms = 0;
        for (uint32_t v : {o.src0, o.src1}) nterms += st.sums.count(v) ? st.sums[v].cts.size() : 1;
        if (s0 != s1 || l0 != l1 || c0 != c1 || nterms > (size_t)KS_BATCH_MAX) { single(o); continue; }
        LazySum ls;
        ls.size = s0; ls.limbs = l0; ls.scale = c0;
        for (uint32_t v : {o.src0, o.src1}) {
          auto it = st.sums.find(v);
          if (it != st.sums.end()) {
            for (evah_ct *h : it->second.cts) ls.cts.push_back(alias_ct(h));
            for (evah_pt *h : it->second.pts) ls.pts.push_back(h ? alias_pt(h) : nullptr);
          } else {
            ls.cts.push_back(alias_ct(static_cast<evah_ct *>(tab[v].h)));
            ls.pts.push_back(nullptr);
          }
        }
        st.sums[o.dst] = std::move(ls);
        if (!(feeds_only(o.dst, 11) && nterms < (size_t)KS_BATCH_MAX)) (void)ct_of(o.dst); // the chain ends here
      } else {
        single(o);
      }
    }
    // ---- the batchable kinds of this level
    auto each_chunk = [&](std::vector<uint32_t> &g, size_t cap, auto &&fn) {
      if (g.size() == 1) { single(ops[g[0]]); return; }
      for (size_t i = 0; i < g.size(); i += cap) fn(g.data() + i, (uint32_t)std::min(cap, g.size() - i));
    };
    auto store = [&](const uint32_t *is, uint32_t n, std::vector<evah_ct *> &outs) {
      for (uint32_t j = 0; j < n; j++) put(ops[is[j]].dst, outs[j]);
    };
    for (auto &kv : batched_rots)
      each_chunk(kv.second, KS_BATCH_MAX, [&](const uint32_t *is, uint32_t n) {
        std::vector<int32_t> steps(n);
        std::vector<evah_ct *> outs(n, nullptr);
        for (uint32_t j = 0; j < n; j++) steps[j] = ops[is[j]].op == 14 ? ops[is[j]].imm : -ops[is[j]].imm;
        chk(evah_rotate_many(c, ct_of(kv.first), steps.data(), n, outs.data()));
        store(is, n, outs);
      });
    for (auto &kv : rots) {
      // one launch set per level; evah_rotate_pairs shares the digit decomposition of sources that
      // occur more than once (sibling rotations of a convolution) when the set is large enough
      std::vector<uint32_t> &rest = kv.second;
      each_chunk(rest, KS_BATCH_MAX, [&](const uint32_t *is, uint32_t n) {
        std::vector<const evah_ct *> in(n);
        std::vector<int32_t> steps(n);
        std::vector<evah_ct *> outs(n, nullptr);
        for (uint32_t j = 0; j < n; j++) {
          in[j] = ct_of(ops[is[j]].src0);
          steps[j] = ops[is[j]].op == 14 ? ops[is[j]].imm : -ops[is[j]].imm;
        }
        chk(evah_rotate_pairs(c, in.data(), steps.data(), n, outs.data()));
        store(is, n, outs);
      });
    }
    for (auto &kv : fused) {
      auto fused_single = [&](uint32_t i) {
        const evah_op &o = ops[i];
        evah_ct *out = nullptr;
        chk(evah_relinearize_rescale(c, st.relins[o.src0], (uint32_t)o.imm, &out));
        evah_ct_free(c, st.relins[o.src0]);
        st.relins.erase(o.src0);
        put(o.dst, out);
      };
      if (kv.second.size() == 1) { fused_single(kv.second[0]); continue; }
      for (size_t i0 = 0; i0 < kv.second.size(); i0 += KS_BATCH_MAX) {
        const uint32_t n = (uint32_t)std::min<size_t>(KS_BATCH_MAX, kv.second.size() - i0);
        const uint32_t *is = kv.second.data() + i0;
        std::vector<const evah_ct *> in(n);
        std::vector<evah_ct *> outs(n, nullptr);
        for (uint32_t j = 0; j < n; j++) in[j] = st.relins[ops[is[j]].src0];
        chk(evah_relinearize_rescale_many(c, in.data(), n, kv.first.second, outs.data()));
        for (uint32_t j = 0; j < n; j++) {
          evah_ct_free(c, st.relins[ops[is[j]].src0]);
          st.relins.erase(ops[is[j]].src0);
        }
        store(is, n, outs);
      }
    }
    for (auto &kv : fused3)
      for (size_t i0 = 0; i0 < kv.second.size(); i0 += KS_BATCH_MAX) {
        const uint32_t n = (uint32_t)std::min<size_t>(KS_BATCH_MAX, kv.second.size() - i0);
        const uint32_t *is = kv.second.data() + i0;
        std::vector<const evah_ct *> ia(n), ib(n);
        std::vector<evah_ct *> outs(n, nullptr);
        for (uint32_t j = 0; j < n; j++) {
          ia[j] = st.prodrel[ops[is[j]].src0].first;
          ib[j] = st.prodrel[ops[is[j]].src0].second;
        }
        chk(evah_multiply_relinearize_rescale_many(c, ia.data(), ib.data(), n, kv.first.second, outs.data()));
        for (uint32_t j = 0; j < n; j++) {
          evah_ct_free(c, const_cast<evah_ct *>(ia[j]));
          evah_ct_free(c, const_cast<evah_ct *>(ib[j]));
          st.prodrel.erase(ops[is[j]].src0);
        }
        store(is, n, outs);
      }
    for (auto &kv : rescales)
      each_chunk(kv.second, (2 * KS_BATCH_MAX) / std::get<0>(kv.first), [&](const uint32_t *is, uint32_t n) {
        std::vector<const evah_ct *> in(n);
        std::vector<evah_ct *> outs(n, nullptr);
        for (uint32_t j = 0; j < n; j++) in[j] = ct_of(ops[is[j]].src0);
        chk(evah_rescale_many(c, in.data(), n, std::get<2>(kv.first), outs.data()));
        store(is, n, outs);
      });
    for (auto &kv : relins)
      each_chunk(kv.second, KS_BATCH_MAX, [&](const uint32_t *is, uint32_t n) {
        std::vector<const evah_ct *> in(n);
        std::vector<evah_ct *> outs(n, nullptr);
        for (uint32_t j = 0; j < n; j++) in[j] = ct_of(ops[is[j]].src0);
        chk(evah_relinearize_many(c, in.data(), n, outs.data()));
        store(is, n, outs);
      });
    for (auto &kv : mulps)
      each_chunk(kv.second, KS_BATCH_MAX, [&](const uint32_t *is, uint32_t n) {
        std::vector<const evah_ct *> ia(n);
        std::vector<const evah_pt *> ib(n);
        std::vector<evah_ct *> outs(n, nullptr);
        for (uint32_t j = 0; j < n; j++) {
          const evah_op &o = ops[is[j]];
          const uint32_t a = is_plain_ct(o.src0) ? o.src0 : o.src1, b = a == o.src0 ? o.src1 : o.src0;
          ia[j] = static_cast<evah_ct *>(tab[a].h);
          ib[j] = static_cast<evah_pt *>(tab[b].h);
        }
        chk(evah_multiply_plain_many(c, ia.data(), ib.data(), n, outs.data()));
        store(is, n, outs);
      });
    for (auto &kv : muls)
      each_chunk(kv.second, KS_BATCH_MAX, [&](const uint32_t *is, uint32_t n) {
        std::vector<const evah_ct *> ia(n), ib(n);
        std::vector<evah_ct *> outs(n, nullptr);
        for (uint32_t j = 0; j < n; j++) {
          ia[j] = ct_of(ops[is[j]].src0);
          ib[j] = ct_of(ops[is[j]].src1);
        }
        chk(evah_multiply_many(c, ia.data(), ib.data(), n, outs.data()));
        store(is, n, outs);
      });
    // ---- operands whose last reader has run are released (deferred forms hold their own aliases)
    for (uint32_t i : lvl) {
      const evah_op &o = ops[i];
      const uint32_t srcs[2] = {o.src0, o.src1};
      for (int k = 0; k < arity(o.op); k++) {
        const uint32_t v = srcs[k];
        if (--reads[v] == 0 && freeable[v]) {
          drop_sum(v);
          auto lr = st.relins.find(v);
          if (lr != st.relins.end()) { evah_ct_free(c, lr->second); st.relins.erase(lr); }
          if (tab[v].kind != EVAH_VAL_NONE) release(v);
        }
      }
    }
  }
  API_END
}

} // extern "C"
