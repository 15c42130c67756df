// scheduler.hip — libeva_hip.so: evah_execute, the whole-DAG submit.  Replaces the per-node loop
// ProgramTraversal::forwardPass / MulticoreProgramTraversal::forwardPass + SEALExecutor::operator()
// (/root/reference/eva/common/program_traversal.h:36-93, multicore_program_traversal.h:24-83,
// /root/reference/eva/seal/seal_executor.h:279-404) for the encrypted part of a program.  Host code
// only: every device action goes through the entry points of the evaluator units / runtime.hip.
#include "internal.hip.h"
#include <set>

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
    // term j is rotate(cts[j], steps[j]) (0: cts[j] itself); rv[j] = the value slot of the deferred rotation it
    // came from (NONE_V for ordinary terms)
    std::vector<int32_t> steps;
    std::vector<uint32_t> rv;
    uint32_t size = 0, limbs = 0;
    double scale = 0;
  };
  constexpr uint32_t NONE_V = ~0u;
  struct DRot { evah_ct *src; int32_t step; }; // alias of the source
  // An elementwise op (Negate / Add / Sub / Mul on ciphertexts and plaintexts) whose result nobody has needed as a stored
  // ciphertext yet: operands are other such nodes, or aliases of handles.  When a consumer that is not elementwise (or
  // the end of the walk) needs the value, every unevaluated node below it goes out as ONE evah_elementwise_program; only
  // the nodes somebody outside that program reads are stored.
  struct Expr {
    evah_ctx *c;
    uint32_t op, v;                    // Op code; the value slot this node defines
    std::shared_ptr<Expr> ea, eb;      // operand nodes (null: the operand is a handle)
    evah_ct *ca = nullptr, *cb = nullptr; // aliases owned here
    evah_pt *pa = nullptr, *pb = nullptr;
    bool same = false;                 // Mul(a, a): square
    uint32_t size = 0, limbs = 0, batch = 1;
    uint32_t depth = 1;                // longest chain of unevaluated nodes ending here (bounded: EXPR_MAX_DEPTH)
    double scale = 0;
    evah_ct *result = nullptr;         // alias, once the node has been evaluated and stored
    ~Expr() {
      for (evah_ct *h : {ca, cb, result}) if (h) evah_ct_free(c, h);
      for (evah_pt *h : {pa, pb}) if (h) evah_pt_free(c, h);
    }
  };
  struct State {
    evah_ctx *c;
    std::map<uint32_t, LazySum> sums;
    std::map<uint32_t, evah_ct *> relins; // value -> alias of the size-3 operand of a deferred Relinearize
    // value -> aliases of the two operands of a deferred Mul (prods) / of a Relinearize of one (prodrel):
    // Mul -> Relinearize -> Rescale chains without other readers run as one fused call at the Rescale
    std::map<uint32_t, std::pair<evah_ct *, evah_ct *>> prods, prodrel;
    // r6, the other order — Mul -> Rescale -> Relinearize, what lazy relinearization gives a product under the waterline
    // rescalers: prods2 = the operands of a deferred Mul read only by such a Rescale, prodres = of that Rescale (with its
    // divisor); the three run as one evah_multiply_rescale_relinearize_many at the Relinearize
    std::map<uint32_t, std::pair<evah_ct *, evah_ct *>> prods2, prodres;
    std::map<uint32_t, uint32_t> resdiv;
    // value -> alias of the size-3 operand of a deferred Rescale that only a Relinearize reads (a SUM of products under lazy
    // relinearization): the two run as one evah_rescale_relinearize_many at the Relinearize (r6)
    std::map<uint32_t, evah_ct *> resrel;
    // value -> deferred rotation: a Rotate whose readers are all ciphertext x plaintext products inside sums is not
    // evaluated; the sums carry (source, step) terms and go to evah_rotate_weighted_sums (the convolution window)
    std::map<uint32_t, DRot> drots;
    // rotations that sums outside one window share after all: evaluated once, kept until the end of the call
    std::map<uint32_t, evah_ct *> mat;
    // value -> unevaluated elementwise expression (EVAH_EW_FUSE)
    std::map<uint32_t, std::shared_ptr<Expr>> exprs;
    ~State() {
      exprs.clear();
      for (auto &kv : drots) evah_ct_free(c, kv.second.src);
      for (auto &kv : mat) evah_ct_free(c, kv.second);
      for (auto &kv : sums) {
        for (evah_ct *h : kv.second.cts) evah_ct_free(c, h);
        for (evah_pt *h : kv.second.pts) evah_pt_free(c, h);
      }
      for (auto &kv : relins) evah_ct_free(c, kv.second);
      for (auto &kv : resrel) evah_ct_free(c, kv.second);
      for (auto *m : {&prods, &prodrel, &prods2, &prodres})
        for (auto &kv : *m) { evah_ct_free(c, kv.second.first); evah_ct_free(c, kv.second.second); }
    }
  } st{c, {}, {}, {}, {}, {}, {}, {}, {}, {}, {}, {}};
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
  // ---- sums.  vals: LazySum values to evaluate now.  Sums whose terms are rotations deferred from this walk
  // (convolution windows) go out as ONE evah_rotate_weighted_sums: sums with the same term list — the two filters of
  // convolutionXY over one set of rotations — share a window, provided nobody else reads those rotations.
  std::vector<uint32_t> n_reads; // filled by the analysis below; reads[] counts down, this does not
  std::function<void(const std::vector<uint32_t> &)> eval_sums = [&](const std::vector<uint32_t> &vals) {
    auto same_terms = [&](const LazySum &a, const LazySum &b) {
      if (a.cts.size() != b.cts.size() || a.limbs != b.limbs) return false;
      for (size_t j = 0; j < a.cts.size(); j++)
        if (a.cts[j]->d != b.cts[j]->d || a.cts[j]->ps != b.cts[j]->ps || a.steps[j] != b.steps[j] || a.rv[j] != b.rv[j]) return false;
      return true;
    };
    auto rotated = [&](const LazySum &ls) {
      for (int32_t sstep : ls.steps) if (sstep != 0) return true;
      return false;
    };
    std::vector<std::vector<uint32_t>> groups; // windows: one or two sums over the same terms
    std::vector<uint32_t> plain;               // evaluated by evah_weighted_sum
    for (uint32_t v : vals) {
      LazySum &ls = st.sums.at(v);
      if (!rotated(ls)) { plain.push_back(v); continue; }
      bool placed = false;
      for (auto &g : groups)
        if (g.size() < 2 && same_terms(st.sums.at(g[0]), ls)) { g.push_back(v); placed = true; break; }
      if (!placed) groups.push_back({v});
    }
    // a window may skip the rotated ciphertexts only if its sums are the only readers of those rotations
    std::vector<std::vector<uint32_t>> fusedg;
    for (auto &g : groups) {
      std::map<uint32_t, uint32_t> occ;
      for (uint32_t v : g)
        for (uint32_t r : st.sums.at(v).rv) if (r != NONE_V) occ[r]++;
      bool exclusive = true;
      for (auto &kv : occ) exclusive = exclusive && !st.mat.count(kv.first) && kv.second == n_reads[kv.first];
      if (exclusive) fusedg.push_back(g);
      else for (uint32_t v : g) plain.push_back(v);
    }
    // windows of one (level, batch count) per call
    while (!fusedg.empty()) {
      const LazySum &f0 = st.sums.at(fusedg[0][0]);
      std::vector<std::vector<uint32_t>> now, later;
      for (auto &g : fusedg) {
        const LazySum &ls = st.sums.at(g[0]);
        (ls.limbs == f0.limbs && ls.cts[0]->batch == f0.cts[0]->batch ? now : later).push_back(g);
      }
      std::vector<const evah_ct *> cc;
      std::vector<int32_t> ss;
      std::vector<const evah_pt *> pp;
      std::vector<uint32_t> wt, ws, order;
      for (auto &g : now) {
        const LazySum &ls = st.sums.at(g[0]);
        wt.push_back((uint32_t)ls.cts.size());
        ws.push_back((uint32_t)g.size());
        cc.insert(cc.end(), ls.cts.begin(), ls.cts.end());
        ss.insert(ss.end(), ls.steps.begin(), ls.steps.end());
        for (uint32_t v : g) {
          const LazySum &m = st.sums.at(v);
          pp.insert(pp.end(), m.pts.begin(), m.pts.end());
          order.push_back(v);
        }
      }
      std::vector<evah_ct *> outs(order.size(), nullptr);
      chk(evah_rotate_weighted_sums(c, cc.data(), ss.data(), wt.data(), ws.data(), (uint32_t)wt.size(), pp.data(), outs.data()));
      for (size_t i = 0; i < order.size(); i++) {
        drop_sum(order[i]);
        put(order[i], outs[i]);
      }
      fusedg.swap(later);
    }
    // everything else: rotations that are shared beyond a window are evaluated once (st.mat), then the ordinary sums
    {
      std::vector<uint32_t> need;
      for (uint32_t v : plain) {
        const LazySum &ls = st.sums.at(v);
        for (size_t j = 0; j < ls.cts.size(); j++)
          if (ls.steps[j] != 0 && !st.mat.count(ls.rv[j]) && std::find(need.begin(), need.end(), ls.rv[j]) == need.end()) need.push_back(ls.rv[j]);
      }
      std::map<uint32_t, std::pair<const evah_ct *, int32_t>> how;
      for (uint32_t v : plain) {
        const LazySum &ls = st.sums.at(v);
        for (size_t j = 0; j < ls.cts.size(); j++)
          if (ls.steps[j] != 0) how[ls.rv[j]] = {ls.cts[j], ls.steps[j]};
      }
      std::vector<uint32_t> singles, pairs;
      for (uint32_t r : need) (how[r].first->batch > 1 ? singles : pairs).push_back(r);
      for (uint32_t r : singles) {
        evah_ct *o = nullptr;
        chk(evah_rotate(c, how[r].first, how[r].second, &o));
        st.mat[r] = o;
      }
      for (size_t i0 = 0; i0 < pairs.size(); i0 += KS_BATCH_MAX) {
        const uint32_t n = (uint32_t)std::min<size_t>(KS_BATCH_MAX, pairs.size() - i0);
        std::vector<const evah_ct *> in(n);
        std::vector<int32_t> stp(n);
        std::vector<evah_ct *> outs(n, nullptr);
        for (uint32_t j = 0; j < n; j++) { in[j] = how[pairs[i0 + j]].first; stp[j] = how[pairs[i0 + j]].second; }
        const bool same_limbs = std::all_of(in.begin(), in.end(), [&](const evah_ct *a) { return a->limbs == in[0]->limbs; });
        if (n > 1 && same_limbs) {
          chk(evah_rotate_pairs(c, in.data(), stp.data(), n, outs.data()));
        } else {
          for (uint32_t j = 0; j < n; j++) chk(evah_rotate(c, in[j], stp[j], &outs[j]));
        }
        for (uint32_t j = 0; j < n; j++) st.mat[pairs[i0 + j]] = outs[j];
      }
      for (uint32_t v : plain) {
        const LazySum &ls = st.sums.at(v);
        std::vector<const evah_ct *> cc(ls.cts.begin(), ls.cts.end());
        std::vector<const evah_pt *> pp(ls.pts.begin(), ls.pts.end());
        for (size_t j = 0; j < cc.size(); j++)
          if (ls.steps[j] != 0) cc[j] = st.mat.at(ls.rv[j]);
        evah_ct *o = nullptr;
        chk(evah_weighted_sum(c, cc.data(), pp.data(), (uint32_t)cc.size(), &o));
        drop_sum(v);
        put(v, o);
      }
    }
  };
  // a value as a device ciphertext: deferred forms are evaluated on first demand
  std::function<void(const std::vector<uint32_t> &)> force_exprs; // defined below the analysis (needs readers[])
  auto ct_of = [&](uint32_t v) -> evah_ct * {
    if (st.exprs.count(v)) force_exprs({v});
    if (st.sums.count(v)) eval_sums({v});
    auto dr = st.drots.find(v);
    if (dr != st.drots.end()) { // a deferred rotation somebody needs as a ciphertext after all
      evah_ct *o = nullptr;
      chk(evah_rotate(c, dr->second.src, dr->second.step, &o));
      evah_ct_free(c, dr->second.src);
      st.drots.erase(dr);
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
  auto is_ct = [&](uint32_t v) { return slot(v).kind == EVAH_VAL_CT || st.sums.count(v) || st.relins.count(v) || st.drots.count(v) || st.exprs.count(v); };
  auto is_plain_ct = [&](uint32_t v) { return tab[v].kind == EVAH_VAL_CT && !st.sums.count(v) && !st.relins.count(v); };

  API_BEGIN
  use(c);
  // ---- analysis: producers, readers, levels (the list is in topological order, single assignment)
  const int NONE = -1;
  std::vector<int> producer(n_vals, NONE), only_reader(n_vals, NONE);
  std::vector<uint32_t> reads(n_vals, 0), level(n_ops, 0);
  std::vector<std::vector<uint32_t>> readers(n_vals);
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
      readers[v].push_back(i);
      if (o.flags & (k == 0 ? EVAH_OPF_FREE_SRC0 : EVAH_OPF_FREE_SRC1)) freeable[v] = 1;
    }
    producer[o.dst] = (int)i;
    depth = std::max(depth, level[i]);
  }
  n_reads = reads;
  std::vector<std::vector<uint32_t>> buckets(depth + 1);
  for (uint32_t i = 0; i < n_ops; i++)
    if (arity(ops[i].op)) buckets[level[i]].push_back(i);
  // dst is an intermediate nobody else sees and its one reader is an op of kind `by`
  auto feeds_only = [&](uint32_t dst, uint32_t by) {
    return reads[dst] == 1 && only_reader[dst] >= 0 && ops[only_reader[dst]].op == by && freeable[dst];
  };
  // a Rotate all of whose readers are ciphertext x plaintext products that only feed additions (the taps of a
  // convolution window): the rotated ciphertext need not exist (evah_rotate_weighted_sums)
  auto window_rotation = [&](const evah_op &o) {
    if (!c->tun.win_fuse || !freeable[o.dst] || readers[o.dst].empty()) return false;
    for (uint32_t ri : readers[o.dst]) {
      const evah_op &m = ops[ri];
      if (m.op != 13 || m.src0 == m.src1) return false;
      const uint32_t other = m.src0 == o.dst ? m.src1 : m.src0;
      if (slot(other).kind != EVAH_VAL_PT || !feeds_only(m.dst, 11)) return false;
    }
    return true;
  };
  auto shape = [&](uint32_t v, uint32_t &size, uint32_t &limbs, double &scale) {
    auto ls = st.sums.find(v);
    if (ls != st.sums.end()) { size = ls->second.size; limbs = ls->second.limbs; scale = ls->second.scale; return; }
    auto ex = st.exprs.find(v);
    if (ex != st.exprs.end()) { size = ex->second->size; limbs = ex->second->limbs; scale = ex->second->scale; return; }
    chk(evah_ct_info(ct_of(v), &size, &limbs, &scale));
  };

  // ---- elementwise expressions (Expr above; include/eva_hip.h evah_elementwise_program).
  // Evaluate the unevaluated nodes below `roots`: one program per (level, batch size); a node is stored when it is a
  // root or when an op outside the program reads it (a later level's op, elementwise or not).
  force_exprs = [&](const std::vector<uint32_t> &roots) {
    std::vector<std::shared_ptr<Expr>> order; // operands first
    std::set<const Expr *> seen;
    std::function<void(const std::shared_ptr<Expr> &)> visit = [&](const std::shared_ptr<Expr> &n) {
      if (!n || n->result || seen.count(n.get())) return;
      seen.insert(n.get());
      visit(n->ea);
      visit(n->eb);
      order.push_back(n);
    };
    std::set<uint32_t> root_vals;
    for (uint32_t v : roots) {
      auto it = st.exprs.find(v);
      if (it == st.exprs.end()) continue;
      root_vals.insert(v);
      visit(it->second);
    }
    if (order.empty()) return;
    std::set<uint32_t> in_prog;
    for (auto &n : order) in_prog.insert(n->v);
    std::map<std::pair<uint32_t, uint32_t>, std::vector<std::shared_ptr<Expr>>> classes;
    for (auto &n : order) classes[{n->limbs, n->batch}].push_back(n);
    for (auto &kv : classes) {
      const auto &nodes = kv.second;
      std::vector<evah_val> in;
      std::map<const void *, uint32_t> in_idx;
      auto input = [&](uint32_t kind, void *h) {
        auto it = in_idx.find(h);
        if (it != in_idx.end()) return it->second;
        in.push_back(evah_val{kind, h});
        return in_idx[h] = (uint32_t)in.size() - 1;
      };
      // operand `which` of node n: an earlier node of this program (index into nodes, tagged), or an input
      std::map<const Expr *, uint32_t> pos;
      for (uint32_t j = 0; j < nodes.size(); j++) pos[nodes[j].get()] = j;
      constexpr uint32_t NODE = 0x80000000u;
      auto operand = [&](const Expr &n, int which) -> uint32_t {
        const std::shared_ptr<Expr> &e = which ? n.eb : n.ea;
        evah_ct *ch = which ? n.cb : n.ca;
        evah_pt *ph = which ? n.pb : n.pa;
        if (e) {
          if (e->result) return input(EVAH_VAL_CT, e->result);
          auto it = pos.find(e.get());
          if (it == pos.end()) throw std::logic_error("elementwise expression: an operand was dropped before it was evaluated");
          return NODE | it->second;
        }
        if (ch) return input(EVAH_VAL_CT, ch);
        if (ph) return input(EVAH_VAL_PT, ph);
        throw std::logic_error("elementwise expression without an operand");
      };
      std::vector<std::pair<uint32_t, uint32_t>> refs(nodes.size());
      for (uint32_t j = 0; j < nodes.size(); j++) {
        const Expr &n = *nodes[j];
        refs[j].first = operand(n, 0);
        refs[j].second = n.op == 10 ? refs[j].first : (n.same ? refs[j].first : operand(n, 1));
      }
      const uint32_t n_in = (uint32_t)in.size();
      std::vector<evah_ew_op> eops(nodes.size());
      auto idx = [&](uint32_t r) { return (r & NODE) ? n_in + (r & ~NODE) : r; };
      for (uint32_t j = 0; j < nodes.size(); j++) eops[j] = evah_ew_op{nodes[j]->op, idx(refs[j].first), idx(refs[j].second)};
      std::vector<uint32_t> out_vals, out_node;
      for (uint32_t j = 0; j < nodes.size(); j++) {
        const Expr &n = *nodes[j];
        bool store = root_vals.count(n.v) != 0;
        for (uint32_t ri : readers[n.v]) store = store || !in_prog.count(ops[ri].dst);
        if (store) { out_vals.push_back(n_in + j); out_node.push_back(j); }
      }
      if (out_vals.empty()) continue; // (dead code: nobody reads any of it)
      std::vector<evah_ct *> outs(out_vals.size(), nullptr);
      // nothing but independent ciphertext products of stored operands, all of them kept (the products of one level): the
      // dedicated kernel does that in one launch already, at half the interpreter's latency
      bool only_products = out_vals.size() == nodes.size() && nodes.size() <= (size_t)KS_BATCH_MAX && kv.first.second == 1;
      for (auto &n : nodes) only_products = only_products && n->op == 13 && n->ca && (n->same || n->cb) && !n->ea && !n->eb;
      if (only_products) {
        std::vector<const evah_ct *> ia(nodes.size()), ib(nodes.size());
        for (size_t j = 0; j < nodes.size(); j++) { ia[j] = nodes[j]->ca; ib[j] = nodes[j]->same ? nodes[j]->ca : nodes[j]->cb; }
        if (nodes.size() == 1) {
          if (nodes[0]->same) chk(evah_square(c, ia[0], &outs[0]));
          else chk(evah_multiply(c, ia[0], ib[0], &outs[0]));
        } else {
          chk(evah_multiply_many(c, ia.data(), ib.data(), (uint32_t)nodes.size(), outs.data()));
        }
      } else {
        chk(evah_elementwise_program(c, in.data(), n_in, eops.data(), (uint32_t)eops.size(), out_vals.data(), (uint32_t)out_vals.size(), outs.data()));
      }
      for (size_t k = 0; k < outs.size(); k++) {
        Expr &n = *nodes[out_node[k]];
        n.result = alias_ct(outs[k]);
        if (reads[n.v] > 0 || !freeable[n.v]) put(n.v, outs[k]); // still has readers: the table owns it like any other value
        else evah_ct_free(c, outs[k]);
      }
    }
    for (auto &n : order) st.exprs.erase(n->v);
  };
  // Record op `o` as an unevaluated expression node instead of running it?  Only when its result is an intermediate of
  // this walk (freeable, read by somebody), its operands are plain handles or expression nodes, and every check of the
  // entry point it stands for passes — anything else takes the ordinary path, which reports the error.
  std::function<bool(const evah_op &)> try_defer_ew = [&](const evah_op &o) -> bool {
    if (!c->tun.ew_fuse || !(o.op == 10 || o.op == 11 || o.op == 12 || o.op == 13)) return false;
    if (!freeable[o.dst] || n_reads[o.dst] == 0) return false;
    struct Opnd {
      int kind = 0; // 1 ciphertext handle, 2 plaintext handle, 3 expression
      evah_ct *ct = nullptr;
      evah_pt *pt = nullptr;
      std::shared_ptr<Expr> e;
      uint32_t size = 1, limbs = 0, batch = 1;
      double scale = 0;
    };
    auto operand = [&](uint32_t v, Opnd &x) {
      if (v >= n_vals) return false;
      auto it = st.exprs.find(v);
      if (it != st.exprs.end()) {
        x.kind = 3; x.e = it->second;
        x.size = x.e->size; x.limbs = x.e->limbs; x.batch = x.e->batch; x.scale = x.e->scale;
        return true;
      }
      if (st.sums.count(v) || st.relins.count(v) || st.drots.count(v) || st.prods.count(v) || st.prodrel.count(v) || st.prods2.count(v) || st.prodres.count(v) || st.resrel.count(v)) return false;
      if (tab[v].kind == EVAH_VAL_CT && tab[v].h) {
        x.kind = 1; x.ct = static_cast<evah_ct *>(tab[v].h);
        x.size = x.ct->size; x.limbs = x.ct->limbs; x.batch = x.ct->batch; x.scale = x.ct->scale;
        return true;
      }
      if (tab[v].kind == EVAH_VAL_PT && tab[v].h) {
        x.kind = 2; x.pt = static_cast<evah_pt *>(tab[v].h);
        x.limbs = x.pt->limbs; x.scale = x.pt->scale;
        return true;
      }
      return false;
    };
    Opnd a, b;
    if (!operand(o.src0, a)) return false;
    const bool unary = o.op == 10;
    if (!unary && !operand(o.src1, b)) return false;
    if (!unary && o.op != 12 && a.kind == 2) std::swap(a, b); // the ciphertext first (seal_executor.h:116-119, :155-158)
    if (a.kind == 2) return false;
    auto node = std::make_shared<Expr>();
    node->c = c; node->op = o.op; node->v = o.dst;
    node->limbs = a.limbs; node->batch = a.batch;
    if (unary) {
      node->size = a.size; node->scale = a.scale;
    } else if (b.kind == 2) {
      if (a.limbs != b.limbs) return false;
      node->size = a.size;
      if (o.op == 13) {
        node->scale = a.scale * b.scale;
        if (!(node->scale > 0) || (int)std::log2(node->scale) >= c->total_bits[a.limbs]) return false;
      } else {
        if (!same_scale(a.scale, b.scale)) return false;
        node->scale = a.scale;
      }
    } else {
      if (a.limbs != b.limbs || a.batch != b.batch) return false;
      if (o.op == 13) {
        if (a.size != 2 || b.size != 2) return false;
        node->same = o.src0 == o.src1;
        // Mul -> Relinearize -> Rescale without other readers is one fused key-switch call (below): not an expression
        if (!node->same && c->tun.fuse_mac && c->tun.fuse_mul && feeds_only(o.dst, 20) && feeds_only(ops[only_reader[o.dst]].dst, 22) &&
            a.limbs >= 2 && a.batch == 1)
          return false;
        node->size = 3;
        node->scale = a.scale * b.scale;
        if (!(node->scale > 0) || (int)std::log2(node->scale) >= c->total_bits[a.limbs]) return false;
      } else {
        if (!same_scale(a.scale, b.scale)) return false;
        node->size = std::max(a.size, b.size);
        node->scale = a.scale;
      }
    }
    auto take = [&](Opnd &x, std::shared_ptr<Expr> &e, evah_ct *&ch, evah_pt *&ph) {
      if (x.kind == 3) e = x.e;
      else if (x.kind == 1) ch = alias_ct(x.ct);
      else ph = alias_pt(x.pt);
    };
    take(a, node->ea, node->ca, node->pa);
    if (!unary && !node->same) take(b, node->eb, node->cb, node->pb);
    // r5 advisor: nothing bounded the depth of an expression graph — force_exprs' visit and the shared_ptr destructor
    // chain recurse once per dependent node, and a program beyond EW_MAX_INS instructions runs as separate calls anyway.
    // A chain this long is evaluated up to here; the node then starts a new expression on stored operands.
    constexpr uint32_t EXPR_MAX_DEPTH = 48;
    for (const auto &e : {node->ea, node->eb})
      if (e && !e->result) node->depth = std::max(node->depth, e->depth + 1);
    if (node->depth > EXPR_MAX_DEPTH) {
      std::vector<uint32_t> below;
      for (const auto &e : {node->ea, node->eb})
        if (e && !e->result) below.push_back(e->v);
      node.reset(); // (its aliases go; the operands' nodes stay in st.exprs until forced)
      force_exprs(below);
      return try_defer_ew(o); // operands are stored handles now: depth 1
    }
    st.exprs[o.dst] = std::move(node);
    return true;
  };

  for (auto &lvl : buckets) {
    std::map<uint32_t, std::vector<uint32_t>> rots, relins, muls, batched_rots;
    std::map<std::pair<uint32_t, uint32_t>, std::vector<uint32_t>> mulps; // ct x pt products by (size, limbs)
    std::map<std::tuple<uint32_t, uint32_t, uint32_t>, std::vector<uint32_t>> rescales;
    std::map<std::pair<uint32_t, uint32_t>, std::vector<uint32_t>> fused, fused3;
    std::map<std::tuple<uint32_t, uint32_t, uint32_t>, std::vector<uint32_t>> fused3b, fused_rr; // (limbs, divisor, instances per handle)
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
    std::vector<uint32_t> ready; // sums whose chain of additions ends at this level
    { // expressions that this level's key switches, rescales, rotations and outputs read: evaluated together, first
      std::vector<uint32_t> need;
      for (uint32_t i : lvl) {
        const evah_op &o = ops[i];
        if (o.op == 10 || o.op == 11 || o.op == 12 || o.op == 13) continue;
        if (st.exprs.count(o.src0) && std::find(need.begin(), need.end(), o.src0) == need.end()) need.push_back(o.src0);
      }
      if (!need.empty()) force_exprs(need);
    }
    // ---- classify
    for (uint32_t i : lvl) {
      const evah_op &o = ops[i];
      uint32_t size = 0, limbs = 0;
      double scale = 0;
      // a batched handle already covers its instances in one launch set: the *_many forms take
      // single ciphertexts, so on batched operands only sibling rotations are grouped (rotate_many
      // accepts them) and the deferred forms below still apply
      auto batched_val = [&](uint32_t v) { // (r5 advisor: an operand that is still an unevaluated expression has a batch too)
        auto ex = st.exprs.find(v);
        if (ex != st.exprs.end()) return ex->second->batch > 1;
        return tab[v].kind == EVAH_VAL_CT && static_cast<evah_ct *>(tab[v].h)->batch > 1;
      };
      const bool batched = batched_val(o.src0) || ((o.op == 11 || o.op == 12 || o.op == 13) && batched_val(o.src1)) ||
                           (st.relins.count(o.src0) && st.relins[o.src0]->batch > 1);
      if ((o.op == 14 || o.op == 15) && o.imm != 0 && is_ct(o.src0) && window_rotation(o)) {
        evah_ct *src = ct_of(o.src0);
        if (src->size == 2) { // nothing is computed here: the sums that consume the products evaluate the rotation
          st.drots[o.dst] = DRot{alias_ct(src), o.op == 14 ? o.imm : -o.imm};
          continue;
        }
      }
      if (batched && (o.op == 14 || o.op == 15) && o.imm != 0) {
        batched_rots[o.src0].push_back(i);
        continue;
      }
      // Mul (or square) read only by a Rescale that is read only by a Relinearize — lazy relinearization's order: nothing is
      // computed here, the three run as one fused call at the Relinearize (r6; batched handles too: the instances of a handle
      // are entries of the same launch set)
      auto defer_chain2 = [&]() {
        if (!(o.op == 13 && is_ct(o.src0) && is_ct(o.src1) && c->tun.fuse_mac && c->tun.fuse_mul2 && c->sh->relin.d &&
              feeds_only(o.dst, 22) && feeds_only(ops[only_reader[o.dst]].dst, 20)))
          return false;
        evah_ct *x = ct_of(o.src0), *y = ct_of(o.src1);
        if (!(x->size == 2 && y->size == 2 && x->limbs == y->limbs && x->limbs >= 2 && x->batch == y->batch && x->batch <= (uint32_t)KS_BATCH_MAX)) return false;
        if (x->batch > 1 && !c->tun.chain_batched) return false;
        check_scale(c, x->scale * y->scale, x->limbs);
        st.prods2[o.dst] = {alias_ct(x), alias_ct(y)};
        return true;
      };
      if (batched && defer_chain2()) continue;
      // Rescale of a size-3 ciphertext read only by a Relinearize (r6): nothing is computed here either
      auto defer_resrel = [&]() {
        if (!(o.op == 22 && is_ct(o.src0) && !st.relins.count(o.src0) && c->tun.fuse_mac && c->tun.fuse_mul2 && c->tun.chain_step &&
              c->sh->relin.d && feeds_only(o.dst, 20)))
          return false;
        evah_ct *x = ct_of(o.src0);
        if (!(x->size == 3 && x->limbs >= 2 && x->batch <= (uint32_t)KS_BATCH_MAX)) return false;
        if (x->batch > 1 && !c->tun.chain_batched) return false;
        st.resrel[o.dst] = alias_ct(x);
        st.resdiv[o.dst] = (uint32_t)o.imm;
        return true;
      };
      if (batched && defer_resrel()) continue;
      if (batched && (o.op == 22 || (o.op == 20 && !feeds_only(o.dst, 22)) || (o.op == 13 && o.src0 != o.src1 && is_ct(o.src0) && is_ct(o.src1)))) {
        if (o.op == 13 && try_defer_ew(o)) continue; // a product of batched handles inside an elementwise expression
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
      if (o.op == 22 && st.prods2.count(o.src0)) { // Rescale of a deferred product that a Relinearize follows: still deferred
        st.prodres[o.dst] = st.prods2[o.src0];
        st.resdiv[o.dst] = (uint32_t)o.imm;
        st.prods2.erase(o.src0);
      } else if (o.op == 20 && st.resrel.count(o.src0)) {
        fused_rr[{st.resrel[o.src0]->limbs, st.resdiv[o.src0], st.resrel[o.src0]->batch}].push_back(i);
      } else if (o.op == 20 && st.prodres.count(o.src0)) {
        fused3b[{st.prodres[o.src0].first->limbs, st.resdiv[o.src0], st.prodres[o.src0].first->batch}].push_back(i);
      } else if (o.op == 20 && st.prods.count(o.src0)) { // Relinearize of a deferred product: still deferred
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
      } else if (!batched && defer_resrel()) {
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
      } else if (!batched && defer_chain2()) {
      } else if (o.op == 13 && is_ct(o.src0) && is_ct(o.src1) && try_defer_ew(o)) {
        // ciphertext x ciphertext inside an elementwise expression (not a Mul -> Relinearize -> Rescale chain): nothing runs here
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
                 ((st.drots.count(o.src0) && slot(o.src1).kind == EVAH_VAL_PT) || (st.drots.count(o.src1) && slot(o.src0).kind == EVAH_VAL_PT))) {
        const uint32_t a = st.drots.count(o.src0) ? o.src0 : o.src1, b = a == o.src0 ? o.src1 : o.src0;
        const DRot &d = st.drots[a];
        evah_pt *w = static_cast<evah_pt *>(tab[b].h);
        if (w->limbs != d.src->limbs) { single(o); continue; } // multiply_plain reports the mismatch
        LazySum ls;
        ls.size = d.src->size; ls.limbs = d.src->limbs; ls.scale = d.src->scale * w->scale;
        ls.cts.push_back(alias_ct(d.src));
        ls.pts.push_back(alias_pt(w));
        ls.steps.push_back(d.step);
        ls.rv.push_back(a);
        st.sums[o.dst] = std::move(ls);
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
        ls.steps.push_back(0);
        ls.rv.push_back(NONE_V);
        st.sums[o.dst] = std::move(ls);
      } else if (o.op == 13 && try_defer_ew(o)) {
        // ciphertext x plaintext that is not a term of a sum: part of an elementwise expression
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
            ls.steps.insert(ls.steps.end(), it->second.steps.begin(), it->second.steps.end());
            ls.rv.insert(ls.rv.end(), it->second.rv.begin(), it->second.rv.end());
          } else {
            ls.cts.push_back(alias_ct(ct_of(v)));
            ls.pts.push_back(nullptr);
            ls.steps.push_back(0);
            ls.rv.push_back(NONE_V);
          }
        }
        st.sums[o.dst] = std::move(ls);
        // the chain ends here: evaluated with the other sums that end at this level (windows share their rotations)
        if (!(feeds_only(o.dst, 11) && nterms < (size_t)KS_BATCH_MAX)) ready.push_back(o.dst);
      } else if (try_defer_ew(o)) {
        // add / sub / negate (and what is left of the products) on values nobody needs stored yet
      } else {
        single(o);
      }
    }
    {
      std::vector<uint32_t> pending;
      for (uint32_t v : ready) if (st.sums.count(v)) pending.push_back(v); // (not forced by another op of the level meanwhile)
      if (!pending.empty()) eval_sums(pending);
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
    for (auto &kv : fused3b) {
      const size_t per_call = std::max<size_t>(1, (size_t)KS_BATCH_MAX / std::max<uint32_t>(1, std::get<2>(kv.first)));
      for (size_t i0 = 0; i0 < kv.second.size(); i0 += per_call) {
        const uint32_t n = (uint32_t)std::min<size_t>(per_call, kv.second.size() - i0);
        const uint32_t *is = kv.second.data() + i0;
        std::vector<const evah_ct *> ia(n), ib(n);
        std::vector<evah_ct *> outs(n, nullptr);
        for (uint32_t j = 0; j < n; j++) {
          ia[j] = st.prodres[ops[is[j]].src0].first;
          ib[j] = st.prodres[ops[is[j]].src0].second;
        }
        chk(evah_multiply_rescale_relinearize_many(c, ia.data(), ib.data(), n, std::get<1>(kv.first), outs.data()));
        for (uint32_t j = 0; j < n; j++) {
          evah_ct_free(c, const_cast<evah_ct *>(ia[j]));
          evah_ct_free(c, const_cast<evah_ct *>(ib[j]));
          st.prodres.erase(ops[is[j]].src0);
          st.resdiv.erase(ops[is[j]].src0);
        }
        store(is, n, outs);
      }
    }
    for (auto &kv : fused_rr) {
      const size_t per_call = std::max<size_t>(1, (size_t)KS_BATCH_MAX / std::max<uint32_t>(1, std::get<2>(kv.first)));
      for (size_t i0 = 0; i0 < kv.second.size(); i0 += per_call) {
        const uint32_t n = (uint32_t)std::min<size_t>(per_call, kv.second.size() - i0);
        const uint32_t *is = kv.second.data() + i0;
        std::vector<const evah_ct *> in(n);
        std::vector<evah_ct *> outs(n, nullptr);
        for (uint32_t j = 0; j < n; j++) in[j] = st.resrel[ops[is[j]].src0];
        chk(evah_rescale_relinearize_many(c, in.data(), n, std::get<1>(kv.first), outs.data()));
        for (uint32_t j = 0; j < n; j++) {
          evah_ct_free(c, const_cast<evah_ct *>(in[j]));
          st.resrel.erase(ops[is[j]].src0);
          st.resdiv.erase(ops[is[j]].src0);
        }
        store(is, n, outs);
      }
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
          auto dr = st.drots.find(v);
          if (dr != st.drots.end()) { evah_ct_free(c, dr->second.src); st.drots.erase(dr); } // its terms live in the sums now
          auto lr = st.relins.find(v);
          if (lr != st.relins.end()) { evah_ct_free(c, lr->second); st.relins.erase(lr); }
          st.exprs.erase(v); // its readers hold the node (their operand) until they are evaluated
          if (tab[v].kind != EVAH_VAL_NONE) release(v);
        }
      }
    }
  }
  API_END
}

} // extern "C"
