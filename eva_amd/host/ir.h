// ir.h — the DAG container the MI355X executor consumes and the compiler rewrites.
//
// Same vocabulary as EVA's IR so compiled programs mean the same thing — Op and Type codes of
// /root/reference/eva/ir/ops.h:11-25 and types.h:11-15, the run-time attributes of
// attributes.h:12-19 — but a different shape: terms live in one index-addressed arena
// (TermId = position), side tables are plain vectors keyed by TermId, and liveness is
// reachability from the outputs instead of shared_ptr ownership.  That is what the device
// scheduler wants: a flat, topologically sortable op list with dense ids.
#pragma once
#include <algorithm>
#include <cstdint>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

namespace evahost {

enum class Op : int32_t {
  Undef = 0, Input = 1, Output = 2, Constant = 3,
  Negate = 10, Add = 11, Sub = 12, Mul = 13, RotateLeftConst = 14, RotateRightConst = 15,
  Relinearize = 20, ModSwitch = 21, Rescale = 22, Encode = 23
};
enum class Type : int32_t { Undef = 0, Cipher = 1, Raw = 2, Plain = 3 };

inline const char *op_name(Op op) {
  switch (op) {
  case Op::Undef: return "Undef";
  case Op::Input: return "Input";
  case Op::Output: return "Output";
  case Op::Constant: return "Constant";
  case Op::Negate: return "Negate";
  case Op::Add: return "Add";
  case Op::Sub: return "Sub";
  case Op::Mul: return "Mul";
  case Op::RotateLeftConst: return "RotateLeftConst";
  case Op::RotateRightConst: return "RotateRightConst";
  case Op::Relinearize: return "Relinearize";
  case Op::ModSwitch: return "ModSwitch";
  case Op::Rescale: return "Rescale";
  case Op::Encode: return "Encode";
  }
  throw std::runtime_error("Invalid op");
}
inline const char *type_name(Type t) {
  switch (t) {
  case Type::Undef: return "Undef";
  case Type::Cipher: return "Cipher";
  case Type::Raw: return "Raw";
  case Type::Plain: return "Plain";
  }
  throw std::runtime_error("Invalid type");
}

using TermId = uint32_t;
constexpr TermId NO_TERM = 0xFFFFFFFFu;

// Constant vector with broadcast semantics (constant_value.h:16-139): `values.size()` divides
// the program vector size and is replicated to fill it.
struct ConstantValue {
  std::vector<double> values;
  void expand_to(std::vector<double> &out, size_t slots) const {
    if (values.empty() || slots < values.size() || slots % values.size())
      throw std::runtime_error("Size must exactly divide slots");
    out.clear();
    out.reserve(slots);
    for (size_t r = slots / values.size(); r > 0; --r) out.insert(out.end(), values.begin(), values.end());
  }
  bool uniform() const {
    for (double v : values)
      if (v != values[0]) return false;
    return true;
  }
};

struct Term {
  Op op = Op::Undef;
  std::vector<TermId> operands;
  std::vector<TermId> uses; // one entry per operand slot that references this term
  // attributes (attributes.h:12-19); has_* mirror AttributeList::has<>
  bool has_rescale_divisor = false, has_rotation = false, has_type = false, has_range = false,
       has_encode_scale = false, has_encode_level = false;
  uint32_t rescale_divisor = 0;
  int32_t rotation = 0;
  Type type_attr = Type::Undef;
  uint32_t range = 0, encode_scale = 0, encode_level = 0;
  std::shared_ptr<ConstantValue> constant;
};

class Program {
public:
  Program(std::string name, uint64_t vec_size) : name_(std::move(name)), vec_size_((uint32_t)vec_size) {
    if (vec_size == 0) throw std::runtime_error("Vector size must be non-zero");
    if (vec_size & (vec_size - 1)) throw std::runtime_error("Vector size must be a power-of-two");
  }
  Program(const Program &) = delete;
  Program &operator=(const Program &) = delete;

  // ---- construction (program.h:37-110)
  TermId make_term(Op op, const std::vector<TermId> &operands = {}) {
    TermId id = (TermId)terms_.size();
    terms_.emplace_back();
    terms_[id].op = op;
    for (TermId o : operands) add_operand(id, o);
    return id;
  }
  TermId make_dense_constant(std::vector<double> values) {
    if (values.empty() || vec_size_ % values.size()) throw std::runtime_error("DenseConstantValue size must exactly divide size");
    TermId id = make_term(Op::Constant);
    terms_[id].constant = std::make_shared<ConstantValue>(ConstantValue{std::move(values)});
    return id;
  }
  TermId make_uniform_constant(double v) { return make_dense_constant({v}); }
  TermId make_input(const std::string &name, Type type = Type::Cipher) {
    TermId id = make_term(Op::Input);
    terms_[id].has_type = true;
    terms_[id].type_attr = type;
    inputs_.emplace(name, id);
    return id;
  }
  TermId make_output(const std::string &name, TermId t) {
    TermId id = make_term(Op::Output, {t});
    outputs_.emplace(name, id);
    return id;
  }
  TermId make_left_rotation(TermId t, int32_t slots) {
    TermId id = make_term(Op::RotateLeftConst, {t});
    terms_[id].has_rotation = true;
    terms_[id].rotation = slots;
    return id;
  }
  TermId make_right_rotation(TermId t, int32_t slots) {
    TermId id = make_term(Op::RotateRightConst, {t});
    terms_[id].has_rotation = true;
    terms_[id].rotation = slots;
    return id;
  }
  TermId make_rescale(TermId t, uint32_t by) {
    TermId id = make_term(Op::Rescale, {t});
    terms_[id].has_rescale_divisor = true;
    terms_[id].rescale_divisor = by;
    return id;
  }

  // used when rebuilding a program from a file
  void bind_input(const std::string &name, TermId t) { inputs_[name] = t; }
  void bind_output(const std::string &name, TermId t) { outputs_[name] = t; }

  // ---- access
  Term &at(TermId t) { return terms_.at(t); }
  const Term &at(TermId t) const { return terms_.at(t); }
  size_t size() const { return terms_.size(); }
  const std::string &name() const { return name_; }
  void set_name(std::string n) { name_ = std::move(n); }
  uint32_t vec_size() const { return vec_size_; }
  const std::unordered_map<std::string, TermId> &inputs() const { return inputs_; }
  const std::unordered_map<std::string, TermId> &outputs() const { return outputs_; }
  TermId input(const std::string &name) const {
    auto it = inputs_.find(name);
    if (it == inputs_.end()) throw std::out_of_range("No input named " + name);
    return it->second;
  }

  // ---- rewriting (term.h:30-46); uses are kept exact
  void add_operand(TermId t, TermId o) {
    terms_[t].operands.push_back(o);
    terms_[o].uses.push_back(t);
  }
  bool erase_operand(TermId t, TermId o) {
    auto &ops = terms_[t].operands;
    auto it = std::find(ops.begin(), ops.end(), o);
    if (it == ops.end()) return false;
    ops.erase(it);
    erase_use(o, t);
    return true;
  }
  bool replace_operand(TermId t, TermId old_t, TermId new_t) {
    bool replaced = false;
    for (TermId &o : terms_[t].operands)
      if (o == old_t) {
        o = new_t;
        erase_use(old_t, t);
        terms_[new_t].uses.push_back(t);
        replaced = true;
      }
    return replaced;
  }
  void set_operands(TermId t, const std::vector<TermId> &ops) {
    for (TermId o : terms_[t].operands) erase_use(o, t);
    terms_[t].operands.clear();
    for (TermId o : ops) add_operand(t, o);
  }
  // distinct users, in first-use order
  std::vector<TermId> uses_of(TermId t) const {
    std::vector<TermId> u;
    for (TermId x : terms_[t].uses)
      if (std::find(u.begin(), u.end(), x) == u.end()) u.push_back(x);
    return u;
  }
  size_t num_uses(TermId t) const { return terms_[t].uses.size(); }
  void replace_all_uses_with(TermId t, TermId with) {
    for (TermId u : uses_of(t)) replace_operand(u, t, with);
  }
  void replace_other_uses_with(TermId t, TermId with) {
    for (TermId u : uses_of(t))
      if (u != with) replace_operand(u, t, with);
  }

  // Detach terms that lost their last user (the reference frees them through shared_ptr
  // ownership): afterwards `uses` lists name live users only.  Called between visits.
  void gc() {
    std::vector<TermId> work;
    for (TermId t = 0; t < terms_.size(); t++)
      if (terms_[t].uses.empty() && !terms_[t].operands.empty() && terms_[t].op != Op::Output) work.push_back(t);
    while (!work.empty()) {
      TermId t = work.back();
      work.pop_back();
      std::vector<TermId> ops = terms_[t].operands;
      set_operands(t, {});
      for (TermId o : ops)
        if (terms_[o].uses.empty() && !terms_[o].operands.empty() && terms_[o].op != Op::Output) work.push_back(o);
    }
  }

  // ---- liveness: a term is live if an Output reaches it (the reference drops unreferenced
  // terms through shared_ptr ownership; here they simply stay unreachable in the arena)
  std::vector<char> live_mask() const {
    std::vector<char> live(terms_.size(), 0);
    std::vector<TermId> stack;
    for (auto &kv : outputs_) stack.push_back(kv.second);
    for (auto &kv : inputs_) stack.push_back(kv.second); // inputs stay part of the signature
    while (!stack.empty()) {
      TermId t = stack.back();
      stack.pop_back();
      if (live[t]) continue;
      live[t] = 1;
      for (TermId o : terms_[t].operands) stack.push_back(o);
    }
    return live;
  }
  std::vector<TermId> sources() const {
    auto live = live_mask();
    std::vector<TermId> out;
    for (TermId t = 0; t < terms_.size(); t++)
      if (live[t] && terms_[t].operands.empty()) out.push_back(t);
    return out;
  }
  // topological order of the live terms (operands first), deterministic
  std::vector<TermId> topo_order() const {
    auto live = live_mask();
    std::vector<uint32_t> pending(terms_.size(), 0);
    std::vector<TermId> ready, order;
    for (TermId t = 0; t < terms_.size(); t++) {
      if (!live[t]) continue;
      pending[t] = (uint32_t)terms_[t].operands.size();
      if (!pending[t]) ready.push_back(t);
    }
    std::reverse(ready.begin(), ready.end());
    while (!ready.empty()) {
      TermId t = ready.back();
      ready.pop_back();
      order.push_back(t);
      for (TermId u : terms_[t].uses)
        if (live[u] && --pending[u] == 0) ready.push_back(u);
    }
    return order;
  }

  std::unique_ptr<Program> deep_copy() const {
    auto p = std::make_unique<Program>(name_, vec_size_);
    std::vector<TermId> map(terms_.size(), NO_TERM);
    for (TermId t : topo_order()) {
      TermId n = p->make_term(terms_[t].op);
      Term &nt = p->at(n);
      const Term &ot = terms_[t];
      nt.has_rescale_divisor = ot.has_rescale_divisor; nt.rescale_divisor = ot.rescale_divisor;
      nt.has_rotation = ot.has_rotation; nt.rotation = ot.rotation;
      nt.has_type = ot.has_type; nt.type_attr = ot.type_attr;
      nt.has_range = ot.has_range; nt.range = ot.range;
      nt.has_encode_scale = ot.has_encode_scale; nt.encode_scale = ot.encode_scale;
      nt.has_encode_level = ot.has_encode_level; nt.encode_level = ot.encode_level;
      nt.constant = ot.constant;
      for (TermId o : ot.operands) p->add_operand(n, map[o]);
      map[t] = n;
    }
    for (auto &kv : inputs_) p->inputs_[kv.first] = map[kv.second];
    for (auto &kv : outputs_) p->outputs_[kv.first] = map[kv.second];
    return p;
  }

  std::string to_dot() const {
    std::string s = "digraph \"" + name_ + "\" {\n";
    for (TermId t : topo_order()) {
      const Term &x = terms_[t];
      s += "t" + std::to_string(t) + " [label=\"" + op_name(x.op);
      if (x.has_rescale_divisor) s += "(" + std::to_string(x.rescale_divisor) + ")";
      if (x.has_rotation) s += "(" + std::to_string(x.rotation) + ")";
      if (x.has_type) s += std::string(" : ") + type_name(x.type_attr);
      s += "\"];\n";
      for (size_t i = 0; i < x.operands.size(); i++)
        s += "t" + std::to_string(x.operands[i]) + " -> t" + std::to_string(t) + " [label=\"" + std::to_string(i) + "\"];\n";
    }
    return s + "}\n";
  }

private:
  void erase_use(TermId o, TermId user) {
    auto &u = terms_[o].uses;
    auto it = std::find(u.begin(), u.end(), user);
    if (it != u.end()) u.erase(it);
  }
  std::string name_;
  uint32_t vec_size_;
  std::vector<Term> terms_;
  std::unordered_map<std::string, TermId> inputs_, outputs_;
};

// Side table keyed by TermId that grows with the arena.
template <class T> class TermTable {
public:
  explicit TermTable(T init = T()) : init_(init) {}
  T &operator[](TermId t) {
    if (t >= v_.size()) v_.resize((size_t)t + 1, init_);
    return v_[t];
  }
  T get(TermId t) const { return t < v_.size() ? v_[t] : init_; }
  void clear() { v_.clear(); }

private:
  T init_;
  std::vector<T> v_;
};

// Forward / backward work-list traversal that tolerates rewrites made by the visitor
// (same contract as ProgramTraversal, /root/reference/eva/common/program_traversal.h:12-20:
// each term visited exactly once; the visitor may only add or rewire terms around the
// current one).
template <bool FORWARD, class Visitor> void traverse(Program &p, Visitor &&visit) {
  TermTable<char> ready(0), processed(0);
  std::vector<TermId> work;
  auto leaves = [&]() {
    std::vector<TermId> out;
    auto live = p.live_mask();
    for (TermId t = 0; t < p.size(); t++) {
      if (!live[t]) continue;
      if (FORWARD ? p.at(t).operands.empty() : p.at(t).uses.empty()) out.push_back(t);
    }
    return out;
  };
  auto preds_done = [&](TermId t) {
    const auto &pre = FORWARD ? p.at(t).operands : p.at(t).uses;
    for (TermId x : pre)
      if (!processed.get(x)) return false;
    return true;
  };
  for (TermId t : leaves()) {
    work.push_back(t);
    ready[t] = 1;
  }
  std::vector<TermId> check;
  while (!work.empty()) {
    TermId t = work.back();
    work.pop_back();
    check = FORWARD ? p.at(t).uses : p.at(t).operands;
    visit(t);
    processed[t] = 1;
    p.gc();
    for (TermId l : leaves()) // rewrites may have created new leaves
      if (!ready.get(l)) {
        work.push_back(l);
        ready[l] = 1;
      }
    const auto &succ = FORWARD ? p.at(t).uses : p.at(t).operands;
    check.insert(check.end(), succ.begin(), succ.end());
    for (TermId s : check)
      if (!ready.get(s) && preds_done(s)) {
        work.push_back(s);
        ready[s] = 1;
      }
  }
}
template <class V> void forward_pass(Program &p, V &&v) { traverse<true>(p, std::forward<V>(v)); }
template <class V> void backward_pass(Program &p, V &&v) { traverse<false>(p, std::forward<V>(v)); }

} // namespace evahost
