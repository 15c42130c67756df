// passes.h — CKKS compiler for the arena IR: produces the compiled DAG (Rescale / Relinearize /
// ModSwitch / Encode inserted, levels and scales fixed) that the MI355X executor runs, plus the
// encryption parameters and the input signature.
//
// Behavioural restatement of /root/reference/eva/ckks/ckks_compiler.h:36-306 and the passes it
// drives; each pass cites the file it follows.  Host-side, runs once per program.
#pragma once
#include <cstdio>
#include <cstdlib>
#include <map>
#include <numeric>
#include <set>
#include <sstream>
#include <tuple>

#include "ir.h"

namespace evahost {

// ---- logging: EVA_VERBOSITY convention of /root/reference/eva/util/logging.cpp:12-68
inline int verbosity() {
  static int v = [] {
    const char *e = std::getenv("EVA_VERBOSITY");
    if (!e) return 0;
    std::string s(e);
    for (auto &c : s) c = (char)std::tolower(c);
    if (s == "silent") return 0;
    if (s == "info") return 1;
    if (s == "debug") return 2;
    if (s == "trace") return 3;
    return std::atoi(e);
  }();
  return v;
}
inline void warn(const std::string &msg) { std::fprintf(stderr, "WARNING: %s\n", msg.c_str()); }

using Types = TermTable<Type>;
using Scales = TermTable<uint32_t>;

// common/type_deducer.h:11-38
struct TypeDeducer {
  Program &p;
  Types &types;
  void operator()(TermId t) {
    const Term &x = p.at(t);
    if (!x.operands.empty()) {
      Type inferred = Type::Raw;
      for (TermId o : x.operands)
        if (types.get(o) == Type::Cipher) inferred = Type::Cipher;
      types[t] = x.op == Op::Encode ? Type::Plain : inferred;
    } else if (x.op == Op::Constant) {
      types[t] = Type::Raw;
    } else {
      types[t] = x.type_attr;
    }
  }
};

// common/constant_folder.h:11-190 — fold ops whose operands are all constants
struct ConstantFolder {
  Program &p;
  Scales &scale;
  std::vector<double> a, b;
  void replace(TermId t, std::vector<double> out, uint32_t s) {
    TermId c = p.make_dense_constant(std::move(out));
    scale[c] = s;
    p.at(c).has_encode_scale = true;
    p.at(c).encode_scale = s;
    p.replace_all_uses_with(t, c);
  }
  static void norm_shift(int32_t &shift, size_t n) {
    while (shift > 0 && (size_t)shift >= n) shift -= (int32_t)n;
    while (shift < 0) shift += (int32_t)n;
  }
  void operator()(TermId t) {
    const Term x = p.at(t);
    if (x.operands.empty()) return;
    for (TermId o : x.operands)
      if (p.at(o).op != Op::Constant) return;
    size_t n = p.vec_size();
    p.at(x.operands[0]).constant->expand_to(a, n);
    if (x.operands.size() > 1) p.at(x.operands[1]).constant->expand_to(b, n);
    std::vector<double> out(n);
    uint32_t s0 = scale.get(x.operands[0]);
    uint32_t s01 = x.operands.size() > 1 ? std::max(s0, scale.get(x.operands[1])) : s0;
    switch (x.op) {
    case Op::Add: for (size_t i = 0; i < n; i++) out[i] = a[i] + b[i]; replace(t, out, s01); break;
    case Op::Sub: for (size_t i = 0; i < n; i++) out[i] = a[i] - b[i]; replace(t, out, s01); break;
    case Op::Mul: for (size_t i = 0; i < n; i++) out[i] = a[i] * b[i]; replace(t, out, s01); break;
    case Op::RotateLeftConst: {
      int32_t sh = x.rotation;
      norm_shift(sh, n);
      for (size_t i = 0; i < n; i++) out[i] = a[(i + sh) % n];
      replace(t, out, s0);
    } break;
    case Op::RotateRightConst: {
      int32_t sh = x.rotation;
      norm_shift(sh, n);
      for (size_t i = 0; i < n; i++) out[(i + sh) % n] = a[i];
      replace(t, out, s0);
    } break;
    case Op::Negate: for (size_t i = 0; i < n; i++) out[i] = -a[i]; replace(t, out, s0); break;
    case Op::Output:
    case Op::Encode: break;
    case Op::Relinearize:
    case Op::ModSwitch:
    case Op::Rescale:
      throw std::logic_error(std::string("Encountered HE specific operation ") + op_name(x.op) + " in unencrypted computation");
    default: throw std::logic_error(std::string("Unhandled op ") + op_name(x.op));
    }
  }
};

// common/reduction_balancer.h:30-66 — merge single-use chains of the same Add/Mul into one
// n-ary node
struct ReductionCombiner {
  Program &p;
  void operator()(TermId t) {
    const Term &x = p.at(t);
    if (x.operands.empty() || x.uses.empty()) return;
    if (x.op != Op::Add && x.op != Op::Mul) return;
    // one use EDGE (reduction_balancer.h:44-45: Term::getUses lists a user once per operand slot,
    // so t*t with t = a*a is left alone instead of being flattened into a 4-ary product)
    if (p.num_uses(t) != 1 || p.at(x.uses[0]).op != x.op) return;
    TermId use = x.uses[0];
    while (p.erase_operand(use, t))
      for (TermId o : std::vector<TermId>(p.at(t).operands)) p.add_operand(use, o);
  }
};

// common/reduction_balancer.h:68-146 — expand n-ary Add/Mul into a balanced tree, cheap
// (plain, low-scale) operands first
struct ReductionLogExpander {
  Program &p;
  Types &type;
  TermTable<int> scale{0};
  void operator()(TermId t) {
    const Term &x = p.at(t);
    if (x.op == Op::Rescale || x.op == Op::ModSwitch)
      throw std::logic_error("Rescale or ModSwitch encountered, but ReductionLogExpander uses scale as a proxy for level and assumes rescaling has not been performed yet.");
    if (x.operands.empty()) {
      scale[t] = (int)x.encode_scale;
    } else if (x.op == Op::Mul) {
      int s = 0;
      for (TermId o : x.operands) s += scale.get(o);
      scale[t] = s;
    } else {
      int s = 0;
      for (TermId o : x.operands) s = std::max(s, scale.get(o));
      scale[t] = s;
    }
    if ((x.op == Op::Add || x.op == Op::Mul) && x.operands.size() > 2) {
      std::map<uint32_t, std::vector<TermId>> sorted;
      for (TermId o : x.operands) {
        uint32_t order = 0;
        if (type.get(o) == Type::Plain || type.get(o) == Type::Raw) order = 1;
        else if (type.get(o) == Type::Cipher) order = 2 + (uint32_t)scale.get(o);
        sorted[order].push_back(o);
      }
      std::vector<TermId> ops, next;
      for (auto &kv : sorted) ops.insert(ops.end(), kv.second.begin(), kv.second.end());
      Op op = x.op;
      while (ops.size() > 2) {
        size_t i = 0;
        for (; i + 1 < ops.size(); i += 2) next.push_back(p.make_term(op, {ops[i], ops[i + 1]}));
        if (i < ops.size()) next.push_back(ops[i]);
        ops.swap(next);
        next.clear();
      }
      p.set_operands(t, ops);
    }
  }
};

// ---- rescalers (ckks/rescaler.h:11-58 shared helpers)
struct RescalerBase {
  Program &p;
  Types &type;
  Scales &scale;
  uint32_t min_scale = 0;
  RescalerBase(Program &p_, Types &t_, Scales &s_) : p(p_), type(t_), scale(s_) {
    for (TermId s : p.sources()) min_scale = std::max(min_scale, scale.get(s));
  }
  TermId insert_rescale(TermId t, uint32_t by) {
    TermId r = p.make_rescale(t, by);
    type[r] = type.get(t);
    scale[r] = scale.get(t) - by;
    p.replace_other_uses_with(t, r);
    return r;
  }
  void insert_rescale_between(TermId t1, TermId t2, uint32_t by) {
    TermId r = p.make_rescale(t1, by);
    type[r] = type.get(t1);
    scale[r] = scale.get(t1) - by;
    p.replace_operand(t2, t1, r);
  }
  void handle_raw_scale(TermId t) {
    uint32_t m = 0;
    for (TermId o : p.at(t).operands) m = std::max(m, scale.get(o));
    if (!p.at(t).operands.empty()) scale[t] = m;
  }
  // bring the lower-scale operands of an Add/Sub up with a multiply by an encoded 1
  // (lazy_waterline_rescaler.h:96-118); returns the common scale
  uint32_t match_addition_scales(TermId t) {
    uint32_t mx = scale.get(p.at(t).operands[0]);
    for (TermId o : p.at(t).operands) mx = std::max(mx, scale.get(o));
    for (TermId o : std::vector<TermId>(p.at(t).operands)) {
      if (scale.get(o) < mx && type.get(o) != Type::Raw) {
        TermId c = p.make_uniform_constant(1);
        scale[c] = mx - scale.get(o);
        p.at(c).has_encode_scale = true;
        p.at(c).encode_scale = scale.get(c);
        TermId m = p.make_term(Op::Mul, {o, c});
        scale[m] = mx;
        p.replace_operand(t, o, m);
      }
    }
    return mx;
  }
  static bool is_add(Op op) { return op == Op::Add || op == Op::Sub; }
};

// ckks/lazy_waterline_rescaler.h:11-153
struct LazyWaterlineRescaler : RescalerBase {
  static constexpr uint32_t fixed = 60;
  TermTable<char> pending{0};
  TermTable<uint32_t> level{0};
  using RescalerBase::RescalerBase;
  void insert_recursive(TermId t) {
    TermId cur = t;
    uint32_t s = scale.get(cur), num = 0;
    while (s >= fixed + min_scale) {
      cur = insert_rescale(cur, fixed);
      ++num;
      s -= fixed;
    }
    level[cur] = level.get(t) + num;
  }
  void operator()(TermId t) {
    if (p.at(t).operands.empty()) return;
    if (type.get(t) == Type::Raw) { handle_raw_scale(t); return; }
    Op op = p.at(t).op;
    if (op == Op::Rescale) return;
    if (op == Op::Mul) {
      uint32_t ms = 0, ml = 0;
      for (TermId o : p.at(t).operands) { ms += scale.get(o); ml = std::max(ml, level.get(o)); }
      scale[t] = ms;
      level[t] = ml;
      if (ms >= fixed + min_scale) pending[t] = 1;
      else return;
    } else {
      scale[t] = scale.get(p.at(t).operands[0]);
      level[t] = level.get(p.at(t).operands[0]);
      if (is_add(op)) {
        uint32_t ml = 0;
        for (TermId o : p.at(t).operands) ml = std::max(ml, level.get(o));
        level[t] = ml;
        scale[t] = match_addition_scales(t);
      }
      if (!pending.get(t)) return;
    }
    bool must = false;
    auto uses = p.at(t).uses;
    if (uses.empty()) throw std::logic_error("rescaler: term without uses");
    for (TermId u : uses)
      if (p.at(u).op == Op::Mul || p.at(u).op == Op::Output || u != uses[0]) { must = true; break; }
    if (must) {
      pending[t] = 0;
      insert_recursive(t);
    } else {
      for (TermId u : uses) pending[u] = 1;
    }
  }
};

// ckks/eager_waterline_rescaler.h:11-93
struct EagerWaterlineRescaler : RescalerBase {
  static constexpr uint32_t fixed = 60;
  using RescalerBase::RescalerBase;
  void operator()(TermId t) {
    if (p.at(t).operands.empty()) return;
    if (type.get(t) == Type::Raw) { handle_raw_scale(t); return; }
    Op op = p.at(t).op;
    if (op == Op::Rescale) return;
    if (op != Op::Mul) {
      scale[t] = scale.get(p.at(t).operands[0]);
      if (is_add(op)) scale[t] = match_addition_scales(t);
      return;
    }
    uint32_t ms = 0;
    for (TermId o : p.at(t).operands) ms += scale.get(o);
    scale[t] = ms;
    TermId cur = t;
    while (ms >= fixed + min_scale) {
      cur = insert_rescale(cur, fixed);
      ms -= fixed;
    }
  }
};

// ckks/always_rescaler.h:10-63
struct AlwaysRescaler : RescalerBase {
  using RescalerBase::RescalerBase;
  void operator()(TermId t) {
    if (p.at(t).operands.empty()) return;
    if (type.get(t) == Type::Raw) { handle_raw_scale(t); return; }
    Op op = p.at(t).op;
    if (op == Op::Rescale) return;
    if (op != Op::Mul) { scale[t] = scale.get(p.at(t).operands[0]); return; }
    uint32_t ms = 0;
    for (TermId o : p.at(t).operands) ms += scale.get(o);
    scale[t] = ms;
    insert_rescale(t, ms - min_scale);
  }
};

// ckks/minimum_rescaler.h:11-122
struct MinimumRescaler : RescalerBase {
  static constexpr uint32_t max_rescale = 60;
  using RescalerBase::RescalerBase;
  void operator()(TermId t) {
    if (p.at(t).operands.empty()) return;
    if (type.get(t) == Type::Raw) { handle_raw_scale(t); return; }
    Op op = p.at(t).op;
    if (op == Op::Rescale) return;
    if (op != Op::Mul) {
      scale[t] = scale.get(p.at(t).operands[0]);
      if (is_add(op)) scale[t] = match_addition_scales(t);
      return;
    }
    std::vector<TermId> ops = p.at(t).operands;
    uint32_t ms = scale.get(ops[0]) + scale.get(ops[1]);
    scale[t] = ms;
    uint32_t mn = std::min(scale.get(ops[0]), scale.get(ops[1]));
    uint32_t by = std::min(mn - min_scale, max_rescale);
    if (2 * by >= max_rescale) {
      insert_rescale_between(ops[0], t, by);
      if (ops[0] != ops[1]) insert_rescale_between(ops[1], t, by);
      scale[t] = ms - 2 * by;
    } else {
      TermId cur = t;
      while (ms >= max_rescale + min_scale) {
        cur = insert_rescale(cur, max_rescale);
        ms -= max_rescale;
      }
    }
  }
};

// ckks/encode_inserter.h:11-60 — Raw operand meeting a Cipher operand gets an Encode node
struct EncodeInserter {
  Program &p;
  Types &type;
  Scales &scale;
  TermId insert(Op op, TermId other, TermId raw) {
    TermId e = p.make_term(Op::Encode, {raw});
    type[e] = Type::Plain;
    scale[e] = (op == Op::Add || op == Op::Sub) ? scale.get(other) : scale.get(raw);
    p.at(e).has_encode_scale = true;
    p.at(e).encode_scale = scale.get(e);
    return e;
  }
  void operator()(TermId t) {
    if (p.at(t).operands.size() != 2) return;
    TermId l = p.at(t).operands[0], r = p.at(t).operands[1];
    Op op = p.at(t).op;
    if (type.get(l) == Type::Cipher && type.get(r) == Type::Raw) p.replace_operand(t, r, insert(op, l, r));
    l = p.at(t).operands[0];
    r = p.at(t).operands[1];
    if (type.get(r) == Type::Cipher && type.get(l) == Type::Raw) p.replace_operand(t, l, insert(op, r, l));
  }
};

// ckks/lazy_relinearizer.h:11-96 and eager_relinearizer.h:11-54
struct Relinearizer {
  Program &p;
  Types &type;
  Scales &scale;
  bool lazy;
  TermTable<char> pending{0};
  bool encrypted_mul(TermId t) {
    if (p.at(t).op != Op::Mul) return false;
    for (TermId o : p.at(t).operands)
      if (type.get(o) != Type::Cipher) return false;
    return true;
  }
  void insert(TermId t) {
    TermId r = p.make_term(Op::Relinearize, {t});
    type[r] = type.get(t);
    scale[r] = scale.get(t);
    p.replace_other_uses_with(t, r);
  }
  void operator()(TermId t) {
    if (p.at(t).operands.empty()) return;
    if (!lazy) {
      if (encrypted_mul(t)) insert(t);
      return;
    }
    if (encrypted_mul(t)) pending[t] = 1;
    else if (!pending.get(t)) return;
    auto uses = p.at(t).uses;
    if (uses.empty()) throw std::logic_error("relinearizer: term without uses");
    bool must = false;
    for (TermId u : uses) {
      Op uo = p.at(u).op;
      if (encrypted_mul(u) || uo == Op::RotateLeftConst || uo == Op::RotateRightConst || uo == Op::Output || u != uses[0]) {
        must = true;
        break;
      }
    }
    if (must) insert(t);
    else for (TermId u : uses) pending[u] = 1;
  }
};

// ckks/mod_switcher.h:11-96 — backward pass; levels counted from the outputs, ModSwitch chains
// inserted on edges whose consumer sits deeper; sources/Encode nodes get EncodeAtLevel
struct ModSwitcher {
  Program &p;
  Types &type;
  Scales &scale;
  TermTable<uint32_t> level{0};
  std::vector<TermId> encodes;
  void operator()(TermId t) {
    if (p.at(t).uses.empty()) return;
    if (type.get(t) == Type::Raw) return;
    if (p.at(t).op == Op::Encode) encodes.push_back(t);
    std::map<uint32_t, std::vector<TermId>> by_level;
    for (TermId u : p.uses_of(t)) by_level[level.get(u)].push_back(u);
    uint32_t tl = 0;
    if (by_level.size() > 1) {
      auto it = by_level.rbegin();
      tl = it->first;
      ++it;
      TermId cur = t;
      uint32_t cl = tl;
      for (; it != by_level.rend(); ++it) {
        while (cl > it->first) {
          TermId m = p.make_term(Op::ModSwitch, {cur});
          scale[m] = scale.get(cur);
          type[m] = type.get(t);
          level[m] = cl;
          cur = m;
          --cl;
        }
        for (TermId u : it->second) p.replace_operand(u, t, cur);
      }
    } else {
      tl = by_level.begin()->first;
    }
    if (p.at(t).op == Op::Rescale) ++tl;
    level[t] = tl;
  }
  void finish() {
    uint32_t mx = 0;
    auto src = p.sources();
    for (TermId s : src) mx = std::max(mx, level.get(s));
    for (TermId s : src) { p.at(s).has_encode_level = true; p.at(s).encode_level = mx - level.get(s); }
    for (TermId e : encodes) { p.at(e).has_encode_level = true; p.at(e).encode_level = mx - level.get(e); }
  }
};

// ckks/seal_lowering.h:11-30 — plain - cipher  ->  plain + (-cipher)
struct BackendLowering {
  Program &p;
  Types &type;
  void operator()(TermId t) {
    const Term &x = p.at(t);
    if (x.op == Op::Sub && type.get(x.operands[0]) != Type::Cipher && type.get(x.operands[1]) == Type::Cipher) {
      TermId a = x.operands[0], b = x.operands[1];
      TermId n = p.make_term(Op::Negate, {b});
      TermId add = p.make_term(Op::Add, {a, n});
      p.replace_all_uses_with(t, add);
    }
  }
};

struct InconsistentParameters : std::runtime_error {
  using std::runtime_error::runtime_error;
};

// ckks/levels_checker.h — all Cipher operands of a term sit at the same level
struct LevelsChecker {
  Program &p;
  Types &types;
  TermTable<uint32_t> levels{0};
  void operator()(TermId t) {
    const Term &x = p.at(t);
    if (x.operands.empty()) { levels[t] = x.encode_level; return; }
    bool have = false;
    uint32_t lv = 0;
    for (TermId o : x.operands)
      if (types.get(o) == Type::Cipher) {
        if (!have) { lv = levels.get(o); have = true; }
        else if (levels.get(o) != lv) throw std::logic_error("Compiled program has Cipher operands at different levels");
      }
    if (x.op == Op::Rescale || x.op == Op::ModSwitch) ++lv;
    levels[t] = lv;
  }
};

// ckks/parameter_checker.h — operands agree on the primes consumed so far
struct ParameterChecker {
  Program &p;
  Types &types;
  TermTable<std::vector<uint32_t>> parms;
  void operator()(TermId t) {
    const Term &x = p.at(t);
    if (types.get(t) == Type::Raw || x.op == Op::Encode) return;
    std::vector<uint32_t> mine;
    if (!x.operands.empty()) {
      for (TermId o : x.operands) {
        const auto op = parms.get(o);
        if (op.empty()) continue;
        if (mine.empty()) { mine = op; continue; }
        if (op.size() != mine.size()) throw InconsistentParameters("Two operands require different number of primes");
        for (size_t i = 0; i < mine.size(); i++) {
          if (mine[i] == 0) mine[i] = op[i];
          else if (op[i] != 0 && mine[i] != op[i]) throw InconsistentParameters("Primes required by two operands do not match");
        }
      }
      if (x.op == Op::ModSwitch) mine.push_back(0);
      else if (x.op == Op::Rescale) mine.push_back(x.rescale_divisor);
    } else {
      mine.assign(x.encode_level, 0);
    }
    parms[t] = mine;
  }
};

// ckks/scales_checker.h
struct ScalesChecker {
  Program &p;
  Types &types;
  TermTable<uint32_t> s{0};
  void operator()(TermId t) {
    const Term &x = p.at(t);
    if (types.get(t) == Type::Raw) return;
    auto nz = [](uint32_t v) { if (v == 0) throw std::logic_error("Compiled program results in a 0 scale term"); return v; };
    if (x.op == Op::Input || x.op == Op::Encode) {
      if (x.encode_scale == 0) {
        if (x.op == Op::Input) throw std::runtime_error("Program has an input with 0 scale");
        throw std::logic_error("Compiled program results in a 0 scale term");
      }
      s[t] = x.encode_scale;
    } else if (x.op == Op::Mul) {
      uint32_t v = 0;
      for (TermId o : x.operands) v += s.get(o);
      s[t] = nz(v);
    } else if (x.op == Op::Rescale) {
      s[t] = nz(s.get(x.operands[0]) - x.rescale_divisor);
    } else if (x.op == Op::Add || x.op == Op::Sub) {
      uint32_t v = 0;
      for (TermId o : x.operands) {
        if (v == 0) v = s.get(o);
        else if (v != s.get(o)) throw std::logic_error("Addition or subtraction in program has operands of non-equal scale");
      }
      s[t] = nz(v);
    } else {
      s[t] = nz(s.get(x.operands[0]));
    }
  }
};

// ckks/encryption_parameter_selector.h:15-208 — prime bit sizes: [output primes..., rescale
// primes in reverse program order..., special]
struct EncryptionParametersSelector {
  Program &p;
  Scales &scales;
  Types &types;
  TermTable<std::vector<uint32_t>> terms;
  void operator()(TermId t) {
    const Term &x = p.at(t);
    if (types.get(t) == Type::Raw || x.op == Op::Encode) return;
    if (x.operands.empty()) return;
    std::vector<uint32_t> parms;
    for (TermId o : x.operands) {
      auto op = terms.get(o);
      if (op.size() > parms.size()) parms = op;
    }
    if (x.op == Op::Rescale) parms.push_back(x.rescale_divisor);
    terms[t] = parms;
  }
  std::vector<uint32_t> result() {
    std::vector<uint32_t> parms;
    uint32_t max_out = 0, max_parm = 0;
    size_t max_len = 0;
    for (auto &kv : p.outputs()) {
      const Term &o = p.at(kv.second);
      max_out = std::max(max_out, o.range + scales.get(kv.second));
      auto op = terms.get(kv.second);
      max_len = std::max(max_len, op.size());
      for (uint32_t v : op) max_parm = std::max(max_parm, v);
    }
    if (max_out > 60) {
      max_parm = 60;
      while (max_out >= 60) { parms.push_back(60); max_out -= 60; }
      if (max_out > 0) parms.push_back(std::max(20u, max_out));
    } else {
      max_parm = std::max(max_parm, max_out);
      parms.push_back(max_parm);
    }
    for (auto &kv : p.outputs()) {
      auto op = terms.get(kv.second);
      if (op.size() == max_len) {
        parms.insert(parms.end(), op.rbegin(), op.rend());
        break;
      }
    }
    parms.push_back(max_parm);
    return parms;
  }
};

// common/rotation_keys_selector.h:15-55 — left = +r, right = -r
struct RotationKeysSelector {
  Program &p;
  Types &type;
  std::set<int> keys;
  void operator()(TermId t) {
    const Term &x = p.at(t);
    if (x.op != Op::RotateLeftConst && x.op != Op::RotateRightConst) return;
    if (type.get(t) == Type::Raw) return;
    keys.insert(x.op == Op::RotateRightConst ? -x.rotation : x.rotation);
  }
};

// ---- configuration (ckks/ckks_config.h:12-38, ckks_config.cpp:15-77)
enum class Rescaler { LazyWaterline, EagerWaterline, Always, Minimum };
struct CKKSConfig {
  bool balance_reductions = true;
  Rescaler rescaler = Rescaler::LazyWaterline;
  bool lazy_relinearize = true;
  uint32_t security_level = 128;
  bool quantum_safe = false;
  bool warn_vec_size = true;
  CKKSConfig() {}
  explicit CKKSConfig(const std::unordered_map<std::string, std::string> &m) {
    auto as_bool = [](const std::string &v, bool &out) {
      std::istringstream is(v);
      bool b;
      is >> std::boolalpha >> b;
      if (is.fail()) return false;
      out = b;
      return true;
    };
    for (auto &kv : m) {
      const auto &o = kv.first, &v = kv.second;
      if (o == "balance_reductions") { if (!as_bool(v, balance_reductions)) warn("Could not parse boolean in balance_reductions=" + v + ". Falling back to default."); }
      else if (o == "rescaler") {
        if (v == "lazy_waterline") rescaler = Rescaler::LazyWaterline;
        else if (v == "eager_waterline") rescaler = Rescaler::EagerWaterline;
        else if (v == "always") rescaler = Rescaler::Always;
        else if (v == "minimum") rescaler = Rescaler::Minimum;
        else warn("Unknown value rescaler=" + v + ". Available rescalers are lazy_waterline, eager_waterline, always, minimum. Falling back to default.");
      } else if (o == "lazy_relinearize") { if (!as_bool(v, lazy_relinearize)) warn("Could not parse boolean in lazy_relinearize=" + v + ". Falling back to default."); }
      else if (o == "security_level") {
        std::istringstream is(v);
        is >> security_level;
        if (is.fail()) throw std::runtime_error("Could not parse unsigned int in security_level=" + v);
      } else if (o == "quantum_safe") { if (!as_bool(v, quantum_safe)) throw std::runtime_error("Could not parse boolean in quantum_safe=" + v); }
      else if (o == "warn_vec_size") { if (!as_bool(v, warn_vec_size)) warn("Could not parse boolean in warn_vec_size=" + v + ". Falling back to default."); }
      else warn("Unknown option " + o);
    }
  }
  std::string to_string(int indent = 0) const {
    std::string in(indent, ' ');
    std::ostringstream s;
    s << std::boolalpha << in << "balance_reductions = " << balance_reductions << '\n' << in << "rescaler = ";
    switch (rescaler) {
    case Rescaler::LazyWaterline: s << "lazy_waterline"; break;
    case Rescaler::EagerWaterline: s << "eager_waterline"; break;
    case Rescaler::Always: s << "always"; break;
    case Rescaler::Minimum: s << "minimum"; break;
    }
    s << '\n' << in << "lazy_relinearize = " << lazy_relinearize << '\n' << in << "security_level = " << security_level
      << '\n' << in << "quantum_safe = " << quantum_safe << '\n' << in << "warn_vec_size = " << warn_vec_size;
    return s.str();
  }
};

// ckks/ckks_parameters.h:14-18, ckks_signature.h:16-34
struct CKKSParameters {
  std::vector<uint32_t> prime_bits;
  std::set<int> rotations;
  uint32_t poly_modulus_degree = 0;
};
struct CKKSEncodingInfo {
  Type input_type;
  int scale, level;
};
struct CKKSSignature {
  int vec_size = 0;
  std::unordered_map<std::string, CKKSEncodingInfo> inputs;
};

// HomomorphicEncryption.org standard: max total coeff-modulus bits per degree
// (what SEAL's seal_he_std_parms_{128,192,256}_{tc,tq} return; used at ckks_compiler.h:175-193)
inline int he_std_max_bits(uint32_t sec, bool quantum, size_t degree) {
  static const int tc128[] = {27, 54, 109, 218, 438, 881}, tc192[] = {19, 37, 75, 152, 305, 611},
                   tc256[] = {14, 29, 58, 118, 237, 476}, tq128[] = {25, 51, 101, 202, 411, 827},
                   tq192[] = {17, 35, 70, 141, 284, 571}, tq256[] = {13, 27, 54, 109, 220, 443};
  int idx = -1;
  for (int i = 0; i < 6; i++)
    if (degree == (size_t)1024 << i) idx = i;
  if (idx < 0) return 0;
  const int *tab = sec <= 128 ? (quantum ? tq128 : tc128) : sec <= 192 ? (quantum ? tq192 : tc192) : (quantum ? tq256 : tc256);
  return tab[idx];
}

// ckks/ckks_compiler.h:36-306
class CKKSCompiler {
public:
  CKKSConfig config;
  CKKSCompiler() {}
  explicit CKKSCompiler(CKKSConfig c) : config(c) {}

  std::tuple<std::unique_ptr<Program>, CKKSParameters, CKKSSignature> compile(const Program &input) {
    auto prog = input.deep_copy();
    Program &p = *prog;
    Types types(Type::Undef);
    Scales scales(0);
    for (TermId s : p.sources()) {
      if (!p.at(s).has_encode_scale) {
        for (auto &kv : p.inputs())
          if (kv.second == s) throw std::runtime_error("The scale for input " + kv.first + " was not set.");
        throw std::runtime_error("The scale for a constant was not set.");
      }
      scales[s] = p.at(s).encode_scale;
    }
    transform(p, types, scales);
    validate(p, types, scales);
    CKKSParameters params = determine_parameters(p, types, scales);
    CKKSSignature sig;
    sig.vec_size = (int)p.vec_size();
    for (auto &kv : p.inputs()) {
      const Term &x = p.at(kv.second);
      sig.inputs.emplace(kv.first, CKKSEncodingInfo{x.type_attr, (int)x.encode_scale, (int)x.encode_level});
    }
    return std::make_tuple(std::move(prog), std::move(params), std::move(sig));
  }

private:
  void transform(Program &p, Types &types, Scales &scales) {
    forward_pass(p, TypeDeducer{p, types});
    forward_pass(p, ConstantFolder{p, scales, {}, {}});
    if (config.balance_reductions) {
      forward_pass(p, ReductionCombiner{p});
      ReductionLogExpander rle{p, types};
      forward_pass(p, rle);
    }
    switch (config.rescaler) {
    case Rescaler::Minimum: { MinimumRescaler r(p, types, scales); forward_pass(p, r); } break;
    case Rescaler::Always: { AlwaysRescaler r(p, types, scales); forward_pass(p, r); } break;
    case Rescaler::EagerWaterline: { EagerWaterlineRescaler r(p, types, scales); forward_pass(p, r); } break;
    case Rescaler::LazyWaterline: { LazyWaterlineRescaler r(p, types, scales); forward_pass(p, r); } break;
    }
    forward_pass(p, TypeDeducer{p, types});
    forward_pass(p, EncodeInserter{p, types, scales});
    forward_pass(p, TypeDeducer{p, types});
    {
      Relinearizer r{p, types, scales, config.lazy_relinearize};
      forward_pass(p, r);
    }
    forward_pass(p, TypeDeducer{p, types});
    {
      ModSwitcher ms{p, types, scales, TermTable<uint32_t>(0), {}};
      backward_pass(p, ms);
      ms.finish();
    }
    forward_pass(p, TypeDeducer{p, types});
    // Every term of a snapshot of the order is offered to the lowering.  (The reference runs it
    // through ProgramTraversal, whose ready-list never reaches the Negate / Add it creates, so a
    // second plaintext-minus-ciphertext downstream of a lowered one stays a Sub that its executor
    // cannot run — seal_lowering.h:24-30, program_traversal.h:36-88; found by tests/test_gpu_fuzz.py.)
    {
      BackendLowering lower{p, types};
      for (TermId t : p.topo_order()) lower(t);
      p.gc();
    }
    forward_pass(p, TypeDeducer{p, types});
  }

  void validate(Program &p, Types &types, Scales &scales) {
    LevelsChecker lc{p, types};
    forward_pass(p, lc);
    try {
      ParameterChecker pc{p, types, TermTable<std::vector<uint32_t>>{}};
      forward_pass(p, pc);
    } catch (const InconsistentParameters &) {
      switch (config.rescaler) {
      case Rescaler::Minimum:
        throw std::runtime_error("The 'minimum' rescaler produced inconsistent parameters. Note that this rescaling policy is not general and thus will not work for all programs. Please use a different rescaler for this program.");
      case Rescaler::Always:
        throw std::runtime_error("The 'always' rescaler produced inconsistent parameters. Note that this rescaling policy is not general. It is only guaranteed to work for programs that have equal scale for all inputs and constants.");
      default:
        throw std::runtime_error("The current rescaler produced inconsistent parameters. This is a bug, as this rescaler should be able to handle all programs.");
      }
    }
    ScalesChecker sc{p, types};
    forward_pass(p, sc);
    (void)scales;
  }

  size_t min_degree_for_bits(int bits) {
    size_t degree = 1024;
    int seen = 0;
    while (true) {
      int mx = he_std_max_bits(config.security_level, config.quantum_safe, degree);
      seen = std::max(seen, mx);
      if (mx == 0)
        throw std::runtime_error("Program requires a " + std::to_string(bits) + " bit modulus, but parameters are available for a maximum of " + std::to_string(seen));
      if (mx >= bits) return degree;
      degree *= 2;
    }
  }

  CKKSParameters determine_parameters(Program &p, Types &types, Scales &scales) {
    if (config.security_level > 256)
      throw std::runtime_error("EVA has support for up to 256 bit security, but " + std::to_string(config.security_level) + " bit security was requested.");
    EncryptionParametersSelector eps{p, scales, types, TermTable<std::vector<uint32_t>>{}};
    forward_pass(p, eps);
    RotationKeysSelector rks{p, types, {}};
    forward_pass(p, rks);
    CKKSParameters ep;
    ep.prime_bits = eps.result();
    ep.rotations = rks.keys;
    int bits = 0;
    for (auto b : ep.prime_bits) bits += (int)b;
    ep.poly_modulus_degree = (uint32_t)min_degree_for_bits(bits);
    uint32_t slots = ep.poly_modulus_degree / 2;
    if (config.warn_vec_size && slots > p.vec_size())
      warn("Program specifies vector size " + std::to_string(p.vec_size()) + " while at least " + std::to_string(slots) +
           " slots are required for security. This does not affect correctness, as the smaller vector size will be transparently emulated. However, using a vector size up to " +
           std::to_string(slots) + " would come at no additional cost.");
    if (slots < p.vec_size()) {
      if (config.warn_vec_size)
        warn("Program uses vector size " + std::to_string(p.vec_size()) + " while only " + std::to_string(slots) +
             " slots are required for security. This does not affect correctness, but higher performance may be available with a smaller vector size.");
      ep.poly_modulus_degree = 2 * p.vec_size();
    }
    if (verbosity() >= 1) {
      std::printf("EVA: Encryption parameters for %s are:\n  Q = [", p.name().c_str());
      for (size_t i = 0; i < ep.prime_bits.size(); i++) std::printf(i ? ",%u" : "%u", ep.prime_bits[i]);
      std::printf("] (total bits %d)\n  N = %u (available slots %u)\n  Rotation keys: %zu\n", bits, ep.poly_modulus_degree,
                  ep.poly_modulus_degree / 2, ep.rotations.size());
    }
    return ep;
  }
};

} // namespace evahost
