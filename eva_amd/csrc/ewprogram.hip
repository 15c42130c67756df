// ewprogram.hip — libeva_hip.so: evah_elementwise_program, a straight-line program of elementwise evaluator calls
// (add / sub / negate / multiply / square and their plaintext forms — seal::Evaluator::add, sub, add_plain, sub_plain,
// negate, multiply, square, multiply_plain as SEALExecutor calls them, /root/reference/eva/seal/seal_executor.h:114-175,
// :191-195) evaluated in ONE launch.
//
// EVA programs end their key switches in runs of such calls on values nobody else reads: Harris' response
// det - k trace^2 is seven of them on three ciphertexts (/root/reference/examples/image_processing.py:92-100), Sobel's
// gradient magnitude and its polynomial square root a dozen (:39-63).  One launch each, every call reads its operands
// from HBM and writes its result back: at N = 2^15 the seven launches of Harris are latency (5 us each for 3 MB), on a
// 32-instance batched handle of config 4 they are bytes (a ciphertext is 42 MB).  Here the ciphertext-level calls are
// lowered to instructions on POLYNOMIAL registers — two coefficients per thread, the registers in LDS (2-4 KB each per
// workgroup, so an instruction's operands are addressed by register number) — and interpreted by one kernel: every input
// word is read once, only the values that leave the program are written.  Every instruction produces the canonical
// residue the corresponding kernel of elementwise.hip produces (add / sub / negate: compare-and-subtract; products:
// 128-bit product, Barrett; a0 b1 + a1 b0 accumulated in 128 bits and reduced once, as k_mul22 does), so the outputs are
// the bits of the separate calls.  Checks and error messages are those of the separate entry points, in program order.
#include "launch.hip.h"

namespace evah {

constexpr int EW_MAX_INS = 160;  // polynomial instructions per launch
constexpr int EW_MAX_PTR = 48;   // polynomials loaded or stored per launch
// polynomial registers: 8 or 16 bytes of LDS per thread each (one or two coefficients per thread).  With two, up to 14
// registers a workgroup is 256 threads (4 KB per register), up to 28 it is 128 threads: at most 56 KB of dynamic LDS
constexpr int EW_MAX_REGS = 28, EW_REGS_WIDE = 14;
constexpr int EW_MAX_SC = 16;    // uniform plaintexts (scalars of Z_q per limb) multiplied by per launch
enum EwOp : uint32_t { EW_LOAD = 0, EW_STORE, EW_ADD, EW_SUB, EW_NEG, EW_MUL, EW_FMA2, EW_MULU };

struct EwProg {
  uint32_t n_ins;
  // instruction j: w[2j] = op | dst << 8 | a << 16 | b << 24, w[2j + 1] = c | d << 8
  //   LOAD  dst <- ptr[a]          STORE ptr[a] <- reg b
  //   ADD / SUB dst <- a (+/-) b   NEG dst <- -a      MUL dst <- a b      FMA2 dst <- a b + c d
  //   MULU dst <- a * scalar b     (r6: a uniform plaintext — evah_pt_uniform, every word of limb i the same residue W_i — is a
  //                                 scalar of Z_{q_i}: its polynomial is never loaded, the product is a Shoup product with the
  //                                 quotient floor(W_i 2^64 / q_i) the workgroup computes once; same canonical residue as MUL)
  uint32_t w[2 * EW_MAX_INS];
  uint32_t n_sc;                 // scalars: sc_ptr[s] = pointer slot of the uniform plaintext (word 0 of limb i is W_i)
  uint32_t sc_reg0;              // the scalar table starts where register sc_reg0 would (behind the register file in LDS)
  uint8_t sc_ptr[EW_MAX_SC];
  u64 *ptr[EW_MAX_PTR];          // polynomial bases (instance 0, limb 0)
  uint32_t bstride[EW_MAX_PTR];  // distance between the instances of a batched handle, in units of N words (0: shared plaintext)
};

// V coefficients per thread.  V = 2 (throughput-sized launches): every interpreted instruction — the scalar fetch of its
// words, the dispatch on its opcode — serves 128 coefficients of a wave instead of 64, global accesses are 16 bytes per lane
// as in the dedicated elementwise kernels, and the two independent modular products of a MUL overlap in the VALU pipeline.
// V = 1 (small launches, which are bound by the latency of ONE wave's instruction stream: Harris' response at N = 2^15 is
// 34 instructions): half the work per instruction, twice the waves.  A thread touches only its own column of the register
// file, so no barrier is needed anywhere.
template <int V> struct EwVec;
template <> struct EwVec<1> {
  using T = u64;
  static __device__ __forceinline__ T load(const u64 *p) { return *p; }
  static __device__ __forceinline__ void store(u64 *p, T v) { *p = v; }
  template <class F> static __device__ __forceinline__ T map1(T x, F f) { return f(x); }
  template <class F> static __device__ __forceinline__ T map2(T x, T y, F f) { return f(x, y); }
  template <class F> static __device__ __forceinline__ T map4(T x, T y, T z, T u, F f) { return f(x, y, z, u); }
};
template <> struct EwVec<2> {
  using T = ulonglong2;
  static __device__ __forceinline__ T load(const u64 *p) { return ld2(p); }
  static __device__ __forceinline__ void store(u64 *p, T v) { st2(p, v); }
  template <class F> static __device__ __forceinline__ T map1(T x, F f) { return make_ulonglong2(f(x.x), f(x.y)); }
  template <class F> static __device__ __forceinline__ T map2(T x, T y, F f) { return make_ulonglong2(f(x.x, y.x), f(x.y, y.y)); }
  template <class F> static __device__ __forceinline__ T map4(T x, T y, T z, T u, F f) {
    return make_ulonglong2(f(x.x, y.x, z.x, u.x), f(x.y, y.y, z.y, u.y));
  }
};
template <int V>
__global__ void __launch_bounds__(256)
k_ew_program(DevCtx cx, EwProg pg) {
  using VT = typename EwVec<V>::T;
  extern __shared__ __attribute__((aligned(16))) u64 ew_lds[];
  VT *ew_regs = reinterpret_cast<VT *>(ew_lds); // [register][thread]
  const uint32_t i = blockIdx.y, inst = blockIdx.z, tid = threadIdx.x, T = blockDim.x;
  const size_t off = (size_t)i * cx.N + V * ((size_t)blockIdx.x * T + tid);
  const DevPrime pm = cx.primes[cx.prime_of(i)];
  // the launch's scalars with their Shoup quotients, behind the register file: (W, floor(W 2^64 / q)) — the quotient from
  // floor(2^128 / q) (DevPrime::r0, r1: at most 2 below), corrected with the remainder
  ulonglong2 *ew_sc = reinterpret_cast<ulonglong2 *>(ew_regs + (size_t)pg.sc_reg0 * T);
  if (pg.n_sc) { // block-uniform; the index into the kernel arguments stays wave-uniform (scalar loads)
    for (uint32_t sI = 0; sI < pg.n_sc; sI++) {
      const u64 w = pg.ptr[pg.sc_ptr[sI]][(size_t)i * cx.N];
      u64 e = w * pm.r1 + __umul64hi(w, pm.r0); // floor(W floor(2^128 / q) / 2^64): the quotient or one below
      u64 rem = 0ull - e * pm.q;                // W 2^64 - e q  (mod 2^64; the true value is below 2 q)
      while (rem >= pm.q) { rem -= pm.q; e++; }
      if (tid == 0) ew_sc[sI] = make_ulonglong2(w, e);
    }
    __syncthreads();
  }
  uint32_t w0 = pg.w[0], w1 = pg.w[1];
  for (uint32_t pc = 0; pc < pg.n_ins; pc++) {
    // wave-uniform: scalar loads from the kernel arguments; the next instruction's words are requested before this one runs
    const uint32_t nx = pc + 1 < pg.n_ins ? pc + 1 : pc;
    const uint32_t n0 = pg.w[2 * nx], n1 = pg.w[2 * nx + 1];
    const uint32_t op = w0 & 0xffu, dst = (w0 >> 8) & 0xffu, a = (w0 >> 16) & 0xffu, b = w0 >> 24;
    VT *rd = ew_regs + dst * T + tid;
    switch (op) {
    case EW_LOAD: *rd = EwVec<V>::load(pg.ptr[a] + off + (size_t)inst * pg.bstride[a] * cx.N); break;
    case EW_STORE: EwVec<V>::store(pg.ptr[a] + off + (size_t)inst * pg.bstride[a] * cx.N, ew_regs[b * T + tid]); break;
    case EW_ADD: *rd = EwVec<V>::map2(ew_regs[a * T + tid], ew_regs[b * T + tid], [&](u64 x, u64 y) { return addmod(x, y, pm.q); }); break;
    case EW_SUB: *rd = EwVec<V>::map2(ew_regs[a * T + tid], ew_regs[b * T + tid], [&](u64 x, u64 y) { return submod(x, y, pm.q); }); break;
    case EW_NEG: *rd = EwVec<V>::map1(ew_regs[a * T + tid], [&](u64 x) { return negmod(x, pm.q); }); break;
    case EW_MUL: *rd = EwVec<V>::map2(ew_regs[a * T + tid], ew_regs[b * T + tid], [&](u64 x, u64 y) { return mulmod(x, y, pm); }); break;
    case EW_MULU: {
      const ulonglong2 sc = ew_sc[b];
      *rd = EwVec<V>::map1(ew_regs[a * T + tid], [&](u64 x) { return mul_shoup(x, sc.x, sc.y, pm.q); });
    } break;
    default: { // EW_FMA2
      const uint32_t cc = w1 & 0xffu, dd = (w1 >> 8) & 0xffu;
      *rd = EwVec<V>::map4(ew_regs[a * T + tid], ew_regs[b * T + tid], ew_regs[cc * T + tid], ew_regs[dd * T + tid],
                           [&](u64 x, u64 y, u64 z, u64 u) {
                             u128_t t = mul128(x, y);
                             acc128(t, z, u);
                             return barrett128(t, pm);
                           });
    } break;
    }
    w0 = n0;
    w1 = n1;
  }
}
static_assert(sizeof(DevCtx) + sizeof(EwProg) + 64 <= 4096, "kernel arguments of k_ew_program");

} // namespace evah

extern "C" {

int evah_elementwise_program(evah_ctx *c, const evah_val *in, uint32_t n_in, const evah_ew_op *ops, uint32_t n_ops, const uint32_t *out_vals,
                             uint32_t n_out, evah_ct **outs) {
  API_BEGIN
  use(c);
  if (n_out < 1) throw std::invalid_argument("elementwise program without outputs");
  // ---- values: shapes as the separate entry points would give them, checks in program order
  struct Val {
    int kind = 0; // 1 ciphertext, 2 plaintext
    uint32_t size = 0, limbs = 0, batch = 1;
    double scale = 0;
    const evah_ct *ct = nullptr; // inputs
    const evah_pt *pt = nullptr;
    int preg[3] = {-1, -1, -1};  // virtual register per polynomial (-1: not loaded / not computed yet)
  };
  std::vector<Val> vals(n_in + n_ops);
  for (uint32_t v = 0; v < n_in; v++) {
    Val &x = vals[v];
    if (in[v].kind == EVAH_VAL_CT && in[v].h) {
      const evah_ct *t = static_cast<const evah_ct *>(in[v].h);
      x.kind = 1; x.size = t->size; x.limbs = t->limbs; x.batch = t->batch; x.scale = t->scale; x.ct = t;
    } else if (in[v].kind == EVAH_VAL_PT && in[v].h) {
      const evah_pt *t = static_cast<const evah_pt *>(in[v].h);
      x.kind = 2; x.size = 1; x.limbs = t->limbs; x.scale = t->scale; x.pt = t;
    } else {
      throw std::invalid_argument("elementwise program: input " + std::to_string(v) + " is empty");
    }
  }
  struct PIns { uint32_t op; int dst, a, b, cc, dd; };
  std::vector<PIns> pins;
  struct PPtr { u64 *p; uint32_t bstride; };
  std::vector<PPtr> pptr;
  int n_vreg = 0;
  const size_t N = c->N;
  auto poly = [&](uint32_t v, uint32_t p) -> int { // virtual register holding polynomial p of value v (loaded on first use)
    Val &x = vals[v];
    if (x.preg[p] >= 0) return x.preg[p];
    if (!x.ct && !x.pt) throw std::logic_error("elementwise program: polynomial of a computed value is missing");
    u64 *base = x.ct ? x.ct->d + (size_t)p * x.ct->ps : x.pt->d;
    const uint32_t bs = x.ct ? (uint32_t)((size_t)x.ct->size * x.ct->ps / N) : 0u;
    pptr.push_back({base, bs});
    x.preg[p] = n_vreg++;
    pins.push_back({EW_LOAD, x.preg[p], (int)pptr.size() - 1, 0, 0, 0});
    return x.preg[p];
  };
  auto emit = [&](uint32_t op, int a, int b = 0, int cc = 0, int dd = 0) {
    const int d = n_vreg++;
    pins.push_back({op, d, a, b, cc, dd});
    return d;
  };
  uint32_t limbs = 0, batch = 0; // of the program's ciphertexts: one grid
  auto same_grid = [&](const Val &x) {
    if (x.kind != 1) return;
    if (!limbs) { limbs = x.limbs; batch = x.batch; }
  };
  for (uint32_t j = 0; j < n_ops; j++) {
    const evah_ew_op &o = ops[j];
    const uint32_t dst = n_in + j;
    if (o.a >= dst || (o.op != 10 && o.b >= dst)) throw std::invalid_argument("elementwise program: an operand is used before it is defined");
    uint32_t ia = o.a, ib = o.op == 10 ? o.a : o.b;
    Val r;
    r.kind = 1;
    if (o.op == 10) { // negate (seal_executor.h:191-195)
      const Val &a = vals[ia];
      if (a.kind != 1) throw std::runtime_error("Unsupported operation encountered");
      r.size = a.size; r.limbs = a.limbs; r.batch = a.batch; r.scale = a.scale;
      for (uint32_t p = 0; p < a.size; p++) r.preg[p] = emit(EW_NEG, poly(ia, p));
    } else if (o.op == 11 || o.op == 12 || o.op == 13) {
      // the ciphertext first (seal_executor.h:116-119, :155-158); sub keeps its order (:137-150)
      if (o.op != 12 && vals[ia].kind != 1) std::swap(ia, ib);
      const Val &a = vals[ia], &b = vals[ib];
      if (a.kind != 1) throw std::runtime_error("Unsupported operation encountered");
      if (b.kind == 1) {
        if (a.limbs != b.limbs) throw std::invalid_argument("encrypted1 and encrypted2 parameter mismatch");
        if (o.op == 13) {
          if (ia == ib) {
            if (a.size != 2) throw std::invalid_argument("square supports size-2 operands only (relinearize first)");
          } else if (a.size != 2 || b.size != 2) {
            throw std::invalid_argument("multiply supports size-2 operands only (relinearize first)");
          }
          r.scale = a.scale * b.scale;
          check_scale(c, r.scale, a.limbs);
          if (a.batch != b.batch) throw std::invalid_argument("batch size mismatch");
          r.size = 3; r.limbs = a.limbs; r.batch = a.batch;
          const int a0 = poly(ia, 0), a1 = poly(ia, 1), b0 = poly(ib, 0), b1 = poly(ib, 1);
          r.preg[0] = emit(EW_MUL, a0, b0);
          r.preg[1] = emit(EW_FMA2, a0, b1, a1, b0);
          r.preg[2] = emit(EW_MUL, a1, b1);
        } else {
          if (!same_scale(a.scale, b.scale)) throw std::invalid_argument("scale mismatch");
          if (a.batch != b.batch) throw std::invalid_argument("batch size mismatch");
          r.size = std::max(a.size, b.size); r.limbs = a.limbs; r.batch = a.batch; r.scale = a.scale;
          for (uint32_t p = 0; p < r.size; p++) {
            if (p < a.size && p < b.size) r.preg[p] = emit(o.op == 11 ? EW_ADD : EW_SUB, poly(ia, p), poly(ib, p));
            else if (p < a.size) r.preg[p] = poly(ia, p);                 // the longer operand's extra polynomial as it is
            else r.preg[p] = o.op == 11 ? poly(ib, p) : emit(EW_NEG, poly(ib, p)); // ... negated when it is the subtrahend
          }
        }
      } else {
        if (a.limbs != b.limbs) throw std::invalid_argument("encrypted and plain parameter mismatch");
        r.size = a.size; r.limbs = a.limbs; r.batch = a.batch;
        if (o.op == 13) {
          r.scale = a.scale * b.scale;
          check_scale(c, r.scale, a.limbs);
          if (b.pt->uniform && c->tun.ew_uniform) { // a scalar per limb: no load of the plaintext, Shoup products
            pptr.push_back({b.pt->d, 0u});
            const int ps = (int)pptr.size() - 1;
            for (uint32_t p = 0; p < a.size; p++) r.preg[p] = emit(EW_MULU, poly(ia, p), 0, ps);
          } else {
            const int w = poly(ib, 0);
            for (uint32_t p = 0; p < a.size; p++) r.preg[p] = emit(EW_MUL, poly(ia, p), w);
          }
        } else {
          if (!same_scale(a.scale, b.scale)) throw std::invalid_argument("scale mismatch");
          r.scale = a.scale;
          r.preg[0] = emit(o.op == 11 ? EW_ADD : EW_SUB, poly(ia, 0), poly(ib, 0));
          for (uint32_t p = 1; p < a.size; p++) r.preg[p] = poly(ia, p);
        }
      }
    } else {
      throw std::runtime_error("Unhandled op " + std::to_string(o.op) + " in an elementwise program");
    }
    same_grid(r);
    if (r.limbs != limbs || r.batch != batch) throw std::invalid_argument("elementwise program: its ciphertexts are not of one level and batch size");
    vals[dst] = r;
  }
  for (uint32_t v = 0; v < n_in; v++) {
    if (vals[v].ct) acquire(c, vals[v].ct->buf);
    if (vals[v].pt) acquire(c, vals[v].pt->buf);
  }
  // ---- outputs: stores of every polynomial (an output that is an input or shares a polynomial with one copies it)
  std::vector<evah_ct *> made(n_out, nullptr);
  try {
    for (uint32_t k = 0; k < n_out; k++) {
      if (out_vals[k] >= n_in + n_ops || vals[out_vals[k]].kind != 1) throw std::invalid_argument("elementwise program: an output is not a ciphertext of the program");
      Val &x = vals[out_vals[k]];
      same_grid(x);
      if (x.limbs != limbs || x.batch != batch) throw std::invalid_argument("elementwise program: its ciphertexts are not of one level and batch size");
      made[k] = ct_new(c, x.size, x.limbs, x.scale, x.batch);
      for (uint32_t p = 0; p < x.size; p++) {
        const int reg = poly(out_vals[k], p);
        pptr.push_back({made[k]->d + (size_t)p * made[k]->ps, (uint32_t)((size_t)x.size * made[k]->ps / N)});
        pins.push_back({EW_STORE, 0, (int)pptr.size() - 1, reg, 0, 0});
      }
    }
    // ---- liveness: instructions nobody needs go (values computed for their checks only), registers by linear scan
    std::vector<char> need(n_vreg, 0), keep(pins.size(), 0);
    for (size_t q = pins.size(); q-- > 0;) {
      const PIns &pi = pins[q];
      if (pi.op == EW_STORE) { keep[q] = 1; need[pi.b] = 1; continue; }
      if (!need[pi.dst]) continue;
      keep[q] = 1;
      if (pi.op == EW_LOAD) continue;
      need[pi.a] = 1;
      if (pi.op == EW_ADD || pi.op == EW_SUB || pi.op == EW_MUL || pi.op == EW_FMA2) need[pi.b] = 1;
      if (pi.op == EW_FMA2) { need[pi.cc] = 1; need[pi.dd] = 1; }
    }
    std::vector<PIns> live;
    for (size_t q = 0; q < pins.size(); q++) if (keep[q]) live.push_back(pins[q]);
    std::vector<int> last(n_vreg, -1);
    auto uses = [](const PIns &pi, int out[4]) {
      int n = 0;
      if (pi.op == EW_STORE) out[n++] = pi.b;
      else if (pi.op != EW_LOAD) {
        out[n++] = pi.a;
        if (pi.op != EW_NEG && pi.op != EW_MULU) out[n++] = pi.b;
        if (pi.op == EW_FMA2) { out[n++] = pi.cc; out[n++] = pi.dd; }
      }
      return n;
    };
    for (size_t q = 0; q < live.size(); q++) {
      int u[4];
      const int n = uses(live[q], u);
      for (int t = 0; t < n; t++) last[u[t]] = (int)q;
    }
    std::vector<int> phys(n_vreg, -1), free_regs;
    int n_regs = 0;
    // pointer slots in use (loads of dropped instructions leave holes: renumber)
    std::vector<int> ptr_map(pptr.size(), -1);
    std::vector<PPtr> ptr_used;
    bool fits = live.size() <= (size_t)EW_MAX_INS && N >= 256; // (a launch is N / 256 workgroups of 256 threads: smaller rings take the separate calls)
    EwProg pg{};
    for (size_t q = 0; q < live.size() && fits; q++) {
      PIns pi = live[q];
      int u[4];
      const int n = uses(pi, u);
      int pu[4] = {0, 0, 0, 0};
      for (int t = 0; t < n; t++) pu[t] = phys[u[t]];
      for (int t = 0; t < n; t++) // a register whose last reader this is can be the destination
        if (last[u[t]] == (int)q && phys[u[t]] >= 0) { free_regs.push_back(phys[u[t]]); phys[u[t]] = -1; }
      int pd = 0;
      if (pi.op != EW_STORE) {
        if (free_regs.empty()) pd = n_regs++;
        else { pd = free_regs.back(); free_regs.pop_back(); }
        phys[pi.dst] = pd;
        if (last[pi.dst] < 0) { free_regs.push_back(pd); phys[pi.dst] = -1; } // (cannot happen after the liveness pass)
      }
      if (pi.op == EW_LOAD || pi.op == EW_STORE) {
        if (ptr_map[pi.a] < 0) { ptr_map[pi.a] = (int)ptr_used.size(); ptr_used.push_back(pptr[pi.a]); }
      }
      int sc_idx = 0;
      if (pi.op == EW_MULU) { // the scalar's slot: one per distinct plaintext
        int slot = -1;
        for (size_t t = 0; t < ptr_used.size(); t++) if (ptr_used[t].p == pptr[pi.cc].p && ptr_used[t].bstride == 0) slot = (int)t;
        if (slot < 0) { slot = (int)ptr_used.size(); ptr_used.push_back(pptr[pi.cc]); }
        for (uint32_t t = 0; t < pg.n_sc; t++) if (pg.sc_ptr[t] == slot) sc_idx = (int)t + 1;
        if (!sc_idx) {
          if (pg.n_sc < (uint32_t)EW_MAX_SC) { pg.sc_ptr[pg.n_sc++] = (uint8_t)slot; sc_idx = (int)pg.n_sc; }
          else { fits = false; break; }
        }
        sc_idx--;
      }
      uint32_t w0 = pi.op, w1 = 0;
      if (pi.op == EW_LOAD) w0 |= (uint32_t)pd << 8 | (uint32_t)ptr_map[pi.a] << 16;
      else if (pi.op == EW_STORE) w0 |= (uint32_t)ptr_map[pi.a] << 16 | (uint32_t)pu[0] << 24;
      else if (pi.op == EW_MULU) w0 |= (uint32_t)pd << 8 | (uint32_t)pu[0] << 16 | (uint32_t)sc_idx << 24;
      else {
        w0 |= (uint32_t)pd << 8 | (uint32_t)pu[0] << 16 | (uint32_t)(pi.op == EW_NEG ? 0 : pu[1]) << 24;
        if (pi.op == EW_FMA2) w1 = (uint32_t)pu[2] | (uint32_t)pu[3] << 8;
      }
      pg.w[2 * q] = w0;
      pg.w[2 * q + 1] = w1;
      fits = fits && n_regs <= EW_MAX_REGS && ptr_used.size() <= (size_t)EW_MAX_PTR;
    }
    if (!fits) {
      // a program beyond one launch's registers / instructions / pointers: the separate entry points, call by call
      for (evah_ct *t : made) evah_ct_free(c, t);
      std::fill(made.begin(), made.end(), nullptr);
      std::vector<evah_ct *> tmp(n_in + n_ops, nullptr);
      struct Free { evah_ctx *c; std::vector<evah_ct *> &v; ~Free() { for (evah_ct *t : v) if (t) evah_ct_free(c, t); } } guard{c, tmp};
      auto ctv = [&](uint32_t v) -> const evah_ct * { return v < n_in ? static_cast<const evah_ct *>(in[v].h) : tmp[v]; };
      auto chk = [&](int rc) { if (rc) throw std::runtime_error(g_err); };
      for (uint32_t j = 0; j < n_ops; j++) {
        const evah_ew_op &o = ops[j];
        uint32_t ia = o.a, ib = o.op == 10 ? o.a : o.b;
        if ((o.op == 11 || o.op == 13) && vals[ia].kind != 1) std::swap(ia, ib);
        evah_ct *r = nullptr;
        if (o.op == 10) chk(evah_negate(c, ctv(ia), &r));
        else if (vals[ib].kind == 1) {
          if (o.op == 11) chk(evah_add(c, ctv(ia), ctv(ib), &r));
          else if (o.op == 12) chk(evah_sub(c, ctv(ia), ctv(ib), &r));
          else if (ia == ib) chk(evah_square(c, ctv(ia), &r));
          else chk(evah_multiply(c, ctv(ia), ctv(ib), &r));
        } else {
          const evah_pt *w = static_cast<const evah_pt *>(in[ib].h);
          if (o.op == 11) chk(evah_add_plain(c, ctv(ia), w, &r));
          else if (o.op == 12) chk(evah_sub_plain(c, ctv(ia), w, &r));
          else chk(evah_multiply_plain(c, ctv(ia), w, &r));
        }
        tmp[n_in + j] = r;
      }
      for (uint32_t k = 0; k < n_out; k++) {
        const evah_ct *src = ctv(out_vals[k]);
        evah_ct *o = new evah_ct(*src); // an alias: outputs that are inputs, or one value named twice, share the buffer
        o->buf->refs++;
        outs[k] = o;
      }
    } else {
      pg.n_ins = (uint32_t)live.size();
      static const bool ew_debug = std::getenv("EVAH_EW_DEBUG") != nullptr; // what each launch holds (scripts/ew_debug_probe.py)
      if (ew_debug) {
        std::string ops_s;
        for (uint32_t j = 0; j < n_ops; j++) ops_s += std::to_string(ops[j].op) + (j + 1 < n_ops ? "," : "");
        std::fprintf(stderr, "EVAH ew program: %u inputs, ops [%s], %u outputs -> %zu instructions, %d registers, %u scalars, limbs %u batch %u\n",
                     n_in, ops_s.c_str(), n_out, live.size(), n_regs, pg.n_sc, limbs, batch);
      }
      for (size_t q = 0; q < ptr_used.size(); q++) { pg.ptr[q] = ptr_used[q].p; pg.bstride[q] = ptr_used[q].bstride; }
      // throughput-sized launches: two coefficients per thread, 256 threads when the registers fit (else 128); small ones
      // (bound by the latency of one wave's instruction stream): one coefficient per thread, 256 threads
      const uint64_t coeffs = (uint64_t)N * limbs * batch;
      if (coeffs >= ((uint64_t)1 << 20) && N >= 512) { // (a 24-instance group of config 4 is 1.97 M coefficients)
        const uint32_t threads = n_regs <= EW_REGS_WIDE ? 256u : 128u;
        pg.sc_reg0 = (uint32_t)std::max(n_regs, 1);
        const size_t lds = (size_t)std::max(n_regs, 1) * threads * sizeof(ulonglong2) + EW_MAX_SC * sizeof(ulonglong2);
        EW_LAUNCH(k_ew_program<2>, dim3((unsigned)(N / (2 * threads)), limbs, batch), dim3(threads), lds, c->stream, c->dev, pg);
      } else {
        pg.sc_reg0 = (uint32_t)std::max(n_regs, 1);
        const size_t lds = (size_t)std::max(n_regs, 1) * 256 * sizeof(u64) + EW_MAX_SC * sizeof(ulonglong2);
        EW_LAUNCH(k_ew_program<1>, dim3((unsigned)(N / 256), limbs, batch), dim3(256), lds, c->stream, c->dev, pg);
      }
      HIPCHK(hipGetLastError());
      for (uint32_t k = 0; k < n_out; k++) outs[k] = made[k];
    }
  } catch (...) {
    for (evah_ct *t : made) if (t) evah_ct_free(c, t);
    throw;
  }
  API_END
}

} // extern "C"
