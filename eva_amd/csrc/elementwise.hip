// elementwise.hip — libeva_hip.so: elementwise evaluator calls (add / sub / negate / multiply / square / multiply_plain and their batched
// forms, weighted sums, mod_switch), the device CKKS encoder and the plaintext uploads.  One C-ABI entry point per
// seal::Evaluator call of SEALExecutor::operator() (/root/reference/eva/seal/seal_executor.h:114-175, :191-207, :217-243);
// include/eva_hip.h names the reference line each one replaces.
#include "launch.hip.h"

namespace evah {

// K1/K2: add / sub / add_plain / sub_plain.  Common polys combined, extra
// polys of the longer operand copied (or negated when it is the subtrahend).
__global__ void __launch_bounds__(256)
k_addsub(DevCtx cx, const u64 *a, size_t a_ps, uint32_t sa, const u64 *b, size_t b_ps, uint32_t sb,
         u64 *out, size_t o_ps, int sub, uint32_t smax) {
  EW_SETUP
  // grid.z = instance * smax + poly (smax = max(sa, sb)); a plaintext operand has b_ps == 0
  const uint32_t inst = p / smax, pp = p % smax;
  const size_t ao = (size_t)(inst * sa + pp) * a_ps + off, bo = (size_t)(inst * sb + pp) * b_ps + off;
  ulonglong2 r;
  if (pp < sa && pp < sb) {
    ulonglong2 x = ld2(a + ao), y = ld2(b + bo);
    r.x = sub ? submod(x.x, y.x, pm.q) : addmod(x.x, y.x, pm.q);
    r.y = sub ? submod(x.y, y.y, pm.q) : addmod(x.y, y.y, pm.q);
  } else if (pp < sa) {
    r = ld2(a + ao);
  } else {
    r = ld2(b + bo);
    if (sub) { r.x = negmod(r.x, pm.q); r.y = negmod(r.y, pm.q); }
  }
  st2(out + p * o_ps + off, r);
}

// K3: negate
__global__ void __launch_bounds__(256)
k_negate(DevCtx cx, const u64 *a, size_t a_ps, u64 *out, size_t o_ps) {
  EW_SETUP
  ulonglong2 r = ld2(a + p * a_ps + off);
  r.x = negmod(r.x, pm.q);
  r.y = negmod(r.y, pm.q);
  st2(out + p * o_ps + off, r);
}

// K4: multiply 2x2 -> 3: (a0b0, a0b1 + a1b0, a1b1); grid.z = instance of a batched handle
__global__ void __launch_bounds__(256)
k_mul22(DevCtx cx, const u64 *a, size_t a_ps, const u64 *b, size_t b_ps, u64 *out, size_t o_ps) {
  EW_SETUP
  a += (size_t)p * 2 * a_ps;
  b += (size_t)p * 2 * b_ps;
  out += (size_t)p * 3 * o_ps;
  ulonglong2 a0 = ld2(a + off), a1 = ld2(a + a_ps + off);
  ulonglong2 b0 = ld2(b + off), b1 = ld2(b + b_ps + off);
  ulonglong2 d0, d1, d2;
  d0.x = mulmod(a0.x, b0.x, pm);
  d0.y = mulmod(a0.y, b0.y, pm);
  d2.x = mulmod(a1.x, b1.x, pm);
  d2.y = mulmod(a1.y, b1.y, pm);
  u128_t t = mul128(a0.x, b1.x);
  acc128(t, a1.x, b0.x);
  d1.x = barrett128(t, pm);
  t = mul128(a0.y, b1.y);
  acc128(t, a1.y, b0.y);
  d1.y = barrett128(t, pm);
  st2(out + off, d0);
  st2(out + o_ps + off, d1);
  st2(out + 2 * o_ps + off, d2);
}

// K4 batched: n independent 2x2 products in one launch; grid.z = instance
__global__ void __launch_bounds__(256)
k_mul22_many(DevCtx cx, MulTab tab, u64 *out_b, size_t o_ps) {
  EW_SETUP
  const u64 *a = tab.a[p], *b = tab.b[p];
  const size_t a_ps = (size_t)tab.a_ps[p] * cx.N, b_ps = (size_t)tab.b_ps[p] * cx.N;
  u64 *out = out_b + (size_t)p * 3 * o_ps;
  ulonglong2 a0 = ld2(a + off), a1 = ld2(a + a_ps + off);
  ulonglong2 b0 = ld2(b + off), b1 = ld2(b + b_ps + off);
  ulonglong2 d0, d1, d2;
  d0.x = mulmod(a0.x, b0.x, pm);
  d0.y = mulmod(a0.y, b0.y, pm);
  d2.x = mulmod(a1.x, b1.x, pm);
  d2.y = mulmod(a1.y, b1.y, pm);
  u128_t t = mul128(a0.x, b1.x);
  acc128(t, a1.x, b0.x);
  d1.x = barrett128(t, pm);
  t = mul128(a0.y, b1.y);
  acc128(t, a1.y, b0.y);
  d1.y = barrett128(t, pm);
  st2(out + off, d0);
  st2(out + o_ps + off, d1);
  st2(out + 2 * o_ps + off, d2);
}

// K6b: out = sum_j ct_j (*) pt_j  (pt_j == nullptr: ct_j itself) — a convolution / linear-layer
// row as ONE pass: every input word is read once, products accumulate unreduced in 128 bits
// (n <= 64 terms of < 2^122) and are reduced once.  Same canonical result as the
// multiply_plain / add sequence it stands for.
struct WsTab {
  const u64 *ct[KS_BATCH_MAX], *pt[KS_BATCH_MAX];
  uint32_t ct_ps[KS_BATCH_MAX]; // poly strides in units of N coefficients
};
__global__ void __launch_bounds__(256)
k_weighted_sum(DevCtx cx, WsTab tab, uint32_t n, u64 *out, size_t o_ps) {
  EW_SETUP
  u128_t a0 = {0, 0}, a1 = {0, 0};
  for (uint32_t j = 0; j < n; j++) {
    const ulonglong2 x = ld2(tab.ct[j] + (size_t)p * tab.ct_ps[j] * cx.N + off);
    if (tab.pt[j]) {
      const ulonglong2 w = ld2(tab.pt[j] + off);
      acc128(a0, x.x, w.x);
      acc128(a1, x.y, w.y);
    } else {
      acc128(a0, x.x, 1);
      acc128(a1, x.y, 1);
    }
  }
  ulonglong2 r;
  r.x = barrett128(a0, pm);
  r.y = barrett128(a1, pm);
  st2(out + p * o_ps + off, r);
}

// K5: square 2 -> 3: (a0^2, 2 a0 a1, a1^2)
__global__ void __launch_bounds__(256)
k_square(DevCtx cx, const u64 *a, size_t a_ps, u64 *out, size_t o_ps) {
  EW_SETUP
  a += (size_t)p * 2 * a_ps;
  out += (size_t)p * 3 * o_ps;
  ulonglong2 a0 = ld2(a + off), a1 = ld2(a + a_ps + off);
  ulonglong2 d0, d1, d2;
  d0.x = mulmod(a0.x, a0.x, pm);
  d0.y = mulmod(a0.y, a0.y, pm);
  d2.x = mulmod(a1.x, a1.x, pm);
  d2.y = mulmod(a1.y, a1.y, pm);
  u64 x = mulmod(a0.x, a1.x, pm), y = mulmod(a0.y, a1.y, pm);
  d1.x = addmod(x, x, pm.q);
  d1.y = addmod(y, y, pm.q);
  st2(out + off, d0);
  st2(out + o_ps + off, d1);
  st2(out + 2 * o_ps + off, d2);
}

// K6: multiply_plain, every poly x pt
__global__ void __launch_bounds__(256)
k_mul_plain(DevCtx cx, const u64 *a, size_t a_ps, const u64 *pt, u64 *out, size_t o_ps) {
  EW_SETUP
  ulonglong2 x = ld2(a + p * a_ps + off), y = ld2(pt + off), r;
  r.x = mulmod(x.x, y.x, pm);
  r.y = mulmod(x.y, y.y, pm);
  st2(out + p * o_ps + off, r);
}

// K6 batched: n independent multiply_plain of one shape in one launch; grid.z = instance * size + poly
struct MpTab {
  const u64 *ct[KS_BATCH_MAX], *pt[KS_BATCH_MAX];
  uint32_t ct_ps[KS_BATCH_MAX]; // poly strides in units of N coefficients
};
__global__ void __launch_bounds__(256)
k_mul_plain_many(DevCtx cx, MpTab tab, uint32_t size, u64 *out, size_t o_ps) {
  EW_SETUP
  const uint32_t inst = p / size, poly = p - inst * size;
  ulonglong2 x = ld2(tab.ct[inst] + (size_t)poly * tab.ct_ps[inst] * cx.N + off), y = ld2(tab.pt[inst] + off), r;
  r.x = mulmod(x.x, y.x, pm);
  r.y = mulmod(x.y, y.y, pm);
  st2(out + p * o_ps + off, r);
}

// ---- CKKS encoder on the device (SEAL 3.6 CKKSEncoder::encode_internal, reached from
// seal_executor.h:242): values -> conjugate-symmetric slot vector -> inverse special FFT in FP64
// (Gentleman-Sande, one launch per stage, roots in the order the stages consume them) with the
// factor scale/N folded into the LAST stage exactly as SEAL's DWTHandler::transform_from_rev
// does (sums scaled, differences times the pre-scaled root) -> round -> residues.  Each complex
// product is four rounded multiplies, a rounded difference and a rounded sum; FMA contraction is
// off, so the doubles — and the plaintext — are those of the host encoder and the CPU oracle.
__global__ void __launch_bounds__(256)
k_enc_scatter(const double *vals, uint32_t n_vals, const uint32_t *slot_map, double2 *c, uint32_t slots) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= slots) return;
  const double v = vals[i % n_vals]; // the vector is replicated over the N/2 slots (seal_executor.h:226-240)
  c[slot_map[i]] = make_double2(v, 0.0);
  c[slot_map[slots + i]] = make_double2(v, -0.0); // conjugate of a real value
}
// stage with gap 2^log_gap: group g uses roots[root0 + g] (already conjugated)
__global__ void __launch_bounds__(256)
k_enc_fft_stage(double2 *c, const double2 *roots, uint32_t root0, uint32_t log_gap, uint32_t half_n) {
#pragma clang fp contract(off)
  const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= half_n) return;
  const uint32_t gap = 1u << log_gap, g = idx >> log_gap, j = idx & (gap - 1);
  const uint32_t a = 2 * g * gap + j, b = a + gap;
  const double2 r = roots[root0 + g];
  const double2 u = c[a], v = c[b];
  c[a] = make_double2(u.x + v.x, u.y + v.y);
  const double dx = u.x - v.x, dy = u.y - v.y;
  const double p = dx * r.x, q = dy * r.y, s = dx * r.y, t = dy * r.x;
  c[b] = make_double2(p - q, s + t);
}
// last stage (one group, gap = N/2): x = (u + v) * fix, y = (u - v) * (root * fix)
__global__ void __launch_bounds__(256)
k_enc_fft_last(double2 *c, double2 scaled_root, double fix, uint32_t half_n) {
#pragma clang fp contract(off)
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= half_n) return;
  const double2 u = c[j], v = c[j + half_n];
  c[j] = make_double2((u.x + v.x) * fix, (u.y + v.y) * fix);
  const double dx = u.x - v.x, dy = u.y - v.y;
  const double p = dx * scaled_root.x, q = dy * scaled_root.y, s = dx * scaled_root.y, t = dy * scaled_root.x;
  c[j + half_n] = make_double2(p - q, s + t);
}
__global__ void __launch_bounds__(256)
k_enc_round(DevCtx cx, const double2 *c, uint32_t limbs, u64 *out) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= cx.N) return;
  const double t = c[j].x;
  const double x = fabs(t) < 4503599627370496.0 ? round(t) : t; // >= 2^52: already an integer
  const bool neg = signbit(x);
  const u64 mant = (u64)fabs(x); // |x| < 2^63 is guaranteed by the caller's bound
  for (uint32_t i = 0; i < limbs; i++) {
    const DevPrime pm = cx.primes[cx.prime_of(i)];
    const u64 r = barrett64(mant, pm.q, pm.brt);
    out[(size_t)i * cx.N + j] = (neg && r) ? pm.q - r : r;
  }
}

// per-limb constant fill (uniform-constant plaintexts); the per-limb values travel as a
// kernel argument so the call needs no host->device copy and no synchronisation
struct LimbVals {
  u64 v[64];
};
__global__ void __launch_bounds__(256) k_fill_limbs(DevCtx cx, LimbVals vals, u64 *out) {
  EW_SETUP
  ulonglong2 r;
  r.x = r.y = vals.v[i];
  st2(out + off, r);
}

} // namespace evah

namespace evah {
// key words in the layout of the radix-2^30 inner product (ks_inner_kernel<MAC3>)
__global__ void __launch_bounds__(256) k_key_split(const u64 *__restrict__ src, u64 *__restrict__ dst, size_t words) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < words) {
    const u64 k = src[i];
    dst[i] = (k & 0x3fffffffull) | ((k >> 30) << 32);
  }
}
// evah_ct_stack: n single ciphertexts (separate allocations, possibly views with their own polynomial stride) -> the
// instances of one batched handle, ONE launch instead of a 2-D copy per instance (r6: 24 blits of ~6 us in front of every
// group of config 4).  grid = (ceil(row words / 512), size, n), 16-byte accesses.
struct StackTab {
  const u64 *p[KS_BATCH_MAX];
  uint32_t ps[KS_BATCH_MAX]; // polynomial stride of source i in units of N words
};
__global__ void __launch_bounds__(256) k_ct_stack(StackTab tab, u64 *dst, size_t dst_ps, uint32_t size, size_t row_words, uint32_t N) {
  const size_t w = 2 * ((size_t)blockIdx.x * 256 + threadIdx.x);
  if (w >= row_words) return;
  const uint32_t pl = blockIdx.y, i = blockIdx.z;
  st2(dst + ((size_t)i * size + pl) * dst_ps + w, ld2(tab.p[i] + (size_t)pl * tab.ps[i] * N + w));
}
void ct_stack_launch(evah_ctx *c, const evah_ct *const *cts, uint32_t n, evah_ct *o) {
  StackTab tab{};
  for (uint32_t i = 0; i < n; i++) {
    tab.p[i] = cts[i]->d;
    tab.ps[i] = (uint32_t)(cts[i]->ps / c->N);
  }
  const size_t row_words = (size_t)o->limbs * c->N; // (N >= 2: even)
  hipLaunchKernelGGL(k_ct_stack, dim3((unsigned)((row_words / 2 + 255) / 256), o->size, n), dim3(256), 0, c->stream, tab, o->d, o->ps,
                     o->size, row_words, (uint32_t)c->N);
  HIPCHK(hipGetLastError());
}
void key_split_launch(evah_ctx *c, const u64 *src, u64 *dst, size_t words) {
  hipLaunchKernelGGL(k_key_split, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, c->stream, src, dst, words);
  HIPCHK(hipGetLastError());
}
} // namespace evah

extern "C" {

int evah_pt_upload_coeff(evah_ctx *c, uint32_t limbs, double scale, const uint64_t *data, evah_pt **out) {
  API_BEGIN
  use(c);
  if (c->capturing) throw std::logic_error("this call synchronises with the host and cannot be captured into a graph");
  if (limbs < 1 || limbs > c->k - 1) throw std::invalid_argument("invalid limb count for this context");
  evah_pt *t = pt_new(c, limbs, scale);
  HIPCHK(hipMemcpyAsync(t->d, data, sizeof(u64) * (size_t)limbs * c->N, hipMemcpyHostToDevice, c->stream));
  OpPlain::Params p{t->d, t->d, 0, 0, limbs, 0, 0, {}};
  ntt_forward<OpPlain>(c, p, limbs);
  HIPCHK(hipStreamSynchronize(c->stream)); // the pageable host buffer may go away after return
  count_h2d(c, sizeof(u64) * (size_t)limbs * c->N, true);
  t->buf->ready_everywhere = true;
  *out = t;
  API_END
}

// encoder tables of a context family (built on first use, never inside a capture)
} // extern "C"
namespace evah {
void enc_tables(evah_ctx *c) {
  if (c->sh->enc_roots) return;
  if (c->capturing) throw std::logic_error("first use of the device encoder cannot be captured into a graph");
  const uint32_t N = c->N, slots = N >> 1, m = 2 * N;
  std::vector<uint32_t> map(N);
  u64 pos = 1;
  for (uint32_t i = 0; i < slots; i++) {
    map[i] = bitrev((uint32_t)((pos - 1) >> 1), c->logN);
    map[slots + i] = bitrev((uint32_t)((m - pos - 1) >> 1), c->logN);
    pos = (pos * 3) & (m - 1);
  }
  // inverse-transform roots in consumption order, the doubles SEAL's ComplexRoots holds (hostmath.h)
  const CkksRoots cr = ckks_roots(N);
  std::vector<double> roots(2 * (size_t)N);
  for (uint32_t j = 0; j < N; j++) {
    roots[2 * j] = cr.inv_seq[j].real();
    roots[2 * j + 1] = cr.inv_seq[j].imag();
  }
  c->sh->enc_last_root[0] = cr.inv_seq[N - 1].real();
  c->sh->enc_last_root[1] = cr.inv_seq[N - 1].imag();
  HIPCHK(hipMalloc(&c->sh->enc_slot_map, sizeof(uint32_t) * N));
  HIPCHK(hipMalloc(&c->sh->enc_roots, sizeof(double2) * N));
  h2d_now(c, c->sh->enc_slot_map, map.data(), sizeof(uint32_t) * N);
  h2d_now(c, c->sh->enc_roots, roots.data(), sizeof(double2) * N);
}
} // namespace evah
extern "C" {

// CKKSEncoder::encode of `n_values` reals replicated over the N/2 slots, at 2^scale_bits... (scale is
// passed as the double SEAL takes), to `limbs` primes, NTT form.  The caller guarantees that every
// coefficient round(x * scale / N) is below 2^62 in magnitude (the host checks a bound on
// sum |values|); larger encodings take the host's multi-precision path + evah_pt_upload_coeff.
int evah_pt_encode(evah_ctx *c, const double *values, uint32_t n_values, uint32_t limbs, double scale, evah_pt **out) {
  API_BEGIN
  use(c);
  if (c->capturing) throw std::logic_error("this call synchronises with the host and cannot be captured into a graph");
  if (limbs < 1 || limbs > c->k - 1) throw std::invalid_argument("invalid limb count for this context");
  const uint32_t N = c->N, slots = N >> 1;
  if (n_values < 1 || n_values > slots || slots % n_values) throw std::invalid_argument("value count must divide the slot count");
  enc_tables(c);
  evah_pt *t = pt_new(c, limbs, scale);
  try {
    Scratch vals(c, n_values), cbuf(c, 2 * (size_t)N); // doubles / double2 in u64-sized words
    HIPCHK(hipMemcpyAsync(vals.d, values, sizeof(double) * n_values, hipMemcpyHostToDevice, c->stream));
    double2 *cd = reinterpret_cast<double2 *>(cbuf.d);
    ProfScope ps(c, KC_EW);
    hipLaunchKernelGGL(k_enc_scatter, dim3((slots + 255) / 256), dim3(256), 0, c->stream, reinterpret_cast<const double *>(vals.d),
                       n_values, c->sh->enc_slot_map, cd, slots);
    const double fix = scale / (double)N;
    for (uint32_t mm = N >> 1, lg = 0; mm > 1; mm >>= 1, lg++) // stage with mm groups starts at root N - 2mm + 1
      hipLaunchKernelGGL(k_enc_fft_stage, dim3((slots + 255) / 256), dim3(256), 0, c->stream, cd, c->sh->enc_roots, N - 2 * mm + 1,
                         lg, slots);
    hipLaunchKernelGGL(k_enc_fft_last, dim3((slots + 255) / 256), dim3(256), 0, c->stream, cd,
                       make_double2(c->sh->enc_last_root[0] * fix, c->sh->enc_last_root[1] * fix), fix, slots);
    hipLaunchKernelGGL(k_enc_round, dim3((N + 255) / 256), dim3(256), 0, c->stream, c->dev, cd, limbs, t->d);
    HIPCHK(hipGetLastError());
    OpPlain::Params p{t->d, t->d, 0, 0, limbs, 0, 0, {}};
    ntt_forward<OpPlain>(c, p, limbs);
    HIPCHK(hipStreamSynchronize(c->stream)); // `values` is pageable host memory
  } catch (...) {
    evah_pt_free(c, t);
    throw;
  }
  t->buf->ready_everywhere = true;
  *out = t;
  API_END
}

int evah_pt_uniform(evah_ctx *c, uint32_t limbs, double scale, const uint64_t *value, evah_pt **out) {
  API_BEGIN
  use(c);
  if (limbs < 1 || limbs > c->k - 1) throw std::invalid_argument("invalid limb count for this context");
  if (limbs > 64) throw std::invalid_argument("too many limbs");
  evah_pt *t = pt_new(c, limbs, scale);
  LimbVals lv;
  for (uint32_t i = 0; i < limbs; i++) lv.v[i] = value[i];
  EW_LAUNCH(k_fill_limbs, ew_grid(c, limbs, 1), dim3(256), 0, c->stream, c->dev, lv, t->d);
  HIPCHK(hipGetLastError());
  t->uniform = true;
  *out = t;
  API_END
}

// ---- evaluator

static int addsub_impl(evah_ctx *c, const evah_ct *a, const evah_ct *b, evah_ct **out, int sub) {
  API_BEGIN
  use(c);
  acquire(c, a->buf);
  acquire(c, b->buf);
  if (a->limbs != b->limbs) throw std::invalid_argument("encrypted1 and encrypted2 parameter mismatch");
  if (!same_scale(a->scale, b->scale)) throw std::invalid_argument("scale mismatch");
  if (a->batch != b->batch) throw std::invalid_argument("batch size mismatch");
  const uint32_t s = std::max(a->size, b->size);
  evah_ct *o = ct_new(c, s, a->limbs, a->scale, a->batch);
  EW_LAUNCH(k_addsub, ew_grid(c, a->limbs, s * a->batch), dim3(256), 0, c->stream, c->dev, a->d, a->ps, a->size,
                     b->d, b->ps, b->size, o->d, o->ps, sub, s);
  HIPCHK(hipGetLastError());
  *out = o;
  API_END
}
int evah_add(evah_ctx *c, const evah_ct *a, const evah_ct *b, evah_ct **out) { return addsub_impl(c, a, b, out, 0); }
int evah_sub(evah_ctx *c, const evah_ct *a, const evah_ct *b, evah_ct **out) { return addsub_impl(c, a, b, out, 1); }

static int addsub_plain_impl(evah_ctx *c, const evah_ct *a, const evah_pt *b, evah_ct **out, int sub) {
  API_BEGIN
  use(c);
  acquire(c, a->buf);
  acquire(c, b->buf);
  if (a->limbs != b->limbs) throw std::invalid_argument("encrypted and plain parameter mismatch");
  if (!same_scale(a->scale, b->scale)) throw std::invalid_argument("scale mismatch");
  evah_ct *o = ct_new(c, a->size, a->limbs, a->scale, a->batch);
  EW_LAUNCH(k_addsub, ew_grid(c, a->limbs, a->size * a->batch), dim3(256), 0, c->stream, c->dev, a->d, a->ps,
                     a->size, b->d, (size_t)0, 1u, o->d, o->ps, sub, a->size);
  HIPCHK(hipGetLastError());
  *out = o;
  API_END
}
int evah_add_plain(evah_ctx *c, const evah_ct *a, const evah_pt *b, evah_ct **out) { return addsub_plain_impl(c, a, b, out, 0); }
int evah_sub_plain(evah_ctx *c, const evah_ct *a, const evah_pt *b, evah_ct **out) { return addsub_plain_impl(c, a, b, out, 1); }

int evah_negate(evah_ctx *c, const evah_ct *a, evah_ct **out) {
  API_BEGIN
  use(c);
  acquire(c, a->buf);
  evah_ct *o = ct_new(c, a->size, a->limbs, a->scale, a->batch);
  EW_LAUNCH(k_negate, ew_grid(c, a->limbs, a->size * a->batch), dim3(256), 0, c->stream, c->dev, a->d, a->ps, o->d, o->ps);
  HIPCHK(hipGetLastError());
  *out = o;
  API_END
}

int evah_multiply(evah_ctx *c, const evah_ct *a, const evah_ct *b, evah_ct **out) {
  API_BEGIN
  use(c);
  acquire(c, a->buf);
  acquire(c, b->buf);
  if (a->limbs != b->limbs) throw std::invalid_argument("encrypted1 and encrypted2 parameter mismatch");
  if (a->size != 2 || b->size != 2) throw std::invalid_argument("multiply supports size-2 operands only (relinearize first)");
  const double ns = a->scale * b->scale;
  check_scale(c, ns, a->limbs);
  if (a->batch != b->batch) throw std::invalid_argument("batch size mismatch");
  evah_ct *o = ct_new(c, 3, a->limbs, ns, a->batch);
  EW_LAUNCH(k_mul22, ew_grid(c, a->limbs, a->batch), dim3(256), 0, c->stream, c->dev, a->d, a->ps, b->d, b->ps, o->d, o->ps);
  HIPCHK(hipGetLastError());
  *out = o;
  API_END
}

// n (<= 64) independent products at one level as ONE launch (same ciphertexts as n evah_multiply
// calls); the outputs are views into one allocation.
int evah_multiply_many(evah_ctx *c, const evah_ct *const *as, const evah_ct *const *bs, uint32_t n, evah_ct **outs) {
  API_BEGIN
  use(c);
  if (n < 1 || n > (uint32_t)KS_BATCH_MAX) throw std::invalid_argument("multiply_many handles 1..64 products per call");
  const uint32_t l = as[0]->limbs;
  const size_t N = c->N, ops = (size_t)l * N;
  MulTab tab{};
  std::vector<double> scales(n);
  for (uint32_t i = 0; i < n; i++) {
    const evah_ct *a = as[i], *b = bs[i];
    if (a->size != 2 || b->size != 2) throw std::invalid_argument("multiply supports size-2 operands only (relinearize first)");
    if (a->batch != 1 || b->batch != 1) throw std::invalid_argument("multiply_many takes single ciphertexts (a batched handle already multiplies in one launch)");
    if (a->limbs != l || b->limbs != l) throw std::invalid_argument("encrypted parameter mismatch in batch");
    scales[i] = a->scale * b->scale;
    check_scale(c, scales[i], l);
    acquire(c, a->buf);
    acquire(c, b->buf);
    tab.a[i] = a->d;
    tab.b[i] = b->d;
    tab.a_ps[i] = (uint32_t)(a->ps / N);
    tab.b_ps[i] = (uint32_t)(b->ps / N);
  }
  Buffer *ob = buf_new(c, (size_t)n * 3 * ops);
  EW_LAUNCH(k_mul22_many, ew_grid(c, l, n), dim3(256), 0, c->stream, c->dev, tab, ob->d, ops);
  HIPCHK(hipGetLastError());
  ob->refs = (int)n;
  for (uint32_t i = 0; i < n; i++) {
    evah_ct *t = new evah_ct;
    t->buf = ob;
    t->d = ob->d + (size_t)i * 3 * ops;
    t->size = 3;
    t->limbs = l;
    t->ps = ops;
    t->scale = scales[i];
    outs[i] = t;
  }
  API_END
}

int evah_square(evah_ctx *c, const evah_ct *a, evah_ct **out) {
  API_BEGIN
  use(c);
  acquire(c, a->buf);
  if (a->size != 2) throw std::invalid_argument("square supports size-2 operands only (relinearize first)");
  const double ns = a->scale * a->scale;
  check_scale(c, ns, a->limbs);
  evah_ct *o = ct_new(c, 3, a->limbs, ns, a->batch);
  EW_LAUNCH(k_square, ew_grid(c, a->limbs, a->batch), dim3(256), 0, c->stream, c->dev, a->d, a->ps, o->d, o->ps);
  HIPCHK(hipGetLastError());
  *out = o;
  API_END
}

int evah_multiply_plain(evah_ctx *c, const evah_ct *a, const evah_pt *b, evah_ct **out) {
  API_BEGIN
  use(c);
  acquire(c, a->buf);
  acquire(c, b->buf);
  if (a->limbs != b->limbs) throw std::invalid_argument("encrypted and plain parameter mismatch");
  const double ns = a->scale * b->scale;
  check_scale(c, ns, a->limbs);
  evah_ct *o = ct_new(c, a->size, a->limbs, ns, a->batch);
  EW_LAUNCH(k_mul_plain, ew_grid(c, a->limbs, a->size * a->batch), dim3(256), 0, c->stream, c->dev, a->d, a->ps, b->d, o->d, o->ps);
  HIPCHK(hipGetLastError());
  *out = o;
  API_END
}

// n (<= 64) independent multiply_plain calls of one shape (size, limbs) as one launch
int evah_multiply_plain_many(evah_ctx *c, const evah_ct *const *cts, const evah_pt *const *pts, uint32_t n, evah_ct **outs) {
  API_BEGIN
  use(c);
  if (n < 1 || n > (uint32_t)KS_BATCH_MAX) throw std::invalid_argument("multiply_plain_many handles 1..64 products per call");
  const uint32_t size = cts[0]->size, l = cts[0]->limbs;
  const size_t N = c->N, ops = (size_t)l * N;
  MpTab tab{};
  std::vector<double> scales(n);
  for (uint32_t i = 0; i < n; i++) {
    const evah_ct *a = cts[i];
    const evah_pt *b = pts[i];
    if (a->batch != 1) throw std::invalid_argument("multiply_plain_many takes single ciphertexts");
    if (a->size != size || a->limbs != l) throw std::invalid_argument("encrypted parameter mismatch in batch");
    if (b->limbs != l) throw std::invalid_argument("encrypted and plain parameter mismatch");
    scales[i] = a->scale * b->scale;
    check_scale(c, scales[i], l);
    acquire(c, a->buf);
    acquire(c, b->buf);
    tab.ct[i] = a->d;
    tab.pt[i] = b->d;
    tab.ct_ps[i] = (uint32_t)(a->ps / N);
  }
  Buffer *ob = buf_new(c, (size_t)n * size * ops);
  EW_LAUNCH(k_mul_plain_many, ew_grid(c, l, n * size), dim3(256), 0, c->stream, c->dev, tab, size, ob->d, ops);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    buf_unref(c, ob);
    HIPCHK(e);
  }
  ob->refs = (int)n;
  for (uint32_t i = 0; i < n; i++) {
    evah_ct *t = new evah_ct;
    t->buf = ob;
    t->d = ob->d + (size_t)i * size * ops;
    t->size = size;
    t->limbs = l;
    t->ps = ops;
    t->scale = scales[i];
    outs[i] = t;
  }
  API_END
}

// sum_j cts[j] (*) pts[j], pts[j] == NULL standing for the ciphertext itself: the value of
// add(... add(multiply_plain(cts[0], pts[0]), multiply_plain(cts[1], pts[1])) ...) in one pass
int evah_weighted_sum(evah_ctx *c, const evah_ct *const *cts, const evah_pt *const *pts, uint32_t n, evah_ct **out) {
  API_BEGIN
  use(c);
  if (n < 1 || n > (uint32_t)KS_BATCH_MAX) throw std::invalid_argument("weighted_sum handles 1..64 terms per call");
  const evah_ct *f = cts[0];
  const double scale = f->scale * (pts[0] ? pts[0]->scale : 1.0);
  WsTab tab{};
  for (uint32_t j = 0; j < n; j++) {
    const evah_ct *a = cts[j];
    if (a->limbs != f->limbs) throw std::invalid_argument("encrypted1 and encrypted2 parameter mismatch");
    if (a->size != f->size) throw std::invalid_argument("weighted_sum terms must have one size");
    if (a->batch != f->batch) throw std::invalid_argument("batch size mismatch");
    if (pts[j] && pts[j]->limbs != a->limbs) throw std::invalid_argument("encrypted and plain parameter mismatch");
    const double sj = a->scale * (pts[j] ? pts[j]->scale : 1.0);
    if (pts[j]) check_scale(c, sj, a->limbs);
    if (!same_scale(sj, scale)) throw std::invalid_argument("scale mismatch");
    acquire(c, a->buf);
    if (pts[j]) acquire(c, pts[j]->buf);
    tab.ct[j] = a->d;
    tab.pt[j] = pts[j] ? pts[j]->d : nullptr;
    tab.ct_ps[j] = (uint32_t)(a->ps / c->N);
  }
  evah_ct *o = ct_new(c, f->size, f->limbs, scale, f->batch);
  EW_LAUNCH(k_weighted_sum, ew_grid(c, f->limbs, f->size * f->batch), dim3(256), 0, c->stream, c->dev, tab, n, o->d, o->ps);
  HIPCHK(hipGetLastError());
  *out = o;
  API_END
}

int evah_mod_switch(evah_ctx *c, const evah_ct *a, evah_ct **out) {
  API_BEGIN
  use(c);
  if (a->limbs < 2) throw std::invalid_argument("end of modulus switching chain reached");
  check_scale(c, a->scale, a->limbs - 1);
  // dropping the last limb is a view: same buffer, same poly stride, one limb fewer
  evah_ct *o = new evah_ct(*a);
  o->limbs = a->limbs - 1;
  o->buf->refs++;
  *out = o;
  API_END
}

int evah_test_ntt(evah_ctx *c, uint32_t prime_idx, int inverse, uint64_t *host) {
  API_BEGIN
  use(c);
  if (c->capturing) throw std::logic_error("this call synchronises with the host and cannot be captured into a graph");
  if (prime_idx >= c->k) throw std::invalid_argument("prime index out of range");
  Scratch s(c, c->N);
  HIPCHK(hipMemcpyAsync(s.d, host, sizeof(u64) * c->N, hipMemcpyHostToDevice, c->stream));
  OpPlain::Params p{s.d, s.d, 0, 0, 1, prime_idx, 0, {}};
  if (inverse) ntt_inverse<OpPlain>(c, p, 1);
  else ntt_forward<OpPlain>(c, p, 1);
  HIPCHK(hipMemcpyAsync(host, s.d, sizeof(u64) * c->N, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  API_END
}

} // extern "C"
