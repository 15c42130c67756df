// windows.hip — libeva_hip.so: evah_rotate_weighted_sums, the convolution window of EVA programs
// (/root/reference/examples/image_processing.py:22-34: rotate_vector, multiply_plain and add of every tap;
// seal_executor.h:181/188, :168, :124) as one launch set whose rotated ciphertexts are never written.
#include "rotation_sets.hip.h"
#include "ntt_window_lin.hip.h"

namespace evah {
// the guarded fallback's sums: the same outputs from rotated ciphertexts rot[pair][2][l N] (rot_chunk_plain)
template <int F>
__global__ void __launch_bounds__(256)
k_window_sums(DevCtx cx, WinSumTab ws, const u64 *rot, size_t rot_ps, size_t out_ps) {
  if (cx.skipped()) return;
  const uint32_t i = blockIdx.y, w = blockIdx.z >> 1, K = blockIdx.z & 1u;
  const size_t off = (size_t)i * cx.N + 2 * ((size_t)blockIdx.x * blockDim.x + threadIdx.x);
  const DevPrime pm = cx.primes[cx.prime_of(i)];
  u128_t acc[F][2];
#pragma unroll
  for (int f = 0; f < F; f++) acc[f][0] = acc[f][1] = {0, 0};
  auto mac = [&](int f, const ulonglong2 &v, const u64 *wt) {
    ulonglong2 x;
    x.x = x.y = 1;
    if (wt) x = ld2(wt + off);
    acc128(acc[f][0], v.x, x.x);
    acc128(acc[f][1], v.y, x.y);
  };
  const uint32_t first = ws.first[w], cnt = ws.count[w];
  for (uint32_t t = first; t < first + cnt; t++) {
    const ulonglong2 v = ld2(rot + (size_t)(2 * t + K) * rot_ps + off);
    mac(0, v, ws.w0[t]);
    if constexpr (F > 1) mac(1, v, ws.w1[t]);
  }
  if (ws.id_src[w]) {
    const ulonglong2 v = ld2(ws.id_src[w] + (size_t)K * ws.id_ps[w] * cx.N + off);
    mac(0, v, ws.id_w0[w]);
    if constexpr (F > 1) mac(1, v, ws.id_w1[w]);
  }
#pragma unroll
  for (int f = 0; f < F; f++) {
    ulonglong2 r;
    r.x = barrett128(acc[f][0], pm);
    r.y = barrett128(acc[f][1], pm);
    st2((f ? ws.out1[w] : ws.out0[w]) + (size_t)K * out_ps + off, r);
  }
}
template <int P>
static void launch_moddown_sum(evah_ctx *c, uint32_t l, uint32_t n_win, int F, const WinSumTab &wt, const PermTab &perms, const u64 *mid, size_t mid_ps,
                               const u64 *prod, size_t prod_ps, size_t out_ps) {
  ProfScope ps(c, KC_MODDOWN_B);
  const int logC = 8 - P;
  const size_t lds = ((((size_t)1 << logC) * lds_sub_stride<P>() + 1) & ~(size_t)1) * sizeof(u64) + ((size_t)1 << (logC + P)) * sizeof(ulonglong2);
  const dim3 grid(c->N / 256, l, 2 * n_win);
  if (F == 1) hipLaunchKernelGGL((moddown_sum_kernel<P, 1>), grid, dim3(64), lds, c->stream, c->dev, wt, perms, mid, mid_ps, prod, prod_ps, out_ps, logC);
  else hipLaunchKernelGGL((moddown_sum_kernel<P, 2>), grid, dim3(64), lds, c->stream, c->dev, wt, perms, mid, mid_ps, prod, prod_ps, out_ps, logC);
  HIPCHK(hipGetLastError());
}
} // namespace evah
extern "C" {

// out[s] = sum_t pts[s][t] (*) rotate(cts[t], steps[t]) for the sums s of every window (include/eva_hip.h).
int evah_rotate_weighted_sums(evah_ctx *c, const evah_ct *const *cts, const int32_t *steps, const uint32_t *win_terms, const uint32_t *win_sums,
                              uint32_t n_windows, const evah_pt *const *pts, evah_ct **outs) {
  API_BEGIN
  use(c);
  if (n_windows < 1) throw std::invalid_argument("rotate_weighted_sums needs at least one window");
  uint32_t n_terms = 0, n_sums = 0;
  for (uint32_t w = 0; w < n_windows; w++) {
    if (win_terms[w] < 1 || win_terms[w] > (uint32_t)KS_BATCH_MAX) throw std::invalid_argument("a window has 1..64 terms");
    if (win_sums[w] < 1) throw std::invalid_argument("a window has at least one sum");
    n_terms += win_terms[w];
    n_sums += win_sums[w];
  }
  const uint32_t l = cts[0]->limbs, B = cts[0]->batch;
  const size_t N = c->N, pps = (size_t)l * N, prod_bs = (size_t)2 * (l + 1) * N;
  for (uint32_t t = 0; t < n_terms; t++) {
    const evah_ct *a = cts[t];
    if (a->size != 2) throw std::invalid_argument("rotate expects a size-2 ciphertext (relinearize first)");
    if (a->limbs != l) throw std::invalid_argument("encrypted parameter mismatch in batch");
    if (a->batch != B) throw std::invalid_argument("batch size mismatch");
    acquire(c, a->buf);
  }
  // scales as evah_weighted_sum checks them; shape of every window
  std::vector<double> scales(n_sums);
  bool fusable = c->tun.win_fuse && c->tun.fold_pa;
  // r6: every weight of a rotated term is a uniform plaintext (or 1): the window's mod-down runs ONE forward transform per
  // (sum, polynomial, limb) — ntt_window_lin.hip.h; EVAH_WIN_LINEAR=0 keeps one per rotation
  std::vector<char> win_lin(n_windows, c->tun.win_linear ? 1 : 0);
  uint32_t n_rot = 0;
  std::vector<const evah_ct *> distinct; // sources of rotated terms
  std::vector<uint32_t> src_of(n_terms, 0);
  {
    uint32_t t0 = 0, p0 = 0, s0 = 0;
    for (uint32_t w = 0; w < n_windows; w++) {
      const uint32_t nt = win_terms[w], ns = win_sums[w];
      for (uint32_t s = 0; s < ns; s++)
        for (uint32_t j = 0; j < nt; j++) {
          const evah_pt *pt = pts[p0 + s * nt + j];
          if (pt && pt->limbs != l) throw std::invalid_argument("encrypted and plain parameter mismatch");
          const double sj = cts[t0 + j]->scale * (pt ? pt->scale : 1.0);
          if (pt) {
            check_scale(c, sj, l);
            acquire(c, pt->buf);
            if (steps[t0 + j] != 0 && !pt->uniform) win_lin[w] = 0;
          }
          if (j == 0) scales[s0 + s] = sj;
          else if (!same_scale(sj, scales[s0 + s])) throw std::invalid_argument("scale mismatch");
        }
      uint32_t ids = 0;
      for (uint32_t j = 0; j < nt; j++) {
        if (steps[t0 + j] == 0) { ids++; continue; }
        const evah_ct *a = cts[t0 + j];
        uint32_t si = 0;
        while (si < distinct.size() && !(distinct[si]->d == a->d && distinct[si]->ps == a->ps)) si++;
        if (si == distinct.size()) distinct.push_back(a);
        src_of[t0 + j] = si;
      }
      if (ns > 2 || ids > 1 || ids == nt) fusable = false;
      n_rot += nt - ids;
      t0 += nt;
      p0 += nt * ns;
      s0 += ns;
    }
  }
  fusable = fusable && distinct.size() * B <= (size_t)KS_BATCH_MAX && hoist_wanted(c, l, n_rot, B);
  // the hoisting tables of every rotated term; when one does not fit the device the window takes the general form
  std::vector<RotPair> term_pair(n_terms);
  for (uint32_t t = 0; t < n_terms && fusable; t++) {
    if (steps[t] == 0) continue;
    term_pair[t] = rot_pair(c, cts[t]->d, cts[t]->ps, 0, steps[t], l, "rotate_weighted_sums");
    fusable = hoist_prepare(c, term_pair[t], l);
  }
  auto chk = [&](int rc) {
    if (rc) throw std::runtime_error(g_err);
  };
  std::vector<evah_ct *> made; // outputs created so far (released if a later step throws)
  struct Temps {
    evah_ctx *c;
    std::vector<evah_ct *> v;
    ~Temps() { for (evah_ct *t : v) if (t) evah_ct_free(c, t); }
  } rotated{c, std::vector<evah_ct *>(n_terms, nullptr)};
  try {
    if (!fusable) {
      // the general form: the rotations as launch sets (rotate_pairs / rotate_many), then one weighted sum per sum
      if (B == 1) {
        std::vector<uint32_t> idx;
        for (uint32_t t = 0; t < n_terms; t++) if (steps[t] != 0) idx.push_back(t);
        for (size_t i0 = 0; i0 < idx.size(); i0 += KS_BATCH_MAX) {
          const uint32_t n = (uint32_t)std::min<size_t>(KS_BATCH_MAX, idx.size() - i0);
          std::vector<const evah_ct *> in(n);
          std::vector<int32_t> st(n);
          std::vector<evah_ct *> out(n, nullptr);
          for (uint32_t j = 0; j < n; j++) { in[j] = cts[idx[i0 + j]]; st[j] = steps[idx[i0 + j]]; }
          chk(evah_rotate_pairs(c, in.data(), st.data(), n, out.data()));
          for (uint32_t j = 0; j < n; j++) rotated.v[idx[i0 + j]] = out[j];
        }
      } else {
        for (uint32_t si = 0; si < distinct.size(); si++) {
          std::vector<uint32_t> idx;
          for (uint32_t t = 0; t < n_terms; t++) if (steps[t] != 0 && src_of[t] == si) idx.push_back(t);
          for (size_t i0 = 0; i0 < idx.size(); i0 += KS_BATCH_MAX) {
            const uint32_t n = (uint32_t)std::min<size_t>(KS_BATCH_MAX, idx.size() - i0);
            std::vector<int32_t> st(n);
            std::vector<evah_ct *> out(n, nullptr);
            for (uint32_t j = 0; j < n; j++) st[j] = steps[idx[i0 + j]];
            chk(evah_rotate_many(c, distinct[si], st.data(), n, out.data()));
            for (uint32_t j = 0; j < n; j++) rotated.v[idx[i0 + j]] = out[j];
          }
        }
      }
      uint32_t t0 = 0, p0 = 0;
      for (uint32_t w = 0; w < n_windows; w++) {
        const uint32_t nt = win_terms[w], ns = win_sums[w];
        std::vector<const evah_ct *> cc(nt);
        for (uint32_t j = 0; j < nt; j++) cc[j] = steps[t0 + j] ? rotated.v[t0 + j] : cts[t0 + j];
        for (uint32_t s = 0; s < ns; s++) {
          evah_ct *o = nullptr;
          chk(evah_weighted_sum(c, cc.data(), pts + p0 + s * nt, nt, &o));
          made.push_back(o);
        }
        t0 += nt;
        p0 += nt * ns;
      }
    } else {
      for (uint32_t s = 0; s < n_sums; s++) made.push_back(ct_new(c, 2, l, scales[s], B));
      const size_t out_ps = made[0]->ps;
      // (window, instance) units: the rotated terms of a window for one instance of the batch
      struct Unit { uint32_t first, count, w, b, t0, p0, s0; };
      std::vector<RotPair> pairs;
      std::vector<uint32_t> pair_term; // pair -> its term (position among cts / steps)
      std::vector<Unit> units;
      {
        uint32_t t0 = 0, p0 = 0, s0 = 0;
        for (uint32_t w = 0; w < n_windows; w++) {
          const uint32_t nt = win_terms[w], ns = win_sums[w];
          for (uint32_t b = 0; b < B; b++) {
            Unit u{(uint32_t)pairs.size(), 0, w, b, t0, p0, s0};
            for (uint32_t j = 0; j < nt; j++) {
              if (steps[t0 + j] == 0) continue;
              RotPair p = term_pair[t0 + j];
              p.src = cts[t0 + j]->d + (size_t)b * 2 * cts[t0 + j]->ps;
              p.src_idx = src_of[t0 + j] * B + b;
              pairs.push_back(p);
              pair_term.push_back(t0 + j);
              u.count++;
            }
            units.push_back(u);
          }
          t0 += nt;
          p0 += nt * ns;
          s0 += ns;
        }
      }
      std::vector<const u64 *> srcs(distinct.size() * B);
      std::vector<size_t> src_ps(distinct.size() * B);
      for (uint32_t si = 0; si < distinct.size(); si++)
        for (uint32_t b = 0; b < B; b++) {
          srcs[si * B + b] = distinct[si]->d + (size_t)b * 2 * distinct[si]->ps;
          src_ps[si * B + b] = distinct[si]->ps;
        }
      // chunks: whole units, at most KS_BATCH_MAX pairs and WIN_MAX units, one number of sums per launch
      struct Chunk { uint32_t u0, nu, first, np; int F; bool lin; WinSumTab wt; };
      std::vector<Chunk> chunks;
      for (uint32_t u = 0; u < units.size(); u++) {
        const int F = (int)win_sums[units[u].w];
        const bool lin = win_lin[units[u].w] != 0;
        if (chunks.empty() || chunks.back().F != F || chunks.back().lin != lin || chunks.back().nu == (uint32_t)WIN_MAX ||
            chunks.back().np + units[u].count > (uint32_t)KS_BATCH_MAX)
          chunks.push_back(Chunk{u, 0, units[u].first, 0, F, lin, WinSumTab{}});
        Chunk &ch = chunks.back();
        WinSumTab &wt = ch.wt;
        const Unit &un = units[u];
        const uint32_t wi = ch.nu++, nt = win_terms[un.w];
        wt.first[wi] = (uint8_t)(un.first - ch.first);
        wt.count[wi] = (uint8_t)un.count;
        for (uint32_t q = 0; q < un.count; q++) {
          const uint32_t j = pair_term[un.first + q] - un.t0;
          const evah_pt *a0 = pts[un.p0 + j], *a1 = F > 1 ? pts[un.p0 + nt + j] : nullptr;
          wt.w0[un.first - ch.first + q] = a0 ? a0->d : nullptr;
          wt.w1[un.first - ch.first + q] = a1 ? a1->d : nullptr;
        }
        for (uint32_t j = 0; j < nt; j++) {
          if (steps[un.t0 + j] != 0) continue;
          const evah_ct *a = cts[un.t0 + j];
          const evah_pt *a0 = pts[un.p0 + j], *a1 = F > 1 ? pts[un.p0 + nt + j] : nullptr;
          wt.id_src[wi] = a->d + (size_t)un.b * 2 * a->ps;
          wt.id_ps[wi] = (uint32_t)(a->ps / N);
          wt.id_w0[wi] = a0 ? a0->d : nullptr;
          wt.id_w1[wi] = a1 ? a1->d : nullptr;
          // a weight of 1 on the unrotated term is distinguished from "no such term" by id_src
        }
        wt.out0[wi] = made[un.s0]->d + (size_t)un.b * 2 * out_ps;
        wt.out1[wi] = F > 1 ? made[un.s0 + 1]->d + (size_t)un.b * 2 * out_ps : nullptr;
        ch.np += un.count;
      }
      ZeroFlag flag(c, chunks.size());
      const size_t dg_bs = (size_t)(l + 1) * l * N;
      {
        Scratch t(c, srcs.size() * l * N), dg(c, srcs.size() * dg_bs);
        hoist_digits(c, l, srcs, src_ps, flag, t.d, dg.d);
        for (const Chunk &ch : chunks) {
          const RotPair *pr = pairs.data() + ch.first;
          const uint32_t np = ch.np;
          HoistMacTab mt{};
          HoistFixTab ft{};
          const HoistTiles tiles = hoist_tables(pr, np, N, mt, ft);
          Scratch prod(c, np * prod_bs), r(c, (size_t)np * 2 * N), mid(c, (ch.lin ? (size_t)ch.nu * ch.F : (size_t)np) * 2 * pps);
          hoist_mac_launch(c, mt, tiles, dg.d, dg_bs, prod.d, prod_bs, l, true);
          {
            ProfScope ps(c, KC_KSMAC);
            hipLaunchKernelGGL(k_hoist_fix, dim3(c->N / 256, l + 1, np), dim3(256), 0, c->stream, c->dev, flag.d, ft, prod.d, prod_bs, l);
            HIPCHK(hipGetLastError());
          }
          // mod-down: INTT of the special rows (read through the pairs' permutations), first (strided) pass of the
          // forward transforms into mid, then the second pass with the window's sums as its epilogue
          OpPlainG::Params sp{prod.d + (size_t)l * N, r.d, (size_t)(l + 1) * N, N, 1, c->k - 1, 1, {}};
          for (uint32_t q = 0; q < np; q++) sp.perm_tab.p[q] = pr[q].perm;
          if (ch.lin) {
            ntt_inverse<OpPlainG>(c, sp, 2 * np);
            OpWinLin::Params wp{r.d, mid.d, c->k - 1, l, (uint32_t)ch.F, ch.wt};
            launch_pass_p<true, false, OpWinLin>(c, (c->logN + 1) / 2, wp, l * ch.nu * ch.F * 2);
            switch (c->logN / 2) {
            case 5: launch_winlin_pass2<5>(c, l, ch.nu, ch.F, ch.wt, sp.perm_tab, mid.d, pps, prod.d, (size_t)(l + 1) * N, out_ps); break;
            case 6: launch_winlin_pass2<6>(c, l, ch.nu, ch.F, ch.wt, sp.perm_tab, mid.d, pps, prod.d, (size_t)(l + 1) * N, out_ps); break;
            case 7: launch_winlin_pass2<7>(c, l, ch.nu, ch.F, ch.wt, sp.perm_tab, mid.d, pps, prod.d, (size_t)(l + 1) * N, out_ps); break;
            case 8: launch_winlin_pass2<8>(c, l, ch.nu, ch.F, ch.wt, sp.perm_tab, mid.d, pps, prod.d, (size_t)(l + 1) * N, out_ps); break;
            default: throw std::runtime_error("unsupported poly_modulus_degree for the window sums");
            }
            continue;
          }
          OpModDown::Params mp{r.d, N, prod.d, (size_t)(l + 1) * N, nullptr, pps, ~0u, mid.d, pps, c->k - 1, l};
          if (fuse_small_launch(c, 2 * np * l)) {
            launch_pass_p<false, true, OpPlainG>(c, c->logN / 2, sp, 2 * np);
            launch_inv_fwd<OpModDown>(c, mp, 2 * np * l);
          } else {
            ntt_inverse<OpPlainG>(c, sp, 2 * np);
            launch_pass_p<true, false, OpModDown>(c, (c->logN + 1) / 2, mp, 2 * np * l);
          }
          switch (c->logN / 2) {
          case 5: launch_moddown_sum<5>(c, l, ch.nu, ch.F, ch.wt, sp.perm_tab, mid.d, pps, prod.d, (size_t)(l + 1) * N, out_ps); break;
          case 6: launch_moddown_sum<6>(c, l, ch.nu, ch.F, ch.wt, sp.perm_tab, mid.d, pps, prod.d, (size_t)(l + 1) * N, out_ps); break;
          case 7: launch_moddown_sum<7>(c, l, ch.nu, ch.F, ch.wt, sp.perm_tab, mid.d, pps, prod.d, (size_t)(l + 1) * N, out_ps); break;
          case 8: launch_moddown_sum<8>(c, l, ch.nu, ch.F, ch.wt, sp.perm_tab, mid.d, pps, prod.d, (size_t)(l + 1) * N, out_ps); break;
          default: throw std::runtime_error("unsupported poly_modulus_degree for the window sums");
          }
        }
      }
      // exact fallback (more zero digit coefficients than k_hoist_fix handles): the unhoisted rotations, then the sums
      GuardScope gs(c, reinterpret_cast<const uint32_t *>(flag.d));
      for (size_t ci = 0; ci < chunks.size(); ci++) {
        const Chunk &ch = chunks[ci];
        if (c->tun.fb_persist) {
          rot_fallback_launch(c, l, pairs.data() + ch.first, ch.np, nullptr, &ch.wt, ch.nu, ch.F, out_ps, flag.bar(ci));
          continue;
        }
        Scratch rot(c, (size_t)ch.np * 2 * pps);
        rot_chunk_plain(c, l, pairs.data() + ch.first, ch.np, rot.d);
        const dim3 grid(c->N / 512, l, 2 * ch.nu);
        if (ch.F == 1) EW_LAUNCH((k_window_sums<1>), grid, dim3(256), 0, c->stream, c->dev, ch.wt, rot.d, pps, out_ps);
        else EW_LAUNCH((k_window_sums<2>), grid, dim3(256), 0, c->stream, c->dev, ch.wt, rot.d, pps, out_ps);
        HIPCHK(hipGetLastError());
      }
    }
  } catch (...) {
    for (evah_ct *t : made) evah_ct_free(c, t);
    throw;
  }
  for (uint32_t s = 0; s < n_sums; s++) outs[s] = made[s];
  API_END
}
} // extern "C"
