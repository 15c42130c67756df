// microbench_bfly.hip — issue cost of the lazy NTT butterflies of eva_amd/csrc/ntt.hip.h on gfx950,
// in the variants round 3 weighed (profiles/r03_tuning_notes.md):
//   fwd  base      conditional subtraction as compare + select + 64-bit add (every other stage)
//   fwd  bit       the same reduction read off the top bit: x = (X mod 2^s) + (X >> s) * (2^s mod q)
//                  with 2^s ~ 8q (s = 63 for SEAL's 60-bit primes): shift, and, one v_mad_u64_u32
//   fwd  madlo     the low-word cross products accumulated through v_mad_u64_u32 (inline asm) instead
//                  of v_mul_lo_u32 + v_add3_u32
//   inv  base/bit  Gentleman-Sande butterfly, reduction of the sum likewise (2^s ~ 4q)
// Every variant computes the same residues mod q; main() checks that before timing.
// Build: hipcc --offload-arch=gfx950 -O3 -o scripts/microbench_bfly scripts/microbench_bfly.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef unsigned long long u64;

struct Pm {
  u64 q, nq, q4, q5, q8, nq5, nq8;
  uint32_t fs, fmask, fc; // forward: shift - 32, mask of the kept high-word bits, 2^s mod q
  uint32_t is, imask, ic; // inverse
};

__device__ __forceinline__ u64 qt_est(u64 x, u64 ws) {
  const uint32_t x0 = (uint32_t)x, x1 = (uint32_t)(x >> 32), s0 = (uint32_t)ws, s1 = (uint32_t)(ws >> 32);
  return (u64)x1 * s1 + (u64)__umulhi(x1, s0) + (u64)__umulhi(x0, s1);
}
__device__ __forceinline__ u64 mul_add(u64 x, u64 w, u64 ws, u64 nq, u64 a) { return (a + x * w) + qt_est(x, ws) * nq; }

// a + x * w + t * nq (mod 2^64) with every cross product going through v_mad_u64_u32
__device__ __forceinline__ u64 mad64(uint32_t a, uint32_t b, u64 c) {
  u64 d;
  asm("v_mad_u64_u32 %0, vcc, %1, %2, %3" : "=&v"(d) : "v"(a), "v"(b), "v"(c) : "vcc");
  return d;
}
__device__ __forceinline__ u64 mul_add_madlo(u64 x, u64 w, u64 ws, u64 nq, u64 a) {
  const uint32_t x0 = (uint32_t)x, x1 = (uint32_t)(x >> 32), w0 = (uint32_t)w, w1 = (uint32_t)(w >> 32);
  const u64 t = qt_est(x, ws);
  const uint32_t t0 = (uint32_t)t, t1 = (uint32_t)(t >> 32), n0 = (uint32_t)nq, n1 = (uint32_t)(nq >> 32);
  u64 r = mad64(x0, w0, a);
  r = mad64(t0, n0, r);
  u64 h = mad64(x0, w1, r >> 32); // only the low word of h matters from here on
  h = mad64(x1, w0, h);
  h = mad64(t0, n1, h);
  h = mad64(t1, n0, h);
  return (h << 32) | (uint32_t)r;
}

template <int V, bool REDUCE> __device__ __forceinline__ void bfly_fwd(u64 &X, u64 &Y, u64 w, u64 ws, const Pm &p) {
  u64 x = X;
  if (REDUCE) {
    if (V & 1) {
      const uint32_t hi = (uint32_t)(X >> 32);
      x = (((u64)(hi & p.fmask) << 32) | (uint32_t)X) + (u64)(hi >> p.fs) * p.fc;
    } else {
      x = X + (X >= p.q8 ? p.nq8 : 0);
    }
  }
  X = (V & 2) ? mul_add_madlo(Y, w, ws, p.nq, x) : mul_add(Y, w, ws, p.nq, x);
  Y = ((x << 1) + p.q4) - X;
}
template <int V> __device__ __forceinline__ void bfly_inv(u64 &X, u64 &Y, u64 w, u64 ws, const Pm &p) {
  const u64 s = X + Y, d = X + p.q5 - Y;
  if (V & 1) {
    const uint32_t hi = (uint32_t)(s >> 32);
    X = (((u64)(hi & p.imask) << 32) | (uint32_t)s) + (u64)(hi >> p.is) * p.ic;
  } else {
    X = s + (s >= p.q5 ? p.nq5 : 0);
  }
  Y = (V & 2) ? mul_add_madlo(d, w, ws, p.nq, 0) : mul_add(d, w, ws, p.nq, 0);
}

constexpr int ITERS = 512;
// 8 values per thread, 3 stages per iteration (12 butterflies), like one register round of the passes
template <int V, bool INV> __global__ void __launch_bounds__(256) k_round(u64 *out, Pm p, const ulonglong2 *tw, int check) {
  u64 x[8];
#pragma unroll
  for (int i = 0; i < 8; i++) x[i] = (threadIdx.x * 0x9E3779B97F4A7C15ull + i * 0x1234567ull) % p.q;
  ulonglong2 t[6]; // per-lane (w, floor(w 2^64 / q)) pairs in VGPRs, as twiddles from LDS are
#pragma unroll
  for (int s = 0; s < 6; s++) t[s] = tw[s * 256 + threadIdx.x];
  for (int it = 0; it < ITERS / 2; it++) { // two rounds of three stages: plain / reducing stages alternate
#pragma unroll
    for (int s = 0; s < 6; s++) {
      const int half = 4 >> (s % 3);
#pragma unroll
      for (int u = 0; u < 8; u++) {
        if (u & half) continue;
        if (INV) bfly_inv<V>(x[u], x[u + half], t[s].x, t[s].y, p);
        else if ((s & 1) == 0) bfly_fwd<V, false>(x[u], x[u + half], t[s].x, t[s].y, p);
        else bfly_fwd<V, true>(x[u], x[u + half], t[s].x, t[s].y, p);
      }
    }
  }
  if (check) {
    for (int i = 0; i < 8; i++) out[(blockIdx.x * 256 + threadIdx.x) * 8 + i] = x[i] % p.q;
  } else {
    u64 r = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) r += x[i];
    if (r == 0x12345678u) out[0] = r;
  }
}

template <class F> static int timeit(const char *name, F launch) {
  hipEvent_t e0, e1;
  CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  const int blocks = 256 * 8;
  launch(blocks, 0);
  CHK(hipDeviceSynchronize());
  CHK(hipEventRecord(e0));
  for (int r = 0; r < 5; r++) launch(blocks, 0);
  CHK(hipEventRecord(e1));
  CHK(hipEventSynchronize(e1));
  float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
  ms /= 5;
  const double wave_bflies = (double)blocks * 4 * ITERS * 12;
  printf("%-28s %8.3f ms   %7.2f ns per wave-butterfly per SIMD (%.1f cycles at 2.0 GHz)\n", name, ms,
         ms * 1e6 * 1024 / wave_bflies, ms * 1e-3 * 2.0e9 * 1024 / wave_bflies);
  return 0;
}

int main() {
  const u64 q = 0xffffffffe740001ull; // first prime of the N = 2^16, [60]*11 chain
  Pm p{};
  p.q = q; p.nq = 0 - q; p.q4 = 4 * q; p.q5 = 5 * q; p.q8 = 8 * q; p.nq5 = 0 - 5 * q; p.nq8 = 0 - 8 * q;
  p.fs = 31; p.fmask = 0x7fffffffu; p.fc = (uint32_t)((1ull << 63) % q);
  p.is = 30; p.imask = 0x3fffffffu; p.ic = (uint32_t)((1ull << 62) % q);
  std::vector<ulonglong2> twh(6 * 256);
  for (size_t i = 0; i < twh.size(); i++) {
    const u64 w = (0x123456789abcdefull * (i + 1)) % q;
    twh[i] = make_ulonglong2(w, (u64)((((unsigned __int128)w) << 64) / q));
  }
  ulonglong2 *tw;
  CHK(hipMalloc(&tw, twh.size() * sizeof(ulonglong2)));
  CHK(hipMemcpy(tw, twh.data(), twh.size() * sizeof(ulonglong2), hipMemcpyHostToDevice));
  u64 *out; const size_t words = (size_t)8 * 256 * 8;
  CHK(hipMalloc(&out, words * 8));
  std::vector<u64> ref(words), got(words);
#define LAUNCH(V, INV) [&](int b, int chk) { hipLaunchKernelGGL((k_round<V, INV>), dim3(b), dim3(256), 0, 0, out, p, tw, chk); }
#define CHECK(V, INV, NAME) { LAUNCH(V, INV)(8, 1); CHK(hipMemcpy(got.data(), out, words * 8, hipMemcpyDeviceToHost)); \
    if (V == 0) ref = got; else if (got != ref) { printf("%s: residues differ from the base variant\n", NAME); return 1; } }
  CHECK(0, false, "fwd base") CHECK(1, false, "fwd bit") CHECK(2, false, "fwd madlo") CHECK(3, false, "fwd bit+madlo")
  CHECK(0, true, "inv base") CHECK(1, true, "inv bit") CHECK(2, true, "inv madlo") CHECK(3, true, "inv bit+madlo")
  printf("all variants agree mod q\n");
  if (timeit("fwd base", LAUNCH(0, false))) return 1;
  if (timeit("fwd bit-reduce", LAUNCH(1, false))) return 1;
  if (timeit("fwd mad-lo chain", LAUNCH(2, false))) return 1;
  if (timeit("fwd bit-reduce + mad-lo", LAUNCH(3, false))) return 1;
  if (timeit("inv base", LAUNCH(0, true))) return 1;
  if (timeit("inv bit-reduce", LAUNCH(1, true))) return 1;
  if (timeit("inv mad-lo chain", LAUNCH(2, true))) return 1;
  if (timeit("inv bit-reduce + mad-lo", LAUNCH(3, true))) return 1;
  return 0;
}
