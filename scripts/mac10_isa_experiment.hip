#include <hip/hip_runtime.h>
#include <cstdint>
typedef unsigned long long u64;
typedef unsigned __int128 u128;
struct u128_t { u64 lo, hi; };
// partial products of acc += a*b with the carries of the two 64-bit sums that can overflow
struct MacParts { u64 lo, hi, t, c1, ct; };
__device__ __forceinline__ MacParts mac_parts(const u128_t &acc, u64 a, u64 b) {
  const uint32_t a0 = (uint32_t)a, a1 = (uint32_t)(a >> 32), b0 = (uint32_t)b, b1 = (uint32_t)(b >> 32);
  MacParts p;
  u64 t0, d0, d1;
  asm("v_mad_u64_u32 %0, %1, %2, %3, 0" : "=v"(t0), "=s"(d0) : "v"(a1), "v"(b0));
  asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(p.t), "=s"(p.ct) : "v"(a0), "v"(b1), "v"(t0));
  asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(p.hi), "=s"(d1) : "v"(a1), "v"(b1), "v"(acc.hi));
  asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(p.lo), "=s"(p.c1) : "v"(a0), "v"(b0), "v"(acc.lo));
  return p;
}
#define MAC_STEP(w1, w2, w3, c1, ct, t0, t1, K) \
  K == 1 ? "v_addc_co_u32_e64 " w2 ", " c1 ", " w2 ", 0, " c1 "\n\t" : \
  K == 2 ? "v_addc_co_u32_e64 " w3 ", " c1 ", " w3 ", 0, " c1 "\n\t" : ""
__device__ __forceinline__ void acc128x4(u128_t &A, u64 aa, u64 ab, u128_t &B, u64 ba, u64 bb, u128_t &C, u64 ca, u64 cb, u128_t &D,
                                         u64 da, u64 db) {
  MacParts pa = mac_parts(A, aa, ab), pb = mac_parts(B, ba, bb), pc = mac_parts(C, ca, cb), pd = mac_parts(D, da, db);
#define HALVES(p, X) \
  uint32_t X##0 = (uint32_t)p.lo, X##1 = (uint32_t)(p.lo >> 32), X##2 = (uint32_t)p.hi, X##3 = (uint32_t)(p.hi >> 32); \
  const uint32_t X##t0 = (uint32_t)p.t, X##t1 = (uint32_t)(p.t >> 32);
  HALVES(pa, a) HALVES(pb, b) HALVES(pc, c) HALVES(pd, d)
  // operands: A: %0 w1 %1 w2 %2 w3 %3 c1 %4 ct | B: %5..%9 | C: %10..%14 | D: %15..%19 | t0/t1: A %20 %21, B %22 %23, C %24 %25, D %26 %27
  asm("s_nop 1\n\t"
      "v_addc_co_u32_e64 %1, %3, %1, 0, %3\n\t"   "v_addc_co_u32_e64 %6, %8, %6, 0, %8\n\t"   "v_addc_co_u32_e64 %11, %13, %11, 0, %13\n\t" "v_addc_co_u32_e64 %16, %18, %16, 0, %18\n\t"
      "v_addc_co_u32_e64 %2, %3, %2, 0, %3\n\t"   "v_addc_co_u32_e64 %7, %8, %7, 0, %8\n\t"   "v_addc_co_u32_e64 %12, %13, %12, 0, %13\n\t" "v_addc_co_u32_e64 %17, %18, %17, 0, %18\n\t"
      "v_add_co_u32_e64 %0, %3, %0, %20\n\t"      "v_add_co_u32_e64 %5, %8, %5, %22\n\t"      "v_add_co_u32_e64 %10, %13, %10, %24\n\t"     "v_add_co_u32_e64 %15, %18, %15, %26\n\t"
      "v_addc_co_u32_e64 %1, %3, %1, %21, %3\n\t" "v_addc_co_u32_e64 %6, %8, %6, %23, %8\n\t" "v_addc_co_u32_e64 %11, %13, %11, %25, %13\n\t" "v_addc_co_u32_e64 %16, %18, %16, %27, %18\n\t"
      "v_addc_co_u32_e64 %2, %3, %2, 0, %3\n\t"   "v_addc_co_u32_e64 %7, %8, %7, 0, %8\n\t"   "v_addc_co_u32_e64 %12, %13, %12, 0, %13\n\t" "v_addc_co_u32_e64 %17, %18, %17, 0, %18\n\t"
      "v_addc_co_u32_e64 %2, %4, %2, 0, %4\n\t"   "v_addc_co_u32_e64 %7, %9, %7, 0, %9\n\t"   "v_addc_co_u32_e64 %12, %14, %12, 0, %14\n\t" "v_addc_co_u32_e64 %17, %19, %17, 0, %19"
      : "+v"(a1), "+v"(a2), "+v"(a3), "+s"(pa.c1), "+s"(pa.ct), "+v"(b1), "+v"(b2), "+v"(b3), "+s"(pb.c1), "+s"(pb.ct),
        "+v"(c1), "+v"(c2), "+v"(c3), "+s"(pc.c1), "+s"(pc.ct), "+v"(d1), "+v"(d2), "+v"(d3), "+s"(pd.c1), "+s"(pd.ct)
      : "v"(at0), "v"(at1), "v"(bt0), "v"(bt1), "v"(ct0), "v"(ct1), "v"(dt0), "v"(dt1));
  A.lo = ((u64)a1 << 32) | a0; A.hi = ((u64)a3 << 32) | a2;
  B.lo = ((u64)b1 << 32) | b0; B.hi = ((u64)b3 << 32) | b2;
  C.lo = ((u64)c1 << 32) | c0; C.hi = ((u64)c3 << 32) | c2;
  D.lo = ((u64)d1 << 32) | d0; D.hi = ((u64)d3 << 32) | d2;
}
__global__ void k5(const u64 *a, const u64 *b, u64 *o, int n) {
  u128_t acc[4];
  for (int j = 0; j < 4; j++) acc[j] = {o[threadIdx.x + 128 * j], o[threadIdx.x + 64 + 128 * j]};
  for (int i = 0; i < n; i++) {
    const u64 *pa = a + i * 256 + threadIdx.x, *pb = b + i * 256 + threadIdx.x;
    acc128x4(acc[0], pa[0], pb[0], acc[1], pa[64], pb[64], acc[2], pa[128], pb[128], acc[3], pa[192], pb[192]);
  }
  for (int j = 0; j < 4; j++) { o[threadIdx.x + 128 * j] = acc[j].lo; o[threadIdx.x + 64 + 128 * j] = acc[j].hi; }
}
