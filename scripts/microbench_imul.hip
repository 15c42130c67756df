// microbench_imul.hip — issue cost of the integer-multiply building blocks on gfx950.
// Prints SIMD cycles per wave64 instruction (assuming 2.4 GHz, 1024 SIMDs busy).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int ITERS = 2048, UNR = 8;

template <int OP> __global__ void __launch_bounds__(256) k(uint32_t *out, uint32_t seed) {
  uint32_t a[UNR], b = seed | 1u;
  unsigned long long acc[UNR];
  double d[UNR];
#pragma unroll
  for (int i = 0; i < UNR; i++) { a[i] = threadIdx.x * 2654435761u + i * 40503u + seed; acc[i] = a[i]; d[i] = a[i]; }
  for (int it = 0; it < ITERS; it++) {
#pragma unroll
    for (int i = 0; i < UNR; i++) {
      if (OP == 0) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
      if (OP == 1) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
      if (OP == 2) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[i]) : "v"(a[i]), "v"(b) : "vcc");
      if (OP == 3) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(b));
      if (OP == 4) asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(b));
      if (OP == 5) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(a[i]) : "v"(b));
      if (OP == 6) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(d[i]) : "v"(1.0000001));
      if (OP == 7) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
      if (OP == 8) asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(a[i]) : "v"(b) : "vcc");
      if (OP == 9) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(acc[i]) : "v"((unsigned long long)b));
      if (OP == 10) asm volatile("v_mad_u32_u16 %0, %0, %1, %0" : "+v"(a[i]) : "v"(b));
      if (OP == 11) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(1.0000001));
      if (OP == 12) asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(acc[i]) : "v"(a[i]), "v"(b) : "vcc");
      if (OP == 13) asm volatile("v_dot4_u32_u8 %0, %0, %1, %0" : "+v"(a[i]) : "v"(b));
    }
  }
  uint32_t r = 0;
#pragma unroll
  for (int i = 0; i < UNR; i++) r += a[i] + (uint32_t)acc[i] + (uint32_t)d[i];
  if (r == 0x12345678u) out[0] = r;
}

// composite: the 64-bit Shoup mulmod as hipcc compiles it
__device__ __forceinline__ unsigned long long shoup(unsigned long long x, unsigned long long w, unsigned long long ws, unsigned long long q) {
  return x * w - __umul64hi(x, ws) * q;
}
__global__ void __launch_bounds__(256) k_shoup(unsigned long long *out, unsigned long long w, unsigned long long ws, unsigned long long q) {
  unsigned long long x[UNR];
#pragma unroll
  for (int i = 0; i < UNR; i++) x[i] = threadIdx.x * 0x9E3779B97F4A7C15ull + i;
  for (int it = 0; it < ITERS; it++) {
#pragma unroll
    for (int i = 0; i < UNR; i++) x[i] = shoup(x[i], w + i, ws, q);
  }
  unsigned long long r = 0;
#pragma unroll
  for (int i = 0; i < UNR; i++) r += x[i];
  if (r == 0x12345678u) out[0] = r;
}
__global__ void __launch_bounds__(256) k_bfly(unsigned long long *out, unsigned long long w, unsigned long long ws, unsigned long long q) {
  unsigned long long x[UNR];
  const unsigned long long q2 = q * 2;
#pragma unroll
  for (int i = 0; i < UNR; i++) x[i] = (threadIdx.x * 0x9E3779B97F4A7C15ull + i) % q;
  for (int it = 0; it < ITERS; it++) {
#pragma unroll
    for (int i = 0; i < UNR; i += 2) {
      unsigned long long X = x[i], Y = x[i + 1];
      unsigned long long xx = X - (X >= q2 ? q2 : 0);
      unsigned long long t = shoup(Y, w + i, ws, q);
      x[i] = xx + t;
      x[i + 1] = xx + q2 - t;
    }
  }
  unsigned long long r = 0;
#pragma unroll
  for (int i = 0; i < UNR; i++) r += x[i];
  if (r == 0x12345678u) out[0] = r;
}

template <class F> static int timeit(const char *name, F launch, double ops_per_thread) {
  hipEvent_t e0, e1;
  CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  const int blocks = 256 * 8;
  launch(blocks);
  CHK(hipDeviceSynchronize());
  CHK(hipEventRecord(e0));
  for (int r = 0; r < 5; r++) launch(blocks);
  CHK(hipEventRecord(e1));
  CHK(hipEventSynchronize(e1));
  float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
  ms /= 5;
  double wave_instrs = (double)blocks * 4 * ops_per_thread;   // 4 waves per 256-thread block
  double simd_cycles = ms * 1e-3 * 2.4e9 * 1024;
  printf("%-22s %8.3f ms   %6.2f SIMD-cycles per wave64 op\n", name, ms, simd_cycles / wave_instrs);
  return 0;
}

int main() {
  uint32_t *out; CHK(hipMalloc(&out, 64));
  const double n = (double)ITERS * UNR;
#define RUN(OP, NAME) if (timeit(NAME, [&](int b) { hipLaunchKernelGGL(k<OP>, dim3(b), dim3(256), 0, 0, out, 12345u); }, n)) return 1;
  RUN(0, "v_mul_lo_u32") RUN(1, "v_mul_hi_u32") RUN(2, "v_mad_u64_u32") RUN(12, "v_mad_i64_i32")
  RUN(3, "v_mul_u32_u24") RUN(4, "v_mul_hi_u32_u24") RUN(5, "v_mad_u32_u24") RUN(10, "v_mad_u32_u16")
  RUN(13, "v_dot4_u32_u8")
  RUN(6, "v_fma_f64") RUN(11, "v_mul_f64") RUN(7, "v_add_u32") RUN(8, "v_add_co_u32") RUN(9, "v_lshl_add_u64")
  unsigned long long q = 0xffffffffffc0001ull, w = 0x123456789abcdefull % q;
  unsigned long long ws = (unsigned long long)((((unsigned __int128)w) << 64) / q);
  if (timeit("shoup mulmod (64b)", [&](int b) { hipLaunchKernelGGL(k_shoup, dim3(b), dim3(256), 0, 0, (unsigned long long *)out, w, ws, q); }, n)) return 1;
  if (timeit("harvey butterfly", [&](int b) { hipLaunchKernelGGL(k_bfly, dim3(b), dim3(256), 0, 0, (unsigned long long *)out, w, ws, q); }, n / 2)) return 1;
  return 0;
}
