// ISA-count experiment (no GPU needed): the forward butterfly with the Shoup-quotient twiddle
// product used by ntt.hip.h versus a Solinas-style fold for primes q = 2^60 - delta (every
// CoeffModulus::Create 60-bit prime: delta < 2^26), c = 2^64 mod q = 16 delta < 2^30.
#include <hip/hip_runtime.h>
typedef unsigned long long u64;
__device__ __forceinline__ u64 mul_tw_lazy5_add(u64 x, u64 w, u64 ws, u64 nq, u64 a) {
  const uint32_t x0 = (uint32_t)x, x1 = (uint32_t)(x >> 32), s0 = (uint32_t)ws, s1 = (uint32_t)(ws >> 32);
  const u64 qt = (u64)x1 * s1 + (u64)__umulhi(x1, s0) + (u64)__umulhi(x0, s1);
  return (a + x * w) + qt * nq;
}
// x*w mod q, lazy: T = x*w = T1:T0; T1*c = U1:U0; result = T0 + U0 + U1*c with the (<= 2) carries folded by c again
__device__ __forceinline__ u64 mul_solinas_lazy(u64 x, u64 w, uint32_t c) {
  const unsigned __int128 T = (unsigned __int128)x * w;
  const u64 T0 = (u64)T, T1 = (u64)(T >> 64);
  const unsigned __int128 U = (unsigned __int128)T1 * c;
  const u64 U0 = (u64)U;
  const uint32_t U1 = (uint32_t)(U >> 64);
  unsigned __int128 R = (unsigned __int128)T0 + U0 + (u64)U1 * c;
  const u64 lo = (u64)R;
  const uint32_t hi = (uint32_t)(R >> 64);
  const u64 r = lo + (u64)hi * c;                 // can wrap once more only when lo > 2^64 - 2c: fold that too
  return r < lo ? r + c : r;
}
extern "C" __global__ void bfly_shoup(u64 *d, const ulonglong2 *tw, u64 nq, u64 q4) {
  const int i = threadIdx.x;
  u64 X = d[i], Y = d[i + 64];
  const ulonglong2 w = tw[i];
  u64 x = X;
  X = mul_tw_lazy5_add(Y, w.x, w.y, nq, x);
  Y = ((x << 1) + q4) - X;
  d[i] = X; d[i + 64] = Y;
}
extern "C" __global__ void bfly_solinas(u64 *d, const u64 *tw, uint32_t c, u64 q4) {
  const int i = threadIdx.x;
  u64 X = d[i], Y = d[i + 64];
  const u64 t = mul_solinas_lazy(Y, tw[i], c);
  d[i] = X + t; d[i + 64] = X + q4 - t;
}
