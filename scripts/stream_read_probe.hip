// stream_read_probe.hip — what a read-once stream reaches on this GPU (the ceiling k_hoist_mac's key stream is measured against):
//   hipcc --offload-arch=gfx950 -O3 scripts/stream_read_probe.hip -o /tmp/stream_probe && /tmp/stream_probe
// Reads 1.25 GiB with 8- or 16-byte accesses per lane, default or non-temporal policy, blocked (each workgroup walks a
// contiguous 48 KiB block, as the blocked key copy is read) or grid-strided; prints GB/s per variant (best of 5).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned long long u64;
template <int V, bool NT, bool BLOCKED>
__global__ void __launch_bounds__(256) k_read(const u64 *p, size_t words, u64 *out, uint32_t steps) {
  u64 acc = 0;
  if (BLOCKED) { // workgroup b owns `steps` consecutive chunks of 256 * V words
    const u64 *q = p + (size_t)blockIdx.x * steps * 256 * V + (size_t)threadIdx.x * V;
#pragma unroll 4
    for (uint32_t s = 0; s < steps; s++) {
#pragma unroll
      for (int v = 0; v < V; v++) acc += NT ? __builtin_nontemporal_load(q + (size_t)s * 256 * V + v) : q[(size_t)s * 256 * V + v];
    }
  } else {
    const size_t stride = (size_t)gridDim.x * 256 * V;
#pragma unroll 4
    for (size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * V; i < words; i += stride) {
#pragma unroll
      for (int v = 0; v < V; v++) acc += NT ? __builtin_nontemporal_load(p + i + v) : p[i + v];
    }
  }
  if (acc == 0x123456789abcdefull) out[0] = acc; // never true: keeps the loads
}
template <int V, bool NT, bool BLOCKED> static void run(const char *name, const u64 *d, size_t words, u64 *out) {
  const uint32_t steps = 24; // 24 x 2 KiB (V = 1) = 48 KiB per workgroup
  const uint32_t grid = BLOCKED ? (uint32_t)(words / ((size_t)steps * 256 * V)) : 256 * 16;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9f;
  for (int r = 0; r < 6; r++) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k_read<V, NT, BLOCKED>), dim3(grid), dim3(256), 0, 0, d, words, out, steps);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (r && ms < best) best = ms;
  }
  printf("%-44s %8.1f us  %7.1f GB/s\n", name, best * 1e3, words * 8.0 / best / 1e6);
}
int main() {
  const size_t words = (size_t)160 << 20; // 1.25 GiB
  u64 *d, *out;
  hipMalloc(&d, words * 8); hipMalloc(&out, 8);
  hipMemset(d, 1, words * 8);
  run<1, false, true>("8 B/lane blocked", d, words, out);
  run<1, true, true>("8 B/lane blocked nt", d, words, out);
  run<2, false, true>("16 B/lane blocked", d, words, out);
  run<2, true, true>("16 B/lane blocked nt", d, words, out);
  run<1, false, false>("8 B/lane grid-strided (4096 wgs)", d, words, out);
  run<2, false, false>("16 B/lane grid-strided (4096 wgs)", d, words, out);
  run<2, true, false>("16 B/lane grid-strided nt", d, words, out);
  return 0;
}
