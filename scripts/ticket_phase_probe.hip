// ticket_phase_probe.hip — what does a dependency between two phases cost on MI355X: a kernel boundary inside a replayed
// hipGraph (what evah_execute's captured DAGs pay 44 times per Harris replay) or a ticket-ordered phase boundary inside ONE
// persistent kernel (rot_fallback.hip.h's scheme)?  Build: hipcc --offload-arch=gfx950 -O3 scripts/ticket_phase_probe.hip -o
// scripts/ticket_phase_probe; run on the GPU box.  Prints microseconds per phase for K dependent phases of C chunks each,
// with `work` iterations of dependent integer multiply-adds per thread in every chunk.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s failed: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

__device__ __forceinline__ unsigned long long spin_work(unsigned long long v, int work) {
  for (int i = 0; i < work; i++) v = v * 6364136223846793005ull + 1442695040888963407ull;
  return v;
}
__global__ void __launch_bounds__(256) k_phase(unsigned long long *buf, int work, int phase) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  buf[i] = spin_work(buf[i] + phase, work);
}
// K phases of C chunks, ordered by tickets: bar[0] = tickets, bar[1] = finished chunks
__global__ void __launch_bounds__(256) k_persistent(unsigned long long *buf, int work, unsigned K, unsigned C, unsigned *bar) {
  __shared__ unsigned s_ticket;
  if (threadIdx.x == 0) s_ticket = __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  unsigned ticket = __builtin_amdgcn_readfirstlane(s_ticket);
  while (ticket / C < K) {
    const unsigned ph = ticket / C, ch = ticket % C;
    if (ph > 0) {
      if (threadIdx.x == 0)
        while (__hip_atomic_load(bar + 1, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < ph * C) __builtin_amdgcn_s_sleep(1);
      __syncthreads();
      __threadfence();
    }
    const size_t i = (size_t)ch * blockDim.x + threadIdx.x;
    buf[i] = spin_work(buf[i] + ph, work);
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) {
      __hip_atomic_fetch_add(bar + 1, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      s_ticket = __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    ticket = __builtin_amdgcn_readfirstlane(s_ticket);
  }
}

int main() {
  const unsigned K = 44;
  hipStream_t st;
  CHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  hipEvent_t e0, e1;
  CHK(hipEventCreate(&e0));
  CHK(hipEventCreate(&e1));
  for (unsigned C : {256u, 1024u, 4096u}) {
    unsigned long long *buf;
    unsigned *bar;
    CHK(hipMalloc(&buf, sizeof(unsigned long long) * C * 256));
    CHK(hipMalloc(&bar, 8));
    CHK(hipMemset(buf, 0, sizeof(unsigned long long) * C * 256));
    for (int work : {0, 200, 2000}) {
      // A: K dependent kernels in a captured graph
      hipGraph_t g;
      hipGraphExec_t ge;
      CHK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
      for (unsigned p = 0; p < K; p++) hipLaunchKernelGGL(k_phase, dim3(C), dim3(256), 0, st, buf, work, (int)p);
      CHK(hipStreamEndCapture(st, &g));
      CHK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
      const int reps = 50;
      for (int w = 0; w < 5; w++) CHK(hipGraphLaunch(ge, st));
      CHK(hipStreamSynchronize(st));
      CHK(hipEventRecord(e0, st));
      for (int r = 0; r < reps; r++) CHK(hipGraphLaunch(ge, st));
      CHK(hipEventRecord(e1, st));
      CHK(hipEventSynchronize(e1));
      float ms_graph = 0;
      CHK(hipEventElapsedTime(&ms_graph, e0, e1));
      CHK(hipGraphExecDestroy(ge));
      CHK(hipGraphDestroy(g));
      // B: one persistent kernel, K ticket-ordered phases (grids of 256 / 1024 workgroups)
      float ms_p[2] = {0, 0};
      const unsigned grids[2] = {256, 1024};
      for (int gi = 0; gi < 2; gi++) {
        for (int w = 0; w < 3; w++) {
          CHK(hipMemsetAsync(bar, 0, 8, st));
          hipLaunchKernelGGL(k_persistent, dim3(grids[gi]), dim3(256), 0, st, buf, work, K, C, bar);
        }
        CHK(hipStreamSynchronize(st));
        CHK(hipEventRecord(e0, st));
        for (int r = 0; r < reps; r++) {
          CHK(hipMemsetAsync(bar, 0, 8, st));
          hipLaunchKernelGGL(k_persistent, dim3(grids[gi]), dim3(256), 0, st, buf, work, K, C, bar);
        }
        CHK(hipEventRecord(e1, st));
        CHK(hipEventSynchronize(e1));
        CHK(hipEventElapsedTime(&ms_p[gi], e0, e1));
      }
      std::printf("chunks/phase %5u work %5d : graph of %u kernels %7.2f us/phase | persistent grid 256: %7.2f us/phase, grid 1024: %7.2f us/phase\n", C, work, K,
                  ms_graph * 1e3 / reps / K, ms_p[0] * 1e3 / reps / K, ms_p[1] * 1e3 / reps / K);
    }
    CHK(hipFree(buf));
    CHK(hipFree(bar));
  }
  return 0;
}
