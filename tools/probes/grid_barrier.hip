// What would ONE persistent kernel per decode step cost in synchronisation?
// A decode step is a chain of 59 dependent stages; as one kernel each stage boundary becomes a device-wide barrier over the resident
// workgroups, and whatever a stage wrote has to be visible to workgroups on the other seven XCDs (their L2s are not coherent with each
// other: a release at agent scope writes the XCD's dirty L2 lines back, an acquire invalidates).  This probe runs 256 workgroups x 256
// threads (one per CU) through N barriers of the usual sense-reversing kind (agent-scope fetch-add + spin on a generation word), each
// preceded by every thread writing 16 bytes (the activations a stage produces: 1 MiB per stage in all) and followed by a read of another
// workgroup's slot, and prints the time per barrier — next to the ~8 us a dependent graph node costs (tools/bench_dec_linear.py).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/grid_barrier tools/probes/grid_barrier.hip && /tmp/grid_barrier
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k(unsigned* sync, f4* buf, int nbar, int payload, float* sink) {
  unsigned gen = 0;
  float acc = 0.0f;
  for (int i = 0; i < nbar; ++i) {
    if (payload) { const f4 w = {(float)i, acc, 0.0f, 0.0f}; buf[(size_t)blockIdx.x * 256 + threadIdx.x] = w; }
    __syncthreads();
    if (threadIdx.x == 0) {
      __atomic_thread_fence(__ATOMIC_RELEASE);  // (agent scope: this XCD's dirty lines go back to memory)
      const unsigned arrived = __hip_atomic_fetch_add(&sync[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (arrived == gridDim.x * (gen + 1) - 1) __hip_atomic_store(&sync[1], gen + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      while (__hip_atomic_load(&sync[1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) <= gen) __builtin_amdgcn_s_sleep(1);
    }
    ++gen;
    __syncthreads();
    if (payload) {  // a neighbour's slot, written on another XCD (consecutive block ids sit on consecutive XCDs)
      const f4 v = __builtin_nontemporal_load(&buf[(size_t)((blockIdx.x + 1) % gridDim.x) * 256 + threadIdx.x]);
      acc += v.x;
    }
  }
  if (acc == -1.0f) sink[0] = acc;
}

int main() {
  unsigned* sync; f4* buf; float* sink;
  hipMalloc(&sync, 64); hipMalloc(&buf, 256 * 256 * sizeof(f4)); hipMalloc(&sink, 4);
  int ncu = 256;
  hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int payload = 0; payload < 2; ++payload)
    for (int grid : {ncu, ncu / 2, 64}) {
      const int nbar = 2000;
      float best = 1e9f;
      for (int rep = 0; rep < 3; ++rep) {
        hipMemset(sync, 0, 64);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, sync, buf, nbar, payload, sink);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
      }
      printf("%3d workgroups, %s: %.2f us per device-wide barrier (%d barriers in %.3f ms)\n", grid,
             payload ? "16 B per thread written before / read after" : "barrier only", best * 1e3f / nbar, nbar, best);
    }
  return 0;
}
