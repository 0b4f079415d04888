// How fast can the 32 workgroups of one XCD-aligned group re-read a shared 512-KB buffer with loads of different scopes?
// (A persistent weight-stationary recurrence for 512 rows -- rnn_fast -- would have every workgroup of a group read the group's whole
// h_{t-1}, 128 rows x 512 floats, once per timestep: 16 MB per group and step.  Served by the XCD's L2 that is 1.1 TB/s per XCD and
// fine; if agent-scope loads -- the only ones that are correct wherever the dispatcher puts a workgroup -- go to the memory side every
// time it is 9 TB/s in total and not feasible.)  Every workgroup reads its group's buffer `reps` times; plain loads, agent-scope
// atomic loads, system-scope atomic loads.  Printed: aggregate GB/s.
// build: hipcc --offload-arch=gfx950 -O3 scripts/bench_scope_bw.hip -o mmego_amd/build/bench_scope_bw
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define WORDS (128 * 512)   // floats per group buffer (256 KB); x2 for the 8-byte variant

template <int MODE>
__global__ __launch_bounds__(256) void read_kernel(const float* buf, float* sink, int reps) {
  const int g = blockIdx.x & 7, tid = threadIdx.x;
  const float* b = buf + (size_t)g * WORDS;
  float acc = 0.f;
  for (int r = 0; r < reps; ++r) {
    for (int i0 = 0; i0 < WORDS; i0 += 256 * 16) {
      float v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const float* p = b + i0 + u * 256 + tid;
        if (MODE == 0) v[u] = __builtin_nontemporal_load(p);
        else if (MODE == 1) v[u] = *(const volatile float*)p;
        else if (MODE == 2) v[u] = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else if (MODE == 3) v[u] = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else v[u] = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
#pragma unroll
      for (int u = 0; u < 16; ++u) acc += v[u];
    }
  }
  sink[blockIdx.x * 256 + tid] = acc;
}

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 50;
  float *buf, *sink;
  if (hipMalloc(&buf, sizeof(float) * 8 * WORDS) != hipSuccess || hipMalloc(&sink, sizeof(float) * 256 * 256) != hipSuccess) return 1;
  if (hipMemset(buf, 0, sizeof(float) * 8 * WORDS) != hipSuccess) return 1;
  hipEvent_t e0, e1;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return 1;
  const char* names[5] = {"nontemporal loads", "volatile (plain) loads", "workgroup-scope atomic loads", "agent-scope atomic loads", "system-scope atomic loads"};
  for (int mode = 0; mode < 5; ++mode) {
    float best = 1e9f;
    for (int it = 0; it < 5; ++it) {
      (void)hipEventRecord(e0);
      switch (mode) {
        case 0: hipLaunchKernelGGL(read_kernel<0>, dim3(256), dim3(256), 0, 0, buf, sink, reps); break;
        case 1: hipLaunchKernelGGL(read_kernel<1>, dim3(256), dim3(256), 0, 0, buf, sink, reps); break;
        case 2: hipLaunchKernelGGL(read_kernel<2>, dim3(256), dim3(256), 0, 0, buf, sink, reps); break;
        case 3: hipLaunchKernelGGL(read_kernel<3>, dim3(256), dim3(256), 0, 0, buf, sink, reps); break;
        default: hipLaunchKernelGGL(read_kernel<4>, dim3(256), dim3(256), 0, 0, buf, sink, reps); break;
      }
      (void)hipEventRecord(e1);
      (void)hipEventSynchronize(e1);
      float ms;
      (void)hipEventElapsedTime(&ms, e0, e1);
      if (it && ms < best) best = ms;
    }
    const double bytes = 256.0 * reps * WORDS * 4.0;
    printf("%-30s %8.1f us for %d reads of 256 KB by each of 256 workgroups: %8.1f GB/s aggregate (%.1f GB/s per workgroup)\n", names[mode],
           best * 1e3, reps, bytes / (best * 1e-3) / 1e9, bytes / 256.0 / (best * 1e-3) / 1e9);
  }
  return 0;
}
