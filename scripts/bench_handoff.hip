// What a timestep hand-off costs inside ONE persistent launch (a weight-stationary rnn_slow would replace the
// 8 launches of a layer's recurrence by one kernel whose 256 workgroups exchange h_t -- 64 rows x 512 floats per direction -- through
// memory once per timestep).  This probe runs exactly that exchange with no arithmetic: every workgroup stores its 64 x 4 slice of
// h_t, releases, bumps an arrival counter, waits (bounded spin) until all workgroups of its direction have arrived, acquires, and
// reads the whole 64 x 512 h_t back (as the next step's operand).  Printed: us per timestep, against 9.1 us for the per-step launch
// it would replace (lstm_step_small_kernel in the replayed graph) of which ~4.5 us is the launch boundary itself.
// Every spin is bounded (a stuck grid exits with an error flag instead of hanging the GPU).
// build: hipcc --offload-arch=gfx950 -O3 scripts/bench_handoff.hip -o mmego_amd/build/bench_handoff
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define H 512
#define ROWS 64
#define NWG_DIR 128

// variant B: no cache-wide fences -- the slice goes out as agent-scope (write-through) stores, the read-back as agent-scope loads
__global__ __launch_bounds__(256) void handoff_b_kernel(float* hbuf, unsigned* counters, int steps, unsigned* err, float* sink, int read_back) {
  const int wg = blockIdx.x, d = wg / NWG_DIR, slice = wg % NWG_DIR, tid = threadIdx.x;
  float acc = 0.f;
  for (int s = 0; s < steps; ++s) {
    float* hcur = hbuf + ((size_t)(s & 1) * 2 + d) * ROWS * H;
    __hip_atomic_store(&hcur[(tid >> 2) * H + slice * 4 + (tid & 3)], (float)(s + 1) + acc * 1e-30f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_s_waitcnt(0);                         // this thread's store has been acknowledged
    __syncthreads();
    if (tid == 0) {
      __hip_atomic_fetch_add(&counters[d], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned want = (unsigned)NWG_DIR * (unsigned)(s + 1);
      int spins = 0;
      while (__hip_atomic_load(&counters[d], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
        if (++spins > (1 << 22)) { atomicOr(err, 1u); break; }
      }
    }
    __syncthreads();
    if (read_back) {
      for (int i = tid; i < ROWS * H; i += 256 * 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = __hip_atomic_load(&hcur[i + 256 * u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u];
      }
    }
  }
  if (sink) sink[wg * 256 + tid] = acc;
}

__global__ __launch_bounds__(256) void handoff_kernel(float* hbuf, unsigned* counters, int steps, unsigned* err, float* sink, int read_back) {
  const int wg = blockIdx.x, d = wg / NWG_DIR, slice = wg % NWG_DIR, tid = threadIdx.x;
  float acc = 0.f;
  for (int s = 0; s < steps; ++s) {
    float* hcur = hbuf + ((size_t)(s & 1) * 2 + d) * ROWS * H;
    // this workgroup's slice of h_t: 64 rows x 4 units (one float per thread)
    hcur[(tid >> 2) * H + slice * 4 + (tid & 3)] = (float)(s + 1) + acc * 1e-30f;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __syncthreads();
    if (tid == 0) {
      atomicAdd(&counters[d], 1u);
      const unsigned want = (unsigned)NWG_DIR * (unsigned)(s + 1);
      int spins = 0;
      while (__hip_atomic_load(&counters[d], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
        if (++spins > (1 << 22)) { atomicOr(err, 1u); break; }
        __builtin_amdgcn_s_sleep(1);
      }
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    if (read_back) {                                       // the next step's operand: all 64 x 512 of h_t, 16-byte loads
      const float4* src = reinterpret_cast<const float4*>(hcur);
#pragma unroll 8
      for (int i = tid; i < ROWS * H / 4; i += 256) { const float4 v = src[i]; acc += v.x + v.y + v.z + v.w; }
    }
  }
  if (sink) sink[wg * 256 + tid] = acc;
}

// variant C: the exchange kept inside ONE XCD -- group g = workgroup index % 8 (the dispatcher's round robin over the 8 XCDs), 32
// workgroups per group, each group exchanging its own 16 rows x 512 floats through a counter of its own.  Stores and loads as in
// variant B.  xcc[wg] receives HW_REG_XCC_ID so that the host can say whether a group really sat on one XCD.
__global__ __launch_bounds__(256) void handoff_c_kernel(float* hbuf, unsigned* counters, int steps, unsigned* err, float* sink, int read_back,
                                                        unsigned* xcc) {
  const int wg = blockIdx.x, g = wg & 7, slice = wg >> 3, tid = threadIdx.x;      // 32 slices of 16 units per group
  if (tid == 0) xcc[wg] = __builtin_amdgcn_s_getreg((3 << 11) | 20);              // HW_REG_XCC_ID, bits 0..3
  float acc = 0.f;
  for (int s = 0; s < steps; ++s) {
    float* hcur = hbuf + ((size_t)(s & 1) * 8 + g) * 16 * H;                      // [16 rows][512]
    __hip_atomic_store(&hcur[(tid >> 4) * H + slice * 16 + (tid & 15)], (float)(s + 1) + acc * 1e-30f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    if (tid == 0) {
      __hip_atomic_fetch_add(&counters[g], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned want = 32u * (unsigned)(s + 1);
      int spins = 0;
      while (__hip_atomic_load(&counters[g], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
        if (++spins > (1 << 22)) { atomicOr(err, 1u); break; }
      }
    }
    __syncthreads();
    if (read_back) {
      for (int i = tid; i < 16 * H; i += 256 * 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = __hip_atomic_load(&hcur[i + 256 * u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u];
      }
    }
  }
  if (sink) sink[wg * 256 + tid] = acc;
}

int main(int argc, char** argv) {
  const int steps = argc > 1 ? atoi(argv[1]) : 64;
  float *hbuf, *sink;
  unsigned *counters, *err;
  hipMalloc(&hbuf, sizeof(float) * 4 * ROWS * H);      // (variant C: 2 x 8 groups x 16 rows: the same size)
  hipMalloc(&sink, sizeof(float) * 256 * 256);
  hipMalloc(&counters, 8 * sizeof(unsigned));
  unsigned* xcc;
  hipMalloc(&xcc, 256 * sizeof(unsigned));
  hipMalloc(&err, sizeof(unsigned));
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rb = 0; rb < 4; ++rb) {
    float best = 1e9f;
    for (int it = 0; it < 6; ++it) {
      hipMemset(counters, 0, 2 * sizeof(unsigned));
      hipMemset(err, 0, sizeof(unsigned));
      hipEventRecord(e0);
      if (rb < 2) hipLaunchKernelGGL(handoff_kernel, dim3(2 * NWG_DIR), dim3(256), 0, 0, hbuf, counters, steps, err, sink, rb);
      else hipLaunchKernelGGL(handoff_b_kernel, dim3(2 * NWG_DIR), dim3(256), 0, 0, hbuf, counters, steps, err, sink, rb - 2);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      if (it && ms < best) best = ms;
    }
    unsigned herr = 0;
    hipMemcpy(&herr, err, sizeof(unsigned), hipMemcpyDeviceToHost);
    float chk = 0.f;
    hipMemcpy(&chk, sink, sizeof(float), hipMemcpyDeviceToHost);
    printf("handoff [%s] %s: %.2f us per timestep over %d steps (256 workgroups, counter per direction)%s  [check %.0f]\n",
           rb < 2 ? "A: agent release / acquire fences" : "B: agent-scope stores and loads, no fence",
           (rb & 1) ? "store + counter + read 64x512 h" : "store + counter only", best * 1e3f / steps, steps,
           herr ? "  ** SPIN TIMEOUT **" : "", chk);
  }
  for (int rb = 0; rb < 2; ++rb) {
    float best = 1e9f;
    for (int it = 0; it < 6; ++it) {
      hipMemset(counters, 0, 8 * sizeof(unsigned));
      hipMemset(err, 0, sizeof(unsigned));
      hipEventRecord(e0);
      hipLaunchKernelGGL(handoff_c_kernel, dim3(256), dim3(256), 0, 0, hbuf, counters, steps, err, sink, rb, xcc);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      if (it && ms < best) best = ms;
    }
    unsigned herr = 0, hx[256];
    hipMemcpy(&herr, err, sizeof(unsigned), hipMemcpyDeviceToHost);
    hipMemcpy(hx, xcc, sizeof(hx), hipMemcpyDeviceToHost);
    int same = 1;
    for (int w = 0; w < 256; ++w) same &= (hx[w] & 15) == (hx[w & 7] & 15);
    printf("handoff [C: one XCD per group (workgroup %% 8), 32 workgroups, agent-scope stores and loads] %s: %.2f us per timestep over %d "
           "steps%s; groups on one XCD each: %s (XCC ids of workgroups 0..7: %u %u %u %u %u %u %u %u)\n",
           rb ? "store + counter + read 16x512 h" : "store + counter only", best * 1e3f / steps, steps, herr ? "  ** SPIN TIMEOUT **" : "",
           same ? "yes" : "NO", hx[0] & 15, hx[1] & 15, hx[2] & 15, hx[3] & 15, hx[4] & 15, hx[5] & 15, hx[6] & 15, hx[7] & 15);
  }
  return 0;
}
