// r06 (last sentence): what one CU can pull from its XCD's L2, by path and by occupancy -- the "~20 B / clk / CU" that
// bounds the split3 / bf16 products (DESIGN.md sections 3, 7a, 7c) as a measured table instead of a remark.
//   256 workgroups (one per CU; block b reads the region of group b % 8 = its XCD under round-robin dispatch), every workgroup streams a
//   region of `kb` KB `reps` times; regions of 2 MB per XCD stay L2-resident (4 MB of L2 per XCD), 64 MB do not (HBM / fabric rate).
//   path 0: global_load_dwordx4 into VGPRs, 8 loads in flight per wave;  path 1: global_load_lds_dwordx4 (LDS-DMA), 8 transfers in flight.
//   Printed per (path, waves per CU, region): GB/s per CU, B / clk / CU at the clock the kernel itself measures (s_memtime / s_memrealtime).
// build (cross-compiles without a GPU):  hipcc -O3 --offload-arch=gfx950 scripts/bench_l2_delivery.hip -o mmego_amd/lib/variants/bench_l2_delivery
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));

template <int PATH>
__global__ __launch_bounds__(1024) void stream_kernel(const f4* __restrict__ buf, long nvec, int reps, float* sink, unsigned long long* stamps, int priv) {
  extern __shared__ __attribute__((aligned(16))) f4 lds[];
  const int tid = threadIdx.x, nthr = blockDim.x, lane = tid & 63, w = tid >> 6;
  // priv = 0: the 32 workgroups of an XCD stream the SAME region (operand sharing, as the tiles of a product do); 1: every workgroup its
  // own region (distinct lines per CU: what the L2 can deliver to 32 CUs at once)
  const f4* region = buf + (long)(priv ? blockIdx.x : (blockIdx.x & 7)) * nvec;
  unsigned long long t0 = 0, r0 = 0;
  if (tid == 0) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
  f4 acc = {0.f, 0.f, 0.f, 0.f};
  const long per_round = (long)nthr * 8;                 // vectors per round of the workgroup (8 in flight per thread)
  for (int r = 0; r < reps; ++r) {
    for (long i0 = 0; i0 + per_round <= nvec; i0 += per_round) {
      if (PATH == 0) {
        f4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = region[i0 + (long)u * nthr + tid];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u];
      } else {
        // a wave's transfer = 64 lanes x 16 B = 1 KB into its own LDS slot u (8 slots x 1 KB per wave)
#pragma unroll
        for (int u = 0; u < 8; ++u)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(region + i0 + (long)u * nthr + tid),
                                           (__attribute__((address_space(3))) void*)(lds + (w * 8 + u) * 64), 16, 0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
    }
  }
  if (PATH == 1) acc = lds[(w * 8) * 64 + lane];
  if (tid == 0) {
    stamps[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - t0;
    stamps[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - r0;
  }
  if (acc.x + acc.y + acc.z + acc.w == 12345.678f) sink[blockIdx.x * nthr + tid] = acc.x;
}

int main() {
  const long max_bytes = 8L * 64 * 1024 * 1024;          // 8 regions of up to 64 MB
  f4* buf;
  float* sink;
  unsigned long long* stamps;
  if (hipMalloc(&buf, max_bytes) != hipSuccess || hipMalloc(&sink, 256 * 1024 * 4) != hipSuccess || hipMalloc(&stamps, 512 * 8) != hipSuccess) return 1;
  hipMemset(buf, 0x3c, max_bytes);
  hipFuncSetAttribute((const void*)stream_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
  printf("region   path                      waves/CU  region size  GB/s per CU   shader clock   B/clk/CU   aggregate TB/s\n");
  for (int priv = 0; priv < 2; ++priv)
  for (int path = 0; path < 2; ++path)
    for (long kb : (priv ? std::vector<long>{128L, 2048L} : std::vector<long>{2048L, 65536L}))
      for (int waves : {4, 8, 16}) {
        const int nthr = waves * 64;
        const long nvec = kb * 1024 / 16;
        const int reps = kb == 128 ? 600 : (kb == 2048 ? 40 : 2);
        const size_t lds = path ? (size_t)waves * 8 * 1024 : 0;
        for (int it = 0; it < 2; ++it) {                 // (first pass warms the L2s)
          if (path == 0) stream_kernel<0><<<256, nthr, 0, 0>>>(buf, nvec, reps, sink, stamps, priv);
          else stream_kernel<1><<<256, nthr, lds, 0>>>(buf, nvec, reps, sink, stamps, priv);
          if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 2; }
        }
        std::vector<unsigned long long> h(512);
        hipMemcpy(h.data(), stamps, 512 * 8, hipMemcpyDeviceToHost);
        double cyc = 0, rt = 0;
        for (int b = 0; b < 256; ++b) { cyc += (double)h[2 * b]; rt += (double)h[2 * b + 1]; }
        cyc /= 256; rt /= 256;                            // s_memrealtime: 100 MHz
        const long per_round = (long)nthr * 8;
        const double bytes = (double)reps * (double)(nvec / per_round * per_round) * 16.0;
        const double sec = rt / 1e8, ghz = cyc / rt / 10.0;
        printf("%-8s %-25s %5d     %6ld KB   %9.1f     %6.2f GHz   %7.1f    %8.2f\n", priv ? "own" : "shared", path ? "global_load_lds_dwordx4" : "global_load_dwordx4", waves, kb,
               bytes / sec / 1e9, ghz, bytes / cyc, bytes * 256 / sec / 1e12);
      }
  return 0;
}
