// Calibration: sustained fp32 (and bf16) MFMA rate of the chip with no memory traffic at all (register operands only).
// Build: hipcc -O3 --offload-arch=gfx950 scripts/mfma_peak.hip -o /tmp/mfma_peak ; run: /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int KIND>
__global__ __launch_bounds__(256) void spin(float* out, int iters) {
  float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
  if (KIND == 0) {
    f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c3, 0, 0, 0);
      }
    }
    out[blockIdx.x * 256 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
  } else if (KIND == 3) {
    // fp32 32x32x2 in the GEMM's own pattern: 2 x 2 accumulator tiles, operands a0/a1 x b0/b1 from a set of registers holding
    // RANDOM-looking floats (hashed bits in [0.5, 2)), a different set every k-step -- register-only, but with the operand toggling
    // of real data (r03: is the nominal peak reachable on real operands at all?)
    float ra[8], rb[8];
    for (int e = 0; e < 8; ++e) {
      unsigned x = (threadIdx.x * 2654435761u) ^ (e * 0x9E3779B9u) ^ (blockIdx.x * 40503u);
      x ^= x >> 15; x *= 0x2c1b3c6du; x ^= x >> 12; x *= 0x297a2d39u; x ^= x >> 15;
      ra[e] = __builtin_bit_cast(float, 0x3f000000u | (x & 0x00ffffffu));
      rb[e] = __builtin_bit_cast(float, 0x3f000000u | ((x * 2246822519u) & 0x00ffffffu));
    }
    f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 8; u += 2) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(ra[u], rb[u], c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(ra[u], rb[u + 1], c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(ra[u + 1], rb[u], c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(ra[u + 1], rb[u + 1], c3, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(rb[u], ra[u], c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(rb[u], ra[u + 1], c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(rb[u + 1], ra[u], c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(rb[u + 1], ra[u + 1], c3, 0, 0, 0);
      }
      // keep the accumulators bounded (random products would overflow): one cheap rescale per 32 MFMAs
      if ((i & 15) == 15) { c0 *= 1e-3f; c1 *= 1e-3f; c2 *= 1e-3f; c3 *= 1e-3f; }
    }
    out[blockIdx.x * 256 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
  } else if (KIND == 2) {
    // bf16 32x32x16 with realistic (non-constant) operand bits: the data pattern moves the power draw and with it the clock
    u32x4 ua, ub;
    for (int e = 0; e < 4; ++e) {
      ua[e] = 0x3f803f80u + 0x01230007u * (threadIdx.x + 17 * e);
      ub[e] = 0x3f003e80u + 0x00510003u * (threadIdx.x * 3 + e);
    }
    const bf16x8 xa = __builtin_bit_cast(bf16x8, ua), xb = __builtin_bit_cast(bf16x8, ub);
    f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa, xb, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xb, xa, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa, xa, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xb, xb, c3, 0, 0, 0);
      }
    }
    out[blockIdx.x * 256 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
  } else {
    f32x4 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0}, c4 = {0}, c5 = {0}, c6 = {0}, c7 = {0};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c3, 0, 0, 0);
        c4 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c4, 0, 0, 0);
        c5 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c5, 0, 0, 0);
        c6 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c6, 0, 0, 0);
        c7 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c7, 0, 0, 0);
      }
    }
    out[blockIdx.x * 256 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3] + c4[0] + c5[1] + c6[2] + c7[3];
  }
}

template <int KIND>
static void run(const char* name, int blocks, int iters, double flops_per_inst, int inst_per_iter) {
  float* out;
  hipMalloc(&out, (size_t)blocks * 256 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(spin<KIND>, dim3(blocks), dim3(256), 0, 0, out, iters);
  hipDeviceSynchronize();
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(spin<KIND>, dim3(blocks), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double fl = (double)blocks * 4 * iters * inst_per_iter * flops_per_inst;
    printf("%s blocks=%d iters=%d: %.3f ms, %.1f TFLOP/s\n", name, blocks, iters, ms, fl / ms / 1e9);
  }
  hipFree(out);
}

int main() {
  // short (like one GEMM launch, ~0.3 ms) and long (2 CU-resident WGs, ~30 ms) runs
  run<0>("32x32x2 short", 256, 2000, 2.0 * 32 * 32 * 2, 32);
  run<0>("32x32x2 long ", 512, 100000, 2.0 * 32 * 32 * 2, 32);
  run<3>("32x32x2 random operands short", 256, 2000, 2.0 * 32 * 32 * 2, 32);
  run<3>("32x32x2 random operands long ", 512, 100000, 2.0 * 32 * 32 * 2, 32);
  run<3>("32x32x2 random operands, 2 WGs/CU x 4 waves, ~0.5 ms", 512, 2000, 2.0 * 32 * 32 * 2, 32);
  run<1>("16x16x4 short", 256, 2000, 2.0 * 16 * 16 * 4, 64);
  run<1>("16x16x4 long ", 512, 100000, 2.0 * 16 * 16 * 4, 64);
  run<2>("bf16 32x32x16 short", 256, 2000, 2.0 * 32 * 32 * 16, 32);
  run<2>("bf16 32x32x16 long ", 512, 100000, 2.0 * 32 * 32 * 16, 32);
  return 0;
}
