// Test aid (r05 finding): a register-heavy, purely arithmetic kernel -- NREG live fp32 values per lane, FMA chains, optionally
// rsqrt / fp64 -- whose output must not depend on what runs beside it.  scripts/coexec_stress.py launches it on one stream while a
// bf16-MFMA kernel runs on another and compares its output with a run on an idle GPU.
#include <hip/hip_runtime.h>
template <int NREG, int MODE>
__global__ __launch_bounds__(64) void victim_kernel(float* out, int iters, float a) {
  float r[NREG];
  const int t = blockIdx.x * 64 + threadIdx.x;
#pragma unroll
  for (int i = 0; i < NREG; ++i) r[i] = 1.0f + 1e-3f * (float)((t * 131 + i * 17) % 1000);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NREG; ++i) {
      float v = fmaf(r[i], a, 0.25f * r[(i + 1) % NREG]);
      if (MODE == 1) v = v * __frsqrt_rn(1.0f + v * v);
      if (MODE == 2) v = (float)((double)v * 0.999 + (double)r[(i + 7) % NREG] * 1e-3);
      if (MODE == 3) v = v / (1.5f + r[(i + 3) % NREG] * r[(i + 3) % NREG]);          // IEEE division: v_div_scale / v_div_fmas (VCC) / v_div_fixup
      if (MODE == 4) v = v > r[(i + 3) % NREG] ? v * 0.5f : v * 0.75f + 0.1f;          // compare + select (VCC / SGPR-pair masks)
      if (MODE == 5) v = sqrtf(1.0f + v * v) * 0.5f;                                    // IEEE square root
      r[i] = v;
    }
  }
#pragma unroll
  for (int i = 0; i < NREG; ++i) out[(long)t * NREG + i] = r[i];
}
// packed fp32 (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32) chains
typedef float vf2 __attribute__((ext_vector_type(2)));
template <int NREG>
__global__ __launch_bounds__(64) void victim_pk_kernel(float* out, int iters, float a) {
  vf2 r[NREG / 2];
  const int t = blockIdx.x * 64 + threadIdx.x;
#pragma unroll
  for (int i = 0; i < NREG / 2; ++i) r[i] = (vf2){1.0f + 1e-3f * (float)((t * 131 + i * 34) % 1000), 1.0f + 1e-3f * (float)((t * 131 + i * 34 + 17) % 1000)};
  const vf2 a2 = {a, a * 0.99f}, q2 = {0.25f, 0.26f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NREG / 2; ++i) r[i] = r[i] * a2 + q2 * r[(i + 1) % (NREG / 2)];
  }
#pragma unroll
  for (int i = 0; i < NREG / 2; ++i) { out[(long)t * NREG + 2 * i] = r[i].x; out[(long)t * NREG + 2 * i + 1] = r[i].y; }
}
extern "C" int victim_launch(void* stream, int nreg, int mode, float* out, int nblk, int iters) {
  hipStream_t st = (hipStream_t)stream;
#define V(N, M) victim_kernel<N, M><<<nblk, 64, 0, st>>>(out, iters, 0.7f)
  if (nreg == 200 && mode == 0) V(200, 0);
  else if (nreg == 200 && mode == 1) V(200, 1);
  else if (nreg == 200 && mode == 2) V(200, 2);
  else if (nreg == 200 && mode == 9) victim_pk_kernel<200><<<nblk, 64, 0, st>>>(out, iters, 0.7f);
  else if (nreg == 200 && mode == 3) V(200, 3);
  else if (nreg == 200 && mode == 4) V(200, 4);
  else if (nreg == 200 && mode == 5) V(200, 5);
  else if (nreg == 40 && mode == 3) V(40, 3);
  else if (nreg == 40 && mode == 0) V(40, 0);
  else if (nreg == 40 && mode == 1) V(40, 1);
  else if (nreg == 100 && mode == 0) V(100, 0);
  else return -1;
  return (int)hipGetLastError();
}
