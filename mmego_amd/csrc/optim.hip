// Fused multi-tensor Adam over one flat parameter buffer (replaces torch.optim.Adam's foreach chain,
// reference Processor/Train/Train_Upper.py:60,182), plus the dropout mask of the LSTM inter-layer dropout.
// Pure HBM streaming: 16 B/param read (p,g,m,v) + 12 B/param written (p,m,v) = 28 B/param, float4 accesses,
// grid-stride over <= 2048 workgroups.
#include "common.h"

// Ranges of the flat buffer (in float4 units) that the update leaves untouched: parameters that never receive a gradient.
// torch.optim.Adam skips a parameter whose .grad is None (IMU_Net.fc3, never used in forward: Net/IMU_Net.py:55), so with
// weight decay > 0 those tensors must not decay here either.
#define ADAM_MAX_SKIP 4
struct AdamSkip { long lo[ADAM_MAX_SKIP], hi[ADAM_MAX_SKIP]; int n; };

// state[0] = step count (as double), state[1] = step_size = lr / (1 - b1^t), state[2] = sqrt(1 - b2^t)
// Kept in device memory so that a captured HIP graph replays with the right bias corrections.  The step count advances
// INSIDE the update launch: every workgroup reads state[0] = t-1 when it starts and works with t; the workgroup whose ticket
// add comes last (all the others have read the state by then: their add follows their read) stores the new state and puts the
// ticket back to 0.  No separate tick launch.
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, long n4,
                                                   double* state, double lr, double beta1, double beta2d, int* ticket,
                                                   float beta2, float omb1, float omb2, float eps, float weight_decay,
                                                   AdamSkip skip) {
  __shared__ double bc[3];
  if (threadIdx.x == 0) {
    const double t = state[0] + 1.0;
    bc[0] = t;
    bc[1] = lr / (1.0 - pow(beta1, t));
    bc[2] = sqrt(1.0 - pow(beta2d, t));
  }
  __syncthreads();
  const float step_size = (float)bc[1];
  const float bc2_sqrt = (float)bc[2];
  float4* p4 = reinterpret_cast<float4*>(p);
  const float4* g4 = reinterpret_cast<const float4*>(g);
  float4* m4 = reinterpret_cast<float4*>(m);
  float4* v4 = reinterpret_cast<float4*>(v);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    bool skipped = false;
#pragma unroll
    for (int r = 0; r < ADAM_MAX_SKIP; ++r) skipped |= (r < skip.n && i >= skip.lo[r] && i < skip.hi[r]);
    if (skipped) continue;                     // a tensor that never receives a gradient (torch.optim.Adam skips grad=None)
    float4 pp = p4[i], gg = g4[i], mm = m4[i], vv = v4[i];
    float* pa = reinterpret_cast<float*>(&pp);
    float* ga = reinterpret_cast<float*>(&gg);
    float* ma = reinterpret_cast<float*>(&mm);
    float* va = reinterpret_cast<float*>(&vv);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float gr = ga[k];
      if (weight_decay != 0.f) gr = gr + weight_decay * pa[k];
      // torch: exp_avg.lerp_(grad, 1-beta1) == m + (g - m) * (1 - beta1)
      ma[k] = ma[k] + (gr - ma[k]) * omb1;
      va[k] = va[k] * beta2 + omb2 * gr * gr;
      float denom = sqrtf(va[k]) / bc2_sqrt + eps;
      pa[k] = pa[k] - step_size * (ma[k] / denom);
    }
    p4[i] = pp; m4[i] = mm; v4[i] = vv;
  }
  if (threadIdx.x == 0) {
    if (atomicAdd(ticket, 1) == (int)gridDim.x - 1) {
      state[0] = bc[0]; state[1] = bc[1]; state[2] = bc[2];
      *ticket = 0;
    }
  }
}

static inline int ew_blocks(long total) {
  long b = (total + 255) / 256;
  return (int)(b > 2048 ? 2048 : (b < 1 ? 1 : b));
}

extern "C" int mmego_adam_step(void* stream, float* p, const float* g, float* m, float* v, long n, double* state,
                               double lr, double beta1, double beta2, double eps, double weight_decay, const long* skip,
                               int nskip, int* ticket) {
  MMEGO_REQUIRE(p && g && m && v && state && ticket && n > 0 && (n % 4) == 0);
  MMEGO_REQUIRE(nskip >= 0 && nskip <= ADAM_MAX_SKIP && (nskip == 0 || skip));
  AdamSkip sk;
  sk.n = nskip;
  for (int r = 0; r < ADAM_MAX_SKIP; ++r) {
    sk.lo[r] = sk.hi[r] = 0;
    if (r < nskip) {
      MMEGO_REQUIRE(skip[2 * r] >= 0 && skip[2 * r] <= skip[2 * r + 1] && skip[2 * r + 1] <= n && (skip[2 * r] % 4) == 0 &&
                    (skip[2 * r + 1] % 4) == 0);
      sk.lo[r] = skip[2 * r] / 4;
      sk.hi[r] = skip[2 * r + 1] / 4;
    }
  }
  MMEGO_REQUIRE((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(adam_kernel, dim3(ew_blocks(n / 4)), dim3(256), 0, st, p, g, m, v, n / 4, state, lr, beta1, beta2, ticket,
                     (float)beta2, (float)(1.0 - beta1), (float)(1.0 - beta2), (float)eps, (float)weight_decay, sk);
  MMEGO_LAUNCH_CHECK();
  return MMEGO_OK;
}
