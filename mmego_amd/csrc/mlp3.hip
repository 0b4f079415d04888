// Eval-mode pointwise MLP: three (k=1 conv + BatchNorm + ReLU) stages over rows in ONE kernel, intermediates never leave the
// CU.  Replaces, in eval mode, the three conv/BN/ReLU stages of the PointNet blocks (reference Net/Upper_Net.py:242-266 BasePointNet,
// :270-301 GlobalPointNet, Net/Lower_Net.py:40-72) whose per-point activations (32 + 48 + 64 channels x 4 B per point) are
// otherwise written to and re-read from HBM: at the large-batch shape (8.4 M points) that is ~10 GB of traffic per net.
// BatchNorm (running statistics) is folded into the convs while the weights are staged in LDS (the arithmetic of
// mmego_bn_fold_linear, bn.hip: no fold launches in front of the kernel), so a stage is y = relu(W x + b).
//   * workgroup = 64 rows per iteration (persistent loop), 4 waves; v_mfma_f32_32x32x2_f32 tiles: stage outputs padded to
//     32 / 64 / 64 channels, k padded to even; operands read from LDS as [row][k] with odd row strides (conflict-free).
//   * limits: Cin <= 32, C1 <= 32, C2 <= 64, C3 <= 64 (every PointNet of the path: 6-8-16-24, 28-32-48-64, 6-16-32-61).
#include "common.h"

#define M3_ROWS 64
#define M3_S32 33     // row stride of 32-wide LDS tiles
#define M3_S64 65     // row stride of 64-wide LDS tiles

struct Mlp3Bn { const float* gamma; const float* beta; const float* rmean; const float* rvar; };   // gamma == NULL: W, b are folded already
struct Mlp3P {
  const float* X; long ldx; long rows; int Cin;
  const float* W1; const float* b1; int C1;
  const float* W2; const float* b2; int C2;
  const float* W3; const float* b3; int C3;
  float* Y; long ldy;
  Mlp3Bn bn[3]; float eps;
  int pre;                 // the first `pre` input columns of every row are also written in FRONT of the row's outputs: Y[row * ldy - pre + c]
};

// BatchNorm folding: s = gamma / sqrt(var + eps); Wf = s W; bf = (b - mean) s + beta -- the expressions of bn_fold_linear_kernel
// (bn.hip), so that folding here and folding beforehand give the same bits.

// one 32x32 output tile: acc = A[rt*32.., 0:K] . W[ct*32.., 0:K]^T ; A row stride as, W row stride ws; K even
__device__ __forceinline__ f32x16 m3_tile(const float* A, int as, const float* W, int ws, int K, int lane) {
  const int r = lane & 31, h = lane >> 5;
  f32x16 acc = {0};
  const float* ap = A + r * as + h;
  const float* wp = W + r * ws + h;
  for (int k = 0; k < K; k += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[k], wp[k], acc, 0, 0, 0);
  return acc;
}

__global__ __launch_bounds__(256) void mlp3_eval_kernel(Mlp3P p) {
  __shared__ float W1s[32 * M3_S32], W2s[64 * M3_S32], W3s[64 * M3_S64];
  __shared__ float B1s[32], B2s[64], B3s[64];
  // (the input tile shares its LDS with the stage-2 output: Xs is last read in stage 1, Y2s first written behind the barrier that
  // ends stage 1 -- 8.4 KB less, and at 75 KB two workgroups fit on a CU)
  __shared__ float Y1s[M3_ROWS * M3_S32], Y2s[M3_ROWS * M3_S64];
  float* const Xs = Y2s;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // Weights (zero padded, BatchNorm folded) -> LDS, once per workgroup.  Two phases, every load unconditional from a clamped index and
  // all of a phase's loads in flight together: first the per-channel scales and folded biases (threads 0..63), then the 28 weight
  // elements of each thread.  (Written as `(n < C && k < K) ? fold(W[..]) : 0` every element was a branch around three dependent
  // loads -- W, gamma, running_var -- with a full wait each: 28 serial round trips before the first tile.)
  __shared__ float S1s[32], S2s[64], S3s[64];
#define M3_PIN(v) asm volatile("" : "+v"(v))
  if (tid < 64) {
    const int n1 = min(tid, p.C1 - 1), n2 = min(tid, p.C2 - 1), n3 = min(tid, p.C3 - 1);
    const bool fold = p.bn[0].gamma != nullptr;       // (all three layers carry BatchNorm vectors, or none does)
    float s1 = 1.f, s2 = 1.f, s3 = 1.f, b1, b2, b3;
    if (fold) {
      float g1 = p.bn[0].gamma[n1], v1 = p.bn[0].rvar[n1], m1 = p.bn[0].rmean[n1], e1 = p.bn[0].beta[n1];
      float g2 = p.bn[1].gamma[n2], v2 = p.bn[1].rvar[n2], m2 = p.bn[1].rmean[n2], e2 = p.bn[1].beta[n2];
      float g3 = p.bn[2].gamma[n3], v3 = p.bn[2].rvar[n3], m3 = p.bn[2].rmean[n3], e3 = p.bn[2].beta[n3];
      float c1 = p.b1 ? p.b1[n1] : 0.f, c2 = p.b2 ? p.b2[n2] : 0.f, c3 = p.b3 ? p.b3[n3] : 0.f;
      s1 = g1 / sqrtf(v1 + p.eps); s2 = g2 / sqrtf(v2 + p.eps); s3 = g3 / sqrtf(v3 + p.eps);
      b1 = (c1 - m1) * s1 + e1; b2 = (c2 - m2) * s2 + e2; b3 = (c3 - m3) * s3 + e3;
    } else {
      b1 = p.b1[n1]; b2 = p.b2[n2]; b3 = p.b3[n3];
    }
    if (tid < 32) { S1s[tid] = s1; B1s[tid] = tid < p.C1 ? b1 : 0.f; }
    S2s[tid] = s2; B2s[tid] = tid < p.C2 ? b2 : 0.f;
    S3s[tid] = s3; B3s[tid] = tid < p.C3 ? b3 : 0.f;
  }
  __syncthreads();
  {
    const bool fold = p.bn[0].gamma != nullptr;
    float w1[4], w2[8], w3[16];
#pragma unroll
    for (int u = 0; u < 4; ++u) { const int i = tid + 256 * u, n = i >> 5, k = i & 31; w1[u] = p.W1[min(n, p.C1 - 1) * p.Cin + min(k, p.Cin - 1)]; }
#pragma unroll
    for (int u = 0; u < 8; ++u) { const int i = tid + 256 * u, n = i >> 5, k = i & 31; w2[u] = p.W2[min(n, p.C2 - 1) * p.C1 + min(k, p.C1 - 1)]; }
#pragma unroll
    for (int u = 0; u < 16; ++u) { const int i = tid + 256 * u, n = i >> 6, k = i & 63; w3[u] = p.W3[min(n, p.C3 - 1) * p.C2 + min(k, p.C2 - 1)]; }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = tid + 256 * u, n = i >> 5, k = i & 31;
      M3_PIN(w1[u]);
      const float wf = fold ? S1s[n] * w1[u] : w1[u];
      W1s[n * M3_S32 + k] = (n < p.C1 && k < p.Cin) ? wf : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = tid + 256 * u, n = i >> 5, k = i & 31;
      M3_PIN(w2[u]);
      const float wf = fold ? S2s[n] * w2[u] : w2[u];
      W2s[n * M3_S32 + k] = (n < p.C2 && k < p.C1) ? wf : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int i = tid + 256 * u, n = i >> 6, k = i & 63;
      M3_PIN(w3[u]);
      const float wf = fold ? S3s[n] * w3[u] : w3[u];
      W3s[n * M3_S64 + k] = (n < p.C3 && k < p.C2) ? wf : 0.f;
    }
  }
#undef M3_PIN
  const int K1 = (p.Cin + 1) & ~1, K2 = (p.C1 + 1) & ~1, K3 = (p.C2 + 1) & ~1;
  const int rt = wave & 1, ct = wave >> 1;               // 2 x 2 tiles of 32 x 32 over the 64 x 64 stage output
  const int col = ct * 32 + (lane & 31);
  const long ntiles = (p.rows + M3_ROWS - 1) / M3_ROWS;
  // this thread's 8 elements of a 64 x 32 input tile: row (tid >> 5) + 8 j, column tid & 31; the NEXT tile's elements are
  // fetched while the current tile is computed (the loop is otherwise one exposed global-load round trip per tile)
  const int xk = tid & 31, xr = tid >> 5;
  const int xkc = xk < p.Cin ? xk : p.Cin - 1;
  float xv[8];
  // (clamped addresses, all eight loads issued, THEN the selects; the asm keeps each loaded value live outside its select so that
  // the compiler cannot turn load + select back into a branch around the load -- which costs an s_waitcnt vmcnt(0) per load)
#define M3_FETCH(tile)                                                                              \
  do {                                                                                              \
    const long rb_ = (tile) * M3_ROWS;                                                              \
    _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                                 \
      const long rr_ = rb_ + xr + 8 * j;                                                            \
      xv[j] = p.X[(rr_ < p.rows ? rr_ : p.rows - 1) * p.ldx + xkc];                                 \
    }                                                                                               \
    _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                                 \
      const long rr_ = rb_ + xr + 8 * j;                                                            \
      asm volatile("" : "+v"(xv[j]));                                                               \
      xv[j] = (rr_ < p.rows && xk < p.Cin) ? xv[j] : 0.f;                                           \
    }                                                                                               \
  } while (0)
  // (prefetches are unconditional, past the last tile on a clamped index: loads under a condition have their values copied -- and
  // waited for -- where the condition ends, in front of the tile's MFMAs)
  M3_FETCH((long)blockIdx.x < ntiles ? (long)blockIdx.x : ntiles - 1);
  for (long t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const long r0 = t * M3_ROWS;
    // (r06) Lower_Net's cat(xyz, features) rows: the xyz columns come from THIS launch, next to the features of the same rows, so that
    // the L2 sees whole lines -- written by the selection kernel they were 12 bytes into every 512-byte row of a 1-GB buffer at config
    // 5: 150 us of isolated partial-line writes
    if (p.pre > 0 && tid < M3_ROWS * p.pre) {
      const int row = tid / p.pre, c = tid - row * p.pre;
      if (r0 + row < p.rows) p.Y[(r0 + row) * p.ldy - p.pre + c] = p.X[(r0 + row) * p.ldx + c];
    }
    __syncthreads();                                      // previous iteration's readers of Xs / Y1s / Y2s are done
#pragma unroll
    for (int j = 0; j < 8; ++j) Xs[(xr + 8 * j) * M3_S32 + xk] = xv[j];
    __syncthreads();
    M3_FETCH(t + gridDim.x < ntiles ? t + gridDim.x : t);
    __builtin_amdgcn_sched_barrier(0);
    if (ct == 0) {                                        // stage 1: 64 x 32 outputs = 2 tiles (waves 0, 1)
      f32x16 acc = m3_tile(Xs + rt * 32 * M3_S32, M3_S32, W1s, M3_S32, K1, lane);
      const float bv = B1s[lane & 31];
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int row = rt * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
        Y1s[row * M3_S32 + (lane & 31)] = fmaxf(acc[reg] + bv, 0.f);
      }
    }
    __syncthreads();
    if (ct * 32 < p.C2) {                                 // stage 2: 64 x 64 outputs = 4 tiles (column tiles that are all padding: skipped)
      f32x16 acc = m3_tile(Y1s + rt * 32 * M3_S32, M3_S32, W2s + ct * 32 * M3_S32, M3_S32, K2, lane);
      const float bv = B2s[col];
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int row = rt * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
        Y2s[row * M3_S64 + col] = fmaxf(acc[reg] + bv, 0.f);
      }
    }
    __syncthreads();
    if (ct * 32 < p.C3) {                                 // stage 3 -> global
      f32x16 acc = m3_tile(Y2s + rt * 32 * M3_S64, M3_S64, W3s + ct * 32 * M3_S64, M3_S64, K3, lane);
      const float bv = B3s[col];
      // (final values in registers of their own before the first store, whole tiles stored without a predicate per element: a
      // value computed under a store's predicate shares one register, and overwriting a store's data register waits for the store)
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) acc[reg] = fmaxf(acc[reg] + bv, 0.f);
      if (col < p.C3) {
        float* yp = p.Y + (r0 + rt * 32 + 4 * (lane >> 5)) * p.ldy + col;
        if (r0 + M3_ROWS <= p.rows) {
#pragma unroll
          for (int reg = 0; reg < 16; ++reg) yp[(long)((reg & 3) + 8 * (reg >> 2)) * p.ldy] = acc[reg];
        } else {
#pragma unroll
          for (int reg = 0; reg < 16; ++reg) {
            const long row = r0 + rt * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
            if (row < p.rows) yp[(long)((reg & 3) + 8 * (reg >> 2)) * p.ldy] = acc[reg];
          }
        }
      }
    }
  }
}

extern "C" int mmego_mlp3_eval(void* stream, const float* X, long ldx, long rows, int Cin, const float* W1, const float* b1, int C1,
                               const float* W2, const float* b2, int C2, const float* W3, const float* b3, int C3, float* Y,
                               long ldy, const float* const* bn, float eps, int pre) {
  MMEGO_REQUIRE(X && Y && W1 && W2 && W3 && rows > 0);
  MMEGO_REQUIRE(bn || (b1 && b2 && b3));
  MMEGO_REQUIRE(Cin >= 1 && Cin <= 32 && C1 >= 1 && C1 <= 32 && C2 >= 1 && C2 <= 64 && C3 >= 1 && C3 <= 64);
  MMEGO_REQUIRE(pre >= 0 && pre <= 4 && pre <= Cin);
  Mlp3P p = {X, ldx, rows, Cin, W1, b1, C1, W2, b2, C2, W3, b3, C3, Y, ldy, {}, eps, pre};
  for (int i = 0; i < 3; ++i) {
    if (bn) {
      MMEGO_REQUIRE(bn[4 * i] && bn[4 * i + 1] && bn[4 * i + 2] && bn[4 * i + 3]);
      p.bn[i] = {bn[4 * i], bn[4 * i + 1], bn[4 * i + 2], bn[4 * i + 3]};
    } else {
      p.bn[i] = {nullptr, nullptr, nullptr, nullptr};
    }
  }
  const long ntiles = (rows + M3_ROWS - 1) / M3_ROWS;
  const unsigned grid = (unsigned)(ntiles < 2048 ? ntiles : 2048);    // 75 KB of LDS: two workgroups per CU
  hipLaunchKernelGGL(mlp3_eval_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, p);
  MMEGO_LAUNCH_CHECK();
  return MMEGO_OK;
}
