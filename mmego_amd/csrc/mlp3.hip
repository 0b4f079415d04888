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
};

// s = gamma / sqrt(var + eps); Wf = s W; bf = (b - mean) s + beta: the expressions of bn_fold_linear_kernel (bn.hip), so
// that folding here and folding beforehand give the same bits
__device__ __forceinline__ float m3_scale(const Mlp3Bn& bn, float eps, int n) { return bn.gamma ? bn.gamma[n] / sqrtf(bn.rvar[n] + eps) : 1.f; }
__device__ __forceinline__ float m3_w(const Mlp3Bn& bn, float eps, int n, float w) { return bn.gamma ? m3_scale(bn, eps, n) * w : w; }
__device__ __forceinline__ float m3_b(const Mlp3Bn& bn, float eps, int n, const float* b) {
  if (!bn.gamma) return b[n];
  return ((b ? b[n] : 0.f) - bn.rmean[n]) * m3_scale(bn, eps, n) + bn.beta[n];
}

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
  __shared__ float Xs[M3_ROWS * M3_S32], Y1s[M3_ROWS * M3_S32], Y2s[M3_ROWS * M3_S64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // weights (zero padded) -> LDS, once per workgroup
  for (int i = tid; i < 32 * 32; i += 256) { int n = i >> 5, k = i & 31; W1s[n * M3_S32 + k] = (n < p.C1 && k < p.Cin) ? m3_w(p.bn[0], p.eps, n, p.W1[n * p.Cin + k]) : 0.f; }
  for (int i = tid; i < 64 * 32; i += 256) { int n = i >> 5, k = i & 31; W2s[n * M3_S32 + k] = (n < p.C2 && k < p.C1) ? m3_w(p.bn[1], p.eps, n, p.W2[n * p.C1 + k]) : 0.f; }
  for (int i = tid; i < 64 * 64; i += 256) { int n = i >> 6, k = i & 63; W3s[n * M3_S64 + k] = (n < p.C3 && k < p.C2) ? m3_w(p.bn[2], p.eps, n, p.W3[n * p.C2 + k]) : 0.f; }
  if (tid < 32) B1s[tid] = tid < p.C1 ? m3_b(p.bn[0], p.eps, tid, p.b1) : 0.f;
  if (tid < 64) { B2s[tid] = tid < p.C2 ? m3_b(p.bn[1], p.eps, tid, p.b2) : 0.f; B3s[tid] = tid < p.C3 ? m3_b(p.bn[2], p.eps, tid, p.b3) : 0.f; }
  const int K1 = (p.Cin + 1) & ~1, K2 = (p.C1 + 1) & ~1, K3 = (p.C2 + 1) & ~1;
  const int rt = wave & 1, ct = wave >> 1;               // 2 x 2 tiles of 32 x 32 over the 64 x 64 stage output
  const int col = ct * 32 + (lane & 31);
  const long ntiles = (p.rows + M3_ROWS - 1) / M3_ROWS;
  // this thread's 8 elements of a 64 x 32 input tile: row (tid >> 5) + 8 j, column tid & 31; the NEXT tile's elements are
  // fetched while the current tile is computed (the loop is otherwise one exposed global-load round trip per tile)
  const int xk = tid & 31, xr = tid >> 5;
  float xv[8];
#define M3_FETCH(tile)                                                                              \
  do {                                                                                              \
    const long rb_ = (tile) * M3_ROWS;                                                              \
    _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                                 \
      const long rr_ = rb_ + xr + 8 * j;                                                            \
      xv[j] = (rr_ < p.rows && xk < p.Cin) ? p.X[rr_ * p.ldx + xk] : 0.f;                           \
    }                                                                                               \
  } while (0)
  if ((long)blockIdx.x < ntiles) M3_FETCH((long)blockIdx.x);
  for (long t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const long r0 = t * M3_ROWS;
    __syncthreads();                                      // previous iteration's readers of Xs / Y1s / Y2s are done
#pragma unroll
    for (int j = 0; j < 8; ++j) Xs[(xr + 8 * j) * M3_S32 + xk] = xv[j];
    __syncthreads();
    if (t + gridDim.x < ntiles) M3_FETCH(t + gridDim.x);
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
    {                                                     // stage 2: 64 x 64 outputs = 4 tiles
      f32x16 acc = m3_tile(Y1s + rt * 32 * M3_S32, M3_S32, W2s + ct * 32 * M3_S32, M3_S32, K2, lane);
      const float bv = B2s[col];
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int row = rt * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
        Y2s[row * M3_S64 + col] = fmaxf(acc[reg] + bv, 0.f);
      }
    }
    __syncthreads();
    {                                                     // stage 3 -> global
      f32x16 acc = m3_tile(Y2s + rt * 32 * M3_S64, M3_S64, W3s + ct * 32 * M3_S64, M3_S64, K3, lane);
      const float bv = B3s[col];
      if (col < p.C3) {
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
          const long row = r0 + rt * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
          if (row < p.rows) p.Y[row * p.ldy + col] = fmaxf(acc[reg] + bv, 0.f);
        }
      }
    }
  }
}

extern "C" int mmego_mlp3_eval(void* stream, const float* X, long ldx, long rows, int Cin, const float* W1, const float* b1, int C1,
                               const float* W2, const float* b2, int C2, const float* W3, const float* b3, int C3, float* Y,
                               long ldy, const float* const* bn, float eps) {
  MMEGO_REQUIRE(X && Y && W1 && W2 && W3 && rows > 0);
  MMEGO_REQUIRE(bn || (b1 && b2 && b3));
  MMEGO_REQUIRE(Cin >= 1 && Cin <= 32 && C1 >= 1 && C1 <= 32 && C2 >= 1 && C2 <= 64 && C3 >= 1 && C3 <= 64);
  Mlp3P p = {X, ldx, rows, Cin, W1, b1, C1, W2, b2, C2, W3, b3, C3, Y, ldy, {}, eps};
  for (int i = 0; i < 3; ++i) {
    if (bn) {
      MMEGO_REQUIRE(bn[4 * i] && bn[4 * i + 1] && bn[4 * i + 2] && bn[4 * i + 3]);
      p.bn[i] = {bn[4 * i], bn[4 * i + 1], bn[4 * i + 2], bn[4 * i + 3]};
    } else {
      p.bn[i] = {nullptr, nullptr, nullptr, nullptr};
    }
  }
  const long ntiles = (rows + M3_ROWS - 1) / M3_ROWS;
  const unsigned grid = (unsigned)(ntiles < 1024 ? ntiles : 1024);    // 83.5 KB of LDS: one workgroup per CU, 4 tiles each at most
  hipLaunchKernelGGL(mlp3_eval_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, p);
  MMEGO_LAUNCH_CHECK();
  return MMEGO_OK;
}
