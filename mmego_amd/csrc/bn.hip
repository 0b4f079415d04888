// BatchNorm (batch statistics, channels-last rows x C), its backward, and the small row-wise helpers
// around it.  Replaces the native_batch_norm / relu / cat chains of reference Net/Upper_Net.py:242-301,
// Net/Lower_Net.py:40-72 and Net/GCN.py:106-147.  All HBM-bound: one coalesced pass per kernel, threads
// run along the channel axis (64 consecutive floats per wave row), partial statistics are combined with
// Chan's formula in fp64 so that train-mode statistics are at least as accurate as a two-pass CPU BN.
#include "common.h"

// rows per block of the column-reduction kernels: about 1024 partial blocks for long tensors (the reductions are
// streaming kernels: with one 256-thread block per CU they were load-latency bound at ~2 TB/s, a third of what the
// elementwise kernels reach on the same bytes), never fewer than 16 rows per block.
static inline long rows_per_block(long rows) {
  long r = (rows + 1023) / 1024;
  if (r < 16) r = 16;
  return (r + 3) / 4 * 4;
}

// Column-reduction thread tile: TC = min(64, next pow2 >= C) lanes across the channels (coalesced), TR = 256 / TC lanes
// down the rows, so that narrow tensors (C = 8..32 in the PointNets) still keep every lane busy.
static inline int col_tile(int C) {
  int tc = 8;
  while (tc < C && tc < 64) tc <<= 1;
  return tc;
}

// partial[c][blk] = (n, mean, M2) over this block's rows of column c
// X2 (optional, same shape): a second tensor whose statistics ride along as the virtual channels [C, 2C) (two BatchNorms of
// one st_gcn block in one launch; C a multiple of TC then).
__global__ __launch_bounds__(256) void colstats_kernel(const float* __restrict__ X, long ldx, long rows, int C,
                                                       float* __restrict__ partial, long RPB, int TC,
                                                       const float* __restrict__ X2, long ldx2) {
  __shared__ float sh[256][3];
  const int TR = 256 / TC;
  const int cx = threadIdx.x % TC, ry = threadIdx.x / TC;
  const int cv = blockIdx.y * TC + cx;
  int c = cv;
  if (X2 && cv >= C) { X = X2; ldx = ldx2; c = cv - C; }
  const bool live = cv < (X2 ? 2 * C : C);
  const long r0 = (long)blockIdx.x * RPB;
  const long r1 = min(rows, r0 + RPB);
  float n = 0.f, s = 0.f, ss = 0.f, shift = 0.f;
  if (live) {
    shift = X[r0 * ldx + c];  // block-local shift keeps the sum of squares well conditioned
    for (long r = r0 + ry; r < r1; r += TR) {
      float d = X[r * ldx + c] - shift;
      s += d;
      ss += d * d;
      n += 1.f;
    }
  }
  sh[threadIdx.x][0] = n; sh[threadIdx.x][1] = s; sh[threadIdx.x][2] = ss;
  __syncthreads();
  if (ry == 0 && live) {
    float N = 0.f, S = 0.f, SS = 0.f;
    for (int j = 0; j < TR; ++j) { N += sh[j * TC + cx][0]; S += sh[j * TC + cx][1]; SS += sh[j * TC + cx][2]; }
    float mean_d = S / N;
    float* out = partial + ((long)cv * gridDim.x + blockIdx.x) * 3;    // [channel][block]: the finalize lanes read consecutive records
    out[0] = N;
    out[1] = shift + mean_d;
    out[2] = fmaxf(SS - S * mean_d, 0.f);
  }
}

// Combine partials (Chan, fp64); update running stats exactly like torch (momentum, unbiased running var).
// outputs: mean[C], invstd[C], a[C] = gamma*invstd, b[C] = beta     so that y = (x-mean)*a + b
struct BnStatsSecond { const float* gamma; const float* beta; float* running_mean; float* running_var; float momentum; float eps;
                       float* mean_out; float* invstd_out; float* a_out; float* b_out; };

__global__ __launch_bounds__(64) void bn_finalize_kernel(const float* __restrict__ partial, int nblk, int C, const float* gamma,
                                   const float* beta, float* running_mean, float* running_var, float momentum,
                                   float eps, float* mean_out, float* invstd_out, float* a_out, float* b_out, BnStatsSecond second) {
  int c = blockIdx.x;
  const int lane = threadIdx.x;
  partial += (long)c * nblk * 3;                          // this (virtual) channel's records
  if (c >= C) {                                           // the second BatchNorm of a pair (colstats_kernel's X2)
    c -= C;
    gamma = second.gamma; beta = second.beta; running_mean = second.running_mean; running_var = second.running_var;
    momentum = second.momentum; eps = second.eps;
    mean_out = second.mean_out; invstd_out = second.invstd_out; a_out = second.a_out; b_out = second.b_out;
  }
  // Exact pooled statistics of the (<= 1024) block partials, fp64, no division inside the loops:
  //   N = sum n_b,  mean = sum n_b mean_b / N,  M2 = sum [M2_b + n_b (mean_b - mean)^2]
  // All of a lane's partials (<= 16) are fetched in ONE round of loads; the kernel is pure load latency otherwise.
  // (the per-channel parameters the epilogue needs travel with the same round of loads)
  const float g_c = gamma[c], be_c = beta[c];
  const float rm_c = running_mean ? running_mean[c] : 0.f, rv_c = running_mean ? running_var[c] : 0.f;
  float pn[16], pm[16], p2[16];
#pragma unroll
  for (int u = 0; u < 16; ++u) {
    const int k = lane + 64 * u;
    const bool ok = k < nblk;
    const float* p = partial + (long)(ok ? k : 0) * 3;
    // (load first, select afterwards: the address is clamped, so the load is always legal -- written as `ok ? p[0] : 0` the
    // compiler must not speculate it and puts a branch and an s_waitcnt vmcnt(0) around every single load: 48 dependent round
    // trips instead of one)
    const float v0 = p[0], v1 = p[1], v2 = p[2];
    pn[u] = ok ? v0 : 0.f;
    pm[u] = ok ? v1 : 0.f;
    p2[u] = ok ? v2 : 0.f;
  }
  double N = 0.0, S = 0.0;
#pragma unroll
  for (int u = 0; u < 16; ++u) {
    N += (double)pn[u];
    S += (double)pn[u] * (double)pm[u];
  }
  N = wave_sum_d(N);
  S = wave_sum_d(S);
  const double mean = S / N;
  double M2 = 0.0;
#pragma unroll
  for (int u = 0; u < 16; ++u) {
    const double dm = (double)pm[u] - mean;
    M2 += (double)p2[u] + (double)pn[u] * dm * dm;
  }
  M2 = wave_sum_d(M2);
  if (lane != 0) return;
  double var = M2 / N;
  float invstd = (float)(1.0 / sqrt(var + (double)eps));
  mean_out[c] = (float)mean;
  invstd_out[c] = invstd;
  a_out[c] = g_c * invstd;
  b_out[c] = be_c;
  if (running_mean) {
    running_mean[c] = (1.f - momentum) * rm_c + momentum * (float)mean;
    double unbiased = N > 1.0 ? M2 / (N - 1.0) : var;
    running_var[c] = (1.f - momentum) * rv_c + momentum * (float)unbiased;
  }
}

// eval mode: y = (x - running_mean) * gamma/sqrt(running_var+eps) + beta
__global__ void bn_eval_affine_kernel(int C, const float* gamma, const float* beta, const float* running_mean,
                                      const float* running_var, float eps, float* mean_out, float* invstd_out,
                                      float* a_out, float* b_out) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float invstd = 1.0f / sqrtf(running_var[c] + eps);
  mean_out[c] = running_mean[c];
  invstd_out[c] = invstd;
  a_out[c] = gamma[c] * invstd;
  b_out[c] = beta[c];
}

// Eval mode: fold BatchNorm(running stats) into the preceding k=1 conv / Linear:
//   y = ((W x + b) - mean) * s + beta,  s = gamma / sqrt(var + eps)   ==   (s W) x + ((b - mean) * s + beta)
// Wf [N, K] = s[n] * W[n, k],  bf[n] = (b[n] - mean[n]) * s[n] + beta[n].  One launch instead of the affine pass over the
// activations (and no pre-BN tensor in memory): the product then runs with bias + ReLU in its epilogue.
__global__ __launch_bounds__(256) void bn_fold_linear_kernel(const float* __restrict__ W, const float* __restrict__ b, int N, int K,
                                                             const float* gamma, const float* beta, const float* rmean,
                                                             const float* rvar, float eps, float* __restrict__ Wf,
                                                             float* __restrict__ bf) {
  const long total = (long)N * K;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int n = (int)(i / K);
    const float s = gamma[n] / sqrtf(rvar[n] + eps);
    Wf[i] = s * W[i];
    if (i - (long)n * K == 0) bf[n] = ((b ? b[n] : 0.f) - rmean[n]) * s + beta[n];
  }
}

// Y[r, c] = act( (X1-m1)*a1+b1 [+ (X2-m2)*a2+b2] )
__global__ __launch_bounds__(256) void affine_act_kernel(const float* __restrict__ X1, long ld1, const float* m1,
                                                         const float* a1, const float* b1,
                                                         const float* __restrict__ X2, long ld2, const float* m2,
                                                         const float* a2, const float* b2, float* __restrict__ Y,
                                                         long ldy, long rows, int C, int relu) {
  long total = rows * C;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    long r = i / C;
    int c = (int)(i - r * C);
    float v = (X1[r * ld1 + c] - m1[c]) * a1[c] + b1[c];
    if (X2) v += (X2[r * ld2 + c] - m2[c]) * a2[c] + b2[c];
    if (relu) v = fmaxf(v, 0.f);
    Y[r * ldy + c] = v;
  }
}

__global__ __launch_bounds__(256) void copy2d_kernel(const float* __restrict__ X, long ldx, float* __restrict__ Y,
                                                     long ldy, long rows, int C, int accumulate) {
  long total = rows * C;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    long r = i / C;
    int c = (int)(i - r * C);
    float v = X[r * ldx + c];
    if (accumulate) v += Y[r * ldy + c];
    Y[r * ldy + c] = v;
  }
}

// two 2-D copies in one launch (concatenation of two sources into one buffer): blocks [0, nb1) copy the first
__global__ __launch_bounds__(256) void copy2d_pair_kernel(const float* __restrict__ X1, long ldx1, float* __restrict__ Y1, long ldy1,
                                                          long rows1, int C1, const float* __restrict__ X2, long ldx2,
                                                          float* __restrict__ Y2, long ldy2, long rows2, int C2, int nb1) {
  const bool second = (int)blockIdx.x >= nb1;
  const float* X = second ? X2 : X1;
  float* Y = second ? Y2 : Y1;
  const long ldx = second ? ldx2 : ldx1, ldy = second ? ldy2 : ldy1, total = second ? rows2 * C2 : rows1 * C1;
  const int C = second ? C2 : C1;
  const long nb = second ? (long)gridDim.x - nb1 : nb1;
  for (long i = (long)(blockIdx.x - (second ? nb1 : 0)) * blockDim.x + threadIdx.x; i < total; i += nb * blockDim.x) {
    const long r = i / C;
    const int c = (int)(i - r * C);
    Y[r * ldy + c] = X[r * ldx + c];
  }
}

// dZ = dY * [Ymask > 0];  partial[c][blk] = (sum dZ, sum dZ*xhat)   with xhat = (X-mean)*invstd
// A second BatchNorm that shares dY and the mask (st_gcn: relu(BN(tcn) + BN(residual)), GCN.py:140-147) rides along as the
// virtual channels [C, 2C): Bn2 holds its X / mean / invstd (pair.X == nullptr: single).
struct BnSecond { const float* X; long ldx; const float* mean; const float* invstd; const float* a; float* dgamma; float* dbeta; float* dX; long lddx; };

__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const float* __restrict__ dY, long lddy,
                                                            const float* __restrict__ Ymask, long ldm,
                                                            const float* __restrict__ X, long ldx, const float* mean,
                                                            const float* invstd, long rows, int C,
                                                            float* __restrict__ partial, long RPB, int TC, BnSecond pair) {
  __shared__ float sh[256][2];
  const int TR = 256 / TC;
  const int cx = threadIdx.x % TC, ry = threadIdx.x / TC;
  const int cv = blockIdx.y * TC + cx;                     // virtual channel
  const bool second = pair.X != nullptr && cv >= C;
  const int c = second ? cv - C : cv;
  const int Ctot = pair.X ? 2 * C : C;
  if (second) { X = pair.X; ldx = pair.ldx; mean = pair.mean; invstd = pair.invstd; }
  const long r0 = (long)blockIdx.x * RPB;
  const long r1 = min(rows, r0 + RPB);
  float s1 = 0.f, s2 = 0.f;
  if (cv < Ctot) {
    const float mu = mean[c], is = invstd[c];
    for (long r = r0 + ry; r < r1; r += TR) {
      float g = dY[r * lddy + c];
      if (Ymask && !(Ymask[r * ldm + c] > 0.f)) g = 0.f;
      s1 += g;
      s2 += g * ((X[r * ldx + c] - mu) * is);
    }
  }
  sh[threadIdx.x][0] = s1; sh[threadIdx.x][1] = s2;
  __syncthreads();
  if (ry == 0 && cv < Ctot) {
    float a = 0.f, b = 0.f;
    for (int j = 0; j < TR; ++j) { a += sh[j * TC + cx][0]; b += sh[j * TC + cx][1]; }
    partial[((long)cv * gridDim.x + blockIdx.x) * 2 + 0] = a;           // [virtual channel][block]
    partial[((long)cv * gridDim.x + blockIdx.x) * 2 + 1] = b;
  }
}

__global__ __launch_bounds__(64) void bn_bwd_finalize_kernel(const float* __restrict__ partial, int nblk, int C, long rows, float* dgamma,
                                       float* dbeta, float* c1, float* c2, BnSecond pair) {
  const int c = blockIdx.x, lane = threadIdx.x;           // (virtual channel: [C, 2C) = the second BatchNorm of a pair)
  float q1[16], q2[16];                 // one round of loads (<= 1024 partial blocks)
#pragma unroll
  for (int u = 0; u < 16; ++u) {
    const int k = lane + 64 * u;
    const bool ok = k < nblk;
    const float* p = partial + ((long)c * nblk + (ok ? k : 0)) * 2;
    const float v0 = p[0], v1 = p[1];      // (load, then select: see bn_finalize_kernel)
    q1[u] = ok ? v0 : 0.f;
    q2[u] = ok ? v1 : 0.f;
  }
  double s1 = 0.0, s2 = 0.0;
#pragma unroll
  for (int u = 0; u < 16; ++u) {
    s1 += (double)q1[u];
    s2 += (double)q2[u];
  }
  s1 = wave_sum_d(s1);
  s2 = wave_sum_d(s2);
  if (lane != 0) return;
  if (c >= C) { pair.dbeta[c - C] = (float)s1; pair.dgamma[c - C] = (float)s2; }
  else { dbeta[c] = (float)s1; dgamma[c] = (float)s2; }
  c1[c] = (float)(s1 / (double)rows);                     // (c1 / c2: [Ctot] each)
  c2[c] = (float)(s2 / (double)rows);
}

// dX = a * (dZ - c1 - xhat*c2)
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ dY, long lddy,
                                                           const float* __restrict__ Ymask, long ldm,
                                                           const float* __restrict__ X, long ldx, const float* mean,
                                                           const float* invstd, const float* a, const float* c1,
                                                           const float* c2, float* __restrict__ dX, long lddx,
                                                           long rows, int C, BnSecond pair) {
  long total = rows * C;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    long r = i / C;
    int c = (int)(i - r * C);
    float g = dY[r * lddy + c];
    if (Ymask && !(Ymask[r * ldm + c] > 0.f)) g = 0.f;
    float xh = (X[r * ldx + c] - mean[c]) * invstd[c];
    dX[r * lddx + c] = a[c] * (g - c1[c] - xh * c2[c]);
    if (pair.X) {                                          // the second BatchNorm of a pair: same dY and mask
      float xh2 = (pair.X[r * pair.ldx + c] - pair.mean[c]) * pair.invstd[c];
      pair.dX[r * pair.lddx + c] = pair.a[c] * (g - c1[C + c] - xh2 * c2[C + c]);
    }
  }
}

// partial[c][blk] = sum over the block's rows of X[r,c]
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ X, long ldx, long rows, int C,
                                                             float* __restrict__ partial, long RPB, int TC) {
  __shared__ float sh[256];
  const int TR = 256 / TC;
  const int cx = threadIdx.x % TC, ry = threadIdx.x / TC;
  const int c = blockIdx.y * TC + cx;
  const long r0 = (long)blockIdx.x * RPB;
  const long r1 = min(rows, r0 + RPB);
  float s = 0.f;
  if (c < C)
    for (long r = r0 + ry; r < r1; r += TR) s += X[r * ldx + c];
  sh[threadIdx.x] = s;
  __syncthreads();
  if (ry == 0 && c < C) {
    float a = 0.f;
    for (int j = 0; j < TR; ++j) a += sh[j * TC + cx];
    partial[(long)c * gridDim.x + blockIdx.x] = a;                     // [channel][block]
  }
}

// Long, wide tensors (r06; the gate gradients of stage-1 IMU_Net training: 10 240 rows x 2048 columns per direction -- the 16-row blocks
// of the kernel above are 20 480 workgroups with four scalar loads per thread there: 45 us for 84 MB): 128 rows x 64 columns per
// workgroup, a thread takes 4 consecutive columns (16-byte loads) of every 16th row, all 8 loads in flight at once.
__global__ __launch_bounds__(256) void colsum_partial4_kernel(const float* __restrict__ X, long ldx, long rows, int C,
                                                              float* __restrict__ partial) {
  __shared__ f32x4 sh4[256];
  const int cx = threadIdx.x & 15, ry = threadIdx.x >> 4;
  const int c = (blockIdx.y * 16 + cx) * 4;
  const long r0 = (long)blockIdx.x * 128;
  f32x4 v[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const long r = r0 + ry + 16 * u;
    const long rc = r < rows ? r : rows - 1;                              // (clamped address: load, then select)
    v[u] = c < C ? *reinterpret_cast<const f32x4*>(X + rc * ldx + c) : f32x4{0.f, 0.f, 0.f, 0.f};
    if (r >= rows) v[u] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  f32x4 s4 = v[0];
#pragma unroll
  for (int u = 1; u < 8; ++u) s4 += v[u];
  sh4[threadIdx.x] = s4;
  __syncthreads();
  if (ry == 0 && c < C) {
    f32x4 a = sh4[cx];
    for (int j = 1; j < 16; ++j) a += sh4[j * 16 + cx];
#pragma unroll
    for (int e = 0; e < 4; ++e) partial[(long)(c + e) * gridDim.x + blockIdx.x] = a[e];       // [channel][block]
  }
}

// Short tensors (<= 1024 rows): one block per column tile sums all rows -- one launch instead of two (a kernel boundary
// costs more than these reductions themselves).  Same fixed summation structure on every run (deterministic).
// seg > 0: columns [seg, C) go to outB / out2B (index c - seg): the gate gradients of both directions of a BiLSTM layer in one launch.
__global__ __launch_bounds__(256) void colsum_small_kernel(const float* __restrict__ X, long ldx, long rows, int C, float* out,
                                                           float* out2, int accumulate, int TC, int seg, float* outB,
                                                           float* out2B, const float* __restrict__ scale) {
  __shared__ double sh[256];
  const int TR = 256 / TC;
  const int cx = threadIdx.x % TC, ry = threadIdx.x / TC;
  const int c = blockIdx.x * TC + cx;
  double s = 0.0;
  if (c < C) {
    long r = ry;
    for (; r + 3L * TR < rows; r += 4L * TR) {           // four loads in flight per thread
      float v0 = X[r * ldx + c], v1 = X[(r + TR) * ldx + c], v2 = X[(r + 2L * TR) * ldx + c], v3 = X[(r + 3L * TR) * ldx + c];
      s += ((double)v0 + (double)v1) + ((double)v2 + (double)v3);
    }
    for (; r < rows; r += TR) s += (double)X[r * ldx + c];
  }
  sh[threadIdx.x] = s;
  __syncthreads();
  if (ry == 0 && c < C) {
    double a = 0.0;
    for (int j = 0; j < TR; ++j) a += sh[j * TC + cx];
    float* o = out;
    float* o2 = out2;
    int cc = c;
    if (seg > 0 && c >= seg) { o = outB; o2 = out2B; cc = c - seg; }
    float af = (float)a;
    if (scale) af *= scale[c];
    const float v = accumulate ? o[cc] + af : af;
    o[cc] = v;
    if (o2) o2[cc] = v;
  }
}

// out2 (optional) receives a copy of the result (nn.LSTM's bias_ih / bias_hh share one gradient)
__global__ __launch_bounds__(64) void colsum_final_kernel(const float* __restrict__ partial, int nblk, int C, float* out, float* out2,
                                                          int accumulate) {
  const int c = blockIdx.x, lane = threadIdx.x;
  float q[16];                          // one round of loads (<= 1024 partial blocks)
#pragma unroll
  for (int u = 0; u < 16; ++u) {
    const int k = lane + 64 * u;
    const float v = partial[(long)c * nblk + (k < nblk ? k : 0)];      // (clamped address: load, then select)
    q[u] = k < nblk ? v : 0.f;
  }
  double s = 0.0;
#pragma unroll
  for (int u = 0; u < 16; ++u) s += (double)q[u];
  s = wave_sum_d(s);
  if (lane == 0) {
    const float v = accumulate ? out[c] + (float)s : (float)s;
    out[c] = v;
    if (out2) out2[c] = v;
  }
}

// Y[i, :] = X[idx[i], :]   (rows of W floats; minibatch assembly from the HBM-resident dataset)
__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ X, const long long* __restrict__ idx,
                                                          float* __restrict__ Y, long nout, long W, long nsrc) {
  const long total = nout * W;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long r = i / W, c = i - r * W;
    const long long s = idx[r];
    Y[i] = (s >= 0 && s < nsrc) ? X[s * W + c] : 0.f;      // out-of-range indices give zero rows instead of a fault
  }
}

// G[r,c] = 0 where H[r,c] <= 0
__global__ __launch_bounds__(256) void relu_mask_kernel(float* __restrict__ G, long ldg, const float* __restrict__ H,
                                                        long ldh, long rows, int C) {
  long total = rows * C;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    long r = i / C;
    int c = (int)(i - r * C);
    if (!(H[r * ldh + c] > 0.f)) G[r * ldg + c] = 0.f;
  }
}

__global__ __launch_bounds__(256) void fill_kernel(float* __restrict__ X, long n, float v) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) X[i] = v;
}

static inline int ew_blocks(long total) {
  long b = (total + 255) / 256;
  return (int)(b > 2048 ? 2048 : (b < 1 ? 1 : b));
}

extern "C" int mmego_colstats_nblk(long rows) { return cdiv(rows, rows_per_block(rows)); }

extern "C" int mmego_bn_train_stats(void* stream, const float* X, long ldx, long rows, int C, const float* gamma,
                                    const float* beta, float* running_mean, float* running_var, float momentum,
                                    float eps, float* partial_ws, float* mean, float* invstd, float* a, float* b) {
  MMEGO_REQUIRE(X && rows > 0 && C > 0 && partial_ws && mean && invstd && a && b);
  hipStream_t st = (hipStream_t)stream;
  const long RPB = rows_per_block(rows);
  int nblk = cdiv(rows, RPB);
  MMEGO_REQUIRE(nblk <= 1024);
  const int TC = col_tile(C);
  BnStatsSecond none = {};
  hipLaunchKernelGGL(colstats_kernel, dim3(nblk, cdiv(C, TC)), dim3(256), 0, st, X, ldx, rows, C, partial_ws, RPB, TC,
                     (const float*)nullptr, 0L);
  MMEGO_LAUNCH_CHECK();
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(C), dim3(64), 0, st, partial_ws, nblk, C, gamma, beta,
                     running_mean, running_var, momentum, eps, mean, invstd, a, b, none);
  MMEGO_LAUNCH_CHECK();
  return MMEGO_OK;
}

extern "C" int mmego_bn_finalize(void* stream, const float* partial, int nblk, int C, const float* gamma, const float* beta,
                                 float* running_mean, float* running_var, float momentum, float eps, float* mean, float* invstd,
                                 float* a, float* b) {
  MMEGO_REQUIRE(partial && nblk >= 1 && nblk <= 1024 && C > 0 && gamma && beta && mean && invstd && a && b);
  MMEGO_REQUIRE((running_mean == nullptr) == (running_var == nullptr));
  BnStatsSecond none = {};
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(C), dim3(64), 0, (hipStream_t)stream, partial, nblk, C, gamma, beta, running_mean,
                     running_var, momentum, eps, mean, invstd, a, b, none);
  MMEGO_LAUNCH_CHECK();
  return MMEGO_OK;
}

extern "C" int mmego_bn_train_stats_pair(void* stream, long rows, int C, float* partial_ws,
                                         const float* X1, long ldx1, const float* gamma1, const float* beta1, float* running_mean1,
                                         float* running_var1, float momentum1, float eps1, float* mean1, float* invstd1, float* a1,
                                         float* b1,
                                         const float* X2, long ldx2, const float* gamma2, const float* beta2, float* running_mean2,
                                         float* running_var2, float momentum2, float eps2, float* mean2, float* invstd2, float* a2,
                                         float* b2) {
  MMEGO_REQUIRE(rows > 0 && C > 0 && partial_ws && X1 && mean1 && invstd1 && a1 && b1 && X2 && mean2 && invstd2 && a2 && b2);
  MMEGO_REQUIRE(gamma1 && beta1 && gamma2 && beta2 && (running_mean1 == nullptr) == (running_var1 == nullptr) &&
                (running_mean2 == nullptr) == (running_var2 == nullptr));
  hipStream_t st = (hipStream_t)stream;
  const long RPB = rows_per_block(rows);
  int nblk = cdiv(rows, RPB);
  MMEGO_REQUIRE(nblk <= 1024);
  const int TC = col_tile(C);
  MMEGO_REQUIRE((C % TC) == 0);                          // (a column tile must not straddle the two tensors)
  BnStatsSecond second = {gamma2, beta2, running_mean2, running_var2, momentum2, eps2, mean2, invstd2, a2, b2};
  hipLaunchKernelGGL(colstats_kernel, dim3(nblk, cdiv(2 * C, TC)), dim3(256), 0, st, X1, ldx1, rows, C, partial_ws, RPB, TC, X2, ldx2);
  MMEGO_LAUNCH_CHECK();
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(2 * C), dim3(64), 0, st, partial_ws, nblk, C, gamma1, beta1,
                     running_mean1, running_var1, momentum1, eps1, mean1, invstd1, a1, b1, second);
  MMEGO_LAUNCH_CHECK();
  return MMEGO_OK;
}

extern "C" int mmego_bn_eval_affine(void* stream, int C, const float* gamma, const float* beta,
                                    const float* running_mean, const float* running_var, float eps, float* mean,
                                    float* invstd, float* a, float* b) {
  MMEGO_REQUIRE(C > 0 && gamma && beta && running_mean && running_var);
  hipLaunchKernelGGL(bn_eval_affine_kernel, dim3(cdiv(C, 64)), dim3(64), 0, (hipStream_t)stream, C, gamma, beta,
                     running_mean, running_var, eps, mean, invstd, a, b);
  MMEGO_LAUNCH_CHECK();
  return MMEGO_OK;
}

extern "C" int mmego_bn_fold_linear(void* stream, const float* W, const float* b, int N, int K, const float* gamma,
                                    const float* beta, const float* running_mean, const float* running_var, float eps,
                                    float* Wf, float* bf) {
  MMEGO_REQUIRE(W && N > 0 && K > 0 && gamma && beta && running_mean && running_var && Wf && bf);
  hipLaunchKernelGGL(bn_fold_linear_kernel, dim3(ew_blocks((long)N * K)), dim3(256), 0, (hipStream_t)stream, W, b, N, K, gamma, beta,
                     running_mean, running_var, eps, Wf, bf);
  MMEGO_LAUNCH_CHECK();
  return MMEGO_OK;
}

extern "C" int mmego_affine_act(void* stream, const float* X1, long ld1, const float* m1, const float* a1,
                                const float* b1, const float* X2, long ld2, const float* m2, const float* a2,
                                const float* b2, float* Y, long ldy, long rows, int C, int relu) {
  MMEGO_REQUIRE(X1 && Y && rows > 0 && C > 0);
  hipLaunchKernelGGL(affine_act_kernel, dim3(ew_blocks(rows * C)), dim3(256), 0, (hipStream_t)stream, X1, ld1, m1, a1,
                     b1, X2, ld2, m2, a2, b2, Y, ldy, rows, C, relu);
  MMEGO_LAUNCH_CHECK();
  return MMEGO_OK;
}

extern "C" int mmego_copy2d(void* stream, const float* X, long ldx, float* Y, long ldy, long rows, int C,
                            int accumulate) {
  MMEGO_REQUIRE(X && Y && rows > 0 && C > 0);
  hipLaunchKernelGGL(copy2d_kernel, dim3(ew_blocks(rows * C)), dim3(256), 0, (hipStream_t)stream, X, ldx, Y, ldy, rows,
                     C, accumulate);
  MMEGO_LAUNCH_CHECK();
  return MMEGO_OK;
}

extern "C" int mmego_copy2d_pair(void* stream, const float* X1, long ldx1, float* Y1, long ldy1, long rows1, int C1, const float* X2,
                                 long ldx2, float* Y2, long ldy2, long rows2, int C2) {
  MMEGO_REQUIRE(X1 && Y1 && X2 && Y2 && rows1 > 0 && C1 > 0 && rows2 > 0 && C2 > 0);
  const int nb1 = ew_blocks(rows1 * C1), nb2 = ew_blocks(rows2 * C2);
  hipLaunchKernelGGL(copy2d_pair_kernel, dim3(nb1 + nb2), dim3(256), 0, (hipStream_t)stream, X1, ldx1, Y1, ldy1, rows1, C1, X2, ldx2, Y2,
                     ldy2, rows2, C2, nb1);
  MMEGO_LAUNCH_CHECK();
  return MMEGO_OK;
}

static int bn_backward_launch(hipStream_t st, const float* dY, long lddy, const float* Ymask, long ldm, const float* X, long ldx,
                              const float* mean, const float* invstd, const float* a, long rows, int C, float* partial_ws,
                              float* c12_ws, float* dgamma, float* dbeta, float* dX, long lddx, BnSecond pair) {
  const long RPB = rows_per_block(rows);
  int nblk = cdiv(rows, RPB);
  MMEGO_REQUIRE(nblk <= 1024);
  const int TC = col_tile(C);
  const int Ctot = pair.X ? 2 * C : C;
  MMEGO_REQUIRE(!pair.X || (C % TC) == 0);               // (a column tile must not straddle the two BatchNorms)
  hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(nblk, cdiv(Ctot, TC)), dim3(256), 0, st, dY, lddy, Ymask, ldm, X, ldx,
                     mean, invstd, rows, C, partial_ws, RPB, TC, pair);
  MMEGO_LAUNCH_CHECK();
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(Ctot), dim3(64), 0, st, partial_ws, nblk, C, rows, dgamma,
                     dbeta, c12_ws, c12_ws + Ctot, pair);
  MMEGO_LAUNCH_CHECK();
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(ew_blocks(rows * C)), dim3(256), 0, st, dY, lddy, Ymask, ldm, X, ldx,
                     mean, invstd, a, c12_ws, c12_ws + Ctot, dX, lddx, rows, C, pair);
  MMEGO_LAUNCH_CHECK();
  return MMEGO_OK;
}

extern "C" int mmego_bn_backward(void* stream, const float* dY, long lddy, const float* Ymask, long ldm,
                                 const float* X, long ldx, const float* mean, const float* invstd, const float* a,
                                 long rows, int C, float* partial_ws, float* c12_ws, float* dgamma, float* dbeta,
                                 float* dX, long lddx) {
  MMEGO_REQUIRE(dY && X && mean && invstd && a && rows > 0 && C > 0 && partial_ws && c12_ws && dgamma && dbeta && dX);
  BnSecond none = {nullptr, 0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0};
  return bn_backward_launch((hipStream_t)stream, dY, lddy, Ymask, ldm, X, ldx, mean, invstd, a, rows, C, partial_ws, c12_ws, dgamma,
                            dbeta, dX, lddx, none);
}

extern "C" int mmego_bn_backward_pair(void* stream, const float* dY, long lddy, const float* Ymask, long ldm, long rows, int C,
                                      float* partial_ws, float* c12_ws,
                                      const float* X1, long ldx1, const float* mean1, const float* invstd1, const float* a1,
                                      float* dgamma1, float* dbeta1, float* dX1, long lddx1,
                                      const float* X2, long ldx2, const float* mean2, const float* invstd2, const float* a2,
                                      float* dgamma2, float* dbeta2, float* dX2, long lddx2) {
  MMEGO_REQUIRE(dY && rows > 0 && C > 0 && partial_ws && c12_ws);
  MMEGO_REQUIRE(X1 && mean1 && invstd1 && a1 && dgamma1 && dbeta1 && dX1 && X2 && mean2 && invstd2 && a2 && dgamma2 && dbeta2 && dX2);
  BnSecond second = {X2, ldx2, mean2, invstd2, a2, dgamma2, dbeta2, dX2, lddx2};
  return bn_backward_launch((hipStream_t)stream, dY, lddy, Ymask, ldm, X1, ldx1, mean1, invstd1, a1, rows, C, partial_ws, c12_ws,
                            dgamma1, dbeta1, dX1, lddx1, second);
}

extern "C" int mmego_colsum(void* stream, const float* X, long ldx, long rows, int C, float* partial_ws, float* out,
                            float* out2, int accumulate, const float* scale) {
  MMEGO_REQUIRE(X && rows > 0 && C > 0 && partial_ws && out);
  MMEGO_REQUIRE(!scale || rows <= 1024);
  hipStream_t st = (hipStream_t)stream;
  if (rows <= 1024) {
    // narrow column tiles (16 lanes across, 16 down the rows): more blocks and 4x shorter per-thread row loops than the
    // 64-wide tile of the long-tensor path -- this kernel's time is the depth of its dependent load rounds
    const int TCs = C >= 16 ? 16 : col_tile(C);
    hipLaunchKernelGGL(colsum_small_kernel, dim3(cdiv(C, TCs)), dim3(256), 0, st, X, ldx, rows, C, out, out2, accumulate, TCs, 0,
                       (float*)nullptr, (float*)nullptr, scale);
    MMEGO_LAUNCH_CHECK();
    return MMEGO_OK;
  }
  const long RPB = rows_per_block(rows);
  int nblk = cdiv(rows, RPB);
  MMEGO_REQUIRE(nblk <= 1024);
  if (rows >= 4096 && C >= 256 && (C % 4) == 0 && (ldx % 4) == 0 && (((uintptr_t)X) & 15) == 0 && cdiv(rows, 128) <= nblk) {
    nblk = cdiv(rows, 128);                 // (fewer blocks than the workspace was sized for)
    hipLaunchKernelGGL(colsum_partial4_kernel, dim3(nblk, cdiv(C, 64)), dim3(256), 0, st, X, ldx, rows, C, partial_ws);
  } else {
    const int TC = col_tile(C);
    hipLaunchKernelGGL(colsum_partial_kernel, dim3(nblk, cdiv(C, TC)), dim3(256), 0, st, X, ldx, rows, C, partial_ws, RPB, TC);
  }
  MMEGO_LAUNCH_CHECK();
  hipLaunchKernelGGL(colsum_final_kernel, dim3(C), dim3(64), 0, st, partial_ws, nblk, C, out, out2, accumulate);
  MMEGO_LAUNCH_CHECK();
  return MMEGO_OK;
}

// two short column sums (<= 1024 rows each) in one launch: blocks [0, nb1) take the first tensor
__global__ __launch_bounds__(256) void colsum_small2_kernel(const float* __restrict__ X1, long ld1, long rows1, int C1, float* out1,
                                                            const float* __restrict__ X2, long ld2, long rows2, int C2, float* out2,
                                                            int nb1) {
  __shared__ double sh[256];
  const bool second = (int)blockIdx.x >= nb1;
  const float* X = second ? X2 : X1;
  const long ldx = second ? ld2 : ld1, rows = second ? rows2 : rows1;
  const int C = second ? C2 : C1;
  float* out = second ? out2 : out1;
  const int cx = threadIdx.x & 15, ry = threadIdx.x >> 4;
  const int c = (blockIdx.x - (second ? nb1 : 0)) * 16 + cx;
  double s = 0.0;
  if (c < C) {
    long r = ry;
    for (; r + 3L * 16 < rows; r += 4L * 16) {
      float v0 = X[r * ldx + c], v1 = X[(r + 16) * ldx + c], v2 = X[(r + 32) * ldx + c], v3 = X[(r + 48) * ldx + c];
      s += ((double)v0 + (double)v1) + ((double)v2 + (double)v3);
    }
    for (; r < rows; r += 16) s += (double)X[r * ldx + c];
  }
  sh[threadIdx.x] = s;
  __syncthreads();
  if (ry == 0 && c < C) {
    double a = 0.0;
    for (int j = 0; j < 16; ++j) a += sh[j * 16 + cx];
    out[c] = (float)a;
  }
}

extern "C" int mmego_colsum2(void* stream, const float* X1, long ld1, long rows1, int C1, float* out1, const float* X2, long ld2,
                             long rows2, int C2, float* out2) {
  MMEGO_REQUIRE(X1 && out1 && X2 && out2 && rows1 > 0 && rows1 <= 1024 && rows2 > 0 && rows2 <= 1024 && C1 > 0 && C2 > 0);
  const int nb1 = cdiv(C1, 16), nb2 = cdiv(C2, 16);
  hipLaunchKernelGGL(colsum_small2_kernel, dim3(nb1 + nb2), dim3(256), 0, (hipStream_t)stream, X1, ld1, rows1, C1, out1, X2, ld2, rows2, C2,
                     out2, nb1);
  MMEGO_LAUNCH_CHECK();
  return MMEGO_OK;
}

extern "C" int mmego_colsum_pair(void* stream, const float* X, long ldx, long rows, int C, float* outA, float* outA2, float* outB,
                                 float* outB2, int accumulate) {
  MMEGO_REQUIRE(X && rows > 0 && rows <= 1024 && C >= 16 && (C % 16) == 0 && outA && outB);
  hipLaunchKernelGGL(colsum_small_kernel, dim3(cdiv(2 * C, 16)), dim3(256), 0, (hipStream_t)stream, X, ldx, rows, 2 * C, outA, outA2,
                     accumulate, 16, C, outB, outB2, (const float*)nullptr);
  MMEGO_LAUNCH_CHECK();
  return MMEGO_OK;
}

extern "C" int mmego_gather_rows(void* stream, const float* X, long nsrc, long W, const long long* idx, long nout, float* Y) {
  MMEGO_REQUIRE(X && idx && Y && nsrc > 0 && W > 0 && nout > 0);
  hipLaunchKernelGGL(gather_rows_kernel, dim3(ew_blocks(nout * W)), dim3(256), 0, (hipStream_t)stream, X, idx, Y, nout, W, nsrc);
  MMEGO_LAUNCH_CHECK();
  return MMEGO_OK;
}

extern "C" int mmego_relu_mask(void* stream, float* G, long ldg, const float* H, long ldh, long rows, int C) {
  MMEGO_REQUIRE(G && H && rows > 0 && C > 0);
  hipLaunchKernelGGL(relu_mask_kernel, dim3(ew_blocks(rows * C)), dim3(256), 0, (hipStream_t)stream, G, ldg, H, ldh,
                     rows, C);
  MMEGO_LAUNCH_CHECK();
  return MMEGO_OK;
}

extern "C" int mmego_fill(void* stream, float* X, long n, float v) {
  MMEGO_REQUIRE(X && n > 0);
  hipLaunchKernelGGL(fill_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, X, n, v);
  MMEGO_LAUNCH_CHECK();
  return MMEGO_OK;
}
