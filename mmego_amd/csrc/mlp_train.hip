// Train-mode pointwise MLP layers (k=1 conv -> BatchNorm with batch statistics -> ReLU), fused per layer.
// Replaces, in training, the per-stage chains of reference Net/Upper_Net.py:242-301 (PointNet, GlobalPointNet),
// Net/Lower_Net.py:40-72 (BasePointNet) and Net/Upper_Net.py:147-177 (LocalPointNet):
//   forward, per stage : product + column statistics + finalize + affine/ReLU        4 launches -> 1 (+1 for the last stage)
//   backward, per stage: BN reduce + finalize + apply, dW (+ split-K reduce), dX     6-7 launches -> 1 (+2 per 3-stage block)
// Batch statistics need every row, so a layer boundary stays a kernel boundary; everything else of a layer is one pass:
//   * mlp_fwd_layer   z_l = act_{l-1}(z_{l-1}) W_l^T + b_l.  The PREVIOUS layer's BatchNorm + ReLU is applied while the
//                     operand tile is loaded (y_{l-1} never exists in memory), its statistics are finalized in the prologue
//                     from per-workgroup partial sums (every workgroup redundantly, in a fixed order: no finalize launch;
//                     workgroup 0 stores mean / invstd / scale for the backward pass and updates the running statistics),
//                     and this layer's column sums (sum z, sum z^2, fp64) leave with the epilogue.
//   * mlp_bn_act      y = relu(bn(z)) for the last stage (same prologue).
//   * mlp_bwd_layer   g = dy . [y > 0],  dz = a (g - mean(g) - xhat mean(g xhat)),  dX = dz W (= dy of the layer below),
//                     dW partial = dz^T act(x) per workgroup, and the (sum g', sum g' xhat') partials of the layer BELOW
//                     while dX is still in registers; d(gamma), d(beta) by workgroup 0.
//   * mlp_bn_bwd_reduce / mlp_dw_reduce: the first reduction of a block's chain and the final fixed-order sum of the
//                     per-workgroup dW partials of up to three layers.
// All tiles: 64 rows x <= 64 channels, v_mfma_f32_32x32x2_f32, operands in LDS with a 65-float row stride.  HBM-bound: a
// layer reads dy, z, x and writes dx once (the unfused chain made ~12 passes).  Channel widths <= 64 (every PointNet of the path).
#include "common.h"

#define MT_S 65                 // LDS row stride of 64-wide tiles
#define MT_MAXBLK 256           // workgroups (= partial records) per layer launch: one per CU

__device__ __forceinline__ float mt_bn(float z, float mean, float a, float b) { return __builtin_fmaf(z - mean, a, b); }
// element at a 32-bit BYTE offset from a uniform base pointer: one SGPR pair + one offset register per access
__device__ __forceinline__ float mt_ld(const float* base, unsigned boff) { return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + boff); }
__device__ __forceinline__ void mt_st(float* base, unsigned boff, float v) { *reinterpret_cast<float*>(reinterpret_cast<char*>(base) + boff) = v; }

// acc[32x32] = A[32 rows][K] . B[32 rows][K]^T, both row-major with stride MT_S; K even
__device__ __forceinline__ f32x16 mt_tile_nt(const float* A, const float* B, int K, int lane) {
  const int r = lane & 31, h = lane >> 5;
  f32x16 acc = {0};
  const float* ap = A + r * MT_S + h;
  const float* bp = B + r * MT_S + h;
  for (int k = 0; k < K; k += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[k], bp[k], acc, 0, 0, 0);
  return acc;
}
// acc[32x32] += A^T . B with A[K rows][32 cols at A], B[K rows][32 cols at B] (K = 64 rows of a tile)
__device__ __forceinline__ f32x16 mt_tile_tn(const float* A, const float* B, f32x16 acc, int lane) {
  const int r = lane & 31, h = lane >> 5;
  const float* ap = A + h * MT_S + r;
  const float* bp = B + h * MT_S + r;
#pragma unroll 8
  for (int k = 0; k < 64; k += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[k * MT_S], bp[k * MT_S], acc, 0, 0, 0);
  return acc;
}

// Finalize batch statistics from per-workgroup partials part[nblk][2][64] (sum, sum of squares; fp64), every thread of every
// workgroup taking part in a fixed order.  Leaves mean / a (= gamma invstd) / b (= beta) / invstd of channel c in sm[0..3][c].
// Workgroup 0 also stores them (state[4][C]) and updates the running statistics like torch (momentum, unbiased variance).
// fixed-order sum of the NQ row-group partials of column c
template <int NQ>
__device__ __forceinline__ double mt_red_sum(double (*red)[2][64], int k, int c) {
  double s = red[0][k][c];
#pragma unroll
  for (int q = 1; q < NQ; ++q) s += red[q][k][c];
  return s;
}

// sums of the partial records part[nblk][2][64] per column, all NQ * 64 threads taking part (fixed order) -> red[.][k][c]
template <int NQ>
__device__ __forceinline__ void mt_gather_partials(const double* __restrict__ part, int nblk, int C, double (*red)[2][64]) {
  const int tid = threadIdx.x, c = tid & 63, q = tid >> 6;
  double s1 = 0.0, s2 = 0.0;
  if (c < C) {
    for (int b0 = q; b0 < nblk; b0 += 8 * NQ) {           // 16 loads in flight per thread
      double v1[8], v2[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int b = b0 + u * NQ;
        // (clamped index, load, then select: a predicate around the load itself costs a branch and a full wait per load)
        const int bc = b < nblk ? b : nblk - 1;
        const double w1 = part[((long)bc * 2 + 0) * 64 + c], w2 = part[((long)bc * 2 + 1) * 64 + c];
        v1[u] = b < nblk ? w1 : 0.0;
        v2[u] = b < nblk ? w2 : 0.0;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) { s1 += v1[u]; s2 += v2[u]; }
    }
  }
  red[q][0][c] = s1;
  red[q][1][c] = s2;
  __syncthreads();
}

template <int NQ>
__device__ __forceinline__ void mt_finalize_stats(const double* __restrict__ part, int nblk, int C, long N, const float* gamma,
                                                  const float* beta, float eps, float* rmean, float* rvar, float momentum,
                                                  float* state, float (*sm)[64], double (*red)[2][64]) {
  const int tid = threadIdx.x, c = tid & 63, q = tid >> 6;
  mt_gather_partials<NQ>(part, nblk, C, red);
  if (q == 0) {
    float mean = 0.f, a = 0.f, b = 0.f, invstd = 0.f;
    if (c < C) {
      const double S1 = mt_red_sum<NQ>(red, 0, c);
      const double S2 = mt_red_sum<NQ>(red, 1, c);
      const double m = S1 / (double)N;
      double var = S2 / (double)N - m * m;
      var = var > 0.0 ? var : 0.0;
      invstd = (float)(1.0 / sqrt(var + (double)eps));
      mean = (float)m;
      a = gamma[c] * invstd;
      b = beta[c];
      if (blockIdx.x == 0) {
        if (state) { state[c] = mean; state[C + c] = invstd; state[2 * C + c] = a; state[3 * C + c] = b; }
        if (rmean) {
          rmean[c] = (1.f - momentum) * rmean[c] + momentum * mean;
          const double unbiased = N > 1 ? var * (double)N / (double)(N - 1) : var;
          rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)unbiased;
        }
      }
    }
    sm[0][c] = mean; sm[1][c] = a; sm[2][c] = b; sm[3][c] = invstd;
  }
  __syncthreads();
}

// per-workgroup column sums (two quantities, fp64) -> part[blockIdx.x][2][64]; every lane contributes (v1, v2) for column `col`,
// `slot` in 0..NQ-1 identifies which of the NQ lanes/waves sharing a column it is
template <int NQ>
__device__ __forceinline__ void mt_store_partials(double v1, double v2, int col, int slot, double* part, double (*red)[2][64]) {
  __syncthreads();
  red[slot][0][col] = v1;
  red[slot][1][col] = v2;
  __syncthreads();
  const int tid = threadIdx.x;
  if (tid < 128) {
    const int c = tid & 63, k = tid >> 6;
    part[((long)blockIdx.x * 2 + k) * 64 + c] = mt_red_sum<NQ>(red, k, c);
  }
}

struct MlpFwdP {
  const float* X; long ldx; long rows; int Cin;
  const double* in_part; int in_nblk; const float* in_gamma; const float* in_beta; float in_eps;   // in_part == null: identity input
  float* in_rmean; float* in_rvar; float in_momentum; float* in_state;
  const float* W; const float* bias; int Cout;
  float* Z; long ldz;
  double* out_part;
  long rows_per_wg;
};

// 1024 threads = 4 groups of 4 waves; every group owns one 64-row tile of a 256-row round, so a workgroup keeps four tiles'
// loads in flight (one tile at a time left the kernel waiting on one memory round trip per tile).
// FULL: see mlp_bwd_layer_kernel -- complete workgroups, uniform base pointers + 32-bit byte offsets, no row tests
#define MTF_NG 4
template <bool FULL>
__global__ __launch_bounds__(1024) void mlp_fwd_layer_kernel(MlpFwdP p) {
  __shared__ float Ws[64 * MT_S], Xs[MTF_NG][64 * MT_S], Bs[64], sm[4][64];
  __shared__ double red[16][2][64];
  const int tid = threadIdx.x, grp = tid >> 8, t = tid & 255, lane = tid & 63, wave = t >> 6;
  const bool act = p.in_part != nullptr;
  const int K = (p.Cin + 1) & ~1;
  const int rt = wave & 1, ct = wave >> 1, col = ct * 32 + (lane & 31);
  const long rbeg = (long)blockIdx.x * p.rows_per_wg, rend = min(p.rows, rbeg + p.rows_per_wg);
  const int xk = t & 63, xr = t >> 6;
  float* const Xg = Xs[grp];
  float xv[16];
// (clamped addresses, all 16 loads issued, THEN the selects: written as `ok ? X[..] : 0` -- or even as load + select, which the
// compiler turns back into a branch around the load -- every load got its own s_cbranch_execz and s_waitcnt vmcnt(0), 16
// dependent round trips per tile; MT_PIN keeps the loaded value live outside the select so the load stays unconditional)
#define MT_PIN(v) asm volatile("" : "+v"(v))
#define MTF_FETCH(r0_)                                                                              \
  if (FULL) {                                                                                       \
    const unsigned lr_ = (unsigned)((r0_) - rbeg) + xr;                                             \
    _Pragma("unroll") for (int j = 0; j < 16; ++j) xv[j] = mt_ld(Xb, box + (lr_ + 4 * j) * sx);     \
  } else                                                                                            \
  _Pragma("unroll") for (int j = 0; j < 16; ++j) {                                                  \
    const long rr_ = (r0_) + xr + 4 * j;                                                            \
    xv[j] = p.X[(rr_ < rend ? rr_ : rend - 1) * p.ldx + xkc];                                       \
  }                                                                                                 \
  _Pragma("unroll") for (int j = 0; j < 16; ++j) {                                                  \
    const long rr_ = (r0_) + xr + 4 * j;                                                            \
    MT_PIN(xv[j]);                                                                                  \
    xv[j] = ((FULL || rr_ < rend) && xk < p.Cin) ? xv[j] : 0.f;                                     \
  }
  const int xkc = xk < p.Cin ? xk : p.Cin - 1;
  const float* __restrict__ Xb = p.X + rbeg * p.ldx;
  float* __restrict__ Zb = p.Z + rbeg * p.ldz;
  const unsigned sx = 4u * (unsigned)p.ldx, box = 4u * (unsigned)xkc, sz = 4u * (unsigned)p.ldz;
  if (rbeg < rend) { MTF_FETCH(rbeg + 64 * grp) }          // the first tile's loads fly while the statistics are finalized
  // (behind the first tile's loads, so that both are in flight together)
  // weights (zero padded) -> LDS: four elements per thread, every load unconditional from a clamped index and all in flight
  // together, the padding applied to the values (written as `(n < Cout && k < Cin) ? W[..] : 0` each element was a branch around
  // its load with a full wait: four consecutive memory round trips -- eight in the backward kernel -- before anything else)
  {
    float wv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = tid + 1024 * u, n = i >> 6, k = i & 63;
      wv[u] = p.W[min(n, p.Cout - 1) * p.Cin + min(k, p.Cin - 1)];
    }
    float bvv = 0.f;
    if (p.bias) bvv = p.bias[min(tid & 63, p.Cout - 1)];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = tid + 1024 * u, n = i >> 6, k = i & 63;
      asm volatile("" : "+v"(wv[u]));
      Ws[n * MT_S + k] = (n < p.Cout && k < p.Cin) ? wv[u] : 0.f;
    }
    asm volatile("" : "+v"(bvv));
    if (tid < 64) Bs[tid] = tid < p.Cout ? bvv : 0.f;
  }
  if (act) mt_finalize_stats<16>(p.in_part, p.in_nblk, p.Cin, p.rows, p.in_gamma, p.in_beta, p.in_eps, p.in_rmean, p.in_rvar,
                                 p.in_momentum, p.in_state, sm, red);
  else __syncthreads();
  const float mu = act && xk < p.Cin ? sm[0][xk] : 0.f, aa = act && xk < p.Cin ? sm[1][xk] : 1.f, bb = act && xk < p.Cin ? sm[2][xk] : 0.f;
  double s1 = 0.0, s2 = 0.0;
  const float bv = Bs[col];
  for (long rr0 = rbeg; rr0 < rend; rr0 += 64 * MTF_NG) {
    const long r0 = rr0 + 64 * grp;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const long rr = r0 + xr + 4 * j;
      float v = xv[j];
      if (act) v = ((FULL || rr < rend) && xk < p.Cin) ? fmaxf(mt_bn(v, mu, aa, bb), 0.f) : 0.f;
      Xg[(xr + 4 * j) * MT_S + xk] = v;
    }
    __syncthreads();
    if (rr0 + 64 * MTF_NG < rend) { MTF_FETCH(r0 + 64 * MTF_NG) }
    if (ct * 32 < p.Cout && r0 < rend) {                  // (a wave whose 32 output channels are all padding has nothing to do)
      f32x16 acc = mt_tile_nt(Xg + rt * 32 * MT_S, Ws + ct * 32 * MT_S, K, lane);
      if (FULL) {
        if (col < p.Cout) {
          const unsigned ob = 4u * (unsigned)col + (unsigned)((int)(r0 - rbeg) + rt * 32 + 4 * (lane >> 5)) * sz;
#pragma unroll
          for (int reg = 0; reg < 16; ++reg) acc[reg] += bv;
#pragma unroll
          for (int reg = 0; reg < 16; ++reg) mt_st(Zb, ob + (unsigned)((reg & 3) + 8 * (reg >> 2)) * sz, acc[reg]);
#pragma unroll
          for (int reg = 0; reg < 16; ++reg) {
            s1 += (double)acc[reg];
            s2 += (double)acc[reg] * (double)acc[reg];
          }
        }
      } else {
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const long row = r0 + rt * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
        if (row < rend && col < p.Cout) {
          const float z = acc[reg] + bv;
          p.Z[row * p.ldz + col] = z;
          s1 += (double)z;
          s2 += (double)z * (double)z;
        }
      }
      }
    }
  }
  mt_store_partials<16>(s1, s2, col, grp * 4 + rt * 2 + (lane >> 5), p.out_part, red);
}

struct MlpActP {
  const float* Z; long ldz; long rows; int C;
  const double* part; int nblk; const float* gamma; const float* beta; float eps;
  float* rmean; float* rvar; float momentum; float* state;
  float* Y; long ldy;
};

// lane = column, 16 row groups per workgroup, rows_per_wg rows per workgroup (the layer kernels' partition); 32-bit byte offsets from
// the workgroup's first row.  (r04 form: a flat element index with a 64-bit division per element -- ~100 instructions each.)
__global__ __launch_bounds__(1024) void mlp_bn_act_kernel(MlpActP p, long rows_per_wg) {
  __shared__ float sm[4][64];
  __shared__ double red[16][2][64];
  mt_finalize_stats<16>(p.part, p.nblk, p.C, p.rows, p.gamma, p.beta, p.eps, p.rmean, p.rvar, p.momentum, p.state, sm, red);
  const int tid = threadIdx.x, c = tid & 63, q = tid >> 6;
  const long rbeg = (long)blockIdx.x * rows_per_wg, rend = min(p.rows, rbeg + rows_per_wg);
  if (c >= p.C || rbeg >= rend) return;
  const float mu = sm[0][c], a = sm[1][c], b = sm[2][c];
  const int n = (int)(rend - rbeg);
  const float* __restrict__ Zb = p.Z + rbeg * p.ldz;
  float* __restrict__ Yb = p.Y + rbeg * p.ldy;
  const unsigned sz = 4u * (unsigned)p.ldz, sy = 4u * (unsigned)p.ldy, bo = 4u * (unsigned)c;
  for (int r = q; r < n; r += 16 * 16) {                  // 16 rows per thread and round, all loads in flight
    float zv[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) zv[u] = mt_ld(Zb, bo + (unsigned)min(r + 16 * u, n - 1) * sz);
#pragma unroll
    for (int u = 0; u < 16; ++u)
      if (r + 16 * u < n) mt_st(Yb, bo + (unsigned)(r + 16 * u) * sy, fmaxf(mt_bn(zv[u], mu, a, b), 0.f));
  }
}

// (sum g, sum g xhat) per workgroup for the LAST stage of a block: g = dy . [bn(z) > 0]
struct MlpRedP {
  const float* dY; long lddy; const float* Z; long ldz; long rows; int C; const float* state; double* part; long rows_per_wg;
};

__global__ __launch_bounds__(1024) void mlp_bn_bwd_reduce_kernel(MlpRedP p) {
  __shared__ double red[16][2][64];
  const int tid = threadIdx.x, c = tid & 63, q = tid >> 6;
  const long rbeg = (long)blockIdx.x * p.rows_per_wg, rend = min(p.rows, rbeg + p.rows_per_wg);
  double s1 = 0.0, s2 = 0.0;
  if (c < p.C && rbeg < rend) {
    const float mu = p.state[c], is = p.state[p.C + c], a = p.state[2 * p.C + c], b = p.state[3 * p.C + c];
    const int n = (int)(rend - rbeg);
    const float* __restrict__ Zb = p.Z + rbeg * p.ldz;
    const float* __restrict__ Gb = p.dY + rbeg * p.lddy;
    const unsigned sz = 4u * (unsigned)p.ldz, sg = 4u * (unsigned)p.lddy, bo = 4u * (unsigned)c;
    for (int r = q; r < n; r += 128) {                    // 8 rows per round: 16 loads in flight per thread; 32-bit byte offsets
      float zz[8], gg[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int rc = min(r + 16 * u, n - 1);             // (clamped: unconditional loads, then select)
        const float zl = mt_ld(Zb, bo + (unsigned)rc * sz), gl = mt_ld(Gb, bo + (unsigned)rc * sg);
        zz[u] = r + 16 * u < n ? zl : 0.f;
        gg[u] = r + 16 * u < n ? gl : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const float g = mt_bn(zz[u], mu, a, b) > 0.f ? gg[u] : 0.f;
        s1 += (double)g;
        s2 += (double)g * (double)((zz[u] - mu) * is);
      }
    }
  }
  mt_store_partials<16>(s1, s2, c, q, p.part, red);
}

struct MlpBwdP {
  const float* dY; long lddy; const float* Z; long ldz; long rows; int Cout;
  const float* state;                 // this layer's BatchNorm: [4][Cout] mean, invstd, a, b
  const double* g_part; int g_nblk;   // partials (sum g, sum g xhat) of this layer
  float* dgamma; float* dbeta;
  const float* Xin; long ldxin; int Cin; const float* in_state;    // in_state != null: Xin = pre-BN z of the layer below
  const float* W;                     // [Cout][Cin]
  float* dX; long lddx;               // null: no input gradient wanted
  double* gprev_part;                 // partials of the layer below (needs dX and in_state)
  float* dW_part;                     // [gridDim.x][64 * 64]
  long rows_per_wg;
  // gather form of Xin (LocalPointNet's first layer, Upper_Net.py:100-119): row (f, slot) = cat(anchor, xyz - anchor, features) of point
  // gidx[row] of frame f, read from the per-point feature rows -- the gathered tensor is not kept in memory
  const long long* gidx; const float* gfeats; long gldf; const float* ganch; int gN;
};

// 512 threads = 2 groups of 4 waves, each group one 64-row tile of a 128-row round (three tiles of LDS per group)
// FULL: every workgroup of the launch owns rows_per_wg complete rows (rows % rows_per_wg == 0: every shape of the training step).  In-kernel
// stamps (scripts/mlp_bwd_probe.hip) put the two 128-row rounds of a workgroup at ~15 k cycles EACH whatever the channel count -- 8 or
// 64 -- with ~1400 instructions per wave and round, a third of them 64-bit row x stride address arithmetic, row clamps and per-element
// predicates: the kernel is bound by instruction issue (two waves per SIMD), not by memory or the matrix pipe (regrouping its loads,
// the gather rounds or the LDS operand reads changed nothing).  The FULL form addresses everything through uniform base pointers + 32-bit
// byte offsets and drops every row test.
#define MTB_NG 2
template <bool GATHER, bool FULL>
__global__ __launch_bounds__(512) void mlp_bwd_layer_kernel(MlpBwdP p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, grp = tid >> 8, t = tid & 255, lane = tid & 63, wave = t >> 6;
  float* DZs = smem + grp * 3 * 64 * MT_S;   // [64 rows][Cout]
  float* Xs = DZs + 64 * MT_S;               // [64 rows][Cin]  activated layer input
  float* Zp = Xs + 64 * MT_S;                // [64 rows][Cin]  raw z of the layer below
  float* Wt = smem + MTB_NG * 3 * 64 * MT_S; // [Cin][Cout]     (W transposed: B operand of dX = dz W)
  float (*st)[64] = reinterpret_cast<float (*)[64]>(Wt + 64 * MT_S);      // [8][64]: this layer mean, invstd, a, b; below: mean, invstd, a, b
  float (*cc)[64] = st + 8;                  // [2][64]: c1 = sum g / N, c2 = sum g xhat / N
  double (*red)[2][64] = reinterpret_cast<double (*)[2][64]>(cc + 2);     // [8][2][64]
  const long rbeg = (long)blockIdx.x * p.rows_per_wg, rend = min(p.rows, rbeg + p.rows_per_wg);
  const int xk = t & 63, xr = t >> 6;
  float gy[16], gz[16], gx[16];
#define MTB_FETCH(r0_)                                                                              \
  if (GATHER) {                                        /* the layer input gathered through the group indices */ \
    long long gq_[16];                                                                              \
    _Pragma("unroll") for (int j = 0; j < 16; ++j) {                                                \
      const long rr_ = (r0_) + xr + 4 * j;                                                          \
      const long rc_ = rr_ < rend ? rr_ : rend - 1;                                                 \
      gq_[j] = p.gidx[rc_];                                                                         \
      gy[j] = p.dY[rc_ * p.lddy + xko]; gz[j] = p.Z[rc_ * p.ldz + xko];                             \
    }                                                                                               \
    _Pragma("unroll") for (int j = 0; j < 16; ++j) {                                                \
      const long rr_ = (r0_) + xr + 4 * j;                                                          \
      const long rc_ = rr_ < rend ? rr_ : rend - 1;                                                 \
      const long fr_ = rc_ / 216;                                                                   \
      const int an_ = (int)(rc_ - fr_ * 216) >> 3;                                                  \
      const float pv_ = p.gfeats[(fr_ * p.gN + gq_[j]) * p.gldf + (xki >= 3 ? xki - 3 : 0)];        \
      const float av_ = p.ganch[an_ * 3 + (xki < 3 ? xki : (xki < 6 ? xki - 3 : 0))];               \
      gx[j] = xki < 3 ? av_ : (xki < 6 ? pv_ - av_ : pv_);                                          \
    }                                                                                               \
  } else if (FULL) {                                                                                \
    const unsigned lr_ = (unsigned)((r0_) - rbeg) + xr;                                             \
    _Pragma("unroll") for (int j = 0; j < 16; ++j) {                                                \
      gy[j] = mt_ld(dYb, boy + (lr_ + 4 * j) * sy); gz[j] = mt_ld(Zb, boz + (lr_ + 4 * j) * sz); gx[j] = mt_ld(Xb, box + (lr_ + 4 * j) * sx); \
    }                                                                                               \
  } else                                                                                            \
  _Pragma("unroll") for (int j = 0; j < 16; ++j) {                                                  \
    const long rr_ = (r0_) + xr + 4 * j;                                                            \
    const long rc_ = rr_ < rend ? rr_ : rend - 1;      /* clamped row / columns: unconditional loads (see MTF_FETCH) */ \
    gy[j] = p.dY[rc_ * p.lddy + xko]; gz[j] = p.Z[rc_ * p.ldz + xko]; gx[j] = p.Xin[rc_ * p.ldxin + xki]; \
  }                                                                                                 \
  _Pragma("unroll") for (int j = 0; j < 16; ++j) {                                                  \
    const long rr_ = (r0_) + xr + 4 * j;                                                            \
    const bool oko_ = (FULL || rr_ < rend) && xk < p.Cout;                                          \
    MT_PIN(gy[j]); MT_PIN(gz[j]); MT_PIN(gx[j]);                                                    \
    gy[j] = oko_ ? gy[j] : 0.f;                                                                     \
    gz[j] = oko_ ? gz[j] : 0.f;                                                                     \
    gx[j] = ((FULL || rr_ < rend) && xk < p.Cin) ? gx[j] : 0.f;                                     \
  }
  const int xko = xk < p.Cout ? xk : p.Cout - 1, xki = xk < p.Cin ? xk : p.Cin - 1;
  // FULL: uniform bases at the workgroup's first row, 32-bit byte offsets (column part + row x stride)
  const float* __restrict__ dYb = p.dY + rbeg * p.lddy;
  const float* __restrict__ Zb = p.Z + rbeg * p.ldz;
  const float* __restrict__ Xb = GATHER ? p.Z : p.Xin + rbeg * p.ldxin;
  float* __restrict__ dXb = p.dX ? p.dX + rbeg * p.lddx : nullptr;
  const unsigned sy = 4u * (unsigned)p.lddy, sz = 4u * (unsigned)p.ldz, sx = 4u * (unsigned)p.ldxin, sdx = 4u * (unsigned)p.lddx;
  const unsigned boy = 4u * (unsigned)xko, boz = boy, box = 4u * (unsigned)xki;
  MMEGO_STAMP_AT(blockIdx.x, 0, tid == 0);
  if (rbeg < rend) { MTB_FETCH(rbeg + 64 * grp) }          // the first tile's loads fly while the partial sums are gathered
  {  // W^T and the two BatchNorm states -> LDS: unconditional loads from clamped indices, all in flight, padding applied to the values
    float wv[8], sv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = tid + 512 * u, k = i >> 6, n = i & 63;        // k = cout, n = cin
      wv[u] = p.W[min(k, p.Cout - 1) * p.Cin + min(n, p.Cin - 1)];
    }
    const int co = min(tid & 63, p.Cout - 1), ci = min(tid & 63, p.Cin - 1);
    const float* ins = p.in_state ? p.in_state : p.state;          // (a valid address either way; unused without in_state)
    const int cis = p.in_state ? ci : co, Cs = p.in_state ? p.Cin : p.Cout;
#pragma unroll
    for (int q = 0; q < 4; ++q) { sv[q] = p.state[q * p.Cout + co]; sv[4 + q] = ins[q * Cs + cis]; }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = tid + 512 * u, k = i >> 6, n = i & 63;
      asm volatile("" : "+v"(wv[u]));
      Wt[n * MT_S + k] = (k < p.Cout && n < p.Cin) ? wv[u] : 0.f;
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) asm volatile("" : "+v"(sv[q]));
    if (tid < 64) {
      const bool ok = tid < p.Cout, oki = p.in_state && tid < p.Cin;
#pragma unroll
      for (int q = 0; q < 4; ++q) st[q][tid] = ok ? sv[q] : 0.f;
      st[4][tid] = oki ? sv[4] : 0.f;
      st[5][tid] = oki ? sv[5] : 0.f;
      st[6][tid] = oki ? sv[6] : 1.f;
      st[7][tid] = oki ? sv[7] : 0.f;
    }
  }
  {  // finalize this layer's (sum g, sum g xhat): c1, c2; d(gamma), d(beta) by workgroup 0
    mt_gather_partials<8>(p.g_part, p.g_nblk, p.Cout, red);
    const int c = tid & 63;
    if (tid < 64) {
      const double S1 = mt_red_sum<8>(red, 0, c), S2 = mt_red_sum<8>(red, 1, c);
      cc[0][c] = c < p.Cout ? (float)(S1 / (double)p.rows) : 0.f;
      cc[1][c] = c < p.Cout ? (float)(S2 / (double)p.rows) : 0.f;
      if (blockIdx.x == 0 && c < p.Cout) { p.dbeta[c] = (float)S1; p.dgamma[c] = (float)S2; }
    }
    __syncthreads();
  }
  const int Ko = (p.Cout + 1) & ~1;
  const int rt = wave & 1, ct = wave >> 1, col = ct * 32 + (lane & 31);
  const bool act = p.in_state != nullptr;
  const bool want_dx = p.dX != nullptr, want_prev = p.gprev_part != nullptr;
  f32x16 accw = {0};
  double s1 = 0.0, s2 = 0.0;
  MMEGO_STAMP_AT(blockIdx.x, 1, tid == 0);
  for (long rr0 = rbeg; rr0 < rend; rr0 += 64 * MTB_NG) {
    const long r0 = rr0 + 64 * grp;
    __syncthreads();
    {
      const float mu = st[0][xk], is = st[1][xk], a = st[2][xk], b = st[3][xk], c1 = cc[0][xk], c2 = cc[1][xk];
      const float mui = st[4][xk], ai = st[6][xk], bi = st[7][xk];
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const long rr = r0 + xr + 4 * j;
        const bool oko = (FULL || rr < rend) && xk < p.Cout, oki = (FULL || rr < rend) && xk < p.Cin;
        const float z = gz[j];
        const float g = mt_bn(z, mu, a, b) > 0.f ? gy[j] : 0.f;
        DZs[(xr + 4 * j) * MT_S + xk] = oko ? a * (g - c1 - ((z - mu) * is) * c2) : 0.f;
        const float xin = gx[j];
        Zp[(xr + 4 * j) * MT_S + xk] = xin;
        Xs[(xr + 4 * j) * MT_S + xk] = oki ? (act ? fmaxf(mt_bn(xin, mui, ai, bi), 0.f) : xin) : 0.f;
      }
    }
    __syncthreads();
    if (rr0 + 64 * MTB_NG < rend) { MTB_FETCH(r0 + 64 * MTB_NG) }
    if (r0 >= rend) continue;                              // (no barrier below this point of the loop body)
    if (want_dx && ct * 32 < p.Cin) {                     // dX tile: rows rt, input channels ct
      const f32x16 acc = mt_tile_nt(DZs + rt * 32 * MT_S, Wt + ct * 32 * MT_S, Ko, lane);
      const float mui = st[4][col], isi = st[5][col];
      if (FULL) {
        if (col < p.Cin) {                                // (one test per wave half; the stores read the accumulator registers as they are)
          const unsigned ob = 4u * (unsigned)col + (unsigned)((int)(r0 - rbeg) + rt * 32 + 4 * (lane >> 5)) * sdx;
#pragma unroll
          for (int reg = 0; reg < 16; ++reg) mt_st(dXb, ob + (unsigned)((reg & 3) + 8 * (reg >> 2)) * sdx, acc[reg]);
          if (want_prev) {
            float xa[16], za[16];
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
              const int lr = rt * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
              xa[reg] = Xs[lr * MT_S + col];
              za[reg] = Zp[lr * MT_S + col];
            }
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
              const float g = xa[reg] > 0.f ? acc[reg] : 0.f;
              s1 += (double)g;
              s2 += (double)g * (double)((za[reg] - mui) * isi);
            }
          }
        }
      } else {
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int lr = rt * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
        const long row = r0 + lr;
        if (row < rend && col < p.Cin) {
          const float dx = acc[reg];
          p.dX[row * p.lddx + col] = dx;
          if (want_prev) {
            const float g = Xs[lr * MT_S + col] > 0.f ? dx : 0.f;
            s1 += (double)g;
            s2 += (double)g * (double)((Zp[lr * MT_S + col] - mui) * isi);
          }
        }
      }
      }
    }
    // dW[cout tile rt][cin tile ct] += dz^T . act(x) over the tile's 64 rows
    if (rt * 32 < p.Cout && ct * 32 < p.Cin) accw = mt_tile_tn(DZs + rt * 32, Xs + ct * 32, accw, lane);
  }
  // the two groups' dW accumulators, added in a fixed order (group 1's go through LDS)
  __syncthreads();
  MMEGO_STAMP_AT(blockIdx.x, 2, tid == 0);
  float* const xch = smem;                   // [64][64] exchange area (the tiles are no longer needed)
  if (grp == 1) {
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int m = rt * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
      xch[m * 64 + col] = accw[reg];
    }
  }
  __syncthreads();
  if (grp == 0) {
    float* out = p.dW_part + (long)blockIdx.x * 64 * 64;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int m = rt * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
      out[m * 64 + col] = accw[reg] + xch[m * 64 + col];
    }
  }
  if (want_prev) mt_store_partials<8>(s1, s2, col, grp * 4 + rt * 2 + (lane >> 5), p.gprev_part, red);
  MMEGO_STAMP_AT(blockIdx.x, 3, tid == 0);
}

// dW[l][cout][cin] = sum over the workgroups' partials, fixed order, fp64 accumulation; up to 9 layers (three chains) per launch
#define MT_DW_MAX 9
struct MlpDwP { const float* part[MT_DW_MAX]; float* dW[MT_DW_MAX]; int Cout[MT_DW_MAX], Cin[MT_DW_MAX], nblk[MT_DW_MAX]; long stride[MT_DW_MAX]; int nlayers; };

__global__ __launch_bounds__(256) void mlp_dw_reduce_kernel(MlpDwP p) {
  // 16 elements per workgroup, 16 lanes per element: lane s sums partials s, s + 16, ... (all loads in flight), then the 16
  // slice sums are added in a fixed order
  __shared__ double sl[16][17];
  const int l = blockIdx.y;
  if (l >= p.nlayers) return;
  const int Cout = p.Cout[l], Cin = p.Cin[l], nblk = p.nblk[l];
  const long pstride = p.stride[l];                       // floats between two workgroups' partials (64 x 64 for a dW tile)
  const int e = threadIdx.x >> 4, s_ = threadIdx.x & 15;
  for (int i0 = blockIdx.x * 16; i0 < Cout * Cin; i0 += gridDim.x * 16) {
    const int i = i0 + e;
    double s = 0.0;
    if (i < Cout * Cin) {
      const int m = i / Cin, n = i - m * Cin;
      const float* src = p.part[l] + m * 64 + n;
      float v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int b = s_ + 16 * u;
        const float vl = src[(long)(b < nblk ? b : nblk - 1) * pstride];      // (clamped: load, then select)
        v[u] = b < nblk ? vl : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 16; ++u) s += (double)v[u];
    }
    __syncthreads();
    sl[e][s_] = s;
    __syncthreads();
    if (s_ == 0 && i < Cout * Cin) {
      double t = 0.0;
#pragma unroll
      for (int u = 0; u < 16; ++u) t += sl[e][u];
      p.dW[l][i] = (float)t;
    }
  }
}

static inline void mt_grid(long rows, int* nblk, long* rpw) {
  const long tiles = (rows + 255) / 256;                   // rounds of 256 rows (4 x 64-row tiles in flight per workgroup)
  long n = tiles < MT_MAXBLK ? tiles : MT_MAXBLK;
  const long tpw = (tiles + n - 1) / n;
  n = (tiles + tpw - 1) / tpw;
  *nblk = (int)n;
  *rpw = tpw * 256;
}

// Number of workgroups (= partial records) a layer launch over `rows` rows uses: partial buffers are nblk x 2 x 64 doubles
// (statistics) and nblk x 4096 floats (dW).
extern "C" int mmego_mlp_train_nblk(long rows) {
  int nblk; long rpw;
  mt_grid(rows > 0 ? rows : 1, &nblk, &rpw);
  return nblk;
}

extern "C" int mmego_mlp_fwd_layer(void* stream, const float* X, long ldx, long rows, int Cin, const double* in_part,
                                   const float* in_gamma, const float* in_beta, double in_eps, float* in_rmean, float* in_rvar,
                                   double in_momentum, float* in_state, const float* W, const float* bias, int Cout, float* Z,
                                   long ldz, double* out_part) {
  MMEGO_REQUIRE(X && W && Z && out_part && rows > 0 && Cin >= 1 && Cin <= 64 && Cout >= 1 && Cout <= 64 && ldx >= Cin && ldz >= Cout);
  MMEGO_REQUIRE(!in_part || (in_gamma && in_beta));
  MlpFwdP p;
  int nblk;
  mt_grid(rows, &nblk, &p.rows_per_wg);
  p.X = X; p.ldx = ldx; p.rows = rows; p.Cin = Cin;
  p.in_part = in_part; p.in_nblk = nblk; p.in_gamma = in_gamma; p.in_beta = in_beta; p.in_eps = (float)in_eps;
  p.in_rmean = in_rmean; p.in_rvar = in_rvar; p.in_momentum = (float)in_momentum; p.in_state = in_state;
  p.W = W; p.bias = bias; p.Cout = Cout; p.Z = Z; p.ldz = ldz; p.out_part = out_part;
  if (p.rows % p.rows_per_wg == 0 && p.rows_per_wg * 4 * (p.ldx > p.ldz ? p.ldx : p.ldz) < (1L << 31))
    hipLaunchKernelGGL(mlp_fwd_layer_kernel<true>, dim3(nblk), dim3(1024), 0, (hipStream_t)stream, p);
  else
    hipLaunchKernelGGL(mlp_fwd_layer_kernel<false>, dim3(nblk), dim3(1024), 0, (hipStream_t)stream, p);
  MMEGO_LAUNCH_CHECK();
  return MMEGO_OK;
}

// The same with the number of input partial records given (a producer other than mlp_fwd_layer -- mmego_local_group_l1 -- chooses
// its own workgroup count).
extern "C" int mmego_mlp_fwd_layer_n(void* stream, const float* X, long ldx, long rows, int Cin, const double* in_part, int in_nblk,
                                     const float* in_gamma, const float* in_beta, double in_eps, float* in_rmean, float* in_rvar,
                                     double in_momentum, float* in_state, const float* W, const float* bias, int Cout, float* Z,
                                     long ldz, double* out_part) {
  MMEGO_REQUIRE(X && W && Z && out_part && rows > 0 && Cin >= 1 && Cin <= 64 && Cout >= 1 && Cout <= 64 && ldx >= Cin && ldz >= Cout);
  MMEGO_REQUIRE(in_part && in_gamma && in_beta && in_nblk >= 1 && in_nblk <= 1024);
  MlpFwdP p;
  int nblk;
  mt_grid(rows, &nblk, &p.rows_per_wg);
  p.X = X; p.ldx = ldx; p.rows = rows; p.Cin = Cin;
  p.in_part = in_part; p.in_nblk = in_nblk; p.in_gamma = in_gamma; p.in_beta = in_beta; p.in_eps = (float)in_eps;
  p.in_rmean = in_rmean; p.in_rvar = in_rvar; p.in_momentum = (float)in_momentum; p.in_state = in_state;
  p.W = W; p.bias = bias; p.Cout = Cout; p.Z = Z; p.ldz = ldz; p.out_part = out_part;
  if (p.rows % p.rows_per_wg == 0 && p.rows_per_wg * 4 * (p.ldx > p.ldz ? p.ldx : p.ldz) < (1L << 31))
    hipLaunchKernelGGL(mlp_fwd_layer_kernel<true>, dim3(nblk), dim3(1024), 0, (hipStream_t)stream, p);
  else
    hipLaunchKernelGGL(mlp_fwd_layer_kernel<false>, dim3(nblk), dim3(1024), 0, (hipStream_t)stream, p);
  MMEGO_LAUNCH_CHECK();
  return MMEGO_OK;
}

extern "C" int mmego_mlp_bn_act(void* stream, const float* Z, long ldz, long rows, int C, const double* part, const float* gamma,
                                const float* beta, double eps, float* rmean, float* rvar, double momentum, float* state,
                                float* Y, long ldy) {
  MMEGO_REQUIRE(Z && Y && part && gamma && beta && rows > 0 && C >= 1 && C <= 64 && ldz >= C && ldy >= C);
  int nblk; long rpw;
  mt_grid(rows, &nblk, &rpw);
  MlpActP p = {Z, ldz, rows, C, part, nblk, gamma, beta, (float)eps, rmean, rvar, (float)momentum, state, Y, ldy};
  MMEGO_REQUIRE(rpw * 4 * (ldz > ldy ? ldz : ldy) < (1L << 31));
  hipLaunchKernelGGL(mlp_bn_act_kernel, dim3(nblk), dim3(1024), 0, (hipStream_t)stream, p, rpw);
  MMEGO_LAUNCH_CHECK();
  return MMEGO_OK;
}

extern "C" int mmego_mlp_bn_bwd_reduce(void* stream, const float* dY, long lddy, const float* Z, long ldz, long rows, int C,
                                       const float* state, double* part) {
  MMEGO_REQUIRE(dY && Z && state && part && rows > 0 && C >= 1 && C <= 64 && lddy >= C && ldz >= C);
  MlpRedP p;
  int nblk;
  mt_grid(rows, &nblk, &p.rows_per_wg);
  p.dY = dY; p.lddy = lddy; p.Z = Z; p.ldz = ldz; p.rows = rows; p.C = C; p.state = state; p.part = part;
  hipLaunchKernelGGL(mlp_bn_bwd_reduce_kernel, dim3(nblk), dim3(1024), 0, (hipStream_t)stream, p);
  MMEGO_LAUNCH_CHECK();
  return MMEGO_OK;
}

static int mlp_bwd_layer_launch(hipStream_t st, MlpBwdP& p) {
  int nblk;
  mt_grid(p.rows, &nblk, &p.rows_per_wg);
  p.g_nblk = nblk;
  const size_t lds = (size_t)((MTB_NG * 3 + 1) * 64 * MT_S + 10 * 64) * sizeof(float) + sizeof(double) * 8 * 2 * 64;
  static bool attr = false;
  if (!attr) {
    hipError_t e = hipFuncSetAttribute((const void*)mlp_bwd_layer_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)mlp_bwd_layer_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)mlp_bwd_layer_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    attr = true;
  }
  // the fast form: complete workgroups only, 32-bit byte offsets inside a workgroup's rows
  const long ldmax = p.lddy > p.ldz ? (p.lddy > p.ldxin ? p.lddy : p.ldxin) : (p.ldz > p.ldxin ? p.ldz : p.ldxin);
  const bool full = !p.gidx && p.rows % p.rows_per_wg == 0 && p.rows_per_wg * 4 * (ldmax > p.lddx ? ldmax : p.lddx) < (1L << 31);
  if (p.gidx) hipLaunchKernelGGL((mlp_bwd_layer_kernel<true, false>), dim3(nblk), dim3(512), lds, st, p);
  else if (full) hipLaunchKernelGGL((mlp_bwd_layer_kernel<false, true>), dim3(nblk), dim3(512), lds, st, p);
  else hipLaunchKernelGGL((mlp_bwd_layer_kernel<false, false>), dim3(nblk), dim3(512), lds, st, p);
  MMEGO_LAUNCH_CHECK();
  return MMEGO_OK;
}

extern "C" int mmego_mlp_bwd_layer(void* stream, const float* dY, long lddy, const float* Z, long ldz, long rows, int Cout,
                                   const float* state, const double* g_part, float* dgamma, float* dbeta, const float* Xin,
                                   long ldxin, int Cin, const float* in_state, const float* W, float* dX, long lddx,
                                   double* gprev_part, float* dW_part) {
  MMEGO_REQUIRE(dY && Z && state && g_part && dgamma && dbeta && Xin && W && dW_part && rows > 0);
  MMEGO_REQUIRE(Cin >= 1 && Cin <= 64 && Cout >= 1 && Cout <= 64 && lddy >= Cout && ldz >= Cout && ldxin >= Cin);
  MMEGO_REQUIRE(!dX || lddx >= Cin);
  MMEGO_REQUIRE(!gprev_part || (dX && in_state));
  MlpBwdP p;
  p.dY = dY; p.lddy = lddy; p.Z = Z; p.ldz = ldz; p.rows = rows; p.Cout = Cout; p.state = state;
  p.g_part = g_part; p.dgamma = dgamma; p.dbeta = dbeta;
  p.Xin = Xin; p.ldxin = ldxin; p.Cin = Cin; p.in_state = in_state; p.W = W; p.dX = dX; p.lddx = lddx;
  p.gprev_part = gprev_part; p.dW_part = dW_part;
  p.gidx = nullptr; p.gfeats = nullptr; p.gldf = 0; p.ganch = nullptr; p.gN = 0;
  return mlp_bwd_layer_launch((hipStream_t)stream, p);
}

// mmego_mlp_bwd_layer for the FIRST stage of LocalPointNet with its input gathered on the fly: row r = (frame r / 216, slot r % 216) is
// cat(anchors[slot / 8], xyz - anchor, features) of point gidx[r] of that frame, feats [F*N][ldf] = (xyz | D features); Cin = 6 + D.
extern "C" int mmego_mlp_bwd_layer_gather(void* stream, const float* dY, long lddy, const float* Z, long ldz, long rows, int Cout,
                                          const float* state, const double* g_part, float* dgamma, float* dbeta,
                                          const long long* gidx, const float* feats, long ldf, const float* anchors, int N, int D,
                                          const float* W, float* dX, long lddx, float* dW_part) {
  MMEGO_REQUIRE(dY && Z && state && g_part && dgamma && dbeta && gidx && feats && anchors && W && dW_part && rows > 0 && (rows % 216) == 0);
  const int Cin = 6 + D;
  MMEGO_REQUIRE(D >= 0 && Cin <= 64 && Cout >= 1 && Cout <= 64 && lddy >= Cout && ldz >= Cout && ldf >= 3 + D && N > 0);
  MMEGO_REQUIRE(!dX || lddx >= Cin);
  MlpBwdP p;
  p.dY = dY; p.lddy = lddy; p.Z = Z; p.ldz = ldz; p.rows = rows; p.Cout = Cout; p.state = state;
  p.g_part = g_part; p.dgamma = dgamma; p.dbeta = dbeta;
  p.Xin = feats; p.ldxin = 0; p.Cin = Cin; p.in_state = nullptr; p.W = W; p.dX = dX; p.lddx = lddx;
  p.gprev_part = nullptr; p.dW_part = dW_part;
  p.gidx = gidx; p.gfeats = feats; p.gldf = ldf; p.ganch = anchors; p.gN = N;
  return mlp_bwd_layer_launch((hipStream_t)stream, p);
}

extern "C" int mmego_mlp_dw_reduce(void* stream, long rows, int nlayers, const float* part0, float* dW0, int Cout0, int Cin0,
                                   const float* part1, float* dW1, int Cout1, int Cin1, const float* part2, float* dW2, int Cout2,
                                   int Cin2) {
  MMEGO_REQUIRE(nlayers >= 1 && nlayers <= 3 && part0 && dW0 && rows > 0);
  MMEGO_REQUIRE(nlayers < 2 || (part1 && dW1));
  MMEGO_REQUIRE(nlayers < 3 || (part2 && dW2));
  int nblk; long rpw;
  mt_grid(rows, &nblk, &rpw);
  MlpDwP p = {};
  p.part[0] = part0; p.dW[0] = dW0; p.Cout[0] = Cout0; p.Cin[0] = Cin0;
  p.part[1] = part1; p.dW[1] = dW1; p.Cout[1] = Cout1; p.Cin[1] = Cin1;
  p.part[2] = part2; p.dW[2] = dW2; p.Cout[2] = Cout2; p.Cin[2] = Cin2;
  for (int l = 0; l < 3; ++l) {
    p.nblk[l] = nblk;
    p.stride[l] = 64 * 64;
    if (l < nlayers) MMEGO_REQUIRE(p.Cout[l] >= 1 && p.Cout[l] <= 64 && p.Cin[l] >= 1 && p.Cin[l] <= 64);
  }
  p.nlayers = nlayers;
  hipLaunchKernelGGL(mlp_dw_reduce_kernel, dim3(256, nlayers), dim3(256), 0, (hipStream_t)stream, p);
  MMEGO_LAUNCH_CHECK();
  return MMEGO_OK;
}

// The same sum for up to 9 layers of SEVERAL chains in one launch (a net with two or three pointwise-MLP chains had one reduce launch
// per chain at the end of each: the partials stay where they are until the pass ends, one node instead of two or three).
struct MmegoDwRedH { const float* part; float* dW; int Cout, Cin; long rows; int nblk; long stride; };   // host mirror of include/mmego_hip.h's MmegoDwRed
extern "C" int mmego_mlp_dw_reduce_multi(void* stream, int n, const void* descs) {
  MMEGO_REQUIRE(n >= 1 && n <= MT_DW_MAX && descs);
  const MmegoDwRedH* d = static_cast<const MmegoDwRedH*>(descs);
  MlpDwP p = {};
  for (int l = 0; l < n; ++l) {
    MMEGO_REQUIRE(d[l].part && d[l].dW && (d[l].rows > 0 || d[l].nblk > 0) && d[l].Cout >= 1 && d[l].Cout <= 64 && d[l].Cin >= 1 && d[l].Cin <= 64);
    int nblk = d[l].nblk; long rpw;
    if (nblk <= 0) mt_grid(d[l].rows, &nblk, &rpw);
    MMEGO_REQUIRE(nblk <= 256 && d[l].stride >= 0);
    p.part[l] = d[l].part; p.dW[l] = d[l].dW; p.Cout[l] = d[l].Cout; p.Cin[l] = d[l].Cin; p.nblk[l] = nblk;
    p.stride[l] = d[l].stride > 0 ? d[l].stride : 64 * 64;
  }
  p.nlayers = n;
  hipLaunchKernelGGL(mlp_dw_reduce_kernel, dim3(256, n), dim3(256), 0, (hipStream_t)stream, p);
  MMEGO_LAUNCH_CHECK();
  return MMEGO_OK;
}
