// The anchor ("voxel") branch of UpperNetwlocal, MI355X-first (reference Net/Upper_Net.py:10-32 square_distance, :54-72
// point_ball_set, :100-119 AnchorGrouping, :147-177 LocalPointNet with its 8-way softmax pooling, :219-239 LocalModule).
//
//   local_group_l1   a workgroup takes frames in turn: the frame's 128 x 28 feature rows go to LDS once; the 27 x N squared
//                    distances are formed there with the reference's rounding (bit-exact); the 8 nearest points of an anchor are
//                    picked by 8 rounds of a wave-wide minimum (DPP reduction on order-preserving integer codes of the keys, ballot +
//                    find-first for the stable tie rule: lowest point index first) -- 64 keys per instruction instead of a 128-step
//                    serial rank count per (anchor, point); the 27 x 8 gathered rows cat(anchor, xyz - anchor, features) are built in
//                    LDS and go STRAIGHT into LocalPointNet's first k=1 conv on MFMAs: z1 and its BatchNorm partial sums leave,
//                    the int64 group indices leave, and (training only, for the layer's weight gradient) the gathered rows.
//   pool8_bn_act     last stage of the train-mode LocalPointNet: BatchNorm (statistics finalized in the prologue) + ReLU + the
//                    attention score + softmax over a group's 8 members + weighted sum, per 64-row block of one wave; the pooled
//                    vectors are stored directly in the Conv3d input order [frame][channel][anchor] -- the activated 64-channel
//                    tensor (28 MB at the bench shape), the pooling launch's pass over it and the transpose launch are gone.
//   pool8_bwd        backward of that pooling with the activated rows recomputed from the pre-BatchNorm tensor: gradient of the rows,
//                    the BatchNorm sums of the stage (what mlp_bn_bwd_reduce produced in a pass of its own) and the attention
//                    parameter gradients' per-workgroup partials, reading the pooled gradient in the Conv3d order (no transpose).
//   anchor_scatter   gradient of the gathered rows back to the points: the slots of one anchor hit 8 DISTINCT points, so the 27 anchors
//                    are walked in order with 8 x 28 parallel accumulations each (in LDS) -- the same summation order as a serial
//                    walk over the slots, i.e. deterministic, without probing all 216 slots per (point, channel).
#include "common.h"

#define NA 27
#define NS 8
#define NSLOT (NA * NS)          // 216 gathered rows per frame
#define LG_NT 256
#define LG_GS 36                 // LDS row stride of the gathered tile (32 columns used)

__device__ __forceinline__ float sq3_nofma(float x, float y, float z) {
  return __fadd_rn(__fadd_rn(__fmul_rn(x, x), __fmul_rn(y, y)), __fmul_rn(z, z));
}

// order-preserving unsigned code of a float key (-x < +0 < +x < +inf); 0xffffffff is reserved for "taken"
__device__ __forceinline__ unsigned key_code(float d) {
  const unsigned b = __float_as_uint(d);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

// minimum of an unsigned value over the 64 lanes (DPP: within quads, within rows of 16, then across the rows); uniform result
__device__ __forceinline__ unsigned wave_min_u32(unsigned v) {
#define LG_DPP(ctrl, rmask) (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, ctrl, rmask, 0xf, false)
  unsigned t;
  t = LG_DPP(0xB1, 0xf); v = t < v ? t : v;              // quad_perm [1,0,3,2]
  t = LG_DPP(0x4E, 0xf); v = t < v ? t : v;              // quad_perm [2,3,0,1]
  t = LG_DPP(0x124, 0xf); v = t < v ? t : v;             // row_ror:4
  t = LG_DPP(0x128, 0xf); v = t < v ? t : v;             // row_ror:8
  t = LG_DPP(0x142, 0xa); v = t < v ? t : v;             // row_bcast:15 into rows 1 and 3
  t = LG_DPP(0x143, 0xc); v = t < v ? t : v;             // row_bcast:31 into rows 2 and 3
#undef LG_DPP
  return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

struct LocalGroupP {
  const float* feats; long ldf;          // [F*N][ldf]: columns 0:3 xyz, 3:3+D features
  int N, D;
  const float* anchors;                  // [27][3]
  long long* idx;                        // [F][27][8]
  float* grouped;                        // [F*216][6+D] or null
  const float* W1; const float* b1; int C1;       // first k=1 conv of LocalPointNet [C1][6+D] (C1 <= 32) or null: grouping only
  float* Z1; long ldz1; double* part1;   // z1 [F*216][C1], BatchNorm partial sums part1[gridDim.x][2][64] (sum z, sum z^2)
  float* dist_out;                       // [F][27][N] or null
  long F;
};

// One frame: rows -> LDS, the 27 x N squared distances, the 8 nearest points per anchor (sidx), the int64 indices stored, and the
// gathered rows cat(anchor, xyz - anchor, features) built in the LDS tile gs [224][LG_GS] (zero padded).  Ends with a barrier.
// DD >= 0: the feature count as a compile-time constant (with N = 64 NKEY every index division becomes a multiplication)
template <int NKEY, int DD>
__device__ __forceinline__ void lg_group_frame(const float* __restrict__ feats, long ldf, int N_, int D_, long f, float* fs, float* dist,
                                               float* gs, const float* an, int* sidx, long long* __restrict__ idxo,
                                               float* __restrict__ dist_out) {
  constexpr int N = 64 * NKEY;
  const int D = DD >= 0 ? DD : D_;
  const int W = 6 + D, FS = 3 + D;
  (void)N_;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // ---- the frame's rows -> LDS (every load in flight before the first LDS store)
  {
    const float* src = feats + f * (long)N * ldf;
    const int total = N * FS;
    for (int i0 = 0; i0 < total; i0 += LG_NT * 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = i0 + tid + LG_NT * u < total ? i0 + tid + LG_NT * u : total - 1;
        v[u] = src[(long)(i / FS) * ldf + i % FS];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = i0 + tid + LG_NT * u;
        if (i < total) fs[i] = v[u];
      }
    }
  }
  __syncthreads();
  // ---- squared distances, the reference's rounding: dot = fma(az, z, fma(ay, y, ax x)); d = (-2 dot + |a|^2) + |p|^2; +inf for xyz == 0
  for (int i = tid; i < NA * N; i += LG_NT) {
    const int a = i / N, q = i - a * N;
    const float x = fs[q * FS], y = fs[q * FS + 1], z = fs[q * FS + 2];
    const float dot = __fmaf_rn(an[a * 4 + 2], z, __fmaf_rn(an[a * 4 + 1], y, __fmul_rn(an[a * 4], x)));
    float d = __fadd_rn(__fadd_rn(__fmul_rn(-2.0f, dot), an[a * 4 + 3]), sq3_nofma(x, y, z));
    if (x == 0.f && y == 0.f && z == 0.f) d = INFINITY;
    dist[i] = d;
    if (dist_out) dist_out[f * (long)NA * N + i] = d;
  }
  __syncthreads();
  // ---- the 8 nearest points of each anchor (ascending, ties: lowest index first): a wave per anchor
  for (int a = wave; a < NA; a += LG_NT / 64) {
    unsigned u[NKEY];
#pragma unroll
    for (int j = 0; j < NKEY; ++j) u[j] = key_code(dist[a * N + j * 64 + lane]);
    for (int rnd = 0; rnd < NS; ++rnd) {
      unsigned m = u[0];
#pragma unroll
      for (int j = 1; j < NKEY; ++j) m = u[j] < m ? u[j] : m;
      const unsigned wm = wave_min_u32(m);
      int sel = -1;
#pragma unroll
      for (int j = 0; j < NKEY; ++j) {
        const unsigned long long b = __ballot(u[j] == wm);
        if (sel < 0 && b) sel = j * 64 + (int)__ffsll((long long)b) - 1;
      }
#pragma unroll
      for (int j = 0; j < NKEY; ++j)
        if (sel == j * 64 + lane) u[j] = 0xffffffffu;
      if (lane == 0) sidx[a * NS + rnd] = sel;
    }
  }
  __syncthreads();
  if (tid < NSLOT) idxo[f * NSLOT + tid] = (long long)sidx[tid];
  // ---- gathered rows cat(anchor, xyz - anchor, features) -> LDS tile (columns >= 6 + D and rows >= 216: zero)
  for (int i = tid; i < 224 * 32; i += LG_NT) {
    const int row = i >> 5, col = i & 31;
    float v = 0.f;
    if (row < NSLOT && col < W) {
      const int a = row >> 3, q = sidx[row];
      v = col < 3 ? an[a * 4 + col] : (col < 6 ? fs[q * FS + col - 3] - an[a * 4 + col - 3] : fs[q * FS + col - 3]);
    }
    gs[row * LG_GS + col] = v;
  }
  __syncthreads();
}

// NKEY = N / 64 keys per lane
template <int NKEY, int DD>
__global__ __launch_bounds__(LG_NT) void local_group_l1_kernel(LocalGroupP p) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  constexpr int N = 64 * NKEY;
  const int D = DD >= 0 ? DD : p.D, W = 6 + D, FS = 3 + D;     // FS: floats per point row kept in LDS (xyz + features)
  float* fs = sm;                                        // [N][FS] the frame's rows
  float* dist = fs + ((N * FS + 3) & ~3);                // [27][N]
  float* gs = dist + NA * N;                             // [224][LG_GS] gathered rows, later the z1 tile
  float* ws = gs + 224 * LG_GS;                          // [32][LG_GS] W1 (zero padded)
  float* bs = ws + 32 * LG_GS;                           // [32] bias
  float* an = bs + 32;                                   // [27][4] anchors + squared norm
  int* sidx = reinterpret_cast<int*>(an + NA * 4);       // [216]
  double* red = reinterpret_cast<double*>(sidx + ((NSLOT + 1) & ~1));      // [8][2][64]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // constants of the launch: anchors, W1 (zero padded to 32 x 32), bias
  if (tid < NA) {
    const float ax = p.anchors[tid * 3], ay = p.anchors[tid * 3 + 1], az = p.anchors[tid * 3 + 2];
    an[tid * 4] = ax; an[tid * 4 + 1] = ay; an[tid * 4 + 2] = az; an[tid * 4 + 3] = sq3_nofma(ax, ay, az);
  }
  if (p.W1) {
    for (int i = tid; i < 32 * 32; i += LG_NT) {
      const int n = i >> 5, k = i & 31;
      const float w = p.W1[(n < p.C1 ? n : p.C1 - 1) * W + (k < W ? k : W - 1)];
      ws[n * LG_GS + k] = (n < p.C1 && k < W) ? w : 0.f;
    }
    if (tid < 32) bs[tid] = tid < p.C1 ? p.b1[tid] : 0.f;
  }
  double s1 = 0.0, s2 = 0.0;                             // this lane's column sums of z1 over the workgroup's frames
  for (long f = blockIdx.x; f < p.F; f += gridDim.x) {
    __syncthreads();
    lg_group_frame<NKEY, DD>(p.feats, p.ldf, N, D, f, fs, dist, gs, an, sidx, p.idx, p.dist_out);
    if (p.grouped) {
      float* g = p.grouped + f * (long)NSLOT * W;
      for (int i = tid; i < NSLOT * W; i += LG_NT) g[i] = gs[(i / W) * LG_GS + i % W];
    }
    if (!p.W1) continue;
    // ---- z1 = gathered . W1^T + b1 on 32x32x2 MFMAs: 7 row tiles of 32, one 32-column tile, K = 32
    f32x16 acc[2];
    const int r = lane & 31, h = lane >> 5;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int t = wave + 4 * i;
      acc[i] = (f32x16){0};
      if (t < 7) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const f32x4 a4 = *reinterpret_cast<const f32x4*>(gs + (t * 32 + r) * LG_GS + 16 * h + 4 * j);
          const f32x4 b4 = *reinterpret_cast<const f32x4*>(ws + r * LG_GS + 16 * h + 4 * j);
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[e], b4[e], acc[i], 0, 0, 0);
        }
      }
    }
    __syncthreads();                                       // (every wave has read its rows of the gathered tile: it becomes the z1 tile)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int t = wave + 4 * i;
      if (t < 7) {
        const float bb = bs[r];
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
          const int row = t * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
          const float z = acc[i][reg] + bb;
          gs[row * LG_GS + r] = z;
          if (row < NSLOT && r < p.C1) { s1 += (double)z; s2 += (double)z * (double)z; }
        }
      }
    }
    __syncthreads();
    {                                                      // z1 rows out: 16-byte pieces, LDS reads ahead of the stores
      const int c4n = p.C1 >> 2, n4 = NSLOT * c4n;
      float* zo = p.Z1 + f * (long)NSLOT * p.ldz1;
      for (int i0 = 0; i0 < n4; i0 += LG_NT * 4) {
        f32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int i = i0 + tid + LG_NT * u < n4 ? i0 + tid + LG_NT * u : n4 - 1;
          v[u] = *reinterpret_cast<const f32x4*>(gs + (i / c4n) * LG_GS + 4 * (i % c4n));
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int i = i0 + tid + LG_NT * u < n4 ? i0 + tid + LG_NT * u : n4 - 1;
          *reinterpret_cast<f32x4*>(zo + (long)(i / c4n) * p.ldz1 + 4 * (i % c4n)) = v[u];
        }
      }
    }
  }
  if (p.W1) {                                              // per-workgroup BatchNorm partial sums of z1 (mlp_train.hip's record format)
    __syncthreads();
    const int col = lane & 31, slot = wave * 2 + (lane >> 5);
    red[(slot * 2 + 0) * 64 + col] = s1; red[(slot * 2 + 1) * 64 + col] = s2;
    __syncthreads();
    if (tid < 128) {
      const int c = tid & 63, k = tid >> 6;
      double s = 0.0;
      if (c < 32)
        for (int q = 0; q < 8; ++q) s += red[(q * 2 + k) * 64 + c];
      p.part1[((long)blockIdx.x * 2 + k) * 64 + c] = s;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Train-mode tail of LocalPointNet: y = relu(bn(z3)) per row, score = y . w + b, softmax over the 8 rows of a group, pooled vector.
// One wave per 64-row block (8 groups); rows_per_wg rows per workgroup (mlp_train.hip's partition: the statistics partials of z3
// come from mlp_fwd_layer).  BatchNorm state: sm[0..3][c] = mean, a, b, invstd, finalized from the per-workgroup partial sums.
struct Pool8P {
  const float* Z; long ldz; long rows; int C;            // C == 64
  const double* part; int nblk; const float* gamma; const float* beta; float eps; float* rmean; float* rvar; float momentum; float* state;
  const float* aw_w; const float* aw_b;                  // attention Linear(64, 1)
  float* voxT;                                           // [F][64][27] pooled vectors in the Conv3d input order
  float* attn;                                           // [rows] softmax weights
  long rows_per_wg;
};

#define P8_NT 512
#define P8_TS 65

// lane `l` (uniform) of a float register
__device__ __forceinline__ float lane_of(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }

__device__ __forceinline__ float grp8_max(float v) {
  v = fmaxf(v, __shfl_xor(v, 1, 64)); v = fmaxf(v, __shfl_xor(v, 2, 64)); v = fmaxf(v, __shfl_xor(v, 4, 64));
  return v;
}
__device__ __forceinline__ float grp8_sum(float v) {
  v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64);
  return v;
}

// statistics finalize (the arithmetic of mlp_train.hip's mt_finalize_stats: per-workgroup partial sums of z, z^2 in fp64)
template <int NQ>
__device__ __forceinline__ void p8_finalize(const double* __restrict__ part, int nblk, long N, const float* gamma, const float* beta,
                                            float eps, float* rmean, float* rvar, float momentum, float* state, float (*sm)[64],
                                            double (*red)[2][64]) {
  const int tid = threadIdx.x, c = tid & 63, q = tid >> 6;
  const float gm = gamma[c], bt = beta[c];
  float rm = 0.f, rv = 0.f;
  if (blockIdx.x == 0 && rmean) { rm = rmean[c]; rv = rvar[c]; }
  double a1 = 0.0, a2 = 0.0;
  for (int b0 = q; b0 < nblk; b0 += 8 * NQ) {
    double v1[8], v2[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int b = b0 + u * NQ, bc = b < nblk ? b : nblk - 1;
      const double w1 = part[((long)bc * 2 + 0) * 64 + c], w2 = part[((long)bc * 2 + 1) * 64 + c];
      v1[u] = b < nblk ? w1 : 0.0; v2[u] = b < nblk ? w2 : 0.0;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) { a1 += v1[u]; a2 += v2[u]; }
  }
  red[q][0][c] = a1; red[q][1][c] = a2;
  __syncthreads();
  if (q == 0) {
    double S1 = red[0][0][c], S2 = red[0][1][c];
#pragma unroll
    for (int g = 1; g < NQ; ++g) { S1 += red[g][0][c]; S2 += red[g][1][c]; }
    const double m = S1 / (double)N;
    double var = S2 / (double)N - m * m;
    var = var > 0.0 ? var : 0.0;
    const float invstd = (float)(1.0 / sqrt(var + (double)eps));
    const float mean = (float)m, a = gm * invstd;
    if (blockIdx.x == 0) {
      if (state) { state[c] = mean; state[64 + c] = invstd; state[128 + c] = a; state[192 + c] = bt; }
      if (rmean) {
        rmean[c] = (1.f - momentum) * rm + momentum * mean;
        const double unbiased = N > 1 ? var * (double)N / (double)(N - 1) : var;
        rvar[c] = (1.f - momentum) * rv + momentum * (float)unbiased;
      }
    }
    sm[0][c] = mean; sm[1][c] = a; sm[2][c] = bt; sm[3][c] = invstd;
  }
  __syncthreads();
}

__global__ __launch_bounds__(P8_NT) void pool8_bn_act_kernel(Pool8P p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float (*sm)[64] = reinterpret_cast<float (*)[64]>(smem);                        // [4][64]
  double (*red)[2][64] = reinterpret_cast<double (*)[2][64]>(smem + 256);         // [8][2][64]
  float* tiles = smem + 256 + 2 * 8 * 2 * 64;                                     // [8 waves][64][P8_TS]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float* T = tiles + wave * 64 * P8_TS;
  const long rbeg = (long)blockIdx.x * p.rows_per_wg, rend = rbeg + p.rows_per_wg < p.rows ? rbeg + p.rows_per_wg : p.rows;
  const float wv = p.aw_w[lane], bsc = p.aw_b[0];
  p8_finalize<P8_NT / 64>(p.part, p.nblk, p.rows, p.gamma, p.beta, p.eps, p.rmean, p.rvar, p.momentum, p.state, sm, red);
  const float mu = sm[0][lane], aa = sm[1][lane], bb = sm[2][lane];
  for (long r0 = rbeg + 64 * wave; r0 < rend; r0 += 64 * (P8_NT / 64)) {
    // lane = channel: the block's 64 rows, BatchNorm + ReLU, into the wave's tile
    for (int j0 = 0; j0 < 64; j0 += 16) {
      float z[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const long row = r0 + j0 + u;
        z[u] = p.Z[(row < rend ? row : rend - 1) * p.ldz + lane];
      }
#pragma unroll
      for (int u = 0; u < 16; ++u) T[(j0 + u) * P8_TS + lane] = fmaxf(__builtin_fmaf(z[u] - mu, aa, bb), 0.f);
    }
    // lane = row: score, softmax over the 8 rows of its group
    float sc = bsc;
#pragma unroll 16
    for (int c = 0; c < 64; ++c) sc = __builtin_fmaf(T[lane * P8_TS + c], lane_of(wv, c), sc);
    const float mx = grp8_max(sc);
    const float ex = expf(sc - mx);
    const float at = ex / grp8_sum(ex);
    if (r0 + lane < rend) p.attn[r0 + lane] = at;
    // lane = channel: pooled vectors of the 8 groups -> voxT[f][c][anchor]
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      float o = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) o = __builtin_fmaf(lane_of(at, g * 8 + j), T[(g * 8 + j) * P8_TS + lane], o);
      const long gi = (r0 >> 3) + g;                      // group = (frame, anchor)
      if (gi * 8 < rend) {
        const long f = gi / NA;
        p.voxT[(f * 64 + lane) * NA + (gi - f * NA)] = o;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Eval-mode anchor branch up to the pooled vectors in ONE launch (the frozen / --infer path): grouping as above, then LocalPointNet's
// three k=1 conv + BatchNorm (running statistics, folded into the weights while they are staged) + ReLU stages on MFMAs with the
// intermediates in LDS, the attention score, the softmax over each group's 8 members and the weighted sum -- the gathered rows
// (26.8 kB per frame), the per-member 32 / 48 / 64-channel activations and the pooling input never reach HBM.
// Outputs: idx [F][27][8], attn [F*216], voxT [F][64][27].  C1 = 32, C2 = 48, C3 = 64 (LocalPointNet), 6 + D <= 32.
struct LocalEvalP {
  const float* feats; long ldf; int N, D; const float* anchors; long long* idx;
  const float* W[3]; const float* b[3]; const float* gamma[3]; const float* beta[3]; const float* rmean[3]; const float* rvar[3];
  float eps;
  const float* aw_w; const float* aw_b;
  float* voxT; float* attn; long F;
};
#define LE_TS 68                 // row stride of the activation tile (64 columns)

// acc[32x32] = A[32 rows][K] . B[32 rows][K]^T, K a multiple of 32; lane (r, h) takes k = 32 c + 16 h + 4 j + e on both operands
__device__ __forceinline__ f32x16 le_tile(const float* A, int SA, const float* B, int SB, int K, int r, int h) {
  f32x16 acc = {0};
  for (int k0 = 0; k0 < K; k0 += 32) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const f32x4 a4 = *reinterpret_cast<const f32x4*>(A + r * SA + k0 + 16 * h + 4 * j);
      const f32x4 b4 = *reinterpret_cast<const f32x4*>(B + r * SB + k0 + 16 * h + 4 * j);
#pragma unroll
      for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[e], b4[e], acc, 0, 0, 0);
    }
  }
  return acc;
}

template <int NKEY, int DD>
__global__ __launch_bounds__(LG_NT) void local_front_eval_kernel(LocalEvalP p) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  constexpr int N = 64 * NKEY;
  const int D = DD >= 0 ? DD : p.D, W = 6 + D, FS = 3 + D;
  // the activation tile t2 [224][LE_TS] shares its LDS with the frame rows and the distance matrix (dead once the rows are gathered)
  float* t2 = sm;
  float* fs = sm;
  float* dist = fs + ((N * FS + 3) & ~3);
  const int region = 224 * LE_TS > ((N * FS + 3) & ~3) + NA * N ? 224 * LE_TS : ((N * FS + 3) & ~3) + NA * N;
  float* gs = sm + region;                               // [224][LG_GS]
  float* w1 = gs + 224 * LG_GS;                          // [32][36]
  float* w2 = w1 + 32 * LG_GS;                           // [64][36]  (48 rows used)
  float* w3 = w2 + 64 * LG_GS;                           // [64][68]  (48 columns used)
  float* bsv = w3 + 64 * LE_TS;                          // [3][64] folded biases, then [64] attention weight, [1] bias
  float* an = bsv + 4 * 64 + 4;
  int* sidx = reinterpret_cast<int*>(an + NA * 4);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
  if (tid < NA) {
    const float ax = p.anchors[tid * 3], ay = p.anchors[tid * 3 + 1], az = p.anchors[tid * 3 + 2];
    an[tid * 4] = ax; an[tid * 4 + 1] = ay; an[tid * 4 + 2] = az; an[tid * 4 + 3] = sq3_nofma(ax, ay, az);
  }
  {  // weights with the BatchNorm folded: W' = s W, b' = (b - mean) s + beta, s = gamma / sqrt(var + eps)
    const int C[4] = {W, 32, 48, 64};
    float* wd[3] = {w1, w2, w3};
    const int SW[3] = {LG_GS, LG_GS, LE_TS}, KP[3] = {32, 32, 64}, NP[3] = {32, 64, 64};
    if (tid < 64) {
#pragma unroll
      for (int l = 0; l < 3; ++l) {
        const int n = tid < C[l + 1] ? tid : C[l + 1] - 1;
        const float sc = p.gamma[l][n] / sqrtf(p.rvar[l][n] + p.eps);
        bsv[l * 64 + tid] = tid < C[l + 1] ? (p.b[l][n] - p.rmean[l][n]) * sc + p.beta[l][n] : 0.f;
      }
      bsv[3 * 64 + tid] = p.aw_w[tid];
      if (tid == 0) bsv[4 * 64] = p.aw_b[0];
    }
#pragma unroll
    for (int l = 0; l < 3; ++l) {
      for (int i = tid; i < NP[l] * KP[l]; i += LG_NT) {
        const int n = i / KP[l], k = i - n * KP[l];
        const int nc = n < C[l + 1] ? n : C[l + 1] - 1, kc = k < C[l] ? k : C[l] - 1;
        const float sc = p.gamma[l][nc] / sqrtf(p.rvar[l][nc] + p.eps);
        const float w = p.W[l][nc * C[l] + kc];
        wd[l][n * SW[l] + k] = (n < C[l + 1] && k < C[l]) ? sc * w : 0.f;
      }
    }
  }
  for (long f = blockIdx.x; f < p.F; f += gridDim.x) {
    __syncthreads();
    lg_group_frame<NKEY, DD>(p.feats, p.ldf, N, D, f, fs, dist, gs, an, sidx, p.idx, nullptr);
    // ---- stage 1: gathered [224][32] -> t2 columns 0..31 (K = 32)
    for (int t = wave; t < 7; t += 4) {
      const f32x16 acc = le_tile(gs + t * 32 * LG_GS, LG_GS, w1, LG_GS, 32, r, h);
      const float bb = bsv[r];
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) t2[(t * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h) * LE_TS + r] = fmaxf(acc[reg] + bb, 0.f);
    }
    __syncthreads();
    // ---- stage 2: t2[:, 0:32] -> gs-free: 48 columns written to t2 columns 0..47 AFTER every wave has read its A rows: the 14 tiles
    // (7 row x 2 column) are computed first, a barrier, then stored
    {
      f32x16 acc2[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int tl = wave + 4 * i;                       // tile = (row tile tl >> 1, column tile tl & 1)
        if (tl < 14) acc2[i] = le_tile(t2 + (tl >> 1) * 32 * LE_TS, LE_TS, w2 + (tl & 1) * 32 * LG_GS, LG_GS, 32, r, h);
      }
      __syncthreads();
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int tl = wave + 4 * i;
        if (tl < 14) {
          const int col = (tl & 1) * 32 + r;
          const float bb = bsv[64 + col];
#pragma unroll
          for (int reg = 0; reg < 16; ++reg)
            t2[((tl >> 1) * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h) * LE_TS + col] = col < 48 ? fmaxf(acc2[i][reg] + bb, 0.f) : 0.f;
        }
      }
    }
    __syncthreads();
    // ---- stage 3: t2[:, 0:64] (columns 48..63 zero) -> 64 columns, same two-phase form
    {
      f32x16 acc3[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int tl = wave + 4 * i;
        if (tl < 14) acc3[i] = le_tile(t2 + (tl >> 1) * 32 * LE_TS, LE_TS, w3 + (tl & 1) * 32 * LE_TS, LE_TS, 64, r, h);
      }
      __syncthreads();
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int tl = wave + 4 * i;
        if (tl < 14) {
          const int col = (tl & 1) * 32 + r;
          const float bb = bsv[128 + col];
#pragma unroll
          for (int reg = 0; reg < 16; ++reg)
            t2[((tl >> 1) * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h) * LE_TS + col] = fmaxf(acc3[i][reg] + bb, 0.f);
        }
      }
    }
    __syncthreads();
    // ---- pooling: a wave per 64-row block (8 groups): lane = row: score; softmax over the group's 8 lanes; lane = channel: weighted sum
    {
      const int r0 = wave * 64, row = r0 + lane;
      float sc = bsv[4 * 64];
#pragma unroll
      for (int c = 0; c < 64; c += 4) {
        const f32x4 y4 = *reinterpret_cast<const f32x4*>(t2 + row * LE_TS + c);
        const f32x4 w4 = *reinterpret_cast<const f32x4*>(bsv + 3 * 64 + c);
        sc = __builtin_fmaf(y4.x, w4.x, sc); sc = __builtin_fmaf(y4.y, w4.y, sc); sc = __builtin_fmaf(y4.z, w4.z, sc); sc = __builtin_fmaf(y4.w, w4.w, sc);
      }
      const float mx = grp8_max(sc);
      const float ex = expf(sc - mx);
      const float at = ex / grp8_sum(ex);
      if (row < NSLOT) p.attn[f * NSLOT + row] = at;
#pragma unroll
      for (int g = 0; g < 8; ++g) {
        const int a = (r0 >> 3) + g;                       // anchor
        if (a < NA) {                                      // (uniform)
          float o = 0.f;
#pragma unroll
          for (int j = 0; j < 8; ++j) o = __builtin_fmaf(lane_of(at, g * 8 + j), t2[(a * 8 + j) * LE_TS + lane], o);
          p.voxT[(f * 64 + lane) * NA + a] = o;
        }
      }
    }
  }
}

// Backward of the pooling, with the stage's BatchNorm sums and the attention parameter partials.
struct Pool8BwdP {
  const float* Z; long ldz; long rows;                   // z3 [rows][64]
  const float* state;                                    // [4][64] mean, invstd, a, b of the stage's BatchNorm
  const float* attn;                                     // [rows]
  const float* dvoxT;                                    // [F][64][27]
  const float* aw_w;                                     // [64]
  float* dY; long lddy;                                  // gradient of the activated rows [rows][64]
  double* gpart;                                         // [gridDim.x][2][64] (sum g, sum g xhat), g = dY . [y > 0]
  float* awpart;                                         // [gridDim.x][128]: d(attention weight) [64], d(bias) at [64]
  long rows_per_wg;
};

__global__ __launch_bounds__(P8_NT) void pool8_bwd_kernel(Pool8BwdP p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  double (*red)[2][64] = reinterpret_cast<double (*)[2][64]>(smem);               // [8][2][64]
  float* dwred = smem + 2 * 8 * 2 * 64;                                           // [8][65]
  float* tiles = dwred + 8 * 65 + 3;                                              // per wave: T [64][65], DV [8][65], tv [8]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float* T = tiles + wave * (64 * P8_TS + 8 * P8_TS + 8);
  float* DV = T + 64 * P8_TS;
  float* tv = DV + 8 * P8_TS;
  const long rbeg = (long)blockIdx.x * p.rows_per_wg, rend = rbeg + p.rows_per_wg < p.rows ? rbeg + p.rows_per_wg : p.rows;
  const float mu = p.state[lane], is = p.state[64 + lane], aa = p.state[128 + lane], bb = p.state[192 + lane], wv = p.aw_w[lane];
  double s1 = 0.0, s2 = 0.0;                             // (fp32 partial sums over 32 rows each, fp64 across blocks and workgroups)
  float s1f[2] = {0.f, 0.f}, s2f[2] = {0.f, 0.f};
  float dwacc = 0.f, dbacc = 0.f;
  for (long r0 = rbeg + 64 * wave; r0 < rend; r0 += 64 * (P8_NT / 64)) {
    // lane = channel: activated rows (recomputed) -> tile; the 8 groups' pooled gradients; lane = row: its softmax weight
    float dv[8];
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      long gi = (r0 >> 3) + g;
      gi = gi * 8 < rend ? gi : (rend >> 3) - 1;
      const long f = gi / NA;
      dv[g] = p.dvoxT[(f * 64 + lane) * NA + (gi - f * NA)];
    }
    const float at = p.attn[r0 + lane < rend ? r0 + lane : rend - 1];
    for (int j0 = 0; j0 < 64; j0 += 16) {
      float z[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const long row = r0 + j0 + u;
        z[u] = p.Z[(row < rend ? row : rend - 1) * p.ldz + lane];
      }
#pragma unroll
      for (int u = 0; u < 16; ++u) T[(j0 + u) * P8_TS + lane] = fmaxf(__builtin_fmaf(z[u] - mu, aa, bb), 0.f);
    }
    // pooled vectors (recomputed), tv_g = dvec_g . vec_g
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      float o = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) o = __builtin_fmaf(lane_of(at, g * 8 + j), T[(g * 8 + j) * P8_TS + lane], o);
      DV[g * P8_TS + lane] = dv[g];
      const float t = wave_sum(dv[g] * o);
      if (lane == 0) tv[g] = t;
    }
    // lane = row: t_j = dvec_g . y_j;  ds_j = a_j (t_j - tv_g)
    float tj = 0.f;
    {
      const float* dvr = DV + (lane >> 3) * P8_TS;
#pragma unroll 16
      for (int c = 0; c < 64; ++c) tj = __builtin_fmaf(T[lane * P8_TS + c], dvr[c], tj);
    }
    const float ds = r0 + lane < rend ? at * (tj - tv[lane >> 3]) : 0.f;
    const float atv = r0 + lane < rend ? at : 0.f;
    dbacc += ds;                                           // (lane = row here; summed over the lanes at the end)
    // lane = channel: dy_j[c] = a_j dvec_g[c] + ds_j w[c]; BatchNorm sums through the ReLU mask; attention weight gradient.  One group
    // (8 rows) at a time: a group is entirely inside or outside the row range (rows is a multiple of 8)
#pragma unroll 1
    for (int g = 0; g < 8; ++g) {
      const long rg = r0 + 8 * g;
      const bool in = rg < rend;                           // (uniform)
      const float dvg = DV[g * P8_TS + lane];
      float z[8], dy[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) z[u] = p.Z[(in ? rg + u : rend - 1) * p.ldz + lane];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int j = g * 8 + u;
        const float aj = lane_of(atv, j), dsj = lane_of(ds, j);
        dy[u] = __builtin_fmaf(aj, dvg, dsj * wv);
        const float y = T[j * P8_TS + lane];
        dwacc = __builtin_fmaf(dsj, y, dwacc);
        const float gq = y > 0.f ? dy[u] : 0.f;
        const float xh = (z[u] - mu) * is;
        s1f[g & 1] += gq;
        s2f[g & 1] = __builtin_fmaf(gq, xh, s2f[g & 1]);
      }
      if (in) {
#pragma unroll
        for (int u = 0; u < 8; ++u) p.dY[(rg + u) * p.lddy + lane] = dy[u];
      }
    }
    s1 += (double)s1f[0] + (double)s1f[1]; s2 += (double)s2f[0] + (double)s2f[1];
    s1f[0] = s1f[1] = s2f[0] = s2f[1] = 0.f;
  }
  // per-workgroup partials: the waves' sums added in a fixed order
  const float dbw = wave_sum(dbacc);
  red[wave][0][lane] = s1; red[wave][1][lane] = s2;
  dwred[wave * 65 + lane] = dwacc;
  if (lane == 0) dwred[wave * 65 + 64] = dbw;
  __syncthreads();
  if (tid < 128) {
    const int c = tid & 63, k = tid >> 6;
    double s = red[0][k][c];
#pragma unroll
    for (int q = 1; q < P8_NT / 64; ++q) s += red[q][k][c];
    p.gpart[((long)blockIdx.x * 2 + k) * 64 + c] = s;
  } else if (tid < 128 + 65) {
    const int c = tid - 128;
    float s = dwred[c];
#pragma unroll
    for (int q = 1; q < P8_NT / 64; ++q) s += dwred[q * 65 + c];
    p.awpart[(long)blockIdx.x * 128 + c] = s;
  } else if (tid < 256) {
    p.awpart[(long)blockIdx.x * 128 + (tid - 128)] = 0.f;
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// The same two fusions for GlobalPointNet's pooling over the 128 points of a frame (Upper_Net.py:285-301): a workgroup of 8 waves
// takes the 2 frames (256 rows) mlp_train.hip's partition gives it, 4 waves per frame.
//   pool128_bn_act   y = relu(BN(z3)) (statistics finalized in the prologue), score, softmax over the frame's 128 rows, pooled
//                    vector: replaces mlp_bn_act + attn_pool_forward (the activated 64-channel tensor, 16.8 MB at the bench shape,
//                    is neither written nor read)
//   pool128_bwd      row gradients with y recomputed, the stage's BatchNorm sums, attention parameter partials: replaces
//                    attn_pool_backward + 2 column sums + mlp_bn_bwd_reduce
#define P128_TS 65
struct Pool128P {
  const float* Z; long ldz; long rows;
  const double* part; int nblk; const float* gamma; const float* beta; float eps; float* rmean; float* rvar; float momentum; float* state;
  const float* aw_w; const float* aw_b;
  float* vec;                                            // [F][64]
  float* attn;                                           // [rows]
};

__global__ __launch_bounds__(P8_NT) void pool128_bn_act_kernel(Pool128P p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float (*sm)[64] = reinterpret_cast<float (*)[64]>(smem);                        // [4][64]
  double (*red)[2][64] = reinterpret_cast<double (*)[2][64]>(smem + 256);         // [8][2][64]
  float* tiles = smem + 256 + 2 * 8 * 2 * 64;                                     // [2 frames][128][P128_TS]
  float* aux = tiles + 2 * 128 * P128_TS;                                         // per frame: at [128], red4 [4][64], mx[2], sx[2]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = wave >> 2, w4 = wave & 3;
  float* T = tiles + fr * 128 * P128_TS;
  float* at_s = aux + fr * (128 + 256 + 8);
  float* vred = at_s + 128;
  float* mred = vred + 256;
  const long f = (long)blockIdx.x * 2 + fr;
  const long F = p.rows / 128;
  const float wv = p.aw_w[lane], bsc = p.aw_b[0];
  // the frame's rows: requested before the statistics prologue (wave w4 takes rows 32 w4 .. 32 w4 + 31)
  float z[32];
  const long r0 = (f < F ? f : F - 1) * 128 + 32 * w4;
#pragma unroll
  for (int u = 0; u < 32; ++u) z[u] = p.Z[(r0 + u) * p.ldz + lane];
  p8_finalize<P8_NT / 64>(p.part, p.nblk, p.rows, p.gamma, p.beta, p.eps, p.rmean, p.rvar, p.momentum, p.state, sm, red);
  const float mu = sm[0][lane], aa = sm[1][lane], bb = sm[2][lane];
#pragma unroll
  for (int u = 0; u < 32; ++u) T[(32 * w4 + u) * P128_TS + lane] = fmaxf(__builtin_fmaf(z[u] - mu, aa, bb), 0.f);
  __syncthreads();
  // scores: waves 0 / 1 of the frame take rows 0..63 / 64..127 (lane = row)
  float sc = 0.f;
  if (w4 < 2) {
    sc = bsc;
    const float* tr = T + (64 * w4 + lane) * P128_TS;
#pragma unroll 16
    for (int c = 0; c < 64; ++c) sc = __builtin_fmaf(tr[c], lane_of(wv, c), sc);
    const float m = wave_max(sc);
    if (lane == 0) mred[w4] = m;
  }
  __syncthreads();
  if (w4 < 2) {
    const float mx = fmaxf(mred[0], mred[1]);
    const float ex = expf(sc - mx);
    at_s[64 * w4 + lane] = ex;
    const float sx = wave_sum(ex);
    if (lane == 0) mred[2 + w4] = sx;
  }
  __syncthreads();
  const float inv = 1.0f / (mred[2] + mred[3]);
  if (w4 < 2 && f < F) p.attn[f * 128 + 64 * w4 + lane] = at_s[64 * w4 + lane] * inv;
  // pooled vector: lane = channel, wave w4 sums its 32 rows; the four partial sums are added in wave order
  float o = 0.f;
#pragma unroll 8
  for (int u = 0; u < 32; ++u) o = __builtin_fmaf(at_s[32 * w4 + u] * inv, T[(32 * w4 + u) * P128_TS + lane], o);
  vred[w4 * 64 + lane] = o;
  __syncthreads();
  if (w4 == 0 && f < F) p.vec[f * 64 + lane] = ((vred[lane] + vred[64 + lane]) + vred[128 + lane]) + vred[192 + lane];
}

struct Pool128BwdP {
  const float* Z; long ldz; long rows; const float* state; const float* attn; const float* vec; const float* dvec; const float* aw_w;
  float* dY; long lddy; double* gpart; float* awpart;
};

__global__ __launch_bounds__(P8_NT) void pool128_bwd_kernel(Pool128BwdP p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  double (*red)[2][64] = reinterpret_cast<double (*)[2][64]>(smem);               // [8][2][64]
  float* dwred = smem + 2 * 8 * 2 * 64;                                           // [8][65]
  float* tiles = dwred + 8 * 65 + 3;                                              // [2][128][P128_TS]
  float* aux = tiles + 2 * 128 * P128_TS;                                         // per frame: at [128], ds [128], dv [64], tv [4]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = wave >> 2, w4 = wave & 3;
  float* T = tiles + fr * 128 * P128_TS;
  float* at_s = aux + fr * (128 + 128 + 64 + 4);
  float* ds_s = at_s + 128;
  float* dv_s = ds_s + 128;                 // (the last 4 floats of the frame's aux block, at dv_s + 64, are spare)
  const long F = p.rows / 128;
  const long f = (long)blockIdx.x * 2 + fr;
  const bool in = f < F;
  const long fc = in ? f : F - 1;
  const float mu = p.state[lane], is = p.state[64 + lane], aa = p.state[128 + lane], bb = p.state[192 + lane], wv = p.aw_w[lane];
  float z[32];
  const long r0 = fc * 128 + 32 * w4;
#pragma unroll
  for (int u = 0; u < 32; ++u) z[u] = p.Z[(r0 + u) * p.ldz + lane];
  const float dv = p.dvec[fc * 64 + lane], vc = p.vec[fc * 64 + lane];
  if (w4 < 2) at_s[64 * w4 + lane] = p.attn[fc * 128 + 64 * w4 + lane];
  if (w4 == 2) dv_s[lane] = dv;
  const float tvv = wave_sum(dv * vc);                   // dvec . vec (every wave of the frame computes the same value)
#pragma unroll
  for (int u = 0; u < 32; ++u) T[(32 * w4 + u) * P128_TS + lane] = fmaxf(__builtin_fmaf(z[u] - mu, aa, bb), 0.f);
  __syncthreads();
  float dbacc = 0.f;
  if (w4 < 2) {                                          // lane = row: t_j = dvec . y_j;  ds_j = a_j (t_j - tv)
    const float* tr = T + (64 * w4 + lane) * P128_TS;
    float tj = 0.f;
#pragma unroll 16
    for (int c = 0; c < 64; ++c) tj = __builtin_fmaf(tr[c], dv_s[c], tj);
    const float ds = in ? at_s[64 * w4 + lane] * (tj - tvv) : 0.f;
    ds_s[64 * w4 + lane] = ds;
    dbacc = ds;
  }
  __syncthreads();
  // lane = channel, this wave's 32 rows: dy = a_j dvec[c] + ds_j w[c]; BatchNorm sums through the ReLU mask; attention weight gradient
  float s1f = 0.f, s2f = 0.f, dwacc = 0.f;
  float dy[32];
#pragma unroll
  for (int u = 0; u < 32; ++u) {
    const int j = 32 * w4 + u;
    const float aj = in ? at_s[j] : 0.f, dsj = ds_s[j];
    dy[u] = __builtin_fmaf(aj, dv, dsj * wv);
    const float y = T[j * P128_TS + lane];
    dwacc = __builtin_fmaf(dsj, y, dwacc);
    const float gq = y > 0.f ? dy[u] : 0.f;
    s1f += gq;
    s2f = __builtin_fmaf(gq, (z[u] - mu) * is, s2f);
  }
  if (in) {
#pragma unroll
    for (int u = 0; u < 32; ++u) p.dY[(r0 + u) * p.lddy + lane] = dy[u];
  }
  const float dbw = wave_sum(dbacc);
  red[wave][0][lane] = (double)s1f; red[wave][1][lane] = (double)s2f;
  dwred[wave * 65 + lane] = dwacc;
  if (lane == 0) dwred[wave * 65 + 64] = dbw;
  __syncthreads();
  if (tid < 128) {
    const int c = tid & 63, k = tid >> 6;
    double s = red[0][k][c];
#pragma unroll
    for (int q = 1; q < P8_NT / 64; ++q) s += red[q][k][c];
    p.gpart[((long)blockIdx.x * 2 + k) * 64 + c] = s;
  } else if (tid < 128 + 65) {
    const int c = tid - 128;
    float s = dwred[c];
#pragma unroll
    for (int q = 1; q < P8_NT / 64; ++q) s += dwred[q * 65 + c];
    p.awpart[(long)blockIdx.x * 128 + c] = s;
  } else if (tid < 256) {
    p.awpart[(long)blockIdx.x * 128 + (tid - 128)] = 0.f;
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// dfeats[f][p][0:3+D] += sum over the slots with idx == p of dgrouped[slot][3:6+D]  (slot order: deterministic)
__global__ __launch_bounds__(256) void anchor_scatter_kernel(const float* __restrict__ dgrouped, const long long* __restrict__ idx,
                                                             int N, int D, float* __restrict__ dxf, long lddx, long F) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int C = 3 + D, W = 6 + D;
  float* acc = sm;                         // [N][C]
  float* dg = acc + N * C;                 // [216][C]
  int* sidx = reinterpret_cast<int*>(dg + NSLOT * C);
  const int tid = threadIdx.x;
  for (long f = blockIdx.x; f < F; f += gridDim.x) {
    __syncthreads();
    const float* src = dgrouped + f * (long)NSLOT * W;
    {
      const int total = NSLOT * C;
      for (int i0 = 0; i0 < total; i0 += 256 * 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int i = i0 + tid + 256 * u < total ? i0 + tid + 256 * u : total - 1;
          v[u] = src[(long)(i / C) * W + 3 + i % C];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (i0 + tid + 256 * u < total) dg[i0 + tid + 256 * u] = v[u];
      }
    }
    if (tid < NSLOT) sidx[tid] = (int)idx[f * NSLOT + tid];
    for (int i = tid; i < N * C; i += 256) acc[i] = 0.f;
    __syncthreads();
    const int j = tid / C, c = tid - j * C;              // 8 slots x C channels per anchor step (8 C <= 256)
    for (int a = 0; a < NA; ++a) {
      if (j < NS) {
        const int q = sidx[a * NS + j];
        if (q >= 0 && q < N) acc[q * C + c] += dg[(a * NS + j) * C + c];
      }
      __syncthreads();
    }
    float* dst = dxf + f * (long)N * lddx;
    for (int i = tid; i < N * C; i += 256) {
      const int q = i / C, cc = i - q * C;
      dst[(long)q * lddx + cc] += acc[i];
    }
  }
}

// =============================================================================================================================
extern "C" int mmego_local_group_l1(void* stream, const float* feats, long ldf, long F, int N, int D, const float* anchors,
                                    long long* idx, float* grouped, const float* W1, const float* b1, int C1, float* Z1, long ldz1,
                                    double* part1, int nwg, float* dist_out) {
  MMEGO_REQUIRE(feats && anchors && idx && F > 0 && D >= 0 && 6 + D <= 32 && ldf >= 3 + D);
  MMEGO_REQUIRE((N == 64 || N == 128 || N == 256) && nwg >= 1);
  MMEGO_REQUIRE(!W1 || (b1 && Z1 && part1 && C1 >= 4 && C1 <= 32 && (C1 % 4) == 0 && ldz1 >= C1 && (ldz1 % 4) == 0 && ((uintptr_t)Z1 & 15) == 0));
  LocalGroupP p = {feats, ldf, N, D, anchors, idx, grouped, W1, b1, C1, Z1, ldz1, part1, dist_out, F};
  const int FS = 3 + D;
  const size_t fl = (size_t)((N * FS + 3) & ~3) + (size_t)NA * N + 224 * LG_GS + 32 * LG_GS + 32 + NA * 4 + ((NSLOT + 1) & ~1);
  const size_t lds = fl * sizeof(float) + 8 * 2 * 64 * sizeof(double);
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((unsigned)(nwg < F ? nwg : F));
#define LG_LAUNCH(NK_)                                                                                                  \
  do {                                                                                                                  \
    static size_t attr = 0;                                                                                             \
    if (lds > 64 * 1024 && lds > attr) {                                                                                \
      hipError_t e = hipFuncSetAttribute((const void*)local_group_l1_kernel<NK_, DD_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
      if (e != hipSuccess) return (int)e;                                                                               \
      attr = lds;                                                                                                       \
    }                                                                                                                   \
    hipLaunchKernelGGL((local_group_l1_kernel<NK_, DD_>), grid, dim3(LG_NT), lds, st, p);                               \
  } while (0)
  if (N == 128 && D == 25) { constexpr int DD_ = 25; LG_LAUNCH(2); }       // (the reference's shape: constants folded)
  else { constexpr int DD_ = -1; if (N == 64) LG_LAUNCH(1); else if (N == 128) LG_LAUNCH(2); else LG_LAUNCH(4); }
#undef LG_LAUNCH
  MMEGO_LAUNCH_CHECK();
  return MMEGO_OK;
}

extern "C" int mmego_local_front_eval(void* stream, const float* feats, long ldf, long F, int N, int D, const float* anchors,
                                      long long* idx, const float* const* tab, double eps, float* voxT, float* attn) {
  // tab: HOST array of 20 device pointers: per layer l = 0..2 {W, b, gamma, beta, running_mean, running_var} (18), then the attention
  // weight [64] and bias [1]
  MMEGO_REQUIRE(feats && anchors && idx && tab && voxT && attn && F > 0 && D >= 0 && 6 + D <= 32 && ldf >= 3 + D);
  MMEGO_REQUIRE(N == 64 || N == 128 || N == 256);
  for (int i = 0; i < 20; ++i) MMEGO_REQUIRE(tab[i] != nullptr);
  LocalEvalP p;
  p.feats = feats; p.ldf = ldf; p.N = N; p.D = D; p.anchors = anchors; p.idx = idx;
  for (int l = 0; l < 3; ++l) {
    p.W[l] = tab[6 * l]; p.b[l] = tab[6 * l + 1]; p.gamma[l] = tab[6 * l + 2]; p.beta[l] = tab[6 * l + 3];
    p.rmean[l] = tab[6 * l + 4]; p.rvar[l] = tab[6 * l + 5];
  }
  p.eps = (float)eps; p.aw_w = tab[18]; p.aw_b = tab[19]; p.voxT = voxT; p.attn = attn; p.F = F;
  const int FS = 3 + D;
  const size_t head = (size_t)((N * FS + 3) & ~3) + (size_t)NA * N;
  const size_t region = head > (size_t)224 * LE_TS ? head : (size_t)224 * LE_TS;
  const size_t fl = region + 224 * LG_GS + 32 * LG_GS + 64 * LG_GS + 64 * LE_TS + 4 * 64 + 4 + NA * 4 + NSLOT;
  const size_t lds = fl * sizeof(float);
  MMEGO_REQUIRE(lds <= 160 * 1024);
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((unsigned)(F < 512 ? F : 512));
#define LE_LAUNCH(NK_)                                                                                                  \
  do {                                                                                                                  \
    static size_t attr = 0;                                                                                             \
    if (lds > 64 * 1024 && lds > attr) {                                                                                \
      hipError_t e = hipFuncSetAttribute((const void*)local_front_eval_kernel<NK_, DD_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
      if (e != hipSuccess) return (int)e;                                                                               \
      attr = lds;                                                                                                       \
    }                                                                                                                   \
    hipLaunchKernelGGL((local_front_eval_kernel<NK_, DD_>), grid, dim3(LG_NT), lds, st, p);                             \
  } while (0)
  if (N == 128 && D == 25) { constexpr int DD_ = 25; LE_LAUNCH(2); }
  else { constexpr int DD_ = -1; if (N == 64) LE_LAUNCH(1); else if (N == 128) LE_LAUNCH(2); else LE_LAUNCH(4); }
#undef LE_LAUNCH
  MMEGO_LAUNCH_CHECK();
  return MMEGO_OK;
}

static inline void p8_grid(long rows, int* nblk, long* rpw) {      // mlp_train.hip's partition (mt_grid): 256-row rounds, <= 256 workgroups
  const long tiles = (rows + 255) / 256;
  long n = tiles < 256 ? tiles : 256;
  const long tpw = (tiles + n - 1) / n;
  n = (tiles + tpw - 1) / tpw;
  *nblk = (int)n;
  *rpw = tpw * 256;
}

extern "C" int mmego_pool8_bn_act(void* stream, const float* Z, long ldz, long rows, const double* part, const float* gamma,
                                  const float* beta, double eps, float* rmean, float* rvar, double momentum, float* state,
                                  const float* aw_w, const float* aw_b, float* voxT, float* attn) {
  MMEGO_REQUIRE(Z && part && gamma && beta && aw_w && aw_b && voxT && attn && rows > 0 && (rows % (8 * NA)) == 0 && ldz >= 64);
  int nblk; long rpw;
  p8_grid(rows, &nblk, &rpw);
  Pool8P p = {Z, ldz, rows, 64, part, nblk, gamma, beta, (float)eps, rmean, rvar, (float)momentum, state, aw_w, aw_b, voxT, attn, rpw};
  const size_t lds = (size_t)(256 + 2 * 8 * 2 * 64 + 8 * 64 * P8_TS) * sizeof(float);
  static bool attr = false;
  if (!attr) {
    hipError_t e = hipFuncSetAttribute((const void*)pool8_bn_act_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    attr = true;
  }
  hipLaunchKernelGGL(pool8_bn_act_kernel, dim3(nblk), dim3(P8_NT), lds, (hipStream_t)stream, p);
  MMEGO_LAUNCH_CHECK();
  return MMEGO_OK;
}

extern "C" int mmego_pool8_nblk(long rows) {
  int nblk; long rpw;
  p8_grid(rows > 0 ? rows : 1, &nblk, &rpw);
  return nblk;
}

extern "C" int mmego_pool8_backward(void* stream, const float* Z, long ldz, long rows, const float* state, const float* attn,
                                    const float* dvoxT, const float* aw_w, float* dY, long lddy, double* gpart, float* awpart) {
  MMEGO_REQUIRE(Z && state && attn && dvoxT && aw_w && dY && gpart && awpart && rows > 0 && (rows % (8 * NA)) == 0 && ldz >= 64 && lddy >= 64);
  int nblk; long rpw;
  p8_grid(rows, &nblk, &rpw);
  Pool8BwdP p = {Z, ldz, rows, state, attn, dvoxT, aw_w, dY, lddy, gpart, awpart, rpw};
  const size_t lds = (size_t)(2 * 8 * 2 * 64 + 8 * 65 + 3 + 8 * (64 * P8_TS + 8 * P8_TS + 8)) * sizeof(float);
  static bool attr = false;
  if (!attr) {
    hipError_t e = hipFuncSetAttribute((const void*)pool8_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    attr = true;
  }
  hipLaunchKernelGGL(pool8_bwd_kernel, dim3(nblk), dim3(P8_NT), lds, (hipStream_t)stream, p);
  MMEGO_LAUNCH_CHECK();
  return MMEGO_OK;
}

// GlobalPointNet's pooling fused with the last stage's BatchNorm + ReLU (forward) / with the stage's BatchNorm sums and the attention
// parameter partials (backward): rows = F * 128, two frames per workgroup = mlp_train.hip's partition of 256-row rounds (requires
// mmego_mlp_train_nblk(rows) == rows / 256, i.e. rows <= 65 536: larger batches take the separate launches).
extern "C" int mmego_pool128_ok(long rows) {
  int nblk; long rpw;
  p8_grid(rows > 0 ? rows : 1, &nblk, &rpw);
  return rows > 0 && (rows % 256) == 0 && rpw == 256 ? 1 : 0;
}

extern "C" int mmego_pool128_bn_act(void* stream, const float* Z, long ldz, long rows, const double* part, const float* gamma,
                                    const float* beta, double eps, float* rmean, float* rvar, double momentum, float* state,
                                    const float* aw_w, const float* aw_b, float* vec, float* attn) {
  MMEGO_REQUIRE(Z && part && gamma && beta && aw_w && aw_b && vec && attn && mmego_pool128_ok(rows) && ldz >= 64);
  int nblk; long rpw;
  p8_grid(rows, &nblk, &rpw);
  Pool128P p = {Z, ldz, rows, part, nblk, gamma, beta, (float)eps, rmean, rvar, (float)momentum, state, aw_w, aw_b, vec, attn};
  const size_t lds = (size_t)(256 + 2 * 8 * 2 * 64 + 2 * 128 * P128_TS + 2 * (128 + 256 + 8)) * sizeof(float);
  static bool attr = false;
  if (!attr) {
    hipError_t e = hipFuncSetAttribute((const void*)pool128_bn_act_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    attr = true;
  }
  hipLaunchKernelGGL(pool128_bn_act_kernel, dim3(nblk), dim3(P8_NT), lds, (hipStream_t)stream, p);
  MMEGO_LAUNCH_CHECK();
  return MMEGO_OK;
}

extern "C" int mmego_pool128_backward(void* stream, const float* Z, long ldz, long rows, const float* state, const float* attn,
                                      const float* vec, const float* dvec, const float* aw_w, float* dY, long lddy, double* gpart,
                                      float* awpart) {
  MMEGO_REQUIRE(Z && state && attn && vec && dvec && aw_w && dY && gpart && awpart && mmego_pool128_ok(rows) && ldz >= 64 && lddy >= 64);
  int nblk; long rpw;
  p8_grid(rows, &nblk, &rpw);
  Pool128BwdP p = {Z, ldz, rows, state, attn, vec, dvec, aw_w, dY, lddy, gpart, awpart};
  const size_t lds = (size_t)(2 * 8 * 2 * 64 + 8 * 65 + 3 + 2 * 128 * P128_TS + 2 * (128 + 128 + 64 + 4)) * sizeof(float);
  static bool attr = false;
  if (!attr) {
    hipError_t e = hipFuncSetAttribute((const void*)pool128_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    attr = true;
  }
  hipLaunchKernelGGL(pool128_bwd_kernel, dim3(nblk), dim3(P8_NT), lds, (hipStream_t)stream, p);
  MMEGO_LAUNCH_CHECK();
  return MMEGO_OK;
}

extern "C" int mmego_anchor_scatter(void* stream, const float* dgrouped, const long long* idx, long F, int N, int D, float* dxf,
                                    long lddx) {
  MMEGO_REQUIRE(dgrouped && idx && dxf && F > 0 && N > 0 && N <= 512 && D >= 0 && 8 * (3 + D) <= 256 && lddx >= 3 + D);
  const int C = 3 + D;
  const size_t lds = (size_t)(N * C + NSLOT * C) * sizeof(float) + NSLOT * sizeof(int);
  MMEGO_REQUIRE(lds <= 64 * 1024);
  hipLaunchKernelGGL(anchor_scatter_kernel, dim3((unsigned)(F < 1024 ? F : 1024)), dim3(256), lds, (hipStream_t)stream, dgrouped, idx, N, D,
                     dxf, lddx, F);
  MMEGO_LAUNCH_CHECK();
  return MMEGO_OK;
}
