// LocalVoxelNet's training step (Net/Upper_Net.py:180-205: Conv3d(64, 96, k=3) over the whole 3x3x3 anchor grid -- one 1728 -> 96 map
// per frame -- then two 1x1x1 convs 96 -> 128 -> 64, a train-mode BatchNorm3d + ReLU behind each) as 4 forward and 5 backward launches.
//
// The chain is 512 rows (B*T frames) wide at the bench shape: every tensor but the first weight is a few hundred KB, and as generic
// launches (product / column statistics / finalize / affine + ReLU per stage; bn_bwd_reduce / finalize / apply + two products per stage
// backward) it was 27 launches of ~4.8 us each, all launch latency (profiles/r04_wlocal_timeline.txt 314.8-393 us and 828-915 us).
// Here a launch boundary exists only where a BatchNorm needs every row's statistics:
//
//   forward   vox_l1_fwd     z1 = X W1^T + b1                                  records of z1
//             vox_mid_fwd    y1 = relu(bn1(z1)) [BatchNorm finalized in the prologue], z2 = y1 W2^T + b2, records of z2
//             vox_mid_fwd    y2 = relu(bn2(z2)), z3 = y2 W3^T + b3, records of z3
//             vox_out_fwd    y3 = relu(bn3(z3))
//   backward  vox_bwd_sums   g3 = dy3 . [y3 > 0], per-tile sums (sum g3, sum g3 xhat3)
//             vox_mid_bwd    dz3 from the sums, g2 = (dz3 W3) . [y2 > 0], sums of layer 2
//             vox_mid_bwd    dz2, g1 = (dz2 W2) . [y1 > 0], sums of layer 1
//             vox_l1_bwd     dz1, dX = dz1 W1
//             vox_dw         dW1 = dz1^T X, dW2 = dz2^T y1, dW3 = dz3^T y2 (one launch, 16 x 16 output tiles)
//
// Forward statistics travel as the (mean, M2) partial records of gcn_stats.h, one per 16-row tile and channel, finalized by every
// consumer workgroup in fixed order (workgroup 0 writes the state and the running statistics); backward sums as float2 partials per
// tile, added in fp64 in tile order.  Products: v_mfma_f32_16x16x4_f32, a wave per 16 x 16 output tile; an operand's k axis is walked
// in a PERMUTED order inside each 16-k chunk (lane group q takes k = 4q .. 4q+3, MFMA step j multiplies the j-th of both operands)
// so that a k-contiguous operand is one 16-byte load per lane and chunk.  fp32 throughout; the conv biases get no gradient (exactly
// zero in front of a batch-statistics BatchNorm: blocks.mlp3_backward).
#include "common.h"
#include "gcn_stats.h"

typedef float vx_f4 __attribute__((ext_vector_type(4)));
#define VX_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

// (mean, M2) of one column of a 16 x 16 accumulator tile over its first nvalid rows: lane (r = lane & 15: column, q = lane >> 4:
// rows 4q .. 4q+3).  Every lane returns the column's record.
__device__ __forceinline__ float2 vx_tile_record(const vx_f4& z, int nvalid, int q) {
  float s = 0.f;
#pragma unroll
  for (int v = 0; v < 4; ++v) s += (4 * q + v < nvalid) ? z[v] : 0.f;
  s += __shfl_xor(s, 16);
  s += __shfl_xor(s, 32);
  const float mean = s / (float)nvalid;
  float d = 0.f;
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    const float e = z[v] - mean;
    d += (4 * q + v < nvalid) ? e * e : 0.f;
  }
  d += __shfl_xor(d, 16);
  d += __shfl_xor(d, 32);
  return float2{mean, d};
}

// ---- forward, layer 1: Z[R][C] = X[R][K] W[C][K]^T + bias, records [ceil(R/16)][C] ---------------------------------------------------
// grid (row tiles, C / 16), 128 threads: the two waves take the two halves of K (a 16 x 16 tile over K = 1728 is 432 dependent MFMAs:
// 6.6 us on one SIMD) and meet in LDS.  D chunks of both operands in flight per wave.
template <int NCH, int D>
__global__ __launch_bounds__(128) void vox_l1_fwd_kernel(const float* __restrict__ X, long ldx, const float* __restrict__ W,
                                                         const float* __restrict__ bias, float* __restrict__ Z, float2* __restrict__ rec,
                                                         int R, int C) {
  // NCH = 16-k chunks per wave (K = 32 NCH), fully unrolled: the compiler then counts the outstanding requests exactly (in a rolled
  // loop it drained the memory pipe at every loop head)
  constexpr int K = 32 * NCH, kh = 16 * NCH;
  __shared__ float red[16][17];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15, q = lane >> 4;
  const int r0 = blockIdx.x * 16, c0 = blockIdx.y * 16;
  const int rowc = min(r0 + r, R - 1);
  const float* A = X + (long)rowc * ldx + wave * kh + 4 * q;
  const float* B = W + (long)(c0 + r) * K + wave * kh + 4 * q;
  vx_f4 ra[D], rb[D];
#pragma unroll
  for (int u = 0; u < D; ++u) {
    ra[u] = *reinterpret_cast<const vx_f4*>(A + 16 * u);
    rb[u] = *reinterpret_cast<const vx_f4*>(B + 16 * u);
  }
  __builtin_amdgcn_sched_barrier(0);
  vx_f4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const vx_f4 a = ra[i % D], b = rb[i % D];
    if (i + D < NCH) {
      ra[i % D] = *reinterpret_cast<const vx_f4*>(A + 16 * (i + D));
      rb[i % D] = *reinterpret_cast<const vx_f4*>(B + 16 * (i + D));
    }
    __builtin_amdgcn_sched_barrier(0);                   // (the requests stay HERE, D chunks ahead of their use: the scheduler sinks them otherwise)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc = VX_MFMA(a[j], b[j], acc);
  }
  if (wave == 1) {
#pragma unroll
    for (int v = 0; v < 4; ++v) red[4 * q + v][r] = acc[v];
  }
  __syncthreads();
  if (wave == 0) {
    const float bv = bias[c0 + r];
#pragma unroll
    for (int v = 0; v < 4; ++v) acc[v] += red[4 * q + v][r] + bv;
    const int nvalid = min(16, R - r0);
#pragma unroll
    for (int v = 0; v < 4; ++v)
      if (4 * q + v < nvalid) Z[(long)(r0 + 4 * q + v) * C + c0 + r] = acc[v];
    const float2 rc = vx_tile_record(acc, nvalid, q);
    if (q == 0) rec[(long)blockIdx.x * C + c0 + r] = rc;
  }
}

// ---- forward, layers 2 and 3: Yin = relu(bn(Zin)) (stored: the backward pass masks and multiplies with it), Zout = Yin W^T + bias ------
// grid (row tiles), COUT / 16 waves: wave w owns output columns [16w, 16w + 16).  The weight fragments are requested first and travel
// while the BatchNorm is finalized from the records.
template <int CIN, int COUT>
__global__ __launch_bounds__(64 * (COUT / 16)) void vox_mid_fwd_kernel(const float* __restrict__ Zin, BnRefD bn, float* __restrict__ Yin,
                                                                       const float* __restrict__ W, const float* __restrict__ bias,
                                                                       float* __restrict__ Zout, float2* __restrict__ rec, int R) {
  constexpr int NT = 64 * (COUT / 16), LDA = CIN + 4;
  __shared__ double red[NT * 4];
  __shared__ float st[4 * CIN];
  __shared__ __attribute__((aligned(16))) float at[16 * LDA];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, q = lane >> 4;
  const int r0 = blockIdx.x * 16, c0 = wave * 16;
  vx_f4 wb[CIN / 16];
  {
    const float* B = W + (long)(c0 + r) * CIN + 4 * q;
#pragma unroll
    for (int kc = 0; kc < CIN / 16; ++kc) wb[kc] = *reinterpret_cast<const vx_f4*>(B + 16 * kc);
  }
  const float bv = bias[c0 + r];
  bn_from_records<NT>(bn, CIN, R, red, st, blockIdx.x == 0);      // st: mean, a, b, invstd
  for (int idx = tid; idx < 16 * CIN; idx += NT) {
    const int rr = idx / CIN, c = idx - rr * CIN;
    const int row = r0 + rr;
    const float z = Zin[(long)min(row, R - 1) * CIN + c];
    const float a = fmaxf((z - st[c]) * st[CIN + c] + st[2 * CIN + c], 0.f);
    at[rr * LDA + c] = a;
    if (row < R) Yin[(long)row * CIN + c] = a;
  }
  __syncthreads();
  vx_f4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int kc = 0; kc < CIN / 16; ++kc) {
    const vx_f4 a = *reinterpret_cast<const vx_f4*>(&at[r * LDA + 16 * kc + 4 * q]);
#pragma unroll
    for (int j = 0; j < 4; ++j) acc = VX_MFMA(a[j], wb[kc][j], acc);
  }
  const int nvalid = min(16, R - r0);
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    acc[v] += bv;
    if (4 * q + v < nvalid) Zout[(long)(r0 + 4 * q + v) * COUT + c0 + r] = acc[v];
  }
  const float2 rc = vx_tile_record(acc, nvalid, q);
  if (q == 0) rec[(long)blockIdx.x * COUT + c0 + r] = rc;
}

// ---- forward, output: Y = relu(bn(Z)), C <= 128 ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void vox_out_fwd_kernel(const float* __restrict__ Z, BnRefD bn, float* __restrict__ Y, long ldy, int R, int C) {
  __shared__ double red[256 * 4];
  __shared__ float st[4 * 128];
  bn_from_records<256>(bn, C, R, red, st, blockIdx.x == 0);
  const int r0 = blockIdx.x * 16;
  for (int idx = threadIdx.x; idx < 16 * C; idx += 256) {
    const int rr = idx / C, c = idx - rr * C;
    const int row = r0 + rr;
    if (row < R) Y[(long)row * ldy + c] = fmaxf((Z[(long)row * C + c] - st[c]) * st[C + c] + st[2 * C + c], 0.f);
  }
}

// ---- backward, the output stage's sums: G = dY . [Y > 0], prt[tile][c] = (sum G, sum G xhat) over the tile's rows; C <= 128 ----------
__global__ __launch_bounds__(256) void vox_bwd_sums_kernel(const float* __restrict__ dY, long lddy, const float* __restrict__ Y, long ldy,
                                                           const float* __restrict__ Z, const float* __restrict__ state,
                                                           float* __restrict__ G, float2* __restrict__ prt, int R, int C) {
  __shared__ float2 sm[4][64];
  const int tid = threadIdx.x, cl = tid & 63, rq = tid >> 6;
  const int r0 = blockIdx.x * 16;
  for (int cb = 0; cb < C; cb += 64) {
    const int c = cb + cl;
    float s1 = 0.f, s2 = 0.f;
    if (c < C) {
      const float mean = state[c], invstd = state[C + c];
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int row = r0 + 4 * rq + v;
        if (row < R) {
          const float g = Y[(long)row * ldy + c] > 0.f ? dY[(long)row * lddy + c] : 0.f;
          G[(long)row * C + c] = g;
          s1 += g;
          s2 += g * ((Z[(long)row * C + c] - mean) * invstd);
        }
      }
    }
    sm[rq][cl] = float2{s1, s2};
    __syncthreads();
    if (rq == 0 && c < C) {
      float2 t = sm[0][cl];
#pragma unroll
      for (int g = 1; g < 4; ++g) { t.x += sm[g][cl].x; t.y += sm[g][cl].y; }
      prt[(long)blockIdx.x * C + c] = t;
    }
    __syncthreads();
  }
}

// The BatchNorm backward coefficients of C channels from the per-tile sums: cf[0..C) = sum g / R, cf[C..2C) = sum g xhat / R,
// cf[2C..5C) = mean, invstd, a (= gamma invstd) of the forward state.  Threads [0, C); tile order, fp64.  Workgroup `first` writes
// the BatchNorm's parameter gradients.
__device__ __forceinline__ void vx_bwd_coef(const float2* __restrict__ prt, int nrt, const float* __restrict__ state, int C, int R,
                                            float* cf, float* dgamma, float* dbeta, bool first) {
  const int c = threadIdx.x;
  if (c < C) {
    double s1 = 0.0, s2 = 0.0;
    for (int j0 = 0; j0 < nrt; j0 += 8) {
      float2 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = prt[(long)min(j0 + u, nrt - 1) * C + c];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (j0 + u < nrt) { s1 += (double)v[u].x; s2 += (double)v[u].y; }
      }
    }
    cf[c] = (float)(s1 / (double)R);
    cf[C + c] = (float)(s2 / (double)R);
    cf[2 * C + c] = state[c];
    cf[3 * C + c] = state[C + c];
    cf[4 * C + c] = state[2 * C + c];
    if (first) { dbeta[c] = (float)s1; dgamma[c] = (float)s2; }
  }
}

// ---- backward, layers 3 and 2 (W [COUT][CIN], BatchNorm over COUT behind it, the layer's input Yp = relu(bn_p(Zp)) over CIN) ----------
// dZ = a (G - sum g / R - xhat sum g xhat / R); Gp = (dZ W) . [Yp > 0]; prtp = the sums of Gp for the BatchNorm in front.
// grid (row tiles), CIN / 16 waves.
template <int COUT, int CIN>
__global__ __launch_bounds__(64 * (CIN / 16)) void vox_mid_bwd_kernel(const float* __restrict__ G, const float* __restrict__ Z,
                                                                      const float* __restrict__ state, const float2* __restrict__ prt,
                                                                      int nrt, float* __restrict__ dZ, float* dgamma, float* dbeta,
                                                                      const float* __restrict__ W, const float* __restrict__ Yp,
                                                                      const float* __restrict__ Zp, const float* __restrict__ statep,
                                                                      float* __restrict__ Gp, float2* __restrict__ prtp, int R) {
  constexpr int NT = 64 * (CIN / 16), LDZ = COUT + 4;
  static_assert(NT >= COUT, "one thread per channel in the prologue");
  __shared__ float cf[5 * COUT];
  __shared__ __attribute__((aligned(16))) float dzt[16 * LDZ];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, q = lane >> 4;
  const int r0 = blockIdx.x * 16, c0 = wave * 16;
  vx_f4 wb[COUT / 16];                                 // W[16 kc + 4 q + j][c0 + r]: coalesced over r
#pragma unroll
  for (int kc = 0; kc < COUT / 16; ++kc)
#pragma unroll
    for (int j = 0; j < 4; ++j) wb[kc][j] = W[(long)(16 * kc + 4 * q + j) * CIN + c0 + r];
  const float meanp = statep[c0 + r], invp = statep[CIN + c0 + r];
  vx_bwd_coef(prt, nrt, state, COUT, R, cf, dgamma, dbeta, blockIdx.x == 0);
  __syncthreads();
  for (int idx = tid; idx < 16 * COUT; idx += NT) {
    const int rr = idx / COUT, c = idx - rr * COUT;
    const int row = r0 + rr, rowc = min(row, R - 1);
    const float xhat = (Z[(long)rowc * COUT + c] - cf[2 * COUT + c]) * cf[3 * COUT + c];
    float dz = cf[4 * COUT + c] * (G[(long)rowc * COUT + c] - cf[c] - xhat * cf[COUT + c]);
    dz = row < R ? dz : 0.f;
    dzt[rr * LDZ + c] = dz;
    if (row < R) dZ[(long)row * COUT + c] = dz;
  }
  __syncthreads();
  vx_f4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int kc = 0; kc < COUT / 16; ++kc) {
    const vx_f4 a = *reinterpret_cast<const vx_f4*>(&dzt[r * LDZ + 16 * kc + 4 * q]);
#pragma unroll
    for (int j = 0; j < 4; ++j) acc = VX_MFMA(a[j], wb[kc][j], acc);
  }
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    const int row = r0 + 4 * q + v;
    if (row < R) {
      const long o = (long)row * CIN + c0 + r;
      const float g = Yp[o] > 0.f ? acc[v] : 0.f;
      Gp[o] = g;
      s1 += g;
      s2 += g * ((Zp[o] - meanp) * invp);
    }
  }
  s1 += __shfl_xor(s1, 16); s1 += __shfl_xor(s1, 32);
  s2 += __shfl_xor(s2, 16); s2 += __shfl_xor(s2, 32);
  if (q == 0) prtp[(long)blockIdx.x * CIN + c0 + r] = float2{s1, s2};
}

// ---- backward, layer 1: dZ from the sums, dX[R][K] = dZ[R][C] W[C][K] -----------------------------------------------------------------
// grid (row tiles, K / (64 TPW)), four waves of TPW column tiles each; the next tile's weight fragments travel under the current
// tile's MFMAs.
template <int C, int TPW>
__global__ __launch_bounds__(256) void vox_l1_bwd_kernel(const float* __restrict__ G, const float* __restrict__ Z, const float* __restrict__ state,
                                                         const float2* __restrict__ prt, int nrt, float* __restrict__ dZ, float* dgamma,
                                                         float* dbeta, const float* __restrict__ W, float* __restrict__ dX, long lddx,
                                                         int R, int K) {
  constexpr int LDZ = C + 4, NK = C / 16;
  __shared__ float cf[5 * C];
  __shared__ __attribute__((aligned(16))) float dzt[16 * LDZ];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, q = lane >> 4;
  const int r0 = blockIdx.x * 16;
  const int n0 = (blockIdx.y * 4 + wave) * TPW * 16;
  vx_f4 wb[2][NK];
  const float* Wl = W + (long)(4 * q) * K + n0 + r;
#pragma unroll
  for (int kc = 0; kc < NK; ++kc)
#pragma unroll
    for (int j = 0; j < 4; ++j) wb[0][kc][j] = Wl[(long)(16 * kc + j) * K];
  const bool first = blockIdx.x == 0 && blockIdx.y == 0;
  vx_bwd_coef(prt, nrt, state, C, R, cf, dgamma, dbeta, first);
  __syncthreads();
  for (int idx = tid; idx < 16 * C; idx += 256) {
    const int rr = idx / C, c = idx - rr * C;
    const int row = r0 + rr, rowc = min(row, R - 1);
    const float xhat = (Z[(long)rowc * C + c] - cf[2 * C + c]) * cf[3 * C + c];
    float dz = cf[4 * C + c] * (G[(long)rowc * C + c] - cf[c] - xhat * cf[C + c]);
    dz = row < R ? dz : 0.f;
    dzt[rr * LDZ + c] = dz;
    if (row < R && blockIdx.y == 0) dZ[(long)row * C + c] = dz;
  }
  __syncthreads();
  vx_f4 a[NK];
#pragma unroll
  for (int kc = 0; kc < NK; ++kc) a[kc] = *reinterpret_cast<const vx_f4*>(&dzt[r * LDZ + 16 * kc + 4 * q]);
#pragma unroll
  for (int t = 0; t < TPW; ++t) {
    if (t + 1 < TPW) {
#pragma unroll
      for (int kc = 0; kc < NK; ++kc)
#pragma unroll
        for (int j = 0; j < 4; ++j) wb[(t + 1) & 1][kc][j] = Wl[(long)(16 * kc + j) * K + 16 * (t + 1)];
    }
    vx_f4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kc = 0; kc < NK; ++kc)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc = VX_MFMA(a[kc][j], wb[t & 1][kc][j], acc);
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int row = r0 + 4 * q + v;
      if (row < R) dX[(long)row * lddx + n0 + 16 * t + r] = acc[v];
    }
  }
}

// ---- backward, the three weight gradients in one launch: dW_l[Cout][Cin] = dZ_l[R][Cout]^T In_l[R][Cin] -------------------------------
// a wave per 16 x 16 output tile (tiles numbered layer by layer, input-column tile fastest: the four waves of a workgroup mostly share
// their dZ fragment); K = the rows, four 16-row chunks of both operands in flight.
struct VoxDwL { const float* dZ; const float* In; long ldin; float* dW; int Cout, Cin, tile_end; };
struct VoxDwP { VoxDwL l[3]; int R; };
__global__ __launch_bounds__(256) void vox_dw_kernel(VoxDwP p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15, q = lane >> 4;
  const int t = blockIdx.x * 4 + wave;
  if (t >= p.l[2].tile_end) return;
  const int li = t < p.l[0].tile_end ? 0 : (t < p.l[1].tile_end ? 1 : 2);
  const VoxDwL& L = p.l[li];
  const int local = t - (li ? p.l[li - 1].tile_end : 0);
  const int ntn = L.Cin >> 4;
  const int mt = local / ntn, nt = local - mt * ntn;
  const int R = p.R, nch = (R + 15) >> 4;
  const float* A = L.dZ + 16 * mt + r;
  const float* B = L.In + 16 * nt + r;
  const long lda = L.Cout, ldb = L.ldin;
  constexpr int D = 4;
  vx_f4 ra[D], rb[D];
#pragma unroll
  for (int u = 0; u < D; ++u)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row = min(16 * min(u, nch - 1) + 4 * q + j, R - 1);
      ra[u][j] = A[(long)row * lda];
      rb[u][j] = B[(long)row * ldb];
    }
  vx_f4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int i0 = 0; i0 < nch; i0 += D) {
#pragma unroll
    for (int u = 0; u < D; ++u) {
      vx_f4 a = ra[u];
      const vx_f4 b = rb[u];
      const int nx = min(i0 + D + u, nch - 1);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int row = min(16 * nx + 4 * q + j, R - 1);
        ra[u][j] = A[(long)row * lda];
        rb[u][j] = B[(long)row * ldb];
      }
      __builtin_amdgcn_sched_barrier(0);
      const int rb0 = 16 * (i0 + u) + 4 * q;                 // rows past R (a ragged last chunk, chunks past the end): zero
#pragma unroll
      for (int j = 0; j < 4; ++j) a[j] = (rb0 + j < R) ? a[j] : 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) acc = VX_MFMA(a[j], b[j], acc);
    }
  }
#pragma unroll
  for (int v = 0; v < 4; ++v) L.dW[(long)(16 * mt + 4 * q + v) * L.Cin + 16 * nt + r] = acc[v];
}

// ---------------------------------------------------------------------------------------------------------------------------------------
#define VOX_C1 96
#define VOX_C2 128
#define VOX_C3 64

#define VOX_K 1728
extern "C" int mmego_vox_ok(long rows, int K, int C1, int C2, int C3) {
  return rows >= 2 && rows <= 4096 && K == VOX_K && C1 == VOX_C1 && C2 == VOX_C2 && C3 == VOX_C3;
}

extern "C" int mmego_vox_l1_fwd(void* stream, const float* X, long ldx, long rows, int K, const float* W, const float* bias, int C, float* Z,
                                float* rec) {
  MMEGO_REQUIRE(X && W && bias && Z && rec && rows > 0 && rows <= 65535L * 16 && K == VOX_K && C > 0 && C % 16 == 0 && ldx >= K &&
                ldx % 4 == 0 && (((uintptr_t)X | (uintptr_t)W) & 15) == 0);
  hipLaunchKernelGGL((vox_l1_fwd_kernel<VOX_K / 32, 9>), dim3((unsigned)cdiv(rows, 16), C / 16), dim3(128), 0, (hipStream_t)stream, X, ldx, W,
                     bias, Z, reinterpret_cast<float2*>(rec), (int)rows, C);
  MMEGO_LAUNCH_CHECK();
  return MMEGO_OK;
}

static inline BnRefD vox_bn(const void* bn, const float* rec, long rows) {
  BnRefD d = bnref_device(static_cast<const MmegoBnRefH*>(bn));
  d.rec = reinterpret_cast<const float2*>(rec);
  d.nrec = (int)cdiv(rows, 16);
  d.rpr = 16;
  return d;
}

extern "C" int mmego_vox_mid_fwd(void* stream, const float* Zin, const float* rec_in, const void* bn, long rows, int Cin, float* Yin,
                                 const float* W, const float* bias, int Cout, float* Zout, float* rec_out) {
  MMEGO_REQUIRE(Zin && rec_in && bn && Yin && W && bias && Zout && rec_out && rows > 1 && rows <= 65535L * 16 && ((uintptr_t)W & 15) == 0);
  const BnRefD d = vox_bn(bn, rec_in, rows);
  MMEGO_REQUIRE(d.gamma && d.beta && d.state);
  const dim3 grid((unsigned)cdiv(rows, 16));
  if (Cin == VOX_C1 && Cout == VOX_C2)
    hipLaunchKernelGGL((vox_mid_fwd_kernel<VOX_C1, VOX_C2>), grid, dim3(64 * (VOX_C2 / 16)), 0, (hipStream_t)stream, Zin, d, Yin, W, bias, Zout,
                       reinterpret_cast<float2*>(rec_out), (int)rows);
  else if (Cin == VOX_C2 && Cout == VOX_C3)
    hipLaunchKernelGGL((vox_mid_fwd_kernel<VOX_C2, VOX_C3>), grid, dim3(64 * (VOX_C3 / 16)), 0, (hipStream_t)stream, Zin, d, Yin, W, bias, Zout,
                       reinterpret_cast<float2*>(rec_out), (int)rows);
  else
    return MMEGO_EBADARG;
  MMEGO_LAUNCH_CHECK();
  return MMEGO_OK;
}

extern "C" int mmego_vox_out_fwd(void* stream, const float* Z, const float* rec, const void* bn, long rows, int C, float* Y, long ldy) {
  MMEGO_REQUIRE(Z && rec && bn && Y && rows > 1 && rows <= 65535L * 16 && C > 0 && C <= 128 && ldy >= C);
  const BnRefD d = vox_bn(bn, rec, rows);
  MMEGO_REQUIRE(d.gamma && d.beta && d.state);
  hipLaunchKernelGGL(vox_out_fwd_kernel, dim3((unsigned)cdiv(rows, 16)), dim3(256), 0, (hipStream_t)stream, Z, d, Y, ldy, (int)rows, C);
  MMEGO_LAUNCH_CHECK();
  return MMEGO_OK;
}

extern "C" int mmego_vox_bwd_sums(void* stream, const float* dY, long lddy, const float* Y, long ldy, const float* Z, const float* state,
                                  long rows, int C, float* G, float* prt) {
  MMEGO_REQUIRE(dY && Y && Z && state && G && prt && rows > 0 && rows <= 65535L * 16 && C > 0 && C <= 128 && lddy >= C && ldy >= C);
  hipLaunchKernelGGL(vox_bwd_sums_kernel, dim3((unsigned)cdiv(rows, 16)), dim3(256), 0, (hipStream_t)stream, dY, lddy, Y, ldy, Z, state, G,
                     reinterpret_cast<float2*>(prt), (int)rows, C);
  MMEGO_LAUNCH_CHECK();
  return MMEGO_OK;
}

extern "C" int mmego_vox_mid_bwd(void* stream, const float* G, const float* Z, const float* state, const float* prt, long rows, int Cout,
                                 float* dZ, float* dgamma, float* dbeta, const float* W, int Cin, const float* Yp, const float* Zp,
                                 const float* statep, float* Gp, float* prtp) {
  MMEGO_REQUIRE(G && Z && state && prt && dZ && dgamma && dbeta && W && Yp && Zp && statep && Gp && prtp && rows > 0 && rows <= 65535L * 16);
  const dim3 grid((unsigned)cdiv(rows, 16));
  const int nrt = (int)cdiv(rows, 16);
  if (Cout == VOX_C3 && Cin == VOX_C2)
    hipLaunchKernelGGL((vox_mid_bwd_kernel<VOX_C3, VOX_C2>), grid, dim3(64 * (VOX_C2 / 16)), 0, (hipStream_t)stream, G, Z, state,
                       reinterpret_cast<const float2*>(prt), nrt, dZ, dgamma, dbeta, W, Yp, Zp, statep, Gp, reinterpret_cast<float2*>(prtp), (int)rows);
  else if (Cout == VOX_C2 && Cin == VOX_C1)
    hipLaunchKernelGGL((vox_mid_bwd_kernel<VOX_C2, VOX_C1>), grid, dim3(64 * (VOX_C1 / 16)), 0, (hipStream_t)stream, G, Z, state,
                       reinterpret_cast<const float2*>(prt), nrt, dZ, dgamma, dbeta, W, Yp, Zp, statep, Gp, reinterpret_cast<float2*>(prtp), (int)rows);
  else
    return MMEGO_EBADARG;
  MMEGO_LAUNCH_CHECK();
  return MMEGO_OK;
}

extern "C" int mmego_vox_l1_bwd(void* stream, const float* G, const float* Z, const float* state, const float* prt, long rows, int C, float* dZ,
                                float* dgamma, float* dbeta, const float* W, int K, float* dX, long lddx) {
  MMEGO_REQUIRE(G && Z && state && prt && dZ && dgamma && dbeta && W && dX && rows > 0 && rows <= 65535L * 16 && C == VOX_C1 && K > 0 &&
                K % 192 == 0 && lddx >= K);
  hipLaunchKernelGGL((vox_l1_bwd_kernel<VOX_C1, 3>), dim3((unsigned)cdiv(rows, 16), K / 192), dim3(256), 0, (hipStream_t)stream, G, Z, state,
                     reinterpret_cast<const float2*>(prt), (int)cdiv(rows, 16), dZ, dgamma, dbeta, W, dX, lddx, (int)rows, K);
  MMEGO_LAUNCH_CHECK();
  return MMEGO_OK;
}

extern "C" int mmego_vox_dw(void* stream, long rows, const float* dZ1, const float* X, long ldx, float* dW1, int C1, int K, const float* dZ2,
                            const float* Y1, float* dW2, int C2, const float* dZ3, const float* Y2, float* dW3, int C3) {
  MMEGO_REQUIRE(rows > 0 && rows < (1L << 30) && dZ1 && X && dW1 && dZ2 && Y1 && dW2 && dZ3 && Y2 && dW3);
  MMEGO_REQUIRE(C1 > 0 && C2 > 0 && C3 > 0 && K > 0 && ((C1 | C2 | C3 | K) & 15) == 0 && ldx >= K);
  VoxDwP p;
  p.R = (int)rows;
  p.l[0] = VoxDwL{dZ1, X, ldx, dW1, C1, K, (C1 / 16) * (K / 16)};
  p.l[1] = VoxDwL{dZ2, Y1, (long)C1, dW2, C2, C1, p.l[0].tile_end + (C2 / 16) * (C1 / 16)};
  p.l[2] = VoxDwL{dZ3, Y2, (long)C2, dW3, C3, C2, p.l[1].tile_end + (C3 / 16) * (C2 / 16)};
  hipLaunchKernelGGL(vox_dw_kernel, dim3((unsigned)cdiv(p.l[2].tile_end, 4)), dim3(256), 0, (hipStream_t)stream, p);
  MMEGO_LAUNCH_CHECK();
  return MMEGO_OK;
}
