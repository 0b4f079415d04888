// fp32 GEMM on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32 fma chain).
//
// Two kernels:
//   gemm128_nt : C[M,N] = A[M,K] . W[N,K]^T (+bias)(relu), 128x128x16 tiles, float4 global loads,
//                double-buffered LDS.  The hot GEMMs of the path (IMU_Net LSTM input projections,
//                reference Net/IMU_Net.py:58-62) run here.
//   gemm64     : same product with arbitrary element strides on A, B and C, a batch dimension and
//                split-K slabs; 64x64x16 tiles.  Covers every Linear / k=1 conv / temporal conv of
//                Upper_Net / Lower_Net / GCN forward and backward (dX = dY.W, dW = dY^T.X).
//
// LDS tiles are k-major ([k][m]) so the MFMA operand read (lane l: row l&31, k l>>5) is one
// conflict-free ds_read_b32 per operand; the transposed ds_write_b32 is at most 2-way (free).
#include <stdlib.h>

#include "common.h"
#include "../../include/mmego_hip.h"       // MmegoGemmDesc (mmego_gemm_group)

#include "gemm_tile.h"

struct GemmP {
  const float* A;
  const float* B;
  float* C;
  const float* bias;
  long sam, sak, sbk, sbn, scm, scn;
  long sAb, sBb, sCb, sCs, sBiasb;
  int M, N, K;
  int nsplit, kchunk;
  int relu, accumulate;
  int avec, bvec;     // operand is k-contiguous with 16-B aligned rows: float4 loads allowed
  const float* cmul;  // optional elementwise multiplier in C's layout, applied last (a dropout mask on an input gradient)
  float* asum;        // optional: asum[b][m] (+)= sum_k A[b](m,k) -- the bias gradient beside a weight gradient dW = dY^T X
                      // (gemm32kq only; with split-K the slab sums go behind the product slabs and the reducer finishes them)
  // optional LSTM cell-backward epilogue (gemm32kq, unsplit, batch = direction; mmego_lstm_bwd_step): the product is dh_rec of the
  // step before, which only the cell backward of that step consumes -- it is applied to the tile's elements instead of storing C
  const float* cb_dout[2]; long cb_dos;
  const float* cb_gst[2]; const float* cb_cst[2]; const float* cb_cprev[2];
  float* cb_dc[2]; float* cb_dg[2]; long cb_dgs;
};

#define LD64 68
// kernel choice (ops.py mirrors both): the other operand orientations take the tile kernels from this many 64x64 work units; up to
// GEMM_KQ_MAX 64x64 workgroups a product runs on K-quartered 32x32 tiles
constexpr long GEMM_TILE_MIN_OTHER = 256;
constexpr int GEMM_KQ_MAX = 512;

// 64x64xBK tiles, BK = 64 (16 when the K range of a workgroup is shorter than 64).  These products are small
// (<= 1 GFLOP) and their k-loop is bound by the global-load latency of each step, not by the matrix cores (BK = 16:
// 0.4-0.5 us per step, 32 steps for K = 512), so the step is made as deep as LDS allows: 4x fewer latency-bound steps,
// 16 elements per operand per thread in flight, float4 loads where the operand is k-contiguous and 16-B aligned.
template <bool A_KC, bool B_KC, int BK>
__global__ __launch_bounds__(256) void gemm64_kernel(GemmP p) {
  constexpr int EL = BK / 4;                          // elements per thread per operand per step
  __shared__ float As[BK][LD64];
  __shared__ float Bs[BK][LD64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave & 1, wn = wave >> 1;
  const int batch = blockIdx.z / p.nsplit, split = blockIdx.z % p.nsplit;
  const int m0 = blockIdx.x * 64, n0 = blockIdx.y * 64;   // M tiles on grid.x (no 65535 limit: rows can be millions)
  const float* A = p.A + (long)batch * p.sAb;
  const float* B = p.B + (long)batch * p.sBb;
  const int kbeg = split * p.kchunk;
  const int kend = min(p.K, kbeg + p.kchunk);

  // staging coordinates.  k-contiguous operand: row tid>>2, k = 4 (tid&3) + 16 g + c (g < EL/4, c < 4): four lanes cover
  // 64 contiguous bytes; m-contiguous operand: m = tid&63, k = (tid>>6) + 4 j: a wave covers 256 contiguous bytes.
  const int am = A_KC ? (tid >> 2) : (tid & 63), bn = B_KC ? (tid >> 2) : (tid & 63);
  const int akb = A_KC ? (tid & 3) * 4 : (tid >> 6), bkb = B_KC ? (tid & 3) * 4 : (tid >> 6);
  // rows / columns past M / N: CLAMPED (their products land in output elements that are never stored); k past the slab: loaded from
  // a clamped k and zeroed afterwards -- only in a chunk that is not whole (a workgroup-uniform test).  A predicate on a load itself
  // costs a branch and a full s_waitcnt vmcnt(0) per load (see gemm32kq_kernel).
  const float* Arow = A + (long)min(m0 + am, p.M - 1) * p.sam;
  const float* Bcol = B + (long)min(n0 + bn, p.N - 1) * p.sbn;
  const bool avec = A_KC && p.avec, bvec = B_KC && p.bvec;

  float ra[EL], rb[EL];
#define G64_PIN(v) asm volatile("" : "+v"(v))
#define G64_LOAD1(R, PTR, SK, KC, VEC, KB, k0)                                                                \
  do {                                                                                                        \
    const bool whole_ = (k0) + BK <= kend;                     /* uniform */                                  \
    if (KC) {                                                                                                 \
      if ((VEC) && whole_) {                                                                                  \
        _Pragma("unroll") for (int g = 0; g < EL / 4; ++g) {                                                  \
          const f32x4 v = *reinterpret_cast<const f32x4*>((PTR) + (k0) + (KB) + 16 * g);                      \
          R[4 * g] = v.x; R[4 * g + 1] = v.y; R[4 * g + 2] = v.z; R[4 * g + 3] = v.w;                         \
        }                                                                                                     \
      } else {                                                                                                \
        _Pragma("unroll") for (int j = 0; j < EL; ++j)                                                        \
          R[j] = (PTR)[(long)min((k0) + (KB) + 16 * (j >> 2) + (j & 3), kend - 1) * (SK)];                    \
        if (!whole_) {                                                                                        \
          _Pragma("unroll") for (int j = 0; j < EL; ++j) {                                                    \
            G64_PIN(R[j]);                                                                                    \
            R[j] = ((k0) + (KB) + 16 * (j >> 2) + (j & 3) < kend) ? R[j] : 0.0f;                              \
          }                                                                                                   \
        }                                                                                                     \
      }                                                                                                       \
    } else {                                                                                                  \
      _Pragma("unroll") for (int j = 0; j < EL; ++j) R[j] = (PTR)[(long)min((k0) + (KB) + 4 * j, kend - 1) * (SK)]; \
      if (!whole_) {                                                                                          \
        _Pragma("unroll") for (int j = 0; j < EL; ++j) {                                                      \
          G64_PIN(R[j]);                                                                                      \
          R[j] = ((k0) + (KB) + 4 * j < kend) ? R[j] : 0.0f;                                                  \
        }                                                                                                     \
      }                                                                                                       \
    }                                                                                                         \
  } while (0)
#define G64_LOAD(k0)                                                                                          \
  do {                                                                                                        \
    G64_LOAD1(ra, Arow, p.sak, A_KC, avec, akb, k0);                                                          \
    G64_LOAD1(rb, Bcol, p.sbk, B_KC, bvec, bkb, k0);                                                          \
  } while (0)

  f32x16 acc = {0};
  if (kbeg < kend) G64_LOAD(kbeg);
  for (int k0 = kbeg; k0 < kend; k0 += BK) {
    __syncthreads();
#pragma unroll
    for (int j = 0; j < EL; ++j) {
      As[A_KC ? akb + 16 * (j >> 2) + (j & 3) : akb + 4 * j][am] = ra[j];
      Bs[B_KC ? bkb + 16 * (j >> 2) + (j & 3) : bkb + 4 * j][bn] = rb[j];
    }
    __syncthreads();
    if (k0 + BK < kend) G64_LOAD(k0 + BK);
    const int r = lane & 31, h = lane >> 5;
#pragma unroll
    for (int kk = 0; kk < BK; kk += 2) {
      float a = As[kk + h][wm * 32 + r];
      float b = Bs[kk + h][wn * 32 + r];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
  }
#undef G64_LOAD
#undef G64_LOAD1
#undef G64_PIN

  // epilogue: lane holds column (lane&31), rows (reg&3)+8*(reg>>2)+4*(lane>>5).  Whatever the options read (accumulate: C itself,
  // cmul: the mask / factor tensor) is fetched for all 16 elements first, from clamped addresses, and only then do the stores
  // begin: a load between two stores waits for the store in front of it as well (one counter for both on gfx9) -- the 16
  // elements were 16 consecutive memory round trips.
  const int col = n0 + wn * 32 + (lane & 31);
  if (col >= p.N) return;
  float* C = p.C + (long)batch * p.sCb + (long)split * p.sCs;
  const float bv = (p.bias && p.nsplit == 1) ? p.bias[(long)batch * p.sBiasb + col] : 0.0f;
  const int rbase = m0 + wm * 32 + 4 * (lane >> 5);
  const bool final_ = p.nsplit == 1;
  float old[16], cm[16];
  if (final_ && p.accumulate) {
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int row = rbase + (reg & 3) + 8 * (reg >> 2);
      old[reg] = C[(long)(row < p.M ? row : p.M - 1) * p.scm + (long)col * p.scn];
    }
  }
  if (final_ && p.cmul) {
    const float* cmp = p.cmul + (C - p.C);
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int row = rbase + (reg & 3) + 8 * (reg >> 2);
      cm[reg] = cmp[(long)(row < p.M ? row : p.M - 1) * p.scm + (long)col * p.scn];
    }
  }
#pragma unroll
  for (int reg = 0; reg < 16; ++reg) {
    float v = acc[reg] + bv;
    if (final_) {
      if (p.relu == 1) v = fmaxf(v, 0.0f);
      if (p.accumulate) { asm volatile("" : "+v"(old[reg])); v += old[reg]; }
      if (p.cmul) { asm volatile("" : "+v"(cm[reg])); v = p.relu == 2 ? (cm[reg] > 0.f ? v : 0.f) : v * cm[reg]; }
    }
    acc[reg] = v;
  }
  float* cp = C + (long)rbase * p.scm + (long)col * p.scn;
  if (m0 + wm * 32 + 32 <= p.M) {                          // whole 32-row part: stores without a predicate each
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) cp[(long)((reg & 3) + 8 * (reg >> 2)) * p.scm] = acc[reg];
  } else {
#pragma unroll
    for (int reg = 0; reg < 16; ++reg)
      if (rbase + (reg & 3) + 8 * (reg >> 2) < p.M) cp[(long)((reg & 3) + 8 * (reg >> 2)) * p.scm] = acc[reg];
  }
}

// ---- K-quartered 32x32 tiles for products with few output tiles ------------------------------------------------------
// With at most a few hundred 64x64 tiles the chip is mostly idle and a workgroup's run time is its serial chain of k-steps, each
// bound by a global-load round trip.  Here a workgroup owns a 32x32 output tile (4x more workgroups) and its four waves
// each take a QUARTER of the K range (4x shorter chain); the four partial tiles are summed through LDS in a fixed order.
// Operands go straight from global memory into the MFMA operand layout, no LDS staging: in a 32-k chunk lane (r, h) takes
// k = k0 + 16 h + s at MFMA step s for BOTH operands (any k order is valid as long as A and B agree), which makes a
// k-contiguous operand four 16-B loads per lane and an m-contiguous operand 16 loads that are contiguous across lanes.
template <bool A_KC, bool B_KC>
__device__ __forceinline__ void gemm32kq_body(const GemmP& p, const int bx, const int by, const int bz) {
  __shared__ float red[4][32][33];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int batch = bz / p.nsplit, split = bz % p.nsplit;
  const int m0 = bx * 32, n0 = by * 32;
  const float* A = p.A + (long)batch * p.sAb;
  const float* B = p.B + (long)batch * p.sBb;
  const int kbeg0 = split * p.kchunk;
  const int kend0 = min(p.K, kbeg0 + p.kchunk);
  const int kq = (((kend0 - kbeg0) + 3) / 4 + 31) / 32 * 32;      // k per wave, a multiple of the 32-k chunk
  const int kbeg = kbeg0 + wave * kq;
  const int kend = min(kend0, kbeg + kq);
  // Rows / columns past M / N are CLAMPED, not predicated: their products land in output elements that are never stored, and a
  // predicate on the load itself costs a branch and a full s_waitcnt vmcnt(0) per load (the compiler will not speculate it).
  // Only k has to be masked, and only in a wave's last partial chunk: the vector path is taken by whole chunks (a wave-uniform
  // test), the scalar path loads from clamped k and zeroes the tail afterwards.
  const float* Arow = A + (long)min(m0 + r, p.M - 1) * p.sam;
  const float* Bcol = B + (long)min(n0 + r, p.N - 1) * p.sbn;
  const bool avec = A_KC && p.avec, bvec = B_KC && p.bvec;

  float ra[16], rb[16], na[16], nb[16];
#define KQ_PIN(v) asm volatile("" : "+v"(v))
#define KQ_LOAD1(R, PTR, SK, VEC, k0)                                                                         \
  do {                                                                                                        \
    const int kl = (k0) + 16 * h;                                                                             \
    if ((VEC) && (k0) + 32 <= kend) {                                                                         \
      _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                                         \
        const f32x4 v = *reinterpret_cast<const f32x4*>((PTR) + kl + 4 * g);                                  \
        R[4 * g] = v.x; R[4 * g + 1] = v.y; R[4 * g + 2] = v.z; R[4 * g + 3] = v.w;                           \
      }                                                                                                       \
    } else {                                                                                                  \
      _Pragma("unroll") for (int s_ = 0; s_ < 16; ++s_) R[s_] = (PTR)[(long)min(kl + s_, kend - 1) * (SK)];   \
      if ((k0) + 32 > kend) {                                                                                 \
        _Pragma("unroll") for (int s_ = 0; s_ < 16; ++s_) { KQ_PIN(R[s_]); R[s_] = (kl + s_ < kend) ? R[s_] : 0.0f; } \
      }                                                                                                       \
    }                                                                                                         \
  } while (0)
#define KQ_LOAD(RA, RB, k0)                                                                                   \
  do {                                                                                                        \
    KQ_LOAD1(RA, Arow, p.sak, avec, k0);                                                                      \
    KQ_LOAD1(RB, Bcol, p.sbk, bvec, k0);                                                                      \
  } while (0)

  f32x16 acc = {0};
  const bool want_asum = p.asum != nullptr && by == 0;
  float asum = 0.f;
  if (kbeg < kend) {
    KQ_LOAD(ra, rb, kbeg);
    for (int k0 = kbeg; k0 < kend; k0 += 32) {
      const bool more = k0 + 32 < kend;
      if (more) KQ_LOAD(na, nb, k0 + 32);
      if (want_asum) {
#pragma unroll
        for (int s_ = 0; s_ < 16; ++s_) asum += ra[s_];
      }
#pragma unroll
      for (int s_ = 0; s_ < 16; ++s_) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ra[s_], rb[s_], acc, 0, 0, 0);
      if (more) {
#pragma unroll
        for (int s_ = 0; s_ < 16; ++s_) { ra[s_] = na[s_]; rb[s_] = nb[s_]; }
      }
    }
  }
#undef KQ_LOAD
#undef KQ_LOAD1
#undef KQ_PIN
  // partial tiles -> LDS; C layout of the 32x32 MFMA: lane holds column (lane&31), rows (reg&3) + 8 (reg>>2) + 4 (lane>>5)
#pragma unroll
  for (int reg = 0; reg < 16; ++reg) red[wave][(reg & 3) + 8 * (reg >> 2) + 4 * h][r] = acc[reg];
  __shared__ float ared[4][2][32];
  if (want_asum) ared[wave][h][r] = asum;
  __syncthreads();
  if (want_asum && tid < 32 && m0 + tid < p.M) {          // the eight partial row sums in a fixed order
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) v += ared[w][0][tid] + ared[w][1][tid];
    if (p.nsplit == 1) {
      float* dst = p.asum + (long)batch * p.M + m0 + tid;
      *dst = p.accumulate ? *dst + v : v;
    } else {                                               // slab sums: [split][batch][M] behind the product slabs
      p.asum[((long)split * (gridDim.z / p.nsplit) + batch) * p.M + m0 + tid] = v;
    }
  }
  if (p.cb_dg[0]) {
    // LSTM cell backward on the tile (the expressions of lstm_cell_bwd_kernel, imu_train.hip: same bits): whole tiles only (the
    // launcher checks), N = H; every load of the thread's four elements before its first store
    const int col = tid & 31, H = p.N, d = batch;
    float dh[4], gi[4], gf[4], gg[4], go[4], cc[4], cp[4], dcin[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int row = (tid >> 5) + 8 * e;
      const long r = m0 + row, j = n0 + col, i = r * H + j;
      const float rec = ((red[0][row][col] + red[1][row][col]) + red[2][row][col]) + red[3][row][col];
      const float* gs = p.cb_gst[d] + r * 4 * H + j;
      dh[e] = p.cb_dout[d][r * p.cb_dos + j] + rec;
      gi[e] = gs[0]; gf[e] = gs[H]; gg[e] = gs[2 * H]; go[e] = gs[3 * H];
      cc[e] = p.cb_cst[d][i];
      cp[e] = p.cb_cprev[d] ? p.cb_cprev[d][i] : 0.f;
      dcin[e] = p.cb_dc[d][i];
    }
    float o0[4], o1[4], o2[4], o3[4], dco[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      asm volatile("" : "+v"(gi[e]), "+v"(gf[e]), "+v"(gg[e]), "+v"(go[e]), "+v"(cc[e]), "+v"(cp[e]), "+v"(dcin[e]), "+v"(dh[e]));
      const float tc = tanhf(cc[e]);
      const float dcv = dcin[e] + dh[e] * go[e] * (1.f - tc * tc);
      o0[e] = dcv * gg[e] * gi[e] * (1.f - gi[e]);
      o1[e] = dcv * cp[e] * gf[e] * (1.f - gf[e]);
      o2[e] = dcv * gi[e] * (1.f - gg[e] * gg[e]);
      o3[e] = dh[e] * tc * go[e] * (1.f - go[e]);
      dco[e] = dcv * gf[e];
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int row = (tid >> 5) + 8 * e;
      const long r = m0 + row, j = n0 + col;
      float* dg = p.cb_dg[d] + r * p.cb_dgs + j;
      dg[0] = o0[e]; dg[H] = o1[e]; dg[2 * H] = o2[e]; dg[3 * H] = o3[e];
      p.cb_dc[d][r * H + j] = dco[e];
    }
    return;
  }
  // (options' loads for all four elements first, from clamped addresses, then the stores: see gemm64_kernel's epilogue)
  float* C = p.C + (long)batch * p.sCb + (long)split * p.sCs;
  const bool final_ = p.nsplit == 1;
  const int col = tid & 31, colc = n0 + col < p.N ? n0 + col : p.N - 1;
  long off[4];
  float v[4], old[4], cm[4], bv = 0.f;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int row = (tid >> 5) + 8 * e;
    v[e] = ((red[0][row][col] + red[1][row][col]) + red[2][row][col]) + red[3][row][col];
    off[e] = (long)(m0 + row < p.M ? m0 + row : p.M - 1) * p.scm + (long)colc * p.scn;
  }
  if (final_ && p.bias) bv = p.bias[(long)batch * p.sBiasb + colc];
  if (final_ && p.accumulate) {
#pragma unroll
    for (int e = 0; e < 4; ++e) old[e] = C[off[e]];
  }
  if (final_ && p.cmul) {
    const float* cmp = p.cmul + (C - p.C);
#pragma unroll
    for (int e = 0; e < 4; ++e) cm[e] = cmp[off[e]];
  }
  if (final_) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float x = v[e] + bv;
      if (p.relu == 1) x = fmaxf(x, 0.0f);
      if (p.accumulate) { asm volatile("" : "+v"(old[e])); x += old[e]; }
      if (p.cmul) { asm volatile("" : "+v"(cm[e])); x = p.relu == 2 ? (cm[e] > 0.f ? x : 0.f) : x * cm[e]; }
      v[e] = x;
    }
  }
  if (m0 + 32 <= p.M && n0 + 32 <= p.N) {                  // whole tile: stores without a predicate each
#pragma unroll
    for (int e = 0; e < 4; ++e) C[off[e]] = v[e];
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (m0 + (tid >> 5) + 8 * e < p.M && n0 + col < p.N) C[off[e]] = v[e];
  }
}

template <bool A_KC, bool B_KC>
__global__ __launch_bounds__(256) void gemm32kq_kernel(GemmP p) {
  gemm32kq_body<A_KC, B_KC>(p, blockIdx.x, blockIdx.y, blockIdx.z);
}

// Several INDEPENDENT small products in one launch (mmego_gemm_group): grid.z is cut into the products' own z ranges, grid.x / y
// cover the largest tile counts.  For leaves of a backward pass -- weight gradients nothing else reads -- that would otherwise sit
// one behind the other between the kernels of a dependent chain.
#define GEMM_GROUP_MAX 10
struct GemmGroup { GemmP p[GEMM_GROUP_MAX]; int zend[GEMM_GROUP_MAX]; int n; };
template <bool A_KC, bool B_KC>
__global__ __launch_bounds__(256) void gemm32kq_group_kernel(GemmGroup g) {
  int i = 0;
  const int z = blockIdx.z;
  while (i + 1 < g.n && z >= g.zend[i]) ++i;              // (uniform)
  const int z0 = i ? g.zend[i - 1] : 0;
  const GemmP& p = g.p[i];
  if ((int)blockIdx.x * 32 >= p.M || (int)blockIdx.y * 32 >= p.N) return;
  gemm32kq_body<A_KC, B_KC>(p, blockIdx.x, blockIdx.y, z - z0);
}

// ws: [nsplit][nbatch][M][N] contiguous partial products -> C (strided), + bias, relu, accumulate.
// 16 outputs x 16 split-lanes per block: lane group kg sums slabs kg, kg+16, ... (loads unrolled 8 deep: the loop is
// load-latency bound), then the 16 partial sums are combined through LDS in a fixed order (deterministic).
#define SKR_OUT 16
#define SKR_LANES 16
// asum_ws / asum_out (optional): [nsplit][nbatch*M] slab row sums of A (mmego_gemm's asum) -> asum_out[nbatch*M], reduced by the
// blocks behind the product's own (the same lane structure, the same fixed order).
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ ws, float* C, const float* bias, int nsplit,
                                                            int nbatch, int M, int N, long scm, long scn, long sCb,
                                                            int relu, int accumulate, const float* __restrict__ asum_ws,
                                                            float* asum_out) {
  __shared__ float sh[SKR_LANES][SKR_OUT + 1];
  long total = (long)nbatch * M * N;
  const int o = threadIdx.x & (SKR_OUT - 1), kg = threadIdx.x / SKR_OUT;
  long i = (long)blockIdx.x * SKR_OUT + o;
  const long cblocks = (total + SKR_OUT - 1) / SKR_OUT;
  const bool is_asum = (long)blockIdx.x >= cblocks;        // (uniform per block)
  if (is_asum) {
    i = ((long)blockIdx.x - cblocks) * SKR_OUT + o;
    total = (long)nbatch * M;
    ws = asum_ws;
  }
  float s = 0.0f;
  if (i < total) {
    const float* src = ws + i;
    int k = kg;
#pragma unroll 1
    for (; k + 7 * SKR_LANES < nsplit; k += 8 * SKR_LANES) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = src[(long)(k + u * SKR_LANES) * total];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; k < nsplit; k += SKR_LANES) s += src[(long)k * total];
  }
  sh[kg][o] = s;
  __syncthreads();
  if (kg == 0 && i < total) {
    s = sh[0][o];
#pragma unroll
    for (int g = 1; g < SKR_LANES; ++g) s += sh[g][o];
    if (is_asum) {
      asum_out[i] = accumulate ? asum_out[i] + s : s;
      return;
    }
    int n = (int)(i % N);
    long r = i / N;
    int m = (int)(r % M);
    int b = (int)(r / M);
    if (bias) s += bias[n];
    if (relu) s = fmaxf(s, 0.0f);
    float* dst = C + (long)b * sCb + (long)m * scm + (long)n * scn;
    if (accumulate) s += *dst;
    *dst = s;
  }
}

// ---------------------------------------------------------------------------------------------
#define LD128 132

__global__ __launch_bounds__(256) void gemm128_nt_kernel(const float* __restrict__ A, const float* __restrict__ W,
                                                         float* __restrict__ C, const float* __restrict__ bias,
                                                         int M, int N, int K, long lda, long ldw, long ldc, int relu) {
  __shared__ float As[2][16][LD128];
  __shared__ float Bs[2][16][LD128];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave & 1, wn = wave >> 1;
  // XCD-aware tile order: consecutive tiles along N (sharing the A row panel) stay on one XCD's L2
  const int ntn = N / 128, ntm = M / 128, nwg = ntn * ntm;
  int id = blockIdx.x;
  if ((nwg & 7) == 0) id = (id & 7) * (nwg >> 3) + (id >> 3);
  const int m0 = (id / ntn) * 128, n0 = (id % ntn) * 128;

  const int lr = tid >> 2, lk = (tid & 3) * 4;
  const float* Ap = A + (long)(m0 + lr) * lda + lk;
  const float* Wp = W + (long)(n0 + lr) * ldw + lk;
  float4 ra[2], rb[2];
  auto gload = [&](int k0) {
    ra[0] = *reinterpret_cast<const float4*>(Ap + k0);
    ra[1] = *reinterpret_cast<const float4*>(Ap + 64 * lda + k0);
    rb[0] = *reinterpret_cast<const float4*>(Wp + k0);
    rb[1] = *reinterpret_cast<const float4*>(Wp + 64 * ldw + k0);
  };
  auto sstore = [&](int buf) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      As[buf][lk + 0][lr + 64 * h] = ra[h].x; As[buf][lk + 1][lr + 64 * h] = ra[h].y;
      As[buf][lk + 2][lr + 64 * h] = ra[h].z; As[buf][lk + 3][lr + 64 * h] = ra[h].w;
      Bs[buf][lk + 0][lr + 64 * h] = rb[h].x; Bs[buf][lk + 1][lr + 64 * h] = rb[h].y;
      Bs[buf][lk + 2][lr + 64 * h] = rb[h].z; Bs[buf][lk + 3][lr + 64 * h] = rb[h].w;
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = (f32x16){0};

  const int nk = K / 16;
  gload(0);
  sstore(0);
  __syncthreads();
  const int r = lane & 31, h = lane >> 5;
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) gload((kt + 1) * 16);
#pragma unroll
    for (int kk = 0; kk < 16; kk += 2) {
      float a0 = As[buf][kk + h][wm * 64 + r], a1 = As[buf][kk + h][wm * 64 + 32 + r];
      float b0 = Bs[buf][kk + h][wn * 64 + r], b1 = Bs[buf][kk + h][wn * 64 + 32 + r];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
    }
    if (kt + 1 < nk) sstore(buf ^ 1);
    __syncthreads();
  }

#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = n0 + wn * 64 + j * 32 + (lane & 31);
    const float bv = bias ? bias[col] : 0.0f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        int row = m0 + wm * 64 + i * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
        float v = acc[i][j][reg] + bv;
        if (relu) v = fmaxf(v, 0.0f);
        C[(long)row * ldc + col] = v;
      }
    }
  }
}

extern "C" int mmego_gemm(void* stream, const float* A, long sam, long sak, const float* B, long sbk, long sbn,
                          float* C, long scm, long scn, const float* bias, int M, int N, int K, int nbatch, long sAb,
                          long sBb, long sCb, int relu, int accumulate, float* splitk_ws, int nsplit, long sBiasb,
                          const float* cmul, float* asum) {
  MMEGO_REQUIRE(A && B && C && M > 0 && N > 0 && K > 0 && nbatch > 0 && nsplit >= 1);
  // accumulate == 2: DEFERRED split-K -- the slabs (and the slab row sums behind them) stay in splitk_ws[nsplit][M*N] (+ [nsplit][M])
  // and C is not touched; mmego_slab_reduce (kind 0) adds them later, together with a backward pass's other partial products
  const bool defer = accumulate == 2;
  MMEGO_REQUIRE(!defer || (nsplit > 1 && nbatch == 1 && !bias && !relu));
  MMEGO_REQUIRE(!cmul || nsplit == 1);
  MMEGO_REQUIRE(relu == 0 || relu == 1 || (relu == 2 && cmul));
  MMEGO_REQUIRE(sBiasb == 0 || nsplit == 1);          // (the split-K reducer applies one shared bias)
  hipStream_t st = (hipStream_t)stream;
  // Large-tile kernels (gemm_tile.hip): 64-aligned shapes with enough work units to be worth a 64x64+ tile, any operand
  // orientation whose contiguous index has unit stride and 16-B aligned rows.
  {
    const bool a_kc = sak == 1, a_mc = sam == 1 && !a_kc;
    const bool b_kc = sbk == 1, b_mc = sbn == 1 && !b_kc;
    const long lda = a_kc ? sam : sak, ldw = b_kc ? sbn : sbk;
    const int kchunk_t = nsplit > 1 ? cdiv(cdiv(K, nsplit), 64) * 64 : K;
    // k-contiguous x k-contiguous products take the tile kernels at any size (measured better than the K-quartered kernel
    // even for 32 tiles); the other orientations only when there are enough 64x64 work units to fill the chip.
    constexpr long tile_min_other = GEMM_TILE_MIN_OTHER;
    // (a long-K product with few tiles is a serial chain of load-latency-bound chunks: the K-quartered kernel is better)
    const long tile_min_units = (a_kc && b_kc && nsplit == 1) ? (K <= 1024 ? 1 : 200) : tile_min_other;
    const long units64 = (long)(M / 64) * (N / 64) * nsplit * nbatch;
    const bool ok = !cmul && !asum && scn == 1 && (nbatch == 1 || ((sAb % 4) == 0 && (sBb % 4) == 0)) && (a_kc || a_mc) && (b_kc || b_mc) && (lda % 4) == 0 && (ldw % 4) == 0 &&
                    (((uintptr_t)A | (uintptr_t)B) & 15) == 0 && (M % 64) == 0 && (N % 64) == 0 &&
                    (K % 64) == 0 && units64 >= tile_min_units && (nsplit == 1 || (long)(nsplit - 1) * kchunk_t < K);
    if (ok) {
      TileP tp;
      tp.A = A; tp.W = B; tp.C = C; tp.bias = bias;
      tp.M = M; tp.N = N; tp.K = K;
      tp.lda = lda; tp.ldw = ldw; tp.ldc = scm;
      tp.relu = relu; tp.accumulate = accumulate;
      tp.nsplit = nsplit; tp.kchunk = kchunk_t; tp.ws = splitk_ws;
      tp.nbatch = nbatch; tp.sAb = sAb; tp.sWb = sBb; tp.sCb = sCb; tp.sBiasb = sBiasb;
      int rc = mmego_detail::gemm_tile_launch(st, tp, a_kc, b_kc);
      if (rc == 0 && nsplit > 1 && !defer) {
        long total = (long)nbatch * M * N;
        int blocks = (int)((total + SKR_OUT - 1) / SKR_OUT);
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, st, splitk_ws, C, bias, nsplit, nbatch, M, N, scm, scn,
                           sCb, relu, accumulate, (const float*)nullptr, (float*)nullptr);
        MMEGO_LAUNCH_CHECK();
      }
      if (rc != -2) return rc;
    }
  }
  const bool fast = !cmul && !asum && nbatch == 1 && nsplit == 1 && !accumulate && sak == 1 && sbk == 1 && scn == 1 && (M % 128) == 0 &&
                    (N % 128) == 0 && (K % 16) == 0 && (sam % 4) == 0 && (sbn % 4) == 0 &&
                    (((uintptr_t)A | (uintptr_t)B) & 15) == 0;
  if (fast) {
    dim3 grid((M / 128) * (N / 128));
    hipLaunchKernelGGL(gemm128_nt_kernel, grid, dim3(256), 0, st, A, B, C, bias, M, N, K, sam, sbn, scm, relu);
    MMEGO_LAUNCH_CHECK();
    return MMEGO_OK;
  }
  GemmP p;
  p.A = A; p.B = B; p.bias = bias;
  p.sam = sam; p.sak = sak; p.sbk = sbk; p.sbn = sbn;
  p.sAb = sAb; p.sBb = sBb; p.sBiasb = sBiasb;
  p.M = M; p.N = N; p.K = K;
  p.nsplit = nsplit;
  p.relu = relu; p.accumulate = accumulate;
  p.cmul = cmul;
  p.asum = asum;
  p.cb_dg[0] = p.cb_dg[1] = nullptr;
  if (nsplit > 1) {
    MMEGO_REQUIRE(splitk_ws != nullptr);
    int kc = cdiv(K, nsplit);
    p.kchunk = cdiv(kc, 16) * 16;
    p.C = splitk_ws; p.scm = N; p.scn = 1; p.sCb = (long)M * N; p.sCs = (long)nbatch * M * N;
    if (asum) p.asum = splitk_ws + (long)nsplit * nbatch * M * N;      // slab row sums behind the product slabs
  } else {
    p.kchunk = cdiv(K, 16) * 16;
    p.C = C; p.scm = scm; p.scn = scn; p.sCb = sCb; p.sCs = 0;
  }
  MMEGO_REQUIRE(cdiv(N, 32) <= 65535 && (long)nbatch * nsplit <= 65535);
  const long wgs64 = (long)cdiv(M, 64) * cdiv(N, 64) * nbatch * nsplit;
  constexpr int kq_max = GEMM_KQ_MAX;
  const bool kq = wgs64 <= kq_max && p.kchunk >= 64;     // few tiles and a k-chain worth cutting: K-quartered 32x32 tiles
  MMEGO_REQUIRE(!asum || kq);                            // (row sums of A: the K-quartered kernel only; callers check with ops)
  dim3 grid(cdiv(M, kq ? 32 : 64), cdiv(N, kq ? 32 : 64), nbatch * nsplit);
  const bool akc = (sak == 1), bkc = (sbk == 1);
  p.avec = akc && (sam % 4) == 0 && (sAb % 4) == 0 && ((uintptr_t)A & 15) == 0;
  p.bvec = bkc && (sbn % 4) == 0 && (sBb % 4) == 0 && ((uintptr_t)B & 15) == 0;
  const bool deep = p.kchunk >= 64;                  // K range per workgroup
#define G64_LAUNCH(AK, BKC)                                                                        \
  do {                                                                                             \
    if (kq) hipLaunchKernelGGL((gemm32kq_kernel<AK, BKC>), grid, dim3(256), 0, st, p);             \
    else if (deep) hipLaunchKernelGGL((gemm64_kernel<AK, BKC, 64>), grid, dim3(256), 0, st, p);    \
    else hipLaunchKernelGGL((gemm64_kernel<AK, BKC, 16>), grid, dim3(256), 0, st, p);              \
  } while (0)
  if (akc && bkc) G64_LAUNCH(true, true);
  else if (akc) G64_LAUNCH(true, false);
  else if (bkc) G64_LAUNCH(false, true);
  else G64_LAUNCH(false, false);
#undef G64_LAUNCH
  MMEGO_LAUNCH_CHECK();
  if (nsplit > 1 && !defer) {
    long total = (long)nbatch * M * N;
    int blocks = (int)((total + SKR_OUT - 1) / SKR_OUT);
    if (asum) blocks += (int)(((long)nbatch * M + SKR_OUT - 1) / SKR_OUT);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, st, splitk_ws, C, bias, nsplit, nbatch, M, N,
                       scm, scn, sCb, relu, accumulate, (const float*)p.asum, asum);
    MMEGO_LAUNCH_CHECK();
  }
  return MMEGO_OK;
}

int lstm_bwd_step_dma_try(void* stream, int Bn, int H, const float* dg0, const float* dg1, long dgs, const float* wT0,
                                           const float* wT1, const float* dout0, const float* dout1, long dos, const float* gst0,
                                           const float* gst1, const float* cst0, const float* cst1, const float* cprev0,
                                           const float* cprev1, float* dc0, float* dc1, float* dgo0, float* dgo1);

// One step of the backward recurrence of a BiLSTM layer (stage-1 training, reference autograd of nn.LSTM): for both directions
// dh_rec = dgates_s . W_hh (A = dgates of step s, rows Bn, K = 4H; wT = W_hh transposed [H][4H]) and, on the product's tiles, the
// cell backward of the step BEFORE in time order of the backward pass (dh = dout + dh_rec, gate / cell gradients from the forward
// stashes) -- one launch instead of product + lstm_cell_backward, and dh_rec never reaches memory.  Bn, H multiples of 32.
extern "C" int mmego_lstm_bwd_step(void* stream, int Bn, int H, const float* dg0, const float* dg1, long dgs, const float* wT0,
                                   const float* wT1, const float* dout0, const float* dout1, long dos, const float* gst0,
                                   const float* gst1, const float* cst0, const float* cst1, const float* cprev0,
                                   const float* cprev1, float* dc0, float* dc1, float* dgo0, float* dgo1) {
  MMEGO_REQUIRE(Bn > 0 && H > 0 && (Bn % 32) == 0 && (H % 32) == 0 && dg0 && dg1 && wT0 && wT1 && dout0 && dout1 && gst0 && gst1 &&
                cst0 && cst1 && dc0 && dc1 && dgo0 && dgo1 && (cprev0 == nullptr) == (cprev1 == nullptr));
  MMEGO_REQUIRE((dgs % 4) == 0 && ((((uintptr_t)dg0) | ((uintptr_t)dg1) | ((uintptr_t)wT0) | ((uintptr_t)wT1)) & 15) == 0);
  {   // 64-row multiples: the LDS-DMA step kernel (lstm_bwd_step.hip); anything else: the K-quartered small-tile product below
    const int rc = lstm_bwd_step_dma_try(stream, Bn, H, dg0, dg1, dgs, wT0, wT1, dout0, dout1, dos, gst0, gst1, cst0, cst1,
                                               cprev0, cprev1, dc0, dc1, dgo0, dgo1);
    if (rc != -1) return rc;
  }
  GemmP p;
  p.A = dg0; p.B = wT0; p.C = nullptr; p.bias = nullptr;
  p.sam = dgs; p.sak = 1; p.sbk = 1; p.sbn = 4L * H;
  p.sAb = dg1 - dg0; p.sBb = wT1 - wT0; p.sBiasb = 0;
  p.scm = H; p.scn = 1; p.sCb = 0; p.sCs = 0;
  p.M = Bn; p.N = H; p.K = 4 * H;
  p.nsplit = 1; p.kchunk = cdiv(p.K, 16) * 16;
  p.relu = 0; p.accumulate = 0;
  p.cmul = nullptr; p.asum = nullptr;
  p.avec = (p.sAb % 4) == 0; p.bvec = (p.sBb % 4) == 0;
  p.cb_dout[0] = dout0; p.cb_dout[1] = dout1; p.cb_dos = dos;
  p.cb_gst[0] = gst0; p.cb_gst[1] = gst1; p.cb_cst[0] = cst0; p.cb_cst[1] = cst1;
  p.cb_cprev[0] = cprev0; p.cb_cprev[1] = cprev1;
  p.cb_dc[0] = dc0; p.cb_dc[1] = dc1; p.cb_dg[0] = dgo0; p.cb_dg[1] = dgo1; p.cb_dgs = dgs;
  dim3 grid(Bn / 32, H / 32, 2);
  hipLaunchKernelGGL((gemm32kq_kernel<true, true>), grid, dim3(256), 0, (hipStream_t)stream, p);
  MMEGO_LAUNCH_CHECK();
  return MMEGO_OK;
}

// Several independent products in one launch where all of them are small-tile (K-quartered) products of one operand orientation;
// otherwise one mmego_gemm call each, in order -- the results are the same either way.
extern "C" int mmego_gemm_group(void* stream, int n, const MmegoGemmDesc* d) {
  MMEGO_REQUIRE(n >= 1 && d);
  constexpr long tile_min_other = GEMM_TILE_MIN_OTHER;
  constexpr int kq_max = GEMM_KQ_MAX;
  bool ok = n >= 2 && n <= GEMM_GROUP_MAX;
  const bool akc = d[0].sak == 1, bkc = d[0].sbk == 1;
  GemmGroup g;
  unsigned gx = 0, gy = 0;
  int z = 0;
  for (int i = 0; ok && i < n; ++i) {
    const MmegoGemmDesc& a = d[i];
    if (!(a.A && a.B && a.C && a.M > 0 && a.N > 0 && a.K > 0 && a.nbatch > 0)) return MMEGO_EBADARG;
    const long units64 = (long)(a.M / 64) * (a.N / 64) * a.nbatch;
    const long wgs64 = (long)cdiv(a.M, 64) * cdiv(a.N, 64) * a.nbatch;
    ok = a.nsplit == 1 && (a.sak == 1) == akc && (a.sbk == 1) == bkc && !(akc && bkc) && units64 < tile_min_other &&
         wgs64 <= kq_max && a.K >= 64 && a.relu != 2 && cdiv(a.N, 32) <= 65535;
    if (!ok) break;
    GemmP& p = g.p[i];
    p.A = a.A; p.B = a.B; p.bias = a.bias;
    p.sam = a.sam; p.sak = a.sak; p.sbk = a.sbk; p.sbn = a.sbn;
    p.sAb = a.sAb; p.sBb = a.sBb; p.sBiasb = a.sBiasb;
    p.M = a.M; p.N = a.N; p.K = a.K;
    p.nsplit = 1; p.kchunk = cdiv(a.K, 16) * 16;
    p.relu = a.relu; p.accumulate = a.accumulate;
    p.cmul = a.cmul; p.asum = a.asum;
    p.cb_dg[0] = p.cb_dg[1] = nullptr;
    p.C = a.C; p.scm = a.scm; p.scn = a.scn; p.sCb = a.sCb; p.sCs = 0;
    p.avec = akc && (a.sam % 4) == 0 && (a.sAb % 4) == 0 && ((uintptr_t)a.A & 15) == 0;
    p.bvec = bkc && (a.sbn % 4) == 0 && (a.sBb % 4) == 0 && ((uintptr_t)a.B & 15) == 0;
    const unsigned tx = (unsigned)cdiv(a.M, 32), ty = (unsigned)cdiv(a.N, 32);
    gx = tx > gx ? tx : gx; gy = ty > gy ? ty : gy;
    z += a.nbatch;
    g.zend[i] = z;
  }
  if (!ok || z > 65535) {
    for (int i = 0; i < n; ++i) {
      const MmegoGemmDesc& a = d[i];
      int rc = mmego_gemm(stream, a.A, a.sam, a.sak, a.B, a.sbk, a.sbn, a.C, a.scm, a.scn, a.bias, a.M, a.N, a.K, a.nbatch, a.sAb, a.sBb,
                          a.sCb, a.relu, a.accumulate, a.splitk_ws, a.nsplit, a.sBiasb, a.cmul, a.asum);
      if (rc != MMEGO_OK) return rc;
    }
    return MMEGO_OK;
  }
  g.n = n;
  for (int i = n; i < GEMM_GROUP_MAX; ++i) g.zend[i] = z;
  dim3 grid(gx, gy, (unsigned)z);
  hipStream_t st = (hipStream_t)stream;
  if (akc) hipLaunchKernelGGL((gemm32kq_group_kernel<true, false>), grid, dim3(256), 0, st, g);
  else if (bkc) hipLaunchKernelGGL((gemm32kq_group_kernel<false, true>), grid, dim3(256), 0, st, g);
  else hipLaunchKernelGGL((gemm32kq_group_kernel<false, false>), grid, dim3(256), 0, st, g);
  MMEGO_LAUNCH_CHECK();
  return MMEGO_OK;
}
