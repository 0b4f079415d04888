// One timestep of a (bi)LSTM of hidden size H (H % 32 == 0) on the fp32 matrix cores: the recurrent half of
// IMU_Net's 2 x (2-layer, H=512) BiLSTMs (reference Net/IMU_Net.py:58-62,77,82) -- 94 % of the path's FLOPs.
//   gates = xproj_t (+ b_hh) + h_{t-1} . W_hh^T ;  i,f,o = sigmoid, g = tanh ;  c = f c + i g ;  h = o tanh(c)
// Both directions run in one launch.  Two kernels, chosen by the number of batch rows:
//   lstm_step_kernel        Bn >= 128: WG = 64 rows x 32 hidden x 4 gates (v_mfma_f32_16x16x4_f32), [row][k] LDS
//                           tiles read with ds_read_b128 + a k-permutation (one 16-B read feeds 4 MFMA steps),
//                           double-buffered; the cell update is register-local.  WGs sharing a W_hh slice are
//                           placed on one XCD, so each XCD's L2 holds 1/8 of W_hh.
//   lstm_step_small_kernel  Bn < 128 (rnn_slow: 64 rows): the step is W_hh-streaming bound, so the hidden axis is
//                           cut finely (4 hidden x 4 gates per WG -> 256 WGs for H=512, every CU streams 1/256 of
//                           W_hh); gates of one hidden unit sit in 4 lanes and are exchanged with __shfl.
// `first` != 0 means h_{t-1} = 0 and c_{t-1} = 0: the product is skipped altogether.
#include <stdlib.h>

#include "common.h"

struct LstmStepP {
  const float* hprev[2]; long hps;
  const float* whh[2];
  const float* bhh[2];
  const float* xproj[2]; long xs;
  float* hout[2]; long hos;
  float* c[2];
  float* gst[2];   // optional stash for backward: post-activation gates [Bn][4H] and new cell state [Bn][H] of this step
  float* cst[2];
  int Bn, H, ndir, first;
  int packed;      // W_hh is in the chunk-packed layout (see lstm_step_ws_kernel)
};

__device__ __forceinline__ float fast_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + expf(-x)); }
__device__ __forceinline__ float fast_tanh(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + expf(2.0f * x)); }

#define KC 64     // k per staged chunk
#define SLD 68    // LDS row stride (floats) of the [row][k] staging tiles: 64 k + 4 pad (ds_read_b128 ~conflict-free)
#define XLD 132   // xproj tile: 4 gates x 32 hidden + 4 pad (rows 4 apart land 16 banks apart: conflict-free C-layout reads)
#define CLD 36    // c / h tiles: 32 hidden + 4 pad

// WG = 64 batch rows x 32 hidden x 4 gates, 4 waves (32 rows x 16 hidden x 4 gates each), one WG per CU at Bn=512,H=512.
// LDS: 2 x (64 + 128) x 68 floats of operand tiles (102 KB) + c tile (9 KB); the xproj tile and the h tile alias the
// second operand buffer (they are live only before / after the product).
__global__ __launch_bounds__(256) void lstm_step_kernel(LstmStepP p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float (*As)[64][SLD] = reinterpret_cast<float (*)[64][SLD]>(smem);                       // [2][64][SLD]
  float (*Bs)[128][SLD] = reinterpret_cast<float (*)[128][SLD]>(smem + 2 * 64 * SLD);      // [2][128][SLD]
  float (*CP)[CLD] = reinterpret_cast<float (*)[CLD]>(smem + 2 * 64 * SLD + 2 * 128 * SLD); // [64][CLD]
  float (*XP)[XLD] = reinterpret_cast<float (*)[XLD]>(&Bs[1][0][0]);                       // [64][XLD] aliases Bs[1]
  float (*HP)[CLD] = reinterpret_cast<float (*)[CLD]>(&Bs[1][0][0]);                       // [64][CLD] aliases Bs[1]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int H = p.H;
  const int nrb = (p.Bn + 63) / 64, nht = H / 32, npairs = p.ndir * nht;
  int pair, rb;
  {
    const int wg = blockIdx.x;
    if ((npairs & 7) == 0) {  // XCD-aware: blocks b and b+8 share an XCD; give each XCD whole (dir, hidden-tile) pairs
      const int xcd = wg & 7, q = wg >> 3;
      pair = xcd + 8 * (q / nrb);
      rb = q % nrb;
    } else {
      pair = wg / nrb;
      rb = wg % nrb;
    }
  }
  const int d = pair / nht, ht = pair % nht;
  const int j0 = ht * 32, r0 = rb * 64;
  const int rowbase = (wave & 1) * 32, hb = (wave >> 1) * 16;
  const int fr = lane & 15, fq = lane >> 4;
  const bool first = p.first != 0;
  const f32x4 zero4 = (f32x4){0.f, 0.f, 0.f, 0.f};

  // ---- operand tile staging (h_{t-1} rows and the W_hh slice): 16 lanes per 256-B row segment, rows lr + 16*i ----
  const int lk = (tid & 15) * 4, lr = tid >> 4;
  const int nk = H / KC;
  f32x4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3, rb4, rb5, rb6, rb7;
  const float* ap = nullptr;
  const float* wp = nullptr;
  long rs16 = 0;
  bool ok0 = false, ok1 = false, ok2 = false, ok3 = false;
#define STEP_GLOAD(k0)                                                          \
  do {                                                                          \
    ra0 = ok0 ? *reinterpret_cast<const f32x4*>(ap + (k0)) : zero4;            \
    ra1 = ok1 ? *reinterpret_cast<const f32x4*>(ap + rs16 + (k0)) : zero4;     \
    ra2 = ok2 ? *reinterpret_cast<const f32x4*>(ap + 2 * rs16 + (k0)) : zero4; \
    ra3 = ok3 ? *reinterpret_cast<const f32x4*>(ap + 3 * rs16 + (k0)) : zero4; \
    rb0 = *reinterpret_cast<const f32x4*>(wp + (k0));                          \
    rb1 = *reinterpret_cast<const f32x4*>(wp + 16 * H + (k0));                 \
    rb2 = *reinterpret_cast<const f32x4*>(wp + (long)H * H + (k0));            \
    rb3 = *reinterpret_cast<const f32x4*>(wp + (long)H * H + 16 * H + (k0));   \
    rb4 = *reinterpret_cast<const f32x4*>(wp + 2L * H * H + (k0));             \
    rb5 = *reinterpret_cast<const f32x4*>(wp + 2L * H * H + 16 * H + (k0));    \
    rb6 = *reinterpret_cast<const f32x4*>(wp + 3L * H * H + (k0));             \
    rb7 = *reinterpret_cast<const f32x4*>(wp + 3L * H * H + 16 * H + (k0));    \
  } while (0)
#define STEP_SSTORE(buf)                                                        \
  do {                                                                          \
    *reinterpret_cast<f32x4*>(&As[buf][lr][lk]) = ra0;                         \
    *reinterpret_cast<f32x4*>(&As[buf][lr + 16][lk]) = ra1;                    \
    *reinterpret_cast<f32x4*>(&As[buf][lr + 32][lk]) = ra2;                    \
    *reinterpret_cast<f32x4*>(&As[buf][lr + 48][lk]) = ra3;                    \
    *reinterpret_cast<f32x4*>(&Bs[buf][lr][lk]) = rb0;                         \
    *reinterpret_cast<f32x4*>(&Bs[buf][lr + 16][lk]) = rb1;                    \
    *reinterpret_cast<f32x4*>(&Bs[buf][lr + 32][lk]) = rb2;                    \
    *reinterpret_cast<f32x4*>(&Bs[buf][lr + 48][lk]) = rb3;                    \
    *reinterpret_cast<f32x4*>(&Bs[buf][lr + 64][lk]) = rb4;                    \
    *reinterpret_cast<f32x4*>(&Bs[buf][lr + 80][lk]) = rb5;                    \
    *reinterpret_cast<f32x4*>(&Bs[buf][lr + 96][lk]) = rb6;                    \
    *reinterpret_cast<f32x4*>(&Bs[buf][lr + 112][lk]) = rb7;                   \
  } while (0)
  MMEGO_STAMP_AT(blockIdx.x, 0, tid == 0);
  if (!first) {
    ok0 = (r0 + lr) < p.Bn; ok1 = (r0 + lr + 16) < p.Bn; ok2 = (r0 + lr + 32) < p.Bn; ok3 = (r0 + lr + 48) < p.Bn;
    ap = p.hprev[d] + (long)(r0 + lr) * p.hps + lk;
    rs16 = 16 * p.hps;
    wp = p.whh[d] + ((long)j0 + lr) * H + lk;   // Bs row = gate*32 + jl  <->  W row gate*H + j0 + jl  (jl = lr, lr+16)
    STEP_GLOAD(0);                              // issued before the xproj stream so both are in flight together
  }

  // ---- stream in this step's xproj tile (64 rows x 4 gates x 32 hidden) and c tile with full 128-B row segments ----
  {
    const int xr = tid >> 2, xq = (tid & 3) * 8;
    const bool ok = (r0 + xr) < p.Bn;
    const float* xrow = p.xproj[d] + (long)(r0 + xr) * p.xs + j0 + xq;
    f32x4 xv[8];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      xv[2 * g] = ok ? *reinterpret_cast<const f32x4*>(xrow + g * H) : zero4;
      xv[2 * g + 1] = ok ? *reinterpret_cast<const f32x4*>(xrow + g * H + 4) : zero4;
    }
    f32x4 cv0 = zero4, cv1 = zero4;
    if (ok && !first) {
      const float* crow = p.c[d] + (long)(r0 + xr) * H + j0 + xq;
      cv0 = *reinterpret_cast<const f32x4*>(crow);
      cv1 = *reinterpret_cast<const f32x4*>(crow + 4);
    }
    if (!first) STEP_SSTORE(0);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      *reinterpret_cast<f32x4*>(&XP[xr][g * 32 + xq]) = xv[2 * g];
      *reinterpret_cast<f32x4*>(&XP[xr][g * 32 + xq + 4]) = xv[2 * g + 1];
    }
    *reinterpret_cast<f32x4*>(&CP[xr][xq]) = cv0;
    *reinterpret_cast<f32x4*>(&CP[xr][xq + 4]) = cv1;
  }
  __syncthreads();

  // accumulators start from xproj + b_hh (C layout: col = lane&15 (hidden), row = (lane>>4)*4 + reg)
  f32x4 acc[2][4];
  {
    const int jj = hb + fr;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float bh = p.bhh[d] ? p.bhh[d][g * H + j0 + jj] : 0.f;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) acc[i][g][reg] = XP[rowbase + i * 16 + fq * 4 + reg][g * 32 + jj] + bh;
    }
  }
  __syncthreads();   // XP (aliasing operand buffer 1) has been consumed by every wave

  MMEGO_STAMP_AT(blockIdx.x, 1, tid == 0);
  if (!first) {
    for (int kt = 0; kt < nk; ++kt) {
      const int buf = kt & 1;
      if (kt + 1 < nk) STEP_GLOAD((kt + 1) * KC);
#pragma unroll
      for (int kb = 0; kb < KC / 16; ++kb) {
        // k-permuted operands: lane group fq supplies k = 16*kb + 4*fq + s at MFMA step s (same map for A and B)
        f32x4 a[2], b[4];
#pragma unroll
        for (int i = 0; i < 2; ++i) a[i] = *reinterpret_cast<const f32x4*>(&As[buf][rowbase + i * 16 + fr][kb * 16 + 4 * fq]);
#pragma unroll
        for (int g = 0; g < 4; ++g) b[g] = *reinterpret_cast<const f32x4*>(&Bs[buf][g * 32 + hb + fr][kb * 16 + 4 * fq]);
        // step-major order: 8 independent accumulators between two MFMAs on the same one (dependent latency 40 > issue 32)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int g = 0; g < 4; ++g) acc[i][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].x, b[g].x, acc[i][g], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int g = 0; g < 4; ++g) acc[i][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].y, b[g].y, acc[i][g], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int g = 0; g < 4; ++g) acc[i][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].z, b[g].z, acc[i][g], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int g = 0; g < 4; ++g) acc[i][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].w, b[g].w, acc[i][g], 0, 0, 0);
      }
      if (kt + 1 < nk) STEP_SSTORE(buf ^ 1);
      __syncthreads();
    }
  }
  MMEGO_STAMP_AT(blockIdx.x, 2, tid == 0);

  // fused cell update in registers, results staged in LDS for full-line stores (HP aliases operand buffer 1: idle now)
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int lrow = rowbase + i * 16 + fq * 4 + reg;
      float gi = fast_sigmoid(acc[i][0][reg]);
      float gf = fast_sigmoid(acc[i][1][reg]);
      float gg = fast_tanh(acc[i][2][reg]);
      float go = fast_sigmoid(acc[i][3][reg]);
      float cn = gf * CP[lrow][hb + fr] + gi * gg;
      CP[lrow][hb + fr] = cn;
      HP[lrow][hb + fr] = go * fast_tanh(cn);
      if (p.gst[d] && (r0 + lrow) < p.Bn) {     // training only: stash for the backward pass
        float* gs = p.gst[d] + (long)(r0 + lrow) * 4 * H + j0 + hb + fr;
        gs[0] = gi; gs[H] = gf; gs[2 * H] = gg; gs[3 * H] = go;
        p.cst[d][(long)(r0 + lrow) * H + j0 + hb + fr] = cn;
      }
    }
  }
  __syncthreads();
  MMEGO_STAMP_AT(blockIdx.x, 3, tid == 0);
  {
    const int xr = tid >> 2, xq = (tid & 3) * 8;
    if ((r0 + xr) < p.Bn) {
      float* crow = p.c[d] + (long)(r0 + xr) * H + j0 + xq;
      *reinterpret_cast<f32x4*>(crow) = *reinterpret_cast<const f32x4*>(&CP[xr][xq]);
      *reinterpret_cast<f32x4*>(crow + 4) = *reinterpret_cast<const f32x4*>(&CP[xr][xq + 4]);
      float* hrow = p.hout[d] + (long)(r0 + xr) * p.hos + j0 + xq;
      *reinterpret_cast<f32x4*>(hrow) = *reinterpret_cast<const f32x4*>(&HP[xr][xq]);
      *reinterpret_cast<f32x4*>(hrow + 4) = *reinterpret_cast<const f32x4*>(&HP[xr][xq + 4]);
    }
  }
}

#define STEP_LDS_BYTES ((2 * 64 * SLD + 2 * 128 * SLD + 64 * CLD) * sizeof(float))

// ---- warp-specialised variant: 4 compute waves (MFMA + cell update) + 4 loader waves (global -> LDS staging) -----------
// Same tile as lstm_step_kernel.  The loader waves stream the xproj / c tiles and keep two operand chunks in flight
// (registers) ahead of the LDS double buffer, so the compute waves' instruction stream is only ds_read_b128 + MFMA
// between barriers: on each SIMD one compute wave owns the matrix pipe while its loader partner issues VMEM / ds_write.
// The compute waves double-buffer their operand fragments in registers: the ds_reads of k-block kb+1 are issued before
// the 32 MFMAs of k-block kb, and the chunk barrier sits before the MFMAs of a chunk's LAST k-block (whose fragments
// are already in registers), so the first reads of the next chunk also land under MFMAs.
// (in-kernel stamps, scripts/clock_probe.hip: product loop 45.8k cycles in lstm_step_kernel, 42.2k with the wave split,
//  MFMA-only bound 32.8k)
#define WS_LDS_FLOATS (2 * 64 * SLD + 2 * 128 * SLD + 64 * CLD + 64 * XLD)
__global__ __launch_bounds__(512) void lstm_step_ws_kernel(LstmStepP p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float (*As)[64][SLD] = reinterpret_cast<float (*)[64][SLD]>(smem);
  float (*Bs)[128][SLD] = reinterpret_cast<float (*)[128][SLD]>(smem + 2 * 64 * SLD);
  float (*CP)[CLD] = reinterpret_cast<float (*)[CLD]>(smem + 2 * 64 * SLD + 2 * 128 * SLD);
  float (*XP)[XLD] = reinterpret_cast<float (*)[XLD]>(smem + 2 * 64 * SLD + 2 * 128 * SLD + 64 * CLD);
  float (*HP)[CLD] = reinterpret_cast<float (*)[CLD]>(&XP[0][0]);     // h tile reuses the xproj tile after the seed
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool loader = wave >= 4;
  const int H = p.H;
  const int nrb = (p.Bn + 63) / 64, nht = H / 32, npairs = p.ndir * nht;
  int pair, rb;
  {
    const int wg = blockIdx.x;
    if ((npairs & 7) == 0) {
      const int xcd = wg & 7, q = wg >> 3;
      pair = xcd + 8 * (q / nrb);
      rb = q % nrb;
    } else {
      pair = wg / nrb;
      rb = wg % nrb;
    }
  }
  const int d = pair / nht, ht = pair % nht;
  const int j0 = ht * 32, r0 = rb * 64;
#ifdef MMEGO_STAMP
  const bool first = (p.first & 1) != 0;      // diagnostic build only: bit 1 = no operand loads in the loop, bit 2 = no
  const int dbg = p.first >> 1;               // operand ds_writes in the loop
#else
  const bool first = p.first != 0;
  const int dbg = 0;
#endif
  const int nk = first ? 0 : H / KC;
  const f32x4 zero4 = (f32x4){0.f, 0.f, 0.f, 0.f};
  MMEGO_STAMP_AT(blockIdx.x, 0, tid == 0);

  if (loader) {
    const int lt = tid - 256;
    const int lk = (lt & 15) * 4, lr = lt >> 4;
    const int xr = lt >> 2, xq = (lt & 3) * 8;
    const bool okx = (r0 + xr) < p.Bn;
    const bool ok0 = (r0 + lr) < p.Bn, ok1 = (r0 + lr + 16) < p.Bn, ok2 = (r0 + lr + 32) < p.Bn, ok3 = (r0 + lr + 48) < p.Bn;
    const float* ap = first ? nullptr : p.hprev[d] + (long)(r0 + lr) * p.hps + lk;
    const long rs16 = 16 * p.hps;
    // W_hh addressing: row stride wrs, gate stride wgs, chunk stride factor wks (floats per k).  Natural [4H][H] layout:
    // (H, H*H, 1); chunk-packed layout [dir][hidden tile][chunk][gate*32 + j][64 k]: (64, 32*64, 128) -- every 64-k
    // chunk of a workgroup's slice is one contiguous 32 KB block.
    const bool packed = p.packed != 0;
    const long wrs = packed ? 64 : H, wgs = packed ? 32 * 64 : (long)H * H, wks = packed ? 128 : 1;
    const float* wp = packed ? p.whh[d] + (long)ht * 128 * H + lr * 64 + lk : p.whh[d] + ((long)j0 + lr) * H + lk;
    f32x4 a0, a1, a2, a3, b0, b1, b2, b3, b4, b5, b6, b7;       // chunk being written
    f32x4 c0, c1, c2, c3, e0, e1, e2, e3, e4, e5, e6, e7;       // chunk in flight
#define WS_LOAD(A0, A1, A2, A3, B0, B1, B2, B3, B4, B5, B6, B7, k0)             \
  do {                                                                          \
    A0 = ok0 ? *reinterpret_cast<const f32x4*>(ap + (k0)) : zero4;              \
    A1 = ok1 ? *reinterpret_cast<const f32x4*>(ap + rs16 + (k0)) : zero4;       \
    A2 = ok2 ? *reinterpret_cast<const f32x4*>(ap + 2 * rs16 + (k0)) : zero4;   \
    A3 = ok3 ? *reinterpret_cast<const f32x4*>(ap + 3 * rs16 + (k0)) : zero4;   \
    B0 = *reinterpret_cast<const f32x4*>(wp + (k0) * wks);                      \
    B1 = *reinterpret_cast<const f32x4*>(wp + 16 * wrs + (k0) * wks);           \
    B2 = *reinterpret_cast<const f32x4*>(wp + wgs + (k0) * wks);                \
    B3 = *reinterpret_cast<const f32x4*>(wp + wgs + 16 * wrs + (k0) * wks);     \
    B4 = *reinterpret_cast<const f32x4*>(wp + 2 * wgs + (k0) * wks);            \
    B5 = *reinterpret_cast<const f32x4*>(wp + 2 * wgs + 16 * wrs + (k0) * wks); \
    B6 = *reinterpret_cast<const f32x4*>(wp + 3 * wgs + (k0) * wks);            \
    B7 = *reinterpret_cast<const f32x4*>(wp + 3 * wgs + 16 * wrs + (k0) * wks); \
  } while (0)
#define WS_STORE(buf, A0, A1, A2, A3, B0, B1, B2, B3, B4, B5, B6, B7)           \
  do {                                                                          \
    *reinterpret_cast<f32x4*>(&As[buf][lr][lk]) = A0;                           \
    *reinterpret_cast<f32x4*>(&As[buf][lr + 16][lk]) = A1;                      \
    *reinterpret_cast<f32x4*>(&As[buf][lr + 32][lk]) = A2;                      \
    *reinterpret_cast<f32x4*>(&As[buf][lr + 48][lk]) = A3;                      \
    *reinterpret_cast<f32x4*>(&Bs[buf][lr][lk]) = B0;                           \
    *reinterpret_cast<f32x4*>(&Bs[buf][lr + 16][lk]) = B1;                      \
    *reinterpret_cast<f32x4*>(&Bs[buf][lr + 32][lk]) = B2;                      \
    *reinterpret_cast<f32x4*>(&Bs[buf][lr + 48][lk]) = B3;                      \
    *reinterpret_cast<f32x4*>(&Bs[buf][lr + 64][lk]) = B4;                      \
    *reinterpret_cast<f32x4*>(&Bs[buf][lr + 80][lk]) = B5;                      \
    *reinterpret_cast<f32x4*>(&Bs[buf][lr + 96][lk]) = B6;                      \
    *reinterpret_cast<f32x4*>(&Bs[buf][lr + 112][lk]) = B7;                     \
  } while (0)
    // every global load of the prologue is issued before the first wait: xproj tile (HBM), c tile, operand chunks 0, 1
    const float* xrow = p.xproj[d] + (long)(r0 + xr) * p.xs + j0 + xq;
    f32x4 x0, x1, x2, x3, x4, x5, x6, x7, cv0 = zero4, cv1 = zero4;
    x0 = okx ? *reinterpret_cast<const f32x4*>(xrow) : zero4;
    x1 = okx ? *reinterpret_cast<const f32x4*>(xrow + 4) : zero4;
    x2 = okx ? *reinterpret_cast<const f32x4*>(xrow + H) : zero4;
    x3 = okx ? *reinterpret_cast<const f32x4*>(xrow + H + 4) : zero4;
    x4 = okx ? *reinterpret_cast<const f32x4*>(xrow + 2 * H) : zero4;
    x5 = okx ? *reinterpret_cast<const f32x4*>(xrow + 2 * H + 4) : zero4;
    x6 = okx ? *reinterpret_cast<const f32x4*>(xrow + 3 * H) : zero4;
    x7 = okx ? *reinterpret_cast<const f32x4*>(xrow + 3 * H + 4) : zero4;
    if (nk > 0) {
      if (okx) {
        const float* crow = p.c[d] + (long)(r0 + xr) * H + j0 + xq;
        cv0 = *reinterpret_cast<const f32x4*>(crow);
        cv1 = *reinterpret_cast<const f32x4*>(crow + 4);
      }
      WS_LOAD(a0, a1, a2, a3, b0, b1, b2, b3, b4, b5, b6, b7, 0);
      if (nk > 1) WS_LOAD(c0, c1, c2, c3, e0, e1, e2, e3, e4, e5, e6, e7, KC);
    }
    *reinterpret_cast<f32x4*>(&XP[xr][xq]) = x0;
    *reinterpret_cast<f32x4*>(&XP[xr][xq + 4]) = x1;
    *reinterpret_cast<f32x4*>(&XP[xr][32 + xq]) = x2;
    *reinterpret_cast<f32x4*>(&XP[xr][32 + xq + 4]) = x3;
    *reinterpret_cast<f32x4*>(&XP[xr][64 + xq]) = x4;
    *reinterpret_cast<f32x4*>(&XP[xr][64 + xq + 4]) = x5;
    *reinterpret_cast<f32x4*>(&XP[xr][96 + xq]) = x6;
    *reinterpret_cast<f32x4*>(&XP[xr][96 + xq + 4]) = x7;
    *reinterpret_cast<f32x4*>(&CP[xr][xq]) = cv0;
    *reinterpret_cast<f32x4*>(&CP[xr][xq + 4]) = cv1;
    if (nk > 0) WS_STORE(0, a0, a1, a2, a3, b0, b1, b2, b3, b4, b5, b6, b7);
    __syncthreads();                                     // (1) chunk 0 + xproj + c tiles are in LDS
    if (nk < 2) __syncthreads();                         // (S) see the compute waves
    // barrier kt -> kt+1: the compute waves hold chunk kt's last fragments in registers (buffer kt&1 is free again) and
    // chunk kt+1 is complete in the other buffer.  Before it we write chunk kt+1 and fetch chunk kt+2.
    for (int kt = 0; kt + 1 < nk; kt += 2) {
      if (kt + 2 < nk && !(dbg & 1)) WS_LOAD(a0, a1, a2, a3, b0, b1, b2, b3, b4, b5, b6, b7, (kt + 2) * KC);
      if (!(dbg & 2)) WS_STORE(1, c0, c1, c2, c3, e0, e1, e2, e3, e4, e5, e6, e7);
      __syncthreads();
      if (kt + 2 < nk) {
        if (kt + 3 < nk && !(dbg & 1)) WS_LOAD(c0, c1, c2, c3, e0, e1, e2, e3, e4, e5, e6, e7, (kt + 3) * KC);
        if (!(dbg & 2)) WS_STORE(0, a0, a1, a2, a3, b0, b1, b2, b3, b4, b5, b6, b7);
        __syncthreads();
      }
    }
    __syncthreads();                                     // (E) cell update done: CP / HP hold the new c / h tiles
    if (okx) {
      float* crow = p.c[d] + (long)(r0 + xr) * H + j0 + xq;
      *reinterpret_cast<f32x4*>(crow) = *reinterpret_cast<const f32x4*>(&CP[xr][xq]);
      *reinterpret_cast<f32x4*>(crow + 4) = *reinterpret_cast<const f32x4*>(&CP[xr][xq + 4]);
      float* hrow = p.hout[d] + (long)(r0 + xr) * p.hos + j0 + xq;
      *reinterpret_cast<f32x4*>(hrow) = *reinterpret_cast<const f32x4*>(&HP[xr][xq]);
      *reinterpret_cast<f32x4*>(hrow + 4) = *reinterpret_cast<const f32x4*>(&HP[xr][xq + 4]);
    }
    return;
  }

  // ---------------------------------------------- compute waves ----------------------------------------------
  const int rowbase = (wave & 1) * 32, hb = (wave >> 1) * 16;
  const int fr = lane & 15, fq = lane >> 4;
  f32x4 acc00, acc01, acc02, acc03, acc10, acc11, acc12, acc13;   // acc<row half><gate>
  float bh[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) bh[g] = p.bhh[d] ? p.bhh[d][g * H + j0 + hb + fr] : 0.f;
  __syncthreads();                                       // (1)
  float cprev[2][4];
#define WS_SEED(ACC, i, g)                                                                                   \
  do {                                                                                                       \
    ACC[0] = XP[rowbase + (i) * 16 + fq * 4 + 0][(g) * 32 + hb + fr] + bh[g];                                \
    ACC[1] = XP[rowbase + (i) * 16 + fq * 4 + 1][(g) * 32 + hb + fr] + bh[g];                                \
    ACC[2] = XP[rowbase + (i) * 16 + fq * 4 + 2][(g) * 32 + hb + fr] + bh[g];                                \
    ACC[3] = XP[rowbase + (i) * 16 + fq * 4 + 3][(g) * 32 + hb + fr] + bh[g];                                \
  } while (0)
  WS_SEED(acc00, 0, 0); WS_SEED(acc01, 0, 1); WS_SEED(acc02, 0, 2); WS_SEED(acc03, 0, 3);
  WS_SEED(acc10, 1, 0); WS_SEED(acc11, 1, 1); WS_SEED(acc12, 1, 2); WS_SEED(acc13, 1, 3);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) cprev[i][reg] = CP[rowbase + i * 16 + fq * 4 + reg][hb + fr];
  // The h tile (HP) aliases the xproj tile: nobody may write h before every wave has seeded its accumulators.  With
  // nk >= 2 a chunk barrier lies in between; otherwise (first step, H = 64) an explicit one is needed.
  if (nk < 2) __syncthreads();                           // (S)
  MMEGO_STAMP_AT(blockIdx.x, 1, tid == 0);
  if (nk > 0) {
    const float* arow = &As[0][rowbase + fr][4 * fq];
    const float* brow = &Bs[0][hb + fr][4 * fq];
    f32x4 pa0, pa1, pb0, pb1, pb2, pb3;                  // fragment set P
    f32x4 qa0, qa1, qb0, qb1, qb2, qb3;                  // fragment set Q
#define WS_RD(S, buf, kb)                                                                                    \
  do {                                                                                                       \
    S##a0 = *reinterpret_cast<const f32x4*>(arow + (buf) * 64 * SLD + (kb) * 16);                            \
    S##a1 = *reinterpret_cast<const f32x4*>(arow + (buf) * 64 * SLD + 16 * SLD + (kb) * 16);                 \
    S##b0 = *reinterpret_cast<const f32x4*>(brow + (buf) * 128 * SLD + (kb) * 16);                           \
    S##b1 = *reinterpret_cast<const f32x4*>(brow + (buf) * 128 * SLD + 32 * SLD + (kb) * 16);                \
    S##b2 = *reinterpret_cast<const f32x4*>(brow + (buf) * 128 * SLD + 64 * SLD + (kb) * 16);                \
    S##b3 = *reinterpret_cast<const f32x4*>(brow + (buf) * 128 * SLD + 96 * SLD + (kb) * 16);                \
  } while (0)
#define WS_MM4(S, c)                                                                                         \
  do {                                                                                                       \
    acc00 = __builtin_amdgcn_mfma_f32_16x16x4f32(S##a0.c, S##b0.c, acc00, 0, 0, 0);                          \
    acc01 = __builtin_amdgcn_mfma_f32_16x16x4f32(S##a0.c, S##b1.c, acc01, 0, 0, 0);                          \
    acc02 = __builtin_amdgcn_mfma_f32_16x16x4f32(S##a0.c, S##b2.c, acc02, 0, 0, 0);                          \
    acc03 = __builtin_amdgcn_mfma_f32_16x16x4f32(S##a0.c, S##b3.c, acc03, 0, 0, 0);                          \
    acc10 = __builtin_amdgcn_mfma_f32_16x16x4f32(S##a1.c, S##b0.c, acc10, 0, 0, 0);                          \
    acc11 = __builtin_amdgcn_mfma_f32_16x16x4f32(S##a1.c, S##b1.c, acc11, 0, 0, 0);                          \
    acc12 = __builtin_amdgcn_mfma_f32_16x16x4f32(S##a1.c, S##b2.c, acc12, 0, 0, 0);                          \
    acc13 = __builtin_amdgcn_mfma_f32_16x16x4f32(S##a1.c, S##b3.c, acc13, 0, 0, 0);                          \
  } while (0)
#define WS_MM(S) do { WS_MM4(S, x); WS_MM4(S, y); WS_MM4(S, z); WS_MM4(S, w); } while (0)
    WS_RD(p, 0, 0);
    const int last = nk - 1;
    for (int kt = 0; kt < last; ++kt) {
      const int buf = kt & 1;
      WS_RD(q, buf, 1);
      __builtin_amdgcn_sched_barrier(0);
      WS_MM(p);
      __builtin_amdgcn_sched_barrier(0);
      WS_RD(p, buf, 2);
      __builtin_amdgcn_sched_barrier(0);
      WS_MM(q);
      __builtin_amdgcn_sched_barrier(0);
      WS_RD(q, buf, 3);
      __builtin_amdgcn_sched_barrier(0);
      WS_MM(p);
      __builtin_amdgcn_sched_barrier(0);
      __syncthreads();                                   // chunk kt+1 is in the other buffer; buffer kt&1 may be refilled
      WS_RD(p, buf ^ 1, 0);
      __builtin_amdgcn_sched_barrier(0);
      WS_MM(q);
      __builtin_amdgcn_sched_barrier(0);
    }
    {   // last chunk (peeled: its waits must not be merged with the loop's, where six younger reads are in flight)
      const int buf = last & 1;
      WS_RD(q, buf, 1);
      __builtin_amdgcn_sched_barrier(0);
      WS_MM(p);
      __builtin_amdgcn_sched_barrier(0);
      WS_RD(p, buf, 2);
      __builtin_amdgcn_sched_barrier(0);
      WS_MM(q);
      __builtin_amdgcn_sched_barrier(0);
      WS_RD(q, buf, 3);
      __builtin_amdgcn_sched_barrier(0);
      WS_MM(p);
      __builtin_amdgcn_sched_barrier(0);
      WS_MM(q);
    }
  }
  MMEGO_STAMP_AT(blockIdx.x, 2, tid == 0);
#define WS_CELL(A0, A1, A2, A3, i)                                                                           \
  _Pragma("unroll") for (int reg = 0; reg < 4; ++reg) {                                                      \
    const int lrow = rowbase + (i) * 16 + fq * 4 + reg;                                                      \
    float gi = fast_sigmoid(A0[reg]);                                                                        \
    float gf = fast_sigmoid(A1[reg]);                                                                        \
    float gg = fast_tanh(A2[reg]);                                                                           \
    float go = fast_sigmoid(A3[reg]);                                                                        \
    float cn = gf * cprev[i][reg] + gi * gg;                                                                 \
    CP[lrow][hb + fr] = cn;                                                                                  \
    HP[lrow][hb + fr] = go * fast_tanh(cn);                                                                  \
    if (p.gst[d] && (r0 + lrow) < p.Bn) {                                                                    \
      float* gs = p.gst[d] + (long)(r0 + lrow) * 4 * H + j0 + hb + fr;                                       \
      gs[0] = gi; gs[H] = gf; gs[2 * H] = gg; gs[3 * H] = go;                                                \
      p.cst[d][(long)(r0 + lrow) * H + j0 + hb + fr] = cn;                                                   \
    }                                                                                                        \
  }
  WS_CELL(acc00, acc01, acc02, acc03, 0)
  WS_CELL(acc10, acc11, acc12, acc13, 1)
  MMEGO_STAMP_AT(blockIdx.x, 3, tid == 0);
  __syncthreads();                                       // (E)
}

// ---- LDS-DMA variant: operands go global -> LDS directly (global_load_lds_dwordx4), never through VGPRs -------------
// Why: with in-kernel stamps (scripts/clock_probe.hip) the register-staged kernels' product loop takes 45k cycles
// against a 32.8k MFMA bound; removing only the loader waves' global loads gives 35.7k.  The data returning into the
// loader waves' VGPRs (and leaving again through ds_write) competes with the MFMA waves on the same SIMDs.  LDS-DMA
// takes the staging off the register file altogether.
//   * Stage image (3 stages): A = 64 rows x 64 k, W = 128 rows x 64 k, UNPADDED rows of 16 16-B slots.  A DMA
//     wave-instruction writes 1 KB lane-linearly (4 rows), so padding is impossible; bank conflicts are avoided by an
//     XOR swizzle applied on BOTH sides: slot s of row r is stored at slot s ^ (r & 15) (the loader permutes its SOURCE
//     address, the reader its LDS address; the 16 lanes of a ds_read_b128 phase hit 16 distinct slots).
//   * 4 loader waves issue 12 DMAs each per chunk and wait with counted vmcnt; barriers are raw s_barrier so that a
//     chunk stays in flight across them.  Chunk kt+3 is issued right after the barrier that retires chunk kt's stage,
//     i.e. two chunk times of latency tolerance.
//   * The xproj tile (seed of the accumulators) is DMA'd into stage 2 (free until chunk 2 is issued after barrier B0);
//     the c tile has its own 8 KB; the new c / h tiles leave through a free stage as full 128-B lines.
#define DMA_STAGE_FLOATS (192 * 64)
#define DMA_LDS_FLOATS (3 * DMA_STAGE_FLOATS + 64 * 32)
#define GLDS16(gptr, lptr)                                                                                  \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr),                   \
                                   (__attribute__((address_space(3))) void*)(lptr), 16, 0, 0)
__global__ __launch_bounds__(512) void lstm_step_dma_kernel(LstmStepP p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* const XP = smem + 2 * DMA_STAGE_FLOATS;          // [64][128] xproj tile, linear (aliases stage 2)
  float* const CPI = smem + 3 * DMA_STAGE_FLOATS;         // [64][32] c_{t-1} tile, linear
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool loader = wave >= 4;
  const int H = p.H;
  const int nrb = (p.Bn + 63) / 64, nht = H / 32, npairs = p.ndir * nht;
  int pair, rb;
  {
    const int wg = blockIdx.x;
    if ((npairs & 7) == 0) {
      const int xcd = wg & 7, q = wg >> 3;
      pair = xcd + 8 * (q / nrb);
      rb = q % nrb;
    } else {
      pair = wg / nrb;
      rb = wg % nrb;
    }
  }
  const int d = pair / nht, ht = pair % nht;
  const int j0 = ht * 32, r0 = rb * 64;
  const bool first = p.first != 0;
  const int nk = first ? 0 : H / KC;
  float* const OUT = smem + (nk % 3) * DMA_STAGE_FLOATS;  // new c tile [64][CLD], then new h tile [64][CLD]
  MMEGO_STAMP_AT(blockIdx.x, 0, tid == 0);

  if (loader) {
    const int lw = wave - 4, q4 = lane >> 4, sl = lane & 15;
    const int rmax = p.Bn - 1 - r0;                       // rows past the batch read a valid row (their results are dropped)
    // per-lane source pointers of the 4 A and 8 W DMAs of a chunk (k0 = 0), swizzled slot included
    const float* ag[4];
    const float* wg_[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int ra = 4 * (lw + 4 * i) + q4;
      ag[i] = first ? nullptr : p.hprev[d] + (long)(r0 + min(ra, rmax)) * p.hps + 4 * (sl ^ (ra & 15));
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int rw = 4 * (lw + 4 * j) + q4;
      wg_[j] = p.whh[d] + ((long)(rw >> 5) * H + j0 + (rw & 31)) * H + 4 * (sl ^ (rw & 15));
    }
#define DMA_CHUNK(kt)                                                                                       \
  do {                                                                                                      \
    float* st_ = smem + ((kt) % 3) * DMA_STAGE_FLOATS;                                                      \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) GLDS16(ag[i] + (kt) * KC, st_ + 4 * (lw + 4 * i) * 64);   \
    _Pragma("unroll") for (int j = 0; j < 8; ++j) GLDS16(wg_[j] + (kt) * KC, st_ + 4096 + 4 * (lw + 4 * j) * 64); \
  } while (0)
    // prologue: xproj tile, c tile, chunks 0 and 1 -- all in flight before the first wait
    {
      const int piece = lane & 31;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int ii = lw + 4 * j, xr = 2 * ii + (lane >> 5);
        GLDS16(p.xproj[d] + (long)(r0 + min(xr, rmax)) * p.xs + (long)(piece >> 3) * H + j0 + (piece & 7) * 4, XP + 2 * ii * 128);
      }
    }
    if (nk > 0) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int ii = lw + 4 * j, cr = 8 * ii + (lane >> 3);
        GLDS16(p.c[d] + (long)(r0 + min(cr, rmax)) * H + j0 + (lane & 7) * 4, CPI + 8 * ii * 32);
      }
      DMA_CHUNK(0);
      if (nk > 1) {
        DMA_CHUNK(1);
        asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();                         // (1) xproj, c, chunk 0 are in LDS
    // barrier B_kt (kt = 0 .. nk-2): chunk kt+1 has landed; the compute waves hold chunk kt's last fragments in registers
    for (int kt = 0; kt + 1 < nk; ++kt) {
      if (kt >= 1 && kt + 2 < nk) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (kt == 0 && 2 < nk) DMA_CHUNK(2);               // stage 2 held the xproj tile until every wave had seeded
      if (kt + 3 < nk) DMA_CHUNK(kt + 3);
    }
    __builtin_amdgcn_s_barrier();                         // (E) new c / h tiles are in OUT
    {
      const int lt = tid - 256, xr = lt >> 2, xq = (lt & 3) * 8;
      if ((r0 + xr) < p.Bn) {
        float* crow = p.c[d] + (long)(r0 + xr) * H + j0 + xq;
        *reinterpret_cast<f32x4*>(crow) = *reinterpret_cast<const f32x4*>(OUT + xr * CLD + xq);
        *reinterpret_cast<f32x4*>(crow + 4) = *reinterpret_cast<const f32x4*>(OUT + xr * CLD + xq + 4);
        float* hrow = p.hout[d] + (long)(r0 + xr) * p.hos + j0 + xq;
        *reinterpret_cast<f32x4*>(hrow) = *reinterpret_cast<const f32x4*>(OUT + 64 * CLD + xr * CLD + xq);
        *reinterpret_cast<f32x4*>(hrow + 4) = *reinterpret_cast<const f32x4*>(OUT + 64 * CLD + xr * CLD + xq + 4);
      }
    }
    return;
  }

  // ---------------------------------------------- compute waves ----------------------------------------------
  const int rowbase = (wave & 1) * 32, hb = (wave >> 1) * 16;
  const int fr = lane & 15, fq = lane >> 4;
  f32x4 acc00, acc01, acc02, acc03, acc10, acc11, acc12, acc13;   // acc<row half><gate>
  float bh[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) bh[g] = p.bhh[d] ? p.bhh[d][g * H + j0 + hb + fr] : 0.f;
  __syncthreads();                                       // (1)
  float cprev[2][4];
#define DMA_SEED(ACC, i, g)                                                                                 \
  do {                                                                                                      \
    ACC[0] = XP[(rowbase + (i) * 16 + fq * 4 + 0) * 128 + (g) * 32 + hb + fr] + bh[g];                      \
    ACC[1] = XP[(rowbase + (i) * 16 + fq * 4 + 1) * 128 + (g) * 32 + hb + fr] + bh[g];                      \
    ACC[2] = XP[(rowbase + (i) * 16 + fq * 4 + 2) * 128 + (g) * 32 + hb + fr] + bh[g];                      \
    ACC[3] = XP[(rowbase + (i) * 16 + fq * 4 + 3) * 128 + (g) * 32 + hb + fr] + bh[g];                      \
  } while (0)
  DMA_SEED(acc00, 0, 0); DMA_SEED(acc01, 0, 1); DMA_SEED(acc02, 0, 2); DMA_SEED(acc03, 0, 3);
  DMA_SEED(acc10, 1, 0); DMA_SEED(acc11, 1, 1); DMA_SEED(acc12, 1, 2); DMA_SEED(acc13, 1, 3);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) cprev[i][reg] = nk > 0 ? CPI[(rowbase + i * 16 + fq * 4 + reg) * 32 + hb + fr] : 0.f;
  MMEGO_STAMP_AT(blockIdx.x, 1, tid == 0);
  if (nk > 0) {
    // fragment addresses: row * 64 floats + swizzled slot; sw<kb> = ((4 kb + fq) ^ fr) * 4 floats
    const int sw0 = ((0 | fq) ^ fr) << 2, sw1 = ((4 | fq) ^ fr) << 2, sw2 = ((8 | fq) ^ fr) << 2, sw3 = ((12 | fq) ^ fr) << 2;
    const float* arow = smem + (rowbase + fr) * 64;
    const float* brow = smem + 4096 + (hb + fr) * 64;
    f32x4 pa0, pa1, pb0, pb1, pb2, pb3;                  // fragment set P
    f32x4 qa0, qa1, qb0, qb1, qb2, qb3;                  // fragment set Q
#define DMA_RD(S, so, sw)                                                                                   \
  do {                                                                                                      \
    S##a0 = *reinterpret_cast<const f32x4*>(arow + (so) + (sw));                                            \
    S##a1 = *reinterpret_cast<const f32x4*>(arow + (so) + 16 * 64 + (sw));                                  \
    S##b0 = *reinterpret_cast<const f32x4*>(brow + (so) + (sw));                                            \
    S##b1 = *reinterpret_cast<const f32x4*>(brow + (so) + 32 * 64 + (sw));                                  \
    S##b2 = *reinterpret_cast<const f32x4*>(brow + (so) + 64 * 64 + (sw));                                  \
    S##b3 = *reinterpret_cast<const f32x4*>(brow + (so) + 96 * 64 + (sw));                                  \
  } while (0)
    int so = 0;                                          // float offset of the stage being read
    DMA_RD(p, so, sw0);
    const int last = nk - 1;
    for (int kt = 0; kt < last; ++kt) {
      DMA_RD(q, so, sw1);
      __builtin_amdgcn_sched_barrier(0);
      WS_MM(p);
      __builtin_amdgcn_sched_barrier(0);
      DMA_RD(p, so, sw2);
      __builtin_amdgcn_sched_barrier(0);
      WS_MM(q);
      __builtin_amdgcn_sched_barrier(0);
      DMA_RD(q, so, sw3);
      __builtin_amdgcn_sched_barrier(0);
      WS_MM(p);
      __builtin_amdgcn_sched_barrier(0);
      __syncthreads();                                   // B_kt: chunk kt+1 has landed; this stage may be refilled
      so = (so == 2 * DMA_STAGE_FLOATS) ? 0 : so + DMA_STAGE_FLOATS;
      DMA_RD(p, so, sw0);
      __builtin_amdgcn_sched_barrier(0);
      WS_MM(q);
      __builtin_amdgcn_sched_barrier(0);
    }
    {   // last chunk (peeled: its waits must not be merged with the loop's, where six younger reads are in flight)
      DMA_RD(q, so, sw1);
      __builtin_amdgcn_sched_barrier(0);
      WS_MM(p);
      __builtin_amdgcn_sched_barrier(0);
      DMA_RD(p, so, sw2);
      __builtin_amdgcn_sched_barrier(0);
      WS_MM(q);
      __builtin_amdgcn_sched_barrier(0);
      DMA_RD(q, so, sw3);
      __builtin_amdgcn_sched_barrier(0);
      WS_MM(p);
      __builtin_amdgcn_sched_barrier(0);
      WS_MM(q);
    }
  }
  MMEGO_STAMP_AT(blockIdx.x, 2, tid == 0);
#define DMA_CELL(A0, A1, A2, A3, i)                                                                         \
  _Pragma("unroll") for (int reg = 0; reg < 4; ++reg) {                                                     \
    const int lrow = rowbase + (i) * 16 + fq * 4 + reg;                                                     \
    float gi = fast_sigmoid(A0[reg]);                                                                       \
    float gf = fast_sigmoid(A1[reg]);                                                                       \
    float gg = fast_tanh(A2[reg]);                                                                          \
    float go = fast_sigmoid(A3[reg]);                                                                       \
    float cn = gf * cprev[i][reg] + gi * gg;                                                                \
    OUT[lrow * CLD + hb + fr] = cn;                                                                         \
    OUT[64 * CLD + lrow * CLD + hb + fr] = go * fast_tanh(cn);                                              \
    if (p.gst[d] && (r0 + lrow) < p.Bn) {                                                                   \
      float* gs = p.gst[d] + (long)(r0 + lrow) * 4 * H + j0 + hb + fr;                                      \
      gs[0] = gi; gs[H] = gf; gs[2 * H] = gg; gs[3 * H] = go;                                               \
      p.cst[d][(long)(r0 + lrow) * H + j0 + hb + fr] = cn;                                                  \
    }                                                                                                       \
  }
  DMA_CELL(acc00, acc01, acc02, acc03, 0)
  DMA_CELL(acc10, acc11, acc12, acc13, 1)
  MMEGO_STAMP_AT(blockIdx.x, 3, tid == 0);
  __syncthreads();                                       // (E)
}

// ---- LDS-DMA variant with a 72 KB footprint: TWO workgroups per CU -----------------------------------------------------
// Same tile and DMA scheme as lstm_step_dma_kernel, but 32-k chunks (three 24-KB stages) and no xproj / c tiles in LDS:
// the compute waves fetch their own xproj / c_{t-1} elements straight into registers at kernel start and consume them
// AFTER the product (gates = h.W^T + xproj + b).  With 72 KB and <= 128 VGPRs two workgroups share a CU.  That matters
// when independent step kernels exist (the two concurrent stage programs each run an IMU_Net forward): while one waits for
// its first operands, its barriers or its stores, the other owns the matrix pipe -- and kernels of other branches with
// modest LDS needs can co-reside as well.
//   stage image: A 64 rows x 32 k, W 128 rows x 32 k, unpadded 128-B rows of 8 16-B pieces; piece p of row r is stored at
//   piece p ^ ((r >> 1) & 7): the 16 lanes of a ds_read_b128 phase (rows fr = 0..15, same piece) hit 16 distinct 4-bank groups.
#define D2_KC 32
#define D2_STAGE_FLOATS (192 * D2_KC)
#define D2_LDS_FLOATS (3 * D2_STAGE_FLOATS)
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void lstm_step_dma2_kernel(LstmStepP p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool loader = wave >= 4;
  const int H = p.H;
  const int nrb = (p.Bn + 63) / 64, nht = H / 32, npairs = p.ndir * nht;
  int pair, rb;
  {
    const int wg = blockIdx.x;
    if ((npairs & 7) == 0) {
      const int xcd = wg & 7, q = wg >> 3;
      pair = xcd + 8 * (q / nrb);
      rb = q % nrb;
    } else {
      pair = wg / nrb;
      rb = wg % nrb;
    }
  }
  const int d = pair / nht, ht = pair % nht;
  const int j0 = ht * 32, r0 = rb * 64;
  const bool first = p.first != 0;
  const int nk = first ? 0 : H / D2_KC;
  float* const OUT = smem + (nk % 3) * D2_STAGE_FLOATS;   // new c tile [64][CLD], then new h tile [64][CLD] (4608 floats)
  MMEGO_STAMP_AT(blockIdx.x, 0, tid == 0);

  if (loader) {
    const int lw = wave - 4, q8 = lane >> 3, sl = lane & 7;
    const int rmax = p.Bn - 1 - r0;                       // rows past the batch read a valid row (their results are dropped)
    const float* ag[2];
    const float* wg_[4];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int ra = 8 * (lw + 4 * i) + q8;
      ag[i] = first ? nullptr : p.hprev[d] + (long)(r0 + min(ra, rmax)) * p.hps + 4 * (sl ^ ((ra >> 1) & 7));
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int rw = 8 * (lw + 4 * j) + q8;
      wg_[j] = p.whh[d] + ((long)(rw >> 5) * H + j0 + (rw & 31)) * H + 4 * (sl ^ ((rw >> 1) & 7));
    }
#define D2_CHUNK(kt)                                                                                        \
  do {                                                                                                      \
    float* st_ = smem + ((kt) % 3) * D2_STAGE_FLOATS;                                                       \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) GLDS16(ag[i] + (kt) * D2_KC, st_ + 8 * (lw + 4 * i) * 32); \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) GLDS16(wg_[j] + (kt) * D2_KC, st_ + 2048 + 8 * (lw + 4 * j) * 32); \
  } while (0)
    if (nk > 0) {
      D2_CHUNK(0);
      if (nk > 1) D2_CHUNK(1);
      if (nk > 2) D2_CHUNK(2);
      if (nk > 2) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
      else if (nk > 1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();                         // (1) chunk 0 is in LDS
    for (int kt = 0; kt + 1 < nk; ++kt) {                 // B_kt: chunk kt+1 has landed; stage kt % 3 may be refilled
      if (kt + 2 < nk) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (kt + 3 < nk) D2_CHUNK(kt + 3);
    }
    __builtin_amdgcn_s_barrier();                         // (E) new c / h tiles are in OUT
    {
      const int lt = tid - 256, xr = lt >> 2, xq = (lt & 3) * 8;
      if ((r0 + xr) < p.Bn) {
        float* crow = p.c[d] + (long)(r0 + xr) * H + j0 + xq;
        *reinterpret_cast<f32x4*>(crow) = *reinterpret_cast<const f32x4*>(OUT + xr * CLD + xq);
        *reinterpret_cast<f32x4*>(crow + 4) = *reinterpret_cast<const f32x4*>(OUT + xr * CLD + xq + 4);
        float* hrow = p.hout[d] + (long)(r0 + xr) * p.hos + j0 + xq;
        *reinterpret_cast<f32x4*>(hrow) = *reinterpret_cast<const f32x4*>(OUT + 64 * CLD + xr * CLD + xq);
        *reinterpret_cast<f32x4*>(hrow + 4) = *reinterpret_cast<const f32x4*>(OUT + 64 * CLD + xr * CLD + xq + 4);
      }
    }
    return;
  }

  // ---------------------------------------------- compute waves ----------------------------------------------
  const int rowbase = (wave & 1) * 32, hb = (wave >> 1) * 16;
  const int fr = lane & 15, fq = lane >> 4;
  const int rmax = p.Bn - 1 - r0;
  // xproj (+ b_hh) and c_{t-1} of this lane's 8 rows x 4 gates, fetched now, consumed after the product
  float xg[2][4][4], cprev[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int lrow = min(rowbase + i * 16 + fq * 4 + reg, rmax);
      const float* xrow = p.xproj[d] + (long)(r0 + lrow) * p.xs + j0 + hb + fr;
#pragma unroll
      for (int g = 0; g < 4; ++g) xg[i][g][reg] = xrow[(long)g * H];
      cprev[i][reg] = nk > 0 ? p.c[d][(long)(r0 + lrow) * H + j0 + hb + fr] : 0.f;
    }
  float bh[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) bh[g] = p.bhh[d] ? p.bhh[d][g * H + j0 + hb + fr] : 0.f;
  f32x4 acc00 = {0.f, 0.f, 0.f, 0.f}, acc01 = acc00, acc02 = acc00, acc03 = acc00, acc10 = acc00, acc11 = acc00, acc12 = acc00,
        acc13 = acc00;
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                          // (1)  (raw barrier: the xproj loads stay in flight)
  MMEGO_STAMP_AT(blockIdx.x, 1, tid == 0);
  if (nk > 0) {
    const int key = (fr >> 1) & 7;
    const int sw0 = ((0 | fq) ^ key) << 2, sw1 = ((4 | fq) ^ key) << 2;       // k-block 0 / 1 of a 32-k chunk
    const float* arow = smem + (rowbase + fr) * 32;
    const float* brow = smem + 2048 + (hb + fr) * 32;
    f32x4 pa0, pa1, pb0, pb1, pb2, pb3;
#define D2_RD(S, so, sw)                                                                                    \
  do {                                                                                                      \
    S##a0 = *reinterpret_cast<const f32x4*>(arow + (so) + (sw));                                            \
    S##a1 = *reinterpret_cast<const f32x4*>(arow + (so) + 16 * 32 + (sw));                                  \
    S##b0 = *reinterpret_cast<const f32x4*>(brow + (so) + (sw));                                            \
    S##b1 = *reinterpret_cast<const f32x4*>(brow + (so) + 32 * 32 + (sw));                                  \
    S##b2 = *reinterpret_cast<const f32x4*>(brow + (so) + 64 * 32 + (sw));                                  \
    S##b3 = *reinterpret_cast<const f32x4*>(brow + (so) + 96 * 32 + (sw));                                  \
  } while (0)
    // One fragment set only (<= 128 VGPRs so that two workgroups fit a CU): the other workgroup on the CU covers this
    // wave's ds_read latency and barrier waits.
    int so = 0;
    for (int kt = 0; kt < nk; ++kt) {
      D2_RD(p, so, sw0);
      WS_MM(p);
      D2_RD(p, so, sw1);
      if (kt + 1 < nk) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave's reads of the stage are done (fragments in registers)
        __builtin_amdgcn_s_barrier();                       // B_kt
        so = (so == 2 * D2_STAGE_FLOATS) ? 0 : so + D2_STAGE_FLOATS;
      }
      WS_MM(p);
    }
  }
  MMEGO_STAMP_AT(blockIdx.x, 2, tid == 0);
#define D2_CELL(A0, A1, A2, A3, i)                                                                          \
  _Pragma("unroll") for (int reg = 0; reg < 4; ++reg) {                                                     \
    const int lrow = rowbase + (i) * 16 + fq * 4 + reg;                                                     \
    float gi = fast_sigmoid(A0[reg] + (xg[i][0][reg] + bh[0]));                                             \
    float gf = fast_sigmoid(A1[reg] + (xg[i][1][reg] + bh[1]));                                             \
    float gg = fast_tanh(A2[reg] + (xg[i][2][reg] + bh[2]));                                                \
    float go = fast_sigmoid(A3[reg] + (xg[i][3][reg] + bh[3]));                                             \
    float cn = gf * cprev[i][reg] + gi * gg;                                                                \
    OUT[lrow * CLD + hb + fr] = cn;                                                                         \
    OUT[64 * CLD + lrow * CLD + hb + fr] = go * fast_tanh(cn);                                              \
    if (p.gst[d] && (r0 + lrow) < p.Bn) {                                                                   \
      float* gs = p.gst[d] + (long)(r0 + lrow) * 4 * H + j0 + hb + fr;                                      \
      gs[0] = gi; gs[H] = gf; gs[2 * H] = gg; gs[3 * H] = go;                                               \
      p.cst[d][(long)(r0 + lrow) * H + j0 + hb + fr] = cn;                                                  \
    }                                                                                                       \
  }
  D2_CELL(acc00, acc01, acc02, acc03, 0)
  D2_CELL(acc10, acc11, acc12, acc13, 1)
  MMEGO_STAMP_AT(blockIdx.x, 3, tid == 0);
  __syncthreads();                                       // (E)
}

// ---- small-batch variant: WG = 64 rows x (4 hidden x 4 gates), K split over nothing, 4 waves = 4 row tiles ----------
template <int SK>  // k per staged chunk (64, or 32 when H is not a multiple of 64)
__global__ __launch_bounds__(256) void lstm_step_small_kernel(LstmStepP p) {
  __shared__ __attribute__((aligned(16))) float As[2][64][SK + 4];
  __shared__ __attribute__((aligned(16))) float Bs[2][16][SK + 4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int H = p.H;
  const int nrb = (p.Bn + 63) / 64, nht = H / 4;
  const int wg = blockIdx.x;
  const int rb = wg % nrb, pair = wg / nrb;
  const int d = pair / nht, ht = pair % nht;
  const int j0 = ht * 4, r0 = rb * 64;
  const int fr = lane & 15, fq = lane >> 4;
  const int gate = fr >> 2, jl = fr & 3;       // tile column fr = gate*4 + jl  <->  W row gate*H + j0 + jl
  const int j = j0 + jl;
  const int rowt = r0 + wave * 16;              // this wave's 16 rows

  float xpv[4], cprev[4];
  const float bh = p.bhh[d] ? p.bhh[d][gate * H + j] : 0.f;
#pragma unroll
  for (int reg = 0; reg < 4; ++reg) {
    const int row = rowt + fq * 4 + reg;
    const bool ok = row < p.Bn;
    xpv[reg] = ok ? p.xproj[d][(long)row * p.xs + gate * H + j] + bh : 0.f;
    cprev[reg] = (ok && !p.first && gate == 0) ? p.c[d][(long)row * H + j] : 0.f;
  }
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  if (!p.first) {
    const float* hp = p.hprev[d];
    const float* W = p.whh[d];
    // staging per chunk of 64 k: A 64 rows x 16 f32x4 = 1024 f32x4 (4 per thread), B 16 rows x 16 f32x4 = 256 (1 per thread)
    constexpr int LPR = SK / 4;                     // lanes per row segment
    constexpr int RPP = 256 / LPR;                  // rows staged per pass (16 or 32)
    const int lk = (tid % LPR) * 4, lr = tid / LPR;
    const float* wp = W + ((long)((lr & 15) >> 2) * H + j0 + (lr & 3)) * H + lk;   // (lr & 15) = gate*4 + jl
    const f32x4 zero4 = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int nk = H / SK;
    const bool ok0 = (r0 + lr) < p.Bn, ok1 = (r0 + lr + RPP) < p.Bn, ok2 = (r0 + lr + 2 * RPP) < p.Bn, ok3 = (r0 + lr + 3 * RPP) < p.Bn;
    const float* ap = hp + (long)(r0 + lr) * p.hps + lk;
    const long rs16 = RPP * p.hps;
    f32x4 ra0, ra1, ra2 = zero4, ra3 = zero4, rbv = zero4;
#define SM_GLOAD(k0)                                                          \
  do {                                                                        \
    ra0 = ok0 ? *reinterpret_cast<const f32x4*>(ap + (k0)) : zero4;            \
    ra1 = ok1 ? *reinterpret_cast<const f32x4*>(ap + rs16 + (k0)) : zero4;     \
    if (RPP == 16) {                                                          \
      ra2 = ok2 ? *reinterpret_cast<const f32x4*>(ap + 2 * rs16 + (k0)) : zero4; \
      ra3 = ok3 ? *reinterpret_cast<const f32x4*>(ap + 3 * rs16 + (k0)) : zero4; \
    }                                                                         \
    if (lr < 16) rbv = *reinterpret_cast<const f32x4*>(wp + (k0));           \
  } while (0)
#define SM_SSTORE(buf)                                                        \
  do {                                                                        \
    *reinterpret_cast<f32x4*>(&As[buf][lr][lk]) = ra0;                       \
    *reinterpret_cast<f32x4*>(&As[buf][lr + RPP][lk]) = ra1;                 \
    if (RPP == 16) {                                                          \
      *reinterpret_cast<f32x4*>(&As[buf][lr + 32][lk]) = ra2;                \
      *reinterpret_cast<f32x4*>(&As[buf][lr + 48][lk]) = ra3;                \
    }                                                                         \
    if (lr < 16) *reinterpret_cast<f32x4*>(&Bs[buf][lr][lk]) = rbv;          \
  } while (0)
    SM_GLOAD(0);
    SM_SSTORE(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
      const int buf = kt & 1;
      if (kt + 1 < nk) SM_GLOAD((kt + 1) * SK);
#pragma unroll
      for (int kb = 0; kb < SK / 16; ++kb) {
        f32x4 a = *reinterpret_cast<const f32x4*>(&As[buf][wave * 16 + fr][kb * 16 + 4 * fq]);
        f32x4 b = *reinterpret_cast<const f32x4*>(&Bs[buf][fr][kb * 16 + 4 * fq]);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, acc1, 0, 0, 0);
      }
      if (kt + 1 < nk) SM_SSTORE(buf ^ 1);
      __syncthreads();
    }
  }
  // lane holds gate `gate` of hidden j for rows fq*4+reg; fetch f,g,o from lanes fr+4, fr+8, fr+12 of the same group
#pragma unroll
  for (int reg = 0; reg < 4; ++reg) {
    float pre = acc0[reg] + acc1[reg] + xpv[reg];
    float act = (gate == 2) ? fast_tanh(pre) : fast_sigmoid(pre);
    const int base = (lane & 48) + jl;          // lane of gate 0 for this (group, jl)
    float gi = __shfl(act, base, 64), gf = __shfl(act, base + 4, 64), gg = __shfl(act, base + 8, 64),
          go = __shfl(act, base + 12, 64);
    const int row = rowt + fq * 4 + reg;
    if (gate == 0 && row < p.Bn) {
      float cn = gf * cprev[reg] + gi * gg;
      p.c[d][(long)row * H + j] = cn;
      p.hout[d][(long)row * p.hos + j] = go * fast_tanh(cn);
      if (p.gst[d]) {
        float* gs = p.gst[d] + (long)row * 4 * H + j;
        gs[0] = gi; gs[H] = gf; gs[2 * H] = gg; gs[3 * H] = go;
        p.cst[d][(long)row * H + j] = cn;
      }
    }
  }
}

extern "C" int mmego_lstm_step(void* stream, int ndir, int Bn, int H, int first, const float* hprev0, const float* hprev1,
                               long hps, const float* whh0, const float* whh1, const float* bhh0, const float* bhh1,
                               const float* xproj0, const float* xproj1, long xs, float* hout0, float* hout1, long hos,
                               float* c0, float* c1, float* gst0, float* gst1, float* cst0, float* cst1) {
  MMEGO_REQUIRE((ndir == 1 || ndir == 2) && Bn > 0 && H > 0 && (H % 32) == 0);
  MMEGO_REQUIRE(whh0 && xproj0 && hout0 && c0 && (((uintptr_t)whh0) & 15) == 0);
  MMEGO_REQUIRE((xs % 4) == 0 && (hos % 4) == 0 && ((((uintptr_t)xproj0) | ((uintptr_t)hout0) | ((uintptr_t)c0)) & 15) == 0);
  if (!first) MMEGO_REQUIRE(hprev0 && (hps % 4) == 0 && (((uintptr_t)hprev0) & 15) == 0);
  if (ndir == 2) {
    MMEGO_REQUIRE(whh1 && xproj1 && hout1 && c1 && (((uintptr_t)whh1) & 15) == 0);
    MMEGO_REQUIRE(((((uintptr_t)xproj1) | ((uintptr_t)hout1) | ((uintptr_t)c1)) & 15) == 0);
    if (!first) MMEGO_REQUIRE(hprev1 && (((uintptr_t)hprev1) & 15) == 0);
  }
  LstmStepP p;
  p.hprev[0] = hprev0; p.hprev[1] = hprev1; p.hps = hps;
  p.whh[0] = whh0; p.whh[1] = whh1;
  p.bhh[0] = bhh0; p.bhh[1] = bhh1;
  p.xproj[0] = xproj0; p.xproj[1] = xproj1; p.xs = xs;
  p.hout[0] = hout0; p.hout[1] = hout1; p.hos = hos;
  p.c[0] = c0; p.c[1] = c1;
  p.gst[0] = gst0; p.gst[1] = gst1; p.cst[0] = cst0; p.cst[1] = cst1;
  MMEGO_REQUIRE((gst0 == nullptr) == (cst0 == nullptr) && (gst1 == nullptr) == (cst1 == nullptr));
  p.Bn = Bn; p.H = H; p.ndir = ndir; p.first = first;
  p.packed = 0;
#ifdef MMEGO_STAMP
  p.packed = getenv("PROBE_PACKED") != nullptr;   // diagnostic build: address W_hh as if chunk-packed (traffic pattern only)
#endif
  static const int ws_mode = getenv("MMEGO_STEP_WS") ? atoi(getenv("MMEGO_STEP_WS")) : 3;   // 3: 72-KB DMA, 2: 156-KB DMA, 1: wave-split, 0: plain
  if (Bn >= 128 && (H % 32) == 0 && ws_mode == 3) {
    static bool d2_attr = false;
    const size_t lds = (size_t)D2_LDS_FLOATS * sizeof(float);
    if (!d2_attr) {
      hipError_t e = hipFuncSetAttribute((const void*)lstm_step_dma2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return (int)e;
      d2_attr = true;
    }
    int grid = ndir * (H / 32) * cdiv(Bn, 64);
    hipLaunchKernelGGL(lstm_step_dma2_kernel, dim3(grid), dim3(512), lds, (hipStream_t)stream, p);
  } else if (Bn >= 128 && (H % KC) == 0 && ws_mode == 2) {
    static bool dma_attr = false;
    const size_t lds = (size_t)DMA_LDS_FLOATS * sizeof(float);
    if (!dma_attr) {
      hipError_t e = hipFuncSetAttribute((const void*)lstm_step_dma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return (int)e;
      dma_attr = true;
    }
    int grid = ndir * (H / 32) * cdiv(Bn, 64);
    hipLaunchKernelGGL(lstm_step_dma_kernel, dim3(grid), dim3(512), lds, (hipStream_t)stream, p);
  } else if (Bn >= 128 && (H % KC) == 0 && ws_mode) {
    static bool ws_attr = false;
    const size_t lds = (size_t)WS_LDS_FLOATS * sizeof(float);
    if (!ws_attr) {
      hipError_t e = hipFuncSetAttribute((const void*)lstm_step_ws_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return (int)e;
      ws_attr = true;
    }
    int grid = ndir * (H / 32) * cdiv(Bn, 64);
    hipLaunchKernelGGL(lstm_step_ws_kernel, dim3(grid), dim3(512), lds, (hipStream_t)stream, p);
  } else if (Bn >= 128 && (H % KC) == 0) {
    static bool attr_set = false;
    if (!attr_set) {
      hipError_t e = hipFuncSetAttribute((const void*)lstm_step_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)STEP_LDS_BYTES);
      if (e != hipSuccess) return (int)e;
      attr_set = true;
    }
    int grid = ndir * (H / 32) * cdiv(Bn, 64);
    hipLaunchKernelGGL(lstm_step_kernel, dim3(grid), dim3(256), STEP_LDS_BYTES, (hipStream_t)stream, p);
  } else {
    int grid = ndir * (H / 4) * cdiv(Bn, 64);
    if ((H % 64) == 0) hipLaunchKernelGGL(lstm_step_small_kernel<64>, dim3(grid), dim3(256), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(lstm_step_small_kernel<32>, dim3(grid), dim3(256), 0, (hipStream_t)stream, p);
  }
  MMEGO_LAUNCH_CHECK();
  return MMEGO_OK;
}
