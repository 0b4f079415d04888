// One timestep of a (bi)LSTM of hidden size H (H % 32 == 0) on the fp32 matrix cores: the recurrent half of
// IMU_Net's 2 x (2-layer, H=512) BiLSTMs (reference Net/IMU_Net.py:58-62,77,82) -- 94 % of the path's FLOPs.
//   gates = xproj_t (+ b_hh) + h_{t-1} . W_hh^T ;  i,f,o = sigmoid, g = tanh ;  c = f c + i g ;  h = o tanh(c)
// Both directions run in one launch.  Two kernels, chosen by the number of batch rows:
//   lstm_step_dma2_kernel   Bn >= 128: WG = 64 rows x 32 hidden x 4 gates (v_mfma_f32_16x16x4_f32); operands go
//                           global -> LDS by LDS-DMA into a 3-stage ring of 32-k chunks (72 KB: two WGs per CU), four
//                           loader waves + four MFMA waves; the cell update is register-local.  WGs sharing a W_hh
//                           slice are placed on one XCD, so each XCD's L2 holds 1/8 of W_hh.
//                           (Earlier variants -- plain double-buffered, register-staged loader waves, a 156-KB DMA
//                           ring -- all measured slower and were removed)
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
};

// rcp / v_exp_f32 based activations (as in lstm.hip): a few ulp from the libm forms at a fraction of their instruction count
// (libm expf is ~15 VALU instructions; the cell update evaluates 40 of them per lane)
__device__ __forceinline__ float fast_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float fast_tanh(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * x)); }

#define CLD 36    // LDS row stride of the c / h output tiles: 32 hidden + 4 pad

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
#define GLDS16(gptr, lptr)                                                                                  \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr),                   \
                                   (__attribute__((address_space(3))) void*)(lptr), 16, 0, 0)

// ---- LDS-DMA step kernel, 72 KB footprint: TWO workgroups per CU -------------------------------------------------------
// Operands go global -> LDS by LDS-DMA (4 loader waves, counted vmcnt, raw s_barriers), 32-k chunks in a ring of three 24-KB
// stages; no xproj / c tiles in LDS: the compute waves fetch their own xproj elements straight into the accumulators
// (gates = xproj + b + h.W^T) and c_{t-1} into registers at kernel start.  With 72 KB and <= 128 VGPRs two workgroups share a CU.  That matters
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
  // The accumulators are SEEDED with xproj (+ b_hh) of this lane's 8 rows x 4 gates (gates = xproj + b + h.W^T), so the
  // projection needs no registers of its own during the product -- they pay for a second fragment set (below).  The loads
  // land while the loader waves' first operand chunk is in flight.  c_{t-1} is fetched now and consumed after the product.
  float bh[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) bh[g] = p.bhh[d] ? p.bhh[d][g * H + j0 + hb + fr] : 0.f;
  f32x4 acc00, acc01, acc02, acc03, acc10, acc11, acc12, acc13;
  float cprev[2][4];
#define D2_SEED(A0, A1, A2, A3, i)                                                                          \
  _Pragma("unroll") for (int reg = 0; reg < 4; ++reg) {                                                     \
    const int lrow = min(rowbase + (i) * 16 + fq * 4 + reg, rmax);                                          \
    const float* xrow = p.xproj[d] + (long)(r0 + lrow) * p.xs + j0 + hb + fr;                               \
    A0[reg] = xrow[0] + bh[0];                                                                              \
    A1[reg] = xrow[(long)H] + bh[1];                                                                        \
    A2[reg] = xrow[2 * (long)H] + bh[2];                                                                    \
    A3[reg] = xrow[3 * (long)H] + bh[3];                                                                    \
    cprev[i][reg] = nk > 0 ? p.c[d][(long)(r0 + lrow) * H + j0 + hb + fr] : 0.f;                            \
  }
  D2_SEED(acc00, acc01, acc02, acc03, 0)
  D2_SEED(acc10, acc11, acc12, acc13, 1)
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                          // (1)  chunk 0 is in LDS
  MMEGO_STAMP_AT(blockIdx.x, 1, tid == 0);
  if (nk > 0) {
    const int key = (fr >> 1) & 7;
    const int sw0 = ((0 | fq) ^ key) << 2, sw1 = ((4 | fq) ^ key) << 2;       // k-block 0 / 1 of a 32-k chunk
    const float* arow = smem + (rowbase + fr) * 32;
    const float* brow = smem + 2048 + (hb + fr) * 32;
    f32x4 pa0, pa1, pb0, pb1, pb2, pb3, qa0, qa1, qb0, qb1, qb2, qb3;
#define D2_RD(S, so, sw)                                                                                    \
  do {                                                                                                      \
    S##a0 = *reinterpret_cast<const f32x4*>(arow + (so) + (sw));                                            \
    S##a1 = *reinterpret_cast<const f32x4*>(arow + (so) + 16 * 32 + (sw));                                  \
    S##b0 = *reinterpret_cast<const f32x4*>(brow + (so) + (sw));                                            \
    S##b1 = *reinterpret_cast<const f32x4*>(brow + (so) + 32 * 32 + (sw));                                  \
    S##b2 = *reinterpret_cast<const f32x4*>(brow + (so) + 64 * 32 + (sw));                                  \
    S##b3 = *reinterpret_cast<const f32x4*>(brow + (so) + 96 * 32 + (sw));                                  \
  } while (0)
    // Two fragment sets: set p holds the first 16 k of a chunk, set q the second.  Each set's ds_reads are issued one
    // half-chunk (32 MFMAs = 1024 matrix-pipe cycles) ahead of the MFMAs that consume them, so the LDS latency is never
    // exposed; the barrier that hands a stage back to the loaders is waited for with a full set of MFMAs in the pipe.
    int so = 0;
    D2_RD(p, 0, sw0);
    for (int kt = 0; kt < nk; ++kt) {
      D2_RD(q, so, sw1);                                    // last reads of stage kt
      __builtin_amdgcn_sched_barrier(0);
      WS_MM(p);                                             // (waits for set p only: the compiler counts lgkmcnt)
      __builtin_amdgcn_sched_barrier(0);
      if (kt + 1 < nk) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave's reads of stage kt are done (fragments in registers)
        __builtin_amdgcn_s_barrier();                       // B_kt: chunk kt+1 has landed, stage kt may be refilled
        so = (so == 2 * D2_STAGE_FLOATS) ? 0 : so + D2_STAGE_FLOATS;
        D2_RD(p, so, sw0);                                  // first reads of stage kt+1, under set q's MFMAs
        __builtin_amdgcn_sched_barrier(0);
      }
      WS_MM(q);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  MMEGO_STAMP_AT(blockIdx.x, 2, tid == 0);
#define D2_CELL(A0, A1, A2, A3, i)                                                                          \
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
  if (Bn >= 128) {
    static bool d2_attr = false;
    const size_t lds = (size_t)D2_LDS_FLOATS * sizeof(float);
    if (!d2_attr) {
      hipError_t e = hipFuncSetAttribute((const void*)lstm_step_dma2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return (int)e;
      d2_attr = true;
    }
    int grid = ndir * (H / 32) * cdiv(Bn, 64);
    hipLaunchKernelGGL(lstm_step_dma2_kernel, dim3(grid), dim3(512), lds, (hipStream_t)stream, p);
  } else {
    int grid = ndir * (H / 4) * cdiv(Bn, 64);
    if ((H % 64) == 0) hipLaunchKernelGGL(lstm_step_small_kernel<64>, dim3(grid), dim3(256), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(lstm_step_small_kernel<32>, dim3(grid), dim3(256), 0, (hipStream_t)stream, p);
  }
  MMEGO_LAUNCH_CHECK();
  return MMEGO_OK;
}
