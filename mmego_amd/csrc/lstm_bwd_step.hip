// One step of the BACKWARD recurrence of a BiLSTM layer (stage-1 training of IMU_Net: reference autograd of nn.LSTM,
// Processor/Train/Train_IMU.py:114-149), both directions per launch:
//     dh_rec = dgates_{s+1} [Bn][4H] . W_hh     (given transposed: wT [H][4H], so both operands are K-contiguous, K = 4H)
// with the cell backward of step s applied on the product's tiles (dh = dout + dh_rec; dh_rec never reaches memory).
// Same design as the forward step (lstm_step.hip, lstm_step_dma_kernel): WG = 64 batch rows x 32 hidden units, four LOADER waves
// move the operands global -> LDS by LDS-DMA (global_load_lds_dwordx4, counted vmcnt, raw s_barriers), four COMPUTE waves run only
// ds_read_b128 + v_mfma_f32_16x16x4_f32.  The product has the forward step's FLOPs with a four times longer K and a four times
// narrower output, so the chunks are 64 k deep (one barrier per 32 MFMAs of a compute wave, as in the forward kernel):
//   stage image: A 64 rows x 64 k, then W 32 rows x 64 k, unpadded 256-B rows of 16 16-B pieces; piece p of row r sits at
//   p ^ (r & 15): the 16 lanes of a ds_read_b128 phase (rows fr = 0..15, same piece) hit 16 distinct 4-bank groups.
// Three stages of 24 KB.  The cell backward's operands (dout, the gate / cell stashes, dc) are requested before the product loop
// and wait in registers; its results leave straight from registers.
// It replaces the K-quartered 32 x 32-tile product with the cell-backward epilogue (gemm.hip, gemm32kq_kernel: 36.8 us per
// timestep, 268 MB of operands through the vector-memory path per launch against 196 MB through LDS-DMA here).
#include <stdlib.h>

#include "common.h"

struct LstmBwdStepP {
  const float* dg[2]; long dgs;          // dgates of the step before in backward order: rows Bn, K = 4H contiguous
  const float* wT[2];                    // W_hh transposed [H][4H]
  const float* dout[2]; long dos;
  const float* gst[2]; const float* cst[2]; const float* cprev[2];
  float* dc[2]; float* dgo[2];
  int Bn, H;
};

#define BGLDS16(gptr, lptr)                                                                                 \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr),                   \
                                   (__attribute__((address_space(3))) void*)(lptr), 16, 0, 0)

__global__ __launch_bounds__(512) void lstm_bwd_step_dma_kernel(LstmBwdStepP p) {
  constexpr int HT = 32, KC = 64;
  constexpr int STAGE = (64 + HT) * KC;        // floats per stage
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool loader = wave >= 4;
  const int H = p.H, K = 4 * H;
  const int nrb = p.Bn / 64, nht = H / HT, npairs = 2 * nht;
  int pair, rb;
  {
    const int wg = blockIdx.x;
    if ((npairs & 7) == 0) {                   // workgroups sharing a W_hh slice on one XCD (a contiguous run of hidden blocks)
      const int xcd = wg & 7, q = wg >> 3;
      pair = xcd * (npairs >> 3) + (q / nrb);
      rb = q % nrb;
    } else {
      pair = wg / nrb;
      rb = wg % nrb;
    }
  }
  const int d = pair / nht, ht = pair % nht;
  const int j0 = ht * HT, r0 = rb * 64;
  const int nk = K / KC;

  if (loader) {
    const int lw = wave - 4, q16 = lane >> 4, sl = lane & 15;
    const float* ag[4];
    const float* wg_[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int ra = 4 * (lw + 4 * i) + q16;
      ag[i] = p.dg[d] + (long)(r0 + ra) * p.dgs + 4 * (sl ^ (ra & 15));
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int rw = 4 * (lw + 4 * j) + q16;
      wg_[j] = p.wT[d] + (long)(j0 + rw) * K + 4 * (sl ^ (rw & 15));
    }
#define B2_CHUNK(kt)                                                                                        \
  do {                                                                                                      \
    float* st_ = smem + ((kt) % 3) * STAGE;                                                                 \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) BGLDS16(ag[i] + (kt) * KC, st_ + 4 * (lw + 4 * i) * KC);  \
    _Pragma("unroll") for (int j = 0; j < 2; ++j) BGLDS16(wg_[j] + (kt) * KC, st_ + 64 * KC + 4 * (lw + 4 * j) * KC); \
  } while (0)
    constexpr int PC = 6;                      // DMAs per loader wave and chunk
    B2_CHUNK(0);
    if (nk > 1) B2_CHUNK(1);
    if (nk > 2) B2_CHUNK(2);
    if (nk > 2) __builtin_amdgcn_s_waitcnt(0x0F70 | ((2 * PC) & 15) | (((2 * PC) >> 4) << 14));
    else if (nk > 1) __builtin_amdgcn_s_waitcnt(0x0F70 | (PC & 15) | ((PC >> 4) << 14));
    else __builtin_amdgcn_s_waitcnt(0x0F70);
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();              // (1) chunk 0 is in LDS
    for (int kt = 0; kt + 1 < nk; ++kt) {      // B_kt: chunk kt+1 has landed; stage kt % 3 may be refilled
      if (kt + 2 < nk) __builtin_amdgcn_s_waitcnt(0x0F70 | (PC & 15) | ((PC >> 4) << 14));
      else __builtin_amdgcn_s_waitcnt(0x0F70);
      asm volatile("" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (kt + 3 < nk) B2_CHUNK(kt + 3);
    }
#undef B2_CHUNK
    return;
  }

  // ---------------------------------------------- compute waves ----------------------------------------------
  const int rowbase = (wave & 1) * 32, hb = (wave >> 1) * 16;
  const int fr = lane & 15, fq = lane >> 4;
  // the cell backward's operands of this lane's 8 elements (rows rowbase + 16 i + 4 fq + reg, unit j0 + hb + fr): requested now,
  // consumed behind the product loop
  float dh[2][4], gi[2][4], gf[2][4], gg[2][4], go[2][4], cc[2][4], cp[2][4], dcin[2][4];
  const long jj = j0 + hb + fr;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const long r = r0 + rowbase + 16 * i + 4 * fq + reg;
      const float* gs = p.gst[d] + r * 4 * H + jj;
      dh[i][reg] = p.dout[d][r * p.dos + jj];
      gi[i][reg] = gs[0]; gf[i][reg] = gs[H]; gg[i][reg] = gs[2 * H]; go[i][reg] = gs[3 * H];
      cc[i][reg] = p.cst[d][r * H + jj];
      cp[i][reg] = p.cprev[d] ? p.cprev[d][r * H + jj] : 0.f;
      dcin[i][reg] = p.dc[d][r * H + jj];
    }
  f32x4 acc[2];
  acc[0] = (f32x4){0.f, 0.f, 0.f, 0.f};
  acc[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();                // (1) chunk 0 is in LDS
  {
    const float* arow = smem + (rowbase + fr) * KC;
    const float* brow = smem + 64 * KC + (hb + fr) * KC;
    f32x4 pa[2], pb, qa[2], qb;
#define B2_RD(A_, B_, so, kb)                                                                               \
  do {                                                                                                      \
    const int sw_ = ((((kb) << 2) | fq) ^ fr) << 2;                                                         \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) A_[i] = *reinterpret_cast<const f32x4*>(arow + (so) + i * 16 * KC + sw_); \
    B_ = *reinterpret_cast<const f32x4*>(brow + (so) + sw_);                                                \
  } while (0)
#define B2_MM(A_, B_)                                                                                       \
  do {                                                                                                      \
    _Pragma("unroll") for (int c = 0; c < 4; ++c)                                                           \
      _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                         \
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(A_[i][c], B_[c], acc[i], 0, 0, 0);                    \
  } while (0)
    // two fragment sets alternate over the four 16-k blocks of a chunk; each set's reads are issued one block ahead of the MFMAs
    // that consume them, and the barrier that hands a stage back to the loaders is waited for with a block of MFMAs queued
    int so = 0;
    B2_RD(pa, pb, 0, 0);
    for (int kt = 0; kt < nk; ++kt) {
      B2_RD(qa, qb, so, 1);
      __builtin_amdgcn_sched_barrier(0);
      B2_MM(pa, pb);
      __builtin_amdgcn_sched_barrier(0);
      B2_RD(pa, pb, so, 2);
      __builtin_amdgcn_sched_barrier(0);
      B2_MM(qa, qb);
      __builtin_amdgcn_sched_barrier(0);
      B2_RD(qa, qb, so, 3);                    // last reads of stage kt
      __builtin_amdgcn_sched_barrier(0);
      B2_MM(pa, pb);
      __builtin_amdgcn_sched_barrier(0);
      if (kt + 1 < nk) {
        __builtin_amdgcn_s_waitcnt(0xC07F);    // this wave's reads of stage kt are done (the builtin: the compiler's wait-count pass sees it)
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();          // B_kt: chunk kt+1 has landed, stage kt may be refilled
        so = (so == 2 * STAGE) ? 0 : so + STAGE;
        B2_RD(pa, pb, so, 0);                  // first reads of stage kt+1, under set q's MFMAs
        __builtin_amdgcn_sched_barrier(0);
      }
      B2_MM(qa, qb);
      __builtin_amdgcn_sched_barrier(0);
    }
#undef B2_RD
#undef B2_MM
  }
  // LSTM cell backward on the tile (the expressions of lstm_cell_bwd_kernel, imu_train.hip)
  float o0[2][4], o1[2][4], o2[2][4], o3[2][4], dco[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const float dhv = dh[i][reg] + acc[i][reg];
      const float tc = tanhf(cc[i][reg]);
      const float dcv = dcin[i][reg] + dhv * go[i][reg] * (1.f - tc * tc);
      o0[i][reg] = dcv * gg[i][reg] * gi[i][reg] * (1.f - gi[i][reg]);
      o1[i][reg] = dcv * cp[i][reg] * gf[i][reg] * (1.f - gf[i][reg]);
      o2[i][reg] = dcv * gi[i][reg] * (1.f - gg[i][reg] * gg[i][reg]);
      o3[i][reg] = dhv * tc * go[i][reg] * (1.f - go[i][reg]);
      dco[i][reg] = dcv * gf[i][reg];
    }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const long r = r0 + rowbase + 16 * i + 4 * fq + reg;
      float* dgp = p.dgo[d] + r * p.dgs + jj;
      dgp[0] = o0[i][reg]; dgp[H] = o1[i][reg]; dgp[2 * H] = o2[i][reg]; dgp[3 * H] = o3[i][reg];
      p.dc[d][r * H + jj] = dco[i][reg];
    }
}

// returns MMEGO_OK when the launch was taken, -1 when the shape is not this kernel's (the caller falls back to gemm32kq)
int lstm_bwd_step_dma_try(void* stream, int Bn, int H, const float* dg0, const float* dg1, long dgs, const float* wT0,
                                           const float* wT1, const float* dout0, const float* dout1, long dos, const float* gst0,
                                           const float* gst1, const float* cst0, const float* cst1, const float* cprev0,
                                           const float* cprev1, float* dc0, float* dc1, float* dgo0, float* dgo1) {
  const char* env = getenv("MMEGO_LSTM_BWD_DMA");       // (read per call: the tests switch it)
  const int on = env ? atoi(env) : 1;
  if (!on || Bn % 64 != 0 || H % 32 != 0 || (dgs & 3) != 0) return -1;
  if (((((uintptr_t)dg0) | ((uintptr_t)dg1) | ((uintptr_t)wT0) | ((uintptr_t)wT1)) & 15) != 0) return -1;
  LstmBwdStepP p;
  p.dg[0] = dg0; p.dg[1] = dg1; p.dgs = dgs;
  p.wT[0] = wT0; p.wT[1] = wT1;
  p.dout[0] = dout0; p.dout[1] = dout1; p.dos = dos;
  p.gst[0] = gst0; p.gst[1] = gst1; p.cst[0] = cst0; p.cst[1] = cst1; p.cprev[0] = cprev0; p.cprev[1] = cprev1;
  p.dc[0] = dc0; p.dc[1] = dc1; p.dgo[0] = dgo0; p.dgo[1] = dgo1;
  p.Bn = Bn; p.H = H;
  constexpr int lds = 3 * (64 + 32) * 64 * 4;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)lstm_bwd_step_dma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  const int grid = 2 * (Bn / 64) * (H / 32);
  hipLaunchKernelGGL(lstm_bwd_step_dma_kernel, dim3(grid), dim3(512), lds, (hipStream_t)stream, p);
  MMEGO_LAUNCH_CHECK();
  return MMEGO_OK;
}
