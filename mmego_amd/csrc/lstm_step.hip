// One timestep of a (bi)LSTM of hidden size H (H % 32 == 0) on the fp32 matrix cores: the recurrent half of
// IMU_Net's 2 x (2-layer, H=512) BiLSTMs (reference Net/IMU_Net.py:58-62,77,82) -- 94 % of the path's FLOPs.
//   gates = xproj_t (+ b_hh) + h_{t-1} . W_hh^T ;  i,f,o = sigmoid, g = tanh ;  c = f c + i g ;  h = o tanh(c)
// Both directions run in one launch.  Two kernels, chosen by the number of batch rows:
//   lstm_step_dma_kernel<HT>  Bn >= 128: WG = 64 rows x HT (16 | 32) hidden x 4 gates (v_mfma_f32_16x16x4_f32); operands
//                           go global -> LDS by LDS-DMA into a 3-stage ring of 32-k chunks, four loader waves + four
//                           MFMA waves; the cell update is register-local.  WGs sharing a W_hh slice are placed on one
//                           XCD, so each XCD's L2 holds 1/8 of W_hh.
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
#ifdef MMEGO_STAMP
  int dbg;         // diagnostic probe only (scripts/clock_probe.hip): 1 = loaders do not wait for DMAs, 2 = no DMAs, 4 = no ds_reads
#endif
};

// rcp / v_exp_f32 based activations (as in lstm.hip): a few ulp from the libm forms at a fraction of their instruction count
// (libm expf is ~15 VALU instructions; the cell update evaluates 40 of them per lane)
__device__ __forceinline__ float fast_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float fast_tanh(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * x)); }

#define GLDS16(gptr, lptr)                                                                                  \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr),                   \
                                   (__attribute__((address_space(3))) void*)(lptr), 16, 0, 0)
#ifdef MMEGO_STAMP
#define D2_DBG(bit) (p.dbg & (bit))
#else
#define D2_DBG(bit) 0
#endif

// ---- LDS-DMA step kernel -------------------------------------------------------------------------------------------------
// WG = 64 batch rows x HT hidden units x 4 gates, 8 waves: four LOADER waves move the operands global -> LDS by LDS-DMA
// (global_load_lds_dwordx4: never through VGPRs; counted vmcnt, raw s_barriers so that chunks stay in flight across them),
// four COMPUTE waves (one per SIMD) run only ds_read_b128 + v_mfma_f32_16x16x4_f32.  32-k chunks in a ring of three stages.
//   stage image: A 64 rows x 32 k, then W 4*HT rows x 32 k (row = gate*HT + unit), unpadded 128-B rows of 8 16-B pieces;
//   piece p of row r sits at piece p ^ ((r >> 1) & 7): the 16 lanes of a ds_read_b128 phase (rows fr = 0..15, same piece)
//   hit 16 distinct 4-bank groups (PMC: no bank conflicts).
// The accumulators are seeded with xproj + b_hh (gates = xproj + b + h.W^T), fetched straight from global memory while the
// first chunk is in flight; c_{t-1} waits in registers for the cell update, which is register-local.
// HT = 32: 72 KB of LDS, wave tile 32 rows x 16 units x 4 gates (8 accumulator tiles, 6 fragment reads per 32 MFMAs);
// HT = 16: 48 KB, wave tile 16 x 16 x 4 (4 tiles, 5 reads per 16 MFMAs) and twice the workgroups.
// Why two sizes (in-kernel stamps, scripts/clock_probe.hip, Bn = 512, H = 512): with ONE workgroup per CU (256 WGs at
// HT = 32) the product loop takes 40.4 k cycles against 32.8 k of MFMA issue; with the DMAs and the LDS reads taken out it
// still takes 35.7 k -- the matrix pipe drains at every chunk barrier (the in-order compute wave cannot issue past it).
// With HT = 16 there are 512 WGs, two per CU with independent barriers: while one workgroup's compute wave sits at its
// barrier the other one's owns the SIMD's matrix pipe.  HT = 32 remains for grids that fill the chip twice over anyway.
template <int HT>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void lstm_step_dma_kernel(LstmStepP p) {
  constexpr int NA = HT / 16;                  // A fragments (16-row groups) per compute wave
  constexpr int WROWS = 4 * HT;                // W rows per stage
  constexpr int NWP = WROWS / 32;              // W DMA pieces-groups per loader wave and chunk (8 rows each)
  constexpr int STAGE = (64 + WROWS) * 32;     // floats per stage
  constexpr int CLD = HT + 4;                  // row stride of the c / h output tiles in LDS
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool loader = wave >= 4;
  const int H = p.H;
  const int nrb = (p.Bn + 63) / 64, nht = H / HT, npairs = p.ndir * nht;
  int pair, rb;
  {
    const int wg = blockIdx.x;
    if ((npairs & 7) == 0) {                   // workgroups sharing a W_hh slice on one XCD: each L2 holds 1/8 of W_hh
      const int xcd = wg & 7, q = wg >> 3;
      // an XCD takes a CONTIGUOUS run of hidden blocks: at HT = 16 a (row, gate) segment of xproj / c / h is 64 bytes, half a
      // cache line, and the neighbouring hidden block's workgroup reads the other half -- on the same XCD the pair shares one
      // L2 fill (the r02 order was strided: hidden blocks xcd, xcd + 8, ...)
      pair = xcd * (npairs >> 3) + (q / nrb);
      rb = q % nrb;
    } else {
      pair = wg / nrb;
      rb = wg % nrb;
    }
  }
  const int d = pair / nht, ht = pair % nht;
  const int j0 = ht * HT, r0 = rb * 64;
  const bool first = p.first != 0;
  const int nk = first ? 0 : H / 32;
  float* const OUT = smem + (nk % 3) * STAGE;  // new c tile [64][CLD], then new h tile [64][CLD]
  MMEGO_STAMP_AT(blockIdx.x, 0, tid == 0);

  if (loader) {
    const int lw = wave - 4, q8 = lane >> 3, sl = lane & 7;
    const int rmax = p.Bn - 1 - r0;            // rows past the batch read a valid row (their results are dropped)
    const float* ag[2];
    const float* wg_[NWP];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int ra = 8 * (lw + 4 * i) + q8;
      ag[i] = first ? nullptr : p.hprev[d] + (long)(r0 + min(ra, rmax)) * p.hps + 4 * (sl ^ ((ra >> 1) & 7));
    }
#pragma unroll
    for (int j = 0; j < NWP; ++j) {
      const int rw = 8 * (lw + 4 * j) + q8;
      wg_[j] = p.whh[d] + ((long)(rw / HT) * H + j0 + (rw % HT)) * H + 4 * (sl ^ ((rw >> 1) & 7));
    }
#define D2_CHUNK(kt)                                                                                        \
  do {                                                                                                      \
    if (D2_DBG(2)) break;                                                                                   \
    float* st_ = smem + ((kt) % 3) * STAGE;                                                                 \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) GLDS16(ag[i] + (kt) * 32, st_ + 8 * (lw + 4 * i) * 32);   \
    _Pragma("unroll") for (int j = 0; j < NWP; ++j) GLDS16(wg_[j] + (kt) * 32, st_ + 2048 + 8 * (lw + 4 * j) * 32); \
  } while (0)
    constexpr int PC = 2 + NWP;                // DMAs per loader wave and chunk
    if (nk > 0) {
      D2_CHUNK(0);
      if (nk > 1) D2_CHUNK(1);
      if (nk > 2) D2_CHUNK(2);
      if (D2_DBG(1)) { }
      else if (nk > 2) __builtin_amdgcn_s_waitcnt(0x0F70 | ((2 * PC) & 15) | (((2 * PC) >> 4) << 14));
      else if (nk > 1) __builtin_amdgcn_s_waitcnt(0x0F70 | (PC & 15) | ((PC >> 4) << 14));
      else __builtin_amdgcn_s_waitcnt(0x0F70);
    }
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();              // (1) chunk 0 is in LDS
    for (int kt = 0; kt + 1 < nk; ++kt) {      // B_kt: chunk kt+1 has landed; stage kt % 3 may be refilled
      if (D2_DBG(1)) { }
      else if (kt + 2 < nk) __builtin_amdgcn_s_waitcnt(0x0F70 | (PC & 15) | ((PC >> 4) << 14));
      else __builtin_amdgcn_s_waitcnt(0x0F70);
      asm volatile("" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (kt + 3 < nk) D2_CHUNK(kt + 3);
    }
    __builtin_amdgcn_s_barrier();              // (E) new c / h tiles are in OUT
    {
      constexpr int TPR = HT / 8;              // threads per row (two 16-B stores each to c and to h)
      const int lt = tid - 256;
      for (int e = lt; e < 64 * TPR; e += 256) {
        const int xr = e / TPR, xq = (e % TPR) * 8;
        if ((r0 + xr) < p.Bn) {
          float* crow = p.c[d] + (long)(r0 + xr) * H + j0 + xq;
          *reinterpret_cast<f32x4*>(crow) = *reinterpret_cast<const f32x4*>(OUT + xr * CLD + xq);
          *reinterpret_cast<f32x4*>(crow + 4) = *reinterpret_cast<const f32x4*>(OUT + xr * CLD + xq + 4);
          float* hrow = p.hout[d] + (long)(r0 + xr) * p.hos + j0 + xq;
          *reinterpret_cast<f32x4*>(hrow) = *reinterpret_cast<const f32x4*>(OUT + 64 * CLD + xr * CLD + xq);
          *reinterpret_cast<f32x4*>(hrow + 4) = *reinterpret_cast<const f32x4*>(OUT + 64 * CLD + xr * CLD + xq + 4);
        }
      }
    }
    return;
  }

  // ---------------------------------------------- compute waves ----------------------------------------------
  const int rowbase = HT == 32 ? (wave & 1) * 32 : wave * 16;
  const int hb = HT == 32 ? (wave >> 1) * 16 : 0;
  const int fr = lane & 15, fq = lane >> 4;
  const int rmax = p.Bn - 1 - r0;
  float bh[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) bh[g] = p.bhh[d] ? p.bhh[d][g * H + j0 + hb + fr] : 0.f;
  f32x4 acc[NA][4];
  float cprev[NA][4];
#pragma unroll
  for (int i = 0; i < NA; ++i)
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int lrow = min(rowbase + i * 16 + fq * 4 + reg, rmax);
      const float* xrow = p.xproj[d] + (long)(r0 + lrow) * p.xs + j0 + hb + fr;
#pragma unroll
      for (int g = 0; g < 4; ++g) acc[i][g][reg] = xrow[(long)g * H] + bh[g];
      cprev[i][reg] = nk > 0 ? p.c[d][(long)(r0 + lrow) * H + j0 + hb + fr] : 0.f;
    }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                // (1) chunk 0 is in LDS
  MMEGO_STAMP_AT(blockIdx.x, 1, tid == 0);
  if (nk > 0) {
    const int key = (fr >> 1) & 7;
    const int sw0 = ((0 | fq) ^ key) << 2, sw1 = ((4 | fq) ^ key) << 2;       // k-block 0 / 1 of a 32-k chunk
    const float* arow = smem + (rowbase + fr) * 32;
    const float* brow = smem + 2048 + (hb + fr) * 32;
    f32x4 pa[NA], pb[4], qa[NA], qb[4];
#define D2_RD(A_, B_, so, sw)                                                                               \
  do {                                                                                                      \
    if (D2_DBG(4)) break;                                                                                   \
    _Pragma("unroll") for (int i = 0; i < NA; ++i) A_[i] = *reinterpret_cast<const f32x4*>(arow + (so) + i * 16 * 32 + (sw)); \
    _Pragma("unroll") for (int g = 0; g < 4; ++g) B_[g] = *reinterpret_cast<const f32x4*>(brow + (so) + g * HT * 32 + (sw));  \
  } while (0)
#define D2_MM(A_, B_)                                                                                       \
  do {                                                                                                      \
    _Pragma("unroll") for (int c = 0; c < 4; ++c)                                                           \
      _Pragma("unroll") for (int i = 0; i < NA; ++i)                                                        \
        _Pragma("unroll") for (int g = 0; g < 4; ++g)                                                       \
          acc[i][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(A_[i][c], B_[g][c], acc[i][g], 0, 0, 0);         \
  } while (0)
    // Two fragment sets: set p holds the first 16 k of a chunk, set q the second; each set's ds_reads are issued one
    // half-chunk ahead of the MFMAs that consume them, and the barrier that hands a stage back to the loaders is waited
    // for with a full set of MFMAs queued.
    int so = 0;
    D2_RD(pa, pb, 0, sw0);
    for (int kt = 0; kt < nk; ++kt) {
      D2_RD(qa, qb, so, sw1);                  // last reads of stage kt
      __builtin_amdgcn_sched_barrier(0);
      D2_MM(pa, pb);                           // (waits for set p only: the compiler counts lgkmcnt)
      __builtin_amdgcn_sched_barrier(0);
      if (kt + 1 < nk) {
        // this wave's reads of stage kt are done (fragments in registers).  The BUILTIN, not inline asm: the compiler's
        // wait-count pass must know that set q has landed, or it makes q's MFMAs wait for the p reads issued just before them.
        __builtin_amdgcn_s_waitcnt(0xC07F);
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();          // B_kt: chunk kt+1 has landed, stage kt may be refilled
        so = (so == 2 * STAGE) ? 0 : so + STAGE;
        D2_RD(pa, pb, so, sw0);                // first reads of stage kt+1, under set q's MFMAs
        __builtin_amdgcn_sched_barrier(0);
      }
      D2_MM(qa, qb);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  MMEGO_STAMP_AT(blockIdx.x, 2, tid == 0);
#pragma unroll
  for (int i = 0; i < NA; ++i)
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int lrow = rowbase + i * 16 + fq * 4 + reg;
      const float gi = fast_sigmoid(acc[i][0][reg]);
      const float gf = fast_sigmoid(acc[i][1][reg]);
      const float gg = fast_tanh(acc[i][2][reg]);
      const float go = fast_sigmoid(acc[i][3][reg]);
      const float cn = gf * cprev[i][reg] + gi * gg;
      OUT[lrow * CLD + hb + fr] = cn;
      OUT[64 * CLD + lrow * CLD + hb + fr] = go * fast_tanh(cn);
      if (p.gst[d] && (r0 + lrow) < p.Bn) {
        float* gs = p.gst[d] + (long)(r0 + lrow) * 4 * H + j0 + hb + fr;
        gs[0] = gi; gs[H] = gf; gs[2 * H] = gg; gs[3 * H] = go;
        p.cst[d][(long)(r0 + lrow) * H + j0 + hb + fr] = cn;
      }
    }
  MMEGO_STAMP_AT(blockIdx.x, 3, tid == 0);
  __syncthreads();                             // (E)
}

// ---- small-batch variant: WG = 64 rows x (4 hidden x 4 gates), K split over nothing, 4 waves = 4 row tiles ----------
// FULL: Bn is a multiple of the 64 rows of a workgroup -- no row predicates, so the prologue's loads (4 input projections, 4 cell
// states per lane) and every chunk's operand loads are straight-line code.  Predicated row by row they were a chain of
// "branch, load, wait for everything" blocks: four dependent memory round trips before the product even started.
// NK > 0 (= H / SK, whole row blocks): the loads of ALL NK chunks are issued before the first one is used (NK x 5 16-byte
// registers per thread) and each chunk waits only for its own -- with one chunk prefetched at a time, under a condition, the
// step was a chain of NK memory round trips (the prefetched values were waited for in front of the MFMAs they should have
// overlapped): 10.0 us per launch at Bn = 64, H = 512.
template <int SK, bool FULL, int NK = 0>  // SK: k per staged chunk (128 / 64, or 32 when H is not a multiple of 64)
__global__ __launch_bounds__(256) void lstm_step_small_kernel(LstmStepP p) {
  // two stages of A [64 rows][SK + 4] and B [16 gate columns][SK + 4] (dynamic LDS: 84.5 KB at SK = 128)
  extern __shared__ __attribute__((aligned(16))) float sm_small[];
  constexpr int LDK = SK + 4;
  float* const As = sm_small;                       // [2][64][LDK]
  float* const Bs = sm_small + 2 * 64 * LDK;        // [2][16][LDK]
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
    const bool ok = FULL || row < p.Bn;
    xpv[reg] = ok ? p.xproj[d][(long)row * p.xs + gate * H + j] + bh : 0.f;
    // (every lane loads it -- only the gate-0 lanes use it: a lane-divergent predicate would put a branch around the load)
    cprev[reg] = (ok && !p.first) ? p.c[d][(long)row * H + j] : 0.f;
  }
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  if (!p.first) {
    const float* hp = p.hprev[d];
    const float* W = p.whh[d];
    // staging per chunk of SK k: a thread moves f32x4 pieces; LPR lanes cover a row's SK floats, RPP rows per pass of the
    // 256 threads, NA passes for the 64 rows of A, NB passes (or a quarter of the threads) for the 16 rows of B
    constexpr int LPR = SK / 4, RPP = 256 / LPR, NA = 64 / RPP, NB = RPP >= 16 ? 1 : 16 / RPP;
    const int lk = (tid % LPR) * 4, lr = tid / LPR;
    const f32x4 zero4 = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int nk = H / SK;
    const float* ap = hp + (long)(r0 + lr) * p.hps + lk;
    const long rsp = (long)RPP * p.hps;
    const float* wp[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const int rowb = (lr + b * RPP) & 15;         // = gate*4 + jl of the staged W row
      wp[b] = W + ((long)(rowb >> 2) * H + j0 + (rowb & 3)) * H + lk;
    }
    f32x4 ra[NA], rbv[NB];
#define SM_GLOAD(k0)                                                                                          \
  do {                                                                                                        \
    _Pragma("unroll") for (int a = 0; a < NA; ++a)                                                            \
      ra[a] = (FULL || (r0 + lr + a * RPP) < p.Bn) ? *reinterpret_cast<const f32x4*>(ap + a * rsp + (k0)) : zero4; \
    _Pragma("unroll") for (int b = 0; b < NB; ++b)                                                            \
      if (RPP <= 16 || lr < 16) rbv[b] = *reinterpret_cast<const f32x4*>(wp[b] + (k0));                      \
  } while (0)
#define SM_SSTORE(buf)                                                                                        \
  do {                                                                                                        \
    _Pragma("unroll") for (int a = 0; a < NA; ++a)                                                            \
      *reinterpret_cast<f32x4*>(&As[((buf) * 64 + lr + a * RPP) * LDK + lk]) = ra[a];                        \
    _Pragma("unroll") for (int b = 0; b < NB; ++b)                                                            \
      if (RPP <= 16 || lr < 16) *reinterpret_cast<f32x4*>(&Bs[((buf) * 16 + lr + b * RPP) * LDK + lk]) = rbv[b]; \
  } while (0)
#define SM_MFMA(buf)                                                                                          \
  _Pragma("unroll") for (int kb = 0; kb < SK / 16; ++kb) {                                                    \
    f32x4 a = *reinterpret_cast<const f32x4*>(&As[((buf) * 64 + wave * 16 + fr) * LDK + kb * 16 + 4 * fq]);  \
    f32x4 b = *reinterpret_cast<const f32x4*>(&Bs[((buf) * 16 + fr) * LDK + kb * 16 + 4 * fq]);              \
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, acc0, 0, 0, 0);                                    \
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, acc1, 0, 0, 0);                                    \
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, acc0, 0, 0, 0);                                    \
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, acc1, 0, 0, 0);                                    \
  }
    if (NK > 0) {
      // every chunk's loads in flight at once; chunk kt is pinned (waited for) only when it is stored to LDS
      constexpr int NKK = NK > 0 ? NK : 1;
      f32x4 qa[NKK][NA], qb[NKK][NB];
#pragma unroll
      for (int kt = 0; kt < NKK; ++kt) {
#pragma unroll
        for (int a = 0; a < NA; ++a) qa[kt][a] = *reinterpret_cast<const f32x4*>(ap + a * rsp + kt * SK);
#pragma unroll
        for (int b = 0; b < NB; ++b) qb[kt][b] = *reinterpret_cast<const f32x4*>(wp[b] + kt * SK);
      }
#pragma unroll
      for (int kt = 0; kt < NKK; ++kt) {
        const int buf = kt & 1;
#pragma unroll
        for (int a = 0; a < NA; ++a) {
          asm volatile("" : "+v"(qa[kt][a].x), "+v"(qa[kt][a].y), "+v"(qa[kt][a].z), "+v"(qa[kt][a].w));
          *reinterpret_cast<f32x4*>(&As[(buf * 64 + lr + a * RPP) * LDK + lk]) = qa[kt][a];
        }
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          asm volatile("" : "+v"(qb[kt][b].x), "+v"(qb[kt][b].y), "+v"(qb[kt][b].z), "+v"(qb[kt][b].w));
          if (RPP <= 16 || lr < 16) *reinterpret_cast<f32x4*>(&Bs[(buf * 16 + lr + b * RPP) * LDK + lk]) = qb[kt][b];
        }
        __syncthreads();                         // (two LDS stages: chunk kt+1 is stored while others may still read chunk kt-1's
        SM_MFMA(buf)                             //  stage -- which the barrier of chunk kt has already retired)
      }
    } else {
      SM_GLOAD(0);
      SM_SSTORE(0);
      __syncthreads();
      for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) SM_GLOAD((kt + 1) * SK);
        SM_MFMA(buf)
        if (kt + 1 < nk) SM_SSTORE(buf ^ 1);
        __syncthreads();
      }
    }
#undef SM_MFMA
#undef SM_GLOAD
#undef SM_SSTORE
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

// ---- the FIRST timestep of a sequence (h = c = 0): no product, gates = xproj + b_hh -- an elementwise pass ------------------
// (Through the step kernels it cost their whole fixed part -- operand staging skipped, but workgroup set-up, the LDS output tile
// and its barriers kept: 5.8-6.4 us per launch at Bn = 512; eight such launches per U+L step.)  Same expressions as the step
// kernels' cell update with c_{t-1} = 0: same bits.  One thread per row and four hidden units.
__global__ __launch_bounds__(256) void lstm_first_step_kernel(LstmStepP p) {
  const int d = blockIdx.y, H = p.H, H4 = H >> 2;
  const long n = (long)p.Bn * H4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const long row = i / H4;
    const int j = (int)(i - row * H4) * 4;
    const float* xr = p.xproj[d] + row * p.xs + j;
    f32x4 x[4], b[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) x[g] = *reinterpret_cast<const f32x4*>(xr + (long)g * H);
    if (p.bhh[d]) {
#pragma unroll
      for (int g = 0; g < 4; ++g) b[g] = *reinterpret_cast<const f32x4*>(p.bhh[d] + g * H + j);
    } else {
#pragma unroll
      for (int g = 0; g < 4; ++g) b[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    f32x4 gi, gf, gg, go, cn, hn;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      gi[u] = fast_sigmoid(x[0][u] + b[0][u]);
      gf[u] = fast_sigmoid(x[1][u] + b[1][u]);
      gg[u] = fast_tanh(x[2][u] + b[2][u]);
      go[u] = fast_sigmoid(x[3][u] + b[3][u]);
      cn[u] = gi[u] * gg[u];                     // (= gf * 0 + gi * gg of the step kernels)
      hn[u] = go[u] * fast_tanh(cn[u]);
    }
    *reinterpret_cast<f32x4*>(p.c[d] + row * H + j) = cn;
    *reinterpret_cast<f32x4*>(p.hout[d] + row * p.hos + j) = hn;
    if (p.gst[d]) {
      float* gs = p.gst[d] + row * 4 * H + j;
      *reinterpret_cast<f32x4*>(gs) = gi;
      *reinterpret_cast<f32x4*>(gs + H) = gf;
      *reinterpret_cast<f32x4*>(gs + 2 * H) = gg;
      *reinterpret_cast<f32x4*>(gs + 3 * H) = go;
      *reinterpret_cast<f32x4*>(p.cst[d] + row * H + j) = cn;
    }
  }
}

#ifdef MMEGO_STAMP
static int mmego_step_dbg = 0;
#endif
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
#ifdef MMEGO_STAMP
  p.dbg = mmego_step_dbg;
#endif
  const uintptr_t al = (uintptr_t)bhh0 | (uintptr_t)bhh1 | (uintptr_t)gst0 | (uintptr_t)gst1 | (uintptr_t)cst0 | (uintptr_t)cst1 |
                       (uintptr_t)c1 | (uintptr_t)hout1 | (uintptr_t)xproj1;
  if (first && (H % 4) == 0 && (al & 15) == 0) {
    long b = ((long)Bn * (H / 4) + 255) / 256;
    hipLaunchKernelGGL(lstm_first_step_kernel, dim3((unsigned)(b > 2048 ? 2048 : b), ndir), dim3(256), 0, (hipStream_t)stream, p);
  } else if (Bn >= 128) {
    // HT = 32 by default (the HT = 16 two-workgroups-per-CU variant: shorter product loop, 37.9 k against 41.1 k cycles, but its
    // prologue -- twice the workgroups fetching first chunks -- costs more)
    static int ncu = 0;
    if (!ncu) {
      int dev = 0;
      (void)hipGetDevice(&dev);
      if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu <= 0) ncu = 256;
    }
    const int grid32 = ndir * (H / 32) * cdiv(Bn, 64);
    // (both directions in one launch, Bn = H = 512: 26.7 us with HT 16 against 24.3 us with HT 32 per step.)  A launch that
    // would fill at most HALF the chip with HT = 32 -- one direction of the pair, launched on its own stream -- takes HT = 16:
    // then each direction's grid covers every CU once and the two directions' workgroups share the CUs out of phase
    const bool ht16 = 2 * grid32 <= ncu;
    static bool attr32 = false, attr16 = false;
    if (ht16) {
      const size_t lds = (size_t)3 * (64 + 64) * 32 * sizeof(float);
      if (!attr16) {
        hipError_t e = hipFuncSetAttribute((const void*)lstm_step_dma_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        attr16 = true;
      }
      hipLaunchKernelGGL(lstm_step_dma_kernel<16>, dim3(2 * grid32), dim3(512), lds, (hipStream_t)stream, p);
    } else {
      const size_t lds = (size_t)3 * (64 + 128) * 32 * sizeof(float);
      if (!attr32) {
        hipError_t e = hipFuncSetAttribute((const void*)lstm_step_dma_kernel<32>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        attr32 = true;
      }
      hipLaunchKernelGGL(lstm_step_dma_kernel<32>, dim3(grid32), dim3(512), lds, (hipStream_t)stream, p);
    }
  } else {
    int grid = ndir * (H / 4) * cdiv(Bn, 64);
    const bool full = (Bn % 64) == 0;
    // (128-k chunks -- half as many dependent chunk round trips, 84.5 KB of LDS -- measured the same 9.9-10.0 us: not kept)
#define SMALL_LAUNCH(SK_, F_)                                                                                        \
  do {                                                                                                               \
    const size_t lds = (size_t)2 * (64 + 16) * (SK_ + 4) * sizeof(float);                                            \
    static bool attr_set = false;                                                                                    \
    if (!attr_set) {                                                                                                 \
      hipError_t e = hipFuncSetAttribute((const void*)lstm_step_small_kernel<SK_, F_>,                               \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                      \
      if (e != hipSuccess) return (int)e;                                                                            \
      attr_set = true;                                                                                               \
    }                                                                                                                \
    hipLaunchKernelGGL((lstm_step_small_kernel<SK_, F_>), dim3(grid), dim3(256), lds, (hipStream_t)stream, p);       \
  } while (0)
    if (H == 512 && full) {              // whole row blocks at IMU_Net's width: all eight chunks' loads up front
      const size_t lds = (size_t)2 * (64 + 16) * (64 + 4) * sizeof(float);
      hipLaunchKernelGGL((lstm_step_small_kernel<64, true, 8>), dim3(grid), dim3(256), lds, (hipStream_t)stream, p);
    }
    else if ((H % 64) == 0) { if (full) SMALL_LAUNCH(64, true); else SMALL_LAUNCH(64, false); }
    else { if (full) SMALL_LAUNCH(32, true); else SMALL_LAUNCH(32, false); }
#undef SMALL_LAUNCH
  }
  MMEGO_LAUNCH_CHECK();
  return MMEGO_OK;
}
