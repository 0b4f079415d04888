// fp32 GEMM, large-tile kernels: C[M,N] (+)= A . B (+bias)(relu) for 64-aligned shapes, any operand orientation.
//
// The dense products of the path run here: the LSTM input projections of IMU_Net (reference Net/IMU_Net.py:58-62:
// 10240 x 2048 x {512,1024}, 512 x 2048 x 1024), the 64-aligned Linear layers, and -- for stage-1 training
// (Processor/Train/Train_IMU.py:114-149) -- the input-gradient (dX = dY . W) and weight-gradient (dW = dY^T . X) products
// of those layers, which differ only in which index of an operand is contiguous in memory.
//
//  * v_mfma_f32_32x32x2_f32 (exact fp32 fma chain), BM x BN block tile (128x128, 128x64 or 64x64), 4 waves as 2x2, each
//    wave (BM/2) x (BN/2) = TM x TN tiles of 32x32.
//  * Operand staging, 32 k per chunk (64 optional), one LDS buffer + register prefetch of the next chunk (global loads are
//    issued before the MFMA block and land under it); 2 workgroups per CU (36.9 KB each at 128x128) overlap each other's
//    barriers.
//      - k-contiguous operand (X[row][k]): LDS tile [row][k] with a (chunk+4)-float row stride; global->LDS is a plain
//        f32x4 -> ds_write_b128 and the operand fetch is ONE conflict-free ds_read_b128 per 4 MFMA steps through a
//        k-permutation (lane half h supplies k = 8 kb + 4 h + s at step s -- identically for both operands).
//      - row-contiguous operand (X[k][row], e.g. dY^T and X in dW = dY^T . X): LDS tile [k][row] with a (rows+4)-float
//        stride, f32x4 loads/stores along the rows, four conflict-free ds_read_b32 per 4 MFMA steps in the same k order.
//  * Split-K (weight gradients: few tiles, K = batch rows): the K range is cut into `nsplit` slabs whose partial tiles go
//    to a workspace [nsplit][M][N]; the caller reduces them in a fixed order (deterministic).
//  * XCD-aware tile order: consecutive tiles along N (sharing the A row panel) stay on one XCD's L2.
#include <stdlib.h>

#include "common.h"
#include "gemm_tile.h"

#define TLD_MAX 68
// NOTE: staging registers are ext_vector f32x4 (not HIP's float4 struct): arrays of the struct type are left in
// scratch memory by hipcc (ROCm 7.2), which serialises the prefetch.

template <int BM, int BN, int WAVES_M, int WAVES_N, bool A_KC, bool B_KC, int KCH>
__device__ __forceinline__ void gemm_tile_body(const TileP& p, int m0, int n0, int sb, float* smem, int sid) {
  const int split = sb % p.nsplit, batch = sb / p.nsplit;      // sb = batch * nsplit + split
  static_assert(WAVES_M * WAVES_N == 4, "4 waves");
  static_assert(KCH == 64 || KCH == 32, "k per staged chunk");
  constexpr int TLD = KCH + 4;                        // [row][k] row stride: 68 / 36 floats, conflict-free ds_read_b128
  constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;   // wave tile
  constexpr int TM = WM / 32, TN = WN / 32;           // 32x32 MFMA tiles per wave
  constexpr int AV = BM * KCH / 1024, BV = BN * KCH / 1024;   // f32x4 per thread per chunk
  constexpr int LPR = KCH / 4, RPP = 256 / LPR;      // [row][k] staging: lanes per row segment, rows per pass
  constexpr int ALD = BM + 4, BLD = BN + 4;          // row strides of the [k][row] tiles
  constexpr int ARP = 1024 / BM, BRP = 1024 / BN;    // k rows covered per pass of the 256 threads ([k][row] staging)
  float* As = smem;
  float* Bs = smem + BM * TLD;                        // (BM * TLD >= KCH * ALD for BM in {64, 128})
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave % WAVES_M, wn = wave / WAVES_M;
  const int kbeg = split * p.kchunk, kend = min(p.K, kbeg + p.kchunk);
  const int nk = (kend - kbeg) / KCH;

  // staging coordinates
  const int lk = (tid % LPR) * 4, lr = tid / LPR;    // [row][k]: LPR lanes cover one row segment of the chunk, rows lr + RPP i
  const int amk = tid / (BM / 4), am4 = (tid % (BM / 4)) * 4;   // [k][row]: k rows amk + ARP i, 4 rows at am4
  const int bnk = tid / (BN / 4), bn4 = (tid % (BN / 4)) * 4;
  const float* Ab = p.A + (long)batch * p.sAb;
  const float* Wb = p.W + (long)batch * p.sWb;
  const float* Ap = A_KC ? Ab + (long)(m0 + lr) * p.lda + kbeg + lk : Ab + (long)(kbeg + amk) * p.lda + m0 + am4;
  const float* Wp = B_KC ? Wb + (long)(n0 + lr) * p.ldw + kbeg + lk : Wb + (long)(kbeg + bnk) * p.ldw + n0 + bn4;
  const long astep = A_KC ? RPP * p.lda : (long)ARP * p.lda, akstep = A_KC ? KCH : KCH * p.lda;
  const long bstep = B_KC ? RPP * p.ldw : (long)BRP * p.ldw, bkstep = B_KC ? KCH : KCH * p.ldw;
  f32x4 ra[AV], rb[BV];
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = (f32x16){0};

  MMEGO_STAMP_AT(sid, 0, tid == 0);
  if (nk > 0) {
#pragma unroll
    for (int i = 0; i < AV; ++i) ra[i] = *reinterpret_cast<const f32x4*>(Ap + i * astep);
#pragma unroll
    for (int i = 0; i < BV; ++i) rb[i] = *reinterpret_cast<const f32x4*>(Wp + i * bstep);
  }
  const int r = lane & 31, h = lane >> 5;
  MMEGO_STAMP_AT(sid, 1, tid == 0);
  for (int kt = 0; kt < nk; ++kt) {
    __syncthreads();                                  // previous chunk's operand reads are done
#pragma unroll
    for (int i = 0; i < AV; ++i) {
      if (A_KC) *reinterpret_cast<f32x4*>(&As[(lr + RPP * i) * TLD + lk]) = ra[i];
      else *reinterpret_cast<f32x4*>(&As[(amk + ARP * i) * ALD + am4]) = ra[i];
    }
#pragma unroll
    for (int i = 0; i < BV; ++i) {
      if (B_KC) *reinterpret_cast<f32x4*>(&Bs[(lr + RPP * i) * TLD + lk]) = rb[i];
      else *reinterpret_cast<f32x4*>(&Bs[(bnk + BRP * i) * BLD + bn4]) = rb[i];
    }
    __syncthreads();
    if (kt + 1 < nk) {
      const float* An = Ap + (kt + 1) * akstep;
      const float* Wn = Wp + (kt + 1) * bkstep;
#pragma unroll
      for (int i = 0; i < AV; ++i) ra[i] = *reinterpret_cast<const f32x4*>(An + i * astep);
#pragma unroll
      for (int i = 0; i < BV; ++i) rb[i] = *reinterpret_cast<const f32x4*>(Wn + i * bstep);
    }
#pragma unroll
    for (int kb = 0; kb < KCH / 8; ++kb) {
      f32x4 a[TM], b[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        if (A_KC) {
          a[i] = *reinterpret_cast<const f32x4*>(&As[(wm * WM + i * 32 + r) * TLD + kb * 8 + 4 * h]);
        } else {
          const float* q = &As[(kb * 8 + 4 * h) * ALD + wm * WM + i * 32 + r];
          a[i] = (f32x4){q[0], q[ALD], q[2 * ALD], q[3 * ALD]};
        }
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        if (B_KC) {
          b[j] = *reinterpret_cast<const f32x4*>(&Bs[(wn * WN + j * 32 + r) * TLD + kb * 8 + 4 * h]);
        } else {
          const float* q = &Bs[(kb * 8 + 4 * h) * BLD + wn * WN + j * 32 + r];
          b[j] = (f32x4){q[0], q[BLD], q[2 * BLD], q[3 * BLD]};
        }
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].x, b[j].x, acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].y, b[j].y, acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].z, b[j].z, acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].w, b[j].w, acc[i][j], 0, 0, 0);
    }
  }
  MMEGO_STAMP_AT(sid, 2, tid == 0);

  const bool slab = p.nsplit > 1;
  float* C = slab ? p.ws + ((long)split * p.nbatch + batch) * p.M * p.N : p.C + (long)batch * p.sCb;
  const long ldc = slab ? (long)p.N : p.ldc;
  const bool relu = !slab && p.relu, accumulate = !slab && p.accumulate;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = n0 + wn * WN + j * 32 + (lane & 31);
    const float bv = (!slab && p.bias) ? p.bias[(long)batch * p.sBiasb + col] : 0.0f;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      float* cp = C + (long)(m0 + wm * WM + i * 32 + 4 * (lane >> 5)) * ldc + col;
      float old[16];
      if (accumulate) {                                    // all 16 reads before the first store (a load behind a store waits for it too)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) old[reg] = cp[(long)((reg & 3) + 8 * (reg >> 2)) * ldc];
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) asm volatile("" : "+v"(old[reg]));
      }
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        float v = acc[i][j][reg] + bv;
        if (relu) v = fmaxf(v, 0.0f);
        if (accumulate) v += old[reg];
        cp[(long)((reg & 3) + 8 * (reg >> 2)) * ldc] = v;
      }
    }
  }
  MMEGO_STAMP_AT(sid, 3, tid == 0);
}

// XCD-aware tile order: blocks b and b+8 share an XCD; hand each XCD a contiguous run of tile ids
__device__ __forceinline__ int xcd_order(int id, int n) { return (n & 7) == 0 ? (id & 7) * (n >> 3) + (id >> 3) : id; }

// work unit u = (batch * nsplit + split) * tiles + tile
template <int BM, int BN, int WAVES_M, int WAVES_N, bool A_KC, bool B_KC, int KCH>
__global__ __launch_bounds__(256) void gemm_tile_kernel(TileP p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int ntn = p.N / BN, tiles = ntn * (p.M / BM);
  const int u = xcd_order(blockIdx.x, (int)gridDim.x);
  const int split = u / tiles, id = u % tiles;
  gemm_tile_body<BM, BN, WAVES_M, WAVES_N, A_KC, B_KC, KCH>(p, (id / ntn) * BM, (id % ntn) * BN, split, smem, u);
}

// Persistent, tail-balanced launch for outputs of more than one wave of co-resident workgroups.  The grid is exactly the
// 512 co-resident workgroups (2 per CU); workgroup w walks a STATIC list of work units, so no CU can end up with an
// extra tile (measured with per-workgroup stamps: the hardware dispatcher hands freed slots out greedily and 1-3 CUs
// regularly received 6 of the 1280 tiles of the 10240 x 2048 projections instead of 5: 383 us instead of 330).  When the
// unit count leaves at most half a wave of workgroups over, those units are cut into two 128x64 halves, one per
// workgroup: the tail costs half a tile time (2.5 tile times per workgroup for 1280 tiles, not 3).  Per output element
// the arithmetic is the plain kernel's (bit-identical result).
template <bool A_KC, bool B_KC, int KCH>
__global__ __launch_bounds__(256) void gemm_tile_persistent_kernel(TileP p, int nfull, int nhalf, int stagger) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int ntn = p.N / 128, tiles = ntn * (p.M / 128), G = (int)gridDim.x;
  const int w = xcd_order(blockIdx.x, G);                  // each XCD walks a contiguous run of unit ids per round
  // The two workgroups of a CU (blocks b and b + G/2) start together and would reach their store epilogues together,
  // leaving the matrix pipe idle; the second one therefore runs its half tile FIRST, which keeps the pair out of phase.
  const bool halves_first = (int)blockIdx.x >= G / 2 && stagger;
  if (halves_first)
    for (int q = w; q < nhalf; q += G) {
      const int u = nfull + (q >> 1), id = u % tiles;
      gemm_tile_body<128, 64, 2, 2, A_KC, B_KC, KCH>(p, (id / ntn) * 128, (id % ntn) * 128 + (q & 1) * 64, u / tiles, smem, nfull + q);
    }
  for (int u = w; u < nfull; u += G) {
    const int id = u % tiles;
    gemm_tile_body<128, 128, 2, 2, A_KC, B_KC, KCH>(p, (id / ntn) * 128, (id % ntn) * 128, u / tiles, smem, u);
  }
  if (!halves_first)
    for (int q = w; q < nhalf; q += G) {
      const int u = nfull + (q >> 1), id = u % tiles;
      gemm_tile_body<128, 64, 2, 2, A_KC, B_KC, KCH>(p, (id / ntn) * 128, (id % ntn) * 128 + (q & 1) * 64, u / tiles, smem, nfull + q);
    }
}

// ---- 320 x 256 tiles, operands by LDS-DMA (r06) --------------------------------------------------------------------------------------
// For the products whose output is exactly a whole number of rounds of such tiles on the 256 CUs -- the BiLSTM(512) input projections
// of IMU_Net, both directions batched: 10 240 x 2048 x {512, 1024} x 2 = 32 x 8 x 2 = 512 tiles = TWO rounds (the 128 x 128 walk above
// needs five rounds of 512 workgroups, and what it loses against the matrix pipe is not its main loop -- 0.94-0.96 of MFMA issue -- but
// the prologue / epilogue / unequal finish of every tile: ~120 us per launch whatever K).  One 512-thread workgroup per CU, waves 2 (M) x
// 4 (N), wave tile 160 x 64 = 5 x 2 tiles of v_mfma_f32_32x32x2_f32 (160 accumulator registers: a wave has 256).
//   Operands (NT form only: A[m][k], W[n][k], both k-contiguous): a 32-k chunk of the tile is (320 + 256) rows x 128 B = 72 KB; it goes
//   global -> LDS by LDS-DMA (global_load_lds_dwordx4, 72 1-KB transfers per chunk, 9 per wave, no staging registers) into a ring of TWO
//   stages (144 KB).  LDS rows are the UNPADDED 128-byte rows of the chunk; 16-byte piece q of row r sits at piece q ^ ((r >> 1) & 7), so
//   that the 16 lanes of every ds_read_b128 phase of the MFMA's operand fetch (lane l: row l % 32, k pieces 2 kb + l / 32) hit 16 distinct
//   bank quads -- the swizzle of lstm_step.hip, applied by the transfer's per-lane SOURCE address (a transfer writes lane l's 16 bytes
//   to LDS base + 16 l: rows r0 .. r0 + 7, piece position l % 8).  Same k permutation as the kernel above (lane half h supplies
//   k = 8 kb + 4 h + s at MFMA step s, both operands alike).
//   Schedule: ONE barrier per 32-k chunk (20 480 cycles of MFMA issue per SIMD), placed behind the chunk's last fragment read; the wave
//   then requests chunk kt + 2 into the stage it has just finished with and goes on with the last k block's MFMAs.
//   Epilogue: bias added in the accumulators, row-major stores of 128 contiguous bytes per half wave.
struct TileBigArgs {
  int tiles_m, tiles_n;       // 320-row panels, 256-column tiles per batch entry
  // (grid.y = p.nsplit K slabs: slab y covers k in [y * p.kchunk, min(K, (y + 1) * p.kchunk)) and leaves its partial product at
  //  p.ws + y * M * N, row stride N -- the deferred split-K format of mmego_gemm (accumulate = 2): products with fewer than 256 tiles,
  //  e.g. the input-gradient products of stage-1 training, 10 240 x {512, 1024} outputs over K = 4096)
};

template <int DUMMY>
__global__ __launch_bounds__(512, 1) void gemm_tile_big_kernel(TileP p, TileBigArgs g) {
  constexpr int BM = 320, BN = 256, KCH = 32, ROWS = BM + BN, STAGE = ROWS * KCH;   // floats per stage (72 8-row transfers)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = w >> 2, wn = w & 3;
  int tm, tn, batch;
  {
    const int per_batch = g.tiles_m * g.tiles_n, n = (int)gridDim.x;
    if ((n & 63) == 0 && (g.tiles_m & 3) == 0 && (g.tiles_n & 7) == 0) {
      // XCD x (blocks x, x + 8, ...) walks blocks of 4 row panels x 8 column tiles: its A panels stay in its L2 for a round
      const int x = blockIdx.x & 7, l = blockIdx.x >> 3, rounds = (n >> 3) >> 5;
      const int blk = x * rounds + (l >> 5), within = l & 31;
      const int nb_n = g.tiles_n >> 3, nb = (g.tiles_m >> 2) * nb_n;
      batch = blk / nb;
      const int rem = blk - batch * nb;
      tm = (rem / nb_n) * 4 + (within & 3);
      tn = (rem % nb_n) * 8 + (within >> 2);
    } else {
      batch = blockIdx.x / per_batch;
      const int id = blockIdx.x - batch * per_batch;
      tm = id / g.tiles_n;
      tn = id - tm * g.tiles_n;
    }
  }
  const int m0 = tm * BM, n0 = tn * BN;
  const int kbeg = p.nsplit > 1 ? (int)blockIdx.y * p.kchunk : 0;
  const int kend = p.nsplit > 1 ? min(p.K, kbeg + p.kchunk) : p.K;
  const float* Ab = p.A + (long)batch * p.sAb + (long)m0 * p.lda + kbeg;
  const float* Wb = p.W + (long)batch * p.sWb + (long)n0 * p.ldw + kbeg;
  // this wave's transfers: row groups i = w + 8 j (8 rows each) of a stage, j < 9; group i < 40: A rows 8 i .. 8 i + 7, else W rows.
  // lane l: row 8 i + l / 8, LDS piece position l % 8 <- source piece (l % 8) ^ ((row >> 1) & 7)
  const float* gp[9];
#pragma unroll
  for (int j = 0; j < 9; ++j) {
    const int i = w + 8 * j;
    const bool isw = i >= BM / 8;
    const int row = (isw ? i - BM / 8 : i) * 8 + (lane >> 3);
    const int piece = (lane & 7) ^ ((row >> 1) & 7);
    gp[j] = (isw ? Wb + (long)row * p.ldw : Ab + (long)row * p.lda) + 4 * piece;
  }
#ifndef GTB_EXP
#define GTB_EXP 0            // probe builds (scripts/bench_gemm_big.py): 1 = no transfers behind the first two chunks, 2 = no LDS reads, 4 = no C stores
#endif
#define GTB_DMA(kt, stage)                                                                                                  \
  if (!((GTB_EXP & 1) && (kt) > 1)) {                                                                                       \
    _Pragma("unroll") for (int j = 0; j < 9; ++j)                                                                           \
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gp[j] + (long)(kt) * KCH),            \
                                       (__attribute__((address_space(3))) void*)(smem + (stage) * STAGE + (w + 8 * j) * 8 * KCH), 16, 0, 0); \
  }
  const int nk = (kend - kbeg) / KCH;
  const int r = lane & 31, h = lane >> 5, key = (r >> 1) & 7;
  // fragment addresses: row (base + r) of the stage, piece (2 kb + h) ^ key -- the row bases are multiples of 32, so the key is the lane's
  int off[4];
#pragma unroll
  for (int kb = 0; kb < 4; ++kb) off[kb] = r * KCH + 4 * ((2 * kb + h) ^ key);
  f32x16 acc[5][2];
#pragma unroll
  for (int i = 0; i < 5; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = (f32x16){0};
  GTB_DMA(0, 0)
  if (nk > 1) GTB_DMA(1, 1)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
#ifndef GTB_SCHED
#define GTB_SCHED 1
#endif
#if GTB_SCHED == 1
  // r06, second schedule: the MFMAs of a k block go row block by row block (per accumulator the same k order: bit-identical), so row
  // block i's fragment registers are free after its 8 MFMAs and receive the next fragments two row blocks later -- every fragment is
  // requested 24 MFMAs before its use with no second register set (only the two W fragments are double-buffered).  An empty asm that reads
  // the fragments about to be used pins the compiler's wait (lgkmcnt(0) in this loop, whatever it waits for) in front of the new request.
  // The chunk's barrier sits in front of row block 2 of the last k block: every read of the stage is at least 8 MFMAs old there.
#define GTB_USE(x) asm volatile("" ::"v"(x))
#define GTB_SB() __builtin_amdgcn_sched_barrier(0)
  f32x4 a[5], b[2][2];
  {
    const float* As = smem + wm * 160 * KCH;
    const float* Bs = smem + (BM + wn * 64) * KCH;
#pragma unroll
    for (int i = 0; i < 3; ++i) a[i] = *reinterpret_cast<const f32x4*>(As + i * 32 * KCH + off[0]);
#pragma unroll
    for (int j = 0; j < 2; ++j) b[0][j] = *reinterpret_cast<const f32x4*>(Bs + j * 32 * KCH + off[0]);
  }
  for (int kt = 0; kt < nk; ++kt) {
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      // slot i (in front of row block i's MFMAs) requests: i = 0, 1: row blocks 3, 4 of THIS k block (their registers were read last by
      // the previous k block's MFMAs); i = 2, 3, 4: row blocks 0, 1, 2 (and, at i = 2, the W fragments) of the NEXT k block -- from this
      // stage, or (last k block of the chunk) from the other one, behind the barrier at slot 2.  Every request is 24 MFMAs ahead of its use
      // and 8 MFMAs behind the previous one, so the wait in front of it finds nothing young outstanding.
      const float* Ac = smem + (kt & 1) * STAGE + wm * 160 * KCH + off[kb];
      const int nst = kb < 3 ? (kt & 1) : ((kt + 1) & 1);
      const float* An = smem + nst * STAGE + wm * 160 * KCH + off[(kb + 1) & 3];
      const float* Bn = smem + nst * STAGE + (BM + wn * 64) * KCH + off[(kb + 1) & 3];
      const bool more = kb < 3 || kt + 1 < nk;
#pragma unroll
      for (int i = 0; i < 5; ++i) {
        GTB_USE(a[i]);
        if (i == 0) { GTB_USE(b[kb & 1][0]); GTB_USE(b[kb & 1][1]); }
        GTB_SB();
        if (i == 2 && kb == 3) {
          asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_s_barrier();
          if (kt + 2 < nk) GTB_DMA(kt + 2, kt & 1)
        }
        if (i < 2) a[i + 3] = *reinterpret_cast<const f32x4*>(Ac + (i + 3) * 32 * KCH);
        else if (more) a[i - 2] = *reinterpret_cast<const f32x4*>(An + (i - 2) * 32 * KCH);
        if (i == 2 && more) {
#pragma unroll
          for (int j = 0; j < 2; ++j) b[(kb + 1) & 1][j] = *reinterpret_cast<const f32x4*>(Bn + j * 32 * KCH);
        }
        GTB_SB();
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][c], b[kb & 1][j][c], acc[i][j], 0, 0, 0);
        GTB_SB();
      }
    }
  }
#undef GTB_USE
#undef GTB_SB
#else
  f32x4 a[5], b[2];
  for (int kt = 0; kt < nk; ++kt) {
    const float* As = smem + (kt & 1) * STAGE + wm * 160 * KCH;
    const float* Bs = smem + (kt & 1) * STAGE + (BM + wn * 64) * KCH;
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      if (!(GTB_EXP & 2) || (kt == 0 && kb == 0)) {
#pragma unroll
        for (int i = 0; i < 5; ++i) a[i] = *reinterpret_cast<const f32x4*>(As + i * 32 * KCH + off[kb]);
#pragma unroll
        for (int j = 0; j < 2; ++j) b[j] = *reinterpret_cast<const f32x4*>(Bs + j * 32 * KCH + off[kb]);
      }
      if (kb == 3) {
        // the stage's last fragments are requested: wait for them and for this wave's transfers of chunk kt + 1, meet the others, and
        // hand the stage back (chunk kt + 2)
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (kt + 2 < nk) GTB_DMA(kt + 2, kt & 1)
      }
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int i = 0; i < 5; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][c], b[j][c], acc[i][j], 0, 0, 0);
    }
  }
#endif
#undef GTB_DMA
  const bool slabs = p.nsplit > 1;
  float* C = slabs ? p.ws + (long)blockIdx.y * p.M * p.N : p.C + (long)batch * p.sCb;
  const long ldc = slabs ? (long)p.N : p.ldc;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = n0 + wn * 64 + j * 32 + r;
    const float bv = (p.bias && !slabs) ? p.bias[(long)batch * p.sBiasb + col] : 0.0f;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      float* cp = C + (long)(m0 + wm * 160 + i * 32 + 4 * h) * ldc + col;
#pragma unroll
      for (int reg = 0; reg < 16; ++reg)
        if (!(GTB_EXP & 4) || acc[i][j][reg] == 12345.f) cp[(long)((reg & 3) + 8 * (reg >> 2)) * ldc] = acc[i][j][reg] + bv;
    }
  }
}

namespace mmego_detail {

// k per staged chunk: 32 (measured at least as fast as 64 for the persistent kernel -- 207.8 / 348.9 us against 210.5 / 353.1 us on the
// 10240 x 2048 x {512, 1024} projections -- and it halves the LDS footprint: 36.9 KB per 128x128 workgroup, so a 72-KB recurrent-step
// workgroup of another branch fits beside the two GEMM workgroups of a CU).  The 64-k instantiations went with their knob in r05.
constexpr int GT_KCH = 32;

template <int BM, int BN, bool A_KC, bool B_KC, int KCH>
static int launch_plain_k(hipStream_t st, const TileP& p) {
  static bool attr_set = false;
  const size_t lds = (size_t)((BM + BN) * (KCH + 4)) * sizeof(float);
  if (!attr_set && lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_tile_kernel<BM, BN, 2, 2, A_KC, B_KC, KCH>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  const unsigned units = (unsigned)((p.M / BM) * (p.N / BN) * p.nsplit * p.nbatch);
  hipLaunchKernelGGL((gemm_tile_kernel<BM, BN, 2, 2, A_KC, B_KC, KCH>), dim3(units), dim3(256), lds, st, p);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}

template <int BM, int BN, bool A_KC, bool B_KC>
static int launch_plain(hipStream_t st, const TileP& p) { return launch_plain_k<BM, BN, A_KC, B_KC, GT_KCH>(st, p); }

template <bool A_KC, bool B_KC>
static int launch_layout(hipStream_t st, const TileP& p) {
  // 128x128 whenever it yields enough work units to occupy the 256 CUs (measured: 128x128 runs the 10240 x 2048
  // projections at 93-109 TFLOP/s, 64x64 at 88-97), else 64x64 (e.g. M = 512: 23.6 us vs 84 us with 64 big tiles).
  // A 160x128 tile (1024 tiles = exactly 2 waves of 512 for those projections) was measured 3-4 % SLOWER (1x4 wave
  // layout: 6 operand reads per 5 MFMAs), so it is not in the list.
  if (A_KC && B_KC) {
    // 320 x 256 tiles by LDS-DMA where the output is whole rounds of them (MMEGO_GEMM_BIG=0: the 128 x 128 walk, for A/B runs)
    static const bool big = !(getenv("MMEGO_GEMM_BIG") && atoi(getenv("MMEGO_GEMM_BIG")) == 0);
    const long tiles = (long)(p.M / 320) * (p.N / 256) * p.nbatch;
    // (K slabs: only the deferred form -- accumulate = 2, one batch entry, the caller sums the slabs -- and only where tiles x slabs is
    //  whole rounds of the chip)
    const bool slab_ok = p.nsplit == 1 ? !p.accumulate
                                       : (p.accumulate == 2 && p.nbatch == 1 && p.ws && !p.bias && (p.kchunk % 32) == 0 && p.kchunk >= 64 &&
                                          (long)(p.nsplit - 1) * p.kchunk < p.K && (p.K - (long)(p.nsplit - 1) * p.kchunk) >= 64);
    const long units = tiles * p.nsplit;
    if (big && (p.M % 320) == 0 && (p.N % 256) == 0 && (p.K % 32) == 0 && p.K >= 64 && slab_ok && !p.relu &&
        units >= 256 && (units % 256) == 0 && tiles < (1L << 30) && (p.lda % 4) == 0 && (p.ldw % 4) == 0 && (p.sAb % 4) == 0 && (p.sWb % 4) == 0 &&
        ((((uintptr_t)p.A) | ((uintptr_t)p.W)) & 15) == 0) {
      int dev = 0;
      if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return (int)hipErrorInvalidDevice;
      constexpr int lds = 2 * (320 + 256) * 32 * (int)sizeof(float);        // 144 KB
      static bool attr_set[64] = {};
      if (!attr_set[dev]) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_tile_big_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return (int)e;
        attr_set[dev] = true;
      }
      TileBigArgs g = {p.M / 320, p.N / 256};
      hipLaunchKernelGGL((gemm_tile_big_kernel<0>), dim3((unsigned)tiles, (unsigned)p.nsplit), dim3(512), lds, st, p, g);
      hipError_t e = hipGetLastError();
      return e == hipSuccess ? 0 : (int)e;
    }
  }
  const long units128 = (long)(p.M / 128) * (p.N / 128) * p.nsplit * p.nbatch;
  const bool big_ok = (p.M % 128) == 0 && (p.N % 128) == 0 && units128 >= 192;
  if (big_ok) {
    // 2 workgroups per CU x 256 CUs.  MMEGO_GEMM_SLOTS=256 (A/B knob of scripts/bench_overlap.py): ONE workgroup per CU, its LDS
    // request padded past half a CU's LDS so that the dispatcher cannot put two on one CU -- at 204 VGPRs a pair of these
    // workgroups leaves no registers for another kernel's waves, a single one leaves 300 per lane.
    static const int slots = getenv("MMEGO_GEMM_SLOTS") ? atoi(getenv("MMEGO_GEMM_SLOTS")) : 512;
    const int units = (int)units128;
    // (K <= 128: two chunks per tile -- the tile is its 64-KB store; the static walk's two tiles per workgroup then run one after
    // the other, load latency and store each exposed: 59 us for the 32768 x 512 x 64 BiLSTM(64) projections of config 5.  The plain
    // grid keeps four workgroups per CU in flight.)
    if (units > slots && p.K > 128) {
      const int rest = units % slots;
      const bool halves = rest > 0 && rest <= slots / 2;
      const int nfull = halves ? units - rest : units, nhalf = halves ? 2 * rest : 0;
      size_t lds = (size_t)(256 * (GT_KCH + 4)) * sizeof(float);
      if (slots <= 256) {
        static bool attr32 = false;
        lds = 84 * 1024;
        if (!attr32) {
          hipError_t e = hipFuncSetAttribute((const void*)gemm_tile_persistent_kernel<A_KC, B_KC, GT_KCH>,
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
          if (e != hipSuccess) return (int)e;
          attr32 = true;
        }
      }
      // (second half of the grid starts with its half tiles: the walk's tail is staggered)
      hipLaunchKernelGGL((gemm_tile_persistent_kernel<A_KC, B_KC, GT_KCH>), dim3(slots), dim3(256), lds, st, p, nfull, nhalf, 1);
      hipError_t e = hipGetLastError();
      return e == hipSuccess ? 0 : (int)e;
    }
    return launch_plain<128, 128, A_KC, B_KC>(st, p);
  }
  return launch_plain<64, 64, A_KC, B_KC>(st, p);
}

// returns 0 on launch, -2 if the shape does not fit these kernels (caller falls back), >0 on a HIP error.
int gemm_tile_launch(hipStream_t st, const TileP& p, bool a_kc, bool b_kc) {
  if ((p.K % 64) != 0 || (p.M % 64) != 0 || (p.N % 64) != 0 || p.nsplit < 1 || p.nbatch < 1) return -2;
  if (p.nsplit > 1 && ((p.kchunk % 64) != 0 || p.ws == nullptr)) return -2;
  if (a_kc && b_kc) return launch_layout<true, true>(st, p);
  if (a_kc) return launch_layout<true, false>(st, p);
  if (b_kc) return launch_layout<false, true>(st, p);
  return launch_layout<false, false>(st, p);
}

}  // namespace mmego_detail
