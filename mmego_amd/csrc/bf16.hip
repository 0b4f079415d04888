// bf16-input / fp32-accumulate contractions for the frozen IMU_Net forward (BASELINE config 5: "bf16 forward / fp32
// accumulate").  Opt-in: the fp32 kernels (gemm_tile.hip, lstm_step.hip) stay the default and the parity path.
//
// What is bf16 and what is not: the two OPERANDS of every dense product (activations and weights of the BiLSTM input
// projections, h_{t-1} and W_hh of the recurrence -- Net/IMU_Net.py:58-62,77,82) are rounded to bf16 (round to nearest
// even); products are exact in fp32 and accumulate in fp32 on v_mfma_f32_32x32x16_bf16; gate pre-activations, the
// cell state, the cell non-linearities and every output are fp32.
//
//   cvt_bf16_kernel          fp32 -> bf16 (RNE), 2-D with row strides.
//   gemm_bf16_nt_kernel      C[M,N] = A[M,K] . W[N,K]^T + bias  (128x128 tile, 64-k chunks, 4 waves x (64x64)).
//   lstm_step_bf16_*kernel   one BiLSTM timestep: gates = xproj + h_{t-1} . W_hh^T, fused cell update; writes h_t as fp32
//                            (consumers: pooling, heads) and as bf16 (next step's / next layer's product operand).
//                            The recurrent operands live in a fragment-major layout (see below).
#include "common.h"

typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef unsigned short bf16_t;   // raw bf16 bits (the C ABI carries them as unsigned short)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// round to nearest even, NaN stays NaN: v_cvt_pk_bf16_f32 on gfx950 (the integer form -- add 0x7fff + lsb, shift, a branch for NaN --
// was five VALU instructions and a branch per value)
__device__ __forceinline__ unsigned int f2bf_bits(float x) { return (unsigned int)__builtin_bit_cast(unsigned short, (__bf16)x); }

// ---- fp32 -> bf16 ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cvt_bf16_kernel(const float* __restrict__ X, long ldx, long rows, long cols4,
                                                        bf16_t* __restrict__ Y, long ldy) {
  const long total = rows * cols4;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long r = i / cols4, c = (i - r * cols4) * 4;
    const f32x4 v = *reinterpret_cast<const f32x4*>(X + r * ldx + c);
    uint2 o;
    o.x = f2bf_bits(v[0]) | (f2bf_bits(v[1]) << 16);
    o.y = f2bf_bits(v[2]) | (f2bf_bits(v[3]) << 16);
    *reinterpret_cast<uint2*>(Y + r * ldy + c) = o;
  }
}

extern "C" int mmego_cvt_bf16(void* stream, const float* X, long ldx, long rows, long cols, unsigned short* Y, long ldy) {
  MMEGO_REQUIRE(rows >= 0 && cols >= 0 && cols % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0);
  MMEGO_REQUIRE((((uintptr_t)X) & 15) == 0 && (((uintptr_t)Y) & 7) == 0);
  if (rows == 0 || cols == 0) return MMEGO_OK;
  const long total = rows * (cols / 4);
  const int grid = (int)(total / 256 + 1 < 4096 ? total / 256 + 1 : 4096);
  cvt_bf16_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(X, ldx, rows, cols / 4, Y, ldy);
  MMEGO_LAUNCH_CHECK();
  return MMEGO_OK;
}

// rows (b*T + t) of X -> rows (t*Bp + b) of Y: the time-major operand of a layer's input projection, so that 32 consecutive
// rows of the product belong to ONE timestep (see the tile-major xproj layout below)
__global__ __launch_bounds__(256) void cvt_bf16_tm_kernel(const float* __restrict__ X, long ldx, int Bn, int T, long cols4,
                                                           bf16_t* __restrict__ Y, int Bp) {
  const long total = (long)Bn * T * cols4;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long r = i / cols4, c = (i - r * cols4) * 4;
    const long b = r / T, t = r - b * T;
    const f32x4 v = *reinterpret_cast<const f32x4*>(X + r * ldx + c);
    uint2 o;
    o.x = f2bf_bits(v[0]) | (f2bf_bits(v[1]) << 16);
    o.y = f2bf_bits(v[2]) | (f2bf_bits(v[3]) << 16);
    *reinterpret_cast<uint2*>(Y + (t * Bp + b) * (cols4 * 4) + c) = o;
  }
}

extern "C" int mmego_cvt_bf16_tm(void* stream, const float* X, long ldx, int Bn, int T, long cols, unsigned short* Y, int Bp) {
  MMEGO_REQUIRE(Bn > 0 && T > 0 && cols > 0 && cols % 4 == 0 && ldx % 4 == 0 && Bp >= Bn);
  MMEGO_REQUIRE((((uintptr_t)X) & 15) == 0 && (((uintptr_t)Y) & 7) == 0);
  const long total = (long)Bn * T * (cols / 4);
  const int grid = (int)(total / 256 + 1 < 4096 ? total / 256 + 1 : 4096);
  cvt_bf16_tm_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(X, ldx, Bn, T, cols / 4, Y, Bp);
  MMEGO_LAUNCH_CHECK();
  return MMEGO_OK;
}

// ---- shared tile machinery -----------------------------------------------------------------------------------------
// Operand tiles sit in LDS as [row][64 k + 8 pad] bf16 (144-B rows): a lane's MFMA operand is the 16 B at
// (row = lane%32, k = 8*(lane/32) .. +8) of a 16-k step, one ds_read_b128; 144-B rows put 16 consecutive rows' 16-B
// reads on 16 distinct bank quads (conflict-free).
#define BK 64
#define BLD 72

// XCD-aware tile order: blocks b and b+8 share an XCD; hand each XCD a contiguous run of tile ids.
__device__ __forceinline__ int bf_xcd_order(int id, int n) { return (n & 7) == 0 ? (id & 7) * (n >> 3) + (id >> 3) : id; }

__device__ __forceinline__ bf16x8 lds_frag(const bf16_t* tile, int row, int k) {
  return *reinterpret_cast<const bf16x8*>(tile + row * BLD + k);
}

struct GemmBfP {
  const bf16_t* A; long lda;
  const bf16_t* W; long ldw;
  float* C; long ldc;
  bf16_t* Cb; long ldcb;        // optional bf16 copy of the output
  float* Cf;                    // optional tile-major fp32 output (see "TILE-MAJOR xproj" below)
  const float* bias;
  int M, N, K, relu, tiles_m, tiles_n;
};

// 256 threads, tile 128 (M) x 128 (N); wave w owns the 64x64 block (w/2, w%2): 2x2 MFMA 32x32 tiles, 64 accumulators.
// One LDS buffer; the next chunk travels global -> registers while the current one is multiplied.
__global__ __launch_bounds__(256, 2) void gemm_bf16_nt_kernel(GemmBfP p) {
  __shared__ __attribute__((aligned(16))) bf16_t As[128 * BLD];
  __shared__ __attribute__((aligned(16))) bf16_t Bs[128 * BLD];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int wm = w >> 1, wn = w & 1;
  // tile order: runs of 16 row-panels sweep the N tiles, so a W tile is shared by 16 consecutive blocks and the 16 A
  // panels stay in L2 for the whole sweep.
  const int id = bf_xcd_order(blockIdx.x, (int)gridDim.x);
  const int GM = 16;
  const int per_group = GM * p.tiles_n;
  const int group = id / per_group, in_group = id - group * per_group;
  const int gm = min(GM, p.tiles_m - group * GM);
  const int tm = group * GM + in_group % gm, tn = in_group / gm;
  const int m0 = tm * 128, n0 = tn * 128;

  const int lrow = tid >> 3, lseg = (tid & 7) * 8;
  const bf16_t* ag[4];
  const bf16_t* wg[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    ag[i] = p.A + (long)min(m0 + lrow + 32 * i, p.M - 1) * p.lda + lseg;
    wg[i] = p.W + (long)min(n0 + lrow + 32 * i, p.N - 1) * p.ldw + lseg;
  }
  u32x4 ra[4], rb[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    ra[i] = *reinterpret_cast<const u32x4*>(ag[i]);
    rb[i] = *reinterpret_cast<const u32x4*>(wg[i]);
  }
  f32x16 acc[2][2];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[mi][ni][i] = 0.f;

  const int fr = lane & 31, fk = (lane >> 5) * 8;
  for (int k0 = 0; k0 < p.K; k0 += BK) {
    __syncthreads();   // the previous chunk's fragments have been read
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      *reinterpret_cast<u32x4*>(As + (lrow + 32 * i) * BLD + lseg) = ra[i];
      *reinterpret_cast<u32x4*>(Bs + (lrow + 32 * i) * BLD + lseg) = rb[i];
    }
    __syncthreads();
    if (k0 + BK < p.K) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        ra[i] = *reinterpret_cast<const u32x4*>(ag[i] + k0 + BK);
        rb[i] = *reinterpret_cast<const u32x4*>(wg[i] + k0 + BK);
      }
    }
#pragma unroll
    for (int kk = 0; kk < BK; kk += 16) {
      bf16x8 a[2], b[2];
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) a[mi] = lds_frag(As, wm * 64 + mi * 32 + fr, kk + fk);
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) b[ni] = lds_frag(Bs, wn * 64 + ni * 32 + fr, kk + fk);
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
    }
  }
  // C layout of the 32x32 MFMA: register i of lane l is (row = 8*(i/4) + 4*(l/32) + i%4, col = l%32)
#pragma unroll
  for (int ni = 0; ni < 2; ++ni) {
    const int col = n0 + wn * 64 + ni * 32 + fr;
    if (col >= p.N) continue;
    const float bv = p.bias ? p.bias[col] : 0.f;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int row = m0 + wm * 64 + mi * 32 + 8 * (i >> 2) + 4 * (lane >> 5) + (i & 3);
        if (row < p.M) {
          float v = acc[mi][ni][i] + bv;
          if (p.relu) v = fmaxf(v, 0.f);
          if (p.C) p.C[(long)row * p.ldc + col] = v;
          if (p.Cb) p.Cb[(long)row * p.ldcb + col] = (bf16_t)f2bf_bits(v);
        }
      }
    }
  }
  if (p.Cf) {   // an MFMA accumulator tile IS a tile of the tile-major layout: four coalesced 1-KB stores per 32x32 tile
    const int ntn = p.N >> 5;
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      const int c0 = n0 + wn * 64 + ni * 32;
      if (c0 >= p.N) continue;
      const float bv = p.bias ? p.bias[c0 + fr] : 0.f;
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
        const int rr0 = m0 + wm * 64 + mi * 32;
        if (rr0 >= p.M) continue;
        float* t = p.Cf + ((long)(rr0 >> 5) * ntn + (c0 >> 5)) * 1024 + lane * 4;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          f32x4 v;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            v[r] = acc[mi][ni][4 * q + r] + bv;
            if (p.relu) v[r] = fmaxf(v[r], 0.f);
          }
          *reinterpret_cast<f32x4*>(t + q * 256) = v;
        }
      }
    }
  }
}

extern "C" int mmego_gemm_bf16(void* stream, const unsigned short* A, long lda, const unsigned short* W, long ldw, float* C,
                               long ldc, unsigned short* Cb, long ldcb, float* Cf, const float* bias, int M, int N, int K,
                               int relu) {
  MMEGO_REQUIRE(M > 0 && N > 0 && K > 0 && K % BK == 0 && lda % 8 == 0 && ldw % 8 == 0);
  MMEGO_REQUIRE((((uintptr_t)A) & 15) == 0 && (((uintptr_t)W) & 15) == 0 && (C || Cb || Cf));
  MMEGO_REQUIRE(!Cf || (M % 32 == 0 && N % 32 == 0 && (((uintptr_t)Cf) & 15) == 0));
  GemmBfP p;
  p.A = A; p.lda = lda; p.W = W; p.ldw = ldw; p.C = C; p.ldc = ldc; p.Cb = Cb; p.ldcb = ldcb; p.Cf = Cf; p.bias = bias;
  p.M = M; p.N = N; p.K = K; p.relu = relu;
  p.tiles_m = cdiv(M, 128); p.tiles_n = cdiv(N, 128);
  const long tiles = (long)p.tiles_m * p.tiles_n;
  MMEGO_REQUIRE(tiles < (1L << 30));
  gemm_bf16_nt_kernel<<<(int)tiles, 256, 0, (hipStream_t)stream>>>(p);
  MMEGO_LAUNCH_CHECK();
  return MMEGO_OK;
}

// ---- one BiLSTM timestep ---------------------------------------------------------------------------------------------
// FRAGMENT-MAJOR operand layout.  The MFMA wants, per 16-k step s and 32-row block rb, 16 B per lane: lane l holds
// row 32*rb + l%32, k = 16*s + 8*(l/32) .. +8.  Row-major storage makes that 32 different cache lines per wave
// instruction; so both recurrent operands -- W_hh (frozen, laid out once on the host) and h_{t-1} (written by the previous
// step's epilogue) -- are stored as the lanes will read them:
//     frag(X, rb, s)[l][e] = X[32*rb + l%32][16*s + 8*(l/32) + e],   blocks ordered [rb][s], 1 KB each,
// i.e. element (r, k) sits at (((r/32)*(H/16) + k/16)*64 + ((k/8)&1)*32 + r%32)*8 + k%8.  For W_hh the "rows" are
// ordered [hidden block jb][gate n][32 hidden units] (row 128*jb + 32*n + jj = W_hh[n*H + 32*jb + jj]), so one WG's four
// gate tiles are adjacent.  A wave's fragment load is then one fully coalesced 1-KB read and needs no LDS transpose.
//
// TILE-MAJOR xproj.  The projection's result is consumed exactly once, by the cell update, which holds gate pre-activations
// in the MFMA accumulator layout.  So the projection GEMM stores every 32x32 accumulator tile as it sits in registers and
// the step kernel loads it back the same way (coalesced 16 B per lane both times, no transposition anywhere): element
// (m, n) of the [M, N] product lives at
//     ((m/32)*(N/32) + n/32)*1024 + ((m%32)/8)*256 + (n%32 + 32*(((m%32)/4)&1))*4 + m%4.
// Rows are TIME-MAJOR, m = t*Bp + b with Bp = Bn rounded up to 32, so a tile's 32 rows are 32 sequences at one timestep;
// columns are n = d*4H + gate*H + j.
struct StepBfP {
  const bf16_t* hprev[2];               // h_{t-1}, fragment-major, ceil(Bn/32) row blocks
  const bf16_t* whh[2];                 // W_hh, fragment-major as above
  const float* xpf; long mt0[2];        // x.W_ih^T + b_ih + b_hh, tile-major; mt0[d] = first row tile of direction d's timestep
  float* hout[2]; long hos;             // h_t fp32, row-major
  bf16_t* houtb[2]; long hbs;           // h_t bf16, row-major (next layer's projection operand); may be null
  bf16_t* hfrag[2];                     // h_t bf16, fragment-major (next step's operand)
  float* c[2];
  int Bn, H, first;
};

__device__ __forceinline__ float bf_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float bf_tanh(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * x)); }

__device__ __forceinline__ long frag_off(int r, int k, int H) {
  return ((((long)(r >> 5) * (H >> 4) + (k >> 4)) * 64) + ((k >> 3) & 1) * 32 + (r & 31)) * 8 + (k & 7);
}

// this lane's 16 B (registers 4q .. 4q+3) of the xproj tile (row block rb, direction d, gate n, hidden block jb)
__device__ __forceinline__ const f32x4* xp_tile(const StepBfP& p, int d, int rb, int n, int jb, int lane) {
  const int hb = p.H >> 5;
  return reinterpret_cast<const f32x4*>(p.xpf + ((p.mt0[d] + rb) * (long)(8 * hb) + (d * 4 + n) * hb + jb) * 1024) + lane;
}

// cell update of one (row, j) and the three stores of h_t
__device__ __forceinline__ void bf_cell(const StepBfP& p, int d, int row, int j, float pi, float pf, float pg, float po,
                                        float cprev) {
  const float gi = bf_sigmoid(pi), gf = bf_sigmoid(pf), gg = bf_tanh(pg), go = bf_sigmoid(po);
  const float cn = gf * cprev + gi * gg;
  const float hn = go * bf_tanh(cn);
  p.c[d][(long)row * p.H + j] = cn;
  p.hout[d][(long)row * p.hos + j] = hn;
  const bf16_t hb = (bf16_t)f2bf_bits(hn);
  if (p.houtb[d]) p.houtb[d][(long)row * p.hbs + j] = hb;
  p.hfrag[d][frag_off(row, j, p.H)] = hb;
}

// Large batches (Bn > 2048) or H % 256 != 0.  WG = (32*WR*(4/KS)) batch rows x 32 hidden units x 4 gates; 4 waves.
// Wave w: row group w % (4/KS), k-slice w / (4/KS) (KS = 2: the 16-k steps {0,1} or {2,3} of every 64-k chunk; the
// slices are summed through LDS at the end).  64-k chunks are staged in LDS, themselves fragment-major (a straight
// 4-KB copy per row block; lane-linear ds_read_b128, conflict-free without padding); the next chunk travels
// global -> registers while the current one is multiplied.
//   <WR=1, KS=1>: 128 rows / WG, 64 accumulators per lane -- three WGs per CU, so one WG's HBM-bound epilogue (xproj, c,
//                 h: ~1 GB per step at Bn = 32768) overlaps the others' products.  (256 rows / WG with 128 accumulators
//                 re-reads W_hh half as often but runs one WG per CU: measured 695 us per step against 374 us.)
//   <WR=1, KS=2>:  64 rows / WG -- small batches with H % 256 != 0.
// The accumulator tile n (of 4) is gate n for hidden units j0..j0+31, so one lane holds i, f, g, o of its (row, j).
template <int WR, int KS>
__global__ __launch_bounds__(256) void lstm_step_bf16_kernel(StepBfP p) {
  constexpr int RG = 4 / KS;            // row groups (waves along rows)
  constexpr int ROWS = 32 * WR * RG;    // batch rows per WG
  constexpr int RB = ROWS / 32;         // row blocks per WG = 16-B loads per thread and chunk for the h tile
  constexpr int TILE_B = (RB + 4) * 4096;
  constexpr int RED_B = KS == 2 ? RG * WR * 4 * 16 * 64 * 4 : 0;
  constexpr int SMEM_B = TILE_B > RED_B ? TILE_B : RED_B;
  __shared__ __attribute__((aligned(16))) unsigned char smem_raw[SMEM_B];
  u32x4* As = reinterpret_cast<u32x4*>(smem_raw);       // [RB][4 steps][64 lanes] 16-B fragments
  u32x4* Bs = As + RB * 256;                            // [4 gates][4 steps][64 lanes]
  float* red = reinterpret_cast<float*>(smem_raw);

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int wr = w % RG, wk = w / RG;
  const int d = blockIdx.z, H = p.H, S = H >> 4;
  // blocks sharing a W_hh slice (same j0, different rows) are consecutive ids on one XCD
  const int nrb = gridDim.y, nb = gridDim.x * nrb;
  const int id = bf_xcd_order(blockIdx.y * gridDim.x + blockIdx.x, nb);
  const int jb = id / nrb, j0 = jb * 32, r0 = (id % nrb) * ROWS;
  const int fr = lane & 31, fh = lane >> 5;

  f32x16 acc[WR][4];
#pragma unroll
  for (int mi = 0; mi < WR; ++mi)
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[mi][n][i] = 0.f;

  if (!p.first) {
    const int last_rb = (p.Bn - 1) >> 5;
    const u32x4* ag[RB];
    const u32x4* wg[4];
#pragma unroll
    for (int i = 0; i < RB; ++i)
      ag[i] = reinterpret_cast<const u32x4*>(p.hprev[d]) + (long)min((r0 >> 5) + i, last_rb) * S * 64 + tid;
#pragma unroll
    for (int i = 0; i < 4; ++i) wg[i] = reinterpret_cast<const u32x4*>(p.whh[d]) + (long)(jb * 4 + i) * S * 64 + tid;
    u32x4 ra[RB], rb[4];
#pragma unroll
    for (int i = 0; i < RB; ++i) ra[i] = ag[i][0];
#pragma unroll
    for (int i = 0; i < 4; ++i) rb[i] = wg[i][0];
    for (int s0 = 0; s0 < S; s0 += 4) {
      __syncthreads();
#pragma unroll
      for (int i = 0; i < RB; ++i) As[i * 256 + tid] = ra[i];
#pragma unroll
      for (int i = 0; i < 4; ++i) Bs[i * 256 + tid] = rb[i];
      __syncthreads();
      if (s0 + 4 < S) {
#pragma unroll
        for (int i = 0; i < RB; ++i) ra[i] = ag[i][(s0 + 4) * 64];
#pragma unroll
        for (int i = 0; i < 4; ++i) rb[i] = wg[i][(s0 + 4) * 64];
      }
#pragma unroll
      for (int s = 0; s < 4 / KS; ++s) {
        const int sl = wk * (4 / KS) + s;
        u32x4 a[WR], b[4];
#pragma unroll
        for (int mi = 0; mi < WR; ++mi) a[mi] = As[((wr * WR + mi) * 4 + sl) * 64 + lane];
#pragma unroll
        for (int n = 0; n < 4; ++n) b[n] = Bs[(n * 4 + sl) * 64 + lane];
#pragma unroll
        for (int mi = 0; mi < WR; ++mi)
#pragma unroll
          for (int n = 0; n < 4; ++n)
            acc[mi][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[mi]),
                                                                 __builtin_bit_cast(bf16x8, b[n]), acc[mi][n], 0, 0, 0);
      }
    }
    if (KS == 2) {
      __syncthreads();   // operand tiles are dead: their LDS becomes the reduction buffer
      if (wk == 1) {
#pragma unroll
        for (int mi = 0; mi < WR; ++mi)
#pragma unroll
          for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int i = 0; i < 16; ++i) red[(((wr * WR + mi) * 4 + n) * 16 + i) * 64 + lane] = acc[mi][n][i];
      }
      __syncthreads();
      if (wk == 0) {
#pragma unroll
        for (int mi = 0; mi < WR; ++mi)
#pragma unroll
          for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mi][n][i] += red[(((wr * WR + mi) * 4 + n) * 16 + i) * 64 + lane];
      }
    }
  }
  if (wk != 0) return;
  const int j = j0 + fr;
#pragma unroll
  for (int mi = 0; mi < WR; ++mi) {
    const int rb = (r0 >> 5) + wr * WR + mi;
    if (rb * 32 >= p.Bn) continue;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f32x4 x[4];
#pragma unroll
      for (int n = 0; n < 4; ++n) x[n] = xp_tile(p, d, rb, n, jb, lane)[q * 64];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = 4 * q + r;
        const int row = rb * 32 + 8 * q + 4 * fh + r;
        if (row < p.Bn) {
          const float cprev = p.first ? 0.f : p.c[d][(long)row * H + j];
          bf_cell(p, d, row, j, acc[mi][0][i] + x[0][r], acc[mi][1][i] + x[1][r], acc[mi][2][i] + x[2][r],
                  acc[mi][3][i] + x[3][r], cprev);
        }
      }
    }
  }
}

// Small batches (Bn <= 2048, H % 256 == 0; the config-3 shape Bn = 512, H = 512 gives 8 x 16 x 2 = 256 WGs, one per CU):
// the step is a chain of latencies (launch, operand fetch, product, cell), not a throughput problem, so the kernel is
// built to pay ONE memory latency.  No LDS staging: wave w takes k quarter w of the WG's whole 64 rows x 128 W rows, so no
// fragment is fetched twice, and requests all its fragments (GS 16-k steps x 6 coalesced 1-KB reads) before the first
// MFMA, together with the xproj / c values of the cells it will finish.  The four k-quarter partial tiles meet in LDS
// (each wave writes its 128 accumulators, then sums the 32 of the 16 rows x 32 units it owns, in fixed order), and all
// four waves run the cell update.
template <int GS>
__global__ __launch_bounds__(256) void lstm_step_bf16_direct_kernel(StepBfP p) {
  extern __shared__ __attribute__((aligned(16))) float red[];      // [4 waves][2 mi][4 n][16 i][64 lanes] = 128 KB
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int d = blockIdx.z, H = p.H, S = H >> 4, SQ = S >> 2;
  const int nrb = gridDim.y, nb = gridDim.x * nrb;
  const int id = bf_xcd_order(blockIdx.y * gridDim.x + blockIdx.x, nb);
  const int jb = id / nrb, j0 = jb * 32, r0 = (id % nrb) * 64;
  const int fr = lane & 31, fh = lane >> 5;
  const int j = j0 + fr;
  const int own_mi = w >> 1, own_i0 = 8 * (w & 1);     // this wave finishes rows 32*own_mi + 16*(w&1) .. +16:
  const int own_r0 = own_mi * 32 + 16 * (w & 1);       // accumulator registers own_i0 .. own_i0+8 of row block own_mi

  f32x4 xp[4][2];
  float cprev[8];
  {
    const int rb = min((r0 >> 5) + own_mi, (p.Bn - 1) >> 5);
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
      for (int qq = 0; qq < 2; ++qq) xp[n][qq] = xp_tile(p, d, rb, n, jb, lane)[(2 * (w & 1) + qq) * 64];
  }
#pragma unroll
  for (int ii = 0; ii < 8; ++ii) {
    const int row = min(r0 + own_r0 + 8 * (ii >> 2) + 4 * fh + (ii & 3), p.Bn - 1);
    cprev[ii] = p.first ? 0.f : p.c[d][(long)row * H + j];
  }
  float pre[4][8];
#pragma unroll
  for (int n = 0; n < 4; ++n)
#pragma unroll
    for (int ii = 0; ii < 8; ++ii) pre[n][ii] = 0.f;

  if (!p.first) {
    f32x16 acc[2][4];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[mi][n][i] = 0.f;
    const int last_rb = (p.Bn - 1) >> 5;
    const u32x4* ap[2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
      ap[mi] = reinterpret_cast<const u32x4*>(p.hprev[d]) + ((long)min((r0 >> 5) + mi, last_rb) * S + w * SQ) * 64 + lane;
    const u32x4* wp = reinterpret_cast<const u32x4*>(p.whh[d]) + ((long)jb * 4 * S + w * SQ) * 64 + lane;
    const int gstride = S * 64;             // between the gates' fragment runs
    for (int s0 = 0; s0 < SQ; s0 += GS) {
      u32x4 a[GS][2], b[GS][4];
#pragma unroll
      for (int s = 0; s < GS; ++s) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) a[s][mi] = ap[mi][(s0 + s) * 64];
#pragma unroll
        for (int n = 0; n < 4; ++n) b[s][n] = wp[n * gstride + (s0 + s) * 64];
      }
      __builtin_amdgcn_sched_barrier(0);   // every request is out before the first MFMA (the scheduler would interleave them)
#pragma unroll
      for (int s = 0; s < GS; ++s)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int n = 0; n < 4; ++n)
            acc[mi][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[s][mi]),
                                                                 __builtin_bit_cast(bf16x8, b[s][n]), acc[mi][n], 0, 0, 0);
    }
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int i = 0; i < 16; ++i) red[(((w * 2 + mi) * 4 + n) * 16 + i) * 64 + lane] = acc[mi][n][i];
    __syncthreads();
#pragma unroll
    for (int src = 0; src < 4; ++src)
#pragma unroll
      for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int ii = 0; ii < 8; ++ii) pre[n][ii] += red[(((src * 2 + own_mi) * 4 + n) * 16 + own_i0 + ii) * 64 + lane];
  }
#pragma unroll
  for (int ii = 0; ii < 8; ++ii) {
    const int row = r0 + own_r0 + 8 * (ii >> 2) + 4 * fh + (ii & 3);
    if (row < p.Bn)
      bf_cell(p, d, row, j, pre[0][ii] + xp[0][ii >> 2][ii & 3], pre[1][ii] + xp[1][ii >> 2][ii & 3],
              pre[2][ii] + xp[2][ii >> 2][ii & 3], pre[3][ii] + xp[3][ii >> 2][ii & 3], cprev[ii]);
  }
}

extern "C" int mmego_lstm_step_bf16(void* stream, int ndir, int Bn, int H, int first, const unsigned short* hprev0,
                                    const unsigned short* hprev1, const unsigned short* whh0, const unsigned short* whh1,
                                    const float* xpf, long mt0_0, long mt0_1, float* hout0, float* hout1,
                                    long hos, unsigned short* houtb0, unsigned short* houtb1, long hbs,
                                    unsigned short* hfrag0, unsigned short* hfrag1, float* c0, float* c1) {
  MMEGO_REQUIRE((ndir == 1 || ndir == 2) && Bn > 0 && H > 0 && H % 64 == 0);
  MMEGO_REQUIRE(first || (hprev0 && (ndir == 1 || hprev1) && (((uintptr_t)hprev0) & 15) == 0 &&
                          (ndir == 1 || (((uintptr_t)hprev1) & 15) == 0)));
  MMEGO_REQUIRE((((uintptr_t)whh0) & 15) == 0 && (ndir == 1 || (((uintptr_t)whh1) & 15) == 0));
  MMEGO_REQUIRE(hfrag0 && (ndir == 1 || hfrag1) && hfrag0 != hprev0 && (ndir == 1 || hfrag1 != hprev1));
  MMEGO_REQUIRE(xpf && (((uintptr_t)xpf) & 15) == 0 && mt0_0 >= 0 && mt0_1 >= 0);
  StepBfP p;
  p.hprev[0] = hprev0; p.hprev[1] = hprev1;
  p.whh[0] = whh0; p.whh[1] = whh1;
  p.xpf = xpf; p.mt0[0] = mt0_0; p.mt0[1] = mt0_1;
  p.hout[0] = hout0; p.hout[1] = hout1; p.hos = hos;
  p.houtb[0] = houtb0; p.houtb[1] = houtb1; p.hbs = hbs;
  p.hfrag[0] = hfrag0; p.hfrag[1] = hfrag1;
  p.c[0] = c0; p.c[1] = c1;
  p.Bn = Bn; p.H = H; p.first = first;
  hipStream_t st = (hipStream_t)stream;
  if (Bn <= 2048 && H % 256 == 0) {
    const int lds = 4 * 2 * 4 * 16 * 64 * (int)sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
      hipError_t e = hipFuncSetAttribute((const void*)lstm_step_bf16_direct_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
      if (e == hipSuccess) e = hipFuncSetAttribute((const void*)lstm_step_bf16_direct_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
      if (e != hipSuccess) return (int)e;
      attr_set = true;
    }
    dim3 grid(H / 32, cdiv(Bn, 64), ndir);
    if (H % 512 == 0) lstm_step_bf16_direct_kernel<8><<<grid, 256, lds, st>>>(p);
    else lstm_step_bf16_direct_kernel<4><<<grid, 256, lds, st>>>(p);
  } else if (Bn <= 2048) {
    dim3 grid(H / 32, cdiv(Bn, 64), ndir);
    lstm_step_bf16_kernel<1, 2><<<grid, 256, 0, st>>>(p);
  } else {
    dim3 grid(H / 32, cdiv(Bn, 128), ndir);
    lstm_step_bf16_kernel<1, 1><<<grid, 256, 0, st>>>(p);
  }
  MMEGO_LAUNCH_CHECK();
  return MMEGO_OK;
}

// ---- large batches: input projection folded into the recurrent step ------------------------------------------------------
// At Bn = 32768 the separate projection is an HBM round trip of 16 B per gate pre-activation (21.5 GB written by the GEMM and
// read back by the steps, per layer: 86 of the ~140 GB a config-5 forward moves).  Here the step computes
//     gates = [x_t | h_{t-1}] . [W_ih | W_hh]^T + (b_ih + b_hh)
// in one accumulation: the K loop walks up to three operand SEGMENTS (layer 0: x_t, h_{t-1}; upper layers: the previous
// layer's forward and backward h_t, then h_{t-1}), every segment fragment-major for both operands, so the loop body is the
// one of lstm_step_bf16_kernel<1,1> (128 rows x 32 units x 4 gates per WG, 64-k chunks through LDS, 3 WGs per CU).  No
// projection tensor exists; per step the kernel streams x_t, h_{t-1}, c (read + write) and h_t.
// frag-major time-major conversion of a layer-0 input: rows (b*T + t) of X -> fragment-major [T][Bp x C]
__global__ __launch_bounds__(256) void cvt_bf16_frag_tm_kernel(const float* __restrict__ X, long ldx, int Bn, int T, int C,
                                                                bf16_t* __restrict__ Y, int Bp) {
  const long c8 = C >> 3, total = (long)Bn * T * c8;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long r = i / c8;
    const int k = (int)(i - r * c8) * 8;
    const int b = (int)(r / T), t = (int)(r - (long)b * T);
    const f32x4 v0 = *reinterpret_cast<const f32x4*>(X + r * ldx + k);
    const f32x4 v1 = *reinterpret_cast<const f32x4*>(X + r * ldx + k + 4);
    u32x4 o;
    o[0] = f2bf_bits(v0[0]) | (f2bf_bits(v0[1]) << 16);
    o[1] = f2bf_bits(v0[2]) | (f2bf_bits(v0[3]) << 16);
    o[2] = f2bf_bits(v1[0]) | (f2bf_bits(v1[1]) << 16);
    o[3] = f2bf_bits(v1[2]) | (f2bf_bits(v1[3]) << 16);
    *reinterpret_cast<u32x4*>(Y + (long)t * Bp * C + frag_off(b, k, C)) = o;
  }
}

extern "C" int mmego_cvt_bf16_frag_tm(void* stream, const float* X, long ldx, int Bn, int T, int C, unsigned short* Y, int Bp) {
  MMEGO_REQUIRE(Bn > 0 && T > 0 && C > 0 && C % 16 == 0 && ldx % 4 == 0 && Bp >= Bn && Bp % 32 == 0);
  MMEGO_REQUIRE((((uintptr_t)X) & 15) == 0 && (((uintptr_t)Y) & 15) == 0);
  const long total = (long)Bn * T * (C / 8);
  const int grid = (int)(total / 256 + 1 < 8192 ? total / 256 + 1 : 8192);
  cvt_bf16_frag_tm_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(X, ldx, Bn, T, C, Y, Bp);
  MMEGO_LAUNCH_CHECK();
  return MMEGO_OK;
}

// IMU_Net's fc1 (Linear(15, H) + ReLU, Net/IMU_Net.py:53,73) straight into the fused step's layer-0 operand: Y[t] (fragment-major
// [Bp x H] bf16) = bf16(relu(X[b*T + t] . W^T + bias)).  In the bf16 mode at large batch the fp32 activation (1.3 GB at config 5)
// existed only to be converted: product 0.44 ms + conversion 0.89 ms there, this kernel writes the 0.67 GB of bf16 once.
// One workgroup per (timestep, 32-row block); the K <= 16 input values of a row wait in registers, W (padded to 16 columns) in
// LDS; a thread produces the 8 consecutive units of one lane of a 1-KB fragment piece, so a wave's store is one whole piece.
__global__ __launch_bounds__(256) void fc_relu_bf16_frag_tm_kernel(const float* __restrict__ X, long ldx, const float* __restrict__ W,
                                                                    const float* __restrict__ bias, int Bn, int T, int Cin, int H,
                                                                    bf16_t* __restrict__ Y, int Bp, int relu) {
  extern __shared__ __attribute__((aligned(16))) float wsm[];      // [H][16] + bias [H]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < H * 16; i += 256) {
    const int n = i >> 4, k = i & 15;
    wsm[i] = k < Cin ? W[(long)n * Cin + k] : 0.f;
  }
  for (int i = tid; i < H; i += 256) wsm[H * 16 + i] = bias ? bias[i] : 0.f;
  const int t = blockIdx.y, rb = blockIdx.x;
  const int b = rb * 32 + (lane & 31), half = lane >> 5;
  float x[16];
  {
    const float* xr = X + ((long)(b < Bn ? b : Bn - 1) * T + t) * ldx;
#pragma unroll
    for (int k = 0; k < 16; ++k) x[k] = xr[k < Cin ? k : Cin - 1];        // (clamped address; the padded weights are zero)
  }
  __syncthreads();
  const bool live = b < Bn;
  u32x4* dst = reinterpret_cast<u32x4*>(Y + (long)t * Bp * H) + (long)rb * (H >> 4) * 64 + lane;
  for (int k16 = wave; k16 < (H >> 4); k16 += 4) {
    const int n0 = k16 * 16 + half * 8;
    float y[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const f32x4* wr = reinterpret_cast<const f32x4*>(wsm + (n0 + j) * 16);
      float a = wsm[H * 16 + n0 + j];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 w4 = wr[q];
        a = fmaf(x[4 * q], w4[0], a); a = fmaf(x[4 * q + 1], w4[1], a); a = fmaf(x[4 * q + 2], w4[2], a); a = fmaf(x[4 * q + 3], w4[3], a);
      }
      y[j] = live ? (relu ? fmaxf(a, 0.f) : a) : 0.f;
    }
    u32x4 o;
    o[0] = f2bf_bits(y[0]) | (f2bf_bits(y[1]) << 16);
    o[1] = f2bf_bits(y[2]) | (f2bf_bits(y[3]) << 16);
    o[2] = f2bf_bits(y[4]) | (f2bf_bits(y[5]) << 16);
    o[3] = f2bf_bits(y[6]) | (f2bf_bits(y[7]) << 16);
    dst[(long)k16 * 64] = o;
  }
}

// The same product on the matrix pipe for H a multiple of 128 (IMU_Net: 512).  The VALU form above reads every weight from LDS
// once per 32 rows (a broadcast ds_read_b128 per four multiply-adds: 412 us at config 5, bound by the LDS reads); here one workgroup
// owns a 32-row block for ALL timesteps with its share of W^T as v_mfma_f32_32x32x2_f32 operand registers (fp32 operands: the
// result is the fp32 product rounded to bf16, as before), x_t's eight values per lane requested one timestep ahead, and the 32 x 32
// accumulator tile turned into fragment pieces (a lane's 8 consecutive units) through a wave-private LDS tile.
template <int NCTW>
__global__ __launch_bounds__(256) void fc_relu_bf16_frag_tm_mfma_kernel(const float* __restrict__ X, long ldx, const float* __restrict__ W,
                                                                         const float* __restrict__ bias, int Bn, int T, int Cin, int H,
                                                                         bf16_t* __restrict__ Y, int Bp, int relu) {
  __shared__ float tile[4][32 * 33];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int rb = blockIdx.x;
  float wf[NCTW][8], bs[NCTW];
#pragma unroll
  for (int j = 0; j < NCTW; ++j) {
    const int n = (wave * NCTW + j) * 32 + r;
    bs[j] = bias ? bias[n] : 0.f;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const int k = 2 * s + h;
      const float w = W[(long)n * Cin + (k < Cin ? k : Cin - 1)];
      wf[j][s] = k < Cin ? w : 0.f;
    }
  }
  const int b = rb * 32 + r;
  const float* xr = X + (long)(b < Bn ? b : Bn - 1) * T * ldx;
  float xn[8];
#pragma unroll
  for (int s = 0; s < 8; ++s) xn[s] = xr[2 * s + h < Cin ? 2 * s + h : Cin - 1];
  float* tl = tile[wave];
  const int prow = lane & 31, pc8 = lane >> 5;          // this lane's row and 8-column half inside a 16-column piece
  const bool live = rb * 32 + prow < Bn;
  for (int t = 0; t < T; ++t) {
    float xa[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) xa[s] = xn[s];
    {
      const float* xq = xr + (long)(t + 1 < T ? t + 1 : t) * ldx;
#pragma unroll
      for (int s = 0; s < 8; ++s) xn[s] = xq[2 * s + h < Cin ? 2 * s + h : Cin - 1];
    }
    u32x4* dst = reinterpret_cast<u32x4*>(Y + (long)t * Bp * H) + (long)rb * (H >> 4) * 64 + lane;
#pragma unroll
    for (int j = 0; j < NCTW; ++j) {
      f32x16 acc = {0};
#pragma unroll
      for (int s = 0; s < 8; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[s], wf[j][s], acc, 0, 0, 0);
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const float v = acc[e] + bs[j];
        tl[((e & 3) + 8 * (e >> 2) + 4 * h) * 33 + r] = relu ? fmaxf(v, 0.f) : v;
      }
      __builtin_amdgcn_wave_barrier();                  // (wave-private tile: LDS operations of a wave complete in issue order)
#pragma unroll
      for (int pc = 0; pc < 2; ++pc) {
        float y[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) y[q] = live ? tl[prow * 33 + pc * 16 + pc8 * 8 + q] : 0.f;
        u32x4 o;
        o[0] = f2bf_bits(y[0]) | (f2bf_bits(y[1]) << 16);
        o[1] = f2bf_bits(y[2]) | (f2bf_bits(y[3]) << 16);
        o[2] = f2bf_bits(y[4]) | (f2bf_bits(y[5]) << 16);
        o[3] = f2bf_bits(y[6]) | (f2bf_bits(y[7]) << 16);
        dst[(long)(((wave * NCTW + j) * 2 + pc)) * 64] = o;
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
}

extern "C" int mmego_fc_relu_bf16_frag_tm(void* stream, const float* X, long ldx, const float* W, const float* bias, int Bn, int T,
                                          int Cin, int H, unsigned short* Y, int Bp, int relu) {
  MMEGO_REQUIRE(X && W && Y && Bn > 0 && T > 0 && Cin > 0 && Cin <= 16 && H > 0 && H % 16 == 0 && H <= 2048 && Bp >= Bn && Bp % 32 == 0);
  MMEGO_REQUIRE(T <= 65535 && (((uintptr_t)Y) & 15) == 0);
  const size_t lds = (size_t)(H * 16 + H) * sizeof(float);
  static bool attr_set = false;
  if (!attr_set && lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)fc_relu_bf16_frag_tm_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 2048 * 17 * 4);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  if (H % 128 == 0 && H <= 512) {
    const dim3 g(Bp / 32);
    hipStream_t st = (hipStream_t)stream;
    switch (H / 128) {
      case 1: fc_relu_bf16_frag_tm_mfma_kernel<1><<<g, 256, 0, st>>>(X, ldx, W, bias, Bn, T, Cin, H, Y, Bp, relu); break;
      case 2: fc_relu_bf16_frag_tm_mfma_kernel<2><<<g, 256, 0, st>>>(X, ldx, W, bias, Bn, T, Cin, H, Y, Bp, relu); break;
      case 3: fc_relu_bf16_frag_tm_mfma_kernel<3><<<g, 256, 0, st>>>(X, ldx, W, bias, Bn, T, Cin, H, Y, Bp, relu); break;
      default: fc_relu_bf16_frag_tm_mfma_kernel<4><<<g, 256, 0, st>>>(X, ldx, W, bias, Bn, T, Cin, H, Y, Bp, relu); break;
    }
    MMEGO_LAUNCH_CHECK();
    return MMEGO_OK;
  }
  dim3 grid(Bp / 32, T);
  fc_relu_bf16_frag_tm_kernel<<<grid, 256, lds, (hipStream_t)stream>>>(X, ldx, W, bias, Bn, T, Cin, H, Y, Bp, relu);
  MMEGO_LAUNCH_CHECK();
  return MMEGO_OK;
}

struct FusedStepP {
  const bf16_t* a[3][2];     // operand segments [segment][direction], fragment-major [Bp x 16*S]; the last one is h_{t-1}
  const bf16_t* w[3][2];     // weight segments, fragment-major with rows [hidden block][gate][32 units]
  int S[3];                  // 16-k steps per segment (multiples of 4)
  int nseg;                  // segments in use, h_{t-1} last; at the first timestep the host passes nseg without it
  const float* bias;         // [2][4H] b_ih + b_hh
  StepBfP o;                 // outputs, c, Bn, H, first (hprev / whh / xpf unused)
};

// one operand segment through the chunk pipeline: WG = 128 WR rows, wave wr owns WR row blocks of 32 rows.  The activation
// fragments of a wave's OWN row block never touch LDS: they are fragment-major in memory (one coalesced 1-KB read per 16-k
// step), no other wave needs them, so each wave fetches them straight into the registers the MFMAs read -- only the weight
// tile, which all four waves share, is staged.  (Staging both, the LDS array's read + write cycles of a chunk, 576 per
// workgroup, exceeded its 512 MFMA cycles: MFMA busy 0.38.)
template <int WR>
__device__ __forceinline__ void fused_segment(const bf16_t* A, const bf16_t* W, int S, int jb, const int (&rbi)[4 * WR], int tid,
                                              int lane, int wr, u32x4* Bs, f32x16 (&acc)[WR][4]) {
  const u32x4* aw[WR];
#pragma unroll
  for (int mi = 0; mi < WR; ++mi) aw[mi] = reinterpret_cast<const u32x4*>(A) + (long)rbi[wr * WR + mi] * S * 64 + lane;
  const u32x4* wb = reinterpret_cast<const u32x4*>(W) + (long)jb * 4 * S * 64 + tid;
  u32x4 ra[WR][4], rw[4];
#pragma unroll
  for (int mi = 0; mi < WR; ++mi)
#pragma unroll
    for (int sl = 0; sl < 4; ++sl) ra[mi][sl] = aw[mi][sl * 64];
#pragma unroll
  for (int i = 0; i < 4; ++i) rw[i] = wb[(long)i * S * 64];
  for (int s0 = 0; s0 < S; s0 += 4) {
    __syncthreads();
    u32x4 ac[WR][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) Bs[i * 256 + tid] = rw[i];
#pragma unroll
    for (int mi = 0; mi < WR; ++mi)
#pragma unroll
      for (int sl = 0; sl < 4; ++sl) ac[mi][sl] = ra[mi][sl];
    __syncthreads();
    {
      // (unconditional, past the last chunk on a clamped index: loads under a condition have their values copied -- and waited
      // for -- where the condition ends, i.e. in front of this chunk's MFMAs)
      const int sn = s0 + 4 < S ? s0 + 4 : s0;
#pragma unroll
      for (int mi = 0; mi < WR; ++mi)
#pragma unroll
        for (int sl = 0; sl < 4; ++sl) ra[mi][sl] = aw[mi][(sn + sl) * 64];
#pragma unroll
      for (int i = 0; i < 4; ++i) rw[i] = wb[((long)i * S + sn) * 64];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int sl = 0; sl < 4; ++sl) {
      u32x4 b[4];
#pragma unroll
      for (int n = 0; n < 4; ++n) b[n] = Bs[(n * 4 + sl) * 64 + lane];
#pragma unroll
      for (int mi = 0; mi < WR; ++mi)
#pragma unroll
        for (int n = 0; n < 4; ++n)
          acc[mi][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ac[mi][sl]), __builtin_bit_cast(bf16x8, b[n]),
                                                               acc[mi][n], 0, 0, 0);
    }
  }
}

template <int WR>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4 / WR, 4 / WR))) void lstm_step_bf16_fused_kernel(FusedStepP p) {
  __shared__ __attribute__((aligned(16))) u32x4 Bs[4 * 256];       // [gate][4 steps][64 lanes]
  const int tid = threadIdx.x, lane = tid & 63, wr = tid >> 6;
  const int d = blockIdx.z, H = p.o.H;
  const int nrb = gridDim.y, nb = gridDim.x * nrb;
  const int id = bf_xcd_order(blockIdx.y * gridDim.x + blockIdx.x, nb);
  // L2 blocking: consecutive workgroups on an XCD walk 4 x 4 super-tiles of (row block, hidden block), hidden-block groups
  // fastest, so 16 workgroups share 4 activation tiles and 4 weight tiles (8 x 0.4 MB at K = 1536, inside the 4-MB L2) and an
  // activation tile group stays resident across a whole sweep of W.  With (hidden block slow, row block fast) every row tile
  // was re-fetched once per hidden block: 3.1 GB per launch measured against 0.7 GB algorithmic.
  int jb, rbw;
  {
    const int njb = gridDim.x, JG = (njb & 3) == 0 ? 4 : 1, RG = 4;
    const int full = (nrb / RG) * RG * njb;
    if (id < full) {
      const int sg = id / (RG * JG), wi = id - sg * (RG * JG), jgs = njb / JG;
      rbw = (sg / jgs) * RG + wi % RG;
      jb = (sg % jgs) * JG + wi / RG;
    } else {
      const int rem = nrb % RG, i2 = id - full;
      rbw = (nrb / RG) * RG + i2 % rem;
      jb = i2 / rem;
    }
  }
  const int j0 = jb * 32, r0 = rbw * (128 * WR);
  const int fr = lane & 31, fh = lane >> 5;
  const int last_rb = (p.o.Bn - 1) >> 5;
  int rbi[4 * WR];
#pragma unroll
  for (int i = 0; i < 4 * WR; ++i) rbi[i] = min((r0 >> 5) + i, last_rb);

  f32x16 acc[WR][4];
#pragma unroll
  for (int mi = 0; mi < WR; ++mi)
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[mi][n][i] = 0.f;
#pragma unroll
  for (int seg = 0; seg < 3; ++seg)
    if (seg < p.nseg) fused_segment<WR>(p.a[seg][d], p.w[seg][d], p.S[seg], jb, rbi, tid, lane, wr, Bs, acc);

  const int j = j0 + fr;
  float bv[4];
#pragma unroll
  for (int n = 0; n < 4; ++n) bv[n] = p.bias[(d * 4 + n) * H + j];
  // cell update: c_{t-1} of all 16 elements in one round of loads (clamped rows), every result in a register of its own, then
  // the stores -- element by element (load, compute, three stores under the row predicate) each load waited for the stores in
  // front of it: 16 consecutive memory round trips per workgroup
#pragma unroll
  for (int mi = 0; mi < WR; ++mi) {
    const int rb = (r0 >> 5) + wr * WR + mi;
    if (rb * 32 >= p.o.Bn) continue;
    float cn[16], hn[16];
    if (!p.o.first) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int row = rb * 32 + 8 * (i >> 2) + 4 * fh + (i & 3);
        cn[i] = p.o.c[d][(long)(row < p.o.Bn ? row : p.o.Bn - 1) * H + j];
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("" : "+v"(cn[i]));
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const float cprev = p.o.first ? 0.f : cn[i];
      const float gi = bf_sigmoid(acc[mi][0][i] + bv[0]), gf = bf_sigmoid(acc[mi][1][i] + bv[1]);
      const float gg = bf_tanh(acc[mi][2][i] + bv[2]), go = bf_sigmoid(acc[mi][3][i] + bv[3]);
      cn[i] = gf * cprev + gi * gg;
      hn[i] = go * bf_tanh(cn[i]);
    }
    if (rb * 32 + 32 <= p.o.Bn) {                          // whole row block: no predicate per element
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int row = rb * 32 + 8 * (i >> 2) + 4 * fh + (i & 3);
        p.o.c[d][(long)row * H + j] = cn[i];
        if (p.o.hout[d]) p.o.hout[d][(long)row * p.o.hos + j] = hn[i];
        p.o.hfrag[d][frag_off(row, j, H)] = (bf16_t)f2bf_bits(hn[i]);
      }
    } else {
      unsigned short hb[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) hb[i] = f2bf_bits(hn[i]);
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int row = rb * 32 + 8 * (i >> 2) + 4 * fh + (i & 3);
        if (row < p.o.Bn) {
          p.o.c[d][(long)row * H + j] = cn[i];
          if (p.o.hout[d]) p.o.hout[d][(long)row * p.o.hos + j] = hn[i];
          p.o.hfrag[d][frag_off(row, j, H)] = (bf16_t)hb[i];
        }
      }
    }
  }
}

// ---- 256 x 256 tile form of the fused step (r03; both directions, Bn % 256 == 0, H % 64 == 0, every K segment % 64 == 0) ------
// A workgroup owns 256 rows x 64 hidden units x 4 gates (256 x 256 outputs: half the operand bytes per MFMA of the 128-row kernel
// above, through memory and through LDS), eight waves as 4 row groups x 2 unit groups, wave tile 64 rows x 32 units x 4 gates =
// 2 x 4 accumulator tiles (6 fragment reads per 8 MFMAs).  BOTH operands go global -> LDS by LDS-DMA (global_load_lds_dwordx4):
// they are fragment-major in memory, so a 1-KB piece (one MFMA operand of one 32-row block and one 16-k step) is one
// wave-instruction, lands in LDS as it lies and is read back lane-linear (conflict-free, no padding, no VGPR staging, no
// ds_write).  64-k chunks (64 pieces of 1 KB: 8 per wave) in a ring of two 64-KB stages; one barrier per chunk.
// Piece q of a stage: q < 32: activation rows (q >> 2) of the workgroup's 8 row blocks, step q & 3; q >= 32: weight rows
// [unit block (q - 32) >> 4][gate ((q - 32) >> 2) & 3], step q & 3.
// The kernel is PERSISTENT (one 160-KB workgroup per CU walks its share of the tiles, XCD-contiguous): the c tile of a workgroup
// travels by LDS-DMA into the ring stage that is free during the LAST chunk (64 KB: exactly one stage), the first operand chunk
// of the NEXT tile is requested before the cell update starts, and the update's stores drain under that tile's product loop
// (vmcnt counts in issue order: waiting for all but the youngest stores covers the requests issued before them); h_t in bf16
// leaves as whole fragment pieces (16-byte stores) instead of two-byte stores.
// Measured at Bn = 32768, H = 512 (scripts/bench_fused_step.py; the 128-row kernel in brackets): K = 1024 348-367 us (373-380),
// K = 1536 464-482 us (501-505).  Where the time goes (diagnostic builds, in git history): the product loop takes 106-118 us per
// 512 of K however the requests are issued (in one burst behind the barrier, one between every four MFMAs, 2 / 3 / 4 ring stages,
// four waves x two workgroups per CU), 76 us without them -- the LDS-DMA path delivers ~20 B/clk/CU here -- and the cell update
// costs 82 us per launch that do not hide under the product loops (60 us of stores at 5-6 TB/s, 18 us of cell arithmetic); PMC:
// 536 MB fetched per launch against 857 MB for the 128-row kernel, MFMA busy 0.33 at 2.23 GHz against 0.43 at 1.82 GHz -- the
// chip gives clock back as the loop gets denser, so the two kernels deliver nearly the same MFMA rate.
// r06 probes of this kernel (git d15b147, logs in profiles/r06_fused256_probes/; none kept): tile walks that leave an XCD 1 / 2 / 4 unit
// blocks so that its weights stay L2-resident -- 344-361 / 453-478 us against 347-349 / 456 (nothing); every tile reading the SAME
// activation rows (all L2 hits) -- 316 / 410 us (-10 %: what misses cost); a four-wave form with 128 x 128 wave tiles (8 instead of 12
// fragment reads per 16 MFMAs, one wave per SIMD, requests fired between the last MFMAs of a chunk; bit-identical) -- the SAME 100 us per
// 512 of K and 30 us more per launch in the cell update: the loop is not bound by LDS traffic either; every second workgroup started
// 7-27 us late so that store bursts meet product loops -- slower by about a third of the delay.  What the numbers say together: the loop
// runs at ~0.7 of the MFMA rate the clock under this load allows, and 130 us of a 335-us launch are the eight cell updates.
#define GLDS16B(gptr, lptr)                                                                                 \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr),                   \
                                   (__attribute__((address_space(3))) void*)(lptr), 16, 0, 0)

template <bool HOUT>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void lstm_step_bf16_fused256_kernel(FusedStepP p, int ntiles) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem256[];
  constexpr int STAGE = 64 * 1024;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);       // wave-uniform: the piece addresses below are scalar arithmetic
  const int rg = w >> 1, cg = w & 1;
  const int H = p.o.H;
  const int njb = H >> 6, nb = ntiles >> 1;               // tiles per direction
  const int G = (int)gridDim.x;
  const int S0 = p.S[0], S1 = p.nseg > 1 ? p.S[1] : 0, S2 = p.nseg > 2 ? p.S[2] : 0;
  const int NC = (S0 + S1 + S2) >> 2;
  const bool first = p.o.first != 0;
  // tile walk: where the grid is 8 x 32 workgroups and the tiles divide evenly, XCD x (workgroups x, x + 8, ...) takes a
  // contiguous run of tiles and its 32 workgroups work on 4 row blocks x all unit blocks at a time (they walk the same K chunks
  // at the same pace: an activation chunk is fetched into that L2 once for its 8 readers, a weight chunk once for 4)
  const bool xcd_walk = G == 256 && ntiles % 256 == 0;
  const int niter = (ntiles + G - 1) / G;
  auto tile_of = [&](int it) { return xcd_walk ? (int)(blockIdx.x & 7) * (ntiles >> 3) + it * 32 + (int)(blockIdx.x >> 3) : (int)blockIdx.x + it * G; };

  // this wave's 8 pieces of a chunk: waves 0..3 carry the activation pieces, waves 4..7 the weight pieces
  const int q0 = w * 8;
  auto issue = [&](int tile, int g, int stage) {
    const int d = tile >= nb, rem = tile - d * nb, jb = rem % njb, rbw = rem / njb;
    int s0 = g * 4, seg = 0, S = S0;
    if (s0 >= S0) { s0 -= S0; seg = 1; S = S1; if (s0 >= S1) { s0 -= S1; seg = 2; S = S2; } }
    const unsigned char* ab = reinterpret_cast<const unsigned char*>(p.a[seg][d]);
    const unsigned char* wb = reinterpret_cast<const unsigned char*>(p.w[seg][d]);
    unsigned char* st = smem256 + stage * STAGE;
    // piece q0 + i: consecutive pieces of a wave differ by one step (1 KB) inside a row block / gate, by S KB between them
    const long blk0 = q0 < 32 ? (long)(rbw * 8 + (q0 >> 2)) : (long)((jb * 2 + ((q0 - 32) >> 4)) * 4 + (((q0 - 32) >> 2) & 3));
    const unsigned char* g0 = (q0 < 32 ? ab : wb) + ((blk0 * S + s0) << 10) + lane * 16;
    const long blk_stride = (long)S << 10;
#pragma unroll
    for (int i = 0; i < 8; ++i) GLDS16B(g0 + (i >> 2) * blk_stride + ((i & 3) << 10), st + ((q0 + i) << 10));
  };
  // the c tile [256 rows][64 units] fp32 of a workgroup as 64 pieces of 4 rows (lane l: row 4 piece + l / 16, 16 B at column 4 (l % 16))
  auto issue_c = [&](int tile, int stage) {
    const int d = tile >= nb, rem = tile - d * nb, jb = rem % njb, rbw = rem / njb;
    const float* cb = p.o.c[d] + ((long)rbw * 256 + (lane >> 4)) * H + jb * 64 + (lane & 15) * 4;
    unsigned char* st = smem256 + stage * STAGE;
#pragma unroll
    for (int i = 0; i < 8; ++i) GLDS16B(cb + (long)(q0 + i) * 4 * H, st + ((q0 + i) << 10));
  };

  const int fr = lane & 31, fh = lane >> 5;
  // vector-memory operations a lane issues in the cell update BEHIND the next tile's first operand requests: per accumulator row
  // block (EPI_MI) 16 c stores (+ 16 fp32 h_t stores), then EPI_PIECES fragment stores.  The first chunk of the next tile waits
  // with vmcnt(EPI_WAIT): everything older than the youngest EPI_WAIT operations -- the requests included -- has completed.
  constexpr int EPI_MI = 2, EPI_PIECES = 4;
  constexpr int EPI_YOUNGER = EPI_MI * 16 * (HOUT ? 2 : 1) + EPI_PIECES;
  constexpr int EPI_WAIT = EPI_YOUNGER < 63 ? EPI_YOUNGER : 63;
  u32x4 fa[2][2], fb[2][4];
#define F256_RD(buf, st, step)                                                                                              \
  do {                                                                                                                      \
    _Pragma("unroll") for (int mi = 0; mi < 2; ++mi)                                                                        \
      fa[buf][mi] = *reinterpret_cast<const u32x4*>((st) + ((((2 * rg + mi) << 2) + (step)) << 10) + lane * 16);            \
    _Pragma("unroll") for (int n = 0; n < 4; ++n)                                                                           \
      fb[buf][n] = *reinterpret_cast<const u32x4*>((st) + ((32 + ((cg * 4 + n) << 2) + (step)) << 10) + lane * 16);         \
  } while (0)
#define F256_MM(buf)                                                                                                        \
  do {                                                                                                                      \
    _Pragma("unroll") for (int mi = 0; mi < 2; ++mi)                                                                        \
      _Pragma("unroll") for (int n = 0; n < 4; ++n)                                                                         \
        acc[mi][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[buf][mi]),                       \
                                                             __builtin_bit_cast(bf16x8, fb[buf][n]), acc[mi][n], 0, 0, 0);  \
  } while (0)
#define F256_BARRIER()                       \
  do {                                       \
    asm volatile("" ::: "memory");           \
    __builtin_amdgcn_s_barrier();            \
    asm volatile("" ::: "memory");           \
  } while (0)

  int stage = 0;                               // ring stage of the chunk about to be multiplied
  if (tile_of(0) < ntiles) issue(tile_of(0), 0, 0);
  for (int it = 0; it < niter; ++it) {
    const int tile = tile_of(it);
    if (tile >= ntiles) break;
    const int tile_next = it + 1 < niter ? tile_of(it + 1) : ntiles;
    f32x16 acc[2][4];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[mi][n][i] = 0.f;
    for (int c = 0; c < NC; ++c) {
      // this wave's pieces of chunk c have landed.  They were issued BEFORE the previous tile's cell-update stores (if any):
      // all but the youngest 63 operations done covers them, and waits for no more of those stores than it must
      // (younger than those pieces: the cell update's stores, counted by the constants its loops run on -- EPI_YOUNGER above; the
      // wait may leave at most that many operations outstanding, and the counter holds 63)
      if (c == 0 && it > 0) __builtin_amdgcn_s_waitcnt(0x0F70 | (EPI_WAIT & 15) | ((EPI_WAIT >> 4) << 14));
      else __builtin_amdgcn_s_waitcnt(0x0F70);                                        // vmcnt(0)
      F256_BARRIER();                          // everyone's pieces of chunk c are in LDS; everyone has finished the other stage
      if (c + 1 < NC) issue(tile, c + 1, stage ^ 1);
      else if (!first) issue_c(tile, stage ^ 1);
      const unsigned char* st = smem256 + stage * STAGE;
      F256_RD(0, st, 0);
      F256_RD(1, st, 1);
      __builtin_amdgcn_sched_barrier(0);
      F256_MM(0);
      __builtin_amdgcn_sched_barrier(0);
      F256_RD(0, st, 2);
      __builtin_amdgcn_sched_barrier(0);
      F256_MM(1);
      __builtin_amdgcn_sched_barrier(0);
      F256_RD(1, st, 3);
      __builtin_amdgcn_sched_barrier(0);
      F256_MM(0);
      __builtin_amdgcn_sched_barrier(0);
      F256_MM(1);
      __builtin_amdgcn_sched_barrier(0);
      stage ^= 1;
    }
    // `stage` now holds the c tile (its DMAs were issued a whole chunk ago); the other stage is free once everyone has passed
    // this barrier: the next tile's first chunk goes there now, and the ring continues from that stage
    __builtin_amdgcn_s_waitcnt(0x0F70);
    F256_BARRIER();
    const float* cs = reinterpret_cast<const float*>(smem256 + stage * STAGE);
    stage ^= 1;
    if (tile_next < ntiles) issue(tile_next, 0, stage);

    const int d = tile >= nb, rem = tile - d * nb, jb = rem % njb, rbw = rem / njb;
    const int j = (jb * 2 + cg) * 32 + fr;
    float bv[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) bv[n] = p.bias[(d * 4 + n) * H + j];
    // h_t in bf16 leaves fragment-major: this wave's 64 rows x 32 units are four whole 1-KB pieces [row block][16-k group].  They
    // are put together in 4 KB of LDS of the wave's own (behind the two ring stages: 160 KB in all, no other wave touches it,
    // no barrier) and leave as four 16-byte stores per lane -- element by element
    // they were 32 two-byte stores per lane scattered over eight 16-byte segments each (16 us of the launch)
    unsigned char* hst = smem256 + 2 * STAGE + w * 4096;
    float cv[EPI_MI][16];
    if (!first) {
#pragma unroll
      for (int mi = 0; mi < EPI_MI; ++mi)
#pragma unroll
        for (int i = 0; i < 16; ++i) cv[mi][i] = cs[(rg * 64 + mi * 32 + 4 * fh + 8 * (i >> 2) + (i & 3)) * 64 + cg * 32 + fr];
    }
#pragma unroll
    for (int mi = 0; mi < EPI_MI; ++mi) {
      float cn[16], hn[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const float cprev = first ? 0.f : cv[mi][i];
        const float gi = bf_sigmoid(acc[mi][0][i] + bv[0]), gf = bf_sigmoid(acc[mi][1][i] + bv[1]);
        const float gg = bf_tanh(acc[mi][2][i] + bv[2]), go = bf_sigmoid(acc[mi][3][i] + bv[3]);
        cn[i] = gf * cprev + gi * gg;
        hn[i] = go * bf_tanh(cn[i]);
      }
      // addresses = a wave-uniform row base (scalar arithmetic) + one per-lane offset that does not depend on i
      // addresses = a wave-uniform row base + one per-lane offset that does not depend on i
      float* const cwave = p.o.c[d] + ((long)rbw * 256 + rg * 64 + mi * 32) * H;
      float* const hwave = HOUT ? p.o.hout[d] + ((long)rbw * 256 + rg * 64 + mi * 32) * p.o.hos : nullptr;
      const unsigned cl = (unsigned)(4 * fh) * (unsigned)H + (unsigned)j, hl = (unsigned)(4 * fh) * (unsigned)p.o.hos + (unsigned)j;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        cwave[(long)(8 * (i >> 2) + (i & 3)) * H + cl] = cn[i];
        if (HOUT) hwave[(long)(8 * (i >> 2) + (i & 3)) * p.o.hos + hl] = hn[i];
        // piece (mi, fr >> 4) of the wave's four; inside it [k half][row % 32][8 k]
        *reinterpret_cast<bf16_t*>(hst + ((((mi * 2 + (fr >> 4)) * 64 + ((fr >> 3) & 1) * 32 + 4 * fh + 8 * (i >> 2) + (i & 3)) * 8 + (fr & 7)) << 1)) =
            __builtin_bit_cast(bf16_t, (__bf16)hn[i]);      // v_cvt_pk_bf16_f32: round to nearest even, NaN stays NaN (f2bf_bits' branch per element gone)
      }
    }
    {
      bf16_t* hf = p.o.hfrag[d];
#pragma unroll
      for (int pc = 0; pc < EPI_PIECES; ++pc) {           // piece pc = (row block pc >> 1, 16-k group pc & 1) of this wave
        const u32x4 v = *reinterpret_cast<const u32x4*>(hst + ((pc * 64 + lane) << 4));
        const long rb32 = (long)rbw * 8 + rg * 2 + (pc >> 1), k16 = (jb * 2 + cg) * 2 + (pc & 1);
        *reinterpret_cast<u32x4*>(hf + ((rb32 * (H >> 4) + k16) * 64 + lane) * 8) = v;
      }
    }
    // (the cell update read the c stage through ds_read: done before the next chunk-0 barrier releases anyone to overwrite it --
    //  the stage it sits in is refilled only after the NEXT tile's first barrier, which this wave reaches after these reads)
  }
#undef F256_RD
#undef F256_MM
#undef F256_BARRIER
}

extern "C" int mmego_lstm_step_bf16_fused(void* stream, int ndir, int Bn, int H, int first, int nseg,
                                          const unsigned short* a0_0, const unsigned short* a0_1, const unsigned short* w0_0,
                                          const unsigned short* w0_1, int K0, const unsigned short* a1_0,
                                          const unsigned short* a1_1, const unsigned short* w1_0, const unsigned short* w1_1,
                                          int K1, const unsigned short* hprev0, const unsigned short* hprev1,
                                          const unsigned short* whh0, const unsigned short* whh1, const float* bias,
                                          float* hout0, float* hout1, long hos, unsigned short* hfrag0, unsigned short* hfrag1,
                                          float* c0, float* c1) {
  MMEGO_REQUIRE((ndir == 1 || ndir == 2) && Bn > 0 && H > 0 && H % 64 == 0 && (nseg == 1 || nseg == 2));
  MMEGO_REQUIRE(K0 > 0 && K0 % 64 == 0 && a0_0 && w0_0 && (ndir == 1 || (a0_1 && w0_1)));
  MMEGO_REQUIRE(nseg == 1 || (K1 > 0 && K1 % 64 == 0 && a1_0 && w1_0 && (ndir == 1 || (a1_1 && w1_1))));
  MMEGO_REQUIRE(first || (hprev0 && whh0 && (ndir == 1 || (hprev1 && whh1))));
  MMEGO_REQUIRE(bias && hfrag0 && c0 && (ndir == 1 || (hfrag1 && c1)) && hfrag0 != hprev0);
  FusedStepP p;
  int s = 0;
  p.a[s][0] = a0_0; p.a[s][1] = a0_1; p.w[s][0] = w0_0; p.w[s][1] = w0_1; p.S[s] = K0 / 16; ++s;
  if (nseg == 2) { p.a[s][0] = a1_0; p.a[s][1] = a1_1; p.w[s][0] = w1_0; p.w[s][1] = w1_1; p.S[s] = K1 / 16; ++s; }
  if (!first) { p.a[s][0] = hprev0; p.a[s][1] = hprev1; p.w[s][0] = whh0; p.w[s][1] = whh1; p.S[s] = H / 16; ++s; }
  p.nseg = s;
  for (; s < 3; ++s) { p.a[s][0] = p.a[s][1] = p.w[s][0] = p.w[s][1] = nullptr; p.S[s] = 0; }
  for (int q = 0; q < p.nseg; ++q)
    for (int dd = 0; dd < ndir; ++dd)
      MMEGO_REQUIRE((((uintptr_t)p.a[q][dd]) & 15) == 0 && (((uintptr_t)p.w[q][dd]) & 15) == 0);
  p.bias = bias;
  p.o.hprev[0] = p.o.hprev[1] = nullptr; p.o.whh[0] = p.o.whh[1] = nullptr; p.o.xpf = nullptr; p.o.mt0[0] = p.o.mt0[1] = 0;
  p.o.hout[0] = hout0; p.o.hout[1] = hout1; p.o.hos = hos;
  p.o.houtb[0] = p.o.houtb[1] = nullptr; p.o.hbs = 0;
  p.o.hfrag[0] = hfrag0; p.o.hfrag[1] = hfrag1;
  p.o.c[0] = c0; p.o.c[1] = c1;
  p.o.Bn = Bn; p.o.H = H; p.o.first = first;
  // WR = 1 (128 rows per WG, 128 VGPRs, four WGs per CU); WR = 2 (256 rows, two WGs per CU, half the weight-tile reads per MFMA)
  // measured 20.3 - 20.5 ms per config-5 IMU_Net forward either way and went with its knob in r05.  Also measured without
  // effect on that figure: the weight tile double-buffered in LDS (one barrier per chunk), L2 super-tiles of 8 x 2, 16 x 1, 2 x 8
  // (hidden x row blocks) instead of 4 x 4.
  // 256 x 256 tiles (persistent, LDS-DMA operands, eight waves) where the shape allows, else the 128-row kernel
  bool k64 = true;
  for (int q = 0; q < p.nseg; ++q) k64 = k64 && p.S[q] % 4 == 0;
  // (also where the tiles do not fill the chip -- 2048 rows at H = 512: 128 tiles on 128 CUs: config 5 25.3 ms against 25.7 with the
  //  128-row kernel there)
  if (ndir == 2 && Bn % 256 == 0 && H % 64 == 0 && k64) {
    constexpr int lds = 2 * 64 * 1024 + 8 * 4096;      // two ring stages + 4 KB per wave for the h_t fragments
    static bool attr_set = false;
    if (!attr_set) {
      hipError_t e = hipFuncSetAttribute((const void*)lstm_step_bf16_fused256_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
      if (e == hipSuccess) e = hipFuncSetAttribute((const void*)lstm_step_bf16_fused256_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
      if (e != hipSuccess) return (int)e;
      attr_set = true;
    }
    MMEGO_REQUIRE((hout0 == nullptr) == (hout1 == nullptr));
    const int ntiles = 2 * (H / 64) * (Bn / 256);
    constexpr int maxwg = 256;
    dim3 grid(ntiles < maxwg ? ntiles : maxwg, 1, 1);
    if (hout0) lstm_step_bf16_fused256_kernel<true><<<grid, 512, lds, (hipStream_t)stream>>>(p, ntiles);
    else lstm_step_bf16_fused256_kernel<false><<<grid, 512, lds, (hipStream_t)stream>>>(p, ntiles);
  } else {
    dim3 grid(H / 32, cdiv(Bn, 128), ndir);
    lstm_step_bf16_fused_kernel<1><<<grid, 256, 0, (hipStream_t)stream>>>(p);
  }
  MMEGO_LAUNCH_CHECK();
  return MMEGO_OK;
}

// ---- attention pooling over the timesteps of a BiLSTM layer whose h_t live fragment-major in bf16 (r05) ---------------------------
// IMU_Net.py:77-81 (softmax over the 20 samples of a frame of w . h + b, weighted sum) on the fused step's own h_t fragments
// (hf [T][2 directions][Bp x H], fragment-major bf16): the layer's last step launches then write NO fp32 output (the HOUT form of the
// 256 x 256-tile step costs 73 us per launch more: 134 MB of fp32 stores per timestep at config 5) and the pooling reads 2 instead of
// 4 bytes per element.  One workgroup per 32-frame row block, eight waves = 2 directions x 4 column quarters, lane = (frame, 8-k
// half): per timestep a wave fetches its NPW pieces (one coalesced 1-KB read each), the partial scores meet in LDS (one barrier per
// timestep, double-buffered), and the weighted sum is kept ONLINE (running maximum and denominator: one pass over the data); the next
// timestep's pieces are in flight while this one is reduced.  vec [Bn][2H] fp32 row-major; attn [Bn][T] (optional).
template <int NPW>
__global__ __launch_bounds__(512) void attn_pool_frag_bf16_kernel(const bf16_t* __restrict__ hf, int T, int Bp, int Bn,
                                                                  const float* __restrict__ w, const float* __restrict__ bias,
                                                                  float* __restrict__ vec, float* __restrict__ attn) {
  constexpr int H = NPW * 64;
  __shared__ float red[2][8][32];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, d = wv >> 2, q = wv & 3;
  const int rb = blockIdx.x, fr = lane & 31, kh = lane >> 5;
  const int frame = rb * 32 + fr;
  const int col0 = d * H + q * (H / 4) + 8 * kh;               // this lane's columns: col0 + 16 p + e
  float wl[NPW][8];
#pragma unroll
  for (int p = 0; p < NPW; ++p)
#pragma unroll
    for (int e = 0; e < 8; ++e) wl[p][e] = w[col0 + 16 * p + e];
  const float b0 = bias[0];
  // piece (row block rb, 16-k group q NPW + p) of direction d at timestep s
  const long step_stride = 2L * Bp * H;
  const u32x4* base = reinterpret_cast<const u32x4*>(hf + (long)d * Bp * H) + ((long)rb * (H >> 4) + q * NPW) * 64 + lane;
  u32x4 cur[NPW], nxt[NPW];
#pragma unroll
  for (int p = 0; p < NPW; ++p) cur[p] = base[p * 64];
  float acc[NPW][8];
#pragma unroll
  for (int p = 0; p < NPW; ++p)
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[p][e] = 0.f;
  float m = -INFINITY, l = 0.f;
  for (int s = 0; s < T; ++s) {
    const int sn = min(s + 1, T - 1);                          // (clamped, unconditional: a predicated load drains the memory pipe)
    const u32x4* nb = reinterpret_cast<const u32x4*>(reinterpret_cast<const bf16_t*>(base) + sn * step_stride);
#pragma unroll
    for (int p = 0; p < NPW; ++p) nxt[p] = nb[p * 64];
    // (the 64 unpacked values are NOT kept for the weighted sum below: they are unpacked again from `cur` there -- 64 more shifts / ands
    //  per step against 64 registers; without packed-fp32 instructions (r06) the kernel had 17 spilled registers in this loop and went
    //  from 281 to 387 us)
    float ps = 0.f;
#pragma unroll
    for (int p = 0; p < NPW; ++p)
#pragma unroll
      for (int e2 = 0; e2 < 4; ++e2) {
        const unsigned u = cur[p][e2];
        ps += wl[p][2 * e2] * __uint_as_float(u << 16) + wl[p][2 * e2 + 1] * __uint_as_float(u & 0xffff0000u);
      }
    ps += __shfl_xor(ps, 32);
    if (lane < 32) red[s & 1][wv][fr] = ps;
    __syncthreads();
    float sc = b0;
#pragma unroll
    for (int k = 0; k < 8; ++k) sc += red[s & 1][k][fr];
    const float mn = fmaxf(m, sc);
    const float scale = __expf(m - mn), pe = __expf(sc - mn);
    l = l * scale + pe;
    m = mn;
#pragma unroll
    for (int p = 0; p < NPW; ++p)
#pragma unroll
      for (int e2 = 0; e2 < 4; ++e2) {
        const unsigned u = cur[p][e2];
        acc[p][2 * e2] = acc[p][2 * e2] * scale + pe * __uint_as_float(u << 16);
        acc[p][2 * e2 + 1] = acc[p][2 * e2 + 1] * scale + pe * __uint_as_float(u & 0xffff0000u);
      }
    if (attn && wv == 0 && lane < 32 && frame < Bn) attn[(long)frame * T + s] = sc;      // (raw scores; normalized below)
#pragma unroll
    for (int p = 0; p < NPW; ++p) cur[p] = nxt[p];
  }
  if (frame < Bn) {
    const float inv = 1.0f / l;
    float* o = vec + (long)frame * (2 * H) + col0;
#pragma unroll
    for (int p = 0; p < NPW; ++p) {
      *reinterpret_cast<f32x4*>(o + 16 * p) = f32x4{acc[p][0] * inv, acc[p][1] * inv, acc[p][2] * inv, acc[p][3] * inv};
      *reinterpret_cast<f32x4*>(o + 16 * p + 4) = f32x4{acc[p][4] * inv, acc[p][5] * inv, acc[p][6] * inv, acc[p][7] * inv};
    }
    if (attn && wv == 0 && lane < 32)
      for (int s = 0; s < T; ++s) attn[(long)frame * T + s] = __expf(attn[(long)frame * T + s] - m) * inv;
  }
}

extern "C" int mmego_attn_pool_frag_bf16_ok(int H) { return H == 128 || H == 256 || H == 512; }

// hf: [T][2][Bp x H] fragment-major bf16 (the hfrag outputs of mmego_lstm_step_bf16_fused for every timestep, direction 0 then 1);
// w [2H], b [1]: the attention Linear; vec [Bn][2H]; attn [Bn][T] or NULL.  Bp % 32 == 0, Bn <= Bp, H in {128, 256, 512}.
extern "C" int mmego_attn_pool_frag_bf16(void* stream, const unsigned short* hf, int T, int Bp, int Bn, int H, const float* w, const float* b,
                                         float* vec, float* attn) {
  MMEGO_REQUIRE(hf && w && b && vec && T > 0 && Bp > 0 && Bp % 32 == 0 && Bn > 0 && Bn <= Bp && mmego_attn_pool_frag_bf16_ok(H));
  MMEGO_REQUIRE((((uintptr_t)hf | (uintptr_t)vec) & 15) == 0);
  const dim3 grid(Bp / 32);
  const bf16_t* h = reinterpret_cast<const bf16_t*>(hf);
  if (H == 512) attn_pool_frag_bf16_kernel<8><<<grid, 512, 0, (hipStream_t)stream>>>(h, T, Bp, Bn, w, b, vec, attn);
  else if (H == 256) attn_pool_frag_bf16_kernel<4><<<grid, 512, 0, (hipStream_t)stream>>>(h, T, Bp, Bn, w, b, vec, attn);
  else attn_pool_frag_bf16_kernel<2><<<grid, 512, 0, (hipStream_t)stream>>>(h, T, Bp, Bn, w, b, vec, attn);
  MMEGO_LAUNCH_CHECK();
  return MMEGO_OK;
}
