// fp32-ACCURATE dense products on the bf16 matrix pipe ("split3"), for the frozen IMU_Net forward (reference Net/IMU_Net.py:58-62,
// 76-83: fc1 -> rnn_fast BiLSTM(512) over the 20 samples -> attention pooling -> rnn_slow BiLSTM(512) over T -> fc2).
//
// On gfx950 v_mfma_f32_32x32x16_bf16 runs at 16x the rate of the fp32 MFMAs.  An fp32 value is EXACTLY the sum of three bf16
// values: a1 = bf16(a), a2 = bf16(a - a1), a3 = bf16(a - a1 - a2) (round to nearest even; each residual is exact in fp32 and the
// third one fits bf16's 8 significant bits).  A product a.b is then the sum of nine exact piece products, of which
//     a1 b1,   a1 b2, a2 b1,   a2 b2, a1 b3, a3 b1          (relative size 1, 2^-8, 2^-8, 2^-16, 2^-16, 2^-16)
// are kept (NPROD = 6) and a2 b3, a3 b2 (2^-24), a3 b3 (2^-32) are dropped -- below the rounding of ONE fp32 product; NPROD = 9
// keeps them, to measure what they are worth.  All piece products are exact in fp32 and accumulate in fp32 inside the MFMA, so a
// 6-product sum over K has the error profile of an fp32 dot product at 6/16 of its matrix time.
// Outside the exact range: a value that rounds to +-inf in bf16 (|a| > 3.39e38) or is inf / NaN has a1 = that, a2 = a3 = 0
// (and inf x a zero piece is NaN where fp32 would say inf); below 2^-110 the third piece underflows bf16's subnormals (absolute
// error <= 2^-134 per operand).  Neither range occurs in the net's weights or activations; the mode is opt-in
// (IMUNet.precision = "split3") and the native fp32 engine stays the default.
//
// OPERAND LAYOUT ("sfrag": split, fragment-major).  The MFMA wants, per 16-k step s and 32-row block rb, 16 B per lane: lane l
// holds row 32 rb + l % 32, k = 16 s + 8 (l / 32) .. + 8.  Both operands of every product are stored that way, piece by piece:
//     block (rb, s, p) = 1 KB = 64 lanes x 8 bf16, at ((rb * SK + s) * 3 + p) KB,      SK = K / 16, p = piece 0..2,
// so a wave's operand fetch is one coalesced 1-KB read, an LDS image of a tile is a straight copy (lane-linear ds_read_b128, no
// padding, no conflicts), and the producers (fc1, the recurrent step's epilogue, the one-time weight re-layout) write whole
// blocks.  Activation rows are TIME-MAJOR (m = t * Bp + b, Bp = Bn rounded up to 32): a 32-row block is 32 sequences at one
// timestep, so the layer output buffer [T Bp / 32][2H / 16][3] KB is at once the next step's h_{t-1} operand (a strided window:
// one timestep, one direction's H columns) and the next layer's projection operand.
// The projection result is stored TILE-MAJOR fp32 (every 32 x 32 accumulator tile as it sits in registers; bf16.hip has the
// element formula) and the step kernel loads it back the same way.
//
//   s3_cvt_kernel        fp32 row-major -> sfrag pieces (weights once; small activations)
//   s3_fc_relu_kernel    IMU_Net's fc1 + ReLU (K <= 16) straight into the layer-0 operand
//   s3_gemm_kernel       C = A . W^T + bias on 6 (9) piece products, 64 / 128 / 256 x 128 tiles
//   s3_step_kernel       one BiLSTM timestep, both directions: gates = xproj + h_{t-1} . W_hh^T, cell update, h_t as pieces
#include <stdlib.h>

#include "common.h"

// Compile-time experiment mask (scripts/s3_experiments.py builds variant libraries with -DS3_EXP=<mask> to take a kernel's time apart
// by elimination; 0 -- nothing of it exists -- in the product build).  1: piece products replaced by an XOR of the fragments,
// 2: every chunk / step reads the FIRST one's addresses (operands cache-resident), 4: step kernel without reduction and cell update,
// 8: GEMM without its C stores.
#ifndef S3_EXP
#define S3_EXP 0
#endif

typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 s3_bf16x8;
typedef unsigned short s3_bf16_t;
typedef unsigned int s3_u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned s3_bits(float x) { return (unsigned)__builtin_bit_cast(unsigned short, (__bf16)x); }   // RNE: v_cvt_pk_bf16_f32
__device__ __forceinline__ float s3_up(unsigned b) { return __uint_as_float(b << 16); }

// a -> (a1, a2, a3) as bf16 bit patterns; a1 + a2 + a3 == a exactly for every finite a with 2^-110 <= |a| <= 3.38e38 and for 0
__device__ __forceinline__ void s3_split(float a, unsigned& p1, unsigned& p2, unsigned& p3) {
  p1 = s3_bits(a);
  const bool fin = (p1 & 0x7f80u) != 0x7f80u;           // a1 neither inf nor NaN
  const float r1 = fin ? a - s3_up(p1) : 0.f;
  p2 = s3_bits(r1);
  const float r2 = r1 - s3_up(p2);
  p3 = s3_bits(r2);
}

// 8 values -> the three 16-byte pieces of one lane
__device__ __forceinline__ void s3_split8(const float (&y)[8], s3_u32x4& o1, s3_u32x4& o2, s3_u32x4& o3) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    unsigned a1, a2, a3, b1, b2, b3;
    s3_split(y[2 * q], a1, a2, a3);
    s3_split(y[2 * q + 1], b1, b2, b3);
    o1[q] = a1 | (b1 << 16);
    o2[q] = a2 | (b2 << 16);
    o3[q] = a3 | (b3 << 16);
  }
}

// ---- fp32 row-major -> sfrag ------------------------------------------------------------------------------------------------
// Output row r (of Rp, a multiple of 32) takes input row r (tm == 0; rows >= rows_in are zero) or, time-major (tm != 0: r = t Bp + b),
// input row b T + t (b < Bn, else zero).
__global__ __launch_bounds__(256) void s3_cvt_kernel(const float* __restrict__ X, long ldx, int rows_in, int SK, int tm, int Bn, int T,
                                                      int Bp, s3_u32x4* __restrict__ Y, long nblk) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  const int lane = (int)(idx & 63);
  const long blk = idx >> 6;
  if (blk >= nblk) return;
  const long rb = blk / SK;
  const int s = (int)(blk - rb * SK);
  const long r = rb * 32 + (lane & 31);
  long src;
  bool live;
  if (tm) {
    const long t = r / Bp, b = r - t * Bp;
    live = b < Bn && t < T;
    src = b * T + t;
  } else {
    live = r < rows_in;
    src = r;
  }
  float y[8];
  const float* xr = X + (live ? src : 0) * ldx + 16 * s + 8 * (lane >> 5);
  const f32x4 v0 = *reinterpret_cast<const f32x4*>(xr), v1 = *reinterpret_cast<const f32x4*>(xr + 4);
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    y[e] = live ? v0[e] : 0.f;
    y[4 + e] = live ? v1[e] : 0.f;
  }
  s3_u32x4 o1, o2, o3;
  s3_split8(y, o1, o2, o3);
  s3_u32x4* dst = Y + blk * 192 + lane;
  dst[0] = o1;
  dst[64] = o2;
  dst[128] = o3;
}

// X [rows_in][K] fp32 (row stride ldx, 16-byte aligned rows) -> Y sfrag [Rp / 32][K / 16][3][64][8] bf16.  tm = 0: Rp >= rows_in
// rows in order (zero padded); tm = 1: Rp = T * Bp rows t * Bp + b from input rows b * T + t (b < Bn).
extern "C" int mmego_split3_cvt(void* stream, const float* X, long ldx, long rows_in, int K, int tm, int Bn, int T, int Bp, long Rp,
                                unsigned short* Y) {
  MMEGO_REQUIRE(X && Y && rows_in > 0 && K > 0 && K % 16 == 0 && ldx % 4 == 0 && Rp > 0 && Rp % 32 == 0 && rows_in < (1L << 31));
  MMEGO_REQUIRE((((uintptr_t)X) & 15) == 0 && (((uintptr_t)Y) & 15) == 0);
  if (tm) MMEGO_REQUIRE(Bn > 0 && T > 0 && Bp >= Bn && Bp % 32 == 0 && Rp == (long)T * Bp && rows_in == (long)Bn * T);
  else MMEGO_REQUIRE(Rp >= rows_in);
  const long nblk = (Rp / 32) * (K / 16);
  const long nthr = nblk * 64;
  MMEGO_REQUIRE(nthr / 256 + 1 < (1L << 31));
  s3_cvt_kernel<<<(unsigned)((nthr + 255) / 256), 256, 0, (hipStream_t)stream>>>(X, ldx, (int)rows_in, K / 16, tm, Bn, T, Bp,
                                                                                  reinterpret_cast<s3_u32x4*>(Y), nblk);
  MMEGO_LAUNCH_CHECK();
  return MMEGO_OK;
}

// The pieces of X^T: X [R][C] fp32 row-major (row stride ldx) -> sfrag of the [Cp x R] matrix whose row c is X's column c (Cp = C rounded
// up to 32, zero rows behind C; R % 16 == 0).  Lane l of block (rb, s) holds row 32 rb + l % 32 of X^T = COLUMN 32 rb + l % 32 of X and
// k = 16 s + 8 (l / 32) + e = ROW of X: for a fixed e the 32 lanes of a half wave read 32 consecutive floats of one row of X (128 B).
// For the weight-gradient products of stage-1 training (dW = dY^T X: both operands are read along the row axis).
// shift / T (T > 0): the k axis reads row r + shift of X where that row belongs to the same T-row sequence as r (rows b T + t), zero
// otherwise: h_{t-1} (shift -1) or h_{t+1} (shift +1, the reverse direction) of a BiLSTM layer's outputs without a shifted copy.
__global__ __launch_bounds__(256) void s3_cvt_t_kernel(const float* __restrict__ X, long ldx, int C, int SK, s3_u32x4* __restrict__ Y, long nblk,
                                                        int shift, int T) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  const int lane = (int)(idx & 63);
  const long blk = idx >> 6;
  if (blk >= nblk) return;
  const long rb = blk / SK;
  const int s = (int)(blk - rb * SK);
  const int c = (int)rb * 32 + (lane & 31);
  const bool live = c < C;
  const int r0 = 16 * s + 8 * (lane >> 5);
  const float* xc = X + (live ? c : 0);
  float y[8];
  bool ok[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int r = r0 + e;
    int t = 0;
    if (T > 0) t = r % T + shift;
    ok[e] = live && t >= 0 && (T <= 0 || t < T);
    y[e] = xc[(long)(ok[e] ? r + (T > 0 ? shift : 0) : r) * ldx];          // (clamped to a valid row, never predicated)
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) y[e] = ok[e] ? y[e] : 0.f;
  s3_u32x4 o1, o2, o3;
  s3_split8(y, o1, o2, o3);
  s3_u32x4* dst = Y + blk * 192 + lane;
  dst[0] = o1;
  dst[64] = o2;
  dst[128] = o3;
}

// X [R][C] fp32 (row stride ldx) -> Y = sfrag pieces of X^T: [Cp / 32][R / 16][3][64][8] bf16, Cp % 32 == 0, Cp >= C, R % 16 == 0.
extern "C" int mmego_split3_cvt_t(void* stream, const float* X, long ldx, long R, int C, long Cp, unsigned short* Y, int shift, int T) {
  MMEGO_REQUIRE(X && Y && R > 0 && R % 16 == 0 && R < (1L << 30) && C > 0 && Cp >= C && Cp % 32 == 0 && ldx >= C);
  MMEGO_REQUIRE(T >= 0 && (T == 0 ? shift == 0 : (R % T == 0 && shift > -T && shift < T)));
  MMEGO_REQUIRE((((uintptr_t)Y) & 15) == 0);
  const long nblk = (Cp / 32) * (R / 16);
  const long nthr = nblk * 64;
  MMEGO_REQUIRE(nthr / 256 + 1 < (1L << 31));
  s3_cvt_t_kernel<<<(unsigned)((nthr + 255) / 256), 256, 0, (hipStream_t)stream>>>(X, ldx, C, (int)(R / 16), reinterpret_cast<s3_u32x4*>(Y), nblk, shift, T);
  MMEGO_LAUNCH_CHECK();
  return MMEGO_OK;
}

// sfrag -> fp32 row-major (a1 + a2 + a3 in that order of addition: exact for a split of an fp32 value).  Test / debug aid.
__global__ __launch_bounds__(256) void s3_join_kernel(const s3_u32x4* __restrict__ Y, int SK, long nblk, float* __restrict__ X, long ldx) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  const int lane = (int)(idx & 63);
  const long blk = idx >> 6;
  if (blk >= nblk) return;
  const long rb = blk / SK;
  const int s = (int)(blk - rb * SK);
  const s3_u32x4* src = Y + blk * 192 + lane;
  const s3_u32x4 o1 = src[0], o2 = src[64], o3 = src[128];
  float* xr = X + (rb * 32 + (lane & 31)) * ldx + 16 * s + 8 * (lane >> 5);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    xr[2 * q] = (s3_up(o1[q] & 0xffffu) + s3_up(o2[q] & 0xffffu)) + s3_up(o3[q] & 0xffffu);
    xr[2 * q + 1] = (s3_up(o1[q] >> 16) + s3_up(o2[q] >> 16)) + s3_up(o3[q] >> 16);
  }
}

extern "C" int mmego_split3_join(void* stream, const unsigned short* Y, long Rp, int K, float* X, long ldx) {
  MMEGO_REQUIRE(X && Y && Rp > 0 && Rp % 32 == 0 && K > 0 && K % 16 == 0 && ldx >= K);
  const long nblk = (Rp / 32) * (K / 16);
  s3_join_kernel<<<(unsigned)((nblk * 64 + 255) / 256), 256, 0, (hipStream_t)stream>>>(reinterpret_cast<const s3_u32x4*>(Y), K / 16, nblk, X, ldx);
  MMEGO_LAUNCH_CHECK();
  return MMEGO_OK;
}

// ---- IMU_Net's fc1 + ReLU (Linear(15, H), Net/IMU_Net.py:53,73) straight into the layer-0 operand ---------------------------------
// Y (sfrag, rows t * Bp + b, K = H) = split(relu(X[b * T + t] . W^T + bias)).  One workgroup per (32-row block, timestep); W (padded to
// 16 columns) and the bias wait in LDS, a row's <= 16 inputs in registers; a thread produces the 8 consecutive units of one lane of
// a block, so a wave stores three whole 1-KB blocks per 16 units.  fp32 FMAs in k order (the fp32 path's fc1 is a product kernel with
// another summation order: same result to fp32 rounding).
__global__ __launch_bounds__(256) void s3_fc_relu_kernel(const float* __restrict__ X, long ldx, const float* __restrict__ W,
                                                          const float* __restrict__ bias, int Bn, int T, int Cin, int H,
                                                          s3_u32x4* __restrict__ Y, int Bp, int relu) {
  extern __shared__ __attribute__((aligned(16))) float s3_wsm[];      // [H][16] + bias [H]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // (W padded to 16 columns; loads unconditional on clamped addresses, 16 in flight: under `k < Cin ? W[..] : 0` every load was a round
  //  trip of its own -- 32 of them in a row were most of this kernel's 30 us)
  for (int i0 = 0; i0 < H * 16; i0 += 16 * 256) {
    float wv[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int i = min(i0 + u * 256 + tid, H * 16 - 1);
      wv[u] = W[(long)(i >> 4) * Cin + min(i & 15, Cin - 1)];
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int i = i0 + u * 256 + tid;
      if (i < H * 16) s3_wsm[i] = (i & 15) < Cin ? wv[u] : 0.f;
    }
  }
  for (int i = tid; i < H; i += 256) s3_wsm[H * 16 + i] = bias ? bias[i] : 0.f;
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
  const int SK = H >> 4;
  s3_u32x4* dst = Y + ((long)t * (Bp >> 5) + rb) * SK * 192 + lane;
  for (int k16 = wave; k16 < SK; k16 += 4) {
    const int n0 = k16 * 16 + half * 8;
    float y[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const f32x4* wr = reinterpret_cast<const f32x4*>(s3_wsm + (n0 + j) * 16);
      float a = s3_wsm[H * 16 + n0 + j];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 w4 = wr[q];
        a = fmaf(x[4 * q], w4[0], a); a = fmaf(x[4 * q + 1], w4[1], a); a = fmaf(x[4 * q + 2], w4[2], a); a = fmaf(x[4 * q + 3], w4[3], a);
      }
      y[j] = live ? (relu ? fmaxf(a, 0.f) : a) : 0.f;
    }
    s3_u32x4 o1, o2, o3;
    s3_split8(y, o1, o2, o3);
    dst[(long)k16 * 192] = o1;
    dst[(long)k16 * 192 + 64] = o2;
    dst[(long)k16 * 192 + 128] = o3;
  }
}

extern "C" int mmego_split3_fc_relu(void* stream, const float* X, long ldx, const float* W, const float* bias, int Bn, int T, int Cin,
                                    int H, unsigned short* Y, int Bp, int relu) {
  MMEGO_REQUIRE(X && W && Y && Bn > 0 && T > 0 && Cin > 0 && Cin <= 16 && H > 0 && H % 16 == 0 && H <= 2048 && Bp >= Bn && Bp % 32 == 0);
  MMEGO_REQUIRE(T <= 65535 && (((uintptr_t)Y) & 15) == 0);
  const size_t lds = (size_t)(H * 16 + H) * sizeof(float);
  static bool attr_set = false;
  if (!attr_set && lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)s3_fc_relu_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 2048 * 17 * 4);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  dim3 grid(Bp / 32, T);
  s3_fc_relu_kernel<<<grid, 256, lds, (hipStream_t)stream>>>(X, ldx, W, bias, Bn, T, Cin, H, reinterpret_cast<s3_u32x4*>(Y), Bp, relu);
  MMEGO_LAUNCH_CHECK();
  return MMEGO_OK;
}

// ---- the piece products of one (A tile, B tile) pair at one 16-k step --------------------------------------------------------
// small terms first, a1 b1 last (the accumulator is fp32 either way; this order keeps the partial sums' rounding smallest)
template <int NPROD>
__device__ __forceinline__ f32x16 s3_mma(const s3_u32x4 (&a)[3], const s3_u32x4 (&b)[3], f32x16 acc) {
  if (S3_EXP & 1) {
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const s3_u32x4 v = a[q] ^ b[q];
      acc[q] = __uint_as_float(__float_as_uint(acc[q]) ^ v[0] ^ v[1] ^ v[2] ^ v[3]);
    }
    return acc;
  }
#define S3_MM(i, j) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(s3_bf16x8, a[i]), __builtin_bit_cast(s3_bf16x8, b[j]), acc, 0, 0, 0)
  if (NPROD == 9) {
    S3_MM(2, 2);
    S3_MM(1, 2);
    S3_MM(2, 1);
  }
  S3_MM(0, 2);
  S3_MM(2, 0);
  S3_MM(1, 1);
  S3_MM(0, 1);
  S3_MM(1, 0);
  S3_MM(0, 0);
#undef S3_MM
  return acc;
}

// XCD-aware tile order: blocks b and b + 8 share an XCD; hand each XCD a contiguous run of tile ids.
__device__ __forceinline__ int s3_xcd_order(int id, int n) { return (n & 7) == 0 ? (id & 7) * (n >> 3) + (id >> 3) : id; }

// ---- C = A . W^T + bias -----------------------------------------------------------------------------------------------------------
struct S3GemmP {
  const s3_u32x4* A;       // sfrag [Mrb][SK][3][64]
  const s3_u32x4* W;       // sfrag [Nrb][SK][3][64]
  float* Cf;               // tile-major fp32 [Mrb][Nrb][1024] or null
  float* C; long ldc;      // row-major fp32 or null
  const float* bias;       // [32 Nrb] or null
  int Mrb, Nrb, SK, M;     // M: rows really stored to C (row-major output only)
  int tiles_m, tiles_n;
  int gm;                  // row panels per sweep group (tile order)
  int cps;                 // split-K (grid.y = number of K slabs): 32-k chunks per slab; slab y writes its partial product behind the
  long slab;               // others' (C + y * slab floats; Cf likewise); bias only in slab 0
};

// Workgroup tile (32 MI WM) x 128, WM x 2 waves, wave tile (32 MI) x 64 = MI x 2 MFMA tiles.  32-k chunks (2 16-k steps): the chunk's
// blocks of a row block are 6 consecutive KB in memory, copied as they lie into an LDS image [row block][step][piece][lane] through
// registers (the next chunk's loads are in flight while this one is multiplied); per 16-k step a wave reads 3 (MI + 2) fragments
// (lane-linear ds_read_b128) for 6 MI 2 (9 MI 2) MFMAs.  LDS: (WM MI + 4) x 6 KB = 48 KB (WM = MI = 2: 128 x 128 tiles, two or three
// workgroups per CU) / 72 KB (WM = 4: 256 x 128) / 36 KB (WM = 2, MI = 1: 64 x 128, for products with few rows).
template <int WM, int MI, int NPROD>
__global__ __launch_bounds__(WM * 128, 2) void s3_gemm_kernel(S3GemmP p) {
  constexpr int NT = WM * 128;                 // threads
  constexpr int RBA = WM * MI;                 // A row blocks per tile
  constexpr int NLA = RBA * 384 / NT;          // 16-byte loads per thread and chunk
  constexpr int NLW = 4 * 384 / NT;
  static_assert(RBA * 384 % NT == 0 && 4 * 384 % NT == 0, "tile images must divide over the threads");
  __shared__ s3_u32x4 As[RBA * 384];
  __shared__ s3_u32x4 Bs[4 * 384];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int wm = w >> 1, wn = w & 1;
  // tile order: runs of p.gm row panels sweep the N tiles, so that the workgroups resident on an XCD at one time share few A and few W
  // panels (64 resident 256 x 128 tiles as 4 row panels x 16 column tiles)
  const int id = s3_xcd_order(blockIdx.x, (int)gridDim.x);
  const int GM = p.gm;
  const int per_group = GM * p.tiles_n;
  const int group = id / per_group, in_group = id - group * per_group;
  const int gm = min(GM, p.tiles_m - group * GM);
  const int tm = group * GM + in_group % gm, tn = in_group / gm;
  const int rbA0 = tm * RBA, rbW0 = tn * 4;
  const int SK = p.SK;

  // piece i = j NT + tid of an operand's image: row block i / 384, offset i % 384 inside its 6-KB run
  int ga[NLA], gw[NLW];
#pragma unroll
  for (int j = 0; j < NLA; ++j) {
    const int i = j * NT + tid;
    ga[j] = min(rbA0 + i / 384, p.Mrb - 1) * SK * 192 + i % 384;
  }
#pragma unroll
  for (int j = 0; j < NLW; ++j) {
    const int i = j * NT + tid;
    gw[j] = min(rbW0 + i / 384, p.Nrb - 1) * SK * 192 + i % 384;
  }
  const int c0 = (int)blockIdx.y * p.cps;
  s3_u32x4 ra[NLA], rw[NLW];
#pragma unroll
  for (int j = 0; j < NLA; ++j) ra[j] = p.A[ga[j] + c0 * 384];
#pragma unroll
  for (int j = 0; j < NLW; ++j) rw[j] = p.W[gw[j] + c0 * 384];

  f32x16 acc[MI][2];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[mi][ni][i] = 0.f;

  const int nchunk = min(SK >> 1, c0 + p.cps);          // (this K slab: chunks [c0, nchunk))
  for (int c = c0; c < nchunk; ++c) {
    __syncthreads();                            // the previous chunk's fragments have been read
#pragma unroll
    for (int j = 0; j < NLA; ++j) As[j * NT + tid] = ra[j];
#pragma unroll
    for (int j = 0; j < NLW; ++j) Bs[j * NT + tid] = rw[j];
    __syncthreads();
    {
      const int cn = (S3_EXP & 2) ? 0 : min(c + 1, nchunk - 1) * 384;      // (unconditional prefetch; past the end: the last chunk again)
#pragma unroll
      for (int j = 0; j < NLA; ++j) ra[j] = p.A[ga[j] + cn];
#pragma unroll
      for (int j = 0; j < NLW; ++j) rw[j] = p.W[gw[j] + cn];
    }
#pragma unroll
    for (int kc = 0; kc < 2; ++kc) {
      s3_u32x4 a[MI][3], b[2][3];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int q = 0; q < 3; ++q) a[mi][q] = As[(((wm * MI + mi) * 2 + kc) * 3 + q) * 64 + lane];
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int q = 0; q < 3; ++q) b[ni][q] = Bs[(((wn * 2 + ni) * 2 + kc) * 3 + q) * 64 + lane];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = s3_mma<NPROD>(a[mi], b[ni], acc[mi][ni]);
    }
  }
  // accumulator layout of the 32 x 32 MFMA: register i of lane l is (row = 8 (i / 4) + 4 (l / 32) + i % 4, col = l % 32).
  // The bias goes into the accumulators in place and the stores read them where they are: a temporary per store would be one register
  // rewritten 16 times, each rewrite waiting for the store before it.
  const int fr = lane & 31;
#pragma unroll
  for (int ni = 0; ni < 2; ++ni) {
    const int cb = min(rbW0 + wn * 2 + ni, p.Nrb - 1);
    const float bv = (p.bias && blockIdx.y == 0) ? p.bias[cb * 32 + fr] : 0.f;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[mi][ni][i] += bv;
  }
#pragma unroll
  for (int ni = 0; ni < 2; ++ni) {
    const int cb = rbW0 + wn * 2 + ni;                 // column block
    if (cb >= p.Nrb) continue;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      const int rbm = rbA0 + wm * MI + mi;
      if (rbm >= p.Mrb) continue;
      if (p.Cf && !((S3_EXP & 8) && acc[mi][ni][0] != 12345.f)) {     // an accumulator tile IS a tile of the tile-major layout: four 1-KB stores
        float* t = p.Cf + (long)blockIdx.y * p.slab + ((long)rbm * p.Nrb + cb) * 1024 + lane * 4;
#pragma unroll
        for (int q = 0; q < 4; ++q)
          *reinterpret_cast<f32x4*>(t + q * 256) = (f32x4){acc[mi][ni][4 * q], acc[mi][ni][4 * q + 1], acc[mi][ni][4 * q + 2], acc[mi][ni][4 * q + 3]};
      }
      if (p.C) {
        float* cp = p.C + (long)blockIdx.y * p.slab + (long)(rbm * 32 + 4 * (lane >> 5)) * p.ldc + cb * 32 + fr;
        const int rows_left = p.M - (rbm * 32 + 4 * (lane >> 5));         // rows 8 (i / 4) + i % 4 below this lane's first
#pragma unroll
        for (int i = 0; i < 16; ++i)
          if (8 * (i >> 2) + (i & 3) < rows_left) cp[(long)(8 * (i >> 2) + (i & 3)) * p.ldc] = acc[mi][ni][i];
      }
    }
  }
}

// ---- the same product on 320 x 256 tiles, operands by LDS-DMA (r06) ---------------------------------------------------------------
// Why: the kernel above needs (256 + 128) x 32 k x 6 B = 74 KB per 32-k chunk for 6144 cycles of MFMA issue on its SIMDs (two workgroups per
// CU) = 24 B / clk / CU of operand traffic through two barriers per chunk and a register-staged copy -- PMC: MFMA busy 0.63, 3.0 TB/s of L2
// misses, 3.9 x the algorithmic bytes.  (r05 read that as "bound by what a CU can pull from L2"; profiles/r06_l2_delivery.txt measures 52-93
// B / clk / CU for a pure stream, so it was the kernel's structure, not the path.)  Arithmetic intensity is M N / (M + N) per tile: 85 for
// 256 x 128, **142 for 320 x 256** (14 B / clk / CU), and 10 240 x 4096 is exactly 32 x 16 = 512 such tiles = TWO full rounds of the 256 CUs
// (1280 tiles on 512 slots were 2.5).
//   Workgroup = one tile, 512 threads, ONE per CU (256 VGPRs per wave): waves 2 (M) x 4 (N), wave tile 160 x 64 = 5 x 2 MFMA tiles
//   (160 accumulator registers); per 16-k step a wave reads 2 x 3 W fragments once and 5 x 3 A fragments (lane-linear ds_read_b128)
//   for 60 (90) MFMAs.
//   Operands: a 16-k step of the tile is (10 + 8) row blocks x 3 KB = 54 KB, as it lies in memory; it goes global -> LDS by LDS-DMA
//   (global_load_lds_dwordx4: 54 1-KB transfers per step, 7 per wave, no VGPRs -- the register-staged copy of the kernel above would need
//   56 more), into a ring of TWO 54-KB stages; one barrier per step, placed INSIDE the step (see the schedule in the kernel).
//   Tile order: an XCD's 64 tiles are two rounds of 4 row panels x 8 column tiles over the SAME 4 row panels (A panels stay in its L2).
//   TRB = 10 (320-row tiles: the projections) or 8 (256-row tiles, wave tile 128 x 64: products whose row count is a multiple of 256 but
//   not of 320 -- the weight gradients of stage-1 training, 2048 / 4096 rows over K = 10 240).  grid.y = K slabs (p.cps 32-k chunks each):
//   slab y leaves its partial product at C + y * p.slab (mmego_split3_gemm_slabs), bias in slab 0 only.
template <int NPROD, int TRB>
__global__ __launch_bounds__(512, 1) void s3_gemm_big_kernel(S3GemmP p) {
  constexpr int TCB = 8, NBLK = (TRB + TCB) * 3, MI = TRB / 2;   // 1-KB blocks per step and stage; accumulator row blocks per wave
  extern __shared__ __attribute__((aligned(16))) s3_u32x4 s3_big[];   // [2 stages][NBLK][64]
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);        // (wave-uniform by construction: addresses below stay scalar)
  const int wm = w >> 2, wn = w & 3;
  int tm, tn;
  {
    const int n = (int)gridDim.x;
    if ((n & 63) == 0 && (p.tiles_m & 3) == 0 && (p.tiles_n & 7) == 0) {
      const int x = blockIdx.x & 7, l = blockIdx.x >> 3;          // XCD (round-robin dispatch), index inside the XCD's run
      const int per_x = n >> 3, rounds = per_x >> 5;             // 32 tiles = 4 x 8 per round
      const int blk = x * rounds + (l >> 5), within = l & 31;     // block of 4 row panels x 8 column tiles
      const int bn_count = p.tiles_n >> 3;
      tm = (blk / bn_count) * 4 + (within & 3);
      tn = (blk % bn_count) * 8 + (within >> 2);
    } else {
      tm = blockIdx.x % p.tiles_m;
      tn = blockIdx.x / p.tiles_m;
    }
  }
  const int rbA0 = tm * TRB, rbW0 = tn * TCB;
  const int SKall = p.SK;
  const int s_first = (int)blockIdx.y * 2 * p.cps;               // this slab's 16-k steps: [s_first, s_first + SK)
  const int SK = min(SKall - s_first, 2 * p.cps);
  // this wave's transfers: blocks i = w + 8 j of a stage (i < NBLK); block i < 3 TRB: A row block i / 3, piece i % 3; else W.
  // Source = a wave-uniform block address (scalar registers) + 16 lane.
  const s3_u32x4* gp[7];
#pragma unroll
  for (int j = 0; j < 7; ++j) {
    const int i = min(w + 8 * j, NBLK - 1);
    const int isw = i >= TRB * 3, ii = isw ? i - TRB * 3 : i;
    const int rb = (isw ? rbW0 : rbA0) + ii / 3;
    gp[j] = (isw ? p.W : p.A) + (((long)rb * SKall + s_first) * 3 + ii % 3) * 64;
  }
#define S3_BIG_DMA(s, stage)                                                                                          \
  {                                                                                                                   \
    _Pragma("unroll") for (int j = 0; j < 7; ++j)                                                                     \
      if (w + 8 * j < NBLK && !(S3_EXP & 16))                                                                         \
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gp[j] + (long)((S3_EXP & 2) ? 0 : (s)) * 192 + lane), \
                                         (__attribute__((address_space(3))) void*)(s3_big + ((stage) * NBLK + w + 8 * j) * 64), 16, 0, 0); \
  }
  f32x16 acc[MI][2];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[mi][ni][i] = 0.f;
  // Schedule (steps in pairs, so that every register index below is a compile-time constant).  At the top of step s its stage holds
  // its data and its W fragments + first A fragments are requested.  Inside the step: in front of row block mi's 12 (18) MFMAs an empty
  // asm READS that row block's fragments -- the compiler waits with lgkmcnt(0) wherever it waits in a loop that also holds LDS-DMA
  // requests, so the wait has to land where only reads requested a whole MFMA group ago are outstanding -- then row block mi + 1's
  // fragments are requested.  In front of the LAST row block's MFMAs (all of the stage's fragments are in registers) the wave waits for
  // its transfers of step s + 1, meets the others at THE step's one barrier, requests step s + 2 into the stage it has just finished
  // with (waves 0-3 here, waves 4-7 behind the MFMAs: a 1-KB LDS-DMA costs its wave 60-185 cycles of issue, and while one wave of a
  // SIMD issues them the other one issues MFMAs) and the first fragments of step s + 1.  No barrier sits between two steps, and a
  // transfer has a whole step to land.  (r06's first schedule -- requests in front of the previous group's MFMAs without the pinned
  // wait, barrier in front of row block 3 -- was 1.5-2.5 % slower; git history.)
  // A-fragment slots: row block mi of a step of parity par sits in slot (par + mi) & 1 for odd MI, mi & 1 for even MI -- either way the
  // next step's first fragments, requested in front of the last row block's MFMAs, go to the slot those MFMAs do not read.
  s3_u32x4 b[2][2][3], a[2][3];
#define S3_SLOT(par, mi) ((MI & 1) ? (((par) + (mi)) & 1) : ((mi) & 1))
#define S3_BIG_RD_B(slot, stage)                                                                                      \
  if (!(S3_EXP & 32)) _Pragma("unroll") for (int ni = 0; ni < 2; ++ni)                                                \
    _Pragma("unroll") for (int q = 0; q < 3; ++q) b[slot][ni][q] = s3_big[((stage) * NBLK + TRB * 3 + (wn * 2 + ni) * 3 + q) * 64 + lane];
#define S3_BIG_RD_A(slot, stage, mi)                                                                                  \
  if (!(S3_EXP & 32)) _Pragma("unroll") for (int q = 0; q < 3; ++q) a[slot][q] = s3_big[((stage) * NBLK + (wm * MI + (mi)) * 3 + q) * 64 + lane];
#define S3_USE3(x) asm volatile("" ::"v"((x)[0]), "v"((x)[1]), "v"((x)[2]))
#define S3_BIG_STEP(s, par)                                                                                           \
  {                                                                                                                   \
    _Pragma("unroll") for (int mi = 0; mi < MI; ++mi) {                                                               \
      if (!(S3_EXP & 32)) {                                                                                           \
        S3_USE3(a[S3_SLOT(par, mi)]);                                                                                 \
        if (mi == 0) { S3_USE3(b[par][0]); S3_USE3(b[par][1]); }                                                      \
      }                                                                                                               \
      __builtin_amdgcn_sched_barrier(0);                                                                              \
      if (mi < MI - 1) { S3_BIG_RD_A(S3_SLOT(par, mi + 1), par, mi + 1) }                                             \
      if (mi == MI - 1) {                                                                                             \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                              \
        __builtin_amdgcn_s_barrier();                                                                                 \
        if (wm == 0 && (s) + 2 < SK) S3_BIG_DMA((s) + 2, par)                                                         \
        if ((s) + 1 < SK) {                                                                                           \
          S3_BIG_RD_B((par) ^ 1, (par) ^ 1)                                                                           \
          S3_BIG_RD_A(S3_SLOT((par) ^ 1, 0), (par) ^ 1, 0)                                                            \
        }                                                                                                             \
      }                                                                                                               \
      __builtin_amdgcn_sched_barrier(0);                                                                              \
      _Pragma("unroll") for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = s3_mma<NPROD>(a[S3_SLOT(par, mi)], b[par][ni], acc[mi][ni]); \
      __builtin_amdgcn_sched_barrier(0);                                                                              \
      if (mi == MI - 1 && wm == 1 && (s) + 2 < SK) S3_BIG_DMA((s) + 2, par)                                            \
    }                                                                                                                 \
  }
  S3_BIG_DMA(0, 0)
  if (SK > 1) S3_BIG_DMA(1, 1)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  S3_BIG_RD_B(0, 0)
  S3_BIG_RD_A(S3_SLOT(0, 0), 0, 0)
  for (int s = 0; s < SK; s += 2) {
    S3_BIG_STEP(s, 0)
    if (s + 1 < SK) S3_BIG_STEP(s + 1, 1)
  }
#undef S3_BIG_STEP
#undef S3_USE3
#undef S3_SLOT
#undef S3_BIG_RD_A
#undef S3_BIG_RD_B
#undef S3_BIG_DMA
  const int fr = lane & 31;
  float* const Cs = p.C ? p.C + (long)blockIdx.y * p.slab : nullptr;
  float* const Cfs = p.Cf ? p.Cf + (long)blockIdx.y * p.slab : nullptr;
#pragma unroll
  for (int ni = 0; ni < 2; ++ni) {
    const int cb = rbW0 + wn * 2 + ni;
    const float bv = (p.bias && blockIdx.y == 0) ? p.bias[cb * 32 + fr] : 0.f;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      const int rbm = rbA0 + wm * MI + mi;
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[mi][ni][i] += bv;
      if (Cfs) {
        float* t = Cfs + ((long)rbm * p.Nrb + cb) * 1024 + lane * 4;
#pragma unroll
        for (int q = 0; q < 4; ++q)
          *reinterpret_cast<f32x4*>(t + q * 256) = (f32x4){acc[mi][ni][4 * q], acc[mi][ni][4 * q + 1], acc[mi][ni][4 * q + 2], acc[mi][ni][4 * q + 3]};
      }
      if (Cs) {
        float* cp = Cs + (long)(rbm * 32 + 4 * (lane >> 5)) * p.ldc + cb * 32 + fr;
        const int rows_left = p.M - (rbm * 32 + 4 * (lane >> 5));
#pragma unroll
        for (int i = 0; i < 16; ++i)
          if (8 * (i >> 2) + (i & 3) < rows_left) cp[(long)(8 * (i >> 2) + (i & 3)) * p.ldc] = acc[mi][ni][i];
      }
    }
  }
}

// C[m][n] = sum_k A[m][k] W[n][k] + bias[n] on 6 (nprod = 6) or 9 piece products; A sfrag [Mrb][K / 16][3] KB, W sfrag [Nrb][K / 16][3] KB
// (mmego_split3_cvt).  Cf: tile-major fp32 [Mrb][Nrb][1024] and / or C: row-major (rows < M stored, row stride ldc).  K % 32 == 0.
// wm: tile rows / 64 -- 1: 64 x 128 tiles (256-thread workgroups of four 32 x 64 wave tiles: products with few rows), 2: 128 x 128
// tiles (256 threads), 4: 256 x 128 tiles (512 threads); 0: the library's choice.
static int s3_gemm_launch(void* stream, const unsigned short* A, const unsigned short* W, float* Cf, float* C, long ldc, const float* bias,
                          int Mrb, int Nrb, int K, int M, int nprod, int wm, int nsplit, long slab) {
  MMEGO_REQUIRE(A && W && (Cf || C) && Mrb > 0 && Nrb > 0 && K > 0 && K % 32 == 0 && (nprod == 6 || nprod == 9));
  MMEGO_REQUIRE((((uintptr_t)A) & 15) == 0 && (((uintptr_t)W) & 15) == 0 && (!Cf || (((uintptr_t)Cf) & 15) == 0));
  MMEGO_REQUIRE(!C || (M > 0 && M <= Mrb * 32 && ldc >= Nrb * 32));
  MMEGO_REQUIRE((long)Mrb * (K / 16) * 192 < (1L << 31) && (long)Nrb * (K / 16) * 192 < (1L << 31));
  MMEGO_REQUIRE(nsplit >= 1 && nsplit <= 64 && (nsplit == 1 || slab > 0));
  // wm = 0 (the library's choice), 10 or 8: 320 x 256 / 256 x 256 tiles where they divide the product and (with the K slabs) fill the
  // chip -- the projections of rnn_fast (10 240 x 4096 = 512 tiles of 320 x 256), the weight gradients of stage-1 training (2048 / 4096
  // rows: 256 x 256 tiles x K slabs)
  {
    const int cps = cdiv(K / 32, nsplit);
    const bool fits10 = Mrb % 10 == 0 && Nrb % 8 == 0, fits8 = Mrb % 8 == 0 && Nrb % 8 == 0;
    const long t10 = fits10 ? (long)(Mrb / 10) * (Nrb / 8) * nsplit : 0, t8 = fits8 ? (long)(Mrb / 8) * (Nrb / 8) * nsplit : 0;
    int trb = 0;
    if (wm == 10 && fits10) trb = 10;
    else if (wm == 8 && fits8) trb = 8;
    else if (wm == 0 && t10 >= 256) trb = 10;
    else if (wm == 0 && t8 >= 256 && (nsplit == 1 || 2 * cps >= 16)) trb = 8;        // (slabs of at least 256 k)
    if (trb) {
      MMEGO_REQUIRE((long)(nsplit - 1) * cps < K / 32);          // (no empty slab)
      S3GemmP p;
      p.A = reinterpret_cast<const s3_u32x4*>(A); p.W = reinterpret_cast<const s3_u32x4*>(W);
      p.Cf = Cf; p.C = C; p.ldc = ldc; p.bias = bias; p.Mrb = Mrb; p.Nrb = Nrb; p.SK = K / 16; p.M = M;
      p.tiles_m = Mrb / trb; p.tiles_n = Nrb / 8; p.gm = 4; p.cps = cps; p.slab = slab;
      constexpr int lds = 2 * 54 * 1024;
      static bool attr_set[64] = {};
      int dev = 0;
      if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return MMEGO_EBADARG;
      if (!attr_set[dev]) {
        hipError_t e = hipFuncSetAttribute((const void*)s3_gemm_big_kernel<6, 10>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)s3_gemm_big_kernel<9, 10>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)s3_gemm_big_kernel<6, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)s3_gemm_big_kernel<9, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return (int)e;
        attr_set[dev] = true;
      }
      const dim3 grid((unsigned)(p.tiles_m * p.tiles_n), (unsigned)nsplit);
      hipStream_t st = (hipStream_t)stream;
      if (trb == 10) {
        if (nprod == 6) s3_gemm_big_kernel<6, 10><<<grid, 512, lds, st>>>(p);
        else s3_gemm_big_kernel<9, 10><<<grid, 512, lds, st>>>(p);
      } else {
        if (nprod == 6) s3_gemm_big_kernel<6, 8><<<grid, 512, lds, st>>>(p);
        else s3_gemm_big_kernel<9, 8><<<grid, 512, lds, st>>>(p);
      }
      MMEGO_LAUNCH_CHECK();
      return MMEGO_OK;
    }
  }
  if (wm == 10 || wm == 8) return MMEGO_EBADARG;
  if (wm == 0) {
    const long t128 = (long)cdiv(Mrb, 4) * cdiv(Nrb, 4) * nsplit;        // 128 x 128 tiles
    wm = t128 < 256 ? 1 : (Mrb >= 64 && (long)cdiv(Mrb, 8) * cdiv(Nrb, 4) * nsplit >= 256 ? 4 : 2);     // fewer tiles than CUs: smaller tiles
  }
  MMEGO_REQUIRE(wm == 1 || wm == 2 || wm == 4);
  S3GemmP p;
  p.A = reinterpret_cast<const s3_u32x4*>(A); p.W = reinterpret_cast<const s3_u32x4*>(W);
  p.Cf = Cf; p.C = C; p.ldc = ldc; p.bias = bias; p.Mrb = Mrb; p.Nrb = Nrb; p.SK = K / 16; p.M = M;
  p.tiles_m = cdiv(Mrb, 2 * wm); p.tiles_n = cdiv(Nrb, 4);
  p.gm = 4;         // (r05: 4 / 8 / 16 / 32 row panels per sweep group measured 194 / 193 / 197 / 205 us at K = 512, 369 / 373 / 377 / 387 at 1024)
  p.cps = cdiv(K / 32, nsplit); p.slab = slab;
  MMEGO_REQUIRE((long)(nsplit - 1) * p.cps < K / 32);          // (no empty slab: its first loads would lie behind the operands)
  const long tiles = (long)p.tiles_m * p.tiles_n;
  MMEGO_REQUIRE(tiles < (1L << 30));
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((unsigned)tiles, (unsigned)nsplit);
  if (wm == 1) {
    if (nprod == 6) s3_gemm_kernel<2, 1, 6><<<grid, 256, 0, st>>>(p);
    else s3_gemm_kernel<2, 1, 9><<<grid, 256, 0, st>>>(p);
  } else if (wm == 2) {
    if (nprod == 6) s3_gemm_kernel<2, 2, 6><<<grid, 256, 0, st>>>(p);
    else s3_gemm_kernel<2, 2, 9><<<grid, 256, 0, st>>>(p);
  } else {
    if (nprod == 6) s3_gemm_kernel<4, 2, 6><<<grid, 512, 0, st>>>(p);
    else s3_gemm_kernel<4, 2, 9><<<grid, 512, 0, st>>>(p);
  }
  MMEGO_LAUNCH_CHECK();
  return MMEGO_OK;
}

extern "C" int mmego_split3_gemm(void* stream, const unsigned short* A, const unsigned short* W, float* Cf, float* C, long ldc,
                                 const float* bias, int Mrb, int Nrb, int K, int M, int nprod, int wm) {
  return s3_gemm_launch(stream, A, W, Cf, C, ldc, bias, Mrb, Nrb, K, M, nprod, wm, 1, 0);
}

// The same product cut into nsplit slabs of K (long contractions with few output tiles: the weight gradients of stage-1 training,
// K = the 10 240 rows): slab y leaves its partial product row-major at ws + y * (32 Mrb) * (32 Nrb) floats (the format of mmego_gemm's
// accumulate = 2, summed by mmego_slab_reduce kind 0).
extern "C" int mmego_split3_gemm_slabs(void* stream, const unsigned short* A, const unsigned short* W, float* ws, int Mrb, int Nrb, int K,
                                       int nprod, int wm, int nsplit) {
  return s3_gemm_launch(stream, A, W, nullptr, ws, (long)Nrb * 32, nullptr, Mrb, Nrb, K, Mrb * 32, nprod, wm, nsplit,
                        (long)Mrb * 32 * Nrb * 32);
}

// out[i] = sum over the nsplit slabs of ws[y][i], i < n (n % 4 == 0): the K slabs of mmego_split3_gemm_slabs, added in slab order.  A plain
// streaming sum (16-byte loads, every slab's request in flight before the first add): the weight gradients' slabs are 8-67 MB each,
// for which mmego_slab_reduce's 16-outputs-per-workgroup form ran at 0.7 TB/s.
__global__ __launch_bounds__(256) void s3_slab_sum_kernel(const f32x4* __restrict__ ws, int nsplit, long n4, f32x4* __restrict__ out) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  f32x4 acc = ws[i];
  int y = 1;
  for (; y + 3 < nsplit; y += 4) {
    const f32x4 a = ws[(long)y * n4 + i], b = ws[(long)(y + 1) * n4 + i], c = ws[(long)(y + 2) * n4 + i], d = ws[(long)(y + 3) * n4 + i];
    acc += a; acc += b; acc += c; acc += d;
  }
  for (; y < nsplit; ++y) acc += ws[(long)y * n4 + i];
  out[i] = acc;
}

extern "C" int mmego_split3_slab_sum(void* stream, const float* ws, int nsplit, long n, float* out) {
  MMEGO_REQUIRE(ws && out && nsplit >= 1 && n > 0 && n % 4 == 0 && n / 4 / 256 + 1 < (1L << 31));
  MMEGO_REQUIRE((((uintptr_t)ws | (uintptr_t)out) & 15) == 0);
  s3_slab_sum_kernel<<<(unsigned)((n / 4 + 255) / 256), 256, 0, (hipStream_t)stream>>>(reinterpret_cast<const f32x4*>(ws), nsplit, n / 4,
                                                                                    reinterpret_cast<f32x4*>(out));
  MMEGO_LAUNCH_CHECK();
  return MMEGO_OK;
}

// ---- one BiLSTM timestep -------------------------------------------------------------------------------------------------------------
struct S3StepP {
  const s3_u32x4* hprev[2]; long hrb;   // h_{t-1} pieces: block (rb, s, p) at hprev[d] + (rb * hrb + s * 3 + p) * 64; hrb = blocks between row blocks
  const s3_u32x4* whh[2];               // W_hh pieces, sfrag with rows [hidden block jb][gate][32 units], SK = H / 16
  const float* xpf; long mt0[2];        // x . W_ih^T + b_ih + b_hh, tile-major; mt0[d] = first row tile of direction d's timestep
  float* hout[2]; long hos;             // h_t fp32 row-major (may be null)
  s3_u32x4* hnext[2]; long hnrb;        // h_t pieces, addressed like hprev
  float* c[2];
  int Bn, H, first, dbase;             // dbase: direction of slot 0 (a single-direction launch of the reverse direction: 1)
  int S;                               // 16-k steps of the product: H / 16 for a timestep, K / 16 for mmego_split3_proj
};

__device__ __forceinline__ float s3_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float s3_tanh(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * x)); }

// The small-batch step of the bf16 mode (lstm_step_bf16_direct_kernel, bf16.hip) on piece products.  Workgroup = 64 rows x 32 hidden
// units x 4 gates of one direction (Bn = H = 512: 8 x 16 x 2 = 256 workgroups, one per CU); wave w takes k quarter w of the whole tile,
// so no fragment is fetched twice, and streams its SQ 16-k steps through a ring of four fragment sets (18 coalesced 1-KB reads per
// step, requested three steps = 144 MFMAs ahead: with two, every step still waited for memory -- product and fetch added up); the four partial tiles meet in LDS (fixed order) and all four waves run the cell
// update of 16 rows each.  h_t leaves as 12 whole 1-KB blocks per workgroup (pieces put together in LDS), and optionally as fp32 rows.
// MODE 0: a timestep with its product; 1: the product-less first timestep; 2: the product alone, as a projection for few rows (IMU_Net's
// rnn_slow: 512 rows) -- out[row][gate H + j] = A[row][:] . W[gate H + j][:] + bias, row-major, K = 16 p.S: the chunked kernel above
// is latency-bound there (64 x 128 tiles, one workgroup per CU, a memory round trip per 32-k chunk: 55 us against gemm_tile's 50),
// this one keeps three steps of requests in flight per wave (mmego_split3_proj).
template <int SQ, int NPROD, int MODE>
__global__ __launch_bounds__(256, 1) void s3_step_kernel(S3StepP p) {
  constexpr bool FIRST = MODE == 1, PROJ = MODE == 2;
  extern __shared__ __attribute__((aligned(16))) float s3_red[];      // [4 waves][2 mi][4 n][16 i][64 lanes] = 128 KB
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int d = blockIdx.z, H = p.H, S = p.S;
  const int nrb = gridDim.y, nb = gridDim.x * nrb;
  const int id = s3_xcd_order(blockIdx.y * gridDim.x + blockIdx.x, nb);
  const int jb = id / nrb, j0 = jb * 32, r0 = (id % nrb) * 64;
  const int fr = lane & 31, fh = lane >> 5;
  const int j = j0 + fr;
  const int own_mi = w >> 1, own_i0 = 8 * (w & 1);     // this wave finishes rows 32 own_mi + 16 (w & 1) .. + 16:
  const int own_r0 = own_mi * 32 + 16 * (w & 1);       // accumulator registers own_i0 .. own_i0 + 8 of row block own_mi
  const int last_rb = (p.Bn - 1) >> 5;
  const int hb = H >> 5;
  [[maybe_unused]] const int s3_stamp_id = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
  MMEGO_STAMP_AT(s3_stamp_id, 0, tid == 0);

  // operand pointers and the first two steps' fragment requests go out BEFORE the cell update's own operands (xproj tile, c_{t-1}):
  // one memory round trip for all of them (asked for first, the compiler parked the xproj values in AGPRs behind a vmcnt(0) and
  // only then issued the first fragment load)
  f32x16 acc[2][4];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[mi][n][i] = 0.f;
  const s3_u32x4* ap[2];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) ap[mi] = p.hprev[d] + ((long)min((r0 >> 5) + mi, last_rb) * p.hrb + (long)w * SQ * 3) * 64 + lane;
  const s3_u32x4* wp = p.whh[d] + ((long)jb * 4 * S + w * SQ) * 192 + lane;
  const int gstride = S * 192;                        // between the gates' row blocks
  constexpr int NR = SQ >= 4 ? 4 : SQ;                // ring of fragment sets: requests run NR - 1 steps (48 (NR - 1) MFMAs) ahead
  s3_u32x4 a[NR][2][3], b[NR][4][3];
#define S3_LOAD(slot, s)                                                                    \
    {                                                                                       \
      _Pragma("unroll") for (int q = 0; q < 3; ++q) a[slot][0][q] = ap[0][((S3_EXP & 2 ? 0 : (s)) * 3 + q) * 64];        \
      _Pragma("unroll") for (int n = 0; n < 4; ++n)                                         \
        _Pragma("unroll") for (int q = 0; q < 3; ++q) b[slot][n][q] = wp[n * gstride + ((S3_EXP & 2 ? 0 : (s)) * 3 + q) * 64]; \
      _Pragma("unroll") for (int q = 0; q < 3; ++q) a[slot][1][q] = ap[1][((S3_EXP & 2 ? 0 : (s)) * 3 + q) * 64];        \
    }
  if (!FIRST) {                                         // (compile time: the step loop below is straight-line code)
#pragma unroll
    for (int s = 0; s < NR - 1; ++s) S3_LOAD(s, s)
  }
  __builtin_amdgcn_sched_barrier(0);
  // the cell update's own operands (this wave's part of the projection tile, c_{t-1}): requested two steps before the end of the
  // product loop, not in front of it (16 more requests on top of the ring's 54 would run into the 63 a wave may have outstanding)
  f32x4 xp[4][2];
  float cprev[8];
#define S3_CELL_OPERANDS()                                                                                                       \
  {                                                                                                                              \
    const int rbx = min((r0 >> 5) + own_mi, last_rb);                                                                            \
    _Pragma("unroll") for (int n = 0; n < 4; ++n)                                                                                \
      _Pragma("unroll") for (int qq = 0; qq < 2; ++qq)                                                                           \
        xp[n][qq] = (reinterpret_cast<const f32x4*>(p.xpf + ((p.mt0[d] + rbx) * (long)(8 * hb) + ((p.dbase + d) * 4 + n) * hb + jb) * 1024) + lane)[(2 * (w & 1) + qq) * 64]; \
    _Pragma("unroll") for (int ii = 0; ii < 8; ++ii) {                                                                           \
      const int row = min(r0 + own_r0 + 8 * (ii >> 2) + 4 * fh + (ii & 3), p.Bn - 1);                                            \
      cprev[ii] = FIRST ? 0.f : p.c[d][(long)row * H + j];                                                                       \
    }                                                                                                                            \
  }
  if (FIRST) S3_CELL_OPERANDS()
  float bproj[4] = {0.f, 0.f, 0.f, 0.f};
  if (PROJ && p.xpf) {
#pragma unroll
    for (int n = 0; n < 4; ++n) bproj[n] = p.xpf[(long)(p.dbase + d) * 4 * H + n * H + j];       // (MODE 2: xpf = the bias vector [2][4H])
  }
  float pre[4][8];
#pragma unroll
  for (int n = 0; n < 4; ++n)
#pragma unroll
    for (int ii = 0; ii < 8; ++ii) pre[n][ii] = 0.f;

  if (!FIRST) {
    MMEGO_STAMP_AT(s3_stamp_id, 1, tid == 0);
#pragma unroll
    for (int s = 0; s < SQ; ++s) {
      // The requests of step s + NR - 1 go out BETWEEN this step's MFMA groups, three per group: a wave is alone on its SIMD, so
      // while it issues a burst of 18 loads (all four waves at once, the CU's load path takes them at ~1 KB per 16 cycles) it issues
      // no MFMA -- with the burst in front of the step, product time and fetch time ADDED UP (timing by elimination,
      // scripts/s3_experiments.py: 19.5 us per timestep, 15.4 with cache-resident operands, 14.7 without the MFMAs, 11.5 with neither).
      constexpr bool more = true;
      const int sn = s + NR - 1, slot = (s + NR - 1) % NR;
      if (!PROJ && s == (SQ >= 2 ? SQ - 2 : 0)) S3_CELL_OPERANDS()
#pragma unroll
      for (int g = 0; g < 8; ++g) {
        const int mi = g >> 2, n = g & 3;
        acc[mi][n] = s3_mma<NPROD>(a[s % NR][mi], b[s % NR][n], acc[mi][n]);
        if (more && sn < SQ && g < 6) {
          // order of need in step sn: a[0], b[0..3], a[1]
#pragma unroll
          for (int q = 0; q < 3; ++q) {
            if (g == 0) a[slot][0][q] = ap[0][((S3_EXP & 2 ? 0 : sn) * 3 + q) * 64];
            else if (g == 5) a[slot][1][q] = ap[1][((S3_EXP & 2 ? 0 : sn) * 3 + q) * 64];
            else b[slot][g - 1][q] = wp[(g - 1) * gstride + ((S3_EXP & 2 ? 0 : sn) * 3 + q) * 64];
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#undef S3_LOAD
#undef S3_CELL_OPERANDS
    MMEGO_STAMP_AT(s3_stamp_id, 2, tid == 0);
    if (S3_EXP & 4) {                                   // (experiment: no reduction -- every wave keeps its own partial sums)
#pragma unroll
      for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int ii = 0; ii < 8; ++ii) pre[n][ii] = acc[own_mi][n][own_i0 + ii] + acc[1 - own_mi][n][own_i0 + ii];
    } else {
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
          for (int i = 0; i < 16; ++i) s3_red[(((w * 2 + mi) * 4 + n) * 16 + i) * 64 + lane] = acc[mi][n][i];
      __syncthreads();
#pragma unroll
      for (int src = 0; src < 4; ++src)
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
          for (int ii = 0; ii < 8; ++ii) pre[n][ii] += s3_red[(((src * 2 + own_mi) * 4 + n) * 16 + own_i0 + ii) * 64 + lane];
      __syncthreads();                                  // (the reduction buffer becomes the piece image below)
    }
  }
  if (PROJ) {                                           // the product alone: 8 rows x 4 gate columns per lane, row-major
#pragma unroll
    for (int ii = 0; ii < 8; ++ii) {
      const int row = r0 + own_r0 + 8 * (ii >> 2) + 4 * fh + (ii & 3);
      if (row < p.Bn) {
#pragma unroll
        for (int n = 0; n < 4; ++n) p.hout[d][(long)row * p.hos + n * H + j] = pre[n][ii] + bproj[n];
      }
    }
    return;
  }
  // cell update of this lane's 8 (row, unit j) elements; PyTorch gate order i, f, g, o
  s3_bf16_t* img = reinterpret_cast<s3_bf16_t*>(s3_red);            // [2 rb][2 s][3 p][64 lanes][8] bf16 = 12 KB
#pragma unroll
  for (int ii = 0; ii < 8; ++ii) {
    const int rl = own_r0 + 8 * (ii >> 2) + 4 * fh + (ii & 3);      // row inside the workgroup's 64
    const int row = r0 + rl;
    const float gi = s3_sigmoid(pre[0][ii] + xp[0][ii >> 2][ii & 3]);
    const float gf = s3_sigmoid(pre[1][ii] + xp[1][ii >> 2][ii & 3]);
    const float gg = s3_tanh(pre[2][ii] + xp[2][ii >> 2][ii & 3]);
    const float go = s3_sigmoid(pre[3][ii] + xp[3][ii >> 2][ii & 3]);
    const float cn = gf * (FIRST ? 0.f : cprev[ii]) + gi * gg;
    const float hn = row < p.Bn ? go * s3_tanh(cn) : 0.f;
    if (row < p.Bn) {
      p.c[d][(long)row * H + j] = cn;
      if (p.hout[d]) p.hout[d][(long)row * p.hos + j] = hn;
    }
    unsigned q1, q2, q3;
    s3_split(hn, q1, q2, q3);
    // element (row rl, k = fr of the workgroup's 32 units): block (rl / 32, fr / 16), lane (rl % 32) + 32 ((fr / 8) & 1), e = fr % 8
    const int o = ((((rl >> 5) * 2 + (fr >> 4)) * 3) * 64 + (rl & 31) + 32 * ((fr >> 3) & 1)) * 8 + (fr & 7);
    img[o] = (s3_bf16_t)q1;
    img[o + 512] = (s3_bf16_t)q2;
    img[o + 1024] = (s3_bf16_t)q3;
  }
  __syncthreads();
  {
    // 768 16-byte pieces: image index i = ((rb_l * 2 + s_l) * 3 + p) * 64 + lane
    const s3_u32x4* im4 = reinterpret_cast<const s3_u32x4*>(img);
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      const int i = u * 256 + tid;
      const int blk = i >> 6, ln = i & 63;
      const int rbl = blk / 6, rest = blk - rbl * 6;               // rest = s_l * 3 + p
      const int rb = (r0 >> 5) + rbl;
      if (rb <= last_rb) p.hnext[d][((long)rb * p.hnrb + (long)(j0 >> 4) * 3 + rest) * 64 + ln] = im4[i];
    }
  }
  MMEGO_STAMP_AT(s3_stamp_id, 3, tid == 0);
}

// The same step on HALF the hidden units per workgroup (64 rows x 16 units x 4 gates: 64 x 64 outputs), for the two-chain form of a
// layer's recurrence (blocks.lstm_steps_forward_split3: one launch per direction and timestep on two streams).  In-kernel stamps of the
// 32-unit kernel (scripts/s3_probe.hip, 2.02 GHz): prologue 5.6 k cycles (first fragments from cold L2s), product loop 15.6 k (12.3 k of
// MFMA issue), reduction + cell update + stores 7.6 k, ~2.7 us between launches -- more than half of a step is NOT the product loop,
// and with 128 KB of LDS and 444 registers per lane a CU holds one such workgroup, so nothing runs beside those phases.  This kernel
// needs half of everything (64 accumulators, a three-deep ring of 12 fragments, 64 KB of LDS: two workgroups per CU), and the two
// directions' launches -- independent dependency chains -- put one workgroup of each on a CU: one direction's launch gap, prologue
// and cell update run beside the other's product loop.  (Both directions in one 512-workgroup launch would run the pairs in lockstep.)
// NOT THE DEFAULT (blocks.split3_two_chains, a test hook): no faster inside a step.  r05 saw head_fk_loss_kernel<1> of geom.hip come out
// different in a 16-lane group of a wave in ~5 % of the runs while this stack ran beside it; r06 found the instruction -- a packed-fp32
// add / mul / fma whose op_sel takes the HIGH register of its second source pair for the LOW result receives 0.0 for it in lanes 48-63 --
// and the neighbour that does it: s3_gemm_kernel above, not this kernel (scripts/coexec_pk_probe.hip is the
// standalone reproducer).  The library is compiled without packed-fp32 instructions, and a step with bf16-MFMA kernels runs them on one chain.
// W_hh rows [16-unit block][gate][16 units]: a 32-column block holds a gate PAIR -- block 0: i | f, block 1: g | o -- of 16 units, so
// the four gates of a (row, unit) sit in lanes l and l ^ 16 of the two accumulator tiles; one exchange (the pre-activations of the
// rows the partner finishes) and every lane updates 4 cells.  The projection's columns are permuted the same way (weight
// preparation), so its tiles load straight into the accumulator layout as before.
template <int SQ, int NPROD, bool FIRST>
__global__ __launch_bounds__(256, 2) void s3_step16_kernel(S3StepP p) {
  extern __shared__ __attribute__((aligned(16))) float s3_red16[];    // [4 waves][2 mi][2 n][16 i][64 lanes] = 64 KB
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int d = blockIdx.z, H = p.H, S = H >> 4;
  const int nrb = gridDim.y, nb = gridDim.x * nrb;
  const int id = s3_xcd_order(blockIdx.y * gridDim.x + blockIdx.x, nb);
  const int jb = id / nrb, j0 = jb * 16, r0 = (id % nrb) * 64;
  const int fr = lane & 31, fh = lane >> 5;
  const int ju = fr & 15, gsel = fr >> 4;              // unit inside the block; which gate of a pair this lane's column is
  const int own_mi = w >> 1, own_i0 = 8 * (w & 1);
  const int own_r0 = own_mi * 32 + 16 * (w & 1);
  const int last_rb = (p.Bn - 1) >> 5;
  const int hb = H >> 5;

  f32x16 acc[2][2];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[mi][n][i] = 0.f;
  const s3_u32x4* ap[2];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) ap[mi] = p.hprev[d] + ((long)min((r0 >> 5) + mi, last_rb) * p.hrb + (long)w * SQ * 3) * 64 + lane;
  const s3_u32x4* wp = p.whh[d] + ((long)jb * 2 * S + w * SQ) * 192 + lane;
  const int gstride = S * 192;
  constexpr int NR = SQ >= 3 ? 3 : SQ;
  s3_u32x4 a[NR][2][3], b[NR][2][3];
#define S3_LOAD16(slot, s)                                                                  \
    {                                                                                       \
      _Pragma("unroll") for (int q = 0; q < 3; ++q) a[slot][0][q] = ap[0][((s) * 3 + q) * 64];              \
      _Pragma("unroll") for (int n = 0; n < 2; ++n)                                         \
        _Pragma("unroll") for (int q = 0; q < 3; ++q) b[slot][n][q] = wp[n * gstride + ((s) * 3 + q) * 64]; \
      _Pragma("unroll") for (int q = 0; q < 3; ++q) a[slot][1][q] = ap[1][((s) * 3 + q) * 64];              \
    }
  if (!FIRST) {
#pragma unroll
    for (int s = 0; s < NR - 1; ++s) S3_LOAD16(s, s)
  }
  __builtin_amdgcn_sched_barrier(0);
  f32x4 xp[2][2];
  float cprev[4];
#define S3_CELL_OPERANDS16()                                                                                                     \
  {                                                                                                                              \
    const int rbx = min((r0 >> 5) + own_mi, last_rb);                                                                            \
    _Pragma("unroll") for (int n = 0; n < 2; ++n)                                                                                \
      _Pragma("unroll") for (int qq = 0; qq < 2; ++qq)                                                                           \
        xp[n][qq] = (reinterpret_cast<const f32x4*>(p.xpf + ((p.mt0[d] + rbx) * (long)(8 * hb) + (p.dbase + d) * 4 * hb + jb * 2 + n) * 1024) + lane)[(2 * (w & 1) + qq) * 64]; \
    _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                                                              \
      const int row = min(r0 + own_r0 + 8 * gsel + 4 * fh + r, p.Bn - 1);                                                        \
      cprev[r] = FIRST ? 0.f : p.c[d][(long)row * H + j0 + ju];                                                                  \
    }                                                                                                                            \
  }
  if (FIRST) S3_CELL_OPERANDS16()
  float pre[2][8];
#pragma unroll
  for (int n = 0; n < 2; ++n)
#pragma unroll
    for (int ii = 0; ii < 8; ++ii) pre[n][ii] = 0.f;

  if (!FIRST) {
#pragma unroll
    for (int s = 0; s < SQ; ++s) {
      const int sn = s + NR - 1, slot = (s + NR - 1) % NR;
      if (s == (SQ >= 2 ? SQ - 2 : 0)) S3_CELL_OPERANDS16()
#pragma unroll
      for (int g = 0; g < 4; ++g) {                     // (three requests of step s + NR - 1 behind each MFMA group, in the order of need)
        const int mi = g >> 1, n = g & 1;
        acc[mi][n] = s3_mma<NPROD>(a[s % NR][mi], b[s % NR][n], acc[mi][n]);
        if (sn < SQ) {
#pragma unroll
          for (int q = 0; q < 3; ++q) {
            if (g == 0) a[slot][0][q] = ap[0][(sn * 3 + q) * 64];
            else if (g == 3) a[slot][1][q] = ap[1][(sn * 3 + q) * 64];
            else b[slot][g - 1][q] = wp[(g - 1) * gstride + (sn * 3 + q) * 64];
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#undef S3_LOAD16
#undef S3_CELL_OPERANDS16
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int i = 0; i < 16; ++i) s3_red16[(((w * 2 + mi) * 2 + n) * 16 + i) * 64 + lane] = acc[mi][n][i];
    __syncthreads();
#pragma unroll
    for (int src = 0; src < 4; ++src)
#pragma unroll
      for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int ii = 0; ii < 8; ++ii) pre[n][ii] += s3_red16[(((src * 2 + own_mi) * 2 + n) * 16 + own_i0 + ii) * 64 + lane];
    __syncthreads();                                    // (the reduction buffer becomes the piece image below)
  }
  // gate pre-activations of this lane's column (gate 2 n + gsel of unit ju) for its wave's 16 rows; this lane finishes rows ii = 4 gsel + r
  // and needs the partner column's (lane ^ 16) two gates for them: it hands over its own values for the rows the partner finishes
#pragma unroll
  for (int n = 0; n < 2; ++n)
#pragma unroll
    for (int ii = 0; ii < 8; ++ii) pre[n][ii] += xp[n][ii >> 2][ii & 3];
  float mine[2][4], theirs[2][4];
#pragma unroll
  for (int n = 0; n < 2; ++n)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      mine[n][r] = gsel ? pre[n][4 + r] : pre[n][r];
      theirs[n][r] = __shfl_xor(gsel ? pre[n][r] : pre[n][4 + r], 16, 64);
    }
  s3_bf16_t* img = reinterpret_cast<s3_bf16_t*>(s3_red16);          // [2 rb][3 p][64 lanes][8] bf16 = 6 KB
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int rl = own_r0 + 8 * gsel + 4 * fh + r;
    const int row = r0 + rl;
    // PyTorch gate order i, f, g, o = columns (n, gsel) = (0, 0), (0, 1), (1, 0), (1, 1)
    const float gi = s3_sigmoid(gsel ? theirs[0][r] : mine[0][r]);
    const float gf = s3_sigmoid(gsel ? mine[0][r] : theirs[0][r]);
    const float gg = s3_tanh(gsel ? theirs[1][r] : mine[1][r]);
    const float go = s3_sigmoid(gsel ? mine[1][r] : theirs[1][r]);
    const float cn = gf * (FIRST ? 0.f : cprev[r]) + gi * gg;
    const float hn = row < p.Bn ? go * s3_tanh(cn) : 0.f;
    if (row < p.Bn) {
      p.c[d][(long)row * H + j0 + ju] = cn;
      if (p.hout[d]) p.hout[d][(long)row * p.hos + j0 + ju] = hn;
    }
    unsigned q1, q2, q3;
    s3_split(hn, q1, q2, q3);
    const int o = (((rl >> 5) * 3) * 64 + (rl & 31) + 32 * ((ju >> 3) & 1)) * 8 + (ju & 7);
    img[o] = (s3_bf16_t)q1;
    img[o + 512] = (s3_bf16_t)q2;
    img[o + 1024] = (s3_bf16_t)q3;
  }
  __syncthreads();
  {
    // 384 16-byte pieces: image index i = (rb_l * 3 + p) * 64 + lane
    const s3_u32x4* im4 = reinterpret_cast<const s3_u32x4*>(img);
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int i = u * 256 + tid;
      const int blk = i >> 6, ln = i & 63;
      const int rbl = blk / 3, pc = blk - rbl * 3;
      const int rb = (r0 >> 5) + rbl;
      if (i < 384 && rb <= last_rb) p.hnext[d][((long)rb * p.hnrb + (long)jb * 3 + pc) * 64 + ln] = im4[i];
    }
  }
}

// One timestep of both (ndir = 2) or one direction of a BiLSTM layer on piece products.  hprev / hnext: h_{t-1} / h_t as sfrag pieces
// inside a buffer whose row blocks are hrb / hnrb 1-KB blocks apart (a [Bp x H] window of the layer output [T Bp / 32][2H / 16][3] KB:
// pointer = buffer + ((t Bp / 32) (2H / 16) + d H / 16) 3 KB, hrb = (2H / 16) 3); whh: mmego_split3_cvt of W_hh with rows reordered
// [hidden block][gate][32 units]; xpf: tile-major projections (mmego_split3_gemm's Cf), mt0_d = t_d Bp / 32; hout (optional): fp32
// rows with stride hos; c [Bn][H]; dbase: the direction slot 0 stands for (ndir = 1 launches
// of the reverse direction pass 1: it selects the projection's columns).  Bn <= 2048 rows, H = 256 / 512 / 1024.
extern "C" int mmego_split3_step(void* stream, int ndir, int Bn, int H, int first, const unsigned short* hprev0, const unsigned short* hprev1,
                                 long hrb, const unsigned short* whh0, const unsigned short* whh1, const float* xpf, long mt0_0, long mt0_1,
                                 float* hout0, float* hout1, long hos, unsigned short* hnext0, unsigned short* hnext1, long hnrb,
                                 float* c0, float* c1, int nprod, int dbase) {
  MMEGO_REQUIRE(dbase >= 0 && dbase + ndir <= 2);
  MMEGO_REQUIRE((ndir == 1 || ndir == 2) && Bn > 0 && Bn <= 2048 && (H == 256 || H == 512 || H == 1024) && (nprod == 6 || nprod == 9));
  MMEGO_REQUIRE(first || (hprev0 && (ndir == 1 || hprev1) && ((((uintptr_t)hprev0) | ((uintptr_t)hprev1)) & 15) == 0));
  MMEGO_REQUIRE(whh0 && (ndir == 1 || whh1) && ((((uintptr_t)whh0) | ((uintptr_t)whh1)) & 15) == 0);
  MMEGO_REQUIRE(hnext0 && (ndir == 1 || hnext1) && ((((uintptr_t)hnext0) | ((uintptr_t)hnext1)) & 15) == 0 && hnext0 != hprev0);
  MMEGO_REQUIRE(xpf && (((uintptr_t)xpf) & 15) == 0 && mt0_0 >= 0 && mt0_1 >= 0 && c0 && (ndir == 1 || c1) && hrb > 0 && hnrb > 0);
  S3StepP p;
  p.hprev[0] = reinterpret_cast<const s3_u32x4*>(hprev0); p.hprev[1] = reinterpret_cast<const s3_u32x4*>(hprev1); p.hrb = hrb;
  p.whh[0] = reinterpret_cast<const s3_u32x4*>(whh0); p.whh[1] = reinterpret_cast<const s3_u32x4*>(whh1);
  p.xpf = xpf; p.mt0[0] = mt0_0; p.mt0[1] = mt0_1;
  p.hout[0] = hout0; p.hout[1] = hout1; p.hos = hos;
  p.hnext[0] = reinterpret_cast<s3_u32x4*>(hnext0); p.hnext[1] = reinterpret_cast<s3_u32x4*>(hnext1); p.hnrb = hnrb;
  p.c[0] = c0; p.c[1] = c1;
  p.Bn = Bn; p.H = H; p.first = first; p.dbase = dbase; p.S = H / 16;
  const int lds = 4 * 2 * 4 * 16 * 64 * (int)sizeof(float);
  dim3 grid(H / 32, cdiv(Bn, 64), ndir);
  hipStream_t st = (hipStream_t)stream;
  const int sq = H / 64;
#define S3_STEP_LAUNCH(SQ_, NP_, F_)                                                                                        \
  {                                                                                                                         \
    static bool attr_set = false;                                                                                           \
    if (!attr_set) {                                                                                                        \
      hipError_t e = hipFuncSetAttribute((const void*)s3_step_kernel<SQ_, NP_, F_>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); \
      if (e != hipSuccess) return (int)e;                                                                                   \
      attr_set = true;                                                                                                      \
    }                                                                                                                       \
    s3_step_kernel<SQ_, NP_, F_><<<grid, 256, lds, st>>>(p);                                                               \
  }
#define S3_STEP_NP(SQ_, F_)                 \
  {                                         \
    if (nprod == 6) S3_STEP_LAUNCH(SQ_, 6, F_) \
    else S3_STEP_LAUNCH(SQ_, 9, F_)         \
  }
#define S3_STEP_F(SQ_)              \
  {                                 \
    if (first) S3_STEP_NP(SQ_, 1) \
    else S3_STEP_NP(SQ_, 0)     \
  }
  if (sq == 8) S3_STEP_F(8)
  else if (sq == 4) S3_STEP_F(4)
  else S3_STEP_F(16)
#undef S3_STEP_F
#undef S3_STEP_NP
#undef S3_STEP_LAUNCH
  MMEGO_LAUNCH_CHECK();
  return MMEGO_OK;
}

// Projection for few rows: C[m][d 4H + n] = sum_k A[m][k] W_d[n][k] + bias[d 4H + n] for both directions d (grid z), m < M <= 2048, n < 4H,
// K = 256 / 512 / 1024 (the k quarters of a workgroup's four waves are 4 / 8 / 16 steps).  A: sfrag pieces [ceil(M/32)][K/16][3] KB (rows
// in order); W0 / W1: mmego_split3_cvt of each direction's weight [4H][K] with rows reordered [32-unit block][gate][32 units]; C row-major,
// row stride ldc >= 8H.  (s3_step_kernel<.., 2>: 64 x 128 outputs per workgroup, K split over its waves, three steps of requests in flight.)
extern "C" int mmego_split3_proj(void* stream, const unsigned short* A, const unsigned short* W0, const unsigned short* W1, const float* bias,
                                 float* C, long ldc, int M, int H, int K, int nprod) {
  MMEGO_REQUIRE(A && W0 && W1 && C && M > 0 && M <= 2048 && H > 0 && H % 32 == 0 && (K == 256 || K == 512 || K == 1024) && ldc >= 8 * H);
  MMEGO_REQUIRE((nprod == 6 || nprod == 9) && ((((uintptr_t)A) | ((uintptr_t)W0) | ((uintptr_t)W1)) & 15) == 0);
  S3StepP p = {};
  p.hprev[0] = p.hprev[1] = reinterpret_cast<const s3_u32x4*>(A); p.hrb = (long)(K / 16) * 3;
  p.whh[0] = reinterpret_cast<const s3_u32x4*>(W0); p.whh[1] = reinterpret_cast<const s3_u32x4*>(W1);
  p.xpf = bias;
  p.hout[0] = C; p.hout[1] = C + 4 * H; p.hos = ldc;
  p.Bn = M; p.H = H; p.first = 0; p.dbase = 0; p.S = K / 16;
  const int lds = 4 * 2 * 4 * 16 * 64 * (int)sizeof(float);
  dim3 grid(H / 32, cdiv(M, 64), 2);
  hipStream_t st = (hipStream_t)stream;
  const int sq = K / 64;
#define S3_PROJ_LAUNCH(SQ_, NP_)                                                                                            \
  {                                                                                                                         \
    static bool attr_set = false;                                                                                           \
    if (!attr_set) {                                                                                                        \
      hipError_t e = hipFuncSetAttribute((const void*)s3_step_kernel<SQ_, NP_, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); \
      if (e != hipSuccess) return (int)e;                                                                                   \
      attr_set = true;                                                                                                      \
    }                                                                                                                       \
    s3_step_kernel<SQ_, NP_, 2><<<grid, 256, lds, st>>>(p);                                                                 \
  }
  if (sq == 16) { if (nprod == 6) S3_PROJ_LAUNCH(16, 6) else S3_PROJ_LAUNCH(16, 9) }
  else if (sq == 8) { if (nprod == 6) S3_PROJ_LAUNCH(8, 6) else S3_PROJ_LAUNCH(8, 9) }
  else { if (nprod == 6) S3_PROJ_LAUNCH(4, 6) else S3_PROJ_LAUNCH(4, 9) }
#undef S3_PROJ_LAUNCH
  MMEGO_LAUNCH_CHECK();
  return MMEGO_OK;
}

// The same timestep on 16-unit workgroups (s3_step16_kernel): for single-direction launches that run as two chains.  Same arguments;
// whh: W_hh rows reordered [16-unit block][gate][16 units], and the projection's columns (W_ih rows, bias) in that order too.
extern "C" int mmego_split3_step16(void* stream, int ndir, int Bn, int H, int first, const unsigned short* hprev0, const unsigned short* hprev1,
                                   long hrb, const unsigned short* whh0, const unsigned short* whh1, const float* xpf, long mt0_0, long mt0_1,
                                   float* hout0, float* hout1, long hos, unsigned short* hnext0, unsigned short* hnext1, long hnrb,
                                   float* c0, float* c1, int nprod, int dbase) {
  MMEGO_REQUIRE(dbase >= 0 && dbase + ndir <= 2);
  MMEGO_REQUIRE((ndir == 1 || ndir == 2) && Bn > 0 && Bn <= 2048 && (H == 256 || H == 512 || H == 1024) && (nprod == 6 || nprod == 9));
  MMEGO_REQUIRE(first || (hprev0 && (ndir == 1 || hprev1) && ((((uintptr_t)hprev0) | ((uintptr_t)hprev1)) & 15) == 0));
  MMEGO_REQUIRE(whh0 && (ndir == 1 || whh1) && ((((uintptr_t)whh0) | ((uintptr_t)whh1)) & 15) == 0);
  MMEGO_REQUIRE(hnext0 && (ndir == 1 || hnext1) && ((((uintptr_t)hnext0) | ((uintptr_t)hnext1)) & 15) == 0 && hnext0 != hprev0);
  MMEGO_REQUIRE(xpf && (((uintptr_t)xpf) & 15) == 0 && mt0_0 >= 0 && mt0_1 >= 0 && c0 && (ndir == 1 || c1) && hrb > 0 && hnrb > 0);
  S3StepP p;
  p.hprev[0] = reinterpret_cast<const s3_u32x4*>(hprev0); p.hprev[1] = reinterpret_cast<const s3_u32x4*>(hprev1); p.hrb = hrb;
  p.whh[0] = reinterpret_cast<const s3_u32x4*>(whh0); p.whh[1] = reinterpret_cast<const s3_u32x4*>(whh1);
  p.xpf = xpf; p.mt0[0] = mt0_0; p.mt0[1] = mt0_1;
  p.hout[0] = hout0; p.hout[1] = hout1; p.hos = hos;
  p.hnext[0] = reinterpret_cast<s3_u32x4*>(hnext0); p.hnext[1] = reinterpret_cast<s3_u32x4*>(hnext1); p.hnrb = hnrb;
  p.c[0] = c0; p.c[1] = c1;
  p.Bn = Bn; p.H = H; p.first = first; p.dbase = dbase; p.S = H / 16;
  const int lds = 4 * 2 * 2 * 16 * 64 * (int)sizeof(float);
  dim3 grid(H / 16, cdiv(Bn, 64), ndir);
  hipStream_t st = (hipStream_t)stream;
  const int sq = H / 64;
#define S3_STEP_LAUNCH(SQ_, NP_, F_)                                                                                        \
  {                                                                                                                         \
    static bool attr_set = false;                                                                                           \
    if (!attr_set) {                                                                                                        \
      hipError_t e = hipFuncSetAttribute((const void*)s3_step16_kernel<SQ_, NP_, F_>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); \
      if (e != hipSuccess) return (int)e;                                                                                   \
      attr_set = true;                                                                                                      \
    }                                                                                                                       \
    s3_step16_kernel<SQ_, NP_, F_><<<grid, 256, lds, st>>>(p);                                                             \
  }
#define S3_STEP_NP(SQ_, F_)                 \
  {                                         \
    if (nprod == 6) S3_STEP_LAUNCH(SQ_, 6, F_) \
    else S3_STEP_LAUNCH(SQ_, 9, F_)         \
  }
#define S3_STEP_F(SQ_)              \
  {                                 \
    if (first) S3_STEP_NP(SQ_, true) \
    else S3_STEP_NP(SQ_, false)     \
  }
  if (sq == 8) S3_STEP_F(8)
  else if (sq == 4) S3_STEP_F(4)
  else S3_STEP_F(16)
#undef S3_STEP_F
#undef S3_STEP_NP
#undef S3_STEP_LAUNCH
  MMEGO_LAUNCH_CHECK();
  return MMEGO_OK;
}
