// fp32 products on the bf16 matrix pipe ("bf16x9"): every fp32 operand is split EXACTLY into three bf16 pieces
//     a = a_h + a_m + a_l      (8 + 8 + 8 significant bits; bf16 has fp32's exponent range)
// and a.b is the sum of the nine piece products, each of which is exact in fp32 (8 x 8 bits), accumulated in fp32 by
// v_mfma_f32_32x32x16_bf16.  Nothing is rounded away: the result differs from the fp32 MFMA's only by the order of the
// fp32 accumulation roundings (9K/16 accumulator updates per output against K/2 for v_mfma_f32_32x32x2_f32) -- measured
// against fp64 it is as close as the native instruction (tests/test_x9_gpu.py).  The bf16 pipe runs 16x the fp32 MFMA rate,
// so nine passes cost 9/16 of the native time: peak 2.5 PFLOP/s / 9 = 278 TFLOP/s fp32-equivalent against 157.3.
//
// Operand format "split-interleaved": [row][K/8][piece h,m,l][8] bf16 (6 B per element): the three pieces of 8 consecutive k
// are adjacent, so a 32-k chunk of a row is 192 contiguous bytes and one LDS row serves all nine products of a k-step.
#include "common.h"

typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef unsigned short bf16_t;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// truncation split: h = top 16 bits of a, m = top 16 bits of (a - h), l = a - h - m (exactly representable: 8 bits left)
__device__ __forceinline__ void split3(float a, unsigned& h, unsigned& m, unsigned& l) {
  const unsigned ua = __float_as_uint(a);
  h = ua >> 16;
  const float r1 = a - __uint_as_float(ua & 0xffff0000u);
  const unsigned u1 = __float_as_uint(r1);
  m = u1 >> 16;
  const float r2 = r1 - __uint_as_float(u1 & 0xffff0000u);
  l = __float_as_uint(r2) >> 16;
}

// X fp32 [rows, K] (row stride ldx) -> split-interleaved [rows][K/8][3][8]
__global__ __launch_bounds__(256) void x9_split_kernel(const float* __restrict__ X, long ldx, long rows, int k8,
                                                        bf16_t* __restrict__ Y) {
  const long total = rows * k8;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long r = i / k8;
    const int g = (int)(i - r * k8);
    const f32x4 v0 = *reinterpret_cast<const f32x4*>(X + r * ldx + g * 8);
    const f32x4 v1 = *reinterpret_cast<const f32x4*>(X + r * ldx + g * 8 + 4);
    unsigned h[8], m[8], l[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      split3(v0[e], h[e], m[e], l[e]);
      split3(v1[e], h[4 + e], m[4 + e], l[4 + e]);
    }
    u32x4 oh, om, ol;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      oh[e] = h[2 * e] | (h[2 * e + 1] << 16);
      om[e] = m[2 * e] | (m[2 * e + 1] << 16);
      ol[e] = l[2 * e] | (l[2 * e + 1] << 16);
    }
    u32x4* o = reinterpret_cast<u32x4*>(Y + (r * k8 + g) * 24);
    o[0] = oh; o[1] = om; o[2] = ol;
  }
}

extern "C" int mmego_x9_split(void* stream, const float* X, long ldx, long rows, int K, unsigned short* Y) {
  MMEGO_REQUIRE(rows > 0 && K > 0 && K % 8 == 0 && ldx % 4 == 0);
  MMEGO_REQUIRE((((uintptr_t)X) & 15) == 0 && (((uintptr_t)Y) & 15) == 0);
  const long total = rows * (K / 8);
  const int grid = (int)(total / 256 + 1 < 8192 ? total / 256 + 1 : 8192);
  x9_split_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(X, ldx, rows, K / 8, Y);
  MMEGO_LAUNCH_CHECK();
  return MMEGO_OK;
}

__device__ __forceinline__ int x9_xcd_order(int id, int n) { return (n & 7) == 0 ? (id & 7) * (n >> 3) + (id >> 3) : id; }

struct X9GemmP {
  const bf16_t* A;     // split-interleaved [M][K/8][3][8]
  const bf16_t* W;     // split-interleaved [N][K/8][3][8]
  float* C; long ldc;  // row-major output (may be null)
  float* Cf;           // tile-major output (may be null), see mmego_gemm_bf16
  const float* bias;
  int M, N, K, relu, tiles_m, tiles_n;
};

#define X9_BK 32                 // k per staged chunk: 4 groups of 8 k x 3 pieces = 192 B per row
#define X9_ROWB 208              // LDS row stride in bytes (192 + 16: 16 consecutive rows' 16-B reads on distinct bank quads)

// 256 threads, tile 128 x 128, wave w owns the 64x64 block (w/2, w%2).  Per 16-k step a wave reads 2 x 3 + 2 x 3 fragments
// and issues 2 x 2 x 9 MFMAs: 12 LDS reads per 36 MFMAs, the loop is matrix-pipe bound.
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void x9_gemm_nt_kernel(X9GemmP p) {
  __shared__ __attribute__((aligned(16))) unsigned char As[128 * X9_ROWB];
  __shared__ __attribute__((aligned(16))) unsigned char Bs[128 * X9_ROWB];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int wm = w >> 1, wn = w & 1;
  const int id = x9_xcd_order(blockIdx.x, (int)gridDim.x);
  const int GM = 16;
  const int per_group = GM * p.tiles_n;
  const int group = id / per_group, in_group = id - group * per_group;
  const int gm = min(GM, p.tiles_m - group * GM);
  const int tm = group * GM + in_group % gm, tn = in_group / gm;
  const int m0 = tm * 128, n0 = tn * 128;

  // a chunk of a tile is 128 rows x 12 pieces of 16 B: 6 pieces per thread and operand
  const long rowbytes = (long)p.K * 6;
  const char* ag[6];
  const char* wg[6];
  int lo[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const int q = tid + 256 * i, row = q / 12, seg = q - row * 12;
    ag[i] = reinterpret_cast<const char*>(p.A) + (long)min(m0 + row, p.M - 1) * rowbytes + seg * 16;
    wg[i] = reinterpret_cast<const char*>(p.W) + (long)min(n0 + row, p.N - 1) * rowbytes + seg * 16;
    lo[i] = row * X9_ROWB + seg * 16;
  }
  // two chunks in flight: chunk c+2 leaves for its registers when chunk c starts to be multiplied (a 32-k chunk is ~2300
  // matrix-pipe cycles per wave, about one global-load latency: one chunk of lookahead stalled every chunk)
  u32x4 ra0[6], rb0[6], ra1[6], rb1[6];
  const int nchunk = p.K / X9_BK;
#define X9_LOAD(RA, RB, c_)                                                                     \
  do {                                                                                          \
    const long off_ = (long)(c_) * (X9_BK * 6);                                                 \
    _Pragma("unroll") for (int i = 0; i < 6; ++i) {                                             \
      RA[i] = *reinterpret_cast<const u32x4*>(ag[i] + off_);                                    \
      RB[i] = *reinterpret_cast<const u32x4*>(wg[i] + off_);                                    \
    }                                                                                           \
  } while (0)
#define X9_STAGE(RA, RB)                                                                        \
  do {                                                                                          \
    __syncthreads();                                                                            \
    _Pragma("unroll") for (int i = 0; i < 6; ++i) {                                             \
      *reinterpret_cast<u32x4*>(As + lo[i]) = RA[i];                                            \
      *reinterpret_cast<u32x4*>(Bs + lo[i]) = RB[i];                                            \
    }                                                                                           \
    __syncthreads();                                                                            \
  } while (0)
  X9_LOAD(ra0, rb0, 0);
  if (nchunk > 1) X9_LOAD(ra1, rb1, 1);
  f32x16 acc[2][2];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[mi][ni][i] = 0.f;

  const int fr = lane & 31, fh = lane >> 5;
  // nine piece products per output tile, smallest first (piece 0 = h, 1 = m, 2 = l); the four output tiles rotate so that
  // consecutive MFMAs never depend on each other
#define X9_COMPUTE()                                                                            \
  do {                                                                                          \
    _Pragma("unroll") for (int s = 0; s < 2; ++s) {                                             \
      bf16x8 a[2][3], b[2][3];                                                                  \
      const int goff = (2 * s + fh) * 48;                                                       \
      _Pragma("unroll") for (int mi = 0; mi < 2; ++mi)                                          \
        _Pragma("unroll") for (int pc = 0; pc < 3; ++pc)                                        \
          a[mi][pc] = *reinterpret_cast<const bf16x8*>(As + (wm * 64 + mi * 32 + fr) * X9_ROWB + goff + pc * 16); \
      _Pragma("unroll") for (int ni = 0; ni < 2; ++ni)                                          \
        _Pragma("unroll") for (int pc = 0; pc < 3; ++pc)                                        \
          b[ni][pc] = *reinterpret_cast<const bf16x8*>(Bs + (wn * 64 + ni * 32 + fr) * X9_ROWB + goff + pc * 16); \
      constexpr int PA[9] = {2, 2, 1, 2, 0, 1, 1, 0, 0};                                        \
      constexpr int PB[9] = {2, 1, 2, 0, 2, 1, 0, 1, 0};                                        \
      _Pragma("unroll") for (int t = 0; t < 9; ++t)                                             \
        _Pragma("unroll") for (int mi = 0; mi < 2; ++mi)                                        \
          _Pragma("unroll") for (int ni = 0; ni < 2; ++ni)                                      \
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi][PA[t]], b[ni][PB[t]], acc[mi][ni], 0, 0, 0); \
    }                                                                                           \
  } while (0)
  for (int c = 0; c < nchunk; c += 2) {
    X9_STAGE(ra0, rb0);
    if (c + 2 < nchunk) X9_LOAD(ra0, rb0, c + 2);
    X9_COMPUTE();
    if (c + 1 < nchunk) {
      X9_STAGE(ra1, rb1);
      if (c + 3 < nchunk) X9_LOAD(ra1, rb1, c + 3);
      X9_COMPUTE();
    }
  }
#undef X9_LOAD
#undef X9_STAGE
#undef X9_COMPUTE
  if (p.C) {
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      const int col = n0 + wn * 64 + ni * 32 + fr;
      if (col >= p.N) continue;
      const float bv = p.bias ? p.bias[col] : 0.f;
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int row = m0 + wm * 64 + mi * 32 + 8 * (i >> 2) + 4 * fh + (i & 3);
          if (row < p.M) {
            float v = acc[mi][ni][i] + bv;
            if (p.relu) v = fmaxf(v, 0.f);
            p.C[(long)row * p.ldc + col] = v;
          }
        }
    }
  }
  if (p.Cf) {
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

extern "C" int mmego_x9_gemm(void* stream, const unsigned short* A, const unsigned short* W, float* C, long ldc, float* Cf,
                             const float* bias, int M, int N, int K, int relu) {
  MMEGO_REQUIRE(M > 0 && N > 0 && K > 0 && K % X9_BK == 0 && (C || Cf));
  MMEGO_REQUIRE((((uintptr_t)A) & 15) == 0 && (((uintptr_t)W) & 15) == 0);
  MMEGO_REQUIRE(!Cf || (M % 32 == 0 && N % 32 == 0 && (((uintptr_t)Cf) & 15) == 0));
  X9GemmP p;
  p.A = A; p.W = W; p.C = C; p.ldc = ldc; p.Cf = Cf; p.bias = bias; p.M = M; p.N = N; p.K = K; p.relu = relu;
  p.tiles_m = cdiv(M, 128); p.tiles_n = cdiv(N, 128);
  const long tiles = (long)p.tiles_m * p.tiles_n;
  MMEGO_REQUIRE(tiles < (1L << 30));
  x9_gemm_nt_kernel<<<(int)tiles, 256, 0, (hipStream_t)stream>>>(p);
  MMEGO_LAUNCH_CHECK();
  return MMEGO_OK;
}

// ---- one BiLSTM timestep with split products ----------------------------------------------------------------------------
// The recurrent product of lstm_step_bf16_direct_kernel (bf16.hip) with nine piece products per output: same workgroup shape
// (64 rows x 32 hidden units x 4 gates, wave w = k quarter w of the whole tile, partial tiles meet in LDS, all four waves run
// the cell update), same fragment-major operands with a piece index added:
//     frag3(X, rb, s, p)[l][e] = piece_p( X[32*rb + l%32][16*s + 8*(l/32) + e] ),   blocks ordered [rb][s][p], 1 KB each.
// The fragments of two 16-k steps (36 coalesced 1-KB reads) are requested while the previous two steps' 144 MFMAs run.
struct X9StepP {
  const bf16_t* hprev[2];               // h_{t-1}, frag3, ceil(Bn/32) row blocks
  const bf16_t* whh[2];                 // W_hh, frag3 with rows [hidden block][gate][32 units]
  const float* xpf; long mt0[2];        // tile-major projection (see mmego_gemm_bf16), row tile of each direction's timestep
  float* hout[2]; long hos;             // h_t fp32 row-major (may be null)
  bf16_t* hsplit[2]; long hss;          // h_t split-interleaved rows [.][2H/8][3][8] (next layer's projection operand; may be null);
                                        // hss = row stride in ELEMENTS of the fp32 matrix (2H), direction d at column d*H
  bf16_t* hfrag[2];                     // h_t frag3 (next step's operand)
  float* c[2];
  int Bn, H, first;
};

__device__ __forceinline__ float x9_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float x9_tanh(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * x)); }

__global__ __launch_bounds__(256) void lstm_step_x9_kernel(X9StepP p) {
  extern __shared__ __attribute__((aligned(16))) float red[];      // [4 waves][2 mi][4 n][16 i][64 lanes] = 128 KB
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int d = blockIdx.z, H = p.H, S = H >> 4, SQ = S >> 2;
  const int nrb = gridDim.y, nb = gridDim.x * nrb;
  const int id = x9_xcd_order(blockIdx.y * gridDim.x + blockIdx.x, nb);
  const int jb = id / nrb, j0 = jb * 32, r0 = (id % nrb) * 64;
  const int fr = lane & 31, fh = lane >> 5;
  const int j = j0 + fr;
  const int own_mi = w >> 1, own_i0 = 8 * (w & 1);
  const int own_r0 = own_mi * 32 + 16 * (w & 1);
  const int hb = H >> 5;

  f32x4 xp[4][2];
  float cprev[8];
  {
    const int rb = min((r0 >> 5) + own_mi, (p.Bn - 1) >> 5);
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      const f32x4* t = reinterpret_cast<const f32x4*>(p.xpf + ((p.mt0[d] + rb) * (long)(8 * hb) + (d * 4 + n) * hb + jb) * 1024) + lane;
#pragma unroll
      for (int qq = 0; qq < 2; ++qq) xp[n][qq] = t[(2 * (w & 1) + qq) * 64];
    }
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
      ap[mi] = reinterpret_cast<const u32x4*>(p.hprev[d]) + ((long)min((r0 >> 5) + mi, last_rb) * S + w * SQ) * 192 + lane;
    const u32x4* wp = reinterpret_cast<const u32x4*>(p.whh[d]) + ((long)jb * 4 * S + w * SQ) * 192 + lane;
    const int gstride = S * 192;            // between the gates' fragment runs (16-B units)
    u32x4 a0[2][3], b0[4][3], a1[2][3], b1[4][3];
#define X9S_LOAD(A_, B_, s_)                                                                     \
  do {                                                                                          \
    _Pragma("unroll") for (int mi = 0; mi < 2; ++mi)                                            \
      _Pragma("unroll") for (int pc = 0; pc < 3; ++pc) A_[mi][pc] = ap[mi][((s_) * 3 + pc) * 64]; \
    _Pragma("unroll") for (int n = 0; n < 4; ++n)                                               \
      _Pragma("unroll") for (int pc = 0; pc < 3; ++pc) B_[n][pc] = wp[n * gstride + ((s_) * 3 + pc) * 64]; \
  } while (0)
#define X9S_MMA(A_, B_)                                                                         \
  do {                                                                                          \
    constexpr int PA[9] = {2, 2, 1, 2, 0, 1, 1, 0, 0};                                          \
    constexpr int PB[9] = {2, 1, 2, 0, 2, 1, 0, 1, 0};                                          \
    _Pragma("unroll") for (int t = 0; t < 9; ++t)                                               \
      _Pragma("unroll") for (int mi = 0; mi < 2; ++mi)                                          \
        _Pragma("unroll") for (int n = 0; n < 4; ++n)                                           \
          acc[mi][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A_[mi][PA[t]]), \
                                                               __builtin_bit_cast(bf16x8, B_[n][PB[t]]), acc[mi][n], 0, 0, 0); \
  } while (0)
    // one 16-k step = 18 fragment loads and 72 MFMAs (2300 matrix-pipe cycles); the next step's loads run under them
    X9S_LOAD(a0, b0, 0);
    for (int s = 0; s < SQ; s += 2) {
      X9S_LOAD(a1, b1, s + 1);
      __builtin_amdgcn_sched_barrier(0);
      X9S_MMA(a0, b0);
      if (s + 2 < SQ) X9S_LOAD(a0, b0, s + 2);
      __builtin_amdgcn_sched_barrier(0);
      X9S_MMA(a1, b1);
    }
#undef X9S_LOAD
#undef X9S_MMA
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
    if (row < p.Bn) {
      const float gi = x9_sigmoid(pre[0][ii] + xp[0][ii >> 2][ii & 3]), gf = x9_sigmoid(pre[1][ii] + xp[1][ii >> 2][ii & 3]);
      const float gg = x9_tanh(pre[2][ii] + xp[2][ii >> 2][ii & 3]), go = x9_sigmoid(pre[3][ii] + xp[3][ii >> 2][ii & 3]);
      const float cn = gf * cprev[ii] + gi * gg;
      const float hn = go * x9_tanh(cn);
      p.c[d][(long)row * H + j] = cn;
      if (p.hout[d]) p.hout[d][(long)row * p.hos + j] = hn;
      unsigned ph[3];
      split3(hn, ph[0], ph[1], ph[2]);
      bf16_t* f = p.hfrag[d] + ((((long)(row >> 5) * S + (j >> 4)) * 3) * 64 + ((j >> 3) & 1) * 32 + (row & 31)) * 8 + (j & 7);
#pragma unroll
      for (int pc = 0; pc < 3; ++pc) f[pc * 512] = (bf16_t)ph[pc];
      if (p.hsplit[d]) {
        bf16_t* o = p.hsplit[d] + ((long)row * (p.hss >> 3) + (j >> 3)) * 24 + (j & 7);
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) o[pc * 8] = (bf16_t)ph[pc];
      }
    }
  }
}

extern "C" int mmego_lstm_step_x9(void* stream, int ndir, int Bn, int H, int first, const unsigned short* hprev0,
                                  const unsigned short* hprev1, const unsigned short* whh0, const unsigned short* whh1,
                                  const float* xpf, long mt0_0, long mt0_1, float* hout0, float* hout1, long hos,
                                  unsigned short* hsplit0, unsigned short* hsplit1, long hss, unsigned short* hfrag0,
                                  unsigned short* hfrag1, float* c0, float* c1) {
  MMEGO_REQUIRE((ndir == 1 || ndir == 2) && Bn > 0 && H > 0 && H % 128 == 0);
  MMEGO_REQUIRE(first || (hprev0 && whh0 && (ndir == 1 || (hprev1 && whh1))));
  MMEGO_REQUIRE(xpf && hfrag0 && c0 && (ndir == 1 || (hfrag1 && c1)) && hfrag0 != hprev0 && hss % 8 == 0);
  X9StepP p;
  p.hprev[0] = hprev0; p.hprev[1] = hprev1; p.whh[0] = whh0; p.whh[1] = whh1;
  p.xpf = xpf; p.mt0[0] = mt0_0; p.mt0[1] = mt0_1;
  p.hout[0] = hout0; p.hout[1] = hout1; p.hos = hos;
  p.hsplit[0] = hsplit0; p.hsplit[1] = hsplit1; p.hss = hss;
  p.hfrag[0] = hfrag0; p.hfrag[1] = hfrag1;
  p.c[0] = c0; p.c[1] = c1;
  p.Bn = Bn; p.H = H; p.first = first;
  const int lds = 4 * 2 * 4 * 16 * 64 * (int)sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)lstm_step_x9_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  dim3 grid(H / 32, cdiv(Bn, 64), ndir);
  lstm_step_x9_kernel<<<grid, 256, lds, (hipStream_t)stream>>>(p);
  MMEGO_LAUNCH_CHECK();
  return MMEGO_OK;
}
