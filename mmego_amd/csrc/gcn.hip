// ST-GCN layer pieces of Lower_Net's KeyEncoder (reference Net/GCN.py:55-64 ConvTemporalGraphical, :108-122 the 9x1 temporal
// convolution of st_gcn.tcn) on channels-last rows (b, t, v):
//   graph_mix    einsum('nkctv,kvw->nctw', x, A * edge_importance): y[f][w][c] = sum_k sum_v (A.imp)[k][v][w] z[f][v][k*C + c],
//                and its input gradient dz[f][v][k*C + c] = sum_w (A.imp)[k][v][w] dy[f][w][c].  A . edge_importance is formed
//                in LDS by the kernel itself (no `mul` launch, no K batched-product launches with a broadcast 15 x 15 operand),
//                a frame's z tile is staged in LDS once and every output is a 15-term dot product from there.
//   tconv        the 9x1 temporal convolution (padding 4) as an IMPLICIT GEMM: out[r][co] = b[co] + sum_tap sum_ci
//                act(in[r + (tap-4) V][ci]) W[co][ci][tap], rows (b, t, v), taps that leave the sequence contribute zero.  The
//                unfolded operand (im2col_t: 9x the activation, 9.4 GB per config-5 forward) never exists: per tap the A tile is
//                the SAME 64 rows shifted by (tap-4) V rows.  The preceding BatchNorm + ReLU (st_gcn.tcn[0..1]) is applied while
//                the tile is loaded.  The input gradient of the convolution is the same kernel on flipped, transposed weights
//                (tconv_pack mode 1), so col2im_t disappears as well.
//                The weight element (tap, n, k) is addressed through three strides; the conv weight [co][ci][tap] as it is would
//                work (forward (1, Ci*taps, taps), input gradient from the LAST tap with (-1, taps, Ci*taps)) but puts the 64 lanes
//                of a tile load 36 B / 4.6 KB apart (measured: 61 us per launch at the training shape), so the weights are
//                re-packed k-contiguous first:
//   tconv_pack   W[co][ci][tap] -> Wp[tap][co][ci] (mode 0) or Wp[tap][ci][co] with the taps reversed (mode 1); mode 2 writes both,
//                one behind the other (a training step packs once in its forward pass and uses the second half in backward).
// 64 x 64 output tiles, v_mfma_f32_32x32x2_f32, operands in LDS with a 65-float row stride, next chunk prefetched in registers.
#include "common.h"

#define GC_S 66          // even row stride: fragments are read as aligned float2 (lanes r = 0..31 hit 64 distinct banks)

// stats (optional, forward): per-frame BatchNorm partials of the output, stats[(c*F + f)*3 + {0,1,2}] = (V, mean, M2) of channel c
// over the frame's V rows -- the record format of colstats_kernel (bn.hip) with one "block" per frame, so that the batch
// statistics of the BatchNorm behind the einsum (st_gcn.tcn[0]) need no pass of their own over y.
__global__ __launch_bounds__(256) void graph_mix_kernel(const float* __restrict__ X, const float* __restrict__ A,
                                                        const float* __restrict__ imp, float* __restrict__ Y, long F, int V,
                                                        int Kk, int C, int backward, float* __restrict__ stats, long ldx) {
  extern __shared__ float sm[];
  // forward : X = z [F][V][Kk*C] -> Y = y [F][V][C];   backward: X = dy [F][V][C] -> Y = dz [F][V][Kk*C]
  const int Cin = backward ? C : Kk * C, Cout = backward ? Kk * C : C;
  float* As = sm;                          // [Kk][V][V]  A . importance
  float* Xs = sm + Kk * V * V;             // [V][Cin + 1]
  float* Ys = Xs + V * (Cin + 1);          // [V][Cout + 1]  (only with stats)
  for (int i = threadIdx.x; i < Kk * V * V; i += blockDim.x) As[i] = A[i] * imp[i];
  for (long f = blockIdx.x; f < F; f += gridDim.x) {
    __syncthreads();
    const float* xf = X + f * V * ldx;                     // (ldx: row stride of X, >= Cin -- X may be a column slice)
    for (int i = threadIdx.x; i < V * Cin; i += blockDim.x) Xs[(i / Cin) * (Cin + 1) + (i % Cin)] = xf[(long)(i / Cin) * ldx + (i % Cin)];
    __syncthreads();
    float* yf = Y + f * V * Cout;
    for (int i = threadIdx.x; i < V * Cout; i += blockDim.x) {
      const int row = i / Cout, col = i - row * Cout;
      float acc = 0.f;
      if (!backward) {                     // row = w, col = c
        for (int k = 0; k < Kk; ++k)
          for (int v = 0; v < V; ++v) acc += As[(k * V + v) * V + row] * Xs[v * (Cin + 1) + k * C + col];
      } else {                             // row = v, col = k*C + c
        const int k = col / C, c = col - k * C;
        for (int w = 0; w < V; ++w) acc += As[(k * V + row) * V + w] * Xs[w * (Cin + 1) + c];
      }
      yf[i] = acc;
      if (stats) Ys[row * (Cout + 1) + col] = acc;
    }
    if (stats) {
      __syncthreads();
      for (int c = threadIdx.x; c < Cout; c += blockDim.x) {
        float sum = 0.f;
        for (int r = 0; r < V; ++r) sum += Ys[r * (Cout + 1) + c];
        const float mean = sum / (float)V;
        float m2 = 0.f;
        for (int r = 0; r < V; ++r) { const float d = Ys[r * (Cout + 1) + c] - mean; m2 += d * d; }
        float* rec = stats + ((long)c * F + f) * 3;
        rec[0] = (float)V; rec[1] = mean; rec[2] = m2;
      }
    }
  }
}

__global__ __launch_bounds__(256) void tconv_pack_kernel(const float* __restrict__ W, int Co, int Ci, int taps, int mode,
                                                         float* __restrict__ Wp) {
  const long total = (long)Co * Ci * taps;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int tap = (int)(i % taps);
    const long q = i / taps;
    const int ci = (int)(q % Ci), co = (int)(q / Ci);
    const float w = W[i];
    if (mode != 1) Wp[((long)tap * Co + co) * Ci + ci] = w;                                      // mode 0 / 2: forward pack
    if (mode != 0) Wp[(mode == 2 ? total : 0) + ((long)(taps - 1 - tap) * Ci + ci) * Co + co] = w;   // mode 1 / 2: gradient pack
  }
}

struct TconvP {
  const float* X; long ldx;          // input rows (b, t, v) x Cin
  const float* in_state;             // [4][Cin] mean, invstd, a, b of the BatchNorm in front (+ ReLU); null: plain input
  const float* W; long wts, wns, wks; // weight element (tap, n = output channel, k = input channel) at W[tap*wts + n*wns + k*wks]
  const float* bias;                 // [Cout] or null
  float* Y; long ldy;
  long rows; int T, V, Cin, Cout, taps;
};

// acc[32x32] += A[32 rows][K] . B[32 rows][K]^T (row-major LDS tiles, stride GC_S), K a multiple of 4 (zero padded).
// One ds_read_b64 per operand feeds two MFMA steps: lane (r, h) holds k = 4j + 2h, 4j + 2h + 1 -- the same k-permutation on
// both operands, so the sum over k is unchanged.
typedef float f32x2 __attribute__((ext_vector_type(2)));
// Workgroup = NG groups of 4 waves on ONE 64 x 64 output tile; group g takes the (tap, channel-chunk) steps g, g + NG, ... with
// operand tiles of its own, so NG steps' loads are in flight at once, and the groups' partial tiles are added through LDS in a
// fixed order at the end.  NG = 4 for small row counts (the training shape: 120 row tiles, where one step at a time left the
// kernel waiting on 9-18 dependent memory round trips), NG = 1 when the grid alone fills the chip.
template <int NG>
__global__ __launch_bounds__(256 * NG) void tconv_kernel(TconvP p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, grp = tid >> 8, t = tid & 255, lane = tid & 63, wave = t >> 6;
  float* As = smem + grp * 2 * 64 * GC_S;
  float* Bs = As + 64 * GC_S;
  const int rt = wave & 1, ct = wave >> 1;
  const long r0 = (long)blockIdx.x * 64;
  const int n0 = blockIdx.y * 64;
  const int half = p.taps / 2;
  const int xk = t & 63, xr = t >> 6;
  const int nkc = (p.Cin + 63) / 64;                       // 64-channel chunks per tap
  const int nchunks = p.taps * nkc;
  // this thread's 16 rows of the tile (rows xr + 4 j): their frame index t, four per register as signed bytes (taps that leave
  // [0, T) are masked; rows past the end count as t = -128).  Addresses are kept as uniform row pointers (scalar registers)
  // plus ONE 32-bit lane offset per operand: spelled out per element, the 32 loads of a step cost 64 address registers and the
  // kernel spilled.
  int tpk[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    int w = 0;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long row = r0 + xr + 4 * (4 * q + u);
      const int tv = row < p.rows ? (int)((row / p.V) % p.T) : -128;
      w |= (tv & 255) << (8 * u);
    }
    tpk[q] = w;
  }
  const int xoff = xr * (int)p.ldx + xk;                   // lane part of an input address
  const int woff = xr * (int)p.wns + xk * (int)p.wks;      // lane part of a weight address
  float av[16], bv[16];
#define TC_FETCH(ch)                                                                                \
  do {                                                                                              \
    const int tap_ = (ch) / nkc, k0_ = ((ch) % nkc) * 64;                                           \
    const int d_ = tap_ - half;                                                                     \
    const bool kok_ = k0_ + xk < p.Cin;                                                             \
    float mu_ = 0.f, a_ = 1.f, b_ = 0.f;                                                            \
    if (p.in_state && kok_) { mu_ = p.in_state[k0_ + xk]; a_ = p.in_state[2 * p.Cin + k0_ + xk]; b_ = p.in_state[3 * p.Cin + k0_ + xk]; } \
    const float* xrow_ = p.X + (r0 + (long)d_ * p.V) * p.ldx + k0_;               /* uniform */     \
    const float* wrow_ = p.W + (long)tap_ * p.wts + (long)n0 * p.wns + (long)k0_ * p.wks;           \
    _Pragma("unroll") for (int j = 0; j < 16; ++j) {                                                \
      const int ts_ = (int)(signed char)(tpk[j >> 2] >> (8 * (j & 3))) + d_;                        \
      const bool ok_ = kok_ && ts_ >= 0 && ts_ < p.T;                                               \
      float v_ = ok_ ? (xrow_ + (long)(4 * j) * p.ldx)[xoff] : 0.f;                                 \
      if (p.in_state) v_ = ok_ ? fmaxf(__builtin_fmaf(v_ - mu_, a_, b_), 0.f) : 0.f;                \
      av[j] = v_;                                                                                   \
      bv[j] = (n0 + xr + 4 * j < p.Cout && kok_) ? (wrow_ + (long)(4 * j) * p.wns)[woff] : 0.f;     \
    }                                                                                               \
  } while (0)
  if (grp < nchunks) TC_FETCH(grp);
  f32x16 acc = {0};
  for (int ch0 = 0; ch0 < nchunks; ch0 += NG) {
    const int ch = ch0 + grp;
    __syncthreads();
    if (ch < nchunks) {
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        As[(xr + 4 * j) * GC_S + xk] = av[j];
        Bs[(xr + 4 * j) * GC_S + xk] = bv[j];
      }
    }
    __syncthreads();
    if (ch + NG < nchunks) TC_FETCH(ch + NG);
    if (ch < nchunks) {
      const int kleft = p.Cin - (ch % nkc) * 64;
      const int K = kleft >= 64 ? 64 : ((kleft + 3) & ~3);
      const int r = lane & 31, h = lane >> 5;
      const f32x2* bp = reinterpret_cast<const f32x2*>(Bs + (ct * 32 + r) * GC_S + 2 * h);
      const f32x2* ap = reinterpret_cast<const f32x2*>(As + (rt * 32 + r) * GC_S + 2 * h);
#pragma unroll 4
      for (int k = 0; k < K; k += 4) {
        const f32x2 a = ap[k >> 1], b = bp[k >> 1];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
      }
    }
  }
  const int lcol = ct * 32 + (lane & 31);
  if (NG > 1) {                                            // groups 1 .. NG-1 hand their partial tiles to group 0 (fixed order)
    __syncthreads();
    float* xch = smem;                                     // [NG-1][64][64]
    if (grp > 0) {
#pragma unroll
      for (int reg = 0; reg < 16; ++reg)
        xch[((grp - 1) * 64 + rt * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)) * 64 + lcol] = acc[reg];
    }
    __syncthreads();
    if (grp > 0) return;
#pragma unroll
    for (int g = 1; g < NG; ++g)
#pragma unroll
      for (int reg = 0; reg < 16; ++reg)
        acc[reg] += xch[((g - 1) * 64 + rt * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)) * 64 + lcol];
  }
  const int col = n0 + lcol;
  if (col < p.Cout) {
    const float bb = p.bias ? p.bias[col] : 0.f;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const long row = r0 + rt * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
      if (row < p.rows) p.Y[row * p.ldy + col] = acc[reg] + bb;
    }
  }
}

extern "C" int mmego_graph_mix(void* stream, const float* X, const float* A, const float* importance, float* Y, long F, int V, int K,
                               int C, int backward, float* stats, long ldx) {
  MMEGO_REQUIRE(X && A && importance && Y && F > 0 && V >= 1 && V <= 32 && K >= 1 && K <= 4 && C >= 1);
  MMEGO_REQUIRE(ldx >= (backward ? C : K * C));
  MMEGO_REQUIRE(!stats || (!backward && F <= 1024));     // (one record per frame and channel: mmego_bn_finalize takes <= 1024)
  const int Cin = backward ? C : K * C, Cout = backward ? K * C : C;
  const size_t lds = (size_t)(K * V * V + V * (Cin + 1) + (stats ? V * (Cout + 1) : 0)) * sizeof(float);
  MMEGO_REQUIRE(lds <= 64 * 1024);
  const int grid = (int)(F < 2048 ? F : 2048);
  hipLaunchKernelGGL(graph_mix_kernel, dim3(grid), dim3(256), lds, (hipStream_t)stream, X, A, importance, Y, F, V, K, C, backward,
                     stats, ldx);
  MMEGO_LAUNCH_CHECK();
  return MMEGO_OK;
}

extern "C" int mmego_tconv_pack(void* stream, const float* W, int Co, int Ci, int taps, int mode, float* Wp) {
  MMEGO_REQUIRE(W && Wp && Co >= 1 && Ci >= 1 && taps >= 1 && mode >= 0 && mode <= 2);
  long b = ((long)Co * Ci * taps + 255) / 256;
  hipLaunchKernelGGL(tconv_pack_kernel, dim3((int)(b > 1024 ? 1024 : b)), dim3(256), 0, (hipStream_t)stream, W, Co, Ci, taps, mode, Wp);
  MMEGO_LAUNCH_CHECK();
  return MMEGO_OK;
}

extern "C" int mmego_tconv(void* stream, const float* X, long ldx, const float* in_state, const float* W, long wts, long wns, long wks,
                           const float* bias, float* Y, long ldy, int B, int T, int V, int Cin, int Cout, int taps) {
  MMEGO_REQUIRE(X && W && Y && B > 0 && T > 0 && V > 0 && Cin >= 1 && Cout >= 1 && taps >= 1 && (taps & 1) && ldx >= Cin && ldy >= Cout);
  TconvP p = {X, ldx, in_state, W, wts, wns, wks, bias, Y, ldy, (long)B * T * V, T, V, Cin, Cout, taps};
  dim3 grid((unsigned)((p.rows + 63) / 64), (unsigned)((Cout + 63) / 64));
  if ((long)grid.x * grid.y <= 512) {                      // small grid: four steps in flight per workgroup
    const size_t lds = (size_t)4 * 2 * 64 * GC_S * sizeof(float);
    static bool attr = false;
    if (!attr) {
      hipError_t e = hipFuncSetAttribute((const void*)tconv_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return (int)e;
      attr = true;
    }
    hipLaunchKernelGGL(tconv_kernel<4>, grid, dim3(1024), lds, (hipStream_t)stream, p);
  } else {
    hipLaunchKernelGGL(tconv_kernel<1>, grid, dim3(256), (size_t)2 * 64 * GC_S * sizeof(float), (hipStream_t)stream, p);
  }
  MMEGO_LAUNCH_CHECK();
  return MMEGO_OK;
}
