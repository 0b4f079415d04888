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
#include "common.h"

namespace mmego_detail {
int gemm_tile_launch(hipStream_t st, const float* A, const float* W, float* C, const float* bias, int M, int N, int K,
                     long lda, long ldw, long ldc, int relu);
}

struct GemmP {
  const float* A;
  const float* B;
  float* C;
  const float* bias;
  long sam, sak, sbk, sbn, scm, scn;
  long sAb, sBb, sCb, sCs;
  int M, N, K;
  int nsplit, kchunk;
  int relu, accumulate;
};

#define LD64 68

template <bool A_KC, bool B_KC>
__global__ __launch_bounds__(256) void gemm64_kernel(GemmP p) {
  __shared__ float As[16][LD64];
  __shared__ float Bs[16][LD64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave & 1, wn = wave >> 1;
  const int batch = blockIdx.z / p.nsplit, split = blockIdx.z % p.nsplit;
  const int m0 = blockIdx.x * 64, n0 = blockIdx.y * 64;   // M tiles on grid.x (no 65535 limit: rows can be millions)
  const float* A = p.A + (long)batch * p.sAb;
  const float* B = p.B + (long)batch * p.sBb;
  const int kbeg = split * p.kchunk;
  const int kend = min(p.K, kbeg + p.kchunk);

  // staging coordinates: 4 elements per thread per operand
  int am, ak[4], bn, bk[4];
  if (A_KC) { am = tid >> 2; for (int j = 0; j < 4; ++j) ak[j] = (tid & 3) * 4 + j; }
  else      { am = tid & 63; for (int j = 0; j < 4; ++j) ak[j] = (tid >> 6) + 4 * j; }
  if (B_KC) { bn = tid >> 2; for (int j = 0; j < 4; ++j) bk[j] = (tid & 3) * 4 + j; }
  else      { bn = tid & 63; for (int j = 0; j < 4; ++j) bk[j] = (tid >> 6) + 4 * j; }
  const bool am_ok = (m0 + am) < p.M, bn_ok = (n0 + bn) < p.N;
  const float* Arow = A + (long)(m0 + am) * p.sam;
  const float* Bcol = B + (long)(n0 + bn) * p.sbn;

  float ra[4], rb[4];
  auto gload = [&](int k0) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      int ka = k0 + ak[j], kb = k0 + bk[j];
      ra[j] = (am_ok && ka < kend) ? Arow[(long)ka * p.sak] : 0.0f;
      rb[j] = (bn_ok && kb < kend) ? Bcol[(long)kb * p.sbk] : 0.0f;
    }
  };

  f32x16 acc = {0};
  if (kbeg < kend) gload(kbeg);
  for (int k0 = kbeg; k0 < kend; k0 += 16) {
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      As[ak[j]][am] = ra[j];
      Bs[bk[j]][bn] = rb[j];
    }
    __syncthreads();
    if (k0 + 16 < kend) gload(k0 + 16);
    const int r = lane & 31, h = lane >> 5;
#pragma unroll
    for (int kk = 0; kk < 16; kk += 2) {
      float a = As[kk + h][wm * 32 + r];
      float b = Bs[kk + h][wn * 32 + r];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
  }

  // epilogue: lane holds column (lane&31), rows (reg&3)+8*(reg>>2)+4*(lane>>5)
  const int col = n0 + wn * 32 + (lane & 31);
  if (col >= p.N) return;
  float* C = p.C + (long)batch * p.sCb + (long)split * p.sCs;
  const float bv = (p.bias && p.nsplit == 1) ? p.bias[col] : 0.0f;
#pragma unroll
  for (int reg = 0; reg < 16; ++reg) {
    int row = m0 + wm * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
    if (row < p.M) {
      float v = acc[reg] + bv;
      float* dst = C + (long)row * p.scm + (long)col * p.scn;
      if (p.nsplit == 1) {
        if (p.relu) v = fmaxf(v, 0.0f);
        if (p.accumulate) v += *dst;
      }
      *dst = v;
    }
  }
}

// ws: [nsplit][nbatch][M][N] contiguous partial products -> C (strided), + bias, relu, accumulate.
// 64 outputs x 4 split-lanes per block; the 4 partial sums are combined in a fixed order (deterministic).
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* ws, float* C, const float* bias, int nsplit,
                                                            int nbatch, int M, int N, long scm, long scn, long sCb,
                                                            int relu, int accumulate) {
  __shared__ float sh[4][64];
  const long total = (long)nbatch * M * N;
  const int o = threadIdx.x & 63, kg = threadIdx.x >> 6;
  const long i = (long)blockIdx.x * 64 + o;
  float s = 0.0f;
  if (i < total)
    for (int k = kg; k < nsplit; k += 4) s += ws[(long)k * total + i];
  sh[kg][o] = s;
  __syncthreads();
  if (kg == 0 && i < total) {
    s = ((sh[0][o] + sh[1][o]) + sh[2][o]) + sh[3][o];
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
                          long sBb, long sCb, int relu, int accumulate, float* splitk_ws, int nsplit) {
  MMEGO_REQUIRE(A && B && C && M > 0 && N > 0 && K > 0 && nbatch > 0 && nsplit >= 1);
  hipStream_t st = (hipStream_t)stream;
  const bool nt_aligned = nbatch == 1 && nsplit == 1 && !accumulate && sak == 1 && sbk == 1 && scn == 1 && (sam % 4) == 0 &&
                          (sbn % 4) == 0 && (((uintptr_t)A | (uintptr_t)B) & 15) == 0;
  if (nt_aligned) {   // large-tile kernel (gemm_tile.hip) when the shape allows
    int rc = mmego_detail::gemm_tile_launch(st, A, B, C, bias, M, N, K, sam, sbn, scm, relu);
    if (rc != -2) return rc;
  }
  const bool fast = nbatch == 1 && nsplit == 1 && !accumulate && sak == 1 && sbk == 1 && scn == 1 && (M % 128) == 0 &&
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
  p.sAb = sAb; p.sBb = sBb;
  p.M = M; p.N = N; p.K = K;
  p.nsplit = nsplit;
  p.relu = relu; p.accumulate = accumulate;
  if (nsplit > 1) {
    MMEGO_REQUIRE(splitk_ws != nullptr);
    int kc = cdiv(K, nsplit);
    p.kchunk = cdiv(kc, 16) * 16;
    p.C = splitk_ws; p.scm = N; p.scn = 1; p.sCb = (long)M * N; p.sCs = (long)nbatch * M * N;
  } else {
    p.kchunk = cdiv(K, 16) * 16;
    p.C = C; p.scm = scm; p.scn = scn; p.sCb = sCb; p.sCs = 0;
  }
  MMEGO_REQUIRE(cdiv(N, 64) <= 65535 && (long)nbatch * nsplit <= 65535);
  dim3 grid(cdiv(M, 64), cdiv(N, 64), nbatch * nsplit);
  const bool akc = (sak == 1), bkc = (sbk == 1);
  if (akc && bkc) hipLaunchKernelGGL((gemm64_kernel<true, true>), grid, dim3(256), 0, st, p);
  else if (akc) hipLaunchKernelGGL((gemm64_kernel<true, false>), grid, dim3(256), 0, st, p);
  else if (bkc) hipLaunchKernelGGL((gemm64_kernel<false, true>), grid, dim3(256), 0, st, p);
  else hipLaunchKernelGGL((gemm64_kernel<false, false>), grid, dim3(256), 0, st, p);
  MMEGO_LAUNCH_CHECK();
  if (nsplit > 1) {
    long total = (long)nbatch * M * N;
    int blocks = (int)((total + 63) / 64);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, st, splitk_ws, C, bias, nsplit, nbatch, M, N,
                       scm, scn, sCb, relu, accumulate);
    MMEGO_LAUNCH_CHECK();
  }
  return MMEGO_OK;
}
