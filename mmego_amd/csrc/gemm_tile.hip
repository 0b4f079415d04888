// fp32 NT GEMM, large-tile kernel: C[M,N] = A[M,K] . W[N,K]^T (+bias)(relu), K % 64 == 0.
//
// The dominant dense products of the path run here: the LSTM input projections of IMU_Net
// (reference Net/IMU_Net.py:58-62: 10240 x 2048 x {512,1024}, 512 x 2048 x 1024) and the 128-aligned Linear layers.
//
//  * v_mfma_f32_32x32x2_f32 (exact fp32 fma chain), BM x BN block tile (128x128 or 64x64), 4 waves as 2x2, each
//    wave (BM/2) x (BN/2) = TM x TN tiles of 32x32.
//  * Operand tiles live in LDS as [row][k] with a 68-float row stride (64 k + 4 pad): the global->LDS copy is a plain
//    f32x4 -> ds_write_b128 (no transpose), and the operand fetch is ONE ds_read_b128 per 4 MFMA steps using a
//    k-permutation (lane half h supplies k = 8*kb + 4*h + s at step s, identically for A and B).  With this stride
//    every 16-lane group of the b128 read hits 16 distinct 16-B slots (conflict-free).
//  * One LDS buffer + register prefetch of the next 64-k chunk (global loads are issued before the MFMA block and
//    land under it), so 2 workgroups fit a CU (69.6 KB each at 128x128) and overlap each other's barriers.
//  * XCD-aware tile order: consecutive tiles along N (sharing the A row panel) stay on one XCD's L2.
#include <stdlib.h>

#include "common.h"

#define TLD 68
// NOTE: staging registers are ext_vector f32x4 (not HIP's float4 struct): arrays of the struct type are left in
// scratch memory by hipcc (ROCm 7.2), which serialises the prefetch.

template <int BM, int BN, int WAVES_M, int WAVES_N>
__device__ __forceinline__ void gemm_tile_body(const float* __restrict__ A, const float* __restrict__ W, float* __restrict__ C,
                                               const float* __restrict__ bias, int K, long lda, long ldw, long ldc, int relu,
                                               int m0, int n0, float* smem, int sid) {
  static_assert(WAVES_M * WAVES_N == 4, "4 waves");
  constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;   // wave tile
  constexpr int TM = WM / 32, TN = WN / 32;           // 32x32 MFMA tiles per wave
  constexpr int AV = BM / 16, BV = BN / 16;          // f32x4 per thread per 64-k chunk (rows lr + 16*i)
  float (*As)[TLD] = reinterpret_cast<float (*)[TLD]>(smem);
  float (*Bs)[TLD] = reinterpret_cast<float (*)[TLD]>(smem + BM * TLD);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave % WAVES_M, wn = wave / WAVES_M;

  const int lk = (tid & 15) * 4, lr = tid >> 4;      // 16 lanes cover one 256-B row segment
  const float* Ap = A + (long)(m0 + lr) * lda + lk;
  const float* Wp = W + (long)(n0 + lr) * ldw + lk;
  f32x4 ra[AV], rb[BV];
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = (f32x16){0};

  const int nk = K / 64;
  MMEGO_STAMP_AT(sid, 0, tid == 0);
#pragma unroll
  for (int i = 0; i < AV; ++i) ra[i] = *reinterpret_cast<const f32x4*>(Ap + (long)(16 * i) * lda);
#pragma unroll
  for (int i = 0; i < BV; ++i) rb[i] = *reinterpret_cast<const f32x4*>(Wp + (long)(16 * i) * ldw);
  const int r = lane & 31, h = lane >> 5;
  MMEGO_STAMP_AT(sid, 1, tid == 0);
  for (int kt = 0; kt < nk; ++kt) {
    __syncthreads();                                  // previous chunk's operand reads are done
#pragma unroll
    for (int i = 0; i < AV; ++i) *reinterpret_cast<f32x4*>(&As[lr + 16 * i][lk]) = ra[i];
#pragma unroll
    for (int i = 0; i < BV; ++i) *reinterpret_cast<f32x4*>(&Bs[lr + 16 * i][lk]) = rb[i];
    __syncthreads();
    if (kt + 1 < nk) {
      const int k0 = (kt + 1) * 64;
#pragma unroll
      for (int i = 0; i < AV; ++i) ra[i] = *reinterpret_cast<const f32x4*>(Ap + (long)(16 * i) * lda + k0);
#pragma unroll
      for (int i = 0; i < BV; ++i) rb[i] = *reinterpret_cast<const f32x4*>(Wp + (long)(16 * i) * ldw + k0);
    }
#pragma unroll
    for (int kb = 0; kb < 8; ++kb) {
      f32x4 a[TM], b[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const f32x4*>(&As[wm * WM + i * 32 + r][kb * 8 + 4 * h]);
#pragma unroll
      for (int j = 0; j < TN; ++j) b[j] = *reinterpret_cast<const f32x4*>(&Bs[wn * WN + j * 32 + r][kb * 8 + 4 * h]);
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

#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = n0 + wn * WN + j * 32 + (lane & 31);
    const float bv = bias ? bias[col] : 0.0f;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        int row = m0 + wm * WM + i * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
        float v = acc[i][j][reg] + bv;
        if (relu) v = fmaxf(v, 0.0f);
        C[(long)row * ldc + col] = v;
      }
    }
  }
  MMEGO_STAMP_AT(sid, 3, tid == 0);
}

// XCD-aware tile order: blocks b and b+8 share an XCD; hand each XCD a contiguous run of tile ids
__device__ __forceinline__ int xcd_order(int id, int n) { return (n & 7) == 0 ? (id & 7) * (n >> 3) + (id >> 3) : id; }

template <int BM, int BN, int WAVES_M, int WAVES_N>
__global__ __launch_bounds__(256) void gemm_tile_kernel(const float* __restrict__ A, const float* __restrict__ W,
                                                        float* __restrict__ C, const float* __restrict__ bias, int M, int N,
                                                        int K, long lda, long ldw, long ldc, int relu) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int ntn = N / BN, nwg = ntn * (M / BM);
  const int id = xcd_order(blockIdx.x, nwg);
  gemm_tile_body<BM, BN, WAVES_M, WAVES_N>(A, W, C, bias, K, lda, ldw, ldc, relu, (id / ntn) * BM, (id % ntn) * BN, smem, id);
}

// Persistent, tail-balanced launch for outputs of more than one wave of co-resident workgroups.  The grid is exactly the
// 512 co-resident workgroups (2 per CU); workgroup w walks a STATIC tile list, so no CU can end up with an extra tile
// (measured with per-workgroup stamps: the hardware dispatcher hands freed slots out greedily and 1-3 CUs regularly
// received 6 of the 1280 tiles of the 10240 x 2048 projections instead of 5: 383 us instead of 330).  When the tile
// count leaves at most half a wave of workgroups over, those tiles are cut into two 128x64 halves, one per workgroup:
// the tail costs half a tile time (2.5 tile times per workgroup for 1280 tiles, not 3).  Per output element the
// arithmetic is the plain kernel's (bit-identical result).
__global__ __launch_bounds__(256) void gemm_tile_persistent_kernel(const float* __restrict__ A, const float* __restrict__ W,
                                                                   float* __restrict__ C, const float* __restrict__ bias, int M,
                                                                   int N, int K, long lda, long ldw, long ldc, int relu, int nfull,
                                                                   int nhalf, int stagger) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int ntn = N / 128, G = (int)gridDim.x;
  const int w = xcd_order(blockIdx.x, G);                  // each XCD walks a contiguous run of tile ids per round
  // The two workgroups of a CU (blocks b and b + G/2) start together and would reach their store epilogues together,
  // leaving the matrix pipe idle; the second one therefore runs its half tile FIRST, which keeps the pair out of phase.
  const bool halves_first = (int)blockIdx.x >= G / 2 && stagger;
  if (halves_first)
    for (int q = w; q < nhalf; q += G) {
      const int id = nfull + (q >> 1);
      gemm_tile_body<128, 64, 2, 2>(A, W, C, bias, K, lda, ldw, ldc, relu, (id / ntn) * 128, (id % ntn) * 128 + (q & 1) * 64, smem, nfull + q);
    }
  for (int id = w; id < nfull; id += G)
    gemm_tile_body<128, 128, 2, 2>(A, W, C, bias, K, lda, ldw, ldc, relu, (id / ntn) * 128, (id % ntn) * 128, smem, id);
  if (!halves_first)
    for (int q = w; q < nhalf; q += G) {
      const int id = nfull + (q >> 1);
      gemm_tile_body<128, 64, 2, 2>(A, W, C, bias, K, lda, ldw, ldc, relu, (id / ntn) * 128, (id % ntn) * 128 + (q & 1) * 64, smem, nfull + q);
    }
}

namespace mmego_detail {

template <int BM, int BN, int WAVES_M, int WAVES_N>
static int launch_cfg(hipStream_t st, const float* A, const float* W, float* C, const float* bias, int M, int N, int K, long lda,
                      long ldw, long ldc, int relu) {
  static bool attr_set = false;
  const size_t lds = (size_t)((BM + BN) * TLD) * sizeof(float);
  if (!attr_set && lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_tile_kernel<BM, BN, WAVES_M, WAVES_N>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  hipLaunchKernelGGL((gemm_tile_kernel<BM, BN, WAVES_M, WAVES_N>), dim3((unsigned)((M / BM) * (N / BN))), dim3(256), lds, st, A, W, C,
                     bias, M, N, K, lda, ldw, ldc, relu);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}

// returns 0 on launch, -2 if the shape does not fit this kernel (caller falls back), >0 on a HIP error.
// Tile choice: the configuration with the fewest (waves of co-resident workgroups) x (tile area).  A 160x128 tile
// (1024 tiles = exactly 2 waves of 512 for the 10240 x 2048 LSTM projections, vs 2.5 -> 3 waves of 128x128) was measured
// 3-4 % SLOWER (1x4 wave layout: 6 operand reads per 5 MFMAs), so it is not in the list.
int gemm_tile_launch(hipStream_t st, const float* A, const float* W, float* C, const float* bias, int M, int N, int K,
                     long lda, long ldw, long ldc, int relu) {
  if ((K % 64) != 0 || (M % 64) != 0 || (N % 64) != 0) return -2;
  // 128x128 whenever it yields enough tiles to occupy the 256 CUs (measured: 128x128 runs the 10240 x 2048 projections
  // at 93-109 TFLOP/s, 64x64 at 88-97), else 64x64 (e.g. M = 512: 23.6 us vs 84 us with 64 big tiles).
  const bool big_ok = (M % 128) == 0 && (N % 128) == 0 && (long)(M / 128) * (N / 128) >= 192;
  if (big_ok) {
    const int tiles = (M / 128) * (N / 128), slots = 512;      // 2 workgroups per CU x 256 CUs
    static const bool no_persist = getenv("MMEGO_GEMM_NO_PERSIST") != nullptr;
    static const int stagger = getenv("MMEGO_GEMM_NO_STAGGER") == nullptr;
    if (tiles > slots && !no_persist) {
      const int rest = tiles % slots;
      const bool halves = rest > 0 && rest <= slots / 2;
      const int nfull = halves ? tiles - rest : tiles, nhalf = halves ? 2 * rest : 0;
      static bool attr_set = false;
      const size_t lds = (size_t)(256 * TLD) * sizeof(float);
      if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_tile_persistent_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
      }
      hipLaunchKernelGGL(gemm_tile_persistent_kernel, dim3(slots), dim3(256), lds, st, A, W, C, bias, M, N, K, lda, ldw, ldc, relu,
                         nfull, nhalf, stagger);
      hipError_t e = hipGetLastError();
      return e == hipSuccess ? 0 : (int)e;
    }
    return launch_cfg<128, 128, 2, 2>(st, A, W, C, bias, M, N, K, lda, ldw, ldc, relu);
  }
  return launch_cfg<64, 64, 2, 2>(st, A, W, C, bias, M, N, K, lda, ldw, ldc, relu);
}

}  // namespace mmego_detail
