// Parameter block and launcher of the large-tile GEMM kernels (gemm_tile.hip), shared with the dispatch in gemm.hip.
#pragma once
#include <hip/hip_runtime.h>

struct TileP {
  const float* A;     // A_KC: A[m * lda + k]   else A[k * lda + m]
  const float* W;     // B_KC: W[n * ldw + k]   else W[k * ldw + n]
  float* C;           // C[m * ldc + n]
  const float* bias;
  int M, N, K;
  long lda, ldw, ldc;
  int relu, accumulate;
  int nsplit, kchunk; // split-K: slab s covers k in [s * kchunk, min(K, (s+1) * kchunk)), kchunk % 64 == 0
  float* ws;          // [nsplit][nbatch][M][N] partial products when nsplit > 1 (bias / relu / accumulate: the reducer's job)
  int nbatch;         // independent products sharing the shapes; element strides between them (may be negative):
  long sAb, sWb, sCb, sBiasb;
};

namespace mmego_detail {
// returns 0 on launch, -2 if the shape does not fit these kernels (caller falls back), >0 on a HIP error.
int gemm_tile_launch(hipStream_t st, const TileP& p, bool a_kc, bool b_kc);
}
