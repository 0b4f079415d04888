// Batch statistics carried between the kernels of a fused ST-GCN training step as PARTIAL RECORDS (gcn_fused.hip, gcn.hip).
//
// A producer kernel that has a tensor's tile in its hands leaves, per workgroup j and channel c, the record
//   rec[j][c] = (mean_j, M2_j)      mean and sum of squared deviations of channel c over the rows the workgroup owns
// (float2; the row count n_j follows from the geometry: rows_per_rec rows per record, the last one ragged).  The consumer of the
// BatchNorm finalizes the statistics in its PROLOGUE -- every workgroup redundantly, every thread taking part, in a fixed order
// (bit-identical everywhere) -- instead of a finalize launch between the two:
//   N = sum n_j,  mean = sum n_j mean_j / N,  M2 = sum M2_j + sum n_j mean_j^2 - N mean^2      (all in fp64)
// Workgroup 0 stores mean / invstd / a / b for the backward pass and updates the running statistics like torch (momentum,
// unbiased variance).  At most ~128 records per BatchNorm (one round of loads per thread).
#pragma once
#include "common.h"

struct BnRefD {
  const float2* rec; int nrec; int rpr;              // records [nrec][C], rows per record
  const float* gamma; const float* beta; float* rmean; float* rvar; float momentum; float eps;
  float* state;                                      // [4][C] mean, invstd, a, b (workgroup 0 writes it) or null
};

// red: LDS scratch, (NT / CP) * 4 * CP doubles with CP = pow2 >= max(C, 32); st: LDS [4][C] <- mean, a, b, invstd.
// Ends with a barrier (st is readable, red is free again).
template <int NT>
__device__ __forceinline__ void bn_from_records(const BnRefD& bn, int C, long rows, double* red, float* st, bool first_wg) {
  int CP = 32;
  while (CP < C) CP <<= 1;
  const int Q = NT / CP;
  const int tid = threadIdx.x, c = tid & (CP - 1), q = tid / CP;
  const int cc = c < C ? c : C - 1;
  double n_ = 0.0, s_ = 0.0, ss_ = 0.0, m2_ = 0.0;
  for (int j0 = q; j0 < bn.nrec; j0 += 8 * Q) {
    float2 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {                     // clamped index: every load is issued, the select comes afterwards
      const int j = j0 + u * Q;
      v[u] = bn.rec[(long)(j < bn.nrec ? j : bn.nrec - 1) * C + cc];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int j = j0 + u * Q;
      asm volatile("" : "+v"(v[u].x), "+v"(v[u].y));
      const long left = rows - (long)j * bn.rpr;
      const double nn = j < bn.nrec ? (double)(left < bn.rpr ? left : bn.rpr) : 0.0;
      const double m = (double)v[u].x;
      n_ += nn; s_ += nn * m; ss_ += nn * m * m; m2_ += j < bn.nrec ? (double)v[u].y : 0.0;
    }
  }
  red[(q * 4 + 0) * CP + c] = n_; red[(q * 4 + 1) * CP + c] = s_; red[(q * 4 + 2) * CP + c] = ss_; red[(q * 4 + 3) * CP + c] = m2_;
  __syncthreads();
  if (tid < C) {
    double N = 0.0, S = 0.0, SS = 0.0, M2 = 0.0;
    for (int g = 0; g < Q; ++g) {
      N += red[(g * 4 + 0) * CP + tid]; S += red[(g * 4 + 1) * CP + tid]; SS += red[(g * 4 + 2) * CP + tid]; M2 += red[(g * 4 + 3) * CP + tid];
    }
    const double mean = S / N;
    double m2 = M2 + SS - N * mean * mean;
    m2 = m2 > 0.0 ? m2 : 0.0;
    const double var = m2 / N;
    const float invstd = (float)(1.0 / sqrt(var + (double)bn.eps));
    const float a = bn.gamma[tid] * invstd, b = bn.beta[tid];
    st[tid] = (float)mean; st[C + tid] = a; st[2 * C + tid] = b; st[3 * C + tid] = invstd;
    if (first_wg) {
      if (bn.state) { bn.state[tid] = (float)mean; bn.state[C + tid] = invstd; bn.state[2 * C + tid] = a; bn.state[3 * C + tid] = b; }
      if (bn.rmean) {
        bn.rmean[tid] = (1.f - bn.momentum) * bn.rmean[tid] + bn.momentum * (float)mean;
        const double unbiased = N > 1.0 ? m2 / (N - 1.0) : var;
        bn.rvar[tid] = (1.f - bn.momentum) * bn.rvar[tid] + bn.momentum * (float)unbiased;
      }
    }
  }
  __syncthreads();
}

// (mean, M2) of n values given their shifted sums s1 = sum (x - shift), s2 = sum (x - shift)^2
__device__ __forceinline__ float2 rec_from_shifted(float shift, float s1, float s2, int n) {
  const float md = s1 / (float)n;
  return float2{shift + md, fmaxf(s2 - s1 * md, 0.f)};
}

// host-side mirror of include/mmego_hip.h's MmegoBnRef (same field order)
struct MmegoBnRefH {
  const float* rec; int nrec; int rows_per_rec;
  const float* gamma; const float* beta; float* running_mean; float* running_var; float momentum; float eps;
  float* state;
};
static inline BnRefD bnref_device(const MmegoBnRefH* h) {
  BnRefD d;
  d.rec = reinterpret_cast<const float2*>(h->rec); d.nrec = h->nrec; d.rpr = h->rows_per_rec;
  d.gamma = h->gamma; d.beta = h->beta; d.rmean = h->running_mean; d.rvar = h->running_var; d.momentum = h->momentum; d.eps = h->eps;
  d.state = h->state;
  return d;
}
