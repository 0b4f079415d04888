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
// Two halves so that the gathers of two BatchNorms can share one barrier pair (bn_gather ... bn_gather, barrier, bn_finish ...
// bn_finish, barrier); 16 record loads in flight per thread and round.
template <int NT>
__device__ __forceinline__ void bn_gather(const BnRefD& bn, int C, long rows, double* red) {
  int CP = 32;
  while (CP < C) CP <<= 1;
  const int Q = NT / CP;
  const int tid = threadIdx.x, c = tid & (CP - 1), q = tid / CP;
  const int cc = c < C ? c : C - 1;
  double n_ = 0.0, s_ = 0.0, ss_ = 0.0, m2_ = 0.0;
  for (int j0 = q; j0 < bn.nrec; j0 += 16 * Q) {
    float2 v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {                    // clamped index: every load is issued, the select comes afterwards
      const int j = j0 + u * Q;
      v[u] = bn.rec[(long)(j < bn.nrec ? j : bn.nrec - 1) * C + cc];
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int j = j0 + u * Q;
      asm volatile("" : "+v"(v[u].x), "+v"(v[u].y));
      const long left = rows - (long)j * bn.rpr;
      const double nn = j < bn.nrec ? (double)(left < bn.rpr ? left : bn.rpr) : 0.0;
      const double m = (double)v[u].x;
      n_ += nn; s_ += nn * m; ss_ += nn * m * m; m2_ += j < bn.nrec ? (double)v[u].y : 0.0;
    }
  }
  red[(q * 4 + 0) * CP + c] = n_; red[(q * 4 + 1) * CP + c] = s_; red[(q * 4 + 2) * CP + c] = ss_; red[(q * 4 + 3) * CP + c] = m2_;
}

// bn_gather of TWO BatchNorms over the same channel count with the loads of both in flight together (one memory round trip per
// round instead of two: a record written by the kernel before lies in another XCD's reach, ~2 us away)
template <int NT>
__device__ __forceinline__ void bn_gather2(const BnRefD& b1, const BnRefD& b2, int C, long rows, double* red1, double* red2) {
  int CP = 32;
  while (CP < C) CP <<= 1;
  const int Q = NT / CP;
  const int tid = threadIdx.x, c = tid & (CP - 1), q = tid / CP;
  const int cc = c < C ? c : C - 1;
  double n1 = 0.0, s1 = 0.0, ss1 = 0.0, m1 = 0.0, n2 = 0.0, s2 = 0.0, ss2 = 0.0, m2 = 0.0;
  const int nmax = b1.nrec > b2.nrec ? b1.nrec : b2.nrec;
  for (int j0 = q; j0 < nmax; j0 += 16 * Q) {
    float2 v[16], w[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int j = j0 + u * Q;
      v[u] = b1.rec[(long)(j < b1.nrec ? j : b1.nrec - 1) * C + cc];
      w[u] = b2.rec[(long)(j < b2.nrec ? j : b2.nrec - 1) * C + cc];
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int j = j0 + u * Q;
      asm volatile("" : "+v"(v[u].x), "+v"(v[u].y), "+v"(w[u].x), "+v"(w[u].y));
      const long l1 = rows - (long)j * b1.rpr, l2 = rows - (long)j * b2.rpr;
      const double a = j < b1.nrec ? (double)(l1 < b1.rpr ? l1 : b1.rpr) : 0.0, b = j < b2.nrec ? (double)(l2 < b2.rpr ? l2 : b2.rpr) : 0.0;
      const double x = (double)v[u].x, y = (double)w[u].x;
      n1 += a; s1 += a * x; ss1 += a * x * x; m1 += j < b1.nrec ? (double)v[u].y : 0.0;
      n2 += b; s2 += b * y; ss2 += b * y * y; m2 += j < b2.nrec ? (double)w[u].y : 0.0;
    }
  }
  red1[(q * 4 + 0) * CP + c] = n1; red1[(q * 4 + 1) * CP + c] = s1; red1[(q * 4 + 2) * CP + c] = ss1; red1[(q * 4 + 3) * CP + c] = m1;
  red2[(q * 4 + 0) * CP + c] = n2; red2[(q * 4 + 1) * CP + c] = s2; red2[(q * 4 + 2) * CP + c] = ss2; red2[(q * 4 + 3) * CP + c] = m2;
}

// The per-channel parameters a thread of [t0, t0 + C) needs in bn_finish, requested BEFORE the record gather so that they travel in
// the same memory round trip (fetched inside bn_finish they were one more dependent ~2 us trip for every workgroup, and workgroup 0's
// running-statistics read-modify-write one more behind its state stores).
struct BnPre { float g, b, rm, rv; };
__device__ __forceinline__ BnPre bn_preload(const BnRefD& bn, int C, int t0, bool first_wg) {
  int ch = (int)threadIdx.x - t0;
  ch = ch < 0 ? 0 : (ch < C ? ch : C - 1);
  BnPre p;
  p.g = bn.gamma[ch]; p.b = bn.beta[ch];
  p.rm = 0.f; p.rv = 0.f;
  if (first_wg && bn.rmean) { p.rm = bn.rmean[ch]; p.rv = bn.rvar[ch]; }
  return p;
}

// threads [t0, t0 + C) finish channel tid - t0 (a barrier lies between bn_gather and this); stores only, no global load
template <int NT>
__device__ __forceinline__ void bn_finish(const BnRefD& bn, int C, const double* red, float* st, bool first_wg, int t0, BnPre pre) {
  int CP = 32;
  while (CP < C) CP <<= 1;
  const int Q = NT / CP;
  const int ch = (int)threadIdx.x - t0;
  asm volatile("" : "+v"(pre.g), "+v"(pre.b), "+v"(pre.rm), "+v"(pre.rv));
  if (ch >= 0 && ch < C) {
    double N = 0.0, S = 0.0, SS = 0.0, M2 = 0.0;
    for (int g = 0; g < Q; ++g) {
      N += red[(g * 4 + 0) * CP + ch]; S += red[(g * 4 + 1) * CP + ch]; SS += red[(g * 4 + 2) * CP + ch]; M2 += red[(g * 4 + 3) * CP + ch];
    }
    const double mean = S / N;
    double m2 = M2 + SS - N * mean * mean;
    m2 = m2 > 0.0 ? m2 : 0.0;
    const double var = m2 / N;
    const float invstd = (float)(1.0 / sqrt(var + (double)bn.eps));
    const float a = pre.g * invstd, b = pre.b;
    st[ch] = (float)mean; st[C + ch] = a; st[2 * C + ch] = b; st[3 * C + ch] = invstd;
    if (first_wg) {
      if (bn.state) { bn.state[ch] = (float)mean; bn.state[C + ch] = invstd; bn.state[2 * C + ch] = a; bn.state[3 * C + ch] = b; }
      if (bn.rmean) {
        bn.rmean[ch] = (1.f - bn.momentum) * pre.rm + bn.momentum * (float)mean;
        const double unbiased = N > 1.0 ? m2 / (N - 1.0) : var;
        bn.rvar[ch] = (1.f - bn.momentum) * pre.rv + bn.momentum * (float)unbiased;
      }
    }
  }
}

template <int NT>
__device__ __forceinline__ void bn_from_records(const BnRefD& bn, int C, long rows, double* red, float* st, bool first_wg) {
  const BnPre pre = bn_preload(bn, C, 0, first_wg);
  bn_gather<NT>(bn, C, rows, red);
  __syncthreads();
  bn_finish<NT>(bn, C, red, st, first_wg, 0, pre);
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
