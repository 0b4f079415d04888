// Anchor ("voxel") grouping of UpperNetwlocal: for each of the 27 grid anchors take the 8 nearest points of
// the frame (reference Net/Upper_Net.py:10-32 square_distance, 54-72 point_ball_set, 100-119 AnchorGrouping).
//
// One workgroup per frame.  The frame's xyz (N x 3) and the 27 x N distance matrix live in LDS; the top-8
// selection is a rank count over LDS (rank = number of strictly smaller keys + equal keys with a lower index,
// i.e. a stable ascending sort), so the int64 group indices are exact and deterministic.  Distances follow
// the reference's rounding: dot = fma(a2,b2, fma(a1,b1, a0*b0)) (the CPU sgemm order for K=3), norms as
// (x*x + y*y) + z*z, d = (-2*dot + |a|^2) + |p|^2, +inf where the point's xyz is exactly 0 (quirk Q5) --
// checked bit for bit against tests/golden/g2_grouping.npz.
#include "common.h"

#define NA 27
#define NS 8

__device__ __forceinline__ float sq3_nofma(float x, float y, float z) {
  return __fadd_rn(__fadd_rn(__fmul_rn(x, x), __fmul_rn(y, y)), __fmul_rn(z, z));
}

// xf [F, N, ldx] (cols 0:3 xyz, cols 3:3+D features); anchors [27,3]
// idx int64 [F,27,8]; grouped [F*27*8, 6+D] = (anchor, xyz-anchor, features); dist_out [F,27,N] optional
__global__ __launch_bounds__(256) void anchor_group_kernel(const float* __restrict__ xf, long ldx, int N, int D,
                                                           const float* __restrict__ anchors, long long* __restrict__ idx,
                                                           float* __restrict__ grouped, float* __restrict__ dist_out) {
  extern __shared__ float sm[];
  float* px = sm;                 // [N][3]
  float* pn = sm + 3 * N;         // [N] squared norms (inf-masked flag folded into dist below)
  float* dist = sm + 4 * N;       // [27][N]
  const long f = blockIdx.x;
  const float* xfr = xf + f * (long)N * ldx;
  for (int p = threadIdx.x; p < N; p += blockDim.x) {
    float x = xfr[(long)p * ldx], y = xfr[(long)p * ldx + 1], z = xfr[(long)p * ldx + 2];
    px[p * 3] = x; px[p * 3 + 1] = y; px[p * 3 + 2] = z;
    pn[p] = sq3_nofma(x, y, z);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < NA * N; i += blockDim.x) {
    const int a = i / N, p = i - a * N;
    const float ax = anchors[a * 3], ay = anchors[a * 3 + 1], az = anchors[a * 3 + 2];
    const float x = px[p * 3], y = px[p * 3 + 1], z = px[p * 3 + 2];
    float dot = __fmaf_rn(az, z, __fmaf_rn(ay, y, __fmul_rn(ax, x)));
    float d = __fadd_rn(__fadd_rn(__fmul_rn(-2.0f, dot), sq3_nofma(ax, ay, az)), pn[p]);
    if (x == 0.f && y == 0.f && z == 0.f) d = INFINITY;
    dist[i] = d;
    if (dist_out) dist_out[f * (long)NA * N + i] = d;
  }
  __syncthreads();
  const int W = 6 + D;
  for (int i = threadIdx.x; i < NA * N; i += blockDim.x) {
    const int a = i / N, p = i - a * N;
    const float* da = dist + a * N;
    const float k = da[p];
    int rank = 0;
    for (int q = 0; q < N; ++q) {
      float kq = da[q];
      rank += (kq < k) || (kq == k && q < p);
    }
    if (rank < NS) {
      const long slot = (f * NA + a) * NS + rank;
      idx[slot] = p;
      float* g = grouped + slot * W;
      const float ax = anchors[a * 3], ay = anchors[a * 3 + 1], az = anchors[a * 3 + 2];
      g[0] = ax; g[1] = ay; g[2] = az;
      g[3] = px[p * 3] - ax; g[4] = px[p * 3 + 1] - ay; g[5] = px[p * 3 + 2] - az;
      const float* src = xfr + (long)p * ldx + 3;
      for (int c = 0; c < D; ++c) g[6 + c] = src[c];
    }
  }
}

// dxf[f, p, 0:3] += sum over slots with idx==p of dgrouped[slot, 3:6];  dxf[f, p, 3:3+D] += dgrouped[slot, 6:6+D]
__global__ __launch_bounds__(256) void anchor_group_bwd_kernel(const float* __restrict__ dgrouped, const long long* __restrict__ idx,
                                                               int N, int D, float* __restrict__ dxf, long lddx) {
  __shared__ int sidx[NA * NS];
  const long f = blockIdx.x;
  for (int s = threadIdx.x; s < NA * NS; s += blockDim.x) sidx[s] = (int)idx[f * NA * NS + s];
  __syncthreads();
  const int W = 6 + D, C = 3 + D;
  for (int i = threadIdx.x; i < N * C; i += blockDim.x) {   // one (point, channel) per thread: fixed slot order -> deterministic
    const int p = i / C, c = i - p * C;
    const int gc = c < 3 ? 3 + c : 3 + c;                   // grouped column: xyz-offset part 3..5, features 6.. == 3 + c
    float acc = 0.f;
    for (int s = 0; s < NA * NS; ++s)
      if (sidx[s] == p) acc += dgrouped[(f * NA * NS + s) * (long)W + gc];
    dxf[(f * N + p) * lddx + c] += acc;
  }
}

extern "C" int mmego_anchor_group(void* stream, const float* xf, long ldx, long F, int N, int D, const float* anchors,
                                  long long* idx, float* grouped, float* dist_out) {
  MMEGO_REQUIRE(xf && anchors && idx && grouped && F > 0 && N >= NS && N <= 1024 && D >= 0);
  size_t lds = (size_t)(4 * N + NA * N) * sizeof(float);
  hipLaunchKernelGGL(anchor_group_kernel, dim3((unsigned)F), dim3(256), lds, (hipStream_t)stream, xf, ldx, N, D, anchors, idx,
                     grouped, dist_out);
  MMEGO_LAUNCH_CHECK();
  return MMEGO_OK;
}

extern "C" int mmego_anchor_group_backward(void* stream, const float* dgrouped, const long long* idx, long F, int N, int D,
                                           float* dxf, long lddx) {
  MMEGO_REQUIRE(dgrouped && idx && dxf && F > 0 && N > 0 && D >= 0);
  hipLaunchKernelGGL(anchor_group_bwd_kernel, dim3((unsigned)F), dim3(256), 0, (hipStream_t)stream, dgrouped, idx, N, D, dxf, lddx);
  MMEGO_LAUNCH_CHECK();
  return MMEGO_OK;
}
