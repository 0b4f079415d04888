// Per-frame pooling / attention / graph kernels of Upper_Net and Lower_Net (fp32).
//   attn_pool   : Linear(C,1) score, softmax over the group's points, weighted sum (Upper_Net.py:285-301,
//                 163-177; IMU_Net.py:79-80)                      -- one workgroup per group, tile in LDS-free
//                 two-pass form (scores by wave reductions, then a coalesced weighted column sum)
//   group_sum   : sum / mean over the points of a group (Lower_Net.py:112-115: the degenerate Q6 gate, avg-pool)
//   cross_attn  : softmax(Q K^T / 8) V with 64 query points and 15 joint keys per frame (Lower_Net.py:105-109);
//                 K, V (and for backward Q, dO) staged in LDS, 4 lanes per query row
//   graph_dA    : gradient of einsum('nkctv,kvw->nctw') wrt the (K,15,15) adjacency (GCN.py:62)
#include "common.h"

// X [G, P, C]; w [C]; b scalar ptr -> vec [G, C], attn [G, P]
__global__ __launch_bounds__(256) void attn_pool_fwd_kernel(const float* __restrict__ X, const float* __restrict__ w,
                                                            const float* __restrict__ bptr, int P, int C,
                                                            float* __restrict__ vec, float* __restrict__ attn) {
  extern __shared__ float sc[];  // [P] scores -> weights
  __shared__ float red[8];
  const long g = blockIdx.x;
  const float* Xg = X + g * (long)P * C;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const float b = bptr ? bptr[0] : 0.f;
  for (int p = wave; p < P; p += nw) {
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += Xg[(long)p * C + c] * w[c];
    s = wave_sum(s);
    if (lane == 0) sc[p] = s + b;
  }
  __syncthreads();
  float m = -INFINITY;
  for (int p = threadIdx.x; p < P; p += blockDim.x) m = fmaxf(m, sc[p]);
  m = wave_max(m);
  if (lane == 0) red[wave] = m;
  __syncthreads();
  m = red[0];
  for (int i = 1; i < nw; ++i) m = fmaxf(m, red[i]);
  __syncthreads();
  float sum = 0.f;
  for (int p = threadIdx.x; p < P; p += blockDim.x) {
    float e = expf(sc[p] - m);
    sc[p] = e;
    sum += e;
  }
  sum = wave_sum(sum);
  if (lane == 0) red[wave] = sum;
  __syncthreads();
  sum = 0.f;
  for (int i = 0; i < nw; ++i) sum += red[i];
  const float inv = 1.0f / sum;
  for (int p = threadIdx.x; p < P; p += blockDim.x) {
    float a = sc[p] * inv;
    attn[g * P + p] = a;
  }
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float acc = 0.f;
    for (int p = 0; p < P; ++p) acc += Xg[(long)p * C + c] * (sc[p] * inv);
    vec[g * C + c] = acc;
  }
}

// Copy a tile of n4 f32x4 pieces from global memory to LDS with 256 threads: ALL of the tile's loads in flight before the first
// LDS store, and no branch in between.  NQ (pieces per thread) is a compile-time bound and the index is clamped, not predicated:
// written as `if (i < n4) v[q] = src[i]` every load got its own branch and its own s_waitcnt vmcnt(0) -- NQ dependent round
// trips instead of one (the clamped lanes re-read and re-write the tile's last piece: same value, harmless).
template <int NQ>
__device__ __forceinline__ void tile_to_lds(const float* __restrict__ src, float* dst, int n4, int tid) {
  f32x4 v[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) v[q] = reinterpret_cast<const f32x4*>(src)[min(tid + 256 * q, n4 - 1)];
#pragma unroll
  for (int q = 0; q < NQ; ++q) reinterpret_cast<f32x4*>(dst)[min(tid + 256 * q, n4 - 1)] = v[q];
}
__device__ __forceinline__ void tile_to_lds_any(const float* __restrict__ src, float* dst, int n4, int tid) {
  const int nq = (n4 + 255) / 256;                 // (uniform)
  if (nq <= 2) tile_to_lds<2>(src, dst, n4, tid);
  else if (nq <= 4) tile_to_lds<4>(src, dst, n4, tid);
  else if (nq <= 8) tile_to_lds<8>(src, dst, n4, tid);
  else if (nq <= 16) tile_to_lds<16>(src, dst, n4, tid);
  else tile_to_lds<24>(src, dst, n4, tid);
}

// Same, with the group's [P, C] tile staged once in LDS (16-B loads, every byte read from HBM exactly once): used when the tile
// fits (P * C * 4 <= 96 KB; IMU_Net's pooling: 20 x 1024 = 80 KB, the PointNets': 128 x 64 = 32 KB).  C % 4 == 0.
__global__ __launch_bounds__(256) void attn_pool_fwd_lds_kernel(const float* __restrict__ X, const float* __restrict__ w,
                                                                const float* __restrict__ bptr, int P, int C,
                                                                float* __restrict__ vec, float* __restrict__ attn) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* Xs = lds;                       // [P][C]
  float* sc = lds + (long)P * C;         // [P]
  __shared__ float red[8];
  __shared__ __attribute__((aligned(16))) float part[1024];     // partial scores [TPP][P] / partial sums [PG][C]
  const long g = blockIdx.x;
  const float* Xg = X + g * (long)P * C;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
  const int n4 = P * C / 4;
  // the score weights ride with the tile's loads into LDS: read from global memory inside the dot-product loops they were one
  // (cached, but waited-for) load per multiply-add
  __shared__ float wsh[1024];
  const bool wl = C <= 1024;
  float wreg[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) wreg[u] = w[min(tid + 256 * u, C - 1)];
  tile_to_lds_any(Xg, Xs, n4, tid);        // all of the tile's loads in flight at once, then the LDS stores
  const float b = bptr ? bptr[0] : 0.f;
#pragma unroll
  for (int u = 0; u < 4; ++u) if (tid + 256 * u < C && wl) wsh[tid + 256 * u] = wreg[u];
  __syncthreads();
  const float* const wv = wl ? wsh : w;
  // scores: TPP threads per point, each over a contiguous slice of the channels, started at a point-dependent offset so that
  // neighbouring threads hit different LDS banks (the points' rows are C floats apart); partials summed in fixed order
  int TPP = 1;
  while (TPP * 2 * P <= 256 && TPP * 2 * P <= 1024 && (C % (TPP * 2)) == 0) TPP *= 2;
  const bool wide = P * TPP <= 1024 && P <= 1024;
  if (wide) {
    const int len = C / TPP;
    for (int q = tid; q < P * TPP; q += blockDim.x) {
      const int p = q % P, pt = q / P, c0 = pt * len;
      float s = 0.f;
      for (int i = 0; i < len; ++i) {
        int c = i + p;
        c = c0 + (c >= len ? c % len : c);
        s += Xs[p * C + c] * wv[c];
      }
      part[pt * P + p] = s;
    }
    __syncthreads();
    for (int p = tid; p < P; p += blockDim.x) {
      float s = part[p];
      for (int pt = 1; pt < TPP; ++pt) s += part[pt * P + p];
      sc[p] = s + b;
    }
  } else {
    for (int p = wave; p < P; p += nw) {
      float s = 0.f;
      for (int c = lane; c < C; c += 64) s += Xs[p * C + c] * wv[c];
      s = wave_sum(s);
      if (lane == 0) sc[p] = s + b;
    }
  }
  __syncthreads();
  float m = -INFINITY;
  for (int p = threadIdx.x; p < P; p += blockDim.x) m = fmaxf(m, sc[p]);
  m = wave_max(m);
  if (lane == 0) red[wave] = m;
  __syncthreads();
  m = red[0];
  for (int i = 1; i < nw; ++i) m = fmaxf(m, red[i]);
  __syncthreads();
  float sum = 0.f;
  for (int p = threadIdx.x; p < P; p += blockDim.x) {
    float e = expf(sc[p] - m);
    sc[p] = e;
    sum += e;
  }
  sum = wave_sum(sum);
  if (lane == 0) red[wave] = sum;
  __syncthreads();
  sum = 0.f;
  for (int i = 0; i < nw; ++i) sum += red[i];
  const float inv = 1.0f / sum;
  for (int p = threadIdx.x; p < P; p += blockDim.x) attn[g * P + p] = sc[p] * inv;
  // weighted sum: with few channels, PG point groups share the work of a channel (256 / C threads per channel)
  const int PG = (C <= 128 && (256 % C) == 0) ? 256 / C : 1;
  if (PG > 1) {
    const int c = tid % C, pg = tid / C;
    float acc = 0.f;
    for (int p = pg; p < P; p += PG) acc += Xs[p * C + c] * (sc[p] * inv);
    part[pg * C + c] = acc;
    __syncthreads();
    if (tid < C) {
      float a = part[tid];
      for (int q = 1; q < PG; ++q) a += part[q * C + tid];
      vec[g * C + tid] = a;
    }
  } else {
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
      float acc = 0.f;
      for (int p = 0; p < P; ++p) acc += Xs[p * C + c] * (sc[p] * inv);
      vec[g * C + c] = acc;
    }
  }
}

// Few rows x many channels (IMU_Net's pooling over the 20 samples of a frame, C = 1024): the whole [P, C] tile of a group lives
// in REGISTERS -- thread t holds channels 4t..4t+3 of every row (P <= 32 f32x4) -- so HBM is read exactly once with all loads
// in flight together, and no large LDS tile limits the workgroups per CU.  C % 4 == 0, C <= 4 * blockDim.
template <int PMAX>
__global__ __launch_bounds__(256) void attn_pool_fwd_reg_kernel(const float* __restrict__ X, const float* __restrict__ w,
                                                                const float* __restrict__ bptr, int P, int C,
                                                                float* __restrict__ vec, float* __restrict__ attn) {
  __shared__ float part[4][PMAX];
  __shared__ float sc[PMAX];
  const long g = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool act = 4 * tid < C;
  const float* Xg = X + g * (long)P * C + 4 * tid;
  f32x4 xr[PMAX];
#pragma unroll
  for (int p = 0; p < PMAX; ++p) xr[p] = (act && p < P) ? *reinterpret_cast<const f32x4*>(Xg + (long)p * C) : (f32x4){0.f, 0.f, 0.f, 0.f};
  const f32x4 w4 = act ? *reinterpret_cast<const f32x4*>(w + 4 * tid) : (f32x4){0.f, 0.f, 0.f, 0.f};
  const float b = bptr ? bptr[0] : 0.f;
#pragma unroll
  for (int p = 0; p < PMAX; ++p) {
    float s = (xr[p].x * w4.x + xr[p].y * w4.y) + (xr[p].z * w4.z + xr[p].w * w4.w);
    s = wave_sum(s);
    if (lane == 0) part[wave][p] = s;
  }
  __syncthreads();
  if (tid < PMAX) sc[tid] = tid < P ? ((part[0][tid] + part[1][tid]) + (part[2][tid] + part[3][tid])) + b : -INFINITY;
  __syncthreads();
  float m = -INFINITY;
#pragma unroll
  for (int p = 0; p < PMAX; ++p) m = fmaxf(m, sc[p]);
  float e[PMAX], sum = 0.f;
#pragma unroll
  for (int p = 0; p < PMAX; ++p) { e[p] = p < P ? expf(sc[p] - m) : 0.f; sum += e[p]; }
  const float inv = 1.0f / sum;
  if (tid < P) attn[g * P + tid] = e[tid < PMAX ? tid : 0] * inv;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int p = 0; p < PMAX; ++p) acc += xr[p] * (e[p] * inv);
  if (act) *reinterpret_cast<f32x4*>(vec + g * C + 4 * tid) = acc;
}

// dvec [G, C] -> dX [G, P, C], partial_dw [G, C], partial_db [G]
__global__ __launch_bounds__(256) void attn_pool_bwd_kernel(const float* __restrict__ X, const float* __restrict__ w,
                                                            const float* __restrict__ attn,
                                                            const float* __restrict__ dvec, int P, int C,
                                                            float* __restrict__ dX, float* __restrict__ pdw,
                                                            float* __restrict__ pdb) {
  extern __shared__ float sh[];  // [P] da -> ds
  __shared__ float red[8];
  const long g = blockIdx.x;
  const float* Xg = X + g * (long)P * C;
  const float* dv = dvec + g * C;
  const float* ag = attn + g * P;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  for (int p = wave; p < P; p += nw) {
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += Xg[(long)p * C + c] * dv[c];
    s = wave_sum(s);
    if (lane == 0) sh[p] = s;
  }
  __syncthreads();
  float dot = 0.f;
  for (int p = threadIdx.x; p < P; p += blockDim.x) dot += ag[p] * sh[p];
  dot = wave_sum(dot);
  if (lane == 0) red[wave] = dot;
  __syncthreads();
  dot = 0.f;
  for (int i = 0; i < nw; ++i) dot += red[i];
  __syncthreads();
  float dbs = 0.f;
  for (int p = threadIdx.x; p < P; p += blockDim.x) {
    float ds = ag[p] * (sh[p] - dot);
    sh[p] = ds;
    dbs += ds;
  }
  dbs = wave_sum(dbs);
  if (lane == 0) red[wave] = dbs;
  __syncthreads();
  if (threadIdx.x == 0) {
    float s = 0.f;
    for (int i = 0; i < nw; ++i) s += red[i];
    pdb[g] = s;
  }
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    const float wc = w[c], dvc = dv[c];
    float acc = 0.f;
    for (int p = 0; p < P; ++p) {
      float x = Xg[(long)p * C + c];
      acc += sh[p] * x;
      dX[(g * P + p) * (long)C + c] = ag[p] * dvc + sh[p] * wc;
    }
    pdw[g * C + c] = acc;
  }
}

// Same with the group's [P, C] tile staged once in LDS (tile <= 64 KB, C % 4 == 0: the PointNets' pooling, 128 x 64): every
// thread works in every phase (the plain kernel's last loop runs on C of its 256 threads and walks the P rows serially).
__global__ __launch_bounds__(256) void attn_pool_bwd_lds_kernel(const float* __restrict__ X, const float* __restrict__ w,
                                                                const float* __restrict__ attn, const float* __restrict__ dvec,
                                                                int P, int C, float* __restrict__ dX, float* __restrict__ pdw,
                                                                float* __restrict__ pdb) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* Xs = lds;                        // [P][C]
  float* sh = lds + (long)P * C;          // [P]  da -> ds
  float* ags = sh + P;                    // [P]
  float* part = ags + P;                  // [PG][C] partial dw
  __shared__ float red[8];
  const long g = blockIdx.x;
  const float* Xg = X + g * (long)P * C;
  const float* dv = dvec + g * C;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
  const int n4 = P * C / 4;
  // (the group's d(vec) row into LDS with the tile: see attn_pool_fwd_lds_kernel)
  __shared__ float dvs[1024];
  const bool dl = C <= 1024;
  float dreg[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) dreg[u] = dv[min(tid + 256 * u, C - 1)];
  tile_to_lds_any(Xg, Xs, n4, tid);        // tile <= 64 KB: all of its loads in flight at once, then the LDS stores
  for (int p = tid; p < P; p += blockDim.x) ags[p] = attn[g * P + p];
#pragma unroll
  for (int u = 0; u < 4; ++u) if (tid + 256 * u < C && dl) dvs[tid + 256 * u] = dreg[u];
  __syncthreads();
  const float* const dvv = dl ? dvs : dv;
  // da[p] = X[p, :] . dvec: TPP threads per row, each over a contiguous channel slice entered at a row-dependent offset
  // (bank-conflict free), partials summed in fixed order -- as in attn_pool_fwd_lds_kernel
  __shared__ float ps[1024];
  int TPP = 1;
  while (TPP * 2 * P <= 256 && (C % (TPP * 2)) == 0) TPP *= 2;
  if (P * TPP <= 1024) {
    const int len = C / TPP;
    for (int q = tid; q < P * TPP; q += blockDim.x) {
      const int p = q % P, pt = q / P, c0 = pt * len;
      float s = 0.f;
      for (int i = 0; i < len; ++i) {
        int c = i + p;
        c = c0 + (c >= len ? c % len : c);
        s += Xs[p * C + c] * dvv[c];
      }
      ps[pt * P + p] = s;
    }
    __syncthreads();
    for (int p = tid; p < P; p += blockDim.x) {
      float s = ps[p];
      for (int pt = 1; pt < TPP; ++pt) s += ps[pt * P + p];
      sh[p] = s;
    }
  } else {
    for (int p = wave; p < P; p += nw) {
      float s = 0.f;
      for (int c = lane; c < C; c += 64) s += Xs[p * C + c] * dvv[c];
      s = wave_sum(s);
      if (lane == 0) sh[p] = s;
    }
  }
  __syncthreads();
  float dot = 0.f;
  for (int p = tid; p < P; p += blockDim.x) dot += ags[p] * sh[p];
  dot = wave_sum(dot);
  if (lane == 0) red[wave] = dot;
  __syncthreads();
  dot = 0.f;
  for (int i = 0; i < nw; ++i) dot += red[i];
  __syncthreads();
  float dbs = 0.f;
  for (int p = tid; p < P; p += blockDim.x) {
    float ds = ags[p] * (sh[p] - dot);
    sh[p] = ds;
    dbs += ds;
  }
  dbs = wave_sum(dbs);
  if (lane == 0) red[wave] = dbs;
  __syncthreads();
  if (tid == 0) {
    float s = 0.f;
    for (int i = 0; i < nw; ++i) s += red[i];
    pdb[g] = s;
  }
  // dX and the per-group dw: thread (cx, pg) walks rows pg, pg + PG, ... of columns cx, cx + TCc, ...
  const int TCc = C < 256 ? C : 256, PG = 256 / TCc;
  const int cx = tid % TCc, pg = tid / TCc;
  if (pg < PG) {
    for (int c = cx; c < C; c += TCc) {
      const float wc = w[c], dvc = dv[c];
      float acc = 0.f;
      for (int p = pg; p < P; p += PG) {
        acc += sh[p] * Xs[p * C + c];
        dX[(g * P + p) * (long)C + c] = ags[p] * dvc + sh[p] * wc;
      }
      part[pg * C + c] = acc;
    }
  }
  __syncthreads();
  for (int c = tid; c < C; c += blockDim.x) {
    float a = 0.f;
    for (int q = 0; q < PG; ++q) a += part[q * C + c];
    pdw[g * C + c] = a;
  }
}

// Y[g, c] = scale * sum_p X[g, p, c]
__global__ __launch_bounds__(256) void group_sum_kernel(const float* __restrict__ X, int P, int C, float scale,
                                                        float* __restrict__ Y, long ldy, long G) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= G * C) return;
  long g = i / C;
  int c = (int)(i - g * C);
  const float* x = X + g * (long)P * C + c;
  float s = 0.f;
  for (int p0 = 0; p0 < P; p0 += 8) {          // eight rows' loads in flight, summed in row order
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = p0 + u < P ? x[(long)(p0 + u) * C] : 0.f;
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (p0 + u < P) s += v[u];
  }
  Y[g * ldy + c] = s * scale;
}

// Two group sums / two group broadcasts in one launch (Lower_Net's fusion module reduces the point features and the joint features
// of a frame side by side, Lower_Net.py:112-115): blocks [0, nb1) take the first problem, the rest the second.
struct GroupSeg { const float* X; float* Y; long ld; int P, C; float scale; };
__global__ __launch_bounds__(256) void group_sum2_kernel(GroupSeg a, GroupSeg b, long G, int nb1) {
  const bool second = (int)blockIdx.x >= nb1;
  const GroupSeg& s_ = second ? b : a;
  const long i = (long)(blockIdx.x - (second ? nb1 : 0)) * blockDim.x + threadIdx.x;
  if (i >= G * s_.C) return;
  const long g = i / s_.C;
  const int c = (int)(i - g * s_.C);
  const float* x = s_.X + g * (long)s_.P * s_.C + c;
  float s = 0.f;
  for (int p0 = 0; p0 < s_.P; p0 += 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = p0 + u < s_.P ? x[(long)(p0 + u) * s_.C] : 0.f;
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (p0 + u < s_.P) s += v[u];
  }
  s_.Y[g * s_.ld + c] = s * s_.scale;
}
// dX[g, p, c] = scale * dY[g, c] for two problems (X = dY with row stride ld, Y = dX contiguous)
__global__ __launch_bounds__(256) void group_bcast2_kernel(GroupSeg a, GroupSeg b, long G, int nb1) {
  const bool second = (int)blockIdx.x >= nb1;
  const GroupSeg& s_ = second ? b : a;
  const long total = G * s_.P * s_.C;
  const long nb = second ? (long)gridDim.x - nb1 : nb1;
  for (long i = (long)(blockIdx.x - (second ? nb1 : 0)) * blockDim.x + threadIdx.x; i < total; i += nb * blockDim.x) {
    const int c = (int)(i % s_.C);
    const long g = i / ((long)s_.P * s_.C);
    s_.Y[i] = s_.scale * s_.X[g * s_.ld + c];
  }
}

// dX[g, p, c] (+)= scale * dY[g, c]
__global__ __launch_bounds__(256) void group_bcast_kernel(const float* __restrict__ dY, long lddy, int P, int C,
                                                          float scale, float* __restrict__ dX, long G,
                                                          int accumulate) {
  long total = G * P * C;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int c = (int)(i % C);
    long g = i / ((long)P * C);
    float v = scale * dY[g * lddy + c];
    dX[i] = accumulate ? dX[i] + v : v;
  }
}

// ---- cross attention: Q [F,64,64]; K,V [F,15,64] -> O [F,64,64] written at ldo, P [F,64,15] -------------
#define NQ 64
#define NK 15
#define DH 64

__global__ __launch_bounds__(256) void cross_attn_fwd_kernel(const float* __restrict__ Q, const float* __restrict__ K,
                                                             const float* __restrict__ V, float scale,
                                                             float* __restrict__ O, long ldo, float* __restrict__ Pout,
                                                             long ldkv) {
  __shared__ float Ks[NK][DH + 1], Vs[NK][DH + 1];
  const long f = blockIdx.x;
  for (int i = threadIdx.x; i < NK * DH; i += 256) {     // (ldkv: row stride of K and V -- column halves of one [.., 2 DH] buffer)
    Ks[i / DH][i % DH] = K[(f * NK + i / DH) * ldkv + (i % DH)];
    Vs[i / DH][i % DH] = V[(f * NK + i / DH) * ldkv + (i % DH)];
  }
  __syncthreads();
  const int p = threadIdx.x >> 2, sub = threadIdx.x & 3;  // 4 lanes per query row, 16 channels each
  const float* q = Q + (f * NQ + p) * DH + sub * 16;
  float qr[16];
  for (int c = 0; c < 16; ++c) qr[c] = q[c];
  float s[NK];
  float m = -INFINITY;
  for (int j = 0; j < NK; ++j) {
    float d = 0.f;
    for (int c = 0; c < 16; ++c) d += qr[c] * Ks[j][sub * 16 + c];
    d += __shfl_xor(d, 1, 64);
    d += __shfl_xor(d, 2, 64);
    s[j] = d * scale;
    m = fmaxf(m, s[j]);
  }
  float sum = 0.f;
  for (int j = 0; j < NK; ++j) { s[j] = expf(s[j] - m); sum += s[j]; }
  const float inv = 1.0f / sum;
  for (int j = 0; j < NK; ++j) s[j] *= inv;
  if (sub == 0)
    for (int j = 0; j < NK; ++j) Pout[(f * NQ + p) * NK + j] = s[j];
  float* o = O + (f * NQ + p) * ldo + sub * 16;
  for (int c = 0; c < 16; ++c) {
    float acc = 0.f;
    for (int j = 0; j < NK; ++j) acc += s[j] * Vs[j][sub * 16 + c];
    o[c] = acc;
  }
}

// The same forward on the fp32 matrix pipe (r06; v_mfma_f32_16x16x4_f32: exact fp32 products, fp32 accumulation), no LDS at all.
// The scalar kernel above does 480 broadcast LDS reads and 480 FMAs per lane and frame row (402 us for 32 768 frames at config 5, the
// LDS queue full); here a wave owns 16 queries of the frame and computes both products TRANSPOSED, so that every operand is a plain
// contiguous read and no value changes lanes between the two products:
//   S^T[key][query] = K Q^T: lane (m, g) supplies K[key m][16 g + kk] and Q[query m][16 g + kk] at step kk = 0..15 -- each lane reads 16
//     contiguous channels of its key / query row (four 16-byte loads each); result lane (query c, g), register i: key 4 g + i;
//   softmax over a query's 15 keys = over 4 registers and the 4 lane groups g (two shuffles per reduction); key 15 is padding (-inf);
//   O^T[channel][query] = V^T P^T: the B operand at step kk is register kk of the softmax result as it stands (key 4 g + kk), the A
//     operand V[key 4 g + kk][16 ct + m]; result lane (query c, g), register i: channel 16 ct + 4 g + i -- one 16-byte store per tile.
// Requires 16-byte aligned Q, K, V, O and ldkv, ldo multiples of 4 (the launcher falls back to the scalar kernel otherwise).
// POOLED (eval forwards: nothing reads O or P but the sum over the frame's 64 queries that follows, Lower_Net.py:131-133 with gate == 1):
// the O tiles meet in LDS and channel ch's thread adds the 64 queries in ascending order -- the order of group_sum2, so the pooled
// vector has the bits of the two-launch form -- and only [64] floats per frame are written (O + P: 0.66 of this kernel's 1.45 GB at
// config 5, and group_sum2 read the O half back).
// FUSEQ (with POOLED; eval forwards): the queries are computed here -- Q = X Wq^T + bq (Lower_Net.py:100: to_q on the 64-channel point
// features, a 64 x 64 weight) -- instead of being written by a product launch and read back (2 x 537 MB at config 5): Wq sits in LDS,
// Q^T[channel][query] = Wq X^T is one more transposed product (bias as its addend), and its result layout -- lane (query, g), tile ct,
// register i: channel 16 ct + 4 g + i -- is used as it stands: the K operand of S^T = K Q^T simply reads the same channels (four 16-byte
// loads 64 bytes apart instead of one 64-byte run).  Q is then `ldq` floats per row of X.
template <bool POOLED, bool FUSEQ>
__global__ __launch_bounds__(256) void cross_attn_fwd_mfma_kernel(const float* __restrict__ Q, const float* __restrict__ K,
                                                                  const float* __restrict__ V, float scale,
                                                                  float* __restrict__ O, long ldo, float* __restrict__ Pout,
                                                                  long ldkv, const float* __restrict__ Wq, const float* __restrict__ bq,
                                                                  long ldq, long nframes) {
  __shared__ __attribute__((aligned(16))) float osh[POOLED ? NQ * 68 : 4];
  __shared__ __attribute__((aligned(16))) float wsh[FUSEQ ? DH * 68 : 4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = lane & 15, g = lane >> 4;
  if (FUSEQ) {                                              // Wq [64][64] -> LDS rows of 68 floats, once per workgroup: 16 floats per thread
    const int r = threadIdx.x >> 2, c0 = (threadIdx.x & 3) * 16;
#pragma unroll
    for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4*>(wsh + r * 68 + c0 + 4 * j) = *reinterpret_cast<const f32x4*>(Wq + r * DH + c0 + 4 * j);
    __syncthreads();
  }
  // (the FUSEQ launch is persistent -- a workgroup walks frames blockIdx.x, + gridDim.x, ... with its copy of Wq; the others have one
  //  frame per workgroup)
  for (long f = blockIdx.x; f < nframes; f += gridDim.x) {
  const long qrow = f * NQ + 16 * wave + c;
  f32x4 qv[4], kv[4];
  if (FUSEQ) {
    const float* xp = Q + qrow * ldq + 16 * g;              // (Q = X here: the point features)
    f32x4 xv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) xv[j] = *reinterpret_cast<const f32x4*>(xp + 4 * j);
    const float* kp = K + (f * NK + min(c, NK - 1)) * ldkv + 4 * g;
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) kv[ct] = *reinterpret_cast<const f32x4*>(kp + 16 * ct);       // channels 16 ct + 4 g + i
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
      f32x4 acc = *reinterpret_cast<const f32x4*>(bq + 16 * ct + 4 * g);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const f32x4 w4 = *reinterpret_cast<const f32x4*>(wsh + (16 * ct + c) * 68 + 16 * g + 4 * j);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w4[e], xv[j][e], acc, 0, 0, 0);
      }
      qv[ct] = acc;
    }
  } else {
    const float* qp = Q + qrow * DH + 16 * g;
    const float* kp = K + (f * NK + min(c, NK - 1)) * ldkv + 16 * g;          // (lane group row 15: key 14 again; masked below)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      qv[j] = *reinterpret_cast<const f32x4*>(qp + 4 * j);
      kv[j] = *reinterpret_cast<const f32x4*>(kp + 4 * j);
    }
  }
  // V^T fragments: lane (m = c, g): V[key 4 g + kk][16 ct + c]
  float vv[4][4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    const float* vp = V + (f * NK + min(4 * g + kk, NK - 1)) * ldkv + c;
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) vv[ct][kk] = vp[16 * ct];
  }
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int e = 0; e < 4; ++e) s = __builtin_amdgcn_mfma_f32_16x16x4f32(kv[j][e], qv[j][e], s, 0, 0, 0);
  float sv[4], m = -INFINITY;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    sv[i] = (4 * g + i < NK) ? s[i] * scale : -INFINITY;
    m = fmaxf(m, sv[i]);
  }
  m = fmaxf(m, __shfl_xor(m, 16, 64));
  m = fmaxf(m, __shfl_xor(m, 32, 64));
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) { sv[i] = expf(sv[i] - m); sum += sv[i]; }
  sum += __shfl_xor(sum, 16, 64);
  sum += __shfl_xor(sum, 32, 64);
  const float inv = 1.0f / sum;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    sv[i] *= inv;
    if (!POOLED && 4 * g + i < NK) Pout[qrow * NK + 4 * g + i] = sv[i];
  }
  float* op = POOLED ? osh + (16 * wave + c) * 68 + 4 * g : O + qrow * ldo + 4 * g;
#pragma unroll
  for (int ct = 0; ct < 4; ++ct) {
    f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) o = __builtin_amdgcn_mfma_f32_16x16x4f32(vv[ct][kk], sv[kk], o, 0, 0, 0);
    *reinterpret_cast<f32x4*>(op + 16 * ct) = o;
  }
  if (POOLED) {
    __syncthreads();
    if (threadIdx.x < DH) {
      float sum_o = 0.f;
      for (int p0 = 0; p0 < NQ; p0 += 8) {             // (group_sum2_kernel's loop: 8 values fetched, added one by one in order)
        float v8[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v8[u] = osh[(p0 + u) * 68 + threadIdx.x];
#pragma unroll
        for (int u = 0; u < 8; ++u) sum_o += v8[u];
      }
      O[f * ldo + threadIdx.x] = sum_o * 1.0f;
    }
    __syncthreads();                                   // (osh is the next frame's)
  }
  }
}

// dO read at lddo; -> dQ [F,64,64], dK, dV [F,15,64]
__global__ __launch_bounds__(256) void cross_attn_bwd_kernel(const float* __restrict__ Q, const float* __restrict__ K,
                                                             const float* __restrict__ V, const float* __restrict__ Pm,
                                                             const float* __restrict__ dO, long lddo, float scale,
                                                             float* __restrict__ dQ, float* __restrict__ dK,
                                                             float* __restrict__ dV, long ldkv) {
  __shared__ float Ks[NK][DH + 1], Vs[NK][DH + 1];
  __shared__ float Qs[NQ][DH + 1], dOs[NQ][DH + 1];
  __shared__ float Ps[NQ][NK + 1], dSs[NQ][NK + 1];
  const long f = blockIdx.x;
  for (int i = threadIdx.x; i < NK * DH; i += 256) {     // (ldkv: row stride of K, V, dK and dV)
    Ks[i / DH][i % DH] = K[(f * NK + i / DH) * ldkv + (i % DH)];
    Vs[i / DH][i % DH] = V[(f * NK + i / DH) * ldkv + (i % DH)];
  }
  for (int i = threadIdx.x; i < NQ * DH; i += 256) {
    Qs[i / DH][i % DH] = Q[f * NQ * DH + i];
    dOs[i / DH][i % DH] = dO[(f * NQ + i / DH) * lddo + (i % DH)];
  }
  for (int i = threadIdx.x; i < NQ * NK; i += 256) Ps[i / NK][i % NK] = Pm[f * NQ * NK + i];
  __syncthreads();
  const int p = threadIdx.x >> 2, sub = threadIdx.x & 3;
  float dP[NK];
  float dot = 0.f;
  for (int j = 0; j < NK; ++j) {
    float d = 0.f;
    for (int c = 0; c < 16; ++c) d += dOs[p][sub * 16 + c] * Vs[j][sub * 16 + c];
    d += __shfl_xor(d, 1, 64);
    d += __shfl_xor(d, 2, 64);
    dP[j] = d;
    dot += Ps[p][j] * d;
  }
  for (int j = 0; j < NK; ++j) {
    dP[j] = Ps[p][j] * (dP[j] - dot) * scale;  // dS
    if (sub == 0) dSs[p][j] = dP[j];
  }
  float* dq = dQ + (f * NQ + p) * DH + sub * 16;
  for (int c = 0; c < 16; ++c) {
    float acc = 0.f;
    for (int j = 0; j < NK; ++j) acc += dP[j] * Ks[j][sub * 16 + c];
    dq[c] = acc;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < NK * DH; i += 256) {
    const int j = i / DH, c = i % DH;
    float av = 0.f, ak = 0.f;
    for (int pp = 0; pp < NQ; ++pp) {
      av += Ps[pp][j] * dOs[pp][c];
      ak += dSs[pp][j] * Qs[pp][c];
    }
    dV[(f * NK + j) * ldkv + c] = av;
    dK[(f * NK + j) * ldkv + c] = ak;
  }
}

// dA[k, v, w] = sum_{g, c} Z[g, v, k*C + c] * dY[g, w, c]     Z [G, V, K*C], dY [G, V, C]
// Each block stages FPB frames' Z and dY rows in LDS once and every thread owns one (k,v,w) entry;
// partial[blk][k*V*V + v*V + w] is then column-summed in fixed order by the caller (mmego_colsum kernels).
#define GDA_FPB 2   // frames per block: 512 frames -> 256 blocks (8 per block left 3 of 4 CUs idle and chained 8 load latencies)
// A / imp / dZ (optional, all three): the einsum's input gradient dZ[g, v, k*C + c] = sum_w (A . imp)[k, v, w] dY[g, w, c] from
// the same staged dY rows (graph_mix_kernel's backward arithmetic, gcn.hip: same sums in the same order) -- one launch for
// both gradients of the einsum.
__global__ __launch_bounds__(512) void graph_dA_partial_kernel(const float* __restrict__ Z, const float* __restrict__ dY, long G,
                                                               int V, int Kk, int C, float* __restrict__ partial,
                                                               const float* __restrict__ A, const float* __restrict__ imp,
                                                               float* __restrict__ dZ, long ldz, long lddz) {
  extern __shared__ float sm[];
  const int KC = Kk * C;
  float* zs = sm;                       // [V][KC + 1]
  float* ys = sm + V * (KC + 1);        // [V][C + 1]
  float* As = ys + V * (C + 1);         // [Kk][V][V]  A . importance (only with dZ)
  if (dZ)
    for (int i = threadIdx.x; i < Kk * V * V; i += blockDim.x) As[i] = A[i] * imp[i];
  const int nout = Kk * V * V;
  const int e = threadIdx.x;
  const int w = e % V, v = (e / V) % V, k = e / (V * V);
  float acc = 0.f;
  const long g0 = (long)blockIdx.x * GDA_FPB;
  for (long g = g0; g < g0 + GDA_FPB && g < G; ++g) {
    __syncthreads();
    for (int i = threadIdx.x; i < V * KC; i += blockDim.x) zs[(i / KC) * (KC + 1) + (i % KC)] = Z[(g * V + i / KC) * ldz + (i % KC)];
    for (int i = threadIdx.x; i < V * C; i += blockDim.x) ys[(i / C) * (C + 1) + (i % C)] = dY[g * V * C + i];
    __syncthreads();
    if (e < nout) {
      const float* zr = zs + v * (KC + 1) + k * C;
      const float* yr = ys + w * (C + 1);
      float a = 0.f;
      for (int c = 0; c < C; ++c) a += zr[c] * yr[c];
      acc += a;
    }
    if (dZ) {
      float* zf = dZ + g * V * lddz;
      for (int i = threadIdx.x; i < V * KC; i += blockDim.x) {
        const int row = i / KC, col = i - row * KC;
        const int kk = col / C, c = col - kk * C;
        float a = 0.f;
        for (int w2 = 0; w2 < V; ++w2) a += As[(kk * V + row) * V + w2] * ys[w2 * (C + 1) + c];
        zf[(long)row * lddz + col] = a;
      }
    }
  }
  if (e < nout) partial[(long)blockIdx.x * nout + e] = acc;
}

// out[b][c][r] = in[b][r][c]: 64 x 64 tiles through LDS (both the read and the write of a tile walk contiguous addresses; the
// one-element-per-thread form read with a stride of C floats: 268 us for 126 MB at config 5)
__global__ __launch_bounds__(256) void transpose_batched_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                                long Bn, int R, int C) {
  __shared__ float tile[64][65];
  const int tr = (R + 63) / 64, tc = (C + 63) / 64;
  const long ntile = Bn * tr * tc;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (long tI = blockIdx.x; tI < ntile; tI += gridDim.x) {
    const long b = tI / (tr * tc);
    const int q = (int)(tI - b * tr * tc), r0 = (q / tc) * 64, c0 = (q % tc) * 64;
    const float* src = in + b * (long)R * C;
    float* dst = out + b * (long)R * C;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int r = r0 + ty + 4 * i, c = c0 + tx;
      tile[ty + 4 * i][tx] = (r < R && c < C) ? src[(long)r * C + c] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int c = c0 + ty + 4 * i, r = r0 + tx;
      if (r < R && c < C) dst[(long)c * R + r] = tile[tx][ty + 4 * i];
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void mul_kernel(const float* a, const float* b, float* out, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) out[i] = a[i] * b[i];
}

__global__ __launch_bounds__(256) void add_kernel(const float* a, const float* b, float* out, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) out[i] = a[i] + b[i];
}

__global__ void inc_i64_kernel(long long* x, long n, unsigned long long* seed_ctr) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) x[i] += 1;
  if (i == 0 && seed_ctr) seed_ctr[0] = seed_ctr[0] * 6364136223846793005ULL + 1442695040888963407ULL;
}

static inline int ew_blocks(long total) {
  long b = (total + 255) / 256;
  return (int)(b > 2048 ? 2048 : (b < 1 ? 1 : b));
}

extern "C" int mmego_attn_pool_forward(void* stream, const float* X, const float* w, const float* b, long G, int P, int C,
                                       float* vec, float* attn) {
  MMEGO_REQUIRE(X && w && vec && attn && G > 0 && P > 0 && C > 0 && P <= 8192);
  const size_t tile = ((size_t)P * C + P) * sizeof(float);
  const bool vec_ok = (C % 4) == 0 && ((((uintptr_t)X) | ((uintptr_t)w) | ((uintptr_t)vec)) & 15) == 0;
  if (vec_ok && P <= 32 && C <= 1024) {            // the tile fits the register file of one workgroup
    if (P <= 20) hipLaunchKernelGGL((attn_pool_fwd_reg_kernel<20>), dim3((unsigned)G), dim3(256), 0, (hipStream_t)stream, X, w, b, P, C, vec, attn);
    else hipLaunchKernelGGL((attn_pool_fwd_reg_kernel<32>), dim3((unsigned)G), dim3(256), 0, (hipStream_t)stream, X, w, b, P, C, vec, attn);
    MMEGO_LAUNCH_CHECK();
    return MMEGO_OK;
  }
  if (vec_ok && tile <= 96 * 1024) {
    static size_t attr_bytes = 0;
    if (tile > 64 * 1024 && tile > attr_bytes) {
      hipError_t e = hipFuncSetAttribute((const void*)attn_pool_fwd_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
      if (e != hipSuccess) return (int)e;
      attr_bytes = 96 * 1024;
    }
    hipLaunchKernelGGL(attn_pool_fwd_lds_kernel, dim3((unsigned)G), dim3(256), tile, (hipStream_t)stream, X, w, b, P, C, vec, attn);
    MMEGO_LAUNCH_CHECK();
    return MMEGO_OK;
  }
  hipLaunchKernelGGL(attn_pool_fwd_kernel, dim3((unsigned)G), dim3(256), (size_t)P * sizeof(float), (hipStream_t)stream,
                     X, w, b, P, C, vec, attn);
  MMEGO_LAUNCH_CHECK();
  return MMEGO_OK;
}

extern "C" int mmego_attn_pool_backward(void* stream, const float* X, const float* w, const float* attn,
                                        const float* dvec, long G, int P, int C, float* dX, float* pdw, float* pdb) {
  MMEGO_REQUIRE(X && w && attn && dvec && dX && pdw && pdb && G > 0 && P > 0 && C > 0 && P <= 8192);
  {
    const int TCc = C < 256 ? C : 256, PG = 256 / TCc;
    const size_t tile = ((size_t)P * C + 2 * P + (size_t)PG * C) * sizeof(float);
    if ((C % 4) == 0 && (256 % TCc) == 0 && tile <= 64 * 1024 && (((uintptr_t)X) & 15) == 0) {
      hipLaunchKernelGGL(attn_pool_bwd_lds_kernel, dim3((unsigned)G), dim3(256), tile, (hipStream_t)stream, X, w, attn, dvec, P, C, dX,
                         pdw, pdb);
      MMEGO_LAUNCH_CHECK();
      return MMEGO_OK;
    }
  }
  hipLaunchKernelGGL(attn_pool_bwd_kernel, dim3((unsigned)G), dim3(256), (size_t)P * sizeof(float), (hipStream_t)stream,
                     X, w, attn, dvec, P, C, dX, pdw, pdb);
  MMEGO_LAUNCH_CHECK();
  return MMEGO_OK;
}

extern "C" int mmego_group_sum(void* stream, const float* X, long G, int P, int C, float scale, float* Y, long ldy) {
  MMEGO_REQUIRE(X && Y && G > 0 && P > 0 && C > 0);
  hipLaunchKernelGGL(group_sum_kernel, dim3(cdiv(G * C, 256)), dim3(256), 0, (hipStream_t)stream, X, P, C, scale, Y, ldy, G);
  MMEGO_LAUNCH_CHECK();
  return MMEGO_OK;
}

extern "C" int mmego_group_sum2(void* stream, long G, const float* X1, int P1, int C1, float scale1, float* Y1, long ldy1, const float* X2,
                                int P2, int C2, float scale2, float* Y2, long ldy2) {
  MMEGO_REQUIRE(X1 && Y1 && X2 && Y2 && G > 0 && P1 > 0 && C1 > 0 && P2 > 0 && C2 > 0);
  GroupSeg a = {X1, Y1, ldy1, P1, C1, scale1}, b = {X2, Y2, ldy2, P2, C2, scale2};
  const int nb1 = cdiv(G * C1, 256), nb2 = cdiv(G * C2, 256);
  hipLaunchKernelGGL(group_sum2_kernel, dim3(nb1 + nb2), dim3(256), 0, (hipStream_t)stream, a, b, G, nb1);
  MMEGO_LAUNCH_CHECK();
  return MMEGO_OK;
}

extern "C" int mmego_group_bcast2(void* stream, long G, const float* dY1, long lddy1, int P1, int C1, float scale1, float* dX1,
                                  const float* dY2, long lddy2, int P2, int C2, float scale2, float* dX2) {
  MMEGO_REQUIRE(dY1 && dX1 && dY2 && dX2 && G > 0 && P1 > 0 && C1 > 0 && P2 > 0 && C2 > 0);
  GroupSeg a = {dY1, dX1, lddy1, P1, C1, scale1}, b = {dY2, dX2, lddy2, P2, C2, scale2};
  const int nb1 = ew_blocks(G * P1 * C1), nb2 = ew_blocks(G * P2 * C2);
  hipLaunchKernelGGL(group_bcast2_kernel, dim3(nb1 + nb2), dim3(256), 0, (hipStream_t)stream, a, b, G, nb1);
  MMEGO_LAUNCH_CHECK();
  return MMEGO_OK;
}

extern "C" int mmego_group_bcast(void* stream, const float* dY, long lddy, long G, int P, int C, float scale, float* dX,
                                 int accumulate) {
  MMEGO_REQUIRE(dY && dX && G > 0 && P > 0 && C > 0);
  hipLaunchKernelGGL(group_bcast_kernel, dim3(ew_blocks(G * P * C)), dim3(256), 0, (hipStream_t)stream, dY, lddy, P, C,
                     scale, dX, G, accumulate);
  MMEGO_LAUNCH_CHECK();
  return MMEGO_OK;
}

extern "C" int mmego_cross_attn_forward(void* stream, const float* Q, const float* K, const float* V, long F, float scale,
                                        float* O, long ldo, float* P, long ldkv) {
  MMEGO_REQUIRE(Q && K && V && O && P && F > 0 && ldkv >= DH);
  const bool aligned = (((uintptr_t)Q | (uintptr_t)K | (uintptr_t)V | (uintptr_t)O) & 15) == 0 && (ldkv & 3) == 0 && (ldo & 3) == 0;
  if (aligned) hipLaunchKernelGGL((cross_attn_fwd_mfma_kernel<false, false>), dim3((unsigned)F), dim3(256), 0, (hipStream_t)stream, Q, K, V, scale, O, ldo, P, ldkv, (const float*)nullptr, (const float*)nullptr, 0L, F);
  else hipLaunchKernelGGL(cross_attn_fwd_kernel, dim3((unsigned)F), dim3(256), 0, (hipStream_t)stream, Q, K, V, scale, O, ldo, P, ldkv);
  MMEGO_LAUNCH_CHECK();
  return MMEGO_OK;
}

// The eval-mode form: osum[f][0..63] (row stride ldos) = sum over frame f's 64 queries of softmax(Q K^T scale) V; neither the per-query
// outputs nor the probabilities are stored (only backward reads them).  Needs 16-byte aligned Q, K, V and ldkv % 4 == 0.
extern "C" int mmego_cross_attn_forward_pooled(void* stream, const float* Q, const float* K, const float* V, long F, float scale,
                                               float* osum, long ldos, long ldkv) {
  MMEGO_REQUIRE(Q && K && V && osum && F > 0 && ldkv >= DH && ldos >= DH);
  MMEGO_REQUIRE((((uintptr_t)Q | (uintptr_t)K | (uintptr_t)V) & 15) == 0 && (ldkv & 3) == 0);
  hipLaunchKernelGGL((cross_attn_fwd_mfma_kernel<true, false>), dim3((unsigned)F), dim3(256), 0, (hipStream_t)stream, Q, K, V, scale, osum, ldos,
                     (float*)nullptr, ldkv, (const float*)nullptr, (const float*)nullptr, 0L, F);
  MMEGO_LAUNCH_CHECK();
  return MMEGO_OK;
}

// ... with the query projection inside: X [F * 64][ldx >= 64] point features, Wq [64][64], bq [64]; Q = X Wq^T + bq is never stored.
extern "C" int mmego_cross_attn_forward_pooled_q(void* stream, const float* X, long ldx, const float* Wq, const float* bq, const float* K,
                                                 const float* V, long F, float scale, float* osum, long ldos, long ldkv) {
  MMEGO_REQUIRE(X && Wq && bq && K && V && osum && F > 0 && ldkv >= DH && ldos >= DH && ldx >= DH);
  MMEGO_REQUIRE((((uintptr_t)X | (uintptr_t)Wq | (uintptr_t)bq | (uintptr_t)K | (uintptr_t)V) & 15) == 0 && (ldkv & 3) == 0 && (ldx & 3) == 0);
  const unsigned grid = (unsigned)(F < 2048 ? F : 2048);        // 35 KB of LDS: four workgroups per CU, two rounds of them
  hipLaunchKernelGGL((cross_attn_fwd_mfma_kernel<true, true>), dim3(grid), dim3(256), 0, (hipStream_t)stream, X, K, V, scale, osum, ldos,
                     (float*)nullptr, ldkv, Wq, bq, ldx, F);
  MMEGO_LAUNCH_CHECK();
  return MMEGO_OK;
}

extern "C" int mmego_cross_attn_backward(void* stream, const float* Q, const float* K, const float* V, const float* P,
                                         const float* dO, long lddo, long F, float scale, float* dQ, float* dK,
                                         float* dV, long ldkv) {
  MMEGO_REQUIRE(Q && K && V && P && dO && dQ && dK && dV && F > 0 && ldkv >= DH);
  hipLaunchKernelGGL(cross_attn_bwd_kernel, dim3((unsigned)F), dim3(256), 0, (hipStream_t)stream, Q, K, V, P, dO, lddo,
                     scale, dQ, dK, dV, ldkv);
  MMEGO_LAUNCH_CHECK();
  return MMEGO_OK;
}

extern "C" int mmego_graph_dA_nblk(long G) { return cdiv(G, GDA_FPB); }

extern "C" int mmego_graph_dA(void* stream, const float* Z, const float* dY, long G, int V, int K, int C, float* partial_ws,
                              const float* A, const float* imp, float* dZ, long ldz, long lddz) {
  MMEGO_REQUIRE(Z && dY && partial_ws && G > 0 && V > 0 && K > 0 && C > 0 && K * V * V <= 512);
  MMEGO_REQUIRE(!dZ || (A && imp));
  MMEGO_REQUIRE(ldz >= (long)K * C && (!dZ || lddz >= (long)K * C));
  size_t lds = (size_t)(V * (K * C + 1) + V * (C + 1) + K * V * V) * sizeof(float);
  MMEGO_REQUIRE(lds <= 64 * 1024);
  hipLaunchKernelGGL(graph_dA_partial_kernel, dim3(cdiv(G, GDA_FPB)), dim3(512), lds, (hipStream_t)stream, Z, dY, G, V, K, C,
                     partial_ws, A, imp, dZ, ldz, lddz);
  MMEGO_LAUNCH_CHECK();
  return MMEGO_OK;
}

extern "C" int mmego_transpose_batched(void* stream, const float* in, float* out, long Bn, int R, int C) {
  MMEGO_REQUIRE(in && out && Bn > 0 && R > 0 && C > 0);
  const long ntile = Bn * ((R + 63) / 64) * ((C + 63) / 64);
  hipLaunchKernelGGL(transpose_batched_kernel, dim3((unsigned)(ntile < 8192 ? ntile : 8192)), dim3(256), 0, (hipStream_t)stream, in, out,
                     Bn, R, C);
  MMEGO_LAUNCH_CHECK();
  return MMEGO_OK;
}

extern "C" int mmego_mul(void* stream, const float* a, const float* b, float* out, long n) {
  MMEGO_REQUIRE(a && b && out && n > 0);
  hipLaunchKernelGGL(mul_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, a, b, out, n);
  MMEGO_LAUNCH_CHECK();
  return MMEGO_OK;
}

extern "C" int mmego_add(void* stream, const float* a, const float* b, float* out, long n) {
  MMEGO_REQUIRE(a && b && out && n > 0);
  hipLaunchKernelGGL(add_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, a, b, out, n);
  MMEGO_LAUNCH_CHECK();
  return MMEGO_OK;
}

extern "C" int mmego_inc_i64(void* stream, long long* x, long n, unsigned long long* seed_ctr) {
  MMEGO_REQUIRE(n >= 0 && (n == 0 || x) && (n > 0 || seed_ctr));
  hipLaunchKernelGGL(inc_i64_kernel, dim3(n > 0 ? cdiv(n, 64) : 1), dim3(64), 0, (hipStream_t)stream, x, n, seed_ctr);
  MMEGO_LAUNCH_CHECK();
  return MMEGO_OK;
}
