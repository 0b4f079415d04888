// Geometry of the path: head-frame transforms, 6-D -> rotation, forward kinematics, L1 loss, and the
// top-k point selection of Lower_Net.  One thread per frame (or per point); these are latency-sized
// kernels (a few hundred frames), their job is to replace ~60 tiny aten launches per step.
//
// References: Util/Universal_Util/Utils.py:274-292 (Transform2H/2R), Net/Upper_Net.py:122-144,343-364,
// Net/Lower_Net.py:12-37,125-136,216-227, Net/IMU_Net.py:7-47, Processor/Train/Train_Upper.py:53,179.
#include <stdlib.h>

#include "common.h"

// r = (a0*b0 + a1*b1) + a2*b2 with every product and sum rounded separately (no fma contraction): this
// is the operation order of the CPU bmm the reference runs for its 3x3 . 3x1 products, so transformed
// coordinates -- which become sort keys downstream -- are bit-identical to the CPU path.
__device__ __forceinline__ float dot3_nofma(float a0, float a1, float a2, float b0, float b1, float b2) {
  return __fadd_rn(__fadd_rn(__fmul_rn(a0, b0), __fmul_rn(a1, b1)), __fmul_rn(a2, b2));
}

// pts[f, p, 0:3] <- R[f] . (pts[f, p, 0:3] - t[f])   IN PLACE (reference quirk Q1)
// src (optional): read the points from src instead (out of place: a copy + transform in one launch).  src_ld = 3: src holds the
// xyz only; src_ld >= C: whole rows, the channels behind xyz are copied as well (a training step's fresh minibatch).
// keep (optional, [npts][C]) and feats (optional, the first nfeat channels, row stride ldf) receive the transformed rows too --
// the copies Upper_Net's forward makes right behind the transform (the tensor stashed for backward, the xyz + intensity columns of
// the concatenated feature buffer).
__device__ __forceinline__ void t2h_point(long i, float* pts, int P, int C, const float* __restrict__ R, const float* __restrict__ t,
                                          const float* src, long src_ld, float* __restrict__ keep, float* __restrict__ feats, long ldf,
                                          int nfeat) {
  long f = i / P;
  const float* Rf = R + f * 9;
  const float* tf = t + f * 3;
  float* x = pts + i * C;
  const float* xs = src ? src + i * src_ld : x;
  float d0 = __fsub_rn(xs[0], tf[0]), d1 = __fsub_rn(xs[1], tf[1]), d2 = __fsub_rn(xs[2], tf[2]);
  float v[8];
  v[0] = dot3_nofma(Rf[0], Rf[1], Rf[2], d0, d1, d2);
  v[1] = dot3_nofma(Rf[3], Rf[4], Rf[5], d0, d1, d2);
  v[2] = dot3_nofma(Rf[6], Rf[7], Rf[8], d0, d1, d2);
  const bool rest = (src && src_ld >= C) || keep || (feats && nfeat > 3);        // (uniform)
  if (rest) {
    const float* xo = (src && src_ld >= C) ? xs : x;
#pragma unroll
    for (int c = 3; c < 8; ++c) v[c] = c < C ? xo[c] : 0.f;
  }
  x[0] = v[0]; x[1] = v[1]; x[2] = v[2];
  if (src && src_ld >= C) {
#pragma unroll
    for (int c = 3; c < 8; ++c) if (c < C) x[c] = v[c];
  }
  if (keep) {
    float* k = keep + i * C;
#pragma unroll
    for (int c = 0; c < 8; ++c) if (c < C) k[c] = v[c];
  }
  if (feats) {
    float* q = feats + i * ldf;
#pragma unroll
    for (int c = 0; c < 8; ++c) if (c < nfeat) q[c] = v[c];
  }
}

__global__ __launch_bounds__(256) void transform2h_kernel(float* pts, int P, int C, const float* __restrict__ R,
                                                          const float* __restrict__ t, long npts, const float* src, long src_ld,
                                                          float* __restrict__ keep, float* __restrict__ feats, long ldf, int nfeat) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= npts) return;
  t2h_point(i, pts, P, C, R, t, src, src_ld, keep, feats, ldf, nfeat);
}

// Two point sets of the same frames in one launch: pts [F][P][C] in place, and pts2 [F][P2][3] <- transform of src2 [F][P2][3]
// (Lower_Net.py:191-192,205: the radar points and the predicted upper-body joints go to the head frame side by side)
__global__ __launch_bounds__(256) void transform2h_pair_kernel(float* pts, int P, int C, const float* __restrict__ R,
                                                               const float* __restrict__ t, long npts, float* pts2, int P2,
                                                               const float* src2, long npts2) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < npts) t2h_point(i, pts, P, C, R, t, nullptr, 0, nullptr, nullptr, 0, 0);
  else if (i - npts < npts2) t2h_point(i - npts, pts2, P2, 3, R, t, src2, 3, nullptr, nullptr, 0, 0);
}

// out[f, p, :] = R[f]^T . in[f, p, :] + t[f]     (transpose=1, Transform2R)
// out[f, p, :] = R[f] . in[f, p, :]              (transpose=0, add_t=0: its backward)
__global__ __launch_bounds__(256) void rotate_points_kernel(const float* __restrict__ in, float* __restrict__ out, int P,
                                                            const float* __restrict__ R, const float* __restrict__ t,
                                                            long npts, int transpose, int add_t) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= npts) return;
  long f = i / P;
  const float* Rf = R + f * 9;
  float v0 = in[i * 3], v1 = in[i * 3 + 1], v2 = in[i * 3 + 2];
  float o0, o1, o2;
  if (transpose) {
    o0 = dot3_nofma(Rf[0], Rf[3], Rf[6], v0, v1, v2);
    o1 = dot3_nofma(Rf[1], Rf[4], Rf[7], v0, v1, v2);
    o2 = dot3_nofma(Rf[2], Rf[5], Rf[8], v0, v1, v2);
  } else {
    o0 = dot3_nofma(Rf[0], Rf[1], Rf[2], v0, v1, v2);
    o1 = dot3_nofma(Rf[3], Rf[4], Rf[5], v0, v1, v2);
    o2 = dot3_nofma(Rf[6], Rf[7], Rf[8], v0, v1, v2);
  }
  if (add_t) { o0 += t[f * 3]; o1 += t[f * 3 + 1]; o2 += t[f * 3 + 2]; }
  out[i * 3] = o0; out[i * 3 + 1] = o1; out[i * 3 + 2] = o2;
}

// ---- 6-D -> rotation (Gram-Schmidt), two eps rules -------------------------------------------------
// mode 0: F.normalize, v / max(|v|, 1e-12)  (Upper/Lower heads);  mode 1: v / max(|v|, 1e-8) (IMU_Net)
struct Rot6 {
  float x[3], y[3], z[3];
  float na, nw;  // clamped norms of a and of w = x cross b
};

__device__ __forceinline__ void cross3(const float* u, const float* v, float* o) {
  o[0] = u[1] * v[2] - u[2] * v[1];
  o[1] = u[2] * v[0] - u[0] * v[2];
  o[2] = u[0] * v[1] - u[1] * v[0];
}

__device__ __forceinline__ Rot6 rot6d_fwd(const float* six, float eps) {
  Rot6 r;
  const float* a = six;
  const float* b = six + 3;
  r.na = fmaxf(sqrtf(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]), eps);
  for (int i = 0; i < 3; ++i) r.x[i] = a[i] / r.na;
  float w[3];
  cross3(r.x, b, w);
  r.nw = fmaxf(sqrtf(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]), eps);
  for (int i = 0; i < 3; ++i) r.z[i] = w[i] / r.nw;
  cross3(r.z, r.x, r.y);
  return r;
}

// gR: gradient wrt the 3x3 matrix whose COLUMNS are x,y,z (row-major 9 floats) -> gradient wrt the 6 inputs
// MMEGO_HFK_VARIANT (probe builds of scripts/coexec_variants.py only; 0 = the product code): bit 0 = the two eps tests as selects of
// the projection factor instead of branches; bit 1 = this function's divisions as multiplications by v_rcp_f32.
#ifndef MMEGO_HFK_VARIANT
#define MMEGO_HFK_VARIANT 0
#endif
__device__ __forceinline__ float hfk_div(float a, float b) {
#if MMEGO_HFK_VARIANT & 2
  return a * __builtin_amdgcn_rcpf(b);
#else
  return a / b;
#endif
}
__device__ __forceinline__ void rot6d_bwd(const float* six, float eps, const float* gR, float* gsix) {
  Rot6 r = rot6d_fwd(six, eps);
  const float* b = six + 3;
  float gx[3] = {gR[0], gR[3], gR[6]}, gy[3] = {gR[1], gR[4], gR[7]}, gz[3] = {gR[2], gR[5], gR[8]};
  float tmp[3];
  // y = z cross x
  cross3(r.x, gy, tmp);
  for (int i = 0; i < 3; ++i) gz[i] += tmp[i];
  cross3(gy, r.z, tmp);
  for (int i = 0; i < 3; ++i) gx[i] += tmp[i];
  // z = w / max(|w|, eps)
  float w2 = 0.f;
  {
    float w[3];
    cross3(r.x, b, w);
    w2 = sqrtf(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
  }
  float gw[3];
#if MMEGO_HFK_VARIANT & 1
  {
    float dz = r.z[0] * gz[0] + r.z[1] * gz[1] + r.z[2] * gz[2];
    dz = w2 > eps ? dz : 0.f;
    for (int i = 0; i < 3; ++i) gw[i] = hfk_div(gz[i] - r.z[i] * dz, r.nw);
  }
#else
  if (w2 > eps) {
    float dz = r.z[0] * gz[0] + r.z[1] * gz[1] + r.z[2] * gz[2];
    for (int i = 0; i < 3; ++i) gw[i] = hfk_div(gz[i] - r.z[i] * dz, r.nw);
  } else {
    for (int i = 0; i < 3; ++i) gw[i] = hfk_div(gz[i], r.nw);
  }
#endif
  // w = x cross b
  cross3(b, gw, tmp);
  for (int i = 0; i < 3; ++i) gx[i] += tmp[i];
  cross3(gw, r.x, gsix + 3);
  // x = a / max(|a|, eps)
  float a2 = sqrtf(six[0] * six[0] + six[1] * six[1] + six[2] * six[2]);
#if MMEGO_HFK_VARIANT & 1
  {
    float dx = r.x[0] * gx[0] + r.x[1] * gx[1] + r.x[2] * gx[2];
    dx = a2 > eps ? dx : 0.f;
    for (int i = 0; i < 3; ++i) gsix[i] = hfk_div(gx[i] - r.x[i] * dx, r.na);
  }
#else
  if (a2 > eps) {
    float dx = r.x[0] * gx[0] + r.x[1] * gx[1] + r.x[2] * gx[2];
    for (int i = 0; i < 3; ++i) gsix[i] = hfk_div(gx[i] - r.x[i] * dx, r.na);
  } else {
    for (int i = 0; i < 3; ++i) gsix[i] = hfk_div(gx[i], r.na);
  }
#endif
}

// Forward-kinematics plans, compile-time: bone k writes slot child[k] = slot parent[k] + Rot[rot[k]] . body[row[k]].
// (With the plan as a kernel argument the per-frame joint / gradient arrays were indexed at run time and lived in scratch
//  memory: 25-29 us per launch on 4 workgroups.  Constant indices keep them in registers.)
//   WHICH = 0, upper head (Upper_Net.py:122-144, quirk Q4): Config.skeleton_upper_body walked in order
//     (20,3)(3,2)(2,1)(2,4)(2,8)(4,5)(5,6)(6,7)(8,9)(9,10)(10,11)(1,0)(0,12)(0,16), slots = index in
//     upper_joint_map [0..12,16,20] (head = slot 14, seeded from y[84:87]); rotation row = child slot, body row = k.
//   WHICH = 1, lower head (Lower_Net.py:12-37): (12,13)(13,14)(14,15)(16,17)(17,18)(18,19), slots = joint - 12, hips at slots
//     0 / 4 seeded from y[36:39] / y[39:42]; rotation row = index of the child in [13,14,15,17,18,19]; body rows 14..19.
template <int WHICH> struct FkC;
template <> struct FkC<0> {
  static constexpr int nbones = 14, nslots = 15, nrot = 14, ny = 87, nseed = 1;
  static constexpr int parent[14] = {14, 3, 2, 2, 2, 4, 5, 6, 8, 9, 10, 1, 0, 0};
  static constexpr int child[14] = {3, 2, 1, 4, 8, 5, 6, 7, 9, 10, 11, 0, 12, 13};
  static constexpr int rot[14] = {3, 2, 1, 4, 8, 5, 6, 7, 9, 10, 11, 0, 12, 13};
  static constexpr int row[14] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13};
  static constexpr int seed_slot[2] = {14, 0}, seed_off[2] = {84, 0};
};
template <> struct FkC<1> {
  static constexpr int nbones = 6, nslots = 8, nrot = 6, ny = 42, nseed = 2;
  static constexpr int parent[6] = {0, 1, 2, 4, 5, 6};
  static constexpr int child[6] = {1, 2, 3, 5, 6, 7};
  static constexpr int rot[6] = {0, 1, 2, 3, 4, 5};
  static constexpr int row[6] = {14, 15, 16, 17, 18, 19};
  static constexpr int seed_slot[2] = {0, 4}, seed_off[2] = {36, 39};
};

// y: [F, ny] head output (6*nrot rotation params, then positions);  body: [B, 20, 3];  frame n uses body n % B (Q2)
// outputs: q [F, nrot, 3, 3], joints_h [F, nslots, 3] (head frame)
template <int WHICH>
// Rw / tw / world (optional): the frame's head pose; world [F, nslots, 3] = Rw^T joint + tw, the rotate_points(transpose) that
// follows the kinematics in the nets (Upper_Net.py:362-364, Lower_Net.py:225-227), same arithmetic, no extra launch.
__global__ __launch_bounds__(64) void head_fk_fwd_kernel(const float* __restrict__ y, const float* __restrict__ body, int B, long F,
                                                         float* __restrict__ q, float* __restrict__ joints,
                                                         const float* __restrict__ Rw, const float* __restrict__ tw,
                                                         float* __restrict__ world, long long* counters, int ncount,
                                                         unsigned long long* seed_ctr) {
  using P = FkC<WHICH>;
  long f = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (f == 0) {
    // the net's once-per-training-forward tick (mmego_inc_i64's arithmetic) rides on this launch: BatchNorm
    // num_batches_tracked += 1, next dropout seed (every reader of the seed in this forward has run: same stream, earlier)
    for (int i = 0; i < ncount; ++i) counters[i] += 1;
    if (seed_ctr) seed_ctr[0] = seed_ctr[0] * 6364136223846793005ULL + 1442695040888963407ULL;
  }
  if (f >= F) return;
  // the frame's head outputs into registers first: all loads in flight at once (read where they were used -- six floats per rotation
  // -- they were 14 dependent round trips)
  float yv[P::ny];
#pragma unroll
  for (int i = 0; i < P::ny; ++i) yv[i] = y[f * P::ny + i];
  const float* yf = yv;
  const float* bf = body + (f % B) * 60;
  float* qf = q + f * P::nrot * 9;
  float l[P::nslots][3];
  float R[P::nrot][9];
#pragma unroll
  for (int s = 0; s < P::nseed; ++s)
#pragma unroll
    for (int i = 0; i < 3; ++i) l[P::seed_slot[s]][i] = yf[P::seed_off[s] + i];
#pragma unroll
  for (int k = 0; k < P::nrot; ++k) {
    Rot6 r = rot6d_fwd(yf + 6 * k, 1e-12f);
#pragma unroll
    for (int i = 0; i < 3; ++i) { R[k][i * 3 + 0] = r.x[i]; R[k][i * 3 + 1] = r.y[i]; R[k][i * 3 + 2] = r.z[i]; }
#pragma unroll
    for (int i = 0; i < 9; ++i) qf[k * 9 + i] = R[k][i];
  }
#pragma unroll
  for (int k = 0; k < P::nbones; ++k) {
    const float* bv = bf + P::row[k] * 3;
    const float b0 = bv[0], b1 = bv[1], b2 = bv[2];
#pragma unroll
    for (int i = 0; i < 3; ++i)
      l[P::child[k]][i] = l[P::parent[k]][i] + (R[P::rot[k]][i * 3] * b0 + R[P::rot[k]][i * 3 + 1] * b1 + R[P::rot[k]][i * 3 + 2] * b2);
  }
#pragma unroll
  for (int s = 0; s < P::nslots; ++s)
#pragma unroll
    for (int i = 0; i < 3; ++i) joints[(f * P::nslots + s) * 3 + i] = l[s][i];
  if (world) {
    const float* Rf = Rw + f * 9;
    const float t0 = tw[f * 3], t1 = tw[f * 3 + 1], t2 = tw[f * 3 + 2];
#pragma unroll
    for (int s = 0; s < P::nslots; ++s) {
      float* w = world + (f * P::nslots + s) * 3;
      w[0] = dot3_nofma(Rf[0], Rf[3], Rf[6], l[s][0], l[s][1], l[s][2]) + t0;
      w[1] = dot3_nofma(Rf[1], Rf[4], Rf[7], l[s][0], l[s][1], l[s][2]) + t1;
      w[2] = dot3_nofma(Rf[2], Rf[5], Rf[8], l[s][0], l[s][1], l[s][2]) + t2;
    }
  }
}

// dj: [F, nslots, 3] gradient wrt head-frame joints  ->  dy [F, ny]
// Rw (optional): dj is the gradient wrt the WORLD-frame joints; the head-frame gradient Rw dj is formed first.
template <int WHICH>
__global__ __launch_bounds__(64) void head_fk_bwd_kernel(const float* __restrict__ y, const float* __restrict__ body, int B, long F,
                                                         const float* __restrict__ dj, float* __restrict__ dy,
                                                         const float* __restrict__ Rw) {
  using P = FkC<WHICH>;
  long f = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= F) return;
  float yv[P::ny];                       // (preloaded: see the forward kernel)
#pragma unroll
  for (int i = 0; i < P::ny; ++i) yv[i] = y[f * P::ny + i];
  const float* yf = yv;
  const float* bf = body + (f % B) * 60;
  float g[P::nslots][3];
#pragma unroll
  for (int s = 0; s < P::nslots; ++s)
#pragma unroll
    for (int i = 0; i < 3; ++i) g[s][i] = dj[(f * P::nslots + s) * 3 + i];
  if (Rw) {
    const float* Rf = Rw + f * 9;
#pragma unroll
    for (int s = 0; s < P::nslots; ++s) {
      const float v0 = g[s][0], v1 = g[s][1], v2 = g[s][2];
      g[s][0] = dot3_nofma(Rf[0], Rf[1], Rf[2], v0, v1, v2);
      g[s][1] = dot3_nofma(Rf[3], Rf[4], Rf[5], v0, v1, v2);
      g[s][2] = dot3_nofma(Rf[6], Rf[7], Rf[8], v0, v1, v2);
    }
  }
  float gq[P::nrot][9];
#pragma unroll
  for (int k = 0; k < P::nrot; ++k)
#pragma unroll
    for (int i = 0; i < 9; ++i) gq[k][i] = 0.f;
#pragma unroll
  for (int k = P::nbones - 1; k >= 0; --k) {
    const float* bv = bf + P::row[k] * 3;
    const float b0 = bv[0], b1 = bv[1], b2 = bv[2];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      g[P::parent[k]][i] += g[P::child[k]][i];
      gq[P::rot[k]][i * 3 + 0] += g[P::child[k]][i] * b0;
      gq[P::rot[k]][i * 3 + 1] += g[P::child[k]][i] * b1;
      gq[P::rot[k]][i * 3 + 2] += g[P::child[k]][i] * b2;
    }
  }
  float* dyf = dy + f * P::ny;
#pragma unroll
  for (int k = 0; k < P::nrot; ++k) rot6d_bwd(yf + 6 * k, 1e-12f, gq[k], dyf + 6 * k);
#pragma unroll
  for (int i = 6 * P::nrot; i < P::ny; ++i) dyf[i] = 0.f;
#pragma unroll
  for (int s = 0; s < P::nseed; ++s)
#pragma unroll
    for (int i = 0; i < 3; ++i) dyf[P::seed_off[s] + i] = g[P::seed_slot[s]][i];
}

// Forward kinematics + head-to-world transform + L1(sum) loss against the selected target joints + the loss gradient (its sign)
// + the backward of the transform and of the kinematics, in ONE launch: one thread per frame in 64-thread workgroups (one wave per
// workgroup: the frame's ~400 live values fit the unified register file).  The loss is a fixed-order sum: every workgroup publishes its
// partial pair (agent-scope stores), takes a ticket, and the workgroup that draws the last one adds the partials IN INDEX ORDER -- the
// same value whichever workgroup is last -- and resets the ticket for the next launch (graph replays).  Same arithmetic, statement by statement, as
// head_fk_fwd_kernel -> l1_loss_kernel -> head_fk_bwd_kernel (bit-identical results; tests/test_hip_parity.py).  Replaces three
// dependent launches at the turning point of a training step (Train_Upper.py:165-182, Train_Lower.py:199-224).
template <int WHICH>
__global__ __launch_bounds__(64) void head_fk_loss_kernel(const float* __restrict__ y, const float* __restrict__ body, int B, long F,
                                                          float* __restrict__ q, float* __restrict__ joints,
                                                          const float* __restrict__ Rw, const float* __restrict__ tw,
                                                          float* __restrict__ world, long long* counters, int ncount,
                                                          unsigned long long* seed_ctr, const float* __restrict__ target,
                                                          const int* __restrict__ map, int ntgt, float scale,
                                                          float* __restrict__ loss, float* __restrict__ dy, double* part,
                                                          unsigned* ticket) {
  using P = FkC<WHICH>;
  const long f = (long)blockIdx.x * 64 + threadIdx.x;
  if (f == 0) {
    for (int i = 0; i < ncount; ++i) counters[i] += 1;
    if (seed_ctr) seed_ctr[0] = seed_ctr[0] * 6364136223846793005ULL + 1442695040888963407ULL;
  }
  double acc = 0.0, dist = 0.0;
  if (f < F) {
    float yv[P::ny];
#pragma unroll
    for (int i = 0; i < P::ny; ++i) yv[i] = y[f * P::ny + i];
    float tg[P::nslots][3];
#pragma unroll
    for (int s_ = 0; s_ < P::nslots; ++s_) {
      const float* tp = target + (f * ntgt + map[s_]) * 3;
#pragma unroll
      for (int i = 0; i < 3; ++i) tg[s_][i] = tp[i];
    }
    const float* yf = yv;
    const float* bf = body + (f % B) * 60;
    float bd[P::nbones][3];
#pragma unroll
    for (int k = 0; k < P::nbones; ++k)
#pragma unroll
      for (int i = 0; i < 3; ++i) bd[k][i] = bf[P::row[k] * 3 + i];
    const float* Rf = Rw + f * 9;
    float Rr[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) Rr[i] = Rf[i];
    const float t0 = tw[f * 3], t1 = tw[f * 3 + 1], t2 = tw[f * 3 + 2];
    float* qf = q + f * P::nrot * 9;
    float l[P::nslots][3];
    float g[P::nslots][3];
    {
      float R[P::nrot][9];
#pragma unroll
      for (int s_ = 0; s_ < P::nseed; ++s_)
#pragma unroll
        for (int i = 0; i < 3; ++i) l[P::seed_slot[s_]][i] = yf[P::seed_off[s_] + i];
#pragma unroll
      for (int k = 0; k < P::nrot; ++k) {
        Rot6 r = rot6d_fwd(yf + 6 * k, 1e-12f);
#pragma unroll
        for (int i = 0; i < 3; ++i) { R[k][i * 3 + 0] = r.x[i]; R[k][i * 3 + 1] = r.y[i]; R[k][i * 3 + 2] = r.z[i]; }
#pragma unroll
        for (int i = 0; i < 9; ++i) qf[k * 9 + i] = R[k][i];
      }
#pragma unroll
      for (int k = 0; k < P::nbones; ++k) {
        const float b0 = bd[k][0], b1 = bd[k][1], b2 = bd[k][2];
#pragma unroll
        for (int i = 0; i < 3; ++i)
          l[P::child[k]][i] = l[P::parent[k]][i] + (R[P::rot[k]][i * 3] * b0 + R[P::rot[k]][i * 3 + 1] * b1 + R[P::rot[k]][i * 3 + 2] * b2);
      }
    }
#pragma unroll
    for (int s_ = 0; s_ < P::nslots; ++s_) {
#pragma unroll
      for (int i = 0; i < 3; ++i) joints[(f * P::nslots + s_) * 3 + i] = l[s_][i];
      float w[3];
      w[0] = dot3_nofma(Rr[0], Rr[3], Rr[6], l[s_][0], l[s_][1], l[s_][2]) + t0;
      w[1] = dot3_nofma(Rr[1], Rr[4], Rr[7], l[s_][0], l[s_][1], l[s_][2]) + t1;
      w[2] = dot3_nofma(Rr[2], Rr[5], Rr[8], l[s_][0], l[s_][1], l[s_][2]) + t2;
      float* wp = world + (f * P::nslots + s_) * 3;
      wp[0] = w[0]; wp[1] = w[1]; wp[2] = w[2];
      const float dx = w[0] - tg[s_][0], dyy = w[1] - tg[s_][1], dz = w[2] - tg[s_][2];
      acc += ((double)fabsf(dx) + (double)fabsf(dyy)) + (double)fabsf(dz);
      dist += (double)sqrtf(dx * dx + dyy * dyy + dz * dz);
      const float v0 = dx > 0.f ? scale : (dx < 0.f ? -scale : 0.f);
      const float v1 = dyy > 0.f ? scale : (dyy < 0.f ? -scale : 0.f);
      const float v2 = dz > 0.f ? scale : (dz < 0.f ? -scale : 0.f);
      g[s_][0] = dot3_nofma(Rr[0], Rr[1], Rr[2], v0, v1, v2);       // world -> head frame (head_fk_bwd_kernel's first step)
      g[s_][1] = dot3_nofma(Rr[3], Rr[4], Rr[5], v0, v1, v2);
      g[s_][2] = dot3_nofma(Rr[6], Rr[7], Rr[8], v0, v1, v2);
    }
    float gq[P::nrot][9];
#pragma unroll
    for (int k = 0; k < P::nrot; ++k)
#pragma unroll
      for (int i = 0; i < 9; ++i) gq[k][i] = 0.f;
#pragma unroll
    for (int k = P::nbones - 1; k >= 0; --k) {
      const float b0 = bd[k][0], b1 = bd[k][1], b2 = bd[k][2];
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        g[P::parent[k]][i] += g[P::child[k]][i];
        gq[P::rot[k]][i * 3 + 0] += g[P::child[k]][i] * b0;
        gq[P::rot[k]][i * 3 + 1] += g[P::child[k]][i] * b1;
        gq[P::rot[k]][i * 3 + 2] += g[P::child[k]][i] * b2;
      }
    }
    float* dyf = dy + f * P::ny;
#pragma unroll
    for (int k = 0; k < P::nrot; ++k) rot6d_bwd(yf + 6 * k, 1e-12f, gq[k], dyf + 6 * k);
#pragma unroll
    for (int i = 6 * P::nrot; i < P::ny; ++i) dyf[i] = 0.f;
#pragma unroll
    for (int s_ = 0; s_ < P::nseed; ++s_)
#pragma unroll
      for (int i = 0; i < 3; ++i) dyf[P::seed_off[s_] + i] = g[P::seed_slot[s_]][i];
  }
  acc = wave_sum_d(acc);
  dist = wave_sum_d(dist);
  if (threadIdx.x == 0) {
    __hip_atomic_store(&part[2 * blockIdx.x], acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&part[2 * blockIdx.x + 1], dist, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // The pair above must be visible before the ticket is drawn, and the last workgroup must read the pairs after its ticket.
    // Hardware side: agent-scope atomic (write-through, sc1) stores, drained by vmcnt(0), then an agent-scope atomic add; the reader's
    // loads are agent-scope atomics too (never a stale L1 / L2 line) -- the "8-byte agent atomics on both sides" form of the hand-off
    // rules.  Compiler side: relaxed atomics on different addresses may legally be moved across a waitcnt BUILTIN, so the
    // drain is a volatile asm with a memory clobber (no memory access crosses it), and a second one opens the reading branch.  A
    // release / acquire pair in the memory model would say the same at the price of an L2 write-back + invalidate (~1.7 us each)
    // per launch on the critical path of both stage tails.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (t == gridDim.x - 1) {                              // last workgroup: the partials in index order (agent-scope loads: not a stale L2 line)
      asm volatile("" ::: "memory");
      double s0 = 0.0, s1 = 0.0;
      for (unsigned w = 0; w < gridDim.x; ++w) {
        s0 += __hip_atomic_load(&part[2 * w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s1 += __hip_atomic_load(&part[2 * w + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      loss[0] = (float)s0;
      loss[1] = (float)s1;
      __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// IMU head: out[f, 0:9] -> R [f,3,3] (eps rule 1e-8), t [f,3]
__global__ __launch_bounds__(128) void imu_head_kernel(const float* __restrict__ y, long F, float* __restrict__ R,
                                                       float* __restrict__ t) {
  long f = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= F) return;
  Rot6 r = rot6d_fwd(y + f * 9, 1e-8f);
  for (int i = 0; i < 3; ++i) {
    R[f * 9 + i * 3 + 0] = r.x[i]; R[f * 9 + i * 3 + 1] = r.y[i]; R[f * 9 + i * 3 + 2] = r.z[i];
    t[f * 3 + i] = y[f * 9 + 6 + i];
  }
}

// IMU_Net's last Linear (2H -> 9, Net/IMU_Net.py:84) and the head in one launch: y[r][0:9] = b + x[r][:] . W[9][K] as row-wise
// dot products, then 6-D -> R and t.  (As a product with 9 output columns it was 16 workgroups chaining 8 dependent k-steps:
// 13.9 us plus the head launch; here every row is one coalesced read by half a workgroup.)  One wave per 2 rows, lane l takes
// k = 4 l + 256 i; the 64 lane sums meet in a butterfly of fixed shape: deterministic.  K % 256 == 0.
__global__ __launch_bounds__(256) void imu_fc2_head_kernel(const float* __restrict__ X, long ldx, const float* __restrict__ W,
                                                           const float* __restrict__ b, long F, int K, float* __restrict__ y_out,
                                                           float* __restrict__ R, float* __restrict__ t) {
  const int lane = threadIdx.x & 63;
  const long r0 = 2 * ((long)blockIdx.x * 4 + (threadIdx.x >> 6));
  if (r0 >= F) return;
  const long r1 = r0 + 1 < F ? r0 + 1 : r0;
  const f32x4* x0 = reinterpret_cast<const f32x4*>(X + r0 * ldx) + lane;
  const f32x4* x1 = reinterpret_cast<const f32x4*>(X + r1 * ldx) + lane;
  const f32x4* w = reinterpret_cast<const f32x4*>(W) + lane;
  const int K4 = K >> 2;
  float acc[2][9];
#pragma unroll
  for (int n = 0; n < 9; ++n) acc[0][n] = acc[1][n] = 0.f;
  f32x4 a0 = x0[0], a1 = x1[0], wv[9];
#pragma unroll
  for (int n = 0; n < 9; ++n) wv[n] = w[n * K4];
  for (int i = 0; i < K4; i += 64) {
    const f32x4 c0 = a0, c1 = a1;
    f32x4 cw[9];
#pragma unroll
    for (int n = 0; n < 9; ++n) cw[n] = wv[n];
    const int in = i + 64 < K4 ? i + 64 : i;              // (next step's loads, unconditional on a clamped index)
    a0 = x0[in]; a1 = x1[in];
#pragma unroll
    for (int n = 0; n < 9; ++n) wv[n] = w[n * K4 + in];
#pragma unroll
    for (int n = 0; n < 9; ++n) {
      acc[0][n] += (c0.x * cw[n].x + c0.y * cw[n].y) + (c0.z * cw[n].z + c0.w * cw[n].w);
      acc[1][n] += (c1.x * cw[n].x + c1.y * cw[n].y) + (c1.z * cw[n].z + c1.w * cw[n].w);
    }
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1)
#pragma unroll
    for (int n = 0; n < 9; ++n) {
      acc[0][n] += __shfl_xor(acc[0][n], m, 64);
      acc[1][n] += __shfl_xor(acc[1][n], m, 64);
    }
  if (lane < 2 && r0 + lane < F) {
    const long f = r0 + lane;
    float y[9];
#pragma unroll
    for (int n = 0; n < 9; ++n) y[n] = (lane == 0 ? acc[0][n] : acc[1][n]) + b[n];
    if (y_out) {
#pragma unroll
      for (int n = 0; n < 9; ++n) y_out[f * 9 + n] = y[n];
    }
    const Rot6 r = rot6d_fwd(y, 1e-8f);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      R[f * 9 + i * 3 + 0] = r.x[i]; R[f * 9 + i * 3 + 1] = r.y[i]; R[f * 9 + i * 3 + 2] = r.z[i];
      t[f * 3 + i] = y[6 + i];
    }
  }
}

// loss[0] = sum |pred - target[:, map]| ; grad = scale * sign(pred - target) ; loss[1] = sum of the per-joint Euclidean
// distances (the "accuracy" the reference logs per minibatch, Train_Upper.py:183-185: mean = loss[1] / (F * nsel)).
//   pred [F, nsel, 3]; target [F, ntgt, 3]; map[nsel] selects target joints.  One thread per joint; single block with a
//   fixed summation structure: deterministic.
__global__ __launch_bounds__(1024) void l1_loss_kernel(const float* __restrict__ pred, const float* __restrict__ target,
                                                       const int* __restrict__ map, int nsel, int ntgt, long F,
                                                       float scale, float* __restrict__ loss, float* __restrict__ grad) {
  __shared__ double sh[2][16];
  __shared__ int smap[64];
  const long joints = F * nsel;
  if ((int)threadIdx.x < nsel && threadIdx.x < 64) smap[threadIdx.x] = map[threadIdx.x];
  __syncthreads();
  double acc = 0.0, dist = 0.0;
  // eight joints per thread and pass with all their loads in flight (a rolled loop chained map -> target -> pred round trips:
  // ~10 us for 7 K joints)
  for (long base = 0; base < joints; base += 8 * (long)blockDim.x) {
    float pv[8][3], tv[8][3];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      // (clamped joint index: the loads are unconditional -- an `if (i < joints)` around them made every load a branch plus a
      // full wait, 48 dependent round trips per pass; the uses below are predicated)
      const long i = min(base + (long)u * blockDim.x + threadIdx.x, joints - 1);
      const long f = i / nsel;
      const int s = (int)(i - f * nsel);
      const float* pp = pred + i * 3;
      const float* tp = target + (f * ntgt + (nsel <= 64 ? smap[s] : map[s])) * 3;
#pragma unroll
      for (int k = 0; k < 3; ++k) { pv[u][k] = pp[k]; tv[u][k] = tp[k]; }
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int k = 0; k < 3; ++k) { asm volatile("" : "+v"(pv[u][k])); asm volatile("" : "+v"(tv[u][k])); }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const long i = base + (long)u * blockDim.x + threadIdx.x;
      if (i < joints) {
        const float dx = pv[u][0] - tv[u][0], dy = pv[u][1] - tv[u][1], dz = pv[u][2] - tv[u][2];
        acc += ((double)fabsf(dx) + (double)fabsf(dy)) + (double)fabsf(dz);
        dist += (double)sqrtf(dx * dx + dy * dy + dz * dz);
        if (grad) {
          grad[i * 3 + 0] = dx > 0.f ? scale : (dx < 0.f ? -scale : 0.f);
          grad[i * 3 + 1] = dy > 0.f ? scale : (dy < 0.f ? -scale : 0.f);
          grad[i * 3 + 2] = dz > 0.f ? scale : (dz < 0.f ? -scale : 0.f);
        }
      }
    }
  }
  acc = wave_sum_d(acc);
  dist = wave_sum_d(dist);
  if ((threadIdx.x & 63) == 0) { sh[0][threadIdx.x >> 6] = acc; sh[1][threadIdx.x >> 6] = dist; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double s0 = 0.0, s1 = 0.0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) { s0 += sh[0][w]; s1 += sh[1][w]; }
    loss[0] = (float)s0;
    loss[1] = (float)s1;
  }
}

// Keep the `keep` rows with the largest key (column 0), descending, ties lowest-index-first (stable);
// out [F, keep, C], idx int64 [F, keep].  Reference Lower_Net.py:216-227 (torch.sort, tie order unspecified).
// out2 (optional): the first n2 columns of every kept row go there too (row stride ld2) -- Lower_Net's cat(xyz, features) buffer gets
// its xyz part from this launch instead of a copy launch behind it.
__global__ __launch_bounds__(256) void topk_rows_kernel(const float* __restrict__ pts, int N, int C, int keep,
                                                        float* __restrict__ out, long long* __restrict__ idx,
                                                        float* __restrict__ out2, long ld2, int n2) {
  extern __shared__ float keys[];
  const long f = blockIdx.x;
  const float* pf = pts + f * N * C;
  for (int i = threadIdx.x; i < N; i += blockDim.x) keys[i] = pf[i * C];
  __syncthreads();
  for (int i = threadIdx.x; i < N; i += blockDim.x) {
    const float k = keys[i];
    int rank = 0;
    for (int j = 0; j < N; ++j) {
      float kj = keys[j];
      rank += (kj > k) || (kj == k && j < i);
    }
    if (rank < keep) {
      idx[f * keep + rank] = i;
      float* o = out + (f * keep + rank) * C;
      for (int c = 0; c < C; ++c) o[c] = pf[i * C + c];
      if (out2)
        for (int c = 0; c < n2; ++c) out2[(f * keep + rank) * ld2 + c] = pf[i * C + c];
    }
  }
}

// The same selection by an in-register bitonic sort (r06): ONE WAVE per frame, NR = ceil(N / 64) elements per lane (element e = 64 r +
// lane, padded with key -inf / index e), sorted descending by (key, then index ascending) -- a total order, so the network's result is
// THE order the rank kernel above computes, bit for bit (float compares: -0 == +0 ties go by index there and here).  Compare-exchange
// partners 64 r apart are registers of the same lane; partners < 64 apart are lanes (DPP / permlane moves, no LDS).  36 steps x 4
// elements for N = 256 against 256 LDS reads and compares per element.  Small launches (F <= 1024) keep the rank kernel: four waves
// per frame finish a frame sooner than one, and that is what counts when the frames do not fill the chip.
// the value of lane (l ^ J), J = 1 .. 32, without the LDS crossbar (ds_bpermute: the sort was bound by it -- 264 of them per frame):
// DPP quad permutes (1, 2), two bank-masked row shifts (4), a row rotate (8), v_permlane16_swap / v_permlane32_swap (16, 32)
typedef unsigned tk_u32x2 __attribute__((ext_vector_type(2)));
template <int J>
__device__ __forceinline__ int tk_xor_lane(int x, int lane) {
  if constexpr (J == 1) return __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xF, 0xF, true);           // quad_perm [1, 0, 3, 2]
  else if constexpr (J == 2) return __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xF, 0xF, true);      // quad_perm [2, 3, 0, 1]
  else if constexpr (J == 4) {
    const int r = __builtin_amdgcn_update_dpp(0, x, 0x104, 0xF, 0x5, false);                      // row_shl:4 into banks 0, 2
    return __builtin_amdgcn_update_dpp(r, x, 0x114, 0xF, 0xA, false);                             // row_shr:4 into banks 1, 3
  } else if constexpr (J == 8) return __builtin_amdgcn_update_dpp(0, x, 0x128, 0xF, 0xF, true);   // row_ror:8
  else if constexpr (J == 16) {
    const tk_u32x2 r = __builtin_amdgcn_permlane16_swap((unsigned)x, (unsigned)x, false, false);
    return (lane & 16) ? (int)r[0] : (int)r[1];
  } else {
    const tk_u32x2 r = __builtin_amdgcn_permlane32_swap((unsigned)x, (unsigned)x, false, false);
    return (lane & 32) ? (int)r[0] : (int)r[1];
  }
}

// one compare-exchange step between lanes J apart, all NR elements of the lane (KK: the bitonic stage, for the direction)
template <int NR, int KK, int J>
__device__ __forceinline__ void tk_lane_step(float (&k)[NR], int (&id)[NR], int lane) {
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    const float ko = __builtin_bit_cast(float, tk_xor_lane<J>(__builtin_bit_cast(int, k[r]), lane));
    const int io = tk_xor_lane<J>(id[r], lane);
    const int p = 64 * r + lane;
    const bool desc = (p & KK) == 0;
    const bool lower = (lane & J) == 0;                 // this lane holds the lower position of the pair
    const bool mine_first = (k[r] > ko) | ((k[r] == ko) & (id[r] < io));           // (| and &: no branches around the compares)
    // the lower position keeps the element that comes first in a descending pair, second in an ascending one
    const bool keep_mine = mine_first == (lower == desc);
    k[r] = keep_mine ? k[r] : ko;
    id[r] = keep_mine ? id[r] : io;
  }
}

template <int NR, int KK>
__device__ __forceinline__ void tk_lane_steps(float (&k)[NR], int (&id)[NR], int lane) {
  if constexpr (KK >= 64) tk_lane_step<NR, KK, 32>(k, id, lane);
  if constexpr (KK >= 32) tk_lane_step<NR, KK, 16>(k, id, lane);
  if constexpr (KK >= 16) tk_lane_step<NR, KK, 8>(k, id, lane);
  if constexpr (KK >= 8) tk_lane_step<NR, KK, 4>(k, id, lane);
  if constexpr (KK >= 4) tk_lane_step<NR, KK, 2>(k, id, lane);
  tk_lane_step<NR, KK, 1>(k, id, lane);
}

// the stage's steps between registers (partners 64 jr apart), largest distance first
template <int NR, int KK>
__device__ __forceinline__ void tk_reg_steps(float (&k)[NR], int (&id)[NR]) {
#pragma unroll
  for (int j = KK >> 1; j >= 64; j >>= 1) {
    const int jr = j >> 6;
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      if ((r & jr) == 0) {                              // r < partner r ^ jr: position p = 64 r + lane, direction from p & KK
        const int rp = r ^ jr;
        const bool desc = ((64 * r) & KK) == 0;         // (KK >= 128 here: a bit of r)
        const bool first = (k[r] > k[rp]) | ((k[r] == k[rp]) & (id[r] < id[rp]));   // element r comes before element rp
        const bool swap = first != desc;
        const float k0 = swap ? k[rp] : k[r], k1 = swap ? k[r] : k[rp];
        const int i0 = swap ? id[rp] : id[r], i1 = swap ? id[r] : id[rp];
        k[r] = k0; k[rp] = k1; id[r] = i0; id[rp] = i1;
      }
    }
  }
}

template <int NR, int KK>
__device__ __forceinline__ void tk_stage(float (&k)[NR], int (&id)[NR], int lane) {
  if constexpr (KK <= 64 * NR) {
    tk_reg_steps<NR, KK>(k, id);
    tk_lane_steps<NR, KK>(k, id, lane);
  }
}

template <int NR>
__global__ __launch_bounds__(256) void topk_rows_sort_kernel(const float* __restrict__ pts, long F, int N, int C, int keep,
                                                             float* __restrict__ out, long long* __restrict__ idx,
                                                             float* __restrict__ out2, long ld2, int n2) {
  const int lane = threadIdx.x & 63;
  const long f = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (f >= F) return;                                     // (whole waves; no barrier in this kernel)
  const float* pf = pts + f * N * C;
  float k[NR];
  int id[NR];
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    const int e = 64 * r + lane;
    id[r] = e;
    k[r] = e < N ? pf[(long)e * C] : -INFINITY;
  }
  tk_stage<NR, 2>(k, id, lane);
  tk_stage<NR, 4>(k, id, lane);
  tk_stage<NR, 8>(k, id, lane);
  tk_stage<NR, 16>(k, id, lane);
  tk_stage<NR, 32>(k, id, lane);
  tk_stage<NR, 64>(k, id, lane);
  tk_stage<NR, 128>(k, id, lane);
  tk_stage<NR, 256>(k, id, lane);
  tk_stage<NR, 512>(k, id, lane);
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    const int rank = 64 * r + lane;
    if (rank < keep) {
      const int i = id[r];
      idx[f * keep + rank] = i;
      float* o = out + (f * keep + rank) * C;
      for (int c = 0; c < C; ++c) o[c] = pf[(long)i * C + c];
      if (out2)
        for (int c = 0; c < n2; ++c) out2[(f * keep + rank) * ld2 + c] = pf[(long)i * C + c];
    }
  }
}

static void topk_launch(hipStream_t st, const float* pts, long F, int N, int C, int keep, float* out, long long* idx, float* out2, long ld2,
                        int n2) {
  const unsigned grid4 = (unsigned)((F + 3) / 4);
  if (F <= 1024 && N <= 4096)
    hipLaunchKernelGGL(topk_rows_kernel, dim3((unsigned)F), dim3(N < 256 ? ((N + 63) / 64) * 64 : 256), (size_t)N * sizeof(float), st, pts, N, C,
                       keep, out, idx, out2, ld2, n2);
  else if (N <= 64) hipLaunchKernelGGL(topk_rows_sort_kernel<1>, dim3(grid4), dim3(256), 0, st, pts, F, N, C, keep, out, idx, out2, ld2, n2);
  else if (N <= 128) hipLaunchKernelGGL(topk_rows_sort_kernel<2>, dim3(grid4), dim3(256), 0, st, pts, F, N, C, keep, out, idx, out2, ld2, n2);
  else if (N <= 256) hipLaunchKernelGGL(topk_rows_sort_kernel<4>, dim3(grid4), dim3(256), 0, st, pts, F, N, C, keep, out, idx, out2, ld2, n2);
  else if (N <= 512) hipLaunchKernelGGL(topk_rows_sort_kernel<8>, dim3(grid4), dim3(256), 0, st, pts, F, N, C, keep, out, idx, out2, ld2, n2);
  else
    hipLaunchKernelGGL(topk_rows_kernel, dim3((unsigned)F), dim3(256), (size_t)N * sizeof(float), st, pts, N, C, keep, out, idx, out2, ld2, n2);
}

// ------------------------------------------------------------------------------------------------
extern "C" int mmego_transform2h(void* stream, float* pts, long F, int P, int C, const float* R, const float* t, const float* src,
                                 long src_ld, float* keep, float* feats, long ldf, int nfeat) {
  MMEGO_REQUIRE(pts && R && t && F > 0 && P > 0 && C >= 3);
  MMEGO_REQUIRE(!src || src_ld == 3 || src_ld >= C);
  MMEGO_REQUIRE((!keep && !feats && !(src && src_ld >= C)) || C <= 8);
  MMEGO_REQUIRE(!feats || (nfeat >= 1 && nfeat <= C && nfeat <= 8 && ldf >= nfeat));
  long n = F * P;
  hipLaunchKernelGGL(transform2h_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, pts, P, C, R, t, n, src, src_ld, keep,
                     feats, ldf, nfeat);
  MMEGO_LAUNCH_CHECK();
  return MMEGO_OK;
}

extern "C" int mmego_transform2h_pair(void* stream, float* pts, long F, int P, int C, const float* R, const float* t, float* pts2, int P2,
                                      const float* src2) {
  MMEGO_REQUIRE(pts && R && t && pts2 && src2 && F > 0 && P > 0 && C >= 3 && P2 > 0);
  const long n = F * P, n2 = F * P2;
  hipLaunchKernelGGL(transform2h_pair_kernel, dim3(cdiv(n + n2, 256)), dim3(256), 0, (hipStream_t)stream, pts, P, C, R, t, n, pts2, P2, src2, n2);
  MMEGO_LAUNCH_CHECK();
  return MMEGO_OK;
}

extern "C" int mmego_rotate_points(void* stream, const float* in, float* out, long F, int P, const float* R,
                                   const float* t, int transpose, int add_t) {
  MMEGO_REQUIRE(in && out && R && F > 0 && P > 0 && (!add_t || t));
  long n = F * P;
  hipLaunchKernelGGL(rotate_points_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, in, out, P, R, t, n,
                     transpose, add_t);
  MMEGO_LAUNCH_CHECK();
  return MMEGO_OK;
}

// which: 0 = upper head (ny = 87 -> 14 rotations + head), 1 = lower head (ny = 42 -> 6 rotations + 2 hips)
extern "C" int mmego_head_fk_forward(void* stream, int which, const float* y, const float* body, int B, long F, float* q,
                                     float* joints, const float* Rw, const float* tw, float* world, long long* counters,
                                     int ncount, unsigned long long* seed_ctr) {
  MMEGO_REQUIRE((which == 0 || which == 1) && y && body && q && joints && B > 0 && F > 0);
  MMEGO_REQUIRE(!world || (Rw && tw));
  MMEGO_REQUIRE(ncount >= 0 && ncount <= 4096 && (ncount == 0 || counters));
  if (which == 0) hipLaunchKernelGGL(head_fk_fwd_kernel<0>, dim3(cdiv(F, 64)), dim3(64), 0, (hipStream_t)stream, y, body, B, F, q, joints, Rw, tw, world, counters, ncount, seed_ctr);
  else hipLaunchKernelGGL(head_fk_fwd_kernel<1>, dim3(cdiv(F, 64)), dim3(64), 0, (hipStream_t)stream, y, body, B, F, q, joints, Rw, tw, world, counters, ncount, seed_ctr);
  MMEGO_LAUNCH_CHECK();
  return MMEGO_OK;
}

extern "C" int mmego_head_fk_backward(void* stream, int which, const float* y, const float* body, int B, long F,
                                      const float* dj, float* dy, const float* Rw) {
  MMEGO_REQUIRE((which == 0 || which == 1) && y && body && dj && dy && B > 0 && F > 0);
  if (which == 0) hipLaunchKernelGGL(head_fk_bwd_kernel<0>, dim3(cdiv(F, 64)), dim3(64), 0, (hipStream_t)stream, y, body, B, F, dj, dy, Rw);
  else hipLaunchKernelGGL(head_fk_bwd_kernel<1>, dim3(cdiv(F, 64)), dim3(64), 0, (hipStream_t)stream, y, body, B, F, dj, dy, Rw);
  MMEGO_LAUNCH_CHECK();
  return MMEGO_OK;
}

// head_fk_forward + l1_loss + head_fk_backward as ONE launch (F <= 512 frames; map has the net's nslots entries: the target joint of
// every predicted slot).  dy [F, ny]: gradient of the loss wrt the head output y.
extern "C" int mmego_head_fk_loss(void* stream, int which, const float* y, const float* body, int B, long F, float* q, float* joints_h,
                                  const float* Rw, const float* tw, float* world, long long* counters, int ncount,
                                  unsigned long long* seed_ctr, const float* target, const int* map, int ntgt, double scale,
                                  float* loss, float* dy, double* scratch) {
  // scratch: 2 * ceil(F / 64) + 1 doubles -- the workgroups' partial pairs, then (in the last double's storage) the ticket, which must
  // be 0 before the first call and is left 0 by every call
  MMEGO_REQUIRE((which == 0 || which == 1) && y && body && q && joints_h && Rw && tw && world && target && map && loss && dy && scratch);
  MMEGO_REQUIRE(B > 0 && F > 0 && F <= 65536 && ntgt > 0 && ncount >= 0 && ncount <= 4096 && (ncount == 0 || counters));
  const int nb = cdiv(F, 64);
  unsigned* ticket = reinterpret_cast<unsigned*>(scratch + 2 * nb);
  // (r05 asked for 144 KB of untouched LDS here to keep MFMA workgroups off this kernel's CUs; r06 found the cause class -- packed-fp32
  //  instructions beside a bf16-MFMA workgroup, DESIGN.md section 7d -- and the library is built without them: the request is gone.)
  if (which == 0) hipLaunchKernelGGL(head_fk_loss_kernel<0>, dim3(nb), dim3(64), 0, (hipStream_t)stream, y, body, B, F, q, joints_h, Rw, tw, world, counters, ncount, seed_ctr, target, map, ntgt, (float)scale, loss, dy, scratch, ticket);
  else hipLaunchKernelGGL(head_fk_loss_kernel<1>, dim3(nb), dim3(64), 0, (hipStream_t)stream, y, body, B, F, q, joints_h, Rw, tw, world, counters, ncount, seed_ctr, target, map, ntgt, (float)scale, loss, dy, scratch, ticket);
  MMEGO_LAUNCH_CHECK();
  return MMEGO_OK;
}

extern "C" int mmego_imu_head(void* stream, const float* y, long F, float* R, float* t) {
  MMEGO_REQUIRE(y && R && t && F > 0);
  hipLaunchKernelGGL(imu_head_kernel, dim3(cdiv(F, 128)), dim3(128), 0, (hipStream_t)stream, y, F, R, t);
  MMEGO_LAUNCH_CHECK();
  return MMEGO_OK;
}

extern "C" int mmego_imu_fc2_head(void* stream, const float* X, long ldx, const float* W, const float* b, long F, int K, float* y_out,
                                  float* R, float* t) {
  MMEGO_REQUIRE(X && W && b && R && t && F > 0 && K > 0 && (K % 256) == 0 && ldx >= K && (ldx % 4) == 0);
  MMEGO_REQUIRE((((uintptr_t)X | (uintptr_t)W) & 15) == 0);
  hipLaunchKernelGGL(imu_fc2_head_kernel, dim3((unsigned)cdiv(F, 8)), dim3(256), 0, (hipStream_t)stream, X, ldx, W, b, F, K, y_out, R, t);
  MMEGO_LAUNCH_CHECK();
  return MMEGO_OK;
}

extern "C" int mmego_l1_loss(void* stream, const float* pred, const float* target, const int* map, int nsel, int ntgt,
                             long F, float scale, float* loss, float* grad) {
  MMEGO_REQUIRE(pred && target && map && loss && nsel > 0 && ntgt > 0 && F > 0);
  hipLaunchKernelGGL(l1_loss_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, pred, target, map, nsel, ntgt, F,
                     scale, loss, grad);
  MMEGO_LAUNCH_CHECK();
  return MMEGO_OK;
}

extern "C" int mmego_topk_rows(void* stream, const float* pts, long F, int N, int C, int keep, float* out,
                               long long* idx) {
  MMEGO_REQUIRE(pts && out && idx && F > 0 && N > 0 && C > 0 && keep > 0 && keep <= N && N <= 4096);
  topk_launch((hipStream_t)stream, pts, F, N, C, keep, out, idx, nullptr, 0L, 0);
  MMEGO_LAUNCH_CHECK();
  return MMEGO_OK;
}

// mmego_topk_rows whose kept rows' first n2 columns are also written to out2 (row stride ld2): one launch instead of two
extern "C" int mmego_topk_rows2(void* stream, const float* pts, long F, int N, int C, int keep, float* out, long long* idx, float* out2,
                                long ld2, int n2) {
  MMEGO_REQUIRE(pts && out && idx && out2 && F > 0 && N > 0 && C > 0 && keep > 0 && keep <= N && N <= 4096 && n2 >= 1 && n2 <= C && ld2 >= n2);
  topk_launch((hipStream_t)stream, pts, F, N, C, keep, out, idx, out2, ld2, n2);
  MMEGO_LAUNCH_CHECK();
  return MMEGO_OK;
}

// ---- evaluation metric (Processor/Test/Demo_test.py:64-69,121-123,150-163) ------------------------------------
// Per frame: assemble the 21-joint skeleton (lower overwrites the shared hips), per-joint Euclidean error,
// per-bone angle (deg) between predicted and true bone vectors, mean upper / lower joint error.
// E[f, 0:21] joint errors, E[f, 21:41] bone angles, E[f, 41] upper mean, E[f, 42] lower mean.
__global__ __launch_bounds__(128) void pose_errors_kernel(const float* __restrict__ upper, const float* __restrict__ lower,
                                                          const float* __restrict__ target, long F, float* __restrict__ E) {
  const int umap[15] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 16, 20};
  const int lmap[8] = {12, 13, 14, 15, 16, 17, 18, 19};
  const int bones[20][2] = {{20, 3}, {3, 2}, {2, 1}, {2, 4}, {2, 8}, {4, 5}, {5, 6}, {6, 7}, {8, 9}, {9, 10},
                            {10, 11}, {1, 0}, {0, 12}, {0, 16}, {12, 13}, {13, 14}, {14, 15}, {16, 17}, {17, 18}, {18, 19}};
  long f = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= F) return;
  float p[21][3];
  const float* tg = target + f * 63;
  float up_sum = 0.f, lo_sum = 0.f;
  for (int s = 0; s < 15; ++s) {
    float d2 = 0.f;
    for (int i = 0; i < 3; ++i) {
      float v = upper[(f * 15 + s) * 3 + i];
      p[umap[s]][i] = v;
      float d = v - tg[umap[s] * 3 + i];
      d2 += d * d;
    }
    up_sum += sqrtf(d2);
  }
  for (int s = 0; s < 8; ++s) {
    float d2 = 0.f;
    for (int i = 0; i < 3; ++i) {
      float v = lower[(f * 8 + s) * 3 + i];
      p[lmap[s]][i] = v;
      float d = v - tg[lmap[s] * 3 + i];
      d2 += d * d;
    }
    lo_sum += sqrtf(d2);
  }
  float* e = E + f * 43;
  for (int j = 0; j < 21; ++j) {
    float d2 = 0.f;
    for (int i = 0; i < 3; ++i) { float d = p[j][i] - tg[j * 3 + i]; d2 += d * d; }
    e[j] = sqrtf(d2);
  }
  for (int b = 0; b < 20; ++b) {
    const int r = bones[b][0], l = bones[b][1];
    float dot = 0.f, n1 = 0.f, n2 = 0.f;
    for (int i = 0; i < 3; ++i) {
      float a = p[l][i] - p[r][i], c = tg[l * 3 + i] - tg[r * 3 + i];
      dot += a * c; n1 += a * a; n2 += c * c;
    }
    float cs = dot / (fmaxf(sqrtf(n1), 1e-8f) * fmaxf(sqrtf(n2), 1e-8f));
    cs = fminf(fmaxf(cs, -1.0f), 1.0f);
    e[21 + b] = fabsf(acosf(cs) / 3.14159265358f * 180.0f);
  }
  e[41] = up_sum / 15.f;
  e[42] = lo_sum / 8.f;
}

// Upper-body-only figures of Train_Upper.eval_model (Processor/Train/Train_Upper.py:75-88,228-240), per frame:
// U[f, 0:15] Euclidean error of the 15 upper joints (upper_joint_map order), U[f, 15:29] angle (deg) between predicted and true
// bone vectors of the 14 upper-body bones, U[f, 29] sum |pred - target| over the frame's 45 coordinates (the L1 loss share).
__global__ __launch_bounds__(128) void pose_errors_upper_kernel(const float* __restrict__ upper, const float* __restrict__ target,
                                                                long F, float* __restrict__ U) {
  const int umap[15] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 16, 20};
  // skeleton_upper_body as positions in upper_joint_map: (20,3) (3,2) (2,1) (2,4) (2,8) (4,5) (5,6) (6,7) (8,9) (9,10) (10,11) (1,0) (0,12) (0,16)
  const int bones[14][2] = {{14, 3}, {3, 2}, {2, 1}, {2, 4}, {2, 8}, {4, 5}, {5, 6}, {6, 7}, {8, 9}, {9, 10},
                            {10, 11}, {1, 0}, {0, 12}, {0, 13}};
  long f = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= F) return;
  float p[15][3], q[15][3];
  const float* tg = target + f * 63;
  float l1 = 0.f;
  float* u = U + f * 30;
  for (int s = 0; s < 15; ++s) {
    float d2 = 0.f;
    for (int i = 0; i < 3; ++i) {
      const float v = upper[(f * 15 + s) * 3 + i], w = tg[umap[s] * 3 + i];
      p[s][i] = v; q[s][i] = w;
      const float d = v - w;
      d2 += d * d;
      l1 += fabsf(d);
    }
    u[s] = sqrtf(d2);
  }
  for (int b = 0; b < 14; ++b) {
    const int r = bones[b][0], l = bones[b][1];
    float dot = 0.f, n1 = 0.f, n2 = 0.f;
    for (int i = 0; i < 3; ++i) {
      const float a = p[l][i] - p[r][i], c = q[l][i] - q[r][i];
      dot += a * c; n1 += a * a; n2 += c * c;
    }
    float cs = dot / (fmaxf(sqrtf(n1), 1e-8f) * fmaxf(sqrtf(n2), 1e-8f));
    cs = fminf(fmaxf(cs, -1.0f), 1.0f);
    u[15 + b] = fabsf(acosf(cs) / 3.14159265358f * 180.0f);
  }
  u[29] = l1;
}

extern "C" int mmego_pose_errors_upper(void* stream, const float* upper, const float* target, long F, float* U) {
  MMEGO_REQUIRE(upper && target && U && F > 0);
  hipLaunchKernelGGL(pose_errors_upper_kernel, dim3(cdiv(F, 128)), dim3(128), 0, (hipStream_t)stream, upper, target, F, U);
  MMEGO_LAUNCH_CHECK();
  return MMEGO_OK;
}

extern "C" int mmego_pose_errors(void* stream, const float* upper, const float* lower, const float* target, long F,
                                 float* E) {
  MMEGO_REQUIRE(upper && lower && target && E && F > 0);
  hipLaunchKernelGGL(pose_errors_kernel, dim3(cdiv(F, 128)), dim3(128), 0, (hipStream_t)stream, upper, lower, target, F, E);
  MMEGO_LAUNCH_CHECK();
  return MMEGO_OK;
}
