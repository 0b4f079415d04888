// IMU_Net stage-1 training pieces: LSTM cell backward (pointwise), geodesic + position loss, IMU head backward.
// Reference: Processor/Train/Train_IMU.py:21-34 (GeodesicLoss), :138-141 (loss), Net/IMU_Net.py:7-47 (6-D head).
// The recurrent products of the backward pass (dh_{t-1} = dgates . W_hh, dW = dgates^T . [x, h]) are mmego_gemm calls.
#include "common.h"

struct CellBwdP {
  const float* dout[2]; long dos;
  const float* dhrec[2];
  const float* gst[2]; const float* cst[2]; const float* cprev[2];
  float* dc[2];
  float* dgates[2]; long dgs;
  int Bn, H;
};

__global__ __launch_bounds__(256) void lstm_cell_bwd_kernel(CellBwdP p) {
  const int d = blockIdx.y;
  const long n = (long)p.Bn * p.H;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const long row = i / p.H;
    const int j = (int)(i - row * p.H);
    float dh = p.dout[d][row * p.dos + j];
    if (p.dhrec[d]) dh += p.dhrec[d][i];
    const float* gs = p.gst[d] + row * 4 * p.H + j;
    const float gi = gs[0], gf = gs[p.H], gg = gs[2 * p.H], go = gs[3 * p.H];
    const float c = p.cst[d][i];
    const float cprev = p.cprev[d] ? p.cprev[d][i] : 0.f;
    const float tc = tanhf(c);
    const float dcv = p.dc[d][i] + dh * go * (1.f - tc * tc);
    float* dg = p.dgates[d] + row * p.dgs + j;
    dg[0] = dcv * gg * gi * (1.f - gi);
    dg[p.H] = dcv * cprev * gf * (1.f - gf);
    dg[2 * p.H] = dcv * gi * (1.f - gg * gg);
    dg[3 * p.H] = dh * tc * go * (1.f - go);
    p.dc[d][i] = dcv * gf;
  }
}

// single block: deterministic sum
__global__ __launch_bounds__(1024) void imu_loss_kernel(const float* __restrict__ R, const float* __restrict__ t,
                                                        const float* __restrict__ Rg, const float* __restrict__ hg, long F,
                                                        float scale, float* __restrict__ loss, float* __restrict__ dR,
                                                        float* __restrict__ dt) {
  __shared__ double sh[16];
  const float eps = 1e-7f, rad2deg = 180.0f / 3.14159265358f;
  double acc = 0.0;
  for (long f = threadIdx.x; f < F; f += blockDim.x) {
    const float* a = R + f * 9;
    const float* b = Rg + f * 9;
    float tr = 0.f;
    for (int k = 0; k < 9; ++k) tr += a[k] * b[k];                 // trace(R Rgt^T) = sum_ij R_ij Rgt_ij
    const float cosv = (tr - 1.f) * 0.5f;
    const float cc = fminf(fmaxf(cosv, -1.f + eps), 1.f - eps);
    acc += (double)(acosf(cc) * rad2deg);
    // d acos(c)/dc = -1/sqrt(1-c^2) inside the clamp, 0 outside; dc/dR_ij = Rgt_ij / 2
    const float g = (cosv > -1.f + eps && cosv < 1.f - eps) ? (-rad2deg / sqrtf(1.f - cc * cc)) * 0.5f * scale : 0.f;
    if (dR) for (int k = 0; k < 9; ++k) dR[f * 9 + k] = g * b[k];
    float dx = t[f * 3] - hg[f * 3], dy = t[f * 3 + 1] - hg[f * 3 + 1], dz = t[f * 3 + 2] - hg[f * 3 + 2];
    const float nrm = sqrtf(dx * dx + dy * dy + dz * dz);
    acc += 100.0 * (double)nrm;
    if (dt) {
      const float s = nrm > 0.f ? 100.f * scale / nrm : 0.f;
      dt[f * 3] = s * dx; dt[f * 3 + 1] = s * dy; dt[f * 3 + 2] = s * dz;
    }
  }
  acc = wave_sum_d(acc);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double s = 0.0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s += sh[w];
    *loss = (float)s;
  }
}

__device__ __forceinline__ void cross3i(const float* u, const float* v, float* o) {
  o[0] = u[1] * v[2] - u[2] * v[1];
  o[1] = u[2] * v[0] - u[0] * v[2];
  o[2] = u[0] * v[1] - u[1] * v[0];
}

// y[f,0:6] -> R columns x,y,z with v/max(|v|,1e-8) (IMU_Net.py:7-18); y[f,6:9] = t.  (dR, dt) -> dy
__global__ __launch_bounds__(128) void imu_head_bwd_kernel(const float* __restrict__ y, const float* __restrict__ dR,
                                                           const float* __restrict__ dt, long F, float* __restrict__ dy) {
  long f = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= F) return;
  const float eps = 1e-8f;
  const float* a = y + f * 9;
  const float* b = a + 3;
  const float* g = dR + f * 9;
  float na = sqrtf(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);
  float nac = fmaxf(na, eps);
  float x[3] = {a[0] / nac, a[1] / nac, a[2] / nac};
  float w[3];
  cross3i(x, b, w);
  float nw = sqrtf(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
  float nwc = fmaxf(nw, eps);
  float z[3] = {w[0] / nwc, w[1] / nwc, w[2] / nwc};
  float gx[3] = {g[0], g[3], g[6]}, gy[3] = {g[1], g[4], g[7]}, gz[3] = {g[2], g[5], g[8]};
  float tmp[3];
  cross3i(x, gy, tmp);                        // y = z cross x
  for (int i = 0; i < 3; ++i) gz[i] += tmp[i];
  cross3i(gy, z, tmp);
  for (int i = 0; i < 3; ++i) gx[i] += tmp[i];
  float gw[3];
  if (nw > eps) {
    float dz = z[0] * gz[0] + z[1] * gz[1] + z[2] * gz[2];
    for (int i = 0; i < 3; ++i) gw[i] = (gz[i] - z[i] * dz) / nwc;
  } else {
    for (int i = 0; i < 3; ++i) gw[i] = gz[i] / nwc;
  }
  cross3i(b, gw, tmp);                        // w = x cross b
  for (int i = 0; i < 3; ++i) gx[i] += tmp[i];
  float* o = dy + f * 9;
  cross3i(gw, x, o + 3);
  if (na > eps) {
    float dx = x[0] * gx[0] + x[1] * gx[1] + x[2] * gx[2];
    for (int i = 0; i < 3; ++i) o[i] = (gx[i] - x[i] * dx) / nac;
  } else {
    for (int i = 0; i < 3; ++i) o[i] = gx[i] / nac;
  }
  for (int i = 0; i < 3; ++i) o[6 + i] = dt[f * 3 + i];
}

extern "C" int mmego_lstm_cell_backward(void* stream, int ndir, int Bn, int H, const float* dout0, const float* dout1, long dos,
                                        const float* dhrec0, const float* dhrec1, const float* gst0, const float* gst1,
                                        const float* cst0, const float* cst1, const float* cprev0, const float* cprev1,
                                        float* dc0, float* dc1, float* dgates0, float* dgates1, long dgs) {
  MMEGO_REQUIRE((ndir == 1 || ndir == 2) && Bn > 0 && H > 0 && dout0 && gst0 && cst0 && dc0 && dgates0);
  if (ndir == 2) MMEGO_REQUIRE(dout1 && gst1 && cst1 && dc1 && dgates1);
  CellBwdP p;
  p.dout[0] = dout0; p.dout[1] = dout1; p.dos = dos;
  p.dhrec[0] = dhrec0; p.dhrec[1] = dhrec1;
  p.gst[0] = gst0; p.gst[1] = gst1; p.cst[0] = cst0; p.cst[1] = cst1;
  p.cprev[0] = cprev0; p.cprev[1] = cprev1;
  p.dc[0] = dc0; p.dc[1] = dc1;
  p.dgates[0] = dgates0; p.dgates[1] = dgates1; p.dgs = dgs;
  p.Bn = Bn; p.H = H;
  long n = (long)Bn * H;
  int blocks = (int)((n + 255) / 256);
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(lstm_cell_bwd_kernel, dim3(blocks, ndir), dim3(256), 0, (hipStream_t)stream, p);
  MMEGO_LAUNCH_CHECK();
  return MMEGO_OK;
}

extern "C" int mmego_imu_loss(void* stream, const float* R, const float* t, const float* R_gt, const float* head_gt, long F,
                              float scale, float* loss, float* dR, float* dt) {
  MMEGO_REQUIRE(R && t && R_gt && head_gt && loss && F > 0);
  hipLaunchKernelGGL(imu_loss_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, R, t, R_gt, head_gt, F, scale, loss, dR, dt);
  MMEGO_LAUNCH_CHECK();
  return MMEGO_OK;
}

extern "C" int mmego_imu_head_backward(void* stream, const float* y, const float* dR, const float* dt, long F, float* dy) {
  MMEGO_REQUIRE(y && dR && dt && dy && F > 0);
  hipLaunchKernelGGL(imu_head_bwd_kernel, dim3(cdiv(F, 128)), dim3(128), 0, (hipStream_t)stream, y, dR, dt, F, dy);
  MMEGO_LAUNCH_CHECK();
  return MMEGO_OK;
}
