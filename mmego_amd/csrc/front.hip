// Eval-mode front end of Upper_Net in ONE kernel: Transform2H -> PointNet (6-8-16-24) -> concat with the first four point
// columns -> GlobalPointNet (28-32-48-64) -> softmax attention pooling over the frame's points.
// Replaces, for the frozen / evaluated Upper_Net (reference Net/Upper_Net.py:242-266 PointNet, :270-301 GlobalPointNet incl. its
// attention pooling, :381-393 UpperNet.forward up to the sequence model, Util/Universal_Util/Utils.py:284-292 Transform2H), the
// chain transform2h -> mlp3_eval -> mlp3_eval -> attn_pool_forward, whose 28- and 64-channel per-point tensors went through HBM
// between the launches.  Here a frame's points never leave the CU: per frame the kernel reads the radar tile (N x 6 floats), writes
// the transformed points back (quirk Q1: the caller's tensor is transformed in place and Lower_Net reads it afterwards) and
// emits 64 pooled floats + the N attention weights.
//
// Layout.  One workgroup (4 waves) walks frames; a wave owns 16-point SLABS of the frame (slab s of a frame goes to wave s & 3), and
// a slab runs through all six stages inside its wave: the 16 x K activation tile of a stage is written by the wave that reads it
// (D layout -> [row][k] in wave-private LDS -> A layout), so NO workgroup barrier separates the stages; the (BatchNorm-folded,
// zero-padded) weights are staged once per workgroup and only read afterwards.  v_mfma_f32_16x16x4_f32, K padded to multiples of
// 16, operands fetched with one ds_read_b128 per four MFMA steps through a k-permutation shared by both operands (lane (r, q)
// holds k = 16 c + 4 q + s at step s of chunk c); row strides K + 4 floats keep those 16-byte reads conflict-free.
// The pooling is an online softmax per wave (running max / sum / weighted column sums over its slabs), combined across the four
// waves at the end of the frame in a fixed order: deterministic.
// r06: the stages are computed TRANSPOSED, as in front_bf16.hip -- D^T[feature][point] = W . X^T with the weight tile as the A operand:
// the result (lane (point, q), register i: feature 16 ct + 4 q + i) is what the next stage's MFMA step s wants as its B value (k = 16 c
// + 4 q + s), so ReLU happens in registers, the folded bias is the first MFMA's addend and the activation tiles in LDS are gone; the
// concat keeps PointNet's 24 features at k 0..23 and puts the four point columns behind them (weight columns permuted while staged);
// the pooling sums per lane (its point, its 16 features) and is reduced over the points once per frame.
#include "common.h"

#define FR_SLAB 16
// row strides (floats) of the [n][k] weight tiles and [row][k] activation tiles: Kpad + 4
#define FR_S16 20
#define FR_S32 36
#define FR_S48 52

struct FrontLayer { const float* W; const float* b; const float* gamma; const float* beta; const float* rmean; const float* rvar; };
struct FrontP {
  float* x; const float* x_src; const float* R; const float* t; long F; int N;
  FrontLayer l[6];              // PointNet conv1..3, GlobalPointNet conv1..3, each with its eval-mode BatchNorm
  const float* attn_w; const float* attn_b; float eps;
  float* vec; float* attn;
};

// weights in LDS: [n][k] tiles with stride S; offsets in floats
#define FR_W1 0                               // 16 x 16 (8 x 6 real)
#define FR_W2 (FR_W1 + 16 * FR_S16)           // 16 x 16 (16 x 8)
#define FR_W3 (FR_W2 + 16 * FR_S16)           // 32 x 16 (24 x 16)
#define FR_G1 (FR_W3 + 32 * FR_S16)           // 32 x 32 (32 x 28)
#define FR_G2 (FR_G1 + 32 * FR_S32)           // 48 x 32
#define FR_G3 (FR_G2 + 48 * FR_S32)           // 64 x 48
#define FR_WEND (FR_G3 + 64 * FR_S48)
#define FR_B1 FR_WEND                         // biases: 16, 16, 32, 32, 48, 64; then the 64 score weights
#define FR_B2 (FR_B1 + 16)
#define FR_B3 (FR_B2 + 16)
#define FR_C1 (FR_B3 + 32)
#define FR_C2 (FR_C1 + 32)
#define FR_C3 (FR_C2 + 48)
#define FR_AW (FR_C3 + 64)
#define FR_SHARED_END (FR_AW + 64)
#define FR_MAXN 1024

__device__ __forceinline__ float fr_dot3_nofma(float a0, float a1, float a2, float b0, float b1, float b2) {
  return __fadd_rn(__fadd_rn(__fmul_rn(a0, b0), __fmul_rn(a1, b1)), __fmul_rn(a2, b2));
}

// one stage of a slab, transposed: D^T[NCT*16 features][16 points] = bias + W[NCT*16][KCH*16] . X^T; W in LDS ([n][k], stride SW) as the
// A operand (one ds_read_b128 per four MFMA steps: lane (n, q) holds k = 16 c + 4 q + s at step s), the activations' k blocks in registers
// as the B operand (lane (point, q), component s: k = 16 c + 4 q + s -- the accumulator layout of the stage before)
template <int NCT, int KCH, int SW>
__device__ __forceinline__ void fr_stage_t(const float* W, const f32x4* x, const float* bias4, f32x4 (&acc)[NCT], int fr, int fq) {
#pragma unroll
  for (int ct = 0; ct < NCT; ++ct) acc[ct] = *reinterpret_cast<const f32x4*>(bias4 + ct * 16);
#pragma unroll
  for (int c = 0; c < KCH; ++c) {
    f32x4 w[NCT];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) w[ct] = *reinterpret_cast<const f32x4*>(W + (ct * 16 + fr) * SW + 16 * c + 4 * fq);
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct) acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[ct][s], x[c][s], acc[ct], 0, 0, 0);
  }
}

template <int NCT>
__device__ __forceinline__ void fr_relu_t(const f32x4 (&acc)[NCT], f32x4 (&out)[NCT]) {
#pragma unroll
  for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
    for (int i = 0; i < 4; ++i) out[ct][i] = fmaxf(acc[ct][i], 0.f);
}

__global__ __launch_bounds__(256) void upper_front_eval_kernel(FrontP p) {
  __shared__ __attribute__((aligned(16))) float sh[FR_SHARED_END];
  __shared__ float scale_s[208];                   // per-channel BatchNorm scales while the weights are folded
  __shared__ float sc[FR_MAXN];                    // raw scores of the frame's points
  __shared__ float comb[2][4][66];                 // per wave: running max, running sum, 64 weighted column sums (double buffered)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fq = lane >> 4;

  // ---- weights -> LDS (BatchNorm folded: s = gamma / sqrt(var + eps); Wf = s W; bf = (b - mean) s + beta -- bn_fold_linear's
  // expressions), zero padded.  Two phases, as in mlp3.hip: per-channel scales and folded biases (threads 0..63, all 30 loads in
  // flight at once), then every thread's 26 weight elements -- every load unconditional on a clamped index and issued before the
  // first LDS store (a rolled loop was one dependent round trip per iteration: ~26 of them in front of the first point).
#define FR_PIN(v) asm volatile("" : "+v"(v))
  {
    constexpr int Cn[6] = {8, 16, 24, 32, 48, 64}, Kn[6] = {6, 8, 16, 28, 32, 48};
    constexpr int Cp[6] = {16, 16, 32, 32, 48, 64}, Kp[6] = {16, 16, 16, 32, 32, 48};
    constexpr int Sw[6] = {FR_S16, FR_S16, FR_S16, FR_S32, FR_S32, FR_S48};
    constexpr int Wo[6] = {FR_W1, FR_W2, FR_W3, FR_G1, FR_G2, FR_G3}, Bo[6] = {FR_B1, FR_B2, FR_B3, FR_C1, FR_C2, FR_C3};
    constexpr int So[6] = {0, 16, 32, 64, 96, 144};                    // per-channel scales
    float* const scale = scale_s;
    if (tid < 64) {
      float g[6], v[6], m[6], e[6], c[6];
#pragma unroll
      for (int L = 0; L < 6; ++L) {
        const FrontLayer& q = p.l[L];
        const int nc = min(tid, Cn[L] - 1);
        g[L] = q.gamma[nc]; v[L] = q.rvar[nc]; m[L] = q.rmean[nc]; e[L] = q.beta[nc]; c[L] = q.b[nc];
      }
#pragma unroll
      for (int L = 0; L < 6; ++L) {
        FR_PIN(g[L]); FR_PIN(v[L]); FR_PIN(m[L]); FR_PIN(e[L]); FR_PIN(c[L]);
        const float sc_ = g[L] / sqrtf(v[L] + p.eps);
        const float bf = (c[L] - m[L]) * sc_ + e[L];
        if (tid < Cp[L]) { scale[So[L] + tid] = sc_; sh[Bo[L] + tid] = tid < Cn[L] ? bf : 0.f; }
      }
      sh[FR_AW + tid] = p.attn_w[tid];
    }
    __syncthreads();
    float w[26];
    int u0 = 0;
#pragma unroll
    for (int L = 0; L < 6; ++L) {
#pragma unroll
      for (int u = 0; u < Cp[L] * Kp[L] / 256; ++u) {
        const int i = tid + 256 * u, n = i / Kp[L], k = i - n * Kp[L];
        // GlobalPointNet conv1 (L = 3): tile column k < 24 holds input column 4 + k (PointNet feature k), 24..27 the point columns 0..3
        const int ks = L == 3 ? (k < 24 ? k + 4 : (k < 28 ? k - 24 : Kn[L] - 1)) : min(k, Kn[L] - 1);
        w[u0 + u] = p.l[L].W[min(n, Cn[L] - 1) * Kn[L] + ks];
      }
      u0 += Cp[L] * Kp[L] / 256;
    }
    u0 = 0;
#pragma unroll
    for (int L = 0; L < 6; ++L) {
#pragma unroll
      for (int u = 0; u < Cp[L] * Kp[L] / 256; ++u) {
        const int i = tid + 256 * u, n = i / Kp[L], k = i - n * Kp[L];
        FR_PIN(w[u0 + u]);
        sh[Wo[L] + n * Sw[L] + k] = (n < Cn[L] && k < Kn[L]) ? scale[So[L] + n] * w[u0 + u] : 0.f;
      }
      u0 += Cp[L] * Kp[L] / 256;
    }
  }
#undef FR_PIN
  const float attn_b = p.attn_b ? p.attn_b[0] : 0.f;
  __syncthreads();

  const int N = p.N, nslab = N / FR_SLAB;
  int par = 0;
  for (long f = blockIdx.x; f < p.F; f += gridDim.x, par ^= 1) {
    const float* Rf = p.R + f * 9;
    const float* tf = p.t + f * 3;
    float r[9], tt[3];
#pragma unroll
    for (int i = 0; i < 9; ++i) r[i] = Rf[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) tt[i] = tf[i];
    float* xf = p.x + f * (long)N * 6;
    const float* xs = p.x_src ? p.x_src + f * (long)N * 6 : xf;
    // running softmax state of this wave: the maximum (wave-uniform) and, per lane, its point's share of the denominator and of the
    // 16 weighted feature sums it holds (features 16 ct + 4 fq + i); the lanes are added once per frame
    float m_run = -INFINITY, s_part = 0.f;
    float col[4][4];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
      for (int i = 0; i < 4; ++i) col[ct][i] = 0.f;
    // the slab's 16 points: every lane loads the row of point (lane & 15) (24 bytes; the four 16-lane groups load the same rows:
    // no branch around the loads), and the NEXT slab's rows are requested before the current slab is computed
    float2 c01, c23, c45;
    {
      const long row = (long)min(wave, nslab - 1) * FR_SLAB + fr;
      c01 = *reinterpret_cast<const float2*>(xs + row * 6);
      c23 = *reinterpret_cast<const float2*>(xs + row * 6 + 2);
      c45 = *reinterpret_cast<const float2*>(xs + row * 6 + 4);
    }
    for (int s = wave; s < nslab; s += 4) {
      const float2 v01 = c01, v23 = c23, v45 = c45;
      {
        const long rown = (long)(s + 4 < nslab ? s + 4 : s) * FR_SLAB + fr;       // (past the last slab: this slab again)
        c01 = *reinterpret_cast<const float2*>(xs + rown * 6);
        c23 = *reinterpret_cast<const float2*>(xs + rown * 6 + 2);
        c45 = *reinterpret_cast<const float2*>(xs + rown * 6 + 4);
      }
      const float d0 = __fsub_rn(v01.x, tt[0]), d1 = __fsub_rn(v01.y, tt[1]), d2 = __fsub_rn(v23.x, tt[2]);
      const float h0 = fr_dot3_nofma(r[0], r[1], r[2], d0, d1, d2);
      const float h1 = fr_dot3_nofma(r[3], r[4], r[5], d0, d1, d2);
      const float h2 = fr_dot3_nofma(r[6], r[7], r[8], d0, d1, d2);
      if (lane < FR_SLAB) {
        const long row = (long)s * FR_SLAB + lane;
        *reinterpret_cast<float2*>(xf + row * 6) = make_float2(h0, h1);
        *reinterpret_cast<float2*>(xf + row * 6 + 2) = make_float2(h2, v23.y);
        if (p.x_src) *reinterpret_cast<float2*>(xf + row * 6 + 4) = v45;
      }
      // the point's six columns as the first stage's B operand (k = 4 fq + s): group 0: h0 h1 h2 x3, group 1: x4 x5 0 0, groups 2, 3: 0
      const f32x4 xcols = {h0, h1, h2, v23.y};
      const f32x4 x45 = {v45.x, v45.y, 0.f, 0.f};
      const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
      const f32x4 bx = fq == 0 ? xcols : fq == 1 ? x45 : zero4;
      f32x4 a1[1], a2[1], a3[2], g1[2], g2[3], g3[4];
      f32x4 p1[1], p2[1], f01[2], q01[2], r012[3];
      // (the biases and score weights are read from LDS in every slab: the offset is opaque to the compiler, which would otherwise keep
      //  all 17 float4 of them in registers across the loop)
      int bo = 4 * fq;
      asm volatile("" : "+v"(bo));
      const float* const b4 = sh + bo;
      fr_stage_t<1, 1, FR_S16>(sh + FR_W1, &bx, b4 + FR_B1, a1, fr, fq);
      fr_relu_t<1>(a1, p1);
      fr_stage_t<1, 1, FR_S16>(sh + FR_W2, p1, b4 + FR_B2, a2, fr, fq);
      fr_relu_t<1>(a2, p2);
      fr_stage_t<2, 1, FR_S16>(sh + FR_W3, p2, b4 + FR_B3, a3, fr, fq);
      fr_relu_t<2>(a3, f01);
      // concat: features 0..15 | features 16..23, the four point columns, padding (the stage's padded outputs 24..31 are zero)
      if (fq == 2) f01[1] = xcols;
      fr_stage_t<2, 2, FR_S32>(sh + FR_G1, f01, b4 + FR_C1, g1, fr, fq);
      fr_relu_t<2>(g1, q01);
      fr_stage_t<3, 2, FR_S32>(sh + FR_G2, q01, b4 + FR_C2, g2, fr, fq);
      fr_relu_t<3>(g2, r012);
      fr_stage_t<4, 3, FR_S48>(sh + FR_G3, r012, b4 + FR_C3, g3, fr, fq);
      // ---- scores and the online softmax update.  Lane (fr, fq), register i of tile ct: feature 16 ct + 4 fq + i of point fr.
      float y[4][4], part = 0.f;
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) {
        const f32x4 wv = *reinterpret_cast<const f32x4*>(b4 + FR_AW + ct * 16);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          y[ct][i] = fmaxf(g3[ct][i], 0.f);
          part += y[ct][i] * wv[i];
        }
      }
      part += __shfl_xor(part, 16, 64);
      part += __shfl_xor(part, 32, 64);                                          // sum over the four feature groups: the point's score
      part += attn_b;
      if (fq == 0) sc[s * FR_SLAB + fr] = part;
      float smax = part;
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) smax = fmaxf(smax, __shfl_xor(smax, o, 64));   // max over the slab's 16 points
      const float m_new = fmaxf(m_run, smax);
      if (m_new != m_run) {                                                      // wave-uniform; rare after a frame's first slabs
        const float resc = __expf(m_run - m_new);                                // (exp(-inf) = 0 on the first slab)
        s_part *= resc;
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
#pragma unroll
          for (int i = 0; i < 4; ++i) col[ct][i] *= resc;
        m_run = m_new;
      }
      const float e = __expf(part - m_run);
      s_part += e;
#pragma unroll
      for (int ct = 0; ct < 4; ++ct)
#pragma unroll
        for (int i = 0; i < 4; ++i) col[ct][i] += e * y[ct][i];
    }
    // ---- add the 16 points' lanes (once per frame), then combine the four waves (fixed order) and emit the frame's outputs
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) {
      s_part += __shfl_xor(s_part, o, 64);
#pragma unroll
      for (int ct = 0; ct < 4; ++ct)
#pragma unroll
        for (int i = 0; i < 4; ++i) col[ct][i] += __shfl_xor(col[ct][i], o, 64);
    }
    if (fr == 0) {
#pragma unroll
      for (int ct = 0; ct < 4; ++ct)
#pragma unroll
        for (int i = 0; i < 4; ++i) comb[par][wave][2 + ct * 16 + 4 * fq + i] = col[ct][i];
      if (lane == 0) { comb[par][wave][0] = m_run; comb[par][wave][1] = s_part; }
    }
    __syncthreads();
    float M = comb[par][0][0];
#pragma unroll
    for (int w = 1; w < 4; ++w) M = fmaxf(M, comb[par][w][0]);
    float S = 0.f, sw[4];
#pragma unroll
    for (int w = 0; w < 4; ++w) { sw[w] = __expf(comb[par][w][0] - M); S += comb[par][w][1] * sw[w]; }
    const float inv = 1.0f / S;
    if (tid < 64) {
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) v += comb[par][w][2 + tid] * sw[w];
      p.vec[f * 64 + tid] = v * inv;
    }
    for (int n = tid; n < N; n += 256) p.attn[f * (long)N + n] = __expf(sc[n] - M) * inv;
    // (sc is rewritten by the next frame's slabs: every wave must be past the loop above first; comb is double buffered)
    __syncthreads();
  }
}

// w: host-side table of 38 device pointers: for PointNet conv1..3 then GlobalPointNet conv1..3: W, b, gamma, beta, running_mean,
// running_var of the layer's BatchNorm; then the attention Linear's weight [64] and bias [1].
extern "C" int mmego_upper_front_eval(void* stream, float* x, const float* x_src, const float* R, const float* t, long F, int N,
                                      const float* const* w, float eps, float* vec, float* attn) {
  MMEGO_REQUIRE(x && R && t && w && vec && attn && F > 0);
  MMEGO_REQUIRE(N >= FR_SLAB && N <= FR_MAXN && N % FR_SLAB == 0);
  MMEGO_REQUIRE((((uintptr_t)x | (uintptr_t)x_src) & 7) == 0);
  FrontP p;
  p.x = x; p.x_src = x_src; p.R = R; p.t = t; p.F = F; p.N = N;
  for (int L = 0; L < 6; ++L) {
    for (int j = 0; j < 6; ++j) MMEGO_REQUIRE(w[6 * L + j]);
    p.l[L] = {w[6 * L], w[6 * L + 1], w[6 * L + 2], w[6 * L + 3], w[6 * L + 4], w[6 * L + 5]};
  }
  MMEGO_REQUIRE(w[36]);
  p.attn_w = w[36]; p.attn_b = w[37]; p.eps = eps; p.vec = vec; p.attn = attn;
  const unsigned grid = (unsigned)(F < 1024 ? F : 1024);        // ~34 KB of LDS: four workgroups per CU, frames walked persistently
  hipLaunchKernelGGL(upper_front_eval_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, p);
  MMEGO_LAUNCH_CHECK();
  return MMEGO_OK;
}
