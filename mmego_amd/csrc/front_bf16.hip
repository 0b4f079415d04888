// bf16-operand form of front.hip's upper_front_eval (opt-in precision mode, eval forwards: UpperNet.precision = "bf16"; BASELINE
// config 5 "bf16 forward / fp32 accumulate"): same dataflow -- Transform2H -> PointNet (6-8-16-24) -> concat -> GlobalPointNet
// (28-32-48-64) -> softmax attention pooling, a frame's points never leaving the CU (reference Net/Upper_Net.py:242-301, :381-393,
// Util/Universal_Util/Utils.py:284-292) -- with the six stages' operands in bf16: the (BatchNorm-folded) weights are rounded once
// while they are staged, every activation tile is rounded when it is written for the next stage, products are exact in fp32 and
// accumulate in fp32 on v_mfma_f32_16x16x16_bf16 (one MFMA per 16 k instead of four 16x16x4 fp32 steps: an eighth of the matrix
// time).  Transform2H, the write-back of the transformed points (Q1), biases, ReLU, scores and the online softmax stay fp32.
// Lane (r, q) of a 16x16x16 MFMA holds k = 4 q .. 4 q + 3 of row / column r: one 8-byte LDS read per operand and 16 k.
// r06: the stages are computed TRANSPOSED -- D^T[feature][point] = W[feature][k] . X^T[k][point], the weight tile as the A operand --
// because the result layout of that product (lane (point, q), register i: feature 4 q + i) IS the B-operand layout of the next stage's
// MFMA (lane (point, q): k = 4 q .. 4 q + 3): bias, ReLU and the bf16 rounding happen in registers and a slab's activations never
// touch LDS (before: 36 two-byte LDS stores and the reads behind them per slab and wave, a latency chain of six round trips; 591 us at
// config 5, VALU / LDS issue bound).  The concat of GlobalPointNet's input keeps PointNet's 24 features where they are (k 0..23) and
// puts the four point columns BEHIND them (k 24..27); the weight columns of that layer are permuted to match when they are staged.
// The softmax pooling accumulates per lane (its point, its 16 features) and is reduced over the points once per frame.
#include "common.h"

typedef unsigned short bf16_t;
typedef short s16x4 __attribute__((ext_vector_type(4)));

// (a float -> __bf16 conversion is v_cvt_pk_bf16_f32 on gfx950: round to nearest even, NaN stays NaN -- one instruction for two
// values where the integer form took five per value)
typedef float lo_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 lo_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned int frb_bf(float x) { return (unsigned int)__builtin_bit_cast(unsigned short, (__bf16)x); }
__device__ __forceinline__ unsigned int frb_bf2(float lo, float hi) {
  return __builtin_bit_cast(unsigned int, __builtin_convertvector((lo_f32x2){lo, hi}, lo_bf16x2));
}
__device__ __forceinline__ uint2 frb_pack4(float a, float b, float c, float d) {
  return make_uint2(frb_bf2(a, b), frb_bf2(c, d));
}

#define FR_SLAB 16
// row strides (bf16 elements) of the [n][k] weight tiles and [row][k] activation tiles: Kpad + 4 (8-byte aligned rows)
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

// weights in LDS: [n][k] bf16 tiles with stride S; offsets in elements (biases and score weights: a float array of their own)
#define FR_W1 0                               // 16 x 16 (8 x 6 real)
#define FR_W2 (FR_W1 + 16 * FR_S16)           // 16 x 16 (16 x 8)
#define FR_W3 (FR_W2 + 16 * FR_S16)           // 32 x 16 (24 x 16)
#define FR_G1 (FR_W3 + 32 * FR_S16)           // 32 x 32 (32 x 28)
#define FR_G2 (FR_G1 + 32 * FR_S32)           // 48 x 32
#define FR_G3 (FR_G2 + 48 * FR_S32)           // 64 x 48
#define FR_WEND (FR_G3 + 64 * FR_S48)
#define FR_B1 0                               // biases: 16, 16, 32, 32, 48, 64; then the 64 score weights
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

// one stage of a slab, transposed: D^T[NCT*16 features][16 points] = bias + W[NCT*16][KCH*16] . X^T, W in LDS ([n][k] bf16, stride SW) as
// the A operand, the activations' k blocks in registers as the B operand (lane (point fr, fq): k = 16 c + 4 fq .. + 3); the folded bias
// (lane (point, fq), register i: feature 16 ct + 4 fq + i -- the accumulator's own layout) is the first MFMA's addend: no add afterwards
template <int NCT, int KCH, int SW>
__device__ __forceinline__ void fr_stage_t(const bf16_t* W, const s16x4* x, const float* bias4, f32x4 (&acc)[NCT], int fr, int fq) {
#pragma unroll
  for (int ct = 0; ct < NCT; ++ct) acc[ct] = *reinterpret_cast<const f32x4*>(bias4 + ct * 16);
#pragma unroll
  for (int c = 0; c < KCH; ++c) {
    s16x4 w[NCT];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) w[ct] = *reinterpret_cast<const s16x4*>(W + (ct * 16 + fr) * SW + 16 * c + 4 * fq);
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) acc[ct] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(w[ct], x[c], acc[ct], 0, 0, 0);
  }
}

// max(x, 0).  One v_max_f32 only because build.py compiles this file with -fno-honor-nans: otherwise the compiler first canonicalises
// an MFMA result it cannot prove quiet (a second v_max per value; it rewrites v_med3 the same way).  An inline-asm v_max_f32 is NOT an
// option: the hazard recogniser does not see an MFMA result being read inside an asm block and leaves out the wait states (measured:
// results changing from run to run).
__device__ __forceinline__ float fr_relu(float x) { return fmaxf(x, 0.f); }

// relu(D^T) rounded to bf16: accumulator tile ct (lane (point, fq), register i: feature 16 ct + 4 fq + i) IS k block ct of the next
// stage's B operand
template <int NCT>
__device__ __forceinline__ void fr_pack_t(const f32x4 (&acc)[NCT], s16x4 (&out)[NCT]) {
#pragma unroll
  for (int ct = 0; ct < NCT; ++ct)
    out[ct] = __builtin_bit_cast(s16x4, frb_pack4(fr_relu(acc[ct][0]), fr_relu(acc[ct][1]), fr_relu(acc[ct][2]), fr_relu(acc[ct][3])));
}

#ifndef FRONT_BF16_WAVES
#define FRONT_BF16_WAVES 0
#endif
#if FRONT_BF16_WAVES
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(FRONT_BF16_WAVES, FRONT_BF16_WAVES))) void upper_front_eval_bf16_kernel(FrontP p) {
#else
__global__ __launch_bounds__(256) void upper_front_eval_bf16_kernel(FrontP p) {
#endif
  __shared__ __attribute__((aligned(16))) bf16_t shw[FR_WEND];
  __shared__ __attribute__((aligned(16))) float sh[FR_SHARED_END];
  __shared__ float scale[208];                      // per-channel BatchNorm scales while the weights are folded
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
        shw[Wo[L] + n * Sw[L] + k] = (bf16_t)frb_bf((n < Cn[L] && k < Kn[L]) ? scale[So[L] + n] * w[u0 + u] : 0.f);
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
      // the point's six columns as the first stage's B operand (k = 4 fq + i): group 0: h0 h1 h2 x3, group 1: x4 x5 0 0, groups 2, 3: 0
      const uint2 xcols = frb_pack4(h0, h1, h2, v23.y);
      const uint2 x45 = frb_pack4(v45.x, v45.y, 0.f, 0.f);
      const uint2 zero2 = make_uint2(0u, 0u);
      const s16x4 bx = __builtin_bit_cast(s16x4, fq == 0 ? xcols : fq == 1 ? x45 : zero2);
      f32x4 a1[1], a2[1], a3[2], g1[2], g2[3], g3[4];
      s16x4 p1[1], p2[1], f01[2], q01[2], r012[3];
      // (the biases and score weights are read from LDS in every slab -- the offset below is opaque to the compiler, which would
      //  otherwise keep all 17 float4 of them in registers across the loop: 68 VGPRs, a wave per SIMD less)
      int bo = 4 * fq;
      asm volatile("" : "+v"(bo));
      const float* const b4 = sh + bo;
      fr_stage_t<1, 1, FR_S16>(shw + FR_W1, &bx, b4 + FR_B1, a1, fr, fq);
      fr_pack_t<1>(a1, p1);
      fr_stage_t<1, 1, FR_S16>(shw + FR_W2, p1, b4 + FR_B2, a2, fr, fq);
      fr_pack_t<1>(a2, p2);
      fr_stage_t<2, 1, FR_S16>(shw + FR_W3, p2, b4 + FR_B3, a3, fr, fq);
      fr_pack_t<2>(a3, f01);
      // concat: features 0..15 | features 16..23, the four point columns, padding (the stage's padded outputs 24..31 are zero)
      if (fq == 2) f01[1] = __builtin_bit_cast(s16x4, xcols);
      fr_stage_t<2, 2, FR_S32>(shw + FR_G1, f01, b4 + FR_C1, g1, fr, fq);
      fr_pack_t<2>(g1, q01);
      fr_stage_t<3, 2, FR_S32>(shw + FR_G2, q01, b4 + FR_C2, g2, fr, fq);
      fr_pack_t<3>(g2, r012);
      fr_stage_t<4, 3, FR_S48>(shw + FR_G3, r012, b4 + FR_C3, g3, fr, fq);
      // ---- scores and the online softmax update.  Lane (fr, fq), register i of tile ct: feature 16 ct + 4 fq + i of point fr.
      float y[4][4], part = 0.f;
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) {
        const f32x4 wv = *reinterpret_cast<const f32x4*>(b4 + FR_AW + ct * 16);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          y[ct][i] = fr_relu(g3[ct][i]);
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
extern "C" int mmego_upper_front_eval_bf16(void* stream, float* x, const float* x_src, const float* R, const float* t, long F, int N,
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
  const unsigned grid = (unsigned)(F < 1024 ? F : 1024);        // ~40 KB of LDS: four workgroups per CU, frames walked persistently
  hipLaunchKernelGGL(upper_front_eval_bf16_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, p);
  MMEGO_LAUNCH_CHECK();
  return MMEGO_OK;
}
