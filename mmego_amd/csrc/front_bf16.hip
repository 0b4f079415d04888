// bf16-operand form of front.hip's upper_front_eval (opt-in precision mode, eval forwards: UpperNet.precision = "bf16"; BASELINE
// config 5 "bf16 forward / fp32 accumulate"): same dataflow -- Transform2H -> PointNet (6-8-16-24) -> concat -> GlobalPointNet
// (28-32-48-64) -> softmax attention pooling, a frame's points never leaving the CU (reference Net/Upper_Net.py:242-301, :381-393,
// Util/Universal_Util/Utils.py:284-292) -- with the six stages' operands in bf16: the (BatchNorm-folded) weights are rounded once
// while they are staged, every activation tile is rounded when it is written for the next stage, products are exact in fp32 and
// accumulate in fp32 on v_mfma_f32_16x16x16_bf16 (one MFMA per 16 k instead of four 16x16x4 fp32 steps: an eighth of the matrix
// time).  Transform2H, the write-back of the transformed points (Q1), biases, ReLU, scores and the online softmax stay fp32.
// Lane (r, q) of a 16x16x16 MFMA holds k = 4 q .. 4 q + 3 of row / column r: one 8-byte LDS read per operand and 16 k.
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
// wave-private activation tiles: P [16][20] | Q [16][20] (together also G1 [16][36]) | FZ [16][36] | G2 [16][52]
#define FR_ACT_P 0
#define FR_ACT_Q (16 * FR_S16)
#define FR_ACT_FZ (2 * 16 * FR_S16)
#define FR_ACT_G2 (FR_ACT_FZ + 16 * FR_S32)
#define FR_ACT_WAVE (FR_ACT_G2 + 16 * FR_S48)
#define FR_MAXN 1024

__device__ __forceinline__ float fr_dot3_nofma(float a0, float a1, float a2, float b0, float b1, float b2) {
  return __fadd_rn(__fadd_rn(__fmul_rn(a0, b0), __fmul_rn(a1, b1)), __fmul_rn(a2, b2));
}

// one stage of a slab: D[16 rows][NCT*16 cols] = A[16][KCH*16] . W[NCT*16][KCH*16]^T, both operands bf16 in LDS ([row][k], [n][k])
template <int NCT, int KCH, int SA, int SW>
__device__ __forceinline__ void fr_stage(const bf16_t* A, const bf16_t* W, f32x4 (&acc)[NCT], int fr, int fq) {
#pragma unroll
  for (int ct = 0; ct < NCT; ++ct) acc[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int c = 0; c < KCH; ++c) {
    const s16x4 a = *reinterpret_cast<const s16x4*>(A + fr * SA + 16 * c + 4 * fq);
    s16x4 b[NCT];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) b[ct] = *reinterpret_cast<const s16x4*>(W + (ct * 16 + fr) * SW + 16 * c + 4 * fq);
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) acc[ct] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b[ct], acc[ct], 0, 0, 0);
  }
}

// relu(D + bias) -> [row][col0 + ...] of the next stage's A tile (lane (fr, fq), register i: row 4 fq + i, column 16 ct + fr)
template <int NCT, int SD>
__device__ __forceinline__ void fr_store(const f32x4 (&acc)[NCT], const float* bias, bf16_t* D, int col0, int fr, int fq) {
#pragma unroll
  for (int ct = 0; ct < NCT; ++ct) {
    const float bv = bias[ct * 16 + fr];
#pragma unroll
    for (int i = 0; i < 4; ++i) D[(4 * fq + i) * SD + col0 + ct * 16 + fr] = (bf16_t)frb_bf(fmaxf(acc[ct][i] + bv, 0.f));
  }
}

__global__ __launch_bounds__(256) void upper_front_eval_bf16_kernel(FrontP p) {
  __shared__ __attribute__((aligned(16))) bf16_t shw[FR_WEND];
  __shared__ __attribute__((aligned(16))) float sh[FR_SHARED_END];
  __shared__ __attribute__((aligned(16))) bf16_t act[4 * FR_ACT_WAVE];
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
        w[u0 + u] = p.l[L].W[min(n, Cn[L] - 1) * Kn[L] + min(k, Kn[L] - 1)];
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

  bf16_t* const A = act + wave * FR_ACT_WAVE;
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
    // running softmax state of this wave: max, sum, and this lane's share of the weighted column sums (column 16 ct + fr, rows of
    // lane group fq; the four groups are added at the end)
    float m_run = -INFINITY, s_run = 0.f;
    float col[4] = {0.f, 0.f, 0.f, 0.f};
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
      if (lane < FR_SLAB) {
        const long row = (long)s * FR_SLAB + lane;
        const float d0 = __fsub_rn(v01.x, tt[0]), d1 = __fsub_rn(v01.y, tt[1]), d2 = __fsub_rn(v23.x, tt[2]);
        const float h0 = fr_dot3_nofma(r[0], r[1], r[2], d0, d1, d2);
        const float h1 = fr_dot3_nofma(r[3], r[4], r[5], d0, d1, d2);
        const float h2 = fr_dot3_nofma(r[6], r[7], r[8], d0, d1, d2);
        *reinterpret_cast<float2*>(xf + row * 6) = make_float2(h0, h1);
        *reinterpret_cast<float2*>(xf + row * 6 + 2) = make_float2(h2, v23.y);
        if (p.x_src) *reinterpret_cast<float2*>(xf + row * 6 + 4) = v45;
        bf16_t* a0 = A + FR_ACT_P + lane * FR_S16;
        const uint2 z2 = make_uint2(0u, 0u);
        *reinterpret_cast<uint2*>(a0) = frb_pack4(h0, h1, h2, v23.y);
        *reinterpret_cast<uint2*>(a0 + 4) = frb_pack4(v45.x, v45.y, 0.f, 0.f);
        *reinterpret_cast<uint2*>(a0 + 8) = z2;
        *reinterpret_cast<uint2*>(a0 + 12) = z2;
        bf16_t* fz = A + FR_ACT_FZ + lane * FR_S32;                        // concat: the first four point columns ...
        *reinterpret_cast<uint2*>(fz) = frb_pack4(h0, h1, h2, v23.y);
        *reinterpret_cast<uint2*>(fz + 28) = z2;                           // ... and the k padding behind the 24 features
      }
      __builtin_amdgcn_wave_barrier();
      // (wave-private LDS: a wave's LDS operations complete in issue order, so the tile written by some lanes is what the other
      // lanes read; wave_barrier only keeps the compiler from moving code across the hand-over)
      f32x4 a1[1], a2[1], a3[2], g1[2], g2[3], g3[4];
      fr_stage<1, 1, FR_S16, FR_S16>(A + FR_ACT_P, shw + FR_W1, a1, fr, fq);
      fr_store<1, FR_S16>(a1, sh + FR_B1, A + FR_ACT_Q, 0, fr, fq);
      fr_stage<1, 1, FR_S16, FR_S16>(A + FR_ACT_Q, shw + FR_W2, a2, fr, fq);
      fr_store<1, FR_S16>(a2, sh + FR_B2, A + FR_ACT_P, 0, fr, fq);
      fr_stage<2, 1, FR_S16, FR_S16>(A + FR_ACT_P, shw + FR_W3, a3, fr, fq);
      // PointNet's 24 features land behind the four point columns: columns 4..27 of the concat tile (the padded outputs 24..31
      // of the stage are zero and would land in columns 28..35: only columns < 28 are stored)
      {
        const float b0 = sh[FR_B3 + fr], b1 = sh[FR_B3 + 16 + fr];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          bf16_t* d = A + FR_ACT_FZ + (4 * fq + i) * FR_S32 + 4;
          d[fr] = (bf16_t)frb_bf(fmaxf(a3[0][i] + b0, 0.f));
          if (fr < 8) d[16 + fr] = (bf16_t)frb_bf(fmaxf(a3[1][i] + b1, 0.f));
        }
      }
      fr_stage<2, 2, FR_S32, FR_S32>(A + FR_ACT_FZ, shw + FR_G1, g1, fr, fq);
      fr_store<2, FR_S32>(g1, sh + FR_C1, A + FR_ACT_P, 0, fr, fq);            // G1 tile [16][36] over P | Q
      fr_stage<3, 2, FR_S32, FR_S32>(A + FR_ACT_P, shw + FR_G2, g2, fr, fq);
      fr_store<3, FR_S48>(g2, sh + FR_C2, A + FR_ACT_G2, 0, fr, fq);
      fr_stage<4, 3, FR_S48, FR_S48>(A + FR_ACT_G2, shw + FR_G3, g3, fr, fq);
      // ---- scores and the online softmax update.  Lane (fr, fq), register i: point 4 fq + i of the slab, columns 16 ct + fr.
      float y[4][4], part[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) part[i] = 0.f;
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) {
        const float bv = sh[FR_C3 + ct * 16 + fr], wv = sh[FR_AW + ct * 16 + fr];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          y[ct][i] = fmaxf(g3[ct][i] + bv, 0.f);
          part[i] += y[ct][i] * wv;
        }
      }
#pragma unroll
      for (int o = 1; o < 16; o <<= 1)
#pragma unroll
        for (int i = 0; i < 4; ++i) part[i] += __shfl_xor(part[i], o, 64);      // sum over the 16 lanes of a group: all columns
      float smax = -INFINITY;
#pragma unroll
      for (int i = 0; i < 4; ++i) { part[i] += attn_b; smax = fmaxf(smax, part[i]); }
      if (fr == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) sc[s * FR_SLAB + 4 * fq + i] = part[i];
      }
      smax = fmaxf(smax, __shfl_xor(smax, 16, 64));
      smax = fmaxf(smax, __shfl_xor(smax, 32, 64));                              // max over the slab's 16 points
      const float m_new = fmaxf(m_run, smax);
      const float resc = __expf(m_run - m_new);                                  // (exp(-inf) = 0 on the first slab)
      float e[4], esum = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) { e[i] = __expf(part[i] - m_new); esum += e[i]; }
      esum += __shfl_xor(esum, 16, 64);
      esum += __shfl_xor(esum, 32, 64);
      s_run = s_run * resc + esum;
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) {
        float v = col[ct] * resc;
#pragma unroll
        for (int i = 0; i < 4; ++i) v += e[i] * y[ct][i];
        col[ct] = v;
      }
      m_run = m_new;
    }
    // ---- combine the four waves (fixed order) and emit the frame's outputs
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
      col[ct] += __shfl_xor(col[ct], 16, 64);
      col[ct] += __shfl_xor(col[ct], 32, 64);
    }
    if (lane < 16) {
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) comb[par][wave][2 + ct * 16 + lane] = col[ct];
      if (lane == 0) { comb[par][wave][0] = m_run; comb[par][wave][1] = s_run; }
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
