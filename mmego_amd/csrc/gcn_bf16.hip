// Eval-mode ST-GCN block with bf16 OPERANDS and fp32 accumulation (reference Net/GCN.py:67-147 st_gcn, :55-64
// ConvTemporalGraphical; BASELINE config 5 "bf16 forward / fp32 accumulate").  Opt-in (LowerNet.precision = "bf16", eval
// forwards only): the fp32 kernels (gcn.hip, gcn_fused.hip) stay the default and the parity path.
//
// What is bf16 and what is not: the two operands of every dense product are rounded to bf16 (round to nearest even), products
// are exact in fp32 and accumulate in fp32 on v_mfma_f32_32x32x16_bf16; the graph mixing, biases, BatchNorm affines, residual
// sum and ReLU are fp32.  A block is two launches (the frozen BatchNorms are affines known up front, so nothing needs a
// grid-wide statistic):
//   gcn_mix_eval_bf16   8 frames per tile.  The einsum is moved in FRONT of the 1x1 conv (exact in real arithmetic):
//                         y_f = sum_k A_k^T (X_f W_k^T + 1 b_k^T) = [A_0^T X_f | A_1^T X_f | A_2^T X_f] . [W_0 | W_1 | W_2]^T + (sum_k colsum(A_k) b_k^T)
//                       so the mixing works on cin channels instead of K cout, in fp32 on the VALU straight from the loaded rows,
//                       and the product has K cin + cin as its k axis (the last cin columns are X itself: the residual branch's
//                       1x1 conv, accumulated in a second set of tiles).  Out: relu(bn0(y)) as bf16 (the temporal conv's operand)
//                       and bn_r(residual) as fp32.  z [rows][K cout] never exists.
//   tconv_eval_bf16     one sequence per workgroup: its activated rows sit in LDS once (zero halo frames at both ends, so a
//                       tap is a row offset), the 9 x (cin / 64) weight chunks stream through a two-stage LDS ring in
//                       fragment-major order; epilogue out = relu(bn3(conv) + residual) in fp32.  The input is read once from
//                       HBM instead of once per tap and column tile.
// HBM bytes per row of a block (cin -> cout): 4 cin + 2 cout + 4 cout | 2 cout + 4 cout + 4 cout.
#include <stdlib.h>

#include "common.h"

typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef unsigned short bf16_t;   // raw bf16 bits (the C ABI carries them as unsigned short)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// (a float -> __bf16 conversion is v_cvt_pk_bf16_f32 on gfx950: round to nearest even, NaN stays NaN -- one instruction for two
// values where the integer form took five per value)
typedef float lo_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 lo_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned int f2bf(float x) { return (unsigned int)__builtin_bit_cast(unsigned short, (__bf16)x); }
__device__ __forceinline__ unsigned int f2bf2(float lo, float hi) {
  return __builtin_bit_cast(unsigned int, __builtin_convertvector((lo_f32x2){lo, hi}, lo_bf16x2));
}

// ---------------------------------------------------------------------------------------------------------------------------
struct TconvBfP {
  const bf16_t* X;                   // activated input rows (b, t, v) x CIN, dense
  const bf16_t* W;                   // [tap][k chunk][column tile][16-k step][lane][8]: a lane's MFMA B operand is 16 contiguous bytes
  const float* bias;                 // [COUT] or null
  const float* post;                 // [4][COUT] mean, invstd, a, b of the BatchNorm behind, or null
  const float* res; long ldr;        // added behind the affine, in front of the ReLU (the normalised residual branch), or null
  float* Y; long ldy;
  int relu, B, T, V, taps;
};

template <int CIN, int COUT>
__global__ __launch_bounds__(512) void tconv_eval_bf16_kernel(TconvBfP p) {
  constexpr int KC = CIN < 64 ? CIN : 64;              // k per weight chunk
  constexpr int NKC = CIN / KC, NKS = KC / 16;
  constexpr int NT_C = COUT >= 64 ? 2 : 1, NT_R = COUT == 128 ? 2 : 1;
  constexpr int WC = COUT / (32 * NT_C), WR = 8 / WC;  // 8 waves: WR row groups x WC column groups, 256 rows per pass
  constexpr int XS = CIN + 8;                          // bf16 per LDS row: 16 bytes of padding (consecutive rows on distinct bank quads)
  constexpr int WCH = COUT * KC;                       // bf16 per weight chunk
  constexpr int NWL = (WCH / 8 + 511) / 512;           // 16-byte pieces of a chunk per thread
  extern __shared__ __attribute__((aligned(16))) unsigned char smraw[];
  bf16_t* wbuf = reinterpret_cast<bf16_t*>(smraw);     // [2][WCH]
  bf16_t* xs = wbuf + 2 * WCH;                         // [XR][XS]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int V = p.V, TV = p.T * V, half = p.taps / 2, halo = half * V;
  const int nrt = (TV + 255) / 256, XR = nrt * 256 + 2 * halo;
  const int nch = p.taps * NKC;
  const long b = blockIdx.x;

  // weight chunks 0 .. 3: requested first (chunk c waits in register set c % 3 until it is written into the LDS ring one step
  // ahead of its use: three steps of MFMAs cover a chunk's trip from L2)
  u32x4 wq[3][NWL];
#define TB_LOADW(set, chunk)                                                                                       \
  do {                                                                                                             \
    _Pragma("unroll") for (int u = 0; u < NWL; ++u) {                                                              \
      const int i_ = tid + 512 * u;                                                                                \
      wq[set][u] = *reinterpret_cast<const u32x4*>(p.W + (long)(chunk) * WCH + (long)(i_ < WCH / 8 ? i_ : 0) * 8); \
    }                                                                                                              \
  } while (0)
#define TB_STOREW(set, stage)                                                                                      \
  do {                                                                                                             \
    _Pragma("unroll") for (int u = 0; u < NWL; ++u) {                                                              \
      const int i_ = tid + 512 * u;                                                                                \
      if (i_ < WCH / 8) *reinterpret_cast<u32x4*>(wbuf + (stage) * WCH + i_ * 8) = wq[set][u];                     \
    }                                                                                                              \
  } while (0)
  TB_LOADW(0, 0);
  // the sequence's rows -> xs rows [halo, halo + TV); everything else zero (halo frames, rows past the sequence, the row padding)
  {
    const int c8n = CIN / 8, npc = TV * c8n;           // 16-byte pieces of the sequence
    const bf16_t* xb = p.X + b * TV * CIN;
    const u32x4 z = {0u, 0u, 0u, 0u};
    for (int i = tid; i < halo * (XS / 8); i += 512) {
      *reinterpret_cast<u32x4*>(xs + i * 8) = z;
      *reinterpret_cast<u32x4*>(xs + (long)(halo + TV) * XS + i * 8) = z;
    }
    for (int i = tid; i < (XR - 2 * halo - TV) * (XS / 8); i += 512) *reinterpret_cast<u32x4*>(xs + (long)(2 * halo + TV) * XS + i * 8) = z;
    for (int i0 = 0; i0 < npc; i0 += 512 * 8) {
      u32x4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = i0 + tid + 512 * u;
        v[u] = *reinterpret_cast<const u32x4*>(xb + (long)(i < npc ? i : npc - 1) * 8);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = i0 + tid + 512 * u;
        if (i < npc) {
          const int row = i / c8n, c = (i - row * c8n) * 8;
          *reinterpret_cast<u32x4*>(xs + (long)(halo + row) * XS + c) = v[u];
        }
      }
    }
    for (int i = tid; i < TV; i += 512) *reinterpret_cast<u32x4*>(xs + (long)(halo + i) * XS + CIN) = z;      // (row padding: never read, kept clean)
  }
  TB_STOREW(0, 0);
  const int nsteps = nrt * nch;                        // (nch % 3 == 0: the register set of a chunk is static in the loop below)
  TB_LOADW(1, 1 % nch);
  TB_LOADW(2, 2 % nch);
  TB_LOADW(0, 3 % nch);
  __syncthreads();

  const int wrow = wave % WR, wcol = wave / WR;
  const int r = lane & 31, h = lane >> 5;
  int cur = 0;                                         // ring stage holding the chunk about to be multiplied
  for (int rt = 0; rt < nrt; ++rt) {
    const int rb = rt * 256 + wrow * NT_R * 32;        // this wave's first row of the sequence
    // the residual term of this pass's outputs: requested now, consumed behind the product loop
    float rs[NT_R][NT_C][16];
#pragma unroll
    for (int i = 0; i < NT_R; ++i)
#pragma unroll
      for (int j = 0; j < NT_C; ++j) {
        const int col = (wcol * NT_C + j) * 32 + r;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int row = rb + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
          const long g = b * TV + (row < TV ? row : TV - 1);
          rs[i][j][e] = p.res ? p.res[g * p.ldr + col] : 0.f;
        }
      }
    f32x16 acc[NT_R][NT_C];
#pragma unroll
    for (int i = 0; i < NT_R; ++i)
#pragma unroll
      for (int j = 0; j < NT_C; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    for (int ch0 = 0; ch0 < nch; ch0 += 3) {
#pragma unroll
      for (int u3 = 0; u3 < 3; ++u3) {
        const int ch = ch0 + u3;
        const int step = rt * nch + ch;
        const int tap = ch / NKC, kc = ch - tap * NKC;
        const bf16_t* wb = wbuf + cur * WCH;
        const bf16_t* xa = xs + (long)(rb + r + tap * V) * XS + kc * KC + 8 * h;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
          bf16x8 a[NT_R], bb[NT_C];
#pragma unroll
          for (int i = 0; i < NT_R; ++i) a[i] = *reinterpret_cast<const bf16x8*>(xa + (long)i * 32 * XS + ks * 16);
#pragma unroll
          for (int j = 0; j < NT_C; ++j) bb[j] = *reinterpret_cast<const bf16x8*>(wb + (((wcol * NT_C + j) * NKS + ks) * 64 + lane) * 8);
#pragma unroll
          for (int i = 0; i < NT_R; ++i)
#pragma unroll
            for (int j = 0; j < NT_C; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], bb[j], acc[i][j], 0, 0, 0);
        }
        // chunk step + 1 (register set (u3 + 1) % 3, requested three steps ago) -> the other ring stage; chunk step + 4 takes its set
        if (step + 1 < nsteps) TB_STOREW((u3 + 1) % 3, cur ^ 1);
        TB_LOADW((u3 + 1) % 3, (step + 4) % nch);
        cur ^= 1;
        __syncthreads();
      }
    }
    // epilogue: bias -> BatchNorm affine -> + residual -> ReLU, fp32, straight from the accumulators (a store covers two rows x 128 B)
#pragma unroll
    for (int j = 0; j < NT_C; ++j) {
      const int col = (wcol * NT_C + j) * 32 + r;
      const float bs = p.bias ? p.bias[col] : 0.f;
      float mu = 0.f, sa = 1.f, sb = 0.f;
      if (p.post) { mu = p.post[col]; sa = p.post[2 * COUT + col]; sb = p.post[3 * COUT + col]; }
#pragma unroll
      for (int i = 0; i < NT_R; ++i) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int row = rb + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
          float v = acc[i][j][e] + bs;
          if (p.post) v = __builtin_fmaf(v - mu, sa, sb);
          v += rs[i][j][e];
          if (p.relu) v = fmaxf(v, 0.f);
          if (row < TV) p.Y[(b * TV + row) * p.ldy + col] = v;
        }
      }
    }
  }
#undef TB_LOADW
#undef TB_STOREW
}

template <int CIN, int COUT>
static int tconv_bf16_launch(hipStream_t st, const TconvBfP& p) {
  constexpr int KC = CIN < 64 ? CIN : 64;
  const int TV = p.T * p.V, halo = (p.taps / 2) * p.V;
  const long XR = (long)((TV + 255) / 256) * 256 + 2 * halo;
  const long lds = 2L * COUT * KC * 2 + XR * (CIN + 8) * 2;
  if (lds > 160 * 1024) return -2;
  static long attr = 0;
  if (lds > attr) {
    hipError_t e = hipFuncSetAttribute((const void*)tconv_eval_bf16_kernel<CIN, COUT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    attr = lds;
  }
  hipLaunchKernelGGL((tconv_eval_bf16_kernel<CIN, COUT>), dim3(p.B), dim3(512), lds, st, p);
  MMEGO_LAUNCH_CHECK();
  return MMEGO_OK;
}

// 1 when mmego_tconv_eval_bf16 takes the shape (the caller keeps the fp32 kernel otherwise)
extern "C" int mmego_tconv_eval_bf16_ok(int T, int V, int Cin, int Cout, int taps) {
  if (!(Cin == Cout && (Cin == 32 || Cin == 64 || Cin == 128)) || taps < 3 || (taps & 1) == 0 || taps % 3 != 0 || T < 1 || V < 1) return 0;
  const int KC = Cin < 64 ? Cin : 64;
  const long TV = (long)T * V, XR = (TV + 255) / 256 * 256 + 2L * (taps / 2) * V;
  return 2L * Cout * KC * 2 + XR * (Cin + 8) * 2 <= 160 * 1024;
}

// Y[(b,t,v)][co] = relu?( bn?(bias[co] + sum_tap sum_ci X[(b, t + tap - taps/2, v)][ci] W[co][ci][tap]) + res ), X in bf16 (already
// activated), W packed by mmego_tconv_pack_bf16, accumulation and epilogue in fp32
extern "C" int mmego_tconv_eval_bf16(void* stream, const unsigned short* X, const unsigned short* Wp, const float* bias, const float* post,
                                     const float* res, long ldr, float* Y, long ldy, int relu, int B, int T, int V, int Cin, int Cout,
                                     int taps) {
  MMEGO_REQUIRE(B >= 0 && mmego_tconv_eval_bf16_ok(T, V, Cin, Cout, taps) && ldy >= Cout && (!res || ldr >= Cout));
  MMEGO_REQUIRE((((uintptr_t)X) & 15) == 0 && (((uintptr_t)Wp) & 15) == 0);
  if (B == 0) return MMEGO_OK;
  TconvBfP p;
  p.X = X; p.W = Wp; p.bias = bias; p.post = post; p.res = res; p.ldr = ldr; p.Y = Y; p.ldy = ldy;
  p.relu = relu; p.B = B; p.T = T; p.V = V; p.taps = taps;
  hipStream_t st = (hipStream_t)stream;
  if (Cin == 32) return tconv_bf16_launch<32, 32>(st, p);
  if (Cin == 64) return tconv_bf16_launch<64, 64>(st, p);
  return tconv_bf16_launch<128, 128>(st, p);
}

// W [Cout][Cin][taps] fp32 -> bf16 chunks [tap][k chunk][column tile][16-k step][lane][8] (lane (n, hf) of step ks holds
// W[32 ct + n][KC kc + 16 ks + 8 hf .. + 8][tap])
__global__ __launch_bounds__(256) void tconv_pack_bf16_kernel(const float* __restrict__ W, int Co, int Ci, int taps, bf16_t* __restrict__ Wp) {
  const int KC = Ci < 64 ? Ci : 64, NKC = Ci / KC, NKS = KC / 16, NCT = Co / 32;
  const long total = (long)taps * Co * Ci;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    long q = i;
    const int j = (int)(q & 7); q >>= 3;
    const int lane = (int)(q & 63); q >>= 6;
    const int ks = (int)(q % NKS); q /= NKS;
    const int ct = (int)(q % NCT); q /= NCT;
    const int kc = (int)(q % NKC);
    const int tap = (int)(q / NKC);
    const int co = ct * 32 + (lane & 31), ci = kc * KC + ks * 16 + 8 * (lane >> 5) + j;
    Wp[i] = (bf16_t)f2bf(W[((long)co * Ci + ci) * taps + tap]);
  }
}

extern "C" int mmego_tconv_pack_bf16(void* stream, const float* W, int Cout, int Cin, int taps, unsigned short* Wp) {
  MMEGO_REQUIRE(Cout % 32 == 0 && (Cin == 32 || Cin % 64 == 0) && taps >= 1);
  const long total = (long)taps * Cout * Cin;
  const int grid = (int)(total / 256 + 1 < 2048 ? total / 256 + 1 : 2048);
  hipLaunchKernelGGL(tconv_pack_bf16_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, W, Cout, Cin, taps, Wp);
  MMEGO_LAUNCH_CHECK();
  return MMEGO_OK;
}

// ---------------------------------------------------------------------------------------------------------------------------
#define GM_NT 256          // threads of gcn_mix_eval_bf16
#define GM_FPB 8           // frames per tile: 128 tile rows (16 per frame; rows v >= V are padding)

struct GcnMixBfP {
  const float* X;                    // rows (f, v) x CIN fp32, dense
  const float* in_state;             // [4][V CIN] mean, invstd, a, b of data_bn (applied while loading) or null
  const float* A; const float* imp;  // [K][V][V] each
  const bf16_t* Wy; const bf16_t* Wr; // fragment-major: Wy [column tile][KSY][lane][8] over k in [0, 16 KSY), Wr [column tile][KSR][lane][8] over k in [16 KR0, KD)
  const float* biasy;                // [V][COUT]: sum_k colsum_v(A.imp)[k][w] b_k[c]
  const float* biasr;                // [COUT]
  const float* st0; const float* str_; // [4][COUT] mean, invstd, a, b of the BatchNorm behind the einsum / of the residual branch
  bf16_t* Yact; float* Rn;           // [rows][COUT]
  long F; int V;
};

template <int CIN, int COUT, int K>
__global__ __launch_bounds__(GM_NT, CIN >= 64 ? 1 : 2) void gcn_mix_eval_bf16_kernel(GcnMixBfP p) {
  constexpr int KD = ((K + 1) * CIN + 15) / 16 * 16;   // k axis: [A_0^T X | .. | A_{K-1}^T X | X | 0]
  constexpr int NKS = KD / 16, KSY = (K * CIN + 15) / 16, KR0 = (K * CIN) / 16, KSR = NKS - KR0;
  constexpr int XS = KD + 8;
  constexpr int NCT = COUT / 32;
  constexpr int CG = (CIN + 3) / 4;                    // 4-channel groups
  constexpr int WPT = CIN >= 64 ? 8 : (CIN >= 32 ? 4 : 1), NWG = 16 / WPT;      // output joints per mixing task
  constexpr int NTASK = GM_FPB * CG * NWG;             // mixing tasks per tile: (frame, channel group, group of WPT joints)
  constexpr int YS = COUT + 8;
  static_assert(NTASK <= GM_NT, "tile too wide");
  extern __shared__ __attribute__((aligned(16))) unsigned char smraw[];
  float* ai = reinterpret_cast<float*>(smraw);         // [K][16][16] (A . importance)[k][v][w], zero padded
  bf16_t* wy = reinterpret_cast<bf16_t*>(ai + K * 256);
  bf16_t* wr_ = wy + COUT * KSY * 16;
  bf16_t* xk = wr_ + COUT * KSR * 16;                  // [128][XS]; later the bf16 output tile [128][YS]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int V = p.V;
  // ---- once per workgroup: the mixing table, the weights, a clean tile
  for (int i = tid; i < K * 256; i += GM_NT) {
    const int k = i >> 8, v = (i >> 4) & 15, w = i & 15;
    const bool ok = v < V && w < V;
    const int idx = ok ? (k * V + v) * V + w : 0;
    const float a = p.A[idx], im = p.imp[idx];
    ai[i] = ok ? a * im : 0.f;
  }
  for (int i = tid; i < COUT * KSY * 2; i += GM_NT) *reinterpret_cast<u32x4*>(wy + i * 8) = *reinterpret_cast<const u32x4*>(p.Wy + (long)i * 8);
  for (int i = tid; i < COUT * KSR * 2; i += GM_NT) *reinterpret_cast<u32x4*>(wr_ + i * 8) = *reinterpret_cast<const u32x4*>(p.Wr + (long)i * 8);
  {
    const u32x4 z = {0u, 0u, 0u, 0u};
    constexpr int NX = 128 * XS > 128 * YS ? 128 * XS : 128 * YS;
    for (int i = tid; i < NX / 8; i += GM_NT) *reinterpret_cast<u32x4*>(xk + i * 8) = z;
  }
  __syncthreads();

  // this thread's mixing task
  const int tfl = tid / (CG * NWG), tcg = (tid / NWG) % CG, twg = tid % NWG;
  const bool task = tid < NTASK;
  const int c0 = tcg * 4;
  constexpr int NC = CIN % 4 == 0 ? 4 : CIN % 4;       // channels of the (only ragged) last group; CIN = 3: one group of 3
  const int r = lane & 31, h = lane >> 5;
  const long ntile = (p.F + GM_FPB - 1) / GM_FPB;
  for (long tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
    const long f0 = tile * GM_FPB;
    // (pointers laundered per tile: otherwise every tile-invariant read -- 256 registers of weight fragments, the bias table, the
    // BatchNorm states -- is hoisted out of this loop and the kernel spills)
    const bf16_t *wy_ = wy, *wrr = wr_;
    const float *by = p.biasy, *s0 = p.st0, *sr = p.str_, *brp = p.biasr;
    asm volatile("" : "+v"(wy_), "+v"(wrr));
    asm volatile("" : "+s"(by), "+s"(s0), "+s"(sr), "+s"(brp));
    // ---- load this task's column of the frame: x[v][0..3], v < V (clamped frame: a ragged last tile computes on a valid frame and stores nothing)
    float x[16][4];
    if (task) {
      const long f = f0 + tfl < p.F ? f0 + tfl : p.F - 1;
      const float* xf = p.X + f * V * CIN;
#pragma unroll
      for (int v = 0; v < 16; ++v) {
        const int vv = v < V ? v : V - 1;
        if (CIN % 4 == 0) {
          const f32x4 t = *reinterpret_cast<const f32x4*>(xf + vv * CIN + c0);
          x[v][0] = t[0]; x[v][1] = t[1]; x[v][2] = t[2]; x[v][3] = t[3];
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) x[v][e] = e < NC ? xf[vv * CIN + c0 + e] : 0.f;
        }
      }
      if (CIN % 4 != 0 && p.in_state) {                 // (the first block's variant only: the launcher refuses it elsewhere)
        const int C = V * CIN;
#pragma unroll
        for (int v = 0; v < 16; ++v) {
          const int vv = v < V ? v : V - 1;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int ch = vv * CIN + c0 + (e < NC ? e : 0);
            x[v][e] = e < NC ? __builtin_fmaf(x[v][e] - p.in_state[ch], p.in_state[2 * C + ch], p.in_state[3 * C + ch]) : 0.f;
          }
        }
      }
#pragma unroll
      for (int v = 0; v < 16; ++v)
        if (v >= V) { x[v][0] = 0.f; x[v][1] = 0.f; x[v][2] = 0.f; x[v][3] = 0.f; }
      // ---- the K mixed copies and X itself, rounded to bf16, into the tile: rows w of this half
#pragma unroll 1
      for (int k = 0; k < K; ++k) {                     // (not unrolled, the table reads of four joints at a time: bounded live ranges)
        float s[WPT][4];
#pragma unroll
        for (int w = 0; w < WPT; ++w) { s[w][0] = 0.f; s[w][1] = 0.f; s[w][2] = 0.f; s[w][3] = 0.f; }
        const float* ak = ai + k * 256 + twg * WPT;
#pragma unroll
        for (int vq = 0; vq < 4; ++vq) {
          float a[4][WPT];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            if (WPT >= 4) {
#pragma unroll
              for (int w4 = 0; w4 < WPT / 4; ++w4) {
                const f32x4 t = *reinterpret_cast<const f32x4*>(ak + (4 * vq + u) * 16 + 4 * w4);
                a[u][4 * w4 + 0] = t[0]; a[u][4 * w4 + 1] = t[1]; a[u][4 * w4 + 2] = t[2]; a[u][4 * w4 + 3] = t[3];
              }
            } else {
#pragma unroll
              for (int w = 0; w < WPT; ++w) a[u][w] = ak[(4 * vq + u) * 16 + w];
            }
          }
#pragma unroll
          for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int w = 0; w < WPT; ++w)
#pragma unroll
              for (int e = 0; e < 4; ++e) s[w][e] = __builtin_fmaf(a[u][w], x[4 * vq + u][e], s[w][e]);
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int w = 0; w < WPT; ++w) {
          bf16_t* d = xk + (tfl * 16 + twg * WPT + w) * XS + k * CIN + c0;
          if (NC == 4) {
            uint2 o;
            o.x = f2bf2(s[w][0], s[w][1]);
            o.y = f2bf2(s[w][2], s[w][3]);
            *reinterpret_cast<uint2*>(d) = o;
          } else {
#pragma unroll
            for (int e = 0; e < NC; ++e) d[e] = (bf16_t)f2bf(s[w][e]);
          }
        }
      }
#pragma unroll
      for (int w = 0; w < WPT; ++w) {
        bf16_t* d = xk + (tfl * 16 + twg * WPT + w) * XS + K * CIN + c0;
        // x[twg * WPT + w] by bit masks (a select of array elements is turned into a dynamically indexed load and the array into
        // scratch memory)
        unsigned xb[4] = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int q = 0; q < NWG; ++q) {
          const unsigned m = 0u - (unsigned)(twg == q);
#pragma unroll
          for (int e = 0; e < 4; ++e) xb[e] |= __float_as_uint(x[q * WPT + w][e]) & m;
        }
        if (NC == 4) {
          uint2 o;
          o.x = f2bf2(__uint_as_float(xb[0]), __uint_as_float(xb[1]));
          o.y = f2bf2(__uint_as_float(xb[2]), __uint_as_float(xb[3]));
          *reinterpret_cast<uint2*>(d) = o;
        } else {
#pragma unroll
          for (int e = 0; e < NC; ++e) d[e] = (bf16_t)f2bf(__uint_as_float(xb[e]));
        }
      }
    }
    __syncthreads();
    // ---- the products: wave w takes tile rows [32 w, 32 w + 32) (two frames), every column tile
    f32x16 accy[NCT], accr[NCT];
#pragma unroll
    for (int j = 0; j < NCT; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) { accy[j][e] = 0.f; accr[j][e] = 0.f; }
    const bf16_t* xa = xk + (wave * 32 + r) * XS + 8 * h;
    // (operands of step s + 1 requested in front of the MFMAs of step s, two register sets: without the fences the compiler hoists
    // every LDS read of the loop and spills)
    {
      bf16x8 a[2], bb[2][NCT];
#define GM_RD(set, W_, NS_, ks_, k0_)                                                                               \
  do {                                                                                                              \
    a[set] = *reinterpret_cast<const bf16x8*>(xa + ((k0_) + (ks_)) * 16);                                           \
    _Pragma("unroll") for (int j = 0; j < NCT; ++j)                                                                 \
      bb[set][j] = *reinterpret_cast<const bf16x8*>(W_ + ((j * (NS_) + (ks_)) * 64 + lane) * 8);                    \
  } while (0)
      GM_RD(0, wy_, KSY, 0, 0);
#pragma unroll
      for (int ks = 0; ks < KSY; ++ks) {
        if (ks + 1 < KSY) GM_RD((ks + 1) & 1, wy_, KSY, ks + 1, 0);
        else GM_RD((ks + 1) & 1, wrr, KSR, 0, KR0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < NCT; ++j) accy[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks & 1], bb[ks & 1][j], accy[j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int ks = 0; ks < KSR; ++ks) {
        if (ks + 1 < KSR) GM_RD((KSY + ks + 1) & 1, wrr, KSR, ks + 1, KR0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < NCT; ++j)
          accr[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(KSY + ks) & 1], bb[(KSY + ks) & 1][j], accr[j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
#undef GM_RD
    }
    __syncthreads();                                    // (the tile has been read: it becomes the output stage)
    // ---- epilogue.  Residual branch: BatchNorm affine, fp32, straight from the accumulators; einsum branch: BatchNorm + ReLU ->
    // bf16 through LDS (a frame's V rows are contiguous in memory: 16-byte pieces)
    bf16_t* ys = xk;
    float* rnb = p.Rn + f0 * V * COUT;                  // (32-bit offsets from the tile's first row below)
    const int nfr = (int)(p.F - f0 < GM_FPB ? p.F - f0 : GM_FPB);
#pragma unroll
    for (int j = 0; j < NCT; ++j) {
      const int col = j * 32 + r;
      const float m0 = s0[col], a0 = s0[2 * COUT + col], b0 = s0[3 * COUT + col];
      const float mr = sr[col], ar = sr[2 * COUT + col], br = sr[3 * COUT + col];
      const float bsr = brp[col];
#pragma unroll
      for (int e4 = 0; e4 < 4; ++e4) {
        float byv[4];
#pragma unroll
        for (int e1 = 0; e1 < 4; ++e1) {
          const int w = (8 * e4 + 4 * h + e1) & 15;                    // row = 32 wave + e1 + 8 e4 + 4 h: joint w of frame (row >> 4)
          byv[e1] = by[(w < V ? w : V - 1) * COUT + col];
        }
#pragma unroll
        for (int e1 = 0; e1 < 4; ++e1) {
          const int e = 4 * e4 + e1;
          const int row = wave * 32 + e1 + 8 * e4 + 4 * h;
          const int fl = row >> 4, w = row & 15;
          const float y = accy[j][e] + byv[e1];
          ys[row * YS + col] = (bf16_t)f2bf(fmaxf(__builtin_fmaf(y - m0, a0, b0), 0.f));
          const float rn = __builtin_fmaf(accr[j][e] + bsr - mr, ar, br);
          if (w < V && fl < nfr) rnb[(unsigned)((fl * V + w) * COUT + col)] = rn;
        }
      }
    }
    __syncthreads();
    {
      constexpr int C8 = COUT / 8;
      const int npc = nfr * V * C8;
      for (int i = tid; i < npc; i += GM_NT) {
        const int rowg = i / C8, c = (i - rowg * C8) * 8;
        const int fl = rowg / V, w = rowg - fl * V;
        *reinterpret_cast<u32x4*>(p.Yact + (f0 * V + rowg) * COUT + c) = *reinterpret_cast<const u32x4*>(ys + (fl * 16 + w) * YS + c);
      }
    }
    __syncthreads();
    if ((K + 1) * CIN != KD) {                          // the padding columns of the k axis must be clean again (0 x finite)
      const u32x4 z = {0u, 0u, 0u, 0u};
      constexpr int NX = 128 * XS > 128 * YS ? 128 * XS : 128 * YS;
      for (int i = tid; i < NX / 8; i += GM_NT) *reinterpret_cast<u32x4*>(xk + i * 8) = z;
      __syncthreads();
    }
  }
}

template <int CIN, int COUT, int K>
static int gcn_mix_bf16_launch(hipStream_t st, const GcnMixBfP& p) {
  constexpr int KD = ((K + 1) * CIN + 15) / 16 * 16, NKS = KD / 16, KSY = (K * CIN + 15) / 16, KR0 = (K * CIN) / 16, KSR = NKS - KR0;
  constexpr int XS = KD + 8, YS = COUT + 8;
  constexpr long lds = K * 256 * 4 + (long)COUT * (KSY + KSR) * 16 * 2 + 128L * (XS > YS ? XS : YS) * 2;
  static_assert(lds <= 160 * 1024, "LDS");
  static bool attr = false;
  if (!attr) {
    hipError_t e = hipFuncSetAttribute((const void*)gcn_mix_eval_bf16_kernel<CIN, COUT, K>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    attr = true;
  }
  const long ntile = (p.F + GM_FPB - 1) / GM_FPB;
  const int per_cu = (int)(160 * 1024 / lds) < 4 ? (int)(160 * 1024 / lds) : 4;
  const long cap = 256L * (per_cu < 1 ? 1 : per_cu);
  hipLaunchKernelGGL((gcn_mix_eval_bf16_kernel<CIN, COUT, K>), dim3((unsigned)(ntile < cap ? ntile : cap)), dim3(GM_NT), lds, st, p);
  MMEGO_LAUNCH_CHECK();
  return MMEGO_OK;
}

// 1 when mmego_gcn_mix_eval_bf16 takes the shape (the reference's ladder 3 -> 32 -> 64 -> 128 with K <= 3 partitions, V <= 16 joints)
extern "C" int mmego_gcn_mix_eval_bf16_ok(int V, int Cin, int Cout, int K) {
  return V >= 1 && V <= 16 && K >= 1 && K <= 3 && ((Cin == 3 && Cout == 32) || (Cin == 32 && Cout == 64) || (Cin == 64 && Cout == 128));
}

extern "C" int mmego_gcn_mix_eval_bf16(void* stream, const float* X, const float* in_state, const float* A, const float* importance,
                                       const unsigned short* Wy, const unsigned short* Wr, const float* biasy, const float* biasr,
                                       const float* st0, const float* st_r, unsigned short* Yact, float* Rn, long F, int V, int Cin,
                                       int Cout, int K) {
  MMEGO_REQUIRE(F >= 0 && mmego_gcn_mix_eval_bf16_ok(V, Cin, Cout, K) && (in_state == nullptr || Cin % 4 != 0));
  MMEGO_REQUIRE((((uintptr_t)X) & 15) == 0 && (((uintptr_t)Wy) & 15) == 0 && (((uintptr_t)Wr) & 15) == 0 && (((uintptr_t)Yact) & 15) == 0);
  if (F == 0) return MMEGO_OK;
  GcnMixBfP p;
  p.X = X; p.in_state = in_state; p.A = A; p.imp = importance; p.Wy = Wy; p.Wr = Wr; p.biasy = biasy; p.biasr = biasr;
  p.st0 = st0; p.str_ = st_r; p.Yact = Yact; p.Rn = Rn; p.F = F; p.V = V;
  hipStream_t st = (hipStream_t)stream;
#define GM_DISPATCH(K_)                                                  \
  do {                                                                   \
    if (Cin == 3) return gcn_mix_bf16_launch<3, 32, K_>(st, p);          \
    if (Cin == 32) return gcn_mix_bf16_launch<32, 64, K_>(st, p);        \
    return gcn_mix_bf16_launch<64, 128, K_>(st, p);                      \
  } while (0)
  if (K == 1) GM_DISPATCH(1);
  if (K == 2) GM_DISPATCH(2);
  GM_DISPATCH(3);
#undef GM_DISPATCH
}
