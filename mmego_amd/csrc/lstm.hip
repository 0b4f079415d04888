// LSTM recurrences on the matrix cores (fp32, v_mfma_f32_16x16x4_f32), PyTorch gate order i,f,g,o.
//
//  (lstm_step kernels live in lstm_step.hip)
//  lstm_step_kernel   one timestep of a (bi)LSTM of any hidden size H (H % 32 == 0): gates = xproj_t +
//                     h_{t-1}.W_hh^T, fused cell update.  Used for IMU_Net's 2x(2-layer, H=512) BiLSTMs
//                     (reference Net/IMU_Net.py:58-62,77,82), 94 % of the path's FLOPs.  Both directions
//                     in one launch; a workgroup owns 64 batch rows x 32 hidden units x 4 gates so the
//                     cell update is register-local; workgroups that share a W_hh slice are placed on
//                     one XCD (its L2 then holds 1/8 of W_hh).
//  lstm64_fwd/bwd     whole-sequence persistent kernels for the three H=64 BiLSTMs of Upper_Net /
//                     Lower_Net (Upper_Net.py:333, Lower_Net.py:91): batch rows are independent through
//                     the recurrence, so a workgroup keeps W_hh (64 KB) in LDS, 16 rows of h in LDS and c
//                     in registers and walks all T steps with no inter-workgroup traffic.
#include "common.h"
#include <stddef.h>
#include "../../include/mmego_hip.h"       // MmegoLstm64Fwd / MmegoLstm64Bwd (the _multi entry points)

// ---------------------------------------------------------------------------------------------
// H = 64 persistent sequence kernels
// ---------------------------------------------------------------------------------------------
struct Lstm64P {
  const float* xproj[2]; long xs;
  const float* whh[2];
  const float* bhh[2];
  const float* h0[2]; const float* c0[2];
  float* out; long os;
  float* hn[2]; float* cn[2];
  float* gates[2]; float* cst[2]; float* hprev[2];
  int B, T;
  // inverted inter-layer dropout fused into the output store (nn.LSTM(dropout=p) between stacked layers): drop_y / drop_mask
  // have out's layout; the mask of element i is a hash of (i, seed_ctr[0], salt)
  float* drop_y; float* drop_mask; float drop_p; const unsigned long long* seed_ctr; unsigned salt;
};

// rcp / v_exp_f32-based activations: a few ulp from the libm forms at a fraction of their instruction count (the step loop
// of this kernel issued ~490 VALU instructions per 64 MFMAs, most of them libm expf and 64-bit address arithmetic)
__device__ __forceinline__ float l64_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float l64_tanh(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * x)); }

// Workgroup barrier that orders LDS traffic only: __syncthreads() also waits for every outstanding global load / store
// (vmcnt(0)), which would put the prefetch of the next step's inputs and this step's stash stores back on the critical path.
#define L64_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

#define WLD 272  // 256 gate columns + 16: consecutive k rows land on disjoint bank halves

// Per-row pointers kept in arrays lose the "global memory" inference the compiler makes for plain kernel arguments and turn into
// FLAT loads / stores -- which count in lgkmcnt as well as vmcnt, so that every wait for an LDS read also waited for all global
// loads and stores in flight.  Explicit global address space keeps them global_load / global_store (vmcnt only).
typedef const float __attribute__((address_space(1)))* l64_gcptr;
typedef float __attribute__((address_space(1)))* l64_gptr;
typedef float l64_f4 __attribute__((ext_vector_type(4)));
typedef l64_f4 __attribute__((address_space(1)))* l64_gptr4;
typedef const l64_f4 __attribute__((address_space(1)))* l64_gcptr4;

// FULL: B is a multiple of the 16 rows of a workgroup, so every `live` test is a compile-time true; STASH / DROP: the backward
// stashes (gates, cell states, h_{t-1}: all or none) / the fused inter-layer dropout.  Template parameters so that the step loop is
// straight-line code.  That matters more than it looks: with loads and stores predicated row by row (an s_cbranch_execz around
// each) and run-time branches on the optional outputs, the compiler could not count the outstanding memory operations across
// the loop and put ONE s_waitcnt vmcnt(0) into every step -- which waits for the previous step's ~28 stores per lane to be
// acknowledged, not only for the prefetched inputs (2.26 us per step; 1.96 us now).
template <bool FULL, bool STASH, bool DROP>
__device__ __forceinline__ void lstm64_fwd_body(const Lstm64P& p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* hs = smem;              // [64 k][16 rows]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int d = blockIdx.y, r0 = blockIdx.x * 16;
  const int B = p.B, T = p.T;
  const float* W = p.whh[d];
  const int fr = lane & 15, fq = lane >> 4;
  const int j = wave * 16 + fr;  // hidden unit of this lane
  float creg[4], hreg[4], bh[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) bh[g] = p.bhh[d] ? p.bhh[d][g * 64 + j] : 0.f;
#pragma unroll
  for (int reg = 0; reg < 4; ++reg) {
    int row = r0 + fq * 4 + reg;
    bool ok = row < B;
    creg[reg] = (ok && p.c0[d]) ? p.c0[d][row * 64 + j] : 0.f;
    hreg[reg] = (ok && p.h0[d]) ? p.h0[d][row * 64 + j] : 0.f;
    hs[j * 16 + fq * 4 + reg] = hreg[reg];
  }
  __syncthreads();
  // This lane's W_hh operand fragments stay in registers for the whole sequence, loaded straight from global memory: at
  // MFMA step s lane quad fq takes k = 16 fq + s (for h and W_hh alike: any k order is valid as long as both operands agree),
  // so the 16 values of a gate are 4 contiguous 16-B loads.  No W_hh tile in LDS, no staging pass.
  float wreg[16][4];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const float* wrow = W + (g * 64 + j) * 64 + 16 * fq;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(wrow + 4 * q);
      wreg[4 * q + 0][g] = v.x; wreg[4 * q + 1][g] = v.y; wreg[4 * q + 2][g] = v.z; wreg[4 * q + 3][g] = v.w;
    }
  }

  // The input projections do not depend on the recurrence: step s+1's values are fetched while step s computes, so
  // their latency (the longest thing in a step otherwise) is off the critical path.  All addresses advance by a
  // per-step stride from bases computed once (the address arithmetic used to outweigh the MFMAs).
  float xp[4][4], xpn[4][4];
  constexpr bool stash = STASH, keep_h = STASH, drop = DROP;
  const unsigned dkey = drop ? dropout_key(p.seed_ctr[0], p.salt) : 0u;
  const float keep_scale = drop ? 1.0f / (1.0f - p.drop_p) : 1.0f;
  const int t_first = d == 0 ? 0 : T - 1;
  const long dir = d == 0 ? 1 : -1;
  l64_gcptr xq[4];
  l64_gptr oq[4], hq[4], gq[4], cq[4];
  bool live[4];
#pragma unroll
  for (int reg = 0; reg < 4; ++reg) {
    const int row = r0 + fq * 4 + reg;
    live[reg] = FULL || row < B;
    const long rt = (long)(live[reg] ? row : 0) * T + t_first;
    xq[reg] = (l64_gcptr)(p.xproj[d] + rt * p.xs + j);
    oq[reg] = (l64_gptr)(p.out + rt * p.os + d * 64 + j);
    hq[reg] = keep_h ? (l64_gptr)(p.hprev[d] + rt * 64 + j) : (l64_gptr)nullptr;
    const long tr = (long)t_first * B + (live[reg] ? row : 0);
    gq[reg] = stash ? (l64_gptr)(p.gates[d] + tr * 256 + 4 * j) : (l64_gptr)nullptr;   // gate stash: [t][b][unit][i,f,g,o]
    cq[reg] = stash ? (l64_gptr)(p.cst[d] + tr * 64 + j) : (l64_gptr)nullptr;
  }
  const long xstep = dir * p.xs, ostep = dir * p.os, hstep = dir * 64, gstep = dir * (long)B * 256,
             cstep = dir * (long)B * 64;
#define L64_LOAD_XP(DST)                                                                            \
  do {                                                                                              \
    _Pragma("unroll") for (int reg = 0; reg < 4; ++reg) {                                           \
      _Pragma("unroll") for (int g = 0; g < 4; ++g) DST[g][reg] = (FULL || live[reg]) ? xq[reg][g * 64] : 0.f; \
      xq[reg] += xstep;                                                                             \
    }                                                                                               \
  } while (0)
  L64_LOAD_XP(xp);
  for (int s = 0; s < T; ++s) {
    // (unconditional: past the last step the previous address is loaded again -- no branch around the prefetch)
    if (s + 1 >= T) {
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) xq[reg] -= xstep;
    }
    L64_LOAD_XP(xpn);
    if (keep_h) {
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        if (FULL || live[reg]) *hq[reg] = hreg[reg];
        hq[reg] += hstep;
      }
    }
    f32x4 acc[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) acc[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float hk[16];                    // all 16 operand reads in flight before the first MFMA (they were issued in pairs, each
#pragma unroll                       // pair followed by its own wait: eight LDS latencies per step)
    for (int k4 = 0; k4 < 16; ++k4) hk[k4] = hs[(16 * fq + k4) * 16 + fr];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k4 = 0; k4 < 16; ++k4) {
#pragma unroll
      for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(hk[k4], wreg[k4][g], acc[g], 0, 0, 0);
    }
    L64_LDS_BARRIER();  // everyone has finished reading hs for this step
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      float gi = l64_sigmoid(acc[0][reg] + (xp[0][reg] + bh[0]));
      float gf = l64_sigmoid(acc[1][reg] + (xp[1][reg] + bh[1]));
      float gg = l64_tanh(acc[2][reg] + (xp[2][reg] + bh[2]));
      float go = l64_sigmoid(acc[3][reg] + (xp[3][reg] + bh[3]));
      float cn = gf * creg[reg] + gi * gg;
      float hn = go * l64_tanh(cn);
      creg[reg] = cn;
      hreg[reg] = hn;
      hs[j * 16 + fq * 4 + reg] = hn;
      if (FULL || live[reg]) {
        *oq[reg] = hn;
        if (drop) {
          const long off = oq[reg] - (l64_gptr)p.out;
          const float mk = dropout_keep(dkey, (unsigned)off, p.drop_p) ? keep_scale : 0.f;
          p.drop_mask[off] = mk;
          p.drop_y[off] = hn * mk;
        }
        if (stash) {
          *(l64_gptr4)gq[reg] = (l64_f4){gi, gf, gg, go};       // one 16-B store (was four dword stores 256 B apart)
          *cq[reg] = cn;
        }
      }
      oq[reg] += ostep;
      if (stash) { gq[reg] += gstep; cq[reg] += cstep; }
    }
    // (the copy stays HERE: hoisted into the MFMA block by the scheduler, it made every step wait for the loads it had just issued)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) xp[g][reg] = xpn[g][reg];
    L64_LDS_BARRIER();
  }
#undef L64_LOAD_XP
#pragma unroll
  for (int reg = 0; reg < 4; ++reg) {
    int row = r0 + fq * 4 + reg;
    if (row < B) {
      if (p.hn[d]) p.hn[d][row * 64 + j] = hreg[reg];
      if (p.cn[d]) p.cn[d][row * 64 + j] = creg[reg];
    }
  }
}

template <bool FULL, bool STASH, bool DROP>
__global__ __launch_bounds__(256) void lstm64_fwd_kernel(Lstm64P p) { lstm64_fwd_body<FULL, STASH, DROP>(p); }

// Several INDEPENDENT stacks' layers in one launch (grid z = stack): UpperNetwlocal's global and anchor BiLSTM(64) stacks have the same
// shape and no data in common (Net/Upper_Net.py:333-339 and :208-216) -- one after the other each layer was 8 workgroups on a 256-CU
// chip, twice.
#define L64_MAX_MULTI 4
static_assert(sizeof(Lstm64P) == sizeof(MmegoLstm64Fwd) && offsetof(Lstm64P, seed_ctr) == offsetof(MmegoLstm64Fwd, seed_ctr) &&
              offsetof(Lstm64P, B) == offsetof(MmegoLstm64Fwd, B), "MmegoLstm64Fwd is Lstm64P");
struct Lstm64Multi { Lstm64P p[L64_MAX_MULTI]; };
template <bool FULL, bool STASH, bool DROP>
__global__ __launch_bounds__(256) void lstm64_fwd_multi_kernel(Lstm64Multi m) { lstm64_fwd_body<FULL, STASH, DROP>(m.p[blockIdx.z]); }

extern "C" int mmego_lstm64_forward(void* stream, int B, int T, const float* xproj0, const float* xproj1, long xs,
                                    const float* whh0, const float* whh1, const float* bhh0, const float* bhh1,
                                    const float* h0_0, const float* h0_1, const float* c0_0, const float* c0_1, float* out,
                                    long os, float* hn0, float* hn1,
                                    float* cn0, float* cn1, float* gates0, float* gates1, float* cst0, float* cst1,
                                    float* hprev0, float* hprev1, float* drop_y, float* drop_mask, float drop_p,
                                    const unsigned long long* seed_ctr, int salt) {
  MMEGO_REQUIRE((drop_mask == nullptr) == (drop_y == nullptr));
  MMEGO_REQUIRE(!drop_mask || (seed_ctr && drop_p > 0.f && drop_p < 1.f && (long)B * T * os < (1L << 32)));
  MMEGO_REQUIRE(B > 0 && T > 0 && xproj0 && xproj1 && whh0 && whh1 && out);
  MMEGO_REQUIRE((((uintptr_t)whh0 | (uintptr_t)whh1) & 15) == 0);
  MMEGO_REQUIRE((gates0 == nullptr) == (cst0 == nullptr) && (gates1 == nullptr) == (cst1 == nullptr));
  Lstm64P p;
  p.xproj[0] = xproj0; p.xproj[1] = xproj1; p.xs = xs;
  p.whh[0] = whh0; p.whh[1] = whh1;
  p.bhh[0] = bhh0; p.bhh[1] = bhh1;
  p.h0[0] = h0_0; p.h0[1] = h0_1; p.c0[0] = c0_0; p.c0[1] = c0_1;
  p.out = out; p.os = os;
  p.hn[0] = hn0; p.hn[1] = hn1; p.cn[0] = cn0; p.cn[1] = cn1;
  p.gates[0] = gates0; p.gates[1] = gates1; p.cst[0] = cst0; p.cst[1] = cst1;
  p.hprev[0] = hprev0; p.hprev[1] = hprev1;
  p.B = B; p.T = T;
  p.drop_y = drop_y; p.drop_mask = drop_mask; p.drop_p = drop_p; p.seed_ctr = seed_ctr; p.salt = (unsigned)salt;
  size_t lds = (size_t)(64 * 16) * sizeof(float);            // h tile only: W_hh lives in registers
  const bool full = B % 16 == 0, st_ = gates0 != nullptr, dr = drop_mask != nullptr;
  MMEGO_REQUIRE((hprev0 != nullptr) == st_ && (hprev1 != nullptr) == st_ && (gates1 != nullptr) == st_);   // stashes: all or none
#define L64_FWD_LAUNCH(F_, S_, D_)                                                                                    \
  do {                                                                                                                \
    static bool attr_set = false;                                                                                     \
    if (!attr_set) {                                                                                                  \
      hipError_t e = hipFuncSetAttribute((const void*)lstm64_fwd_kernel<F_, S_, D_>,                                  \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                       \
      if (e != hipSuccess) return (int)e;                                                                             \
      attr_set = true;                                                                                                \
    }                                                                                                                 \
    hipLaunchKernelGGL((lstm64_fwd_kernel<F_, S_, D_>), dim3(cdiv(B, 16), 2), dim3(256), lds, (hipStream_t)stream, p); \
  } while (0)
  if (full) {
    if (st_ && dr) L64_FWD_LAUNCH(true, true, true);
    else if (st_) L64_FWD_LAUNCH(true, true, false);
    else if (dr) L64_FWD_LAUNCH(true, false, true);
    else L64_FWD_LAUNCH(true, false, false);
  } else {
    if (st_ && dr) L64_FWD_LAUNCH(false, true, true);
    else if (st_) L64_FWD_LAUNCH(false, true, false);
    else if (dr) L64_FWD_LAUNCH(false, false, true);
    else L64_FWD_LAUNCH(false, false, false);
  }
#undef L64_FWD_LAUNCH
  MMEGO_LAUNCH_CHECK();
  return MMEGO_OK;
}

struct Lstm64BwdP {
  const float* dout; long dos;
  const float* gates[2]; const float* cst[2]; const float* c0[2];
  const float* whh[2];
  float* dgates[2]; long dgs;
  int B, T;
};


// FULL as in lstm64_fwd_kernel: straight-line step loop (no row predicates, the prefetch unconditional, loads from explicit
// global pointers), so the compiler counts its waits instead of draining the memory pipe every step.
template <bool FULL>
__device__ __forceinline__ void lstm64_bwd_body(const Lstm64BwdP& p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* dgs = smem;              // [256 n][16 rows]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int d = blockIdx.y, r0 = blockIdx.x * 16;
  const int B = p.B, T = p.T;
  const int fr = lane & 15, fq = lane >> 4;
  const int j = wave * 16 + fr;
  // W_hh as this lane's 64 MFMA operands (row n = 4*i + fq, column 16*wave + fr), straight from global memory with all
  // loads in flight (the LDS copy it replaces was a rolled loop of 16 dependent global -> LDS round trips per launch)
  float wreg[64];
  {
    const float* W = p.whh[d] + fq * 64 + wave * 16 + fr;
#pragma unroll
    for (int i = 0; i < 64; ++i) wreg[i] = W[i * 256];
  }
  float dcreg[4] = {0.f, 0.f, 0.f, 0.f};
  f32x4 dhrec = {0.f, 0.f, 0.f, 0.f};

  // Everything a step reads from memory (incoming dh, saved gates, cell states) is independent of the recurrence: the
  // next step's values are fetched while this step computes.  c_{t-1} of this step is c_t of the next one.
  float gin[4][6], gnx[4][6];        // per row: dh_in, i, f, g, o, c_prev
  float ccur[4];
  const l64_gcptr g_gates = (l64_gcptr)p.gates[d], g_dout = (l64_gcptr)p.dout, g_cst = (l64_gcptr)p.cst[d];
  const bool has_c0 = p.c0[d] != nullptr;
  const l64_gcptr g_c0 = has_c0 ? (l64_gcptr)p.c0[d] : g_cst;      // (a valid address either way: the value is masked below)
#define L64_LOAD_BWD(DST, step)                                                                                \
  do {                                                                                                         \
    const int tq_ = d == 0 ? (step) : T - 1 - (step);                                                          \
    const int tp_ = d == 0 ? tq_ - 1 : tq_ + 1;                                                                \
    _Pragma("unroll") for (int reg = 0; reg < 4; ++reg) {                                                      \
      const int row_ = r0 + fq * 4 + reg;                                                                      \
      if (FULL || row_ < B) {                                                                                  \
        const l64_f4 gv_ = *(l64_gcptr4)(g_gates + ((long)tq_ * B + row_) * 256 + 4 * j);                      \
        DST[reg][0] = g_dout[((long)row_ * T + tq_) * p.dos + d * 64 + j];                                     \
        DST[reg][1] = gv_.x; DST[reg][2] = gv_.y; DST[reg][3] = gv_.z; DST[reg][4] = gv_.w;                    \
        const l64_gcptr cp_ = ((step) > 0) ? g_cst + ((long)tp_ * B + row_) * 64 + j : g_c0 + (long)row_ * 64 + j; \
        const float cv_ = *cp_;                                                                                \
        DST[reg][5] = ((step) > 0 || has_c0) ? cv_ : 0.f;                                                      \
      } else {                                                                                                 \
        _Pragma("unroll") for (int q = 0; q < 6; ++q) DST[reg][q] = 0.f;                                       \
      }                                                                                                        \
    }                                                                                                          \
  } while (0)
  L64_LOAD_BWD(gin, T - 1);
  {
    const int tl = d == 0 ? T - 1 : 0;
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      int row = r0 + fq * 4 + reg;
      ccur[reg] = (FULL || row < B) ? g_cst[((long)tl * B + row) * 64 + j] : 0.f;
    }
  }
  for (int s = T - 1; s >= 0; --s) {
    const int tt = d == 0 ? s : T - 1 - s;
    const int sn = s > 0 ? s - 1 : 0;          // (unconditional prefetch: the last step loads step 0's values again)
    L64_LOAD_BWD(gnx, sn);
    float dg4[4][4];
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      int row = r0 + fq * 4 + reg;
      if (FULL || row < B) {
        float dh = gin[reg][0] + dhrec[reg];
        float gi = gin[reg][1], gf = gin[reg][2], gg = gin[reg][3], go = gin[reg][4];
        float c = ccur[reg];
        float cprev = gin[reg][5];
        float tc = l64_tanh(c);
        float dc = dcreg[reg] + dh * go * (1.f - tc * tc);
        dg4[0][reg] = dc * gg * gi * (1.f - gi);
        dg4[1][reg] = dc * cprev * gf * (1.f - gf);
        dg4[2][reg] = dc * gi * (1.f - gg * gg);
        dg4[3][reg] = dh * tc * go * (1.f - go);
        dcreg[reg] = dc * gf;
        l64_gptr dst = (l64_gptr)p.dgates[d] + ((long)row * T + tt) * p.dgs + j;
#pragma unroll
        for (int g = 0; g < 4; ++g) dst[g * 64] = dg4[g][reg];
      } else {
#pragma unroll
        for (int g = 0; g < 4; ++g) dg4[g][reg] = 0.f;
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) dgs[(g * 64 + j) * 16 + fq * 4 + reg] = dg4[g][reg];
      ccur[reg] = gin[reg][5];
    }
    L64_LDS_BARRIER();
    // dh_rec[row][k] = sum_n dgates[row][n] * W_hh[n][k]; this wave owns k in [16*wave, 16*wave+16)
    // four accumulator chains (a dependent 16x16x4 MFMA can issue every 32 cycles, an independent one every 8) and the
    // operand reads in groups of 16 ahead of their MFMAs (fixed summation order: (a0 + a1) + (a2 + a3))
    f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f}, a2 = {0.f, 0.f, 0.f, 0.f}, a3 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int nb = 0; nb < 64; nb += 16) {
      float dk[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) dk[u] = dgs[((nb + u) * 4 + fq) * 16 + fr];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < 16; u += 4) {
        a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(dk[u], wreg[nb + u], a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(dk[u + 1], wreg[nb + u + 1], a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(dk[u + 2], wreg[nb + u + 2], a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(dk[u + 3], wreg[nb + u + 3], a3, 0, 0, 0);
      }
    }
    dhrec = (a0 + a1) + (a2 + a3);
    __builtin_amdgcn_sched_barrier(0);       // (keep the copy of the prefetched values at the end of the step: see the forward kernel)
#pragma unroll
    for (int reg = 0; reg < 4; ++reg)
#pragma unroll
      for (int q = 0; q < 6; ++q) gin[reg][q] = gnx[reg][q];
    L64_LDS_BARRIER();
  }
#undef L64_LOAD_BWD
}

template <bool FULL>
__global__ __launch_bounds__(256) void lstm64_bwd_kernel(Lstm64BwdP p) { lstm64_bwd_body<FULL>(p); }
static_assert(sizeof(Lstm64BwdP) == sizeof(MmegoLstm64Bwd) && offsetof(Lstm64BwdP, dgs) == offsetof(MmegoLstm64Bwd, dgs), "MmegoLstm64Bwd is Lstm64BwdP");
struct Lstm64BwdMulti { Lstm64BwdP p[L64_MAX_MULTI]; };
template <bool FULL>
__global__ __launch_bounds__(256) void lstm64_bwd_multi_kernel(Lstm64BwdMulti m) { lstm64_bwd_body<FULL>(m.p[blockIdx.z]); }

extern "C" int mmego_lstm64_backward(void* stream, int B, int T, const float* dout, long dos, const float* gates0,
                                     const float* gates1, const float* cst0, const float* cst1, const float* c0_0,
                                     const float* c0_1, const float* whh0, const float* whh1, float* dgates0,
                                     float* dgates1, long dgs) {
  MMEGO_REQUIRE(B > 0 && T > 0 && dout && gates0 && gates1 && cst0 && cst1 && whh0 && whh1 && dgates0 && dgates1);
  MMEGO_REQUIRE((((uintptr_t)whh0 | (uintptr_t)whh1) & 15) == 0);
  Lstm64BwdP p;
  p.dout = dout; p.dos = dos;
  p.gates[0] = gates0; p.gates[1] = gates1; p.cst[0] = cst0; p.cst[1] = cst1;
  p.c0[0] = c0_0; p.c0[1] = c0_1;
  p.whh[0] = whh0; p.whh[1] = whh1;
  p.dgates[0] = dgates0; p.dgates[1] = dgates1; p.dgs = dgs;
  p.B = B; p.T = T;
  const size_t lds = (size_t)(256 * 16) * sizeof(float);
  if (B % 16 == 0) hipLaunchKernelGGL(lstm64_bwd_kernel<true>, dim3(cdiv(B, 16), 2), dim3(256), lds, (hipStream_t)stream, p);
  else hipLaunchKernelGGL(lstm64_bwd_kernel<false>, dim3(cdiv(B, 16), 2), dim3(256), lds, (hipStream_t)stream, p);
  MMEGO_LAUNCH_CHECK();
  return MMEGO_OK;
}

// n <= 4 independent stacks' layers per launch.  descs: n host structs MmegoLstm64Fwd / MmegoLstm64Bwd (include/mmego_hip.h: the argument
// lists of mmego_lstm64_forward / _backward as structs); every stack must have the same B and T and the same options (stashes for all or
// none, dropout for all or none).
extern "C" int mmego_lstm64_forward_multi(void* stream, int n, const void* descs) {
  MMEGO_REQUIRE(descs && n >= 1 && n <= L64_MAX_MULTI);
  const Lstm64P* h = static_cast<const Lstm64P*>(descs);
  Lstm64Multi m;
  const int B = h[0].B, T = h[0].T;
  const bool st_ = h[0].gates[0] != nullptr, dr = h[0].drop_mask != nullptr;
  for (int i = 0; i < n; ++i) {
    const Lstm64P& p = h[i];
    MMEGO_REQUIRE(p.B == B && p.T == T && B > 0 && T > 0 && p.xproj[0] && p.xproj[1] && p.whh[0] && p.whh[1] && p.out);
    MMEGO_REQUIRE((((uintptr_t)p.whh[0] | (uintptr_t)p.whh[1]) & 15) == 0);
    MMEGO_REQUIRE((p.drop_mask == nullptr) == (p.drop_y == nullptr) && (p.drop_mask != nullptr) == dr);
    MMEGO_REQUIRE(!p.drop_mask || (p.seed_ctr && p.drop_p > 0.f && p.drop_p < 1.f && (long)B * T * p.os < (1L << 32)));
    MMEGO_REQUIRE((p.gates[0] != nullptr) == st_ && (p.gates[1] != nullptr) == st_ && (p.cst[0] != nullptr) == st_ && (p.cst[1] != nullptr) == st_ &&
                  (p.hprev[0] != nullptr) == st_ && (p.hprev[1] != nullptr) == st_);
    m.p[i] = p;
  }
  for (int i = n; i < L64_MAX_MULTI; ++i) m.p[i] = h[0];
  const size_t lds = (size_t)(64 * 16) * sizeof(float);
  const bool full = B % 16 == 0;
#define L64_FWDM_LAUNCH(F_, S_, D_) hipLaunchKernelGGL((lstm64_fwd_multi_kernel<F_, S_, D_>), dim3(cdiv(B, 16), 2, n), dim3(256), lds, (hipStream_t)stream, m)
  if (full) {
    if (st_ && dr) L64_FWDM_LAUNCH(true, true, true);
    else if (st_) L64_FWDM_LAUNCH(true, true, false);
    else if (dr) L64_FWDM_LAUNCH(true, false, true);
    else L64_FWDM_LAUNCH(true, false, false);
  } else {
    if (st_ && dr) L64_FWDM_LAUNCH(false, true, true);
    else if (st_) L64_FWDM_LAUNCH(false, true, false);
    else if (dr) L64_FWDM_LAUNCH(false, false, true);
    else L64_FWDM_LAUNCH(false, false, false);
  }
#undef L64_FWDM_LAUNCH
  MMEGO_LAUNCH_CHECK();
  return MMEGO_OK;
}

extern "C" int mmego_lstm64_backward_multi(void* stream, int n, const void* descs) {
  MMEGO_REQUIRE(descs && n >= 1 && n <= L64_MAX_MULTI);
  const Lstm64BwdP* h = static_cast<const Lstm64BwdP*>(descs);
  Lstm64BwdMulti m;
  const int B = h[0].B, T = h[0].T;
  for (int i = 0; i < n; ++i) {
    const Lstm64BwdP& p = h[i];
    MMEGO_REQUIRE(p.B == B && p.T == T && B > 0 && T > 0 && p.dout && p.gates[0] && p.gates[1] && p.cst[0] && p.cst[1] && p.whh[0] && p.whh[1] &&
                  p.dgates[0] && p.dgates[1]);
    MMEGO_REQUIRE((((uintptr_t)p.whh[0] | (uintptr_t)p.whh[1]) & 15) == 0);
    m.p[i] = p;
  }
  for (int i = n; i < L64_MAX_MULTI; ++i) m.p[i] = h[0];
  const size_t lds = (size_t)(256 * 16) * sizeof(float);
  if (B % 16 == 0) hipLaunchKernelGGL(lstm64_bwd_multi_kernel<true>, dim3(cdiv(B, 16), 2, n), dim3(256), lds, (hipStream_t)stream, m);
  else hipLaunchKernelGGL(lstm64_bwd_multi_kernel<false>, dim3(cdiv(B, 16), 2, n), dim3(256), lds, (hipStream_t)stream, m);
  MMEGO_LAUNCH_CHECK();
  return MMEGO_OK;
}
