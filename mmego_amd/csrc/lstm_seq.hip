// A whole BiLSTM layer's recurrence for <= 64 rows and H = 512 as ONE persistent launch (IMU_Net's rnn_slow: reference
// Net/IMU_Net.py:61-62,82 -- nn.LSTM(2H, H, 2 layers, bidirectional) over the T frames of a sequence), weights stationary.
//
// Per timestep the launch-per-step form (lstm_step_small_kernel, lstm_step.hip) pays a ~4.5-us launch boundary and streams all of
// W_hh (8.4 MB for both directions) from memory again -- kernel boundaries invalidate the XCDs' L2s: 9.4 us per step, MFMA busy 0.09.
// Here the 256 workgroups stay for all T steps:
//   * batch rows are independent through the recurrence, so the chip is cut into 8 GROUPS of 32 workgroups -- group g = workgroup
//     index % 8, which the dispatcher's round robin puts on XCD g (a performance property, nothing depends on it for correctness):
//     direction g / 4, rows [16 (g % 4), +16).  Nothing is exchanged between groups.
//   * inside a group, workgroup c owns hidden units [16 c, 16 c + 16): its four waves hold the 4 gates' W_hh rows for those units
//     as v_mfma_f32_16x16x4_f32 B-operand REGISTERS (128 per lane, loaded once, under the product-less first timestep), the cell
//     state of its 16 x 16 (row, unit) elements lives in one register per thread.
//   * the exchange is IN-BAND: a workgroup stores its 16 x 16 slice of h_t as 8-byte (value, tag) words -- tag = launch generation and
//     timestep, one atomic store each -- into a two-stage exchange buffer and goes on; nobody waits for a store to be acknowledged
//     and there is no arrival counter (store -> ack -> atomic -> poll -> load were four dependent trips to the memory side: 3.5-4.0
//     us per step in scripts/bench_handoff.hip, variant C; 6.2 us per step for the kernel built that way).  A consumer fetches the
//     group's 16 x 512 words in two batches and checks every tag (a batch with a word still missing is fetched again: agent-scope
//     loads are served at L2 speed, a retry is cheap -- polling one word per producer first was 2 us per layer slower), copies the
//     values into LDS, multiplies (128 MFMA steps per wave over four accumulation chains), swaps the four gate tiles through LDS and
//     updates its cells.  The plain h_t goes to the layer's output tensor beside it.
// Every spin is bounded: a grid that cannot make progress (three such launches side by side would wait for each other's CUs) sets
// an error word and runs to its end with wrong results instead of hanging the GPU; the host checks the word (blocks.seq_xcd_errors:
// the trainers and evaluation passes of processors.py read it once per epoch / pass and raise).  A wave that has timed out once -- or, while
// spinning, finds the error word set by another wave (looked at every LQ_SPIN_CHECK retries, nothing on the normal path) -- skips every
// later wait, so a stuck launch ends ~one spin bound after it got stuck whatever T is, not after one bound per wait; workgroups that
// only become resident after that (their tags are long overwritten) leave after LQ_SPIN_CHECK retries.
// The last workgroup to finish resets the counters, so a launch leaves them as it found them (graph replays need no memset node).
// Arithmetic: the products accumulate in another order than the step kernels' (k split over lane groups and four chains), so results
// agree with them to fp32 rounding, not bit for bit; gate non-linearities and the cell update are the step kernels' expressions.
#include <stdlib.h>

#include "common.h"

#define LQ_H 512
#define LQ_RS (LQ_H + 4)                  // LDS row stride of the h tile (floats): 16-byte aligned, rows on distinct bank groups
#define LQ_SPIN_MAX (1 << 20)        // ~1 s of retries: far beyond any wait for co-resident workgroups, short enough that a stuck grid ends
#define LQ_SPIN_CHECK (1 << 10)      // a spinning wave looks at the error word every ~1 ms: once any wave has given up, all do

struct LstmSeqP {
  const float* xproj; long xs;           // input projections [Bn*T rows (b*T + t)][8H]: direction d at column offset d * 4H; xs = row stride
  const float* whh[2]; const float* bhh[2];
  float* out; long os;                   // [Bn*T][2H] (os = row stride): h_t of direction d at column offset d * H
  unsigned* sync;                        // [8 unused | done counter | error word | launch generation], zero before the first launch
  unsigned long long* xbuf;              // [2 stages][8 groups][16 rows][512] (value bits | tag << 32), zero before the first launch
  int Bn, T;
};

// (the step kernels' forms: lstm_step.hip)
__device__ __forceinline__ float lq_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float lq_tanh(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * x)); }

__global__ __launch_bounds__(256, 2) void lstm_seq_xcd_kernel(LstmSeqP p) {
  __shared__ __attribute__((aligned(16))) float hs[16 * LQ_RS];       // h_{t-1} of the group's 16 rows
  __shared__ float gs[4][16][17];                                      // the four gate tiles [gate][row][unit]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wg = blockIdx.x, g = wg & 7, c = wg >> 3;
  const int d = g >> 2, r0 = (g & 3) * 16;
  const int fr = lane & 15, fq = lane >> 4;
  const int T = p.T, H = LQ_H;
  unsigned* done = p.sync + 8;
  unsigned* err = p.sync + 9;
  // (stream order: the launch before this one -- whose last workgroup bumped the generation -- has completed)
  const unsigned gen = __hip_atomic_load(p.sync + 10, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  unsigned long long* xg = p.xbuf + (long)g * 16 * LQ_H;

  // ---- this wave's gate (= wave) of the workgroup's 16 units: W_hh[gate H + 16 c + fr][128 fq .. + 128) -> 128 registers.
  // Lane group fq owns k in [128 fq, 128 fq + 128) for BOTH operands (any k order is valid as long as they agree).
  f32x4 wreg[32];
  {
    const float* wrow = p.whh[d] + ((long)wave * H + 16 * c + fr) * H + 128 * fq;
#pragma unroll
    for (int j = 0; j < 32; ++j) wreg[j] = *reinterpret_cast<const f32x4*>(wrow + 4 * j);
  }
  // ---- the cell-update thread of element (row = tid / 16, unit = tid % 16)
  const int crow = tid >> 4, cu = tid & 15;
  const int brow = r0 + crow;                                   // batch row
  const bool live = brow < p.Bn;
  const int unit = 16 * c + cu;
  float bh[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) bh[q] = p.bhh[d] ? p.bhh[d][q * H + unit] : 0.f;
  const long xrow = (long)(live ? brow : 0) * T;
  const float* xq = p.xproj + d * 4 * H + unit;
  float* oq = p.out + d * H + unit;
  float creg = 0.f;
  float xp[4], xpn[4];
  bool dead = false;                       // wave-uniform: this launch has failed (here or elsewhere); no more waiting
  {
    const int t = d == 0 ? 0 : T - 1;
#pragma unroll
    for (int q = 0; q < 4; ++q) xp[q] = xq[(xrow + t) * p.xs + q * H];
  }
  for (int s = 0; s < T; ++s) {
    const int t = d == 0 ? s : T - 1 - s;
    const int tn = s + 1 < T ? (d == 0 ? t + 1 : t - 1) : t;   // (past the end: this step's address again, no branch around the loads)
#pragma unroll
    for (int q = 0; q < 4; ++q) xpn[q] = xq[(xrow + tn) * p.xs + q * H];
    float pre[4] = {0.f, 0.f, 0.f, 0.f};
    if (s > 0) {
      // ---- the group's h_{t-1}: fetch all 16 x 512 words, check every tag, fetch a batch again while a word is missing
      const unsigned tag = gen * 4096u + (unsigned)s;             // written by the producers at step s - 1
      const unsigned long long* xs_ = xg + (long)((s - 1) & 1) * 8 * 16 * LQ_H;
      {
        // element (row u / 2, k = 256 (u % 2) + tid) in round u: an instruction reads 2 KB of one row; two batches of 16 rounds
        // (the weights hold 128 of the 256 registers)
#pragma unroll
        for (int hb = 0; hb < 2; ++hb) {
          unsigned long long v[16];
          bool ok = true;
#pragma unroll
          for (int u = 0; u < 16; ++u) {
            const int uu = 16 * hb + u;
            v[u] = __hip_atomic_load(xs_ + (uu >> 1) * LQ_H + (uu & 1) * 256 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
#pragma unroll
          for (int u = 0; u < 16; ++u) ok = ok && (unsigned)(v[u] >> 32) == tag;
          int spins = 0;
          while (!dead && !__all(ok)) {                           // (rare: a producer's stores have not all landed yet)
            ok = true;
#pragma unroll
            for (int u = 0; u < 16; ++u) {
              const int uu = 16 * hb + u;
              v[u] = __hip_atomic_load(xs_ + (uu >> 1) * LQ_H + (uu & 1) * 256 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              ok = ok && (unsigned)(v[u] >> 32) == tag;
            }
            ++spins;
            if (spins > LQ_SPIN_MAX) { if (lane == 0) atomicOr(err, 1u); dead = true; }
            else if ((spins & (LQ_SPIN_CHECK - 1)) == 0) {
              // somebody else has given up already (ONE lane asks: an atomic load by 64 lanes is 64 operations on one address)
              unsigned e = 0u;
              if (lane == 0) e = __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              if (__builtin_amdgcn_readfirstlane(e) != 0u) dead = true;
            }
          }
#pragma unroll
          for (int u = 0; u < 16; ++u) {
            const int uu = 16 * hb + u;
            hs[(uu >> 1) * LQ_RS + (uu & 1) * 256 + tid] = __uint_as_float((unsigned)v[u]);
          }
        }
      }
      __syncthreads();
      f32x4 acc[4];
#pragma unroll
      for (int a = 0; a < 4; ++a) acc[a] = (f32x4){0.f, 0.f, 0.f, 0.f};
      const float* ha = hs + fr * LQ_RS + 128 * fq;
#pragma unroll
      for (int j = 0; j < 32; ++j) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(ha + 4 * j);
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, wreg[j].x, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, wreg[j].y, acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, wreg[j].z, acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, wreg[j].w, acc[3], 0, 0, 0);
      }
      // D layout: lane (fr, fq), register i: row 4 fq + i, unit fr
#pragma unroll
      for (int i = 0; i < 4; ++i) gs[wave][4 * fq + i][fr] = (acc[0][i] + acc[1][i]) + (acc[2][i] + acc[3][i]);
      __syncthreads();
#pragma unroll
      for (int q = 0; q < 4; ++q) pre[q] = gs[q][crow][cu];
    }
    // ---- cell update of (row, unit): PyTorch gate order i, f, g, o
    const float gi = lq_sigmoid(pre[0] + (xp[0] + bh[0]));
    const float gf = lq_sigmoid(pre[1] + (xp[1] + bh[1]));
    const float gg = lq_tanh(pre[2] + (xp[2] + bh[2]));
    const float go = lq_sigmoid(pre[3] + (xp[3] + bh[3]));
    creg = gf * creg + gi * gg;
    const float hval = go * lq_tanh(creg);
    if (live) oq[(xrow + t) * p.os] = hval;
    if (s + 1 < T) {
      // (value, tag) in one 8-byte store; stage s & 1 was last read at step s - 1, which every workgroup of the group has left:
      // it produced h_{s-1}... the words of step s - 2 are dead once anybody can be here
      const unsigned long long w = (unsigned long long)__float_as_uint(hval) | ((unsigned long long)(gen * 4096u + (unsigned)(s + 1)) << 32);
      __hip_atomic_store(xg + (long)(s & 1) * 8 * 16 * LQ_H + crow * LQ_H + unit, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) xp[q] = xpn[q];
    __syncthreads();                                              // (gs and hs may be rewritten)
  }
  // ---- the last workgroup of the launch advances the generation (tags of this launch can never match again)
  if (tid == 0) {
    const unsigned n = __hip_atomic_fetch_add(done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (n == gridDim.x - 1) {
      __hip_atomic_store(done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(p.sync + 10, (gen + 1u) & 0xFFFFFu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// How many mmego_lstm_seq_xcd launches the CURRENT device can hold at once (the 256 workgroups of a launch wait for each other: they
// must all be resident).  From the kernel's real occupancy -- hipOccupancyMaxActiveBlocksPerMultiprocessor x CU count, queried per
// device -- not from an assumed two workgroups per CU; 0 on a device or partition that cannot hold one launch (the caller then keeps
// the launch-per-timestep form), 2 on a whole MI355X (256 CUs x 2).
extern "C" int mmego_lstm_seq_xcd_slots(void) {
  static int slots[64];
  static bool have[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
  if (!have[dev]) {
    int cus = 0, per_cu = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, lstm_seq_xcd_kernel, 256, 0) != hipSuccess) per_cu = 0;
    slots[dev] = (int)(((long)cus * per_cu) / 256);
    have[dev] = true;
  }
  return slots[dev];
}

// 1 when mmego_lstm_seq_xcd takes the shape on the current device
extern "C" int mmego_lstm_seq_xcd_ok(int Bn, int H, int T) {
  return Bn >= 1 && Bn <= 64 && H == LQ_H && T >= 1 && T <= 4095 && mmego_lstm_seq_xcd_slots() >= 1;
}

// One BiLSTM layer's recurrence, both directions, all T timesteps in one launch.  xproj [Bn*T][xs >= 8H] rows (b*T + t): W_ih x + b_ih of
// direction d at columns [4H d, 4H d + 4H); out [Bn*T][os >= 2H]: h_t of direction d at columns [H d, H d + H); h_0 = c_0 = 0.
// sync: 16 unsigned words and xbuf: 2 * 8 * 16 * 512 8-byte words, both zero before the first launch and private to one stream of
// launches (word 8: done counter, left zero; word 9: sticky error flag, non-zero after a launch whose workgroups could not all
// become resident -- its results are invalid; word 10: launch generation).
extern "C" int mmego_lstm_seq_xcd(void* stream, const float* xproj, long xs, const float* whh0, const float* whh1, const float* bhh0,
                                  const float* bhh1, float* out, long os, unsigned* sync, unsigned long long* xbuf, int Bn, int H, int T) {
  MMEGO_REQUIRE(xproj && whh0 && whh1 && out && sync && xbuf && mmego_lstm_seq_xcd_ok(Bn, H, T) && xs >= 8 * H && os >= 2 * H);
  MMEGO_REQUIRE((((uintptr_t)whh0 | (uintptr_t)whh1) & 15) == 0);
  LstmSeqP p;
  p.xproj = xproj; p.xs = xs; p.whh[0] = whh0; p.whh[1] = whh1; p.bhh[0] = bhh0; p.bhh[1] = bhh1;
  p.out = out; p.os = os; p.sync = sync; p.xbuf = xbuf; p.Bn = Bn; p.T = T;
  hipLaunchKernelGGL(lstm_seq_xcd_kernel, dim3(256), dim3(256), 0, (hipStream_t)stream, p);
  MMEGO_LAUNCH_CHECK();
  return MMEGO_OK;
}
