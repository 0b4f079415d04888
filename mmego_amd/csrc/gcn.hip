// ST-GCN layer pieces of Lower_Net's KeyEncoder (reference Net/GCN.py:55-64 ConvTemporalGraphical, :108-122 the 9x1 temporal
// convolution of st_gcn.tcn) on channels-last rows (b, t, v):
//   graph_mix    einsum('nkctv,kvw->nctw', x, A * edge_importance): y[f][w][c] = sum_k sum_v (A.imp)[k][v][w] z[f][v][k*C + c],
//                and its input gradient dz[f][v][k*C + c] = sum_w (A.imp)[k][v][w] dy[f][w][c].  A . edge_importance is formed
//                in LDS by the kernel itself (no `mul` launch, no K batched-product launches with a broadcast 15 x 15 operand),
//                a frame's z tile is staged in LDS once and every output is a 15-term dot product from there.
//   tconv        the 9x1 temporal convolution (padding 4) as an IMPLICIT GEMM: out[r][co] = b[co] + sum_tap sum_ci
//                act(in[r + (tap-4) V][ci]) W[co][ci][tap], rows (b, t, v), taps that leave the sequence contribute zero.  The
//                unfolded operand (9x the activation, 9.4 GB per config-5 forward) never exists: per tap the A tile is the SAME
//                64 rows shifted by (tap-4) V rows.  The preceding BatchNorm + ReLU (st_gcn.tcn[0..1]) is applied while the tile
//                is loaded; a training step keeps the activated rows (they come out of the centre tap's loads).  The input
//                gradient of the convolution is the same kernel on flipped, transposed weights (tconv_pack mode 1).
//                The conv weight [co][ci][tap] as it lies would put the 64 lanes of a tile load 36 B apart (measured: 61 us per
//                launch at the training shape), so the weights are re-packed k-contiguous first:
//   tconv_pack   W[co][ci][tap] -> Wp[tap][co][ci] (mode 0) or Wp[tap][ci][co] with the taps reversed (mode 1); mode 2 writes both,
//                one behind the other (a training step packs once in its forward pass and uses the second half in backward).
//   tconv_wgrad  the weight gradient, implicit too: per tap a product over the row axis of dY and the shifted activated rows,
//                row axis split over workgroups, partial tiles added in a fixed order.
// 64 x 64 output tiles, v_mfma_f32_32x32x2_f32, operands in LDS, next chunk prefetched in registers.
#include <stdlib.h>
#include "gcn_stats.h"

#define GC_S 66          // even row stride: fragments are read as aligned float2 (lanes r = 0..31 hit 64 distinct banks)

// stats (optional, forward): per-frame BatchNorm partials of the output, stats[(c*F + f)*3 + {0,1,2}] = (V, mean, M2) of channel c
// over the frame's V rows -- the record format of colstats_kernel (bn.hip) with one "block" per frame, so that the batch
// statistics of the BatchNorm behind the einsum (st_gcn.tcn[0]) need no pass of their own over y.
__global__ __launch_bounds__(256) void graph_mix_kernel(const float* __restrict__ X, const float* __restrict__ A,
                                                        const float* __restrict__ imp, float* __restrict__ Y, long F, int V,
                                                        int Kk, int C, int backward, float* __restrict__ stats, long ldx) {
  extern __shared__ float sm[];
  // forward : X = z [F][V][Kk*C] -> Y = y [F][V][C];   backward: X = dy [F][V][C] -> Y = dz [F][V][Kk*C]
  const int Cin = backward ? C : Kk * C, Cout = backward ? Kk * C : C;
  float* As = sm;                          // [Kk][V][V]  A . importance
  float* Xs = sm + Kk * V * V;             // [V][Cin + 1]
  float* Ys = Xs + V * (Cin + 1);          // [V][Cout + 1]  (only with stats)
  for (int i = threadIdx.x; i < Kk * V * V; i += blockDim.x) As[i] = A[i] * imp[i];
  for (long f = blockIdx.x; f < F; f += gridDim.x) {
    __syncthreads();
    const float* xf = X + f * V * ldx;                     // (ldx: row stride of X, >= Cin -- X may be a column slice)
    for (int i = threadIdx.x; i < V * Cin; i += blockDim.x) Xs[(i / Cin) * (Cin + 1) + (i % Cin)] = xf[(long)(i / Cin) * ldx + (i % Cin)];
    __syncthreads();
    float* yf = Y + f * V * Cout;
    for (int i = threadIdx.x; i < V * Cout; i += blockDim.x) {
      const int row = i / Cout, col = i - row * Cout;
      float acc = 0.f;
      if (!backward) {                     // row = w, col = c
        for (int k = 0; k < Kk; ++k)
          for (int v = 0; v < V; ++v) acc += As[(k * V + v) * V + row] * Xs[v * (Cin + 1) + k * C + col];
      } else {                             // row = v, col = k*C + c
        const int k = col / C, c = col - k * C;
        for (int w = 0; w < V; ++w) acc += As[(k * V + row) * V + w] * Xs[w * (Cin + 1) + c];
      }
      yf[i] = acc;
      if (stats) Ys[row * (Cout + 1) + col] = acc;
    }
    if (stats) {
      __syncthreads();
      for (int c = threadIdx.x; c < Cout; c += blockDim.x) {
        float sum = 0.f;
        for (int r = 0; r < V; ++r) sum += Ys[r * (Cout + 1) + c];
        const float mean = sum / (float)V;
        float m2 = 0.f;
        for (int r = 0; r < V; ++r) { const float d = Ys[r * (Cout + 1) + c] - mean; m2 += d * d; }
        float* rec = stats + ((long)c * F + f) * 3;
        rec[0] = (float)V; rec[1] = mean; rec[2] = m2;
      }
    }
  }
}

__global__ __launch_bounds__(256) void tconv_pack_kernel(const float* __restrict__ W, int Co, int Ci, int taps, int mode,
                                                         float* __restrict__ Wp) {
  const long total = (long)Co * Ci * taps;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int tap = (int)(i % taps);
    const long q = i / taps;
    const int ci = (int)(q % Ci), co = (int)(q / Ci);
    const float w = W[i];
    if (mode != 1) Wp[((long)tap * Co + co) * Ci + ci] = w;                                      // mode 0 / 2: forward pack
    if (mode != 0) Wp[(mode == 2 ? total : 0) + ((long)(taps - 1 - tap) * Ci + ci) * Co + co] = w;   // mode 1 / 2: gradient pack
  }
}

struct TconvP {
  const float* X; long ldx;          // input rows (b, t, v) x Cin
  const float* in_state;             // [4][Cin] mean, invstd, a, b of the BatchNorm in front (+ ReLU); null: plain input
  const float* W;                    // packed [taps][Cout][Cin] (tconv_pack)
  const float* bias;                 // [Cout] or null
  float* Y; long ldy;
  float* act;                        // optional [rows][Cin]: the activated input (kept for the backward pass), written from the
                                     // centre tap's tile loads by the workgroups of the first column tile
  long rows; int T, V, Cin, Cout, taps;
  // fused training step (REC; gcn_fused.hip): the BatchNorm in front comes as partial records and is finalized in the prologue
  // (in_state unused), and the output tile's own (mean, M2) per column leaves as out_rec[row tile][Cout]
  BnRefD in_bn; float2* out_rec;
  // backward statistics epilogue (input-gradient call, Y = d(act)): with g = Y . [bn(ym) > 0], xhat = (ym - mean) invstd of the
  // BatchNorm whose state is bw_st [4][Cout], bw_rec[row tile][Cout] = (sum g, sum g xhat) over the tile's rows
  const float* bw_ym; long ldym; const float* bw_st; float2* bw_rec;
};

// acc[32x32] += A[32 rows][K] . B[32 rows][K]^T (row-major LDS tiles, stride GC_S), K a multiple of 4 (zero padded).
// One ds_read_b64 per operand feeds two MFMA steps: lane (r, h) holds k = 4j + 2h, 4j + 2h + 1 -- the same k-permutation on
// both operands, so the sum over k is unchanged.
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define TC_PIN4(v) asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w))
// Workgroup = NG groups of 4 waves on ONE 64 x 64 output tile; group g takes the (tap, channel-chunk) steps g, g + NG, ... with
// operand tiles of its own, so NG steps' loads are in flight at once, and the groups' partial tiles are added through LDS in a
// fixed order at the end.  NG = 4 for small row counts (the training shape: 120 row tiles, where one step at a time left the
// kernel waiting on 9-18 dependent memory round trips), NG = 1 when the grid alone fills the chip.
// Tile loads: 16-byte loads from CLAMPED addresses, all eight of a step issued before the first is used, masks applied to the
// values afterwards (a predicate on the load itself compiles to branch + load + wait: one memory round trip per load).
template <int NG, bool ACT, bool REC = false>
__global__ __launch_bounds__(256 * NG) void tconv_kernel(TconvP p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, grp = tid >> 8, t = tid & 255, lane = tid & 63, wave = t >> 6;
  float* stl = smem + NG * 2 * 64 * GC_S;                  // REC: [4][Cin] mean, a, b, invstd of the BatchNorm in front
  float* As = smem + grp * 2 * 64 * GC_S;
  float* Bs = As + 64 * GC_S;
  const int rt = wave & 1, ct = wave >> 1;
  const long r0 = (long)blockIdx.x * 64;
  const int n0 = blockIdx.y * 64;
  const int half = p.taps / 2;
  const int c4 = t & 15, rr = t >> 4;                      // this thread's 16-byte column and its rows rr + 16 j of a tile
  const int nkc = (p.Cin + 63) / 64;                       // 64-channel chunks per tap
  const int nchunks = p.taps * nkc;
  // per row of the tile: element offset of its (clamped) input row, its frame index t (rows past the end: far outside), the
  // weight row's offset -- 32-bit (the launcher checks the extents), so that a step's addresses cost a few adds per load
  unsigned xoff[4], woff[4];
  int tv[4];
  bool nok[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const long row = r0 + rr + 16 * j;
    const bool in = row < p.rows;
    xoff[j] = (unsigned)(in ? row : p.rows - 1) * (unsigned)p.ldx;
    tv[j] = in ? (int)(((unsigned)row / (unsigned)p.V) % (unsigned)p.T) : -(1 << 20);
    const int n = n0 + rr + 16 * j;
    nok[j] = n < p.Cout;
    woff[j] = (unsigned)(nok[j] ? n : p.Cout - 1) * (unsigned)p.Cin;
  }
  const int vstep = p.V * (int)p.ldx;                      // one frame, in elements
  f32x4 av[4], bv[4], mu = {0.f, 0.f, 0.f, 0.f}, sa = mu, sb = mu;
  // TC_ISSUE: the loads of step ch (clamped addresses, nothing waits on them); TC_FINISH: masks and activation, run when the
  // values are about to be stored to LDS -- the MFMAs of the step before sit between the two
#define TC_ISSUE(ch)                                                                                \
  do {                                                                                              \
    const int tap_ = (ch) / nkc, k0_ = ((ch) % nkc) * 64;                                           \
    const int d_ = tap_ - half;                                                                     \
    const int kk_ = k0_ + 4 * c4;                                                                   \
    const int kc_ = kk_ < p.Cin ? kk_ : 0;                                                          \
    if (ACT && !REC) {                                                                              \
      mu = *reinterpret_cast<const f32x4*>(p.in_state + kc_);                                       \
      sa = *reinterpret_cast<const f32x4*>(p.in_state + 2 * p.Cin + kc_);                           \
      sb = *reinterpret_cast<const f32x4*>(p.in_state + 3 * p.Cin + kc_);                           \
    }                                                                                               \
    const float* wt_ = p.W + (long)tap_ * p.Cout * p.Cin;                         /* uniform */     \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                 \
      const unsigned ts_ = (unsigned)(tv[j] + d_);                                                  \
      const unsigned o_ = xoff[j] + (unsigned)(ts_ < (unsigned)p.T ? d_ * vstep : 0) + (unsigned)kc_; \
      av[j] = *reinterpret_cast<const f32x4*>(p.X + o_);                                            \
    }                                                                                               \
    _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                   \
      bv[j] = *reinterpret_cast<const f32x4*>(wt_ + (woff[j] + (unsigned)kc_));                     \
  } while (0)
#define TC_FINISH(ch)                                                                               \
  do {                                                                                              \
    const int d_ = (ch) / nkc - half;                                                               \
    const bool kok_ = ((ch) % nkc) * 64 + 4 * c4 < p.Cin;                                           \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) { TC_PIN4(av[j]); TC_PIN4(bv[j]); }               \
    if (ACT && REC) {                      /* the state finalized in this kernel's prologue: from LDS, when the values are used */ \
      const int kq_ = kok_ ? ((ch) % nkc) * 64 + 4 * c4 : 0;                                        \
      mu = *reinterpret_cast<const f32x4*>(stl + kq_);                                              \
      sa = *reinterpret_cast<const f32x4*>(stl + p.Cin + kq_);                                      \
      sb = *reinterpret_cast<const f32x4*>(stl + 2 * p.Cin + kq_);                                  \
    }                                                                                               \
    if (ACT) { TC_PIN4(mu); TC_PIN4(sa); TC_PIN4(sb); }                                             \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                 \
      const int ts_ = tv[j] + d_;                                                                   \
      f32x4 v_ = av[j];                                                                             \
      if (ACT) {                                                                                    \
        v_.x = fmaxf(__builtin_fmaf(v_.x - mu.x, sa.x, sb.x), 0.f);                                 \
        v_.y = fmaxf(__builtin_fmaf(v_.y - mu.y, sa.y, sb.y), 0.f);                                 \
        v_.z = fmaxf(__builtin_fmaf(v_.z - mu.z, sa.z, sb.z), 0.f);                                 \
        v_.w = fmaxf(__builtin_fmaf(v_.w - mu.w, sa.w, sb.w), 0.f);                                 \
      }                                                                                             \
      const f32x4 z_ = {0.f, 0.f, 0.f, 0.f};                                                        \
      av[j] = (kok_ && ts_ >= 0 && ts_ < p.T) ? v_ : z_;                                            \
      bv[j] = (nok[j] && kok_) ? bv[j] : z_;                                                        \
    }                                                                                               \
  } while (0)
  // (issued unconditionally, past the last step on a clamped index: a load under a condition would have its values copied -- and
  // waited for -- where the condition ends)
  TC_ISSUE(grp < nchunks ? grp : nchunks - 1);
  // (REC: the first step's loads fly while the statistics of the BatchNorm in front are finalized from its partial records)
  if (REC) bn_from_records<256 * NG>(p.in_bn, p.Cin, p.rows, reinterpret_cast<double*>(smem), stl, blockIdx.x == 0 && blockIdx.y == 0);
  f32x16 acc = {0};
  const bool live = n0 + ct * 32 < p.Cout && r0 + rt * 32 < p.rows;      // (a wave whose whole 32 x 32 part is padding skips the MFMAs)
  for (int ch0 = 0; ch0 < nchunks; ch0 += NG) {
    const int ch = ch0 + grp;
    __syncthreads();
    if (ch < nchunks) {
      TC_FINISH(ch);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        f32x2* ad = reinterpret_cast<f32x2*>(As + (rr + 16 * j) * GC_S + 4 * c4);
        f32x2* bd = reinterpret_cast<f32x2*>(Bs + (rr + 16 * j) * GC_S + 4 * c4);
        ad[0] = f32x2{av[j].x, av[j].y}; ad[1] = f32x2{av[j].z, av[j].w};
        bd[0] = f32x2{bv[j].x, bv[j].y}; bd[1] = f32x2{bv[j].z, bv[j].w};
      }
      if (ACT && p.act && blockIdx.y == 0 && ch / nkc == half) {
        const int kk = (ch % nkc) * 64 + 4 * c4;
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (kk < p.Cin && tv[j] >= 0) *reinterpret_cast<f32x4*>(p.act + (r0 + rr + 16 * j) * p.Cin + kk) = av[j];
      }
    }
    __syncthreads();
    TC_ISSUE(ch + NG < nchunks ? ch + NG : nchunks - 1);
    __builtin_amdgcn_sched_barrier(0);
    if (ch < nchunks && live) {
      const int kleft = p.Cin - (ch % nkc) * 64;
      const int K = kleft >= 64 ? 64 : ((kleft + 3) & ~3);
      const int r = lane & 31, h = lane >> 5;
      const f32x2* bp = reinterpret_cast<const f32x2*>(Bs + (ct * 32 + r) * GC_S + 2 * h);
      const f32x2* ap = reinterpret_cast<const f32x2*>(As + (rt * 32 + r) * GC_S + 2 * h);
#pragma unroll 4
      for (int k = 0; k < K; k += 4) {
        const f32x2 a = ap[k >> 1], b = bp[k >> 1];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
      }
    }
  }
#undef TC_ISSUE
#undef TC_FINISH
  const int lcol = ct * 32 + (lane & 31);
  if (NG > 1) {                                            // groups 1 .. NG-1 hand their partial tiles to group 0 (fixed order)
    __syncthreads();
    float* xch = smem;                                     // [NG-1][64][64]
    if (grp > 0) {
#pragma unroll
      for (int reg = 0; reg < 16; ++reg)
        xch[((grp - 1) * 64 + rt * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)) * 64 + lcol] = acc[reg];
    }
    __syncthreads();
    if (grp == 0) {
#pragma unroll
      for (int g = 1; g < NG; ++g)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg)
          acc[reg] += xch[((g - 1) * 64 + rt * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)) * 64 + lcol];
    }
  }
  const int col = n0 + lcol;
  if (grp == 0 && col < p.Cout) {
    const float bb = p.bias ? p.bias[col] : 0.f;
    float* yp = p.Y + (r0 + rt * 32 + 4 * (lane >> 5)) * p.ldy + col;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) acc[reg] += bb;     // (final values in registers of their own before the first store: a value
                                                           // computed under a store's predicate reuses one register, and overwriting a
                                                           // store's data register waits for that store)
    if (r0 + 64 <= p.rows) {                               // whole tile: stores without a predicate each (one behind the other)
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) yp[(long)((reg & 3) + 8 * (reg >> 2)) * p.ldy] = acc[reg];
    } else {
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const long row = r0 + rt * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
        if (row < p.rows) yp[(long)((reg & 3) + 8 * (reg >> 2)) * p.ldy] = acc[reg];
      }
    }
  }
  if (p.out_rec || p.bw_rec) {
    // statistics of the output tile for the BatchNorm behind it (forward) / of the BatchNorm + ReLU the gradient passes next
    // (backward): the tile goes through LDS once, thread (column, row quarter) walks 16 rows, quarters added in a fixed order
    __syncthreads();                                       // (the exchange tiles above have been read)
    float* tl = smem;                                      // [64][65]
    float* part = smem + 64 * 65;                          // [4][64][2]
    if (grp == 0) {
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) tl[(rt * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)) * 65 + lcol] = acc[reg];
    }
    __syncthreads();
    const int cx = t & 63, rq = t >> 6;
    const int nvr = (int)(p.rows - r0 < 64 ? p.rows - r0 : 64);
    const int cg = n0 + cx < p.Cout ? n0 + cx : p.Cout - 1;
    float s1 = 0.f, s2 = 0.f;
    if (grp == 0) {
      if (p.out_rec) {                                     // shifted sums: (mean, M2) of the column over the tile's valid rows
        const float shift = tl[cx];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          const int row = rq * 16 + u;
          const float d = row < nvr ? tl[row * 65 + cx] - shift : 0.f;
          s1 += d; s2 = __builtin_fmaf(d, d, s2);
        }
      } else {
        const float mu_ = p.bw_st[cg], is_ = p.bw_st[p.Cout + cg], a_ = p.bw_st[2 * p.Cout + cg], b_ = p.bw_st[3 * p.Cout + cg];
        float ym[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          const long row = r0 + rq * 16 + u;
          ym[u] = p.bw_ym[(row < p.rows ? row : p.rows - 1) * p.ldym + cg];
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          asm volatile("" : "+v"(ym[u]));
          const int row = rq * 16 + u;
          const float g = (row < nvr && __builtin_fmaf(ym[u] - mu_, a_, b_) > 0.f) ? tl[row * 65 + cx] : 0.f;
          s1 += g;
          s2 += g * ((ym[u] - mu_) * is_);
        }
      }
      part[(rq * 64 + cx) * 2] = s1; part[(rq * 64 + cx) * 2 + 1] = s2;
    }
    __syncthreads();
    if (grp == 0 && rq == 0 && n0 + cx < p.Cout) {
      float a = 0.f, b = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) { a += part[(j * 64 + cx) * 2]; b += part[(j * 64 + cx) * 2 + 1]; }
      if (p.out_rec) p.out_rec[(long)blockIdx.x * p.Cout + n0 + cx] = rec_from_shifted(tl[cx], a, b, nvr);
      else p.bw_rec[(long)blockIdx.x * p.Cout + n0 + cx] = float2{a, b};
    }
  }
}

// Weight gradient of the temporal convolution, implicit as well: dW[co][ci][tap] = sum_r dY[r][co] act[r + (tap-4) V][ci] (rows
// whose shifted frame leaves the sequence contribute nothing) -- per tap a product over the ROW axis of two tiles read as they
// lie in memory (k-major: row r of both operands is one LDS row, a lane reads its output channel's column).  The row axis is
// cut into nsplit slabs (grid.x), one 64 x 64 (co, ci) tile per tap and workgroup; the partial tiles go to ws[split][tap][co][ci]
// and tconv_wgrad_reduce adds them in split order into the convolution's [co][ci][tap] layout.
struct TconvWgP {
  const float* dY; long lddy;
  const float* Xa; long ldx;
  float* ws;
  long rows, rows_per_split;
  int T, V, Cin, Cout, taps;
};
#define TW_S 72          // LDS row stride (floats): 16-byte aligned rows
#define TW_K 64          // rows per chunk: 8 16-byte loads per thread in flight, 37 KB of LDS -- several workgroups per CU (one 128-row
                         // chunk per CU left every load round trip exposed: 55 us for the 128-channel block at 7680 rows)
// acc += sum over KN k-major LDS rows of a[k][m] b[k][n]: fragment reads of the next 16 rows are issued before the MFMAs of the
// current 16 (one wave per SIMD: nothing else hides the LDS latency; reading right in front of each MFMA ran at 0.25 of the peak)
template <int KN>
__device__ __forceinline__ void tw_mfma(const float* ap, const float* bp, f32x16& acc) {
  static_assert(KN % 16 == 0, "k range: whole groups of 16 rows");
  float a0[8], b0[8], a1[8], b1[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { a0[i] = ap[2 * i * TW_S]; b0[i] = bp[2 * i * TW_S]; }
#pragma unroll
  for (int k = 0; k < KN; k += 32) {
    if (k + 16 < KN) {
#pragma unroll
      for (int i = 0; i < 8; ++i) { a1[i] = ap[(k + 16 + 2 * i) * TW_S]; b1[i] = bp[(k + 16 + 2 * i) * TW_S]; }
    }
    __builtin_amdgcn_sched_barrier(0);                     // (the scheduler would sink every read to its MFMA again)
#pragma unroll
    for (int i = 0; i < 8; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[i], b0[i], acc, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (k + 32 < KN) {
#pragma unroll
      for (int i = 0; i < 8; ++i) { a0[i] = ap[(k + 32 + 2 * i) * TW_S]; b0[i] = bp[(k + 32 + 2 * i) * TW_S]; }
    }
    __builtin_amdgcn_sched_barrier(0);
    if (k + 16 < KN) {
#pragma unroll
      for (int i = 0; i < 8; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[i], b1[i], acc, 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// KSPLIT (both channel counts <= 32: one 32 x 32 tile): the four waves take a quarter of every chunk's rows each and add
// their tiles through LDS at the end, in wave order.
template <bool KSPLIT>
__global__ __launch_bounds__(256) void tconv_wgrad_kernel(TconvWgP p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Ds = smem;
  float* Xs = smem + TW_K * TW_S;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int rt = KSPLIT ? 0 : wave & 1, ct = KSPLIT ? 0 : wave >> 1;
  const int ntn = (p.Cin + 63) / 64, ntm = (p.Cout + 63) / 64;
  const int tap = blockIdx.y / (ntm * ntn), tile = blockIdx.y % (ntm * ntn);
  const int m0 = (tile / ntn) * 64, n0 = (tile % ntn) * 64;
  const int d = tap - p.taps / 2;
  const long rbeg = (long)blockIdx.x * p.rows_per_split;
  const long rend = rbeg + p.rows_per_split < p.rows ? rbeg + p.rows_per_split : p.rows;
  const int c4 = t & 15, rr = t >> 4;                      // 16-byte column, rows rr + 16 j of a chunk
  const int mm = m0 + 4 * c4, nn = n0 + 4 * c4;
  const bool mok = mm < p.Cout, nok = nn < p.Cin;
  const int mc = mok ? mm : 0, ncl = nok ? nn : 0;
  constexpr int NJ = TW_K / 16;
  f32x4 dv[NJ], xv[NJ];
  // frame index t of every row of a chunk (rows past the split's end: far outside), one division per ROW and chunk instead of
  // one per load (measured: the address arithmetic of 16 loads per thread took three times the chunk's MFMA time), for the
  // chunk in flight and the next one
  int* tab = reinterpret_cast<int*>(smem + 2 * TW_K * TW_S);   // [2][TW_K]
#define TW_TAB(buf, rb)                                                                             \
  if (t < TW_K) {                                                                                   \
    const long row_ = (rb) + t;                                                                     \
    tab[(buf) * TW_K + t] = row_ < rend ? (int)(((unsigned)row_ / (unsigned)p.V) % (unsigned)p.T) : -(1 << 20); \
  }
  const unsigned lastd = (unsigned)(p.rows - 1) * (unsigned)p.lddy, lastx = (unsigned)(p.rows - 1) * (unsigned)p.ldx;
  const int vstep = d * p.V * (int)p.ldx;                  // this tap's shift, in elements
#define TW_ISSUE(buf, rb)                                                                           \
  do {                                                                                              \
    _Pragma("unroll") for (int j = 0; j < NJ; ++j) {                                                \
      const int tq_ = tab[(buf) * TW_K + rr + 16 * j];                                              \
      const unsigned row_ = (unsigned)((rb) + rr + 16 * j);                                         \
      const unsigned od_ = tq_ >= 0 ? row_ * (unsigned)p.lddy : lastd;                              \
      const unsigned ox_ = tq_ >= 0 ? row_ * (unsigned)p.ldx : lastx;                               \
      dv[j] = *reinterpret_cast<const f32x4*>(p.dY + (od_ + (unsigned)mc));                        \
      xv[j] = *reinterpret_cast<const f32x4*>(p.Xa + (ox_ + (unsigned)((unsigned)(tq_ + d) < (unsigned)p.T ? vstep : 0) + (unsigned)ncl)); \
    }                                                                                               \
  } while (0)
#define TW_FINISH(buf)                                                                              \
  do {                                                                                              \
    _Pragma("unroll") for (int j = 0; j < NJ; ++j) { TC_PIN4(dv[j]); TC_PIN4(xv[j]); }              \
    _Pragma("unroll") for (int j = 0; j < NJ; ++j) {                                                \
      const int tq_ = tab[(buf) * TW_K + rr + 16 * j];                                              \
      const f32x4 z_ = {0.f, 0.f, 0.f, 0.f};                                                        \
      dv[j] = (tq_ >= 0 && mok) ? dv[j] : z_;                                                       \
      xv[j] = ((unsigned)(tq_ + d) < (unsigned)p.T && nok) ? xv[j] : z_;                            \
    }                                                                                               \
  } while (0)
  f32x16 acc = {0};
  const bool live = m0 + rt * 32 < p.Cout && n0 + ct * 32 < p.Cin;
  TW_TAB(0, rbeg)
  __syncthreads();
  TW_ISSUE(0, rbeg);                                       // (unconditional, rows past the end clamped: see tconv_kernel)
  int buf = 0;
  for (long rb = rbeg; rb < rend; rb += TW_K, buf ^= 1) {
    TW_TAB(buf ^ 1, rb + TW_K)
    __syncthreads();
    TW_FINISH(buf);
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      *reinterpret_cast<f32x4*>(Ds + (rr + 16 * j) * TW_S + 4 * c4) = dv[j];
      *reinterpret_cast<f32x4*>(Xs + (rr + 16 * j) * TW_S + 4 * c4) = xv[j];
    }
    __syncthreads();
    TW_ISSUE(buf ^ 1, rb + TW_K);
    __builtin_amdgcn_sched_barrier(0);
    if (live) {
      const int r = lane & 31, h = lane >> 5;
      const int kb = KSPLIT ? wave * (TW_K / 4) : 0;
      const float* ap = Ds + (kb + h) * TW_S + rt * 32 + r;
      const float* bp = Xs + (kb + h) * TW_S + ct * 32 + r;
      tw_mfma<KSPLIT ? TW_K / 4 : TW_K>(ap, bp, acc);
    }
  }
#undef TW_ISSUE
#undef TW_FINISH
#undef TW_TAB
  if (KSPLIT) {                                            // waves 1..3 hand their tiles to wave 0
    __syncthreads();
    float* xch = smem;                                     // [3][16][64]
    if (wave > 0) {
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) xch[((wave - 1) * 16 + reg) * 64 + lane] = acc[reg];
    }
    __syncthreads();
    if (wave > 0) return;
#pragma unroll
    for (int w = 0; w < 3; ++w)
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) acc[reg] += xch[(w * 16 + reg) * 64 + lane];
  }
  const int n = n0 + ct * 32 + (lane & 31);
  if (n < p.Cin) {
    float* out = p.ws + ((long)blockIdx.x * p.taps + tap) * p.Cout * p.Cin;
    const int mw = m0 + rt * 32;
    if (mw + 32 <= p.Cout) {                               // whole 32-row part: stores without a predicate each
      float* op = out + (long)(mw + 4 * (lane >> 5)) * p.Cin + n;
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) op[(long)((reg & 3) + 8 * (reg >> 2)) * p.Cin] = acc[reg];
    } else {
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int m = mw + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
        if (m < p.Cout) out[(long)m * p.Cin + n] = acc[reg];
      }
    }
  }
}

__global__ __launch_bounds__(256) void tconv_wgrad_reduce_kernel(const float* __restrict__ ws, int nsplit, int taps, int Co, int Ci,
                                                                 float* __restrict__ dW, int accumulate) {
  const long per = (long)taps * Co * Ci;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < per; i += (long)gridDim.x * blockDim.x) {
    float s = 0.f;
    for (int k = 0; k < nsplit; ++k) s += ws[k * per + i];
    const int ci = (int)(i % Ci);
    const long q = i / Ci;
    const int co = (int)(q % Co), tap = (int)(q / Co);
    float* o = dW + ((long)co * Ci + ci) * taps + tap;
    *o = accumulate ? *o + s : s;
  }
}

extern "C" int mmego_graph_mix(void* stream, const float* X, const float* A, const float* importance, float* Y, long F, int V, int K,
                               int C, int backward, float* stats, long ldx) {
  MMEGO_REQUIRE(X && A && importance && Y && F > 0 && V >= 1 && V <= 32 && K >= 1 && K <= 4 && C >= 1);
  MMEGO_REQUIRE(ldx >= (backward ? C : K * C));
  MMEGO_REQUIRE(!stats || (!backward && F <= 1024));     // (one record per frame and channel: mmego_bn_finalize takes <= 1024)
  const int Cin = backward ? C : K * C, Cout = backward ? K * C : C;
  const size_t lds = (size_t)(K * V * V + V * (Cin + 1) + (stats ? V * (Cout + 1) : 0)) * sizeof(float);
  MMEGO_REQUIRE(lds <= 64 * 1024);
  const int grid = (int)(F < 2048 ? F : 2048);
  hipLaunchKernelGGL(graph_mix_kernel, dim3(grid), dim3(256), lds, (hipStream_t)stream, X, A, importance, Y, F, V, K, C, backward,
                     stats, ldx);
  MMEGO_LAUNCH_CHECK();
  return MMEGO_OK;
}

extern "C" int mmego_tconv_pack(void* stream, const float* W, int Co, int Ci, int taps, int mode, float* Wp) {
  MMEGO_REQUIRE(W && Wp && Co >= 1 && Ci >= 1 && taps >= 1 && mode >= 0 && mode <= 2);
  long b = ((long)Co * Ci * taps + 255) / 256;
  hipLaunchKernelGGL(tconv_pack_kernel, dim3((int)(b > 1024 ? 1024 : b)), dim3(256), 0, (hipStream_t)stream, W, Co, Ci, taps, mode, Wp);
  MMEGO_LAUNCH_CHECK();
  return MMEGO_OK;
}

static int tconv_launch(hipStream_t st, TconvP& p, bool rec) {
  dim3 grid((unsigned)((p.rows + 63) / 64), (unsigned)((p.Cout + 63) / 64));
  const size_t extra = rec ? (size_t)4 * p.Cin * sizeof(float) : 0;
  const bool act = rec || p.in_state;
  if ((long)grid.x * grid.y <= 512) {                      // small grid: four steps in flight per workgroup
    const size_t lds = (size_t)4 * 2 * 64 * GC_S * sizeof(float) + extra;
    static bool attr = false;
    if (!attr) {
      const int mx = (int)((size_t)4 * 2 * 64 * GC_S * sizeof(float) + 4 * 256 * sizeof(float));
      hipError_t e = hipFuncSetAttribute((const void*)tconv_kernel<4, false>, hipFuncAttributeMaxDynamicSharedMemorySize, mx);
      if (e == hipSuccess) e = hipFuncSetAttribute((const void*)tconv_kernel<4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, mx);
      if (e == hipSuccess) e = hipFuncSetAttribute((const void*)tconv_kernel<4, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, mx);
      if (e != hipSuccess) return (int)e;
      attr = true;
    }
    if (rec) hipLaunchKernelGGL((tconv_kernel<4, true, true>), grid, dim3(1024), lds, st, p);
    else if (act) hipLaunchKernelGGL((tconv_kernel<4, true>), grid, dim3(1024), lds, st, p);
    else hipLaunchKernelGGL((tconv_kernel<4, false>), grid, dim3(1024), lds, st, p);
  } else {
    // (one step in flight: the prologue scratch of the record form needs 32 KB, the operand tiles are 33 KB)
    const size_t lds = (size_t)2 * 64 * GC_S * sizeof(float) + extra;
    if (rec) hipLaunchKernelGGL((tconv_kernel<1, true, true>), grid, dim3(256), lds, st, p);
    else if (act) hipLaunchKernelGGL((tconv_kernel<1, true>), grid, dim3(256), lds, st, p);
    else hipLaunchKernelGGL((tconv_kernel<1, false>), grid, dim3(256), lds, st, p);
  }
  MMEGO_LAUNCH_CHECK();
  return MMEGO_OK;
}

extern "C" int mmego_tconv(void* stream, const float* X, long ldx, const float* in_state, const float* Wp, const float* bias, float* Y,
                           long ldy, float* act, int B, int T, int V, int Cin, int Cout, int taps) {
  MMEGO_REQUIRE((long)B * T * V * ldx < (1L << 30) && (long)Cout * Cin < (1L << 30));   // (32-bit element offsets)
  MMEGO_REQUIRE(X && Wp && Y && B > 0 && T > 0 && V > 0 && Cin >= 4 && Cout >= 1 && taps >= 1 && (taps & 1) && ldx >= Cin && ldy >= Cout);
  MMEGO_REQUIRE((Cin % 4) == 0 && (ldx % 4) == 0 && (((uintptr_t)X | (uintptr_t)Wp | (uintptr_t)in_state | (uintptr_t)act) & 15) == 0);
  MMEGO_REQUIRE(!act || in_state);
  TconvP p = {X, ldx, in_state, Wp, bias, Y, ldy, act, (long)B * T * V, T, V, Cin, Cout, taps};
  return tconv_launch((hipStream_t)stream, p, false);
}

// Training forward of the fused step: the BatchNorm (+ ReLU) in front is given by its partial records (finalized in the prologue by
// every workgroup; workgroup 0 writes its state and updates the running statistics), the activated rows are kept in `act`, and the
// (mean, M2) records of the OUTPUT's columns per 64-row tile go to out_rec[ceil(rows / 64)][Cout] for the BatchNorm behind.
extern "C" int mmego_tconv_train(void* stream, const float* X, long ldx, const void* in_bn, const float* Wp, const float* bias, float* Y,
                                 long ldy, float* act, float* out_rec, int B, int T, int V, int Cin, int Cout, int taps) {
  const MmegoBnRefH* h = static_cast<const MmegoBnRefH*>(in_bn);
  MMEGO_REQUIRE((long)B * T * V * ldx < (1L << 30) && (long)Cout * Cin < (1L << 30));
  MMEGO_REQUIRE(X && h && Wp && Y && out_rec && B > 0 && T > 0 && V > 0 && Cin >= 4 && Cin <= 256 && Cout >= 1 && taps >= 1 && (taps & 1) &&
                ldx >= Cin && ldy >= Cout);
  MMEGO_REQUIRE((Cin % 4) == 0 && (ldx % 4) == 0 && (((uintptr_t)X | (uintptr_t)Wp | (uintptr_t)act) & 15) == 0);
  MMEGO_REQUIRE(h->rec && h->nrec >= 1 && h->rows_per_rec >= 1 && h->gamma && h->beta && (h->running_mean == nullptr) == (h->running_var == nullptr));
  TconvP p = {X, ldx, nullptr, Wp, bias, Y, ldy, act, (long)B * T * V, T, V, Cin, Cout, taps};
  p.in_bn = bnref_device(h);
  p.out_rec = reinterpret_cast<float2*>(out_rec);
  return tconv_launch((hipStream_t)stream, p, true);
}

// Input gradient of the temporal convolution (X = dY, Wp = the gradient pack, Y = d(act) [rows][Cout]) with the backward sums of the
// BatchNorm + ReLU in front of the convolution's input taken in the epilogue: bw_rec[ceil(rows / 64)][Cout] = (sum g, sum g xhat) per
// 64-row tile with g = Y . [bn(ymix) > 0], xhat = (ymix - mean) invstd, state = [4][Cout] mean, invstd, a, b of that BatchNorm.
extern "C" int mmego_tconv_bwd_stats(void* stream, const float* dY, long lddy, const float* Wp, float* dAct, long ldda, const float* ymix,
                                     long ldym, const float* state, float* bw_rec, int B, int T, int V, int Cin, int Cout, int taps) {
  MMEGO_REQUIRE((long)B * T * V * lddy < (1L << 30) && (long)Cout * Cin < (1L << 30));
  MMEGO_REQUIRE(dY && Wp && dAct && ymix && state && bw_rec && B > 0 && T > 0 && V > 0 && Cin >= 4 && Cout >= 1 && taps >= 1 && (taps & 1) &&
                lddy >= Cin && ldda >= Cout && ldym >= Cout);
  MMEGO_REQUIRE((Cin % 4) == 0 && (lddy % 4) == 0 && (((uintptr_t)dY | (uintptr_t)Wp) & 15) == 0);
  TconvP p = {dY, lddy, nullptr, Wp, nullptr, dAct, ldda, nullptr, (long)B * T * V, T, V, Cin, Cout, taps};
  p.bw_ym = ymix; p.ldym = ldym; p.bw_st = state; p.bw_rec = reinterpret_cast<float2*>(bw_rec);
  return tconv_launch((hipStream_t)stream, p, false);
}

// ---------------------------------------------------------------------------------------------------------------------------
// Temporal convolution, SEQUENCE-TILED (training shape: T V <= 128 rows per sequence).  The 9 taps of a (t, v) row reach only rows
// of the same sequence, so a workgroup takes ONE sequence and a 32-column slice of the output: the sequence's (activated) input
// rows go to LDS once -- not once per tap -- and every tap is the same tile read 15 d rows further down, rows outside the sequence
// masked to zero.  The weights come fragment-major per tap (mmego_pack_multi kind 2): a wave fetches its B operand straight into
// registers with coalesced 1-KB reads, the next tap's while the current one is multiplied; no LDS for W, no barrier in the tap loop.
// 4 waves = 4 row tiles of 32; K = taps x Cin per wave.  LDS: 128 x (Cin + 4) floats (68 KB at Cin = 128).
//   REC   forward of a training step: BatchNorm (+ReLU) in front from partial records (prologue), `act` kept, out_rec[b][Cout] =
//         (mean, M2) of the output over the sequence's rows
//   else  input gradient with the backward sums of the BatchNorm + ReLU in front of the convolution's input: bw_rec[b][Cout]
// The old tile kernel took 21 / 26 / 40 us at 32 / 64 / 128 channels and 7 680 rows: 9 x Cin / 64 steps of (barrier, tile loads
// from L2, LDS store, barrier, MFMA) with 120 workgroups -- overhead-bound below 128 channels.
struct TconvSeqP {
  const float* X; long ldx;
  const float* Wf;                  // fragment-major [taps][Cout/32][Cin/32][4][64][4]
  const float* bias; float* Y; long ldy; float* act;
  int B, TV, V, Cin, Cout, taps;
  BnRefD in_bn; float2* out_rec;
  const float* bw_ym; long ldym; const float* bw_st; float2* bw_rec;
};

template <int NK, bool REC>
__global__ __launch_bounds__(256) void tconv_seq_kernel(TconvSeqP p) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int Cin = p.Cin, XS = Cin + 4, TV = p.TV;
  float* xs = sm;                                          // [128][XS] the sequence's rows; later the output tile [128][36]
  float* stl = sm + 128 * XS;                              // [4][Cin] (REC)
  float* scr = sm + (128 * XS + 4 * Cin > 2 * 128 * 36 ? 128 * XS + 4 * Cin : 2 * 128 * 36);      // [512] partial sums (behind both output-phase tiles)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
  const long b = blockIdx.x;
  const int ct = blockIdx.y, n0 = ct * 32;
  const long r0 = b * TV;
  // ---- the first tap's weight fragments and the sequence's rows: requested first
  f32x4 wf[2][NK][4];
  const float* wbase = p.Wf + ((long)ct * NK * 4) * 256 + lane * 4;
#define TS_WLOAD(buf, tap)                                                                           \
  _Pragma("unroll") for (int kc = 0; kc < NK; ++kc)                                                  \
    _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                    \
      wf[buf][kc][j] = *reinterpret_cast<const f32x4*>(wbase + (long)(tap) * p.Cout * Cin + (kc * 4 + j) * 256)
  TS_WLOAD(0, 0);
  const int c4n = Cin / 4, n4 = TV * c4n;
  constexpr int NP = NK * 4;                               // 16-byte pieces per thread: 128 rows x Cin / 4 / 256 threads
  f32x4 xv[NP];
#pragma unroll
  for (int u = 0; u < NP; ++u) {
    const int i = tid + 256 * u < n4 ? tid + 256 * u : n4 - 1;
    xv[u] = *reinterpret_cast<const f32x4*>(p.X + (r0 + i / c4n) * p.ldx + 4 * (i % c4n));
  }
  f32x4 ymv[4];                                            // backward statistics: this sequence's ymix columns n0 .. n0 + 31
  if (!REC && p.bw_rec) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = tid + 256 * u < TV * 8 ? tid + 256 * u : TV * 8 - 1;
      ymv[u] = *reinterpret_cast<const f32x4*>(p.bw_ym + (r0 + (i >> 3)) * p.ldym + n0 + 4 * (i & 7));
    }
  }
  if (REC) bn_from_records<256>(p.in_bn, Cin, (long)p.B * TV, reinterpret_cast<double*>(sm), stl, blockIdx.x == 0 && blockIdx.y == 0);
#pragma unroll
  for (int u = 0; u < NP; ++u) {
    asm volatile("" : "+v"(xv[u].x), "+v"(xv[u].y), "+v"(xv[u].z), "+v"(xv[u].w));
    const int i = tid + 256 * u;
    const int row = i / c4n, c = 4 * (i % c4n);
    if (i < 128 * c4n) {
      f32x4 v = xv[u];
      if (REC) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(__builtin_fmaf(v[e] - stl[c + e], stl[Cin + c + e], stl[2 * Cin + c + e]), 0.f);
      }
      const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
      *reinterpret_cast<f32x4*>(xs + row * XS + c) = row < TV ? v : z4;
    }
  }
  __syncthreads();
  if (REC && p.act && ct == 0) {                           // the activated rows, kept for backward: 16-byte pieces, reads ahead of stores
    f32x4 o[NP];
#pragma unroll
    for (int u = 0; u < NP; ++u) {
      const int i = tid + 256 * u < n4 ? tid + 256 * u : n4 - 1;
      o[u] = *reinterpret_cast<const f32x4*>(xs + (i / c4n) * XS + 4 * (i % c4n));
    }
#pragma unroll
    for (int u = 0; u < NP; ++u) {
      const int i = tid + 256 * u < n4 ? tid + 256 * u : n4 - 1;
      *reinterpret_cast<f32x4*>(p.act + (r0 + i / c4n) * Cin + 4 * (i % c4n)) = o[u];
    }
  }
  // ---- taps: acc[32 rows of this wave][32 columns] += shift(xs, 15 d) . W_tap^T.  A lane whose shifted row leaves the sequence
  // reads the tile's zero row 127 instead (T V < 128): no select in the loop; the next chunk's A fragments are requested before the
  // current chunk's 16 MFMAs (one wave per SIMD: nothing else hides the LDS latency), the next tap's weights a whole tap ahead.
  f32x16 acc = {0};
  constexpr int TAPS = 9, half = TAPS / 2;                 // (straight-line tap loop: across a loop's back edge the compiler waits for
                                                           // EVERY outstanding load, i.e. for the weights it has just requested)
  const int rowl = wave * 32 + r;
#define TS_AROW(tp_) (xs + ((rowl < TV && (unsigned)(rowl + ((tp_) - half) * p.V) < (unsigned)TV) ? rowl + ((tp_) - half) * p.V : 127) * XS + 16 * h)
#define TS_ALOAD(buf, ptr, kc_)                                                                       \
  _Pragma("unroll") for (int j = 0; j < 4; ++j) af[buf][j] = *reinterpret_cast<const f32x4*>((ptr) + (kc_) * 32 + 4 * j)
  f32x4 af[2][4];
  const float* ap = TS_AROW(0);
  TS_ALOAD(0, ap, 0);
#pragma unroll
  for (int tp = 0; tp < TAPS; ++tp) {
    const int sub = tp & 1;
    const int tn = tp + 1 < TAPS ? tp + 1 : tp;
    if (tp + 1 < TAPS) { if (sub == 0) TS_WLOAD(1, tn); else TS_WLOAD(0, tn); }
    const float* apn = TS_AROW(tn);
#pragma unroll
    for (int kc = 0; kc < NK; ++kc) {
      const int cur = (tp * NK + kc) & 1;
      if (kc + 1 < NK) { if (cur) TS_ALOAD(0, ap, kc + 1); else TS_ALOAD(1, ap, kc + 1); }
      else if (tp + 1 < TAPS) { if (cur) TS_ALOAD(0, apn, 0); else TS_ALOAD(1, apn, 0); }
      __builtin_amdgcn_sched_barrier(0);                   // (the next chunk's reads stay AHEAD of this chunk's MFMAs)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][j][e], wf[sub][kc][j][e], acc, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    ap = apn;
  }
#undef TS_AROW
#undef TS_ALOAD
#undef TS_WLOAD
  __syncthreads();                                         // (every wave has read the input tile: it becomes the output tile)
  float* os = sm;                                          // [128][36]
  {
    const float bb = p.bias ? p.bias[n0 + r] : 0.f;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) os[(wave * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h) * 36 + r] = acc[reg] + bb;
  }
  __syncthreads();
  {                                                        // output rows: 16-byte pieces, reads ahead of the stores
    f32x4 o[4];
    const int n8 = TV * 8;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = tid + 256 * u < n8 ? tid + 256 * u : n8 - 1;
      o[u] = *reinterpret_cast<const f32x4*>(os + (i >> 3) * 36 + 4 * (i & 7));
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = tid + 256 * u < n8 ? tid + 256 * u : n8 - 1;
      *reinterpret_cast<f32x4*>(p.Y + (r0 + (i >> 3)) * p.ldy + n0 + 4 * (i & 7)) = o[u];
    }
  }
  // ---- statistics over the sequence's TV rows for this workgroup's 32 columns: thread (column, row part of 8)
  const int cx = tid & 31, part = tid >> 5;
  if (REC) {
    if (p.out_rec) {
      const float shift = os[cx];
      float s1 = 0.f, s2 = 0.f;
      for (int rr = part; rr < TV; rr += 8) { const float d = os[rr * 36 + cx] - shift; s1 += d; s2 = __builtin_fmaf(d, d, s2); }
      scr[(part * 32 + cx) * 2] = s1; scr[(part * 32 + cx) * 2 + 1] = s2;
      __syncthreads();
      if (tid < 32) {
        float a = 0.f, c = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) { a += scr[(j * 32 + tid) * 2]; c += scr[(j * 32 + tid) * 2 + 1]; }
        p.out_rec[b * p.Cout + n0 + tid] = rec_from_shifted(shift, a, c, TV);
      }
    }
  } else if (p.bw_rec) {
    // g = dAct . [bn(ym) > 0], xhat = (ym - mean) invstd: the prefetched ymix pieces go through LDS to the (column, part) threads
    float* yt = sm + 128 * 36;                             // [128][36]
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      asm volatile("" : "+v"(ymv[u].x), "+v"(ymv[u].y), "+v"(ymv[u].z), "+v"(ymv[u].w));
      const int i = tid + 256 * u;
      if (i < TV * 8) *reinterpret_cast<f32x4*>(yt + (i >> 3) * 36 + 4 * (i & 7)) = ymv[u];
    }
    const int cg = n0 + cx;
    const float mu_ = p.bw_st[cg], is_ = p.bw_st[p.Cout + cg], a_ = p.bw_st[2 * p.Cout + cg], b_ = p.bw_st[3 * p.Cout + cg];
    __syncthreads();
    float s1 = 0.f, s2 = 0.f;
    for (int rr = part; rr < TV; rr += 8) {
      const float ym = yt[rr * 36 + cx];
      const float g = __builtin_fmaf(ym - mu_, a_, b_) > 0.f ? os[rr * 36 + cx] : 0.f;
      s1 += g;
      s2 += g * ((ym - mu_) * is_);
    }
    scr[(part * 32 + cx) * 2] = s1; scr[(part * 32 + cx) * 2 + 1] = s2;
    __syncthreads();
    if (tid < 32) {
      float a = 0.f, c = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) { a += scr[(j * 32 + tid) * 2]; c += scr[(j * 32 + tid) * 2 + 1]; }
      p.bw_rec[b * p.Cout + n0 + tid] = float2{a, c};
    }
  }
}

static int tconv_seq_launch(hipStream_t st, TconvSeqP& p, bool rec) {
  const int nk = p.Cin / 32;
  size_t fl = (size_t)128 * (p.Cin + 4) + 4 * p.Cin;
  if (fl < 2 * 128 * 36) fl = 2 * 128 * 36;
  size_t lds = (fl + 512) * sizeof(float);
  dim3 grid((unsigned)p.B, (unsigned)(p.Cout / 32));
  // a grid of at most one workgroup per CU whose workgroups are matrix-pipe bound (128 channels: 576 MFMAs per wave): ask for more
  // than half a CU's LDS so that the dispatcher cannot put two of them on one CU while another CU stays empty
  if (nk >= 4 && (long)grid.x * grid.y <= 256 && lds < 84 * 1024) lds = 84 * 1024;
#define TS_LAUNCH(NK_, REC_)                                                                                          \
  do {                                                                                                                \
    static size_t attr = 0;                                                                                           \
    if (lds > 64 * 1024 && lds > attr) {                                                                              \
      hipError_t e = hipFuncSetAttribute((const void*)tconv_seq_kernel<NK_, REC_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
      if (e != hipSuccess) return (int)e;                                                                             \
      attr = lds;                                                                                                     \
    }                                                                                                                 \
    hipLaunchKernelGGL((tconv_seq_kernel<NK_, REC_>), grid, dim3(256), lds, st, p);                                   \
  } while (0)
  if (rec) {
    if (nk == 1) TS_LAUNCH(1, true); else if (nk == 2) TS_LAUNCH(2, true); else if (nk == 4) TS_LAUNCH(4, true); else return MMEGO_EBADARG;
  } else {
    if (nk == 1) TS_LAUNCH(1, false); else if (nk == 2) TS_LAUNCH(2, false); else if (nk == 4) TS_LAUNCH(4, false); else return MMEGO_EBADARG;
  }
#undef TS_LAUNCH
  MMEGO_LAUNCH_CHECK();
  return MMEGO_OK;
}

// 1 when the sequence-tiled kernel takes the shape (else mmego_tconv_train / mmego_tconv_bwd_stats run the tile kernel)
extern "C" int mmego_tconv_seq_ok(int T, int V, int Cin, int Cout) {
  return T * V < 128 && (Cin == 32 || Cin == 64 || Cin == 128) && (Cout % 32) == 0 && Cout <= 256 ? 1 : 0;      // (and 9 taps)
}

// mmego_tconv_train / mmego_tconv_bwd_stats on the sequence-tiled kernel: Wf = the fragment-major pack (mmego_pack_multi kind 2; its
// second half for the input gradient); records per SEQUENCE: out_rec / bw_rec [B][Cout], T V rows per record.
extern "C" int mmego_tconv_seq_train(void* stream, const float* X, long ldx, const void* in_bn, const float* Wf, const float* bias, float* Y,
                                     long ldy, float* act, float* out_rec, int B, int T, int V, int Cin, int Cout, int taps) {
  const MmegoBnRefH* h = static_cast<const MmegoBnRefH*>(in_bn);
  MMEGO_REQUIRE(X && h && Wf && Y && out_rec && B > 0 && taps == 9 && mmego_tconv_seq_ok(T, V, Cin, Cout) && ldx >= Cin && ldy >= Cout);
  MMEGO_REQUIRE((ldx % 4) == 0 && (ldy % 4) == 0 && (((uintptr_t)X | (uintptr_t)Wf | (uintptr_t)Y | (uintptr_t)act) & 15) == 0);
  MMEGO_REQUIRE(h->rec && h->nrec >= 1 && h->rows_per_rec >= 1 && h->gamma && h->beta && (h->running_mean == nullptr) == (h->running_var == nullptr));
  TconvSeqP p = {X, ldx, Wf, bias, Y, ldy, act, B, T * V, V, Cin, Cout, taps};
  p.in_bn = bnref_device(h);
  p.out_rec = reinterpret_cast<float2*>(out_rec);
  p.bw_ym = nullptr; p.ldym = 0; p.bw_st = nullptr; p.bw_rec = nullptr;
  return tconv_seq_launch((hipStream_t)stream, p, true);
}

extern "C" int mmego_tconv_seq_bwd(void* stream, const float* dY, long lddy, const float* Wf, float* dAct, long ldda, const float* ymix,
                                   long ldym, const float* state, float* bw_rec, int B, int T, int V, int Cin, int Cout, int taps) {
  MMEGO_REQUIRE(dY && Wf && dAct && ymix && state && bw_rec && B > 0 && taps == 9 && mmego_tconv_seq_ok(T, V, Cin, Cout));
  MMEGO_REQUIRE(lddy >= Cin && ldda >= Cout && ldym >= Cout && (lddy % 4) == 0 && (ldda % 4) == 0 && (ldym % 4) == 0);
  MMEGO_REQUIRE((((uintptr_t)dY | (uintptr_t)Wf | (uintptr_t)dAct | (uintptr_t)ymix) & 15) == 0);
  TconvSeqP p = {dY, lddy, Wf, nullptr, dAct, ldda, nullptr, B, T * V, V, Cin, Cout, taps};
  p.in_bn = BnRefD{}; p.out_rec = nullptr;
  p.bw_ym = ymix; p.ldym = ldym; p.bw_st = state; p.bw_rec = reinterpret_cast<float2*>(bw_rec);
  return tconv_seq_launch((hipStream_t)stream, p, false);
}

// Every weight re-layout of a training step in ONE launch (host table of up to 8 entries):
//   kind 0  temporal-conv weight W[co][ci][tap] -> mode 2 of mmego_tconv_pack (forward pack, gradient pack behind it)
//   kind 1  a k=1 conv weight W[n][k] (n % 32 == 0, k % 32 == 0) -> FRAGMENT-MAJOR for gcn_front's 32x32x2 MFMAs:
//           Wp[((((n/32) * (K/32) + k/32) * 4 + (k%16)/4) * 64 + (k%32)/16 * 32 + n%32) * 4 + k%4]: a wave's operand fetch is four
//           coalesced 1-KB reads (fetched row by row, 32 rows 128-512 B apart per instruction, the same bytes took 5-6 us per kernel)
struct PackTab { const float* W[8]; float* Wp[8]; int Co[8], Ci[8], taps[8], kind[8], blk0[9]; int n; };
__global__ __launch_bounds__(256) void pack_multi_kernel(PackTab t) {
  int k = 0;
  while (k + 1 < t.n && (int)blockIdx.x >= t.blk0[k + 1]) ++k;
  const int Co = t.Co[k], Ci = t.Ci[k], taps = t.taps[k];
  const long total = (long)Co * Ci * taps;
  const long i = (long)(blockIdx.x - t.blk0[k]) * 256 + threadIdx.x;
  if (i >= total) return;
  const float w = t.W[k][i];
  if (t.kind[k] == 0) {
    const int tap = (int)(i % taps);
    const long q = i / taps;
    const int ci = (int)(q % Ci), co = (int)(q / Ci);
    t.Wp[k][((long)tap * Co + co) * Ci + ci] = w;
    t.Wp[k][total + ((long)(taps - 1 - tap) * Ci + ci) * Co + co] = w;
  } else if (t.kind[k] == 1) {
    const int n = (int)(i / Ci), kk = (int)(i - (long)n * Ci);
    const int ct = n >> 5, r = n & 31, kc = kk >> 5, k32 = kk & 31, h = k32 >> 4, j = (k32 & 15) >> 2, e = k32 & 3;
    t.Wp[k][((((long)ct * (Ci >> 5) + kc) * 4 + j) * 64 + h * 32 + r) * 4 + e] = w;
  } else {
    // kind 2: W[co][ci][tap] -> per tap the fragment-major image of the forward product's B operand (rows n = co, k = ci), and behind
    // all taps the same for the input gradient (taps reversed, rows n = ci, k = co): tconv_seq_kernel's operand fetches
    const int tap = (int)(i % taps);
    const long q = i / taps;
    const int ci = (int)(q % Ci), co = (int)(q / Ci);
    {
      const int ct = co >> 5, r = co & 31, kc = ci >> 5, k32 = ci & 31, h = k32 >> 4, j = (k32 & 15) >> 2, e = k32 & 3;
      t.Wp[k][(long)tap * Co * Ci + ((((long)ct * (Ci >> 5) + kc) * 4 + j) * 64 + h * 32 + r) * 4 + e] = w;
    }
    {
      const int ct = ci >> 5, r = ci & 31, kc = co >> 5, k32 = co & 31, h = k32 >> 4, j = (k32 & 15) >> 2, e = k32 & 3;
      t.Wp[k][total + (long)(taps - 1 - tap) * Co * Ci + ((((long)ct * (Co >> 5) + kc) * 4 + j) * 64 + h * 32 + r) * 4 + e] = w;
    }
  }
}

struct MmegoPackH { const float* W; float* Wp; int Co, Ci, taps, kind; };

extern "C" int mmego_pack_multi(void* stream, int n, const void* descs) {
  const MmegoPackH* h = static_cast<const MmegoPackH*>(descs);
  MMEGO_REQUIRE(h && n >= 1 && n <= 8);
  PackTab t;
  int blk = 0;
  for (int k = 0; k < n; ++k) {
    MMEGO_REQUIRE(h[k].W && h[k].Wp && h[k].Co >= 1 && h[k].Ci >= 1 && h[k].taps >= 1 && h[k].kind >= 0 && h[k].kind <= 2);
    MMEGO_REQUIRE(h[k].kind == 0 || ((h[k].Co % 32) == 0 && (h[k].Ci % 32) == 0));
    MMEGO_REQUIRE(h[k].kind != 1 || h[k].taps == 1);
    t.W[k] = h[k].W; t.Wp[k] = h[k].Wp; t.Co[k] = h[k].Co; t.Ci[k] = h[k].Ci; t.taps[k] = h[k].taps; t.kind[k] = h[k].kind; t.blk0[k] = blk;
    blk += (int)(((long)h[k].Co * h[k].Ci * h[k].taps + 255) / 256);
  }
  t.blk0[n] = blk; t.n = n;
  hipLaunchKernelGGL(pack_multi_kernel, dim3(blk), dim3(256), 0, (hipStream_t)stream, t);
  MMEGO_LAUNCH_CHECK();
  return MMEGO_OK;
}

// rows of one split of the weight-gradient product (a multiple of the chunk) and the number of splits: enough workgroups to cover the
// chip about once, at least 256 rows each
static void tconv_wgrad_split(long rows, int Cin, int Cout, int taps, long* rps, int* nsplit) {
  const long tiles = (long)taps * ((Cin + 63) / 64) * ((Cout + 63) / 64);
  long want = (576 + tiles - 1) / tiles;
  if (want < 1) want = 1;
  long per = (rows + want - 1) / want;
  if (per < 256) per = 256;
  per = (per + TW_K - 1) / TW_K * TW_K;
  *rps = per;
  *nsplit = (int)((rows + per - 1) / per);
}

extern "C" int mmego_tconv_wgrad_nsplit(int B, int T, int V, int Cin, int Cout, int taps) {
  long rps;
  int ns;
  tconv_wgrad_split((long)B * T * V, Cin, Cout, taps, &rps, &ns);
  return ns;
}

extern "C" int mmego_tconv_wgrad(void* stream, const float* dY, long lddy, const float* Xa, long ldx, float* ws, float* dW, int accumulate,
                                 int B, int T, int V, int Cin, int Cout, int taps) {
  MMEGO_REQUIRE((long)B * T * V * ldx < (1L << 30) && (long)B * T * V * lddy < (1L << 30));   // (32-bit element offsets)
  MMEGO_REQUIRE(dY && Xa && ws && dW && B > 0 && T > 0 && V > 0 && Cin >= 4 && Cout >= 4 && taps >= 1 && (taps & 1));
  MMEGO_REQUIRE((Cin % 4) == 0 && (Cout % 4) == 0 && (ldx % 4) == 0 && (lddy % 4) == 0 && ldx >= Cin && lddy >= Cout);
  MMEGO_REQUIRE((((uintptr_t)dY | (uintptr_t)Xa | (uintptr_t)ws) & 15) == 0);
  TconvWgP p = {dY, lddy, Xa, ldx, ws, (long)B * T * V, 0, T, V, Cin, Cout, taps};
  int ns;
  tconv_wgrad_split(p.rows, Cin, Cout, taps, &p.rows_per_split, &ns);
  hipStream_t st = (hipStream_t)stream;
  dim3 grid((unsigned)ns, (unsigned)(taps * ((Cin + 63) / 64) * ((Cout + 63) / 64)));
  const size_t lds = (size_t)2 * TW_K * TW_S * sizeof(float) + 2 * TW_K * sizeof(int);
  static bool attr = false;
  if (!attr) {
    hipError_t e = hipFuncSetAttribute((const void*)tconv_wgrad_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)tconv_wgrad_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    attr = true;
  }
  if (Cin <= 32 && Cout <= 32) hipLaunchKernelGGL(tconv_wgrad_kernel<true>, grid, dim3(256), lds, st, p);
  else hipLaunchKernelGGL(tconv_wgrad_kernel<false>, grid, dim3(256), lds, st, p);
  MMEGO_LAUNCH_CHECK();
  if (accumulate == 2) return MMEGO_OK;                    // deferred: the slabs stay in ws for mmego_slab_reduce (kind 1)
  long b = ((long)taps * Cin * Cout + 255) / 256;
  hipLaunchKernelGGL(tconv_wgrad_reduce_kernel, dim3((int)(b > 1024 ? 1024 : b)), dim3(256), 0, st, ws, ns, taps, Cout, Cin, dW, accumulate);
  MMEGO_LAUNCH_CHECK();
  return MMEGO_OK;
}
