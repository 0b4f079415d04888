// Fused training-step kernels of the ST-GCN inside Lower_Net's KeyEncoder (reference Net/GCN.py:67-147 st_gcn, :55-64
// ConvTemporalGraphical, :332-355 Model.extract_feature), channels-last rows (b, t, v).
//
// The reference's block is  1x1 conv -> einsum with A.importance -> BN -> ReLU -> 9x1 temporal conv -> BN, + residual(1x1 conv -> BN),
// ReLU.  Train-mode BatchNorm needs every row of its input before anything behind it can run, so the block has TWO grid-wide
// synchronisation points forward (behind the einsum, behind the temporal conv) and two backward -- those stay kernel boundaries.
// Everything else that used to be a launch of its own between them is folded into the producer or the consumer:
//   gcn_front      [finalize the previous block's two closing BatchNorms from partial records] -> relu(BN(tcn) + BN(residual)) applied
//                  while the 4-frame input tile is loaded (the activated rows are kept for backward) -> the stacked 1x1 conv of the graph
//                  convolution and the residual branch (one MFMA product) -> the einsum on 16x16x4 MFMAs straight from the product's
//                  LDS tile -> z, the residual pre-activation and the einsum output leave together with their BatchNorm partial records.
//                  Block 0 computes data_bn's batch statistics itself (the frame tensor is 92 KB); with mix = 0 the same kernel is the
//                  closing 1x1 conv `fcn` with its transposed store (the re-viewed output layout, Q8).
//                  Replaces per block: affine_act + product + graph_mix + colstats + 2 bn_finalize  (6 launches -> 1).
//   (tconv, gcn.hip: statistics of the BatchNorm in front finalized in its prologue, its own output's records in the epilogue; in
//   the backward pass the sums of the BatchNorm behind the einsum leave with the input gradient's epilogue)
//   gcn_bn_bwd_reduce / gcn_bn_bwd_apply   the closing BatchNorm pair's backward: reduce to <= 128 records, then finalize-in-prologue +
//                  apply in one launch (3 launches -> 2; d(gamma), d(beta) by workgroup 0)
//   graph_dA_fused finalize + apply of the BatchNorm behind the einsum while the frame's rows are loaded, then both gradients of
//                  the einsum (bn reduce + finalize + apply + graph_dA: 4 launches -> 1 with tconv's epilogue)
//   slab_reduce    every split-K / split-row partial product of a backward pass summed by ONE launch at its end (fixed order)
//   bn_param_grads data_bn's d(gamma), d(beta) alone (its input gradient is never used: the skeleton input is detached)
// All reductions have a fixed order: results are bit-identical from run to run and between the graph / eager engines.
#include <stdlib.h>
#include "gcn_stats.h"

#define GF_NT 512          // threads of gcn_front
#define GF_FPB 4           // frames per workgroup: <= 128 records per BatchNorm at the training shape (512 frames)
#define GF_ST 1024         // floats of statistics state in LDS: [2][4][128]

struct GcnFrontD {
  const float* X1; long ld1; const float* X2; long ld2;
  int in_mode;             // 0: data_bn over the frame tensor X1 [F][V*cin], statistics computed here; 1: relu(bn1(X1) + bn2(X2))
  BnRefD bn1, bn2;
  float* xact;             // [rows][cin]: the activated input (the previous block's output / the normalised skeleton), or null
  const float* W; const float* bias; int cin, nout;        // stacked weight [nout][cin]
  int mix, Kk, cout; const float* A; const float* imp;
  float* Z; long ldz;      // mix: [rows][(Kk + 1) cout] = (z | residual pre-activation)
  float* Y;                // mix: einsum output [rows][cout]
  float2* recY; float2* recR;
  float* outT; int T;      // mix = 0: product stored transposed, outT[b][nout][T*V]
  long F; int V;
};

// Rows [0, nrows) x ncols floats of an LDS tile (row stride S, rows 16-byte aligned) -> global rows of stride ld: all GF_NT threads,
// 16-byte pieces, a thread's LDS reads all ahead of its stores and no predicate on a store (indices past the end are clamped to the
// last piece: those lanes write the same value again) -- a store under a predicate waits for the store before it.
template <int NB>
__device__ __forceinline__ void tile_out(const float* lds, int S, float* g, long ld, int nrows, int ncols, int tid) {
  const int c4n = ncols >> 2, n4 = nrows * c4n;
  for (int i0 = 0; i0 < n4; i0 += GF_NT * NB) {
    f32x4 v[NB];
    int off[NB];
#pragma unroll
    for (int u = 0; u < NB; ++u) {
      int i = i0 + tid + GF_NT * u;
      i = i < n4 ? i : n4 - 1;
      const int row = i / c4n, c = (i - row * c4n) << 2;
      v[u] = *reinterpret_cast<const f32x4*>(lds + row * S + c);
      off[u] = row * 4096 + c;                           // (packed: row, column)
    }
#pragma unroll
    for (int u = 0; u < NB; ++u) *reinterpret_cast<f32x4*>(g + (long)(off[u] >> 12) * ld + (off[u] & 4095)) = v[u];
  }
}

// (mean, M2) records of `ncol` columns (a power of two <= 256) of an LDS tile over its nv valid rows: thread (column, row part) takes
// every P-th row (shifted sums, four loads in flight), the parts are added in a fixed order through `scr` (GF_NT * 2 floats).
// rows16: the tile keeps 16 rows per frame of V (row V a dummy).  Ends with the records stored (no trailing barrier).
__device__ __forceinline__ void tile_records(const float* base, int S, int ncol, int nv, int V, bool rows16, float* scr, float2* rec) {
  const int tid = threadIdx.x, P = GF_NT / ncol;
  const int cid = tid & (ncol - 1), part = tid / ncol;
  const float* col = base + cid;
  const float shift = col[0];
  float s1 = 0.f, s2 = 0.f;
  for (int rr = part; rr < nv; rr += 4 * P) {
    float d[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int q = rr + u * P < nv ? rr + u * P : 0;
      d[u] = col[(rows16 ? q + q / V : q) * S] - shift;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) { const float dd = rr + u * P < nv ? d[u] : 0.f; s1 += dd; s2 = __builtin_fmaf(dd, dd, s2); }
  }
  scr[(part * ncol + cid) * 2] = s1; scr[(part * ncol + cid) * 2 + 1] = s2;
  __syncthreads();
  if (tid < ncol) {
    float a = 0.f, b = 0.f;
    for (int j = 0; j < P; ++j) { a += scr[(j * ncol + tid) * 2]; b += scr[(j * ncol + tid) * 2 + 1]; }
    rec[tid] = rec_from_shifted(shift, a, b, nv);
  }
}

// NCTW: 32-column tiles of the product per wave pair (4 wave pairs x 2 row halves); NK: 32-k chunks (0: scalar product, cin < 32)
template <int NCTW, int NK>
__global__ __launch_bounds__(GF_NT) void gcn_front_kernel(GcnFrontD p) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int V = p.V, cin = p.cin, nout = p.nout, cout = p.cout;
  const long f0 = (long)blockIdx.x * GF_FPB;
  const int nf = (int)(p.F - f0 < GF_FPB ? p.F - f0 : GF_FPB);
  const int nv = nf * V;                                 // valid rows of this tile (<= 60)
  const long r0 = f0 * V, rows = p.F * V;
  const bool phased = NK > 0 && p.mix;                   // blocks with cin >= 32: the product in parts of cout columns (LDS <= 75 KB)
  const int XS = (NK ? cin : 4) + 4, ZS = phased ? cout + 16 : (p.mix ? nout + 16 : nout + 1), YS = cout + 4;
  float* st = sm;                                        // [2][4][cin]: mean, a, b, invstd of bn1 | bn2 (bn_from_records)
  float* zs = sm + GF_ST;                                // [64][ZS] product tile
  float* xs = zs + 64 * ZS;                              // [64][XS] input tile; later ys [60][YS]
  float* ys = xs;
  double* red = reinterpret_cast<double*>(xs);
  const bool first = blockIdx.x == 0;

  // ---- the product's weight fragments: requested first, they fly under the statistics prologue.  Lane (r, h) of a 32x32x2 MFMA
  // takes k = k0 + 16 h + s on BOTH operands (same permutation: the sum over k is unchanged); W comes FRAGMENT-MAJOR (mmego_pack_multi
  // kind 1), so the four 16-byte loads of a (tile, chunk) are coalesced 1-KB reads straight into the operand registers.
  const int r = lane & 31, h = lane >> 5, rt = wave & 1, wp = wave >> 1, NCT = nout / 32;
  // (phased: part i = partition k (i < Kk) or the residual branch (i = Kk), this wave's tile = columns i cout + 32 wp ..; else tile
  // wp + 4 i of the whole product)
  f32x4 wf[NCTW ? NCTW : 1][NK ? NK : 1][4];
  if (NK) {
#pragma unroll
    for (int i = 0; i < NCTW; ++i) {
      int ct = phased ? i * (cout / 32) + (wp < cout / 32 ? wp : 0) : wp + 4 * i;
      ct = ct < NCT ? ct : NCT - 1;
#pragma unroll
      for (int kc = 0; kc < NK; ++kc)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          wf[i][kc][j] = *reinterpret_cast<const f32x4*>(p.W + ((((long)ct * NK + kc) * 4 + j) * 64 + lane) * 4);
    }
  }
  // A . importance as MFMA operand registers: ae[k][s] = (A.imp)[k][v = 4 s + lane/16][w = lane%16]
  float ae[3][4];
  if (p.mix) {
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int v = 4 * s + (lane >> 4), w = lane & 15;
        const bool ok = k < p.Kk && v < V && w < V;
        const int idx = ok ? (k * V + v) * V + w : 0;
        const float a = p.A[idx], im = p.imp[idx];
        ae[k][s] = ok ? a * im : 0.f;
      }
  }

  // ---- statistics of the BatchNorm(s) in front
  if (p.in_mode == 0) {
    // data_bn: BatchNorm1d(V * cin) over the F frames -- every workgroup reads the whole frame tensor (F <= 1024 frames)
    const int C = V * cin, c = tid & 63, q = tid >> 6;       // 8 row groups x 64 channel lanes
    const int cc = c < C ? c : C - 1;
    const float shift = p.X1[cc];
    // this thread's element of the input tile and its piece of the (tiny) weight: requested now, used behind the statistics
    const int xi = tid < nv * cin ? tid : 0, xrow = xi / cin, xch = xi - xrow * cin, xfch = (xrow % V) * cin + xch;
    float xin = p.X1[(f0 + xrow / V) * C + xfch];
    float wst = p.W[tid < nout * cin ? tid : 0], bst = p.bias ? p.bias[tid < nout ? tid : 0] : 0.f;
    BnPre pre = bn_preload(p.bn1, C, 0, first);
    float s1 = 0.f, s2 = 0.f;
    for (long fb = q; fb < p.F; fb += 8 * 32) {
      float v[32];
#pragma unroll
      for (int u = 0; u < 32; ++u) {
        const long f = fb + 8 * u;
        v[u] = p.X1[(f < p.F ? f : p.F - 1) * C + cc];
      }
#pragma unroll
      for (int u = 0; u < 32; ++u) {
        asm volatile("" : "+v"(v[u]));
        const float d = fb + 8 * u < p.F ? v[u] - shift : 0.f;
        s1 += d; s2 += d * d;
      }
    }
    red[(q * 2 + 0) * 64 + c] = (double)s1; red[(q * 2 + 1) * 64 + c] = (double)s2;
    __syncthreads();
    asm volatile("" : "+v"(pre.g), "+v"(pre.b), "+v"(pre.rm), "+v"(pre.rv));
    if (tid < C) {
      double S1 = 0.0, S2 = 0.0;
      for (int g = 0; g < 8; ++g) { S1 += red[(g * 2 + 0) * 64 + tid]; S2 += red[(g * 2 + 1) * 64 + tid]; }
      const double N = (double)p.F, md = S1 / N;
      const double mean = (double)shift + md;
      double m2 = S2 - S1 * md;
      m2 = m2 > 0.0 ? m2 : 0.0;
      const double var = m2 / N;
      const float invstd = (float)(1.0 / sqrt(var + (double)p.bn1.eps));
      const float a = pre.g * invstd, b = pre.b;
      st[tid] = (float)mean; st[C + tid] = a; st[2 * C + tid] = b;
      if (first) {
        if (p.bn1.state) { p.bn1.state[tid] = (float)mean; p.bn1.state[C + tid] = invstd; p.bn1.state[2 * C + tid] = a; p.bn1.state[3 * C + tid] = b; }
        if (p.bn1.rmean) {
          p.bn1.rmean[tid] = (1.f - p.bn1.momentum) * pre.rm + p.bn1.momentum * (float)mean;
          const double unbiased = N > 1.0 ? m2 / (N - 1.0) : var;
          p.bn1.rvar[tid] = (1.f - p.bn1.momentum) * pre.rv + p.bn1.momentum * (float)unbiased;
        }
      }
    }
    __syncthreads();
    // input tile: x[(f, v)][c] = (X1[f][v cin + c] - mean) a + b   (nv * cin <= 512: one element per thread)
    asm volatile("" : "+v"(xin), "+v"(wst), "+v"(bst));
    if (tid < nv * cin) {
      const float x = __builtin_fmaf(xin - st[xfch], st[C + xfch], st[2 * C + xfch]);
      xs[xrow * XS + xch] = x;
      if (p.xact) p.xact[(r0 + xrow) * cin + xch] = x;
    }
    if (tid < nout * cin) st[256 + tid] = wst;            // W [nout][cin] and bias behind the statistics (nout * cin <= 512)
    if (tid < nout) st[768 + tid] = bst;
  } else {
    // the tile's loads first (clamped rows), the two finalizations while they fly
    const int c4n = cin / 4;                               // 16-byte pieces per row
    f32x4 v1[4], v2[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = tid + GF_NT * u, row = i / c4n, c4 = i - row * c4n;
      const int rc = row < nv ? row : nv - 1;
      const bool in = i < 64 * c4n;
      v1[u] = *reinterpret_cast<const f32x4*>(p.X1 + (r0 + rc) * p.ld1 + 4 * (in ? c4 : 0));
      v2[u] = *reinterpret_cast<const f32x4*>(p.X2 + (r0 + rc) * p.ld2 + 4 * (in ? c4 : 0));
    }
    const BnPre pre1 = bn_preload(p.bn1, cin, 0, first), pre2 = bn_preload(p.bn2, cin, 256, first);
    bn_gather2<GF_NT>(p.bn1, p.bn2, cin, rows, red, red + 2048);
    __syncthreads();
    bn_finish<GF_NT>(p.bn1, cin, red, st, first, 0, pre1);
    bn_finish<GF_NT>(p.bn2, cin, red + 2048, st + 4 * cin, first, 256, pre2);
    __syncthreads();
    const float *m1 = st, *a1 = st + cin, *b1 = st + 2 * cin, *m2 = st + 4 * cin, *a2 = st + 5 * cin, *b2 = st + 6 * cin;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = tid + GF_NT * u, row = i / c4n, c4 = i - row * c4n;
      asm volatile("" : "+v"(v1[u].x), "+v"(v1[u].y), "+v"(v1[u].z), "+v"(v1[u].w), "+v"(v2[u].x), "+v"(v2[u].y), "+v"(v2[u].z), "+v"(v2[u].w));
      if (i < 64 * c4n) {
        const int c = 4 * c4;
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float t = __builtin_fmaf(v1[u][e] - m1[c + e], a1[c + e], b1[c + e]);
          t += __builtin_fmaf(v2[u][e] - m2[c + e], a2[c + e], b2[c + e]);
          o[e] = row < nv ? fmaxf(t, 0.f) : 0.f;
        }
        *reinterpret_cast<f32x4*>(xs + row * XS + c) = o;
      }
    }
    if (p.xact) {
      __syncthreads();
      tile_out<2>(xs, XS, p.xact + r0 * cin, cin, nv, cin, tid);
    }
  }
  __syncthreads();

  if (phased) {
    // ---- blocks with cin >= 32: the stacked product one part of cout columns at a time (partition 0 .. Kk-1, then the residual
    // branch); a part's tile leaves for Z, and a partition's goes straight into the einsum (accumulated in registers over the
    // partitions) -- the whole (Kk + 1) cout wide tile never sits in LDS, so a recurrent-step workgroup of the other stage's IMU_Net
    // still fits on the CU beside this one
    const int nch = cout / 16, npair = nf * nch;
    const int vq = lane >> 4, cl = lane & 15;
    f32x4 accm[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int q = 0; q < NCTW; ++q) {
      if (q <= p.Kk) {                                     // (uniform)
        if (wp < cout / 32) {
          f32x16 acc = {0};
#pragma unroll
          for (int kc = 0; kc < NK; ++kc) {
            f32x4 af[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) af[j] = *reinterpret_cast<const f32x4*>(xs + (rt * 32 + r) * XS + kc * 32 + 16 * h + 4 * j);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
              for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[j][e], wf[q][kc][j][e], acc, 0, 0, 0);
          }
          const int col = wp * 32 + r;
          const float bb = p.bias ? p.bias[q * cout + col] : 0.f;
#pragma unroll
          for (int reg = 0; reg < 16; ++reg) zs[(rt * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h) * ZS + col] = acc[reg] + bb;
        }
        __syncthreads();
        tile_out<2>(zs, ZS, p.Z + r0 * p.ldz + q * cout, p.ldz, nv, cout, tid);
        if (q < p.Kk) {
          {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const int pi = wave + (GF_NT / 64) * i;
              if (pi < npair) {
                const int fi = pi / nch, n0 = (pi - fi * nch) * 16;
#pragma unroll
                for (int s_ = 0; s_ < 4; ++s_) {
                  const int v = 4 * s_ + vq;
                  const float b = zs[(fi * V + (v < V ? v : V - 1)) * ZS + n0 + cl];
                  const float a = q == 0 ? ae[0][s_] : (q == 1 ? ae[1][s_] : ae[2][s_]);
                  accm[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, accm[i], 0, 0, 0);
                }
              }
            }
          }
        } else {
          tile_records(zs, ZS, cout, nv, V, false, st, p.recR + (long)blockIdx.x * cout);
        }
        __syncthreads();
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int pi = wave + (GF_NT / 64) * i;
      if (pi < npair) {
        const int fi = pi / nch, n0 = (pi - fi * nch) * 16;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) ys[(fi * 16 + 4 * vq + reg) * YS + n0 + cl] = accm[i][reg];
      }
    }
    __syncthreads();
    for (int fi = 0; fi < nf; ++fi) tile_out<1>(ys + fi * 16 * YS, YS, p.Y + (r0 + fi * V) * cout, cout, V, cout, tid);
    tile_records(ys, YS, cout, nv, V, true, st, p.recY + (long)blockIdx.x * cout);
    return;
  }

  // ---- the stacked 1x1 conv: zs[64][nout] = xs . W^T + bias
  if (NK) {
#pragma unroll
    for (int i = 0; i < NCTW; ++i) {
      const int ct = wp + 4 * i;
      if (ct < NCT) {                     // (uniform per wave)
        f32x16 acc = {0};
#pragma unroll
        for (int kc = 0; kc < NK; ++kc) {
          f32x4 af[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) af[j] = *reinterpret_cast<const f32x4*>(xs + (rt * 32 + r) * XS + kc * 32 + 16 * h + 4 * j);
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[j][e], wf[i][kc][j][e], acc, 0, 0, 0);
        }
        const int col = ct * 32 + r;
        const float bb = p.bias ? p.bias[col] : 0.f;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
          const int row = rt * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
          const float v = acc[reg] + bb;
          zs[row * ZS + col] = v;
        }
      }
    }
  } else {
    for (int row = wave; row < nv; row += GF_NT / 64)
      for (int col = lane; col < nout; col += 64) {
        float v = st[768 + col];
        for (int c = 0; c < cin; ++c) v = __builtin_fmaf(xs[row * XS + c], st[256 + col * cin + c], v);
        zs[row * ZS + col] = v;
      }
  }
  __syncthreads();
  if (p.mix) tile_out<4>(zs, ZS, p.Z + r0 * p.ldz, p.ldz, nv, nout, tid);          // (z | residual pre-activation) leave together

  if (!p.mix) {
    // closing 1x1 conv: transposed store outT[b][col][t V + v] (consecutive threads walk a column's rows: contiguous addresses)
    const int TV = p.T * V, rr = lane;                     // (a wave walks 64 consecutive rows of one column per round)
    if (rr < nv) {
      const long f = f0 + rr / V;
      const long b = f / p.T;
      float* dst = p.outT + b * nout * TV + (int)(f - b * p.T) * V + rr % V;
      for (int col = wave; col < nout; col += GF_NT / 64) dst[(long)col * TV] = zs[rr * ZS + col];
    }
    return;
  }

  // ---- einsum('nkctv,kvw->nctw'): y_f[w][c] = sum_k sum_v (A.imp)[k][v][w] z_f[v][k cout + c] on 16x16x4 MFMAs, one (frame, 16-channel
  // chunk) pair per wave and round; B operand straight from the product's LDS tile
  {
    const int nch = cout / 16, npair = nf * nch;
    const int vq = lane >> 4, cl = lane & 15;
    for (int pi = wave; pi < npair; pi += GF_NT / 64) {
      const int fi = pi / nch, n0 = (pi - fi * nch) * 16;
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        if (k < p.Kk) {
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            const int v = 4 * s + vq;
            const float b = zs[(fi * V + (v < V ? v : V - 1)) * ZS + k * cout + n0 + cl];
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ae[k][s], b, acc, 0, 0, 0);
          }
        }
      }
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) ys[(fi * 16 + 4 * vq + reg) * YS + n0 + cl] = acc[reg];     // (16 rows per frame: row V is a dummy)
    }
  }
  __syncthreads();
  for (int fi = 0; fi < nf; ++fi) tile_out<1>(ys + fi * 16 * YS, YS, p.Y + (r0 + fi * V) * cout, cout, V, cout, tid);
  // ---- BatchNorm partial records of the einsum output and of the residual pre-activation over this tile's rows: thread (column,
  // row part) takes every P-th row (shifted sums, four loads in flight), the parts are added in a fixed order
  {
    const int ncol = 2 * cout, P = GF_NT / ncol;          // cout 32 / 64 / 128: 8 / 4 / 2 row parts
    const int cid = tid % ncol, part = tid / ncol;
    const bool res = cid >= cout;
    const float* col = res ? zs + p.Kk * cout + (cid - cout) : ys + cid;
    const int S = res ? ZS : YS;
    const float shift = col[0];
    float s1 = 0.f, s2 = 0.f;
    if (part < P) {
      for (int rr = part; rr < nv; rr += 4 * P) {
        float d[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {                      // (ys keeps 16 rows per frame, zs V)
          const int q = rr + u * P < nv ? rr + u * P : 0;
          d[u] = col[(res ? q : q + q / V) * S] - shift;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) { const float dd = rr + u * P < nv ? d[u] : 0.f; s1 += dd; s2 = __builtin_fmaf(dd, dd, s2); }
      }
      st[(part * ncol + cid) * 2] = s1; st[(part * ncol + cid) * 2 + 1] = s2;
    }
    __syncthreads();
    if (tid < ncol) {
      float a = 0.f, b = 0.f;
      for (int j = 0; j < P; ++j) { a += st[(j * ncol + tid) * 2]; b += st[(j * ncol + tid) * 2 + 1]; }
      (res ? p.recR : p.recY)[(long)blockIdx.x * cout + (res ? tid - cout : tid)] = rec_from_shifted(shift, a, b, nv);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Closing BatchNorm pair of a block, backward: out = relu(BN3(tz) + BNr(rz)); both BatchNorms see the same dY and mask.
// reduce: rec[j][cv] = (sum g, sum g xhat) over rows [64 j, 64 j + 64), virtual channel cv in [0, 2C): [0, C) = BN3, [C, 2C) = BNr.
struct GcnBnBwdD {
  const float* dY; long lddy; const float* mask; long ldm;
  const float* X1; long ld1; const float* st1;           // state [4][C]: mean, invstd, a, b
  const float* X2; long ld2; const float* st2;
  long rows; int C;
  float2* rec; int nrec;
  float* dg1; float* db1; float* dX1; long lddx1;
  float* dg2; float* db2; float* dX2; long lddx2;
};

__global__ __launch_bounds__(256) void gcn_bn_bwd_reduce_kernel(GcnBnBwdD p) {
  __shared__ float sh[4][64][2];
  const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
  const int cv = blockIdx.y * 64 + cx, C = p.C;
  const bool second = cv >= C;
  const int c = second ? cv - C : cv;
  const bool live = cv < 2 * C;
  const int cc = live ? c : 0;
  const float* X = second ? p.X2 : p.X1;
  const long ldx = second ? p.ld2 : p.ld1;
  const float* stt = second ? p.st2 : p.st1;
  const long rb = (long)blockIdx.x * 64;
  const float mu = stt[cc], is = stt[C + cc];
  float g[16], m[16], x[16];
#pragma unroll
  for (int u = 0; u < 16; ++u) {
    const long rr = rb + ry + 4 * u;
    const long rc = rr < p.rows ? rr : p.rows - 1;
    g[u] = p.dY[rc * p.lddy + cc];
    m[u] = p.mask[rc * p.ldm + cc];
    x[u] = X[rc * ldx + cc];
  }
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int u = 0; u < 16; ++u) {
    asm volatile("" : "+v"(g[u]), "+v"(m[u]), "+v"(x[u]));
    const float gg = (rb + ry + 4 * u < p.rows && m[u] > 0.f) ? g[u] : 0.f;
    s1 += gg;
    s2 += gg * ((x[u] - mu) * is);
  }
  sh[ry][cx][0] = s1; sh[ry][cx][1] = s2;
  __syncthreads();
  if (ry == 0 && live) {
    float a = 0.f, b = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) { a += sh[j][cx][0]; b += sh[j][cx][1]; }
    p.rec[(long)blockIdx.x * 2 * C + cv] = float2{a, b};
  }
}

// sums of (s1, s2) records rec[nrec][CT] per channel, all NT threads taking part, fp64, fixed order -> c12 [2][CT] in LDS
// (c1 = s1 / rows, c2 = s2 / rows); red: (NT / CP) * 2 * CP doubles.  The raw sums go to sums[2][CT] (LDS, float).
template <int NT>
__device__ __forceinline__ void bwd_gather(const float2* rec, int nrec, int CT, double* red) {
  int CP = 32;
  while (CP < CT) CP <<= 1;
  const int Q = NT / CP;
  const int tid = threadIdx.x, c = tid & (CP - 1), q = tid / CP;
  const int cc = c < CT ? c : CT - 1;
  double a = 0.0, b = 0.0;
  for (int j0 = q; j0 < nrec; j0 += 16 * Q) {
    float2 v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int j = j0 + u * Q;
      v[u] = rec[(long)(j < nrec ? j : nrec - 1) * CT + cc];
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      asm volatile("" : "+v"(v[u].x), "+v"(v[u].y));
      const bool ok = j0 + u * Q < nrec;
      a += ok ? (double)v[u].x : 0.0;
      b += ok ? (double)v[u].y : 0.0;
    }
  }
  red[(q * 2 + 0) * CP + c] = a; red[(q * 2 + 1) * CP + c] = b;
}
// (behind a barrier) c12[2][CT] = the sums / rows; the raw sums are returned to threads < CT (d(beta), d(gamma))
template <int NT>
__device__ __forceinline__ float2 bwd_finish(int CT, long rows, const double* red, float* c12) {
  int CP = 32;
  while (CP < CT) CP <<= 1;
  const int Q = NT / CP;
  const int tid = threadIdx.x;
  float2 out = {0.f, 0.f};
  if (tid < CT) {
    double s1 = 0.0, s2 = 0.0;
    for (int g = 0; g < Q; ++g) { s1 += red[(g * 2 + 0) * CP + tid]; s2 += red[(g * 2 + 1) * CP + tid]; }
    out = float2{(float)s1, (float)s2};
    c12[tid] = (float)(s1 / (double)rows); c12[CT + tid] = (float)(s2 / (double)rows);
  }
  return out;
}

// dX = a (g - c1 - xhat c2) for both BatchNorms of the pair; c1, c2 finalized in the prologue from the reduce kernel's records
#define GBA_NT 1024
__global__ __launch_bounds__(GBA_NT) void gcn_bn_bwd_apply_kernel(GcnBnBwdD p, long rows_per_wg) {
  __shared__ double red[GBA_NT / 32 * 2 * 32 > 4 * 2 * 256 ? GBA_NT / 32 * 2 * 32 : 4 * 2 * 256];
  __shared__ float c12[2 * 256], stl[2][3][128];
  const int tid = threadIdx.x, C = p.C, CT = 2 * C;
  const long rbeg = (long)blockIdx.x * rows_per_wg, rend = rbeg + rows_per_wg < p.rows ? rbeg + rows_per_wg : p.rows;
  // first round of this workgroup's loads, then the prologue
  const int c4n = C / 4;
  const long npiece = (rend - rbeg) * c4n;
  f32x4 g0 = {0.f, 0.f, 0.f, 0.f}, m0 = g0, x10 = g0, x20 = g0;
  {
    const long i = tid < npiece ? tid : (npiece > 0 ? npiece - 1 : 0);
    const long row = rbeg + i / c4n;
    const int c = 4 * (int)(i % c4n);
    if (npiece > 0) {
      g0 = *reinterpret_cast<const f32x4*>(p.dY + row * p.lddy + c);
      m0 = *reinterpret_cast<const f32x4*>(p.mask + row * p.ldm + c);
      x10 = *reinterpret_cast<const f32x4*>(p.X1 + row * p.ld1 + c);
      x20 = *reinterpret_cast<const f32x4*>(p.X2 + row * p.ld2 + c);
    }
  }
  // (the BatchNorm states ride in the same memory round trip as the records: registers now, LDS behind the gather)
  const int sc = tid < C ? tid : 0;
  float sv[6] = {p.st1[sc], p.st1[C + sc], p.st1[2 * C + sc], p.st2[sc], p.st2[C + sc], p.st2[2 * C + sc]};
  bwd_gather<GBA_NT>(p.rec, p.nrec, CT, red);
#pragma unroll
  for (int e = 0; e < 6; ++e) asm volatile("" : "+v"(sv[e]));
  if (tid < C) {
    stl[0][0][tid] = sv[0]; stl[0][1][tid] = sv[1]; stl[0][2][tid] = sv[2];
    stl[1][0][tid] = sv[3]; stl[1][1][tid] = sv[4]; stl[1][2][tid] = sv[5];
  }
  __syncthreads();
  const float2 tot = bwd_finish<GBA_NT>(CT, p.rows, red, c12);
  __syncthreads();
  for (long i0 = 0; i0 < npiece; i0 += GBA_NT) {
    const long i = i0 + tid;
    f32x4 g = g0, m = m0, x1 = x10, x2 = x20;
    if (i0 > 0 && i < npiece) {
      const long row = rbeg + i / c4n;
      const int c = 4 * (int)(i % c4n);
      g = *reinterpret_cast<const f32x4*>(p.dY + row * p.lddy + c);
      m = *reinterpret_cast<const f32x4*>(p.mask + row * p.ldm + c);
      x1 = *reinterpret_cast<const f32x4*>(p.X1 + row * p.ld1 + c);
      x2 = *reinterpret_cast<const f32x4*>(p.X2 + row * p.ld2 + c);
    }
    if (i < npiece) {
      const long row = rbeg + i / c4n;
      const int c = 4 * (int)(i % c4n);
      f32x4 o1, o2;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float gg = m[e] > 0.f ? g[e] : 0.f;
        const float xh1 = (x1[e] - stl[0][0][c + e]) * stl[0][1][c + e];
        const float xh2 = (x2[e] - stl[1][0][c + e]) * stl[1][1][c + e];
        o1[e] = stl[0][2][c + e] * (gg - c12[c + e] - xh1 * c12[CT + c + e]);
        o2[e] = stl[1][2][c + e] * (gg - c12[C + c + e] - xh2 * c12[CT + C + c + e]);
      }
      *reinterpret_cast<f32x4*>(p.dX1 + row * p.lddx1 + c) = o1;
      *reinterpret_cast<f32x4*>(p.dX2 + row * p.lddx2 + c) = o2;
    }
  }
  if (blockIdx.x == 0 && tid < CT) {                     // d(beta) = sum g, d(gamma) = sum g xhat (stored last: no load waits behind them)
    if (tid < C) { p.db1[tid] = tot.x; p.dg1[tid] = tot.y; }
    else { p.db2[tid - C] = tot.x; p.dg2[tid - C] = tot.y; }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Both gradients of the einsum (pool.hip graph_dA_partial_kernel's arithmetic) with the backward of the BatchNorm + ReLU behind the
// einsum applied while a frame's rows are loaded: dy[w][c] = a (g - c1 - xhat c2), g = dY0 . [bn(ymix) > 0], xhat from ymix; c1, c2
// finalized in the prologue from the records tconv's input-gradient epilogue left (one per 64-row tile).
#define GDF_FPB 2
struct GraphDAFusedD {
  const float* Z; long ldz; const float* dY0; const float* Ym; const float* st0;     // st0 [4][C]
  const float2* rec; int nrec; float* dg0; float* db0;
  long G; int V, Kk, C;
  float* partial; const float* A; const float* imp; float* dZ; long lddz;
};

// 256 threads; a wave takes the (frame, partition k) pairs of the workgroup's GDF_FPB frames in turn.  Per pair, on 16x16x4 MFMAs:
//   dz_k[v][c] = sum_w (A.imp)[k][v][w] dy[w][c]      A operand = (A.imp)_k (registers), B = the frame's dy rows (LDS)
//   dA_k[v][w] += sum_c z_k[v][c] dy[w][c]            A = z_k rows, B = dy rows, both from LDS, the channel axis as k
// dy is formed while the frame's rows are staged (the BatchNorm + ReLU backward); all of a frame's loads are in flight together and
// the next frame's are requested before the current one is multiplied; dz leaves through LDS in 16-byte pieces.
__global__ __launch_bounds__(256) void graph_dA_fused_kernel(GraphDAFusedD p) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int V = p.V, Kk = p.Kk, C = p.C, KC = Kk * C;
  const int ZS = KC + 4, YS = C + 4;                     // row strides: 16-byte aligned rows, lanes 16 apart in k hit other banks
  float* zs = sm;                                        // [16][ZS] z of the frame (row 15: zero), later dz
  float* ys = zs + 16 * ZS;                              // [16][YS] dy
  float* c12 = ys + 16 * YS;                             // [2][C]
  float* stl = c12 + 2 * C;                              // [4][C]
  float* dAs = stl + 4 * C;                              // [4 waves][256] partial dA tiles
  double* red = reinterpret_cast<double*>(dAs + 1024);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lo = lane & 15, hi = lane >> 4;
  // A . importance as the dz product's A operand: ae[k][s] = (A.imp)[k][v = lane%16][w = 4 s + lane/16]
  float ae[3][4];
#pragma unroll
  for (int k = 0; k < 3; ++k)
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int v = lo, w = 4 * s + hi;
      const bool ok = k < Kk && v < V && w < V;
      const int idx = ok ? (k * V + v) * V + w : 0;
      const float a = p.A[idx], im = p.imp[idx];
      ae[k][s] = ok ? a * im : 0.f;
    }
  float sv[2] = {p.st0[tid < 4 * C ? tid : 0], p.st0[tid + 256 < 4 * C ? tid + 256 : 0]};
  const long g0 = (long)blockIdx.x * GDF_FPB;
  const int nfr = (int)(p.G - g0 < GDF_FPB ? p.G - g0 : GDF_FPB);
  // frame loads: z [V][KC] in 16-byte pieces (<= 4 per thread), ym / dy0 [V][C] (<= 2 per thread each)
  const int zc4 = KC / 4, yc4 = C / 4, nz4 = V * zc4, ny4 = V * yc4;
  f32x4 zv[4], yv[2], dv[2];
#define GDF_ISSUE(g_)                                                                                     \
  do {                                                                                                    \
    _Pragma("unroll") for (int u = 0; u < 4; ++u) {                                                       \
      const int i = tid + 256 * u < nz4 ? tid + 256 * u : nz4 - 1;                                        \
      zv[u] = *reinterpret_cast<const f32x4*>(p.Z + ((g_) * V + i / zc4) * p.ldz + 4 * (i % zc4));        \
    }                                                                                                     \
    _Pragma("unroll") for (int u = 0; u < 2; ++u) {                                                       \
      const int i = tid + 256 * u < ny4 ? tid + 256 * u : ny4 - 1;                                        \
      yv[u] = *reinterpret_cast<const f32x4*>(p.Ym + (g_) * V * C + 4 * i);                               \
      dv[u] = *reinterpret_cast<const f32x4*>(p.dY0 + (g_) * V * C + 4 * i);                              \
    }                                                                                                     \
  } while (0)
  GDF_ISSUE(g0);
  bwd_gather<256>(p.rec, p.nrec, C, red);
  asm volatile("" : "+v"(sv[0]), "+v"(sv[1]));
  if (tid < 4 * C) stl[tid] = sv[0];
  if (tid + 256 < 4 * C) stl[tid + 256] = sv[1];
  for (int i = tid; i < ZS; i += 256) zs[15 * ZS + i] = 0.f;
  for (int i = tid; i < YS; i += 256) ys[15 * YS + i] = 0.f;
  __syncthreads();
  const float2 tot = bwd_finish<256>(C, p.G * V, red, c12);
  __syncthreads();
  f32x4 accA[3] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};     // this wave's dA_k tiles (k = wave, wave + 4, ...)
  for (int fi = 0; fi < nfr; ++fi) {
    const long g = g0 + fi;
    // stage: z as it is, dy = a (g - c1 - xhat c2) with g = dY0 . [bn(ym) > 0]
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      asm volatile("" : "+v"(zv[u].x), "+v"(zv[u].y), "+v"(zv[u].z), "+v"(zv[u].w));
      const int i = tid + 256 * u;
      if (i < nz4) *reinterpret_cast<f32x4*>(zs + (i / zc4) * ZS + 4 * (i % zc4)) = zv[u];
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      asm volatile("" : "+v"(yv[u].x), "+v"(yv[u].y), "+v"(yv[u].z), "+v"(yv[u].w), "+v"(dv[u].x), "+v"(dv[u].y), "+v"(dv[u].z), "+v"(dv[u].w));
      const int i = tid + 256 * u;
      if (i < ny4) {
        const int c = 4 * (i % yc4);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float ym = yv[u][e];
          const float gg = __builtin_fmaf(ym - stl[c + e], stl[2 * C + c + e], stl[3 * C + c + e]) > 0.f ? dv[u][e] : 0.f;
          const float xh = (ym - stl[c + e]) * stl[C + c + e];
          o[e] = stl[2 * C + c + e] * (gg - c12[c + e] - xh * c12[C + c + e]);
        }
        *reinterpret_cast<f32x4*>(ys + (i / yc4) * YS + c) = o;
      }
    }
    __syncthreads();
    if (fi + 1 < nfr) GDF_ISSUE(g + 1);                  // the next frame's rows fly under this frame's products
    // dA_k += z_k . dy^T over the channels (this wave's partitions), then dz_k = (A.imp)_k . dy
#pragma unroll
    for (int kk = 0; kk < 3; ++kk) {
      const int k = wave + 4 * kk;
      if (k < Kk) {                                       // (uniform per wave)
        for (int c0 = 0; c0 < C; c0 += 16) {
          f32x4 a4 = *reinterpret_cast<const f32x4*>(zs + lo * ZS + k * C + c0 + 4 * hi);
          f32x4 b4 = *reinterpret_cast<const f32x4*>(ys + lo * YS + c0 + 4 * hi);
#pragma unroll
          for (int e = 0; e < 4; ++e) accA[kk] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[e], b4[e], accA[kk], 0, 0, 0);
        }
      }
    }
    __syncthreads();                                      // (every wave has read z: the tile is overwritten with dz)
    for (int pi = wave; pi < Kk * (C / 16); pi += 4) {
      const int k = pi / (C / 16), n0 = (pi - k * (C / 16)) * 16;
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const float b = ys[(4 * s + hi) * YS + n0 + lo];
        const float a = k == 0 ? ae[0][s] : (k == 1 ? ae[1][s] : ae[2][s]);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
      }
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) zs[(4 * hi + reg) * ZS + k * C + n0 + lo] = acc[reg];
    }
    __syncthreads();
    {                                                     // dz rows out: 16-byte pieces, reads ahead of the stores
      f32x4 o[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = tid + 256 * u < nz4 ? tid + 256 * u : nz4 - 1;
        o[u] = *reinterpret_cast<const f32x4*>(zs + (i / zc4) * ZS + 4 * (i % zc4));
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = tid + 256 * u < nz4 ? tid + 256 * u : nz4 - 1;
        *reinterpret_cast<f32x4*>(p.dZ + (g * V + i / zc4) * p.lddz + 4 * (i % zc4)) = o[u];
      }
    }
    __syncthreads();                                      // (zs row 15 was overwritten with dz's dummy row: zero it again below)
    for (int i = tid; i < ZS; i += 256) zs[15 * ZS + i] = 0.f;
  }
#undef GDF_ISSUE
  // partial dA of this workgroup: D[i = v][j = w], lane (lo = w, hi): rows v = 4 hi + reg
#pragma unroll
  for (int kk = 0; kk < 3; ++kk) {
    const int k = wave + 4 * kk;
    if (k < Kk) {
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int v = 4 * hi + reg, w = lo;
        if (v < V && w < V) p.partial[(long)blockIdx.x * Kk * V * V + (k * V + v) * V + w] = accA[kk][reg];
      }
    }
  }
  if (blockIdx.x == 0 && tid < C) { p.db0[tid] = tot.x; p.dg0[tid] = tot.y; }
}

// ---------------------------------------------------------------------------------------------------------------------------
// One launch for every deferred partial-product sum of a backward pass.  Entry kinds:
//   0  split-K slabs ws[nsplit][M*N] -> C[m*scm + n] (the layout mmego_gemm leaves with accumulate = 2), optionally the slab row
//      sums behind them ws[nsplit*M*N + k*M + m] -> asum[m]
//   1  temporal-conv weight-gradient slabs ws[nsplit][tap][co][ci] -> dW[co][ci][tap]
//   2  column sums of partial[nsplit][n] scaled per column: out[i] = scale[i] * sum_k partial[k][i]  (edge-importance gradient)
struct SlabDesc { const float* ws; float* out; const float* scale; float* asum; int kind, nsplit, M, N, taps; long scm; int blk0; };
#define SLAB_MAX 24
struct SlabTable { SlabDesc d[SLAB_MAX]; int n; };

__global__ __launch_bounds__(256) void slab_reduce_kernel(SlabTable t) {
  __shared__ float sh[16][17];
  int di = 0;
#pragma unroll 1
  while (di + 1 < t.n && (int)blockIdx.x >= t.d[di + 1].blk0) ++di;
  const SlabDesc& d = t.d[di];
  const int o = threadIdx.x & 15, kg = threadIdx.x >> 4;
  const long total = (long)d.M * d.N;
  const long cblocks = (total + 15) / 16;
  const long blk = (long)blockIdx.x - d.blk0;
  const bool is_asum = blk >= cblocks;                    // (uniform per block; kind 0 only)
  const long i = (is_asum ? blk - cblocks : blk) * 16 + o;
  const long cnt = is_asum ? d.M : total;
  const float* src = is_asum ? d.ws + (long)d.nsplit * total : d.ws;
  float s = 0.f;
  if (i < cnt) {
    int k = kg;
#pragma unroll 1
    for (; k + 7 * 16 < d.nsplit; k += 8 * 16) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = src[(long)(k + u * 16) * cnt + i];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; k < d.nsplit; k += 16) s += src[(long)k * cnt + i];
  }
  sh[kg][o] = s;
  __syncthreads();
  if (kg == 0 && i < cnt) {
    s = sh[0][o];
#pragma unroll
    for (int g = 1; g < 16; ++g) s += sh[g][o];
    if (is_asum) { d.asum[i] = s; return; }
    if (d.kind == 0) {
      const long m = i / d.N;
      d.out[m * d.scm + (i - m * d.N)] = s;
    } else if (d.kind == 1) {                             // i = (tap * Co + co) * Ci + ci ;  M = taps * Co, N = Ci
      const int ci = (int)(i % d.N);
      const long q = i / d.N;
      const int Co = d.M / d.taps;
      const int co = (int)(q % Co), tap = (int)(q / Co);
      d.out[((long)co * d.N + ci) * d.taps + tap] = s;
    } else {
      d.out[i] = d.scale ? s * d.scale[i] : s;
    }
  }
}

// d(gamma)[c] = sum_r dY[r][c] xhat[r][c], d(beta)[c] = sum_r dY[r][c] for a BatchNorm whose input gradient nobody needs
// (rows <= 1024: one workgroup per 16 channels, 16 row lanes)
__global__ __launch_bounds__(256) void bn_param_grads_kernel(const float* __restrict__ dY, long lddy, const float* __restrict__ X, long ldx,
                                                             const float* __restrict__ st, long rows, int C, float* dgamma, float* dbeta) {
  __shared__ double sh[16][16][2];
  const int cx = threadIdx.x & 15, ry = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cx;
  const int cc = c < C ? c : C - 1;
  const float mu = st[cc], is = st[C + cc];
  double s1 = 0.0, s2 = 0.0;
  for (long rb = ry; rb < rows; rb += 16 * 8) {
    float g[8], x[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const long rr = rb + 16 * u;
      const long rc = rr < rows ? rr : rows - 1;
      g[u] = dY[rc * lddy + cc]; x[u] = X[rc * ldx + cc];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      asm volatile("" : "+v"(g[u]), "+v"(x[u]));
      const float gg = rb + 16 * u < rows ? g[u] : 0.f;
      s1 += (double)gg;
      s2 += (double)(gg * ((x[u] - mu) * is));
    }
  }
  sh[ry][cx][0] = s1; sh[ry][cx][1] = s2;
  __syncthreads();
  if (ry == 0 && c < C) {
    double a = 0.0, b = 0.0;
    for (int j = 0; j < 16; ++j) { a += sh[j][cx][0]; b += sh[j][cx][1]; }
    dbeta[c] = (float)a; dgamma[c] = (float)b;
  }
}

// =============================================================================================================================
// C ABI
// =============================================================================================================================
struct MmegoGcnFrontH {          // host-side mirror of include/mmego_hip.h's MmegoGcnFront (same field order)
  const float* X1; long ld1; const float* X2; long ld2; int in_mode;
  MmegoBnRefH bn1, bn2;
  float* xact;
  const float* W; const float* bias; int cin, nout;
  int mix, K, cout; const float* A; const float* importance;
  float* Z; long ldz; float* Y; float* recY; float* recR;
  float* outT; int T;
  long F; int V;
};

extern "C" int mmego_gcn_front_nrec(long F) { return cdiv(F, GF_FPB); }

extern "C" int mmego_gcn_front(void* stream, const void* desc) {
  const MmegoGcnFrontH* h = static_cast<const MmegoGcnFrontH*>(desc);
  MMEGO_REQUIRE(h && h->X1 && h->W && h->F > 0 && h->V >= 1 && h->V <= 15 && h->cin >= 1 && h->nout >= 32 && (h->nout % 32) == 0);
  MMEGO_REQUIRE(h->in_mode == 0 || h->in_mode == 1);
  GcnFrontD p;
  p.X1 = h->X1; p.ld1 = h->ld1; p.X2 = h->X2; p.ld2 = h->ld2; p.in_mode = h->in_mode;
  p.bn1 = bnref_device(&h->bn1); p.bn2 = bnref_device(&h->bn2);
  p.xact = h->xact; p.W = h->W; p.bias = h->bias; p.cin = h->cin; p.nout = h->nout;
  p.mix = h->mix; p.Kk = h->K; p.cout = h->cout; p.A = h->A; p.imp = h->importance;
  p.Z = h->Z; p.ldz = h->ldz; p.Y = h->Y; p.recY = reinterpret_cast<float2*>(h->recY); p.recR = reinterpret_cast<float2*>(h->recR);
  p.outT = h->outT; p.T = h->T; p.F = h->F; p.V = h->V;
  if (p.in_mode == 0) {
    MMEGO_REQUIRE(p.V * p.cin <= 64 && p.F <= 1024 && p.bn1.gamma && p.bn1.beta);
  } else {
    MMEGO_REQUIRE(p.X2 && (p.cin % 32) == 0 && p.cin <= 128 && (p.ld1 % 4) == 0 && (p.ld2 % 4) == 0);
    MMEGO_REQUIRE(p.bn1.rec && p.bn2.rec && p.bn1.nrec >= 1 && p.bn2.nrec >= 1 && p.bn1.rpr >= 1 && p.bn2.rpr >= 1 && p.bn1.gamma && p.bn2.gamma);
    MMEGO_REQUIRE((((uintptr_t)p.X1 | (uintptr_t)p.X2 | (uintptr_t)p.xact | (uintptr_t)p.W) & 15) == 0);
  }
  if (p.mix) {
    MMEGO_REQUIRE(p.Kk >= 1 && p.Kk <= 3 && p.cout >= 16 && (p.cout % 16) == 0 && p.cout <= 128 && p.nout == (p.Kk + 1) * p.cout);
    MMEGO_REQUIRE(p.A && p.imp && p.Z && p.Y && p.recY && p.recR && p.ldz >= p.nout);
  } else {
    MMEGO_REQUIRE(p.outT && p.T >= 1 && (p.F % p.T) == 0);
    p.cout = 4;                      // (unused; keeps the LDS carve-up small)
  }
  const bool scalar = p.cin < 32;
  MMEGO_REQUIRE(scalar == (p.in_mode == 0) || !scalar);
  const bool phased = !scalar && p.mix;
  const int NCT = p.nout / 32, nctw = phased ? p.Kk + 1 : (NCT + 3) / 4, nk = scalar ? 0 : p.cin / 32;
  const int XS = (nk ? p.cin : 4) + 4, ZS = phased ? p.cout + 16 : (p.mix ? p.nout + 16 : p.nout + 1), YS = p.cout + 4;
  const int tail = 64 * XS > 64 * YS ? 64 * XS : 64 * YS;
  const size_t lds = (size_t)(GF_ST + 64 * ZS + (tail > 8192 ? tail : 8192)) * sizeof(float);      // (>= 32 KB of prologue scratch)
  MMEGO_REQUIRE(lds <= 160 * 1024 && 8 * p.cin <= GF_ST);
  hipStream_t st = (hipStream_t)stream;
  dim3 grid((unsigned)cdiv(p.F, GF_FPB));
#define GF_LAUNCH(NCTW_, NK_)                                                                                         \
  do {                                                                                                                \
    static size_t attr = 0;                                                                                           \
    if (lds > 64 * 1024 && lds > attr) {                                                                              \
      hipError_t e = hipFuncSetAttribute((const void*)gcn_front_kernel<NCTW_, NK_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
      if (e != hipSuccess) return (int)e;                                                                             \
      attr = lds;                                                                                                     \
    }                                                                                                                 \
    hipLaunchKernelGGL((gcn_front_kernel<NCTW_, NK_>), grid, dim3(GF_NT), lds, st, p);                                \
  } while (0)
  if (nk == 0 && nctw == 1) GF_LAUNCH(1, 0);
  else if (nk == 1 && nctw == 3) GF_LAUNCH(3, 1);
  else if (nk == 2 && nctw == 3) GF_LAUNCH(3, 2);
  else if (nk == 4 && nctw == 1) GF_LAUNCH(1, 4);
  else if (nk == 2 && nctw == 1) GF_LAUNCH(1, 2);
  else if (nk == 1 && nctw == 2) GF_LAUNCH(2, 1);
  else if (nk == 2 && nctw == 2) GF_LAUNCH(2, 2);
  else if (nk == 1 && nctw == 4) GF_LAUNCH(4, 1);
  else if (nk == 2 && nctw == 4) GF_LAUNCH(4, 2);
  else return MMEGO_EBADARG;
#undef GF_LAUNCH
  MMEGO_LAUNCH_CHECK();
  return MMEGO_OK;
}

static int gcn_bn_bwd_fill(GcnBnBwdD& p, const float* dY, long lddy, const float* mask, long ldm, const float* X1, long ld1,
                           const float* st1, const float* X2, long ld2, const float* st2, long rows, int C, float* rec) {
  MMEGO_REQUIRE(dY && mask && X1 && st1 && X2 && st2 && rec && rows > 0 && C >= 4 && C <= 128 && (C % 4) == 0);
  p.dY = dY; p.lddy = lddy; p.mask = mask; p.ldm = ldm; p.X1 = X1; p.ld1 = ld1; p.st1 = st1; p.X2 = X2; p.ld2 = ld2; p.st2 = st2;
  p.rows = rows; p.C = C; p.rec = reinterpret_cast<float2*>(rec); p.nrec = cdiv(rows, 64);
  p.dg1 = p.db1 = p.dX1 = p.dg2 = p.db2 = p.dX2 = nullptr; p.lddx1 = p.lddx2 = 0;
  return MMEGO_OK;
}

extern "C" int mmego_gcn_bn_bwd_reduce(void* stream, const float* dY, long lddy, const float* mask, long ldm, const float* X1, long ld1,
                                       const float* st1, const float* X2, long ld2, const float* st2, long rows, int C, float* rec) {
  GcnBnBwdD p;
  int rc = gcn_bn_bwd_fill(p, dY, lddy, mask, ldm, X1, ld1, st1, X2, ld2, st2, rows, C, rec);
  if (rc) return rc;
  hipLaunchKernelGGL(gcn_bn_bwd_reduce_kernel, dim3(p.nrec, cdiv(2 * C, 64)), dim3(256), 0, (hipStream_t)stream, p);
  MMEGO_LAUNCH_CHECK();
  return MMEGO_OK;
}

extern "C" int mmego_gcn_bn_bwd_apply(void* stream, const float* dY, long lddy, const float* mask, long ldm, const float* X1, long ld1,
                                      const float* st1, const float* X2, long ld2, const float* st2, long rows, int C, const float* rec,
                                      float* dgamma1, float* dbeta1, float* dX1, long lddx1, float* dgamma2, float* dbeta2, float* dX2,
                                      long lddx2) {
  GcnBnBwdD p;
  int rc = gcn_bn_bwd_fill(p, dY, lddy, mask, ldm, X1, ld1, st1, X2, ld2, st2, rows, C, const_cast<float*>(rec));
  if (rc) return rc;
  MMEGO_REQUIRE(dgamma1 && dbeta1 && dX1 && dgamma2 && dbeta2 && dX2 && p.nrec <= 1024);
  MMEGO_REQUIRE((lddy % 4) == 0 && (ldm % 4) == 0 && (ld1 % 4) == 0 && (ld2 % 4) == 0 && (lddx1 % 4) == 0 && (lddx2 % 4) == 0);
  MMEGO_REQUIRE((((uintptr_t)dY | (uintptr_t)mask | (uintptr_t)X1 | (uintptr_t)X2 | (uintptr_t)dX1 | (uintptr_t)dX2) & 15) == 0);
  p.dg1 = dgamma1; p.db1 = dbeta1; p.dX1 = dX1; p.lddx1 = lddx1; p.dg2 = dgamma2; p.db2 = dbeta2; p.dX2 = dX2; p.lddx2 = lddx2;
  long nwg = cdiv(rows, 64);
  if (nwg > 128) nwg = 128;
  const long rpw = (rows + nwg - 1) / nwg;
  hipLaunchKernelGGL(gcn_bn_bwd_apply_kernel, dim3((unsigned)cdiv(rows, rpw)), dim3(GBA_NT), 0, (hipStream_t)stream, p, rpw);
  MMEGO_LAUNCH_CHECK();
  return MMEGO_OK;
}

extern "C" int mmego_graph_dA_fused_nblk(long G) { return cdiv(G, GDF_FPB); }

extern "C" int mmego_graph_dA_fused(void* stream, const float* Z, long ldz, const float* dY0, const float* Ymix, const float* st0,
                                    const float* rec, int nrec, float* dgamma0, float* dbeta0, long G, int V, int K, int C,
                                    float* partial_ws, const float* A, const float* imp, float* dZ, long lddz) {
  MMEGO_REQUIRE(Z && dY0 && Ymix && st0 && rec && nrec >= 1 && nrec <= 1024 && dgamma0 && dbeta0 && partial_ws && A && imp && dZ);
  MMEGO_REQUIRE(G > 0 && V > 0 && K > 0 && C > 0 && C <= 128 && K * V * V <= 512 && ldz >= (long)K * C && lddz >= (long)K * C);
  GraphDAFusedD p = {Z, ldz, dY0, Ymix, st0, reinterpret_cast<const float2*>(rec), nrec, dgamma0, dbeta0, G, V, K, C, partial_ws, A, imp, dZ, lddz};
  MMEGO_REQUIRE(V <= 15 && K <= 3 && (C % 16) == 0 && V * K * C / 4 <= 1024 && V * C / 4 <= 512 && (ldz % 4) == 0 && (lddz % 4) == 0);
  MMEGO_REQUIRE((((uintptr_t)Z | (uintptr_t)dY0 | (uintptr_t)Ymix | (uintptr_t)dZ) & 15) == 0);
  int CP = 32;
  while (CP < C) CP <<= 1;
  const size_t fl = (size_t)16 * (K * C + 4) + (size_t)16 * (C + 4) + 6 * (size_t)C + 1024;
  const size_t lds = fl * sizeof(float) + (size_t)(256 / CP) * 2 * CP * sizeof(double);
  MMEGO_REQUIRE(lds <= 64 * 1024);
  hipLaunchKernelGGL(graph_dA_fused_kernel, dim3((unsigned)cdiv(G, GDF_FPB)), dim3(256), lds, (hipStream_t)stream, p);
  MMEGO_LAUNCH_CHECK();
  return MMEGO_OK;
}

struct MmegoSlabH { const float* ws; float* out; const float* scale; float* asum; int kind, nsplit, M, N, taps; long scm; };

extern "C" int mmego_slab_reduce(void* stream, int n, const void* descs) {
  const MmegoSlabH* h = static_cast<const MmegoSlabH*>(descs);
  MMEGO_REQUIRE(h && n >= 1 && n <= SLAB_MAX);
  SlabTable t;
  t.n = n;
  int blk = 0;
  for (int i = 0; i < n; ++i) {
    MMEGO_REQUIRE(h[i].ws && h[i].out && h[i].nsplit >= 1 && h[i].M >= 1 && h[i].N >= 1 && h[i].kind >= 0 && h[i].kind <= 2);
    MMEGO_REQUIRE(h[i].kind != 1 || (h[i].taps >= 1 && (h[i].M % h[i].taps) == 0));
    MMEGO_REQUIRE(!h[i].asum || h[i].kind == 0);
    t.d[i].ws = h[i].ws; t.d[i].out = h[i].out; t.d[i].scale = h[i].scale; t.d[i].asum = h[i].asum;
    t.d[i].kind = h[i].kind; t.d[i].nsplit = h[i].nsplit; t.d[i].M = h[i].M; t.d[i].N = h[i].N; t.d[i].taps = h[i].taps;
    t.d[i].scm = h[i].scm; t.d[i].blk0 = blk;
    blk += (int)(((long)h[i].M * h[i].N + 15) / 16) + (h[i].asum ? (h[i].M + 15) / 16 : 0);
  }
  hipLaunchKernelGGL(slab_reduce_kernel, dim3(blk), dim3(256), 0, (hipStream_t)stream, t);
  MMEGO_LAUNCH_CHECK();
  return MMEGO_OK;
}

extern "C" int mmego_bn_param_grads(void* stream, const float* dY, long lddy, const float* X, long ldx, const float* state, long rows,
                                    int C, float* dgamma, float* dbeta) {
  MMEGO_REQUIRE(dY && X && state && dgamma && dbeta && rows > 0 && C > 0);
  hipLaunchKernelGGL(bn_param_grads_kernel, dim3(cdiv(C, 16)), dim3(256), 0, (hipStream_t)stream, dY, lddy, X, ldx, state, rows, C,
                     dgamma, dbeta);
  MMEGO_LAUNCH_CHECK();
  return MMEGO_OK;
}
