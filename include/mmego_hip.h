/* mmego_hip.h -- C ABI of libmmego_hip.so: the MI355X (gfx950) kernels behind the mmEgo hot path.
 *
 * The reference (yenanjing/mmEgo) has no FFI of its own: its boundary is the Python nn.Module
 * interface (Net/IMU_Net.py, Net/Upper_Net.py, Net/Lower_Net.py, Net/GCN.py) and the train_once bodies
 * of Processor/Train/Train_*.py.  This library sits one level below that boundary; each entry point
 * names the chain of aten ops of the reference it replaces.  INTEGRATION.md shows the ctypes stub a
 * maintainer of the reference would add.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer owned by the caller (the library never allocates or frees);
 *  - `stream` is a hipStream_t (pass the stream the caller's tensors are ordered on); all calls are
 *    asynchronous and capturable into a HIP graph (no allocation, no synchronisation inside);
 *  - tensors are fp32, row-major "rows x channels" (channels-last); `ld*` / `s*` are ELEMENT strides;
 *  - return value: 0 = ok, -1 = bad argument, >0 = hipError_t of the failed launch;
 *  - re-entrant per stream; workspaces are passed in by the caller.
 */
#ifndef MMEGO_HIP_H
#define MMEGO_HIP_H
#ifdef __cplusplus
extern "C" {
#endif

/* ---- dense products (gemm.hip) ---------------------------------------------------------------
 * C[b](m,n) (+)= sum_k A[b](m,k) * B[b](k,n) (+ bias[n]) (relu).  A(m,k) = A[m*sam + k*sak + b*sAb],
 * B(k,n) = B[k*sbk + n*sbn + b*sBb], C(m,n) = C[m*scm + n*scn + b*sCb].  nsplit > 1 splits K into slabs
 * in `splitk_ws` (nsplit*nbatch*M*N floats) that are then summed in a fixed order (deterministic).
 * Replaces: nn.Linear / Conv1d(k=1) / Conv2d(1x1, 9x1) / Conv3d forward and their weight/input
 * gradients -- Net/Upper_Net.py:242-301,343-364, Net/Lower_Net.py:40-72,95-123, Net/GCN.py:46-60,
 * and the LSTM input projections of Net/IMU_Net.py:58-62. */
/* (sBiasb: element stride of `bias` between batches, 0 = one bias for all -- lets the two directions' input projections of a
 * BiLSTM layer run as ONE batched product: same A, two weight / bias / output-column blocks.)
 * cmul (may be NULL; nsplit == 1 only): elementwise multiplier in C's layout applied after bias / relu / accumulate -- the
 * inter-layer dropout mask on an LSTM layer's input gradient.  relu == 2 (with cmul): no ReLU on the result; instead the result is
 * set to 0 where cmul <= 0 -- ReLU's backward on an input gradient, cmul = the layer's forward output (Upper_Net.py:350-351).
 * asum (may be NULL): asum[b*M + m] (+)= sum_k A[b](m,k), computed from the operand values the product loads anyway -- the bias
 * gradient beside a weight gradient dW = dY^T X (A = dY^T).  Only for the small-product shapes (about M*N*nsplit*nbatch <=
 * 2 M outputs, K slabs >= 64: bad argument otherwise); with nsplit > 1 splitk_ws needs nsplit*nbatch*M more floats. */
int mmego_gemm(void* stream, const float* A, long sam, long sak, const float* B, long sbk, long sbn, float* C,
               long scm, long scn, const float* bias, int M, int N, int K, int nbatch, long sAb, long sBb, long sCb,
               int relu, int accumulate, float* splitk_ws, int nsplit, long sBiasb, const float* cmul, float* asum);
/* Several INDEPENDENT products in one launch: each entry is one mmego_gemm argument set.  When all of them are small-tile products
 * of one operand orientation (what mmego_gemm would hand to its K-quartered kernel, unsplit), they share a launch -- grid.z is cut
 * into the products' ranges --; otherwise they run as n mmego_gemm calls in order.  Same results either way.  For the leaf
 * products of a backward pass (weight gradients of a BiLSTM stack: the reference's autograd computes them per parameter, e.g.
 * Net/Upper_Net.py:333), which otherwise sit one behind the other in a dependent chain of small kernels. */
typedef struct MmegoGemmDesc {
  const float* A; long sam, sak;
  const float* B; long sbk, sbn;
  float* C; long scm, scn;
  const float* bias;
  int M, N, K, nbatch;
  long sAb, sBb, sCb;
  int relu, accumulate;
  float* splitk_ws; int nsplit;
  long sBiasb;
  const float* cmul; float* asum;
} MmegoGemmDesc;
int mmego_gemm_group(void* stream, int n, const MmegoGemmDesc* descs);

/* ---- BatchNorm and row-wise helpers (bn.hip) ----------------------------------------------------
 * Train-mode statistics of X[rows, C] (+ running-stat update with torch semantics: momentum, unbiased
 * running var) -> mean, invstd, a = gamma*invstd, b = beta, so that y = (x-mean)*a + b.
 * partial_ws: 3*C*mmego_colstats_nblk(rows) floats.   Replaces native_batch_norm (train) of
 * Upper_Net.py:253-255,282-284, Lower_Net.py:51-53, GCN.py:106,117,136,310. */
int mmego_colstats_nblk(long rows);
int mmego_bn_train_stats(void* stream, const float* X, long ldx, long rows, int C, const float* gamma,
                         const float* beta, float* running_mean, float* running_var, float momentum, float eps,
                         float* partial_ws, float* mean, float* invstd, float* a, float* b);
/* The second half of mmego_bn_train_stats alone: partial = nblk records (n, mean, M2) per channel, [C][nblk][3], produced by a
 * kernel that had the tensor in its hands anyway (mmego_graph_mix's `stats`).  nblk <= 1024. */
int mmego_bn_finalize(void* stream, const float* partial, int nblk, int C, const float* gamma, const float* beta,
                      float* running_mean, float* running_var, float momentum, float eps, float* mean, float* invstd, float* a,
                      float* b);
/* The same for two tensors of one shape in the same two launches (st_gcn's BatchNorms of tcn(x) and residual(x), GCN.py:140-147;
 * the second rides along as channels [C, 2C); same results as two calls).  partial_ws: 6*C*nblk floats; C a multiple of the
 * 8..64-wide column tile. */
int mmego_bn_train_stats_pair(void* stream, long rows, int C, float* partial_ws,
                              const float* X1, long ldx1, const float* gamma1, const float* beta1, float* running_mean1,
                              float* running_var1, float momentum1, float eps1, float* mean1, float* invstd1, float* a1, float* b1,
                              const float* X2, long ldx2, const float* gamma2, const float* beta2, float* running_mean2,
                              float* running_var2, float momentum2, float eps2, float* mean2, float* invstd2, float* a2, float* b2);
/* Eval mode: the same four vectors from the running statistics. */
int mmego_bn_eval_affine(void* stream, int C, const float* gamma, const float* beta, const float* running_mean,
                         const float* running_var, float eps, float* mean, float* invstd, float* a, float* b);
/* Eval mode: fold BatchNorm(running statistics) into the preceding k=1 conv / Linear: Wf = s W, bf = (b - mean) s + beta with
 * s = gamma / sqrt(var + eps); the product then applies bias + ReLU in its epilogue (Upper_Net.py:253-255 etc. in eval). */
int mmego_bn_fold_linear(void* stream, const float* W, const float* b, int N, int K, const float* gamma, const float* beta,
                         const float* running_mean, const float* running_var, float eps, float* Wf, float* bf);
/* Eval-mode pointwise MLP: Y = relu(W3 relu(W2 relu(W1 x + b1) + b2) + b3) per row, one kernel, intermediates in LDS
 * (BasePointNet / GlobalPointNet of Upper_Net.py:242-301, Lower_Net.py:40-72 in eval mode).
 * bn == NULL: W, b are BN-folded already (mmego_bn_fold_linear).  Otherwise bn is a HOST array of 12 device pointers
 * {gamma, beta, running_mean, running_var} x 3 layers and W, b are the raw conv parameters (b may be NULL): the kernel folds
 * BatchNorm(eps) into them while staging the weights, with mmego_bn_fold_linear's arithmetic (same bits).
 * Cin, C1 <= 32; C2, C3 <= 64.  X, Y may be column slices (row strides ldx, ldy).
 * pre (0..4, <= Cin): the first `pre` input columns of every row are also written in FRONT of the row's outputs, Y[row * ldy - pre + c]
 * (Lower_Net.py:229-233's cat(xyz, features) written row by row in one launch); 0: nothing outside Y's columns is written. */
int mmego_mlp3_eval(void* stream, const float* X, long ldx, long rows, int Cin, const float* W1, const float* b1, int C1,
                    const float* W2, const float* b2, int C2, const float* W3, const float* b3, int C3, float* Y, long ldy,
                    const float* const* bn, float eps, int pre);
/* The same with the three stages' operands (folded weights, stage inputs) rounded to bf16 and fp32 accumulation (mlp3_bf16.hip; opt-in
 * precision mode of eval forwards).  Biases, ReLU and Y stay fp32. */
int mmego_mlp3_eval_bf16(void* stream, const float* X, long ldx, long rows, int Cin, const float* W1, const float* b1, int C1,
                         const float* W2, const float* b2, int C2, const float* W3, const float* b3, int C3, float* Y, long ldy,
                         const float* const* bn, float eps, int pre);
/* Y = act((X1-m1)*a1+b1 [+ (X2-m2)*a2+b2]) -- BN apply + ReLU, and the st_gcn "tcn(x)+residual" join
 * (GCN.py:140-147). */
int mmego_affine_act(void* stream, const float* X1, long ld1, const float* m1, const float* a1, const float* b1,
                     const float* X2, long ld2, const float* m2, const float* a2, const float* b2, float* Y, long ldy,
                     long rows, int C, int relu);
/* Y[:, :C] (+)= X[:, :C] with row strides: the torch.cat pieces of Upper_Net.py:266, Lower_Net.py:70,111,119. */
int mmego_copy2d(void* stream, const float* X, long ldx, float* Y, long ldy, long rows, int C, int accumulate);
/* Y[i,:] = X[idx[i],:] (rows of W floats, idx int64; indices outside [0,nsrc) give zero rows): minibatch assembly from
 * the HBM-resident dataset -- replaces the DataLoader collate + per-batch host->device copies of Train_Upper.py:140-150. */
int mmego_gather_rows(void* stream, const float* X, long nsrc, long W, const long long* idx, long nout, float* Y);
/* BatchNorm (train) backward through an optional ReLU mask (Ymask > 0): dgamma, dbeta, dX.
 * partial_ws: 2*C*nblk floats, c12_ws: 2*C floats. */
int mmego_bn_backward(void* stream, const float* dY, long lddy, const float* Ymask, long ldm, const float* X, long ldx,
                      const float* mean, const float* invstd, const float* a, long rows, int C, float* partial_ws,
                      float* c12_ws, float* dgamma, float* dbeta, float* dX, long lddx);
/* Two BatchNorms that share dY and the mask -- st_gcn's relu(BN(tcn(x)) + BN(residual(x))), GCN.py:140-147 -- in the same three
 * launches (the second one rides along as channels [C, 2C); results are those of two mmego_bn_backward calls, bit for bit).
 * partial_ws: 4*C*nblk floats, c12_ws: 4*C floats; C a multiple of the 8..64-wide column tile (8, 16, 32, 64, 128, ...). */
int mmego_bn_backward_pair(void* stream, const float* dY, long lddy, const float* Ymask, long ldm, long rows, int C,
                           float* partial_ws, float* c12_ws,
                           const float* X1, long ldx1, const float* mean1, const float* invstd1, const float* a1,
                           float* dgamma1, float* dbeta1, float* dX1, long lddx1,
                           const float* X2, long ldx2, const float* mean2, const float* invstd2, const float* a2,
                           float* dgamma2, float* dbeta2, float* dX2, long lddx2);
/* out[c] (+)= sum_r X[r,c]: bias gradients; out2 (may be NULL) receives a copy (nn.LSTM's bias_ih / bias_hh share
 * one gradient).  partial_ws: C*nblk floats.  scale (may be NULL; rows <= 1024): the sum is multiplied by scale[c] first (the
 * edge-importance gradient = A . sum of the per-workgroup partials of mmego_graph_dA, GCN.py:62). */
int mmego_colsum(void* stream, const float* X, long ldx, long rows, int C, float* partial_ws, float* out, float* out2,
                 int accumulate, const float* scale);
/* The same for X [rows <= 1024, 2C] in ONE launch: columns [0, C) -> outA (and outA2), columns [C, 2C) -> outB (and outB2): the
 * four bias gradients of a BiLSTM layer (both directions x bias_ih / bias_hh) from its gate gradients.  C % 16 == 0. */
int mmego_colsum_pair(void* stream, const float* X, long ldx, long rows, int C, float* outA, float* outA2, float* outB,
                      float* outB2, int accumulate);
/* G = 0 where H <= 0 (ReLU backward for Linear+ReLU pairs, Upper_Net.py:350-351). */
int mmego_relu_mask(void* stream, float* G, long ldg, const float* H, long ldh, long rows, int C);
int mmego_fill(void* stream, float* X, long n, float v);

/* ---- LSTM (lstm.hip) ------------------------------------------------------------------------------
 * One timestep of a (bi)LSTM, any H % 32 == 0: gates = xproj + hprev . W_hh^T, fused cell update, c in
 * place.  Replaces the recurrent half of nn.LSTM for IMU_Net (Net/IMU_Net.py:58-62,77,82); xproj is the
 * input projection (incl. b_ih) produced by mmego_gemm, b_hh is added here.  first != 0: h_{t-1} = c_{t-1} = 0
 * (hprev may be NULL, the product is skipped).  Row strides: hps, xs, hos.  gst_d [Bn][4H] / cst_d [Bn][H]: optional
 * stash of this step's gate activations and cell state for mmego_lstm_cell_backward (training). */
int mmego_lstm_step(void* stream, int ndir, int Bn, int H, int first, const float* hprev0, const float* hprev1,
                    long hps, const float* whh0, const float* whh1, const float* bhh0, const float* bhh1,
                    const float* xproj0, const float* xproj1, long xs, float* hout0, float* hout1, long hos, float* c0,
                    float* c1, float* gst0, float* gst1, float* cst0, float* cst1);
/* Whole-sequence H=64 bidirectional LSTM layer (Upper_Net.py:333, Lower_Net.py:91, Upper_Net.py:210).
 * xproj_d rows are (b*T+t) with row stride xs; out rows (b*T+t) with row stride os, direction d in
 * columns [64d, 64d+64).  Optional stashes for backward (all or none): gates_d [T][B][64][4] (the four gate activations
 * i, f, g, o of a hidden unit side by side: a private format between this call and mmego_lstm64_backward, 16-byte aligned),
 * cst_d [T][B][64], hprev_d [(b*T+t)][64].
 * drop_y / drop_mask (both or neither; out's layout): nn.LSTM(dropout=drop_p)'s inverted inter-layer dropout applied while the
 * outputs are stored -- drop_mask = 0 or 1/(1-p) per element from a counter-based hash of (element index, seed_ctr[0], salt),
 * drop_y = out * drop_mask.  seed_ctr is only read; mmego_inc_i64 advances it once per training forward, salt tells the
 * call sites of one forward apart. */
int mmego_lstm64_forward(void* stream, int B, int T, const float* xproj0, const float* xproj1, long xs,
                         const float* whh0, const float* whh1, const float* bhh0, const float* bhh1, const float* h0_0,
                         const float* h0_1, const float* c0_0, const float* c0_1, float* out, long os, float* hn0,
                         float* hn1, float* cn0, float* cn1,
                         float* gates0, float* gates1, float* cst0, float* cst1, float* hprev0, float* hprev1,
                         float* drop_y, float* drop_mask, float drop_p, const unsigned long long* seed_ctr, int salt);
/* Backward through time of the same layer: dgates_d rows (b*T+t), row stride dgs (pre-activation
 * gradients; weight/input gradients follow as mmego_gemm products). */
int mmego_lstm64_backward(void* stream, int B, int T, const float* dout, long dos, const float* gates0,
                          const float* gates1, const float* cst0, const float* cst1, const float* c0_0,
                          const float* c0_1, const float* whh0, const float* whh1, float* dgates0, float* dgates1,
                          long dgs);

/* The two calls above for n <= 4 INDEPENDENT stacks' layers in ONE launch (grid z = stack; descs: n host structs whose fields are the
 * argument lists above, in order).  All stacks share B and T and the options (stashes for all or none, dropout for all or none).
 * UpperNetwlocal's global and anchor BiLSTM(64) stacks (Net/Upper_Net.py:333-339 and :208-216: same shape, no data in common) run
 * layer by layer side by side this way. */
typedef struct MmegoLstm64Fwd {
  const float* xproj[2]; long xs;
  const float* whh[2]; const float* bhh[2]; const float* h0[2]; const float* c0[2];
  float* out; long os;
  float* hn[2]; float* cn[2];
  float* gates[2]; float* cst[2]; float* hprev[2];
  int B, T;
  float* drop_y; float* drop_mask; float drop_p; const unsigned long long* seed_ctr; unsigned salt;
} MmegoLstm64Fwd;
typedef struct MmegoLstm64Bwd {
  const float* dout; long dos;
  const float* gates[2]; const float* cst[2]; const float* c0[2]; const float* whh[2];
  float* dgates[2]; long dgs;
  int B, T;
} MmegoLstm64Bwd;
int mmego_lstm64_forward_multi(void* stream, int n, const void* descs);
int mmego_lstm64_backward_multi(void* stream, int n, const void* descs);

/* ---- bf16-operand / fp32-accumulate forward of the frozen IMU_Net (bf16.hip; BASELINE config 5) -------------------
 * Opt-in precision mode, never the parity path.  bf16 values cross the ABI as raw bits in unsigned short.
 * Y[r, 0:cols] = bf16(X[r, 0:cols]) (round to nearest even); cols, ldx, ldy multiples of 4. */
int mmego_cvt_bf16(void* stream, const float* X, long ldx, long rows, long cols, unsigned short* Y, long ldy);
/* Same with a row permutation: rows (b*T + t) of X -> rows (t*Bp + b) of Y (Bp >= Bn, pad rows untouched): the time-major
 * operand of a BiLSTM layer's input projection. */
int mmego_cvt_bf16_tm(void* stream, const float* X, long ldx, int Bn, int T, long cols, unsigned short* Y, int Bp);
/* C[M,N] = A[M,K] . W[N,K]^T + bias[N] (relu), operands bf16, products and sums fp32.  Any of three outputs (NULL = skip):
 * C (fp32 row-major), Cb (bf16 row-major), Cf (fp32 TILE-MAJOR, M % 32 == 0 and N % 32 == 0): element (m, n) at
 *     ((m/32)*(N/32) + n/32)*1024 + ((m%32)/8)*256 + (n%32 + 32*(((m%32)/4)&1))*4 + m%4,
 * i.e. every 32x32 tile stored as the MFMA accumulator holds it (coalesced 16 B per lane for producer and consumer).
 * K % 64 == 0.  Replaces the BiLSTM input projections of Net/IMU_Net.py:58-62 in bf16 mode. */
int mmego_gemm_bf16(void* stream, const unsigned short* A, long lda, const unsigned short* W, long ldw, float* C,
                    long ldc, unsigned short* Cb, long ldcb, float* Cf, const float* bias, int M, int N, int K, int relu);
/* One (bi)LSTM timestep as mmego_lstm_step, with h_{t-1} and W_hh in bf16: gates = xproj + h_{t-1} . W_hh^T (xproj
 * already holds b_ih + b_hh), cell update in fp32, c in place.  H % 64 == 0.
 * The recurrent operands are FRAGMENT-MAJOR (the order the MFMA lanes read them, so a wave's operand load is one
 * coalesced 1-KB read): element (r, k) of an [R, H] matrix sits at
 *     (((r/32)*(H/16) + k/16)*64 + ((k/8)&1)*32 + r%32)*8 + k%8        (R padded to a multiple of 32 rows).
 * hprev_d: h_{t-1} in that layout (R = Bn); whh_d: W_hh in that layout with its rows reordered to
 * [hidden block jb][gate n][32 units] (row 128*jb + 32*n + jj = W_hh[n*H + 32*jb + jj]).
 * xpf: the layer's input projection, the tile-major Cf of mmego_gemm_bf16 with M = T*Bp rows ordered t*Bp + b
 * (Bp = Bn rounded up to 32) and N = 8H columns ordered d*4H + gate*H + j; mt0_d = t_d * Bp/32 selects direction d's
 * timestep.
 * h_t is written three times: hout (fp32, row stride hos), houtb (bf16 row-major, row stride hbs, may be NULL: the next
 * layer's projection operand) and hfrag_d (bf16 fragment-major: the next step's hprev_d; must not alias hprev_d). */
int mmego_lstm_step_bf16(void* stream, int ndir, int Bn, int H, int first, const unsigned short* hprev0,
                         const unsigned short* hprev1, const unsigned short* whh0, const unsigned short* whh1,
                         const float* xpf, long mt0_0, long mt0_1, float* hout0, float* hout1,
                         long hos, unsigned short* houtb0, unsigned short* houtb1, long hbs,
                         unsigned short* hfrag0, unsigned short* hfrag1, float* c0, float* c1);

/* Large batches: the input projection folded into the step -- gates = [x_t | h_{t-1}] . [W_ih | W_hh]^T + bias in ONE fp32
 * accumulation, no projection tensor in HBM.  Up to two input segments (layer 0: x_t, K0 = In; upper layers: the previous
 * layer's forward and backward h_t, K0 = K1 = H) plus h_{t-1} (skipped when first != 0); every operand fragment-major as in
 * mmego_lstm_step_bf16 (a*_d: [Bp x K] activations of direction d's timestep; w*_d: the matching column block of W_ih with rows
 * reordered [hidden block][gate][32 units]); bias = [2][4H] b_ih + b_hh.  K0, K1, H multiples of 64.  hout_d (fp32, may be NULL
 * for layers whose output only feeds the next layer) and hfrag_d (fragment-major bf16, always) receive h_t. */
int mmego_lstm_step_bf16_fused(void* stream, int ndir, int Bn, int H, int first, int nseg,
                               const unsigned short* a0_0, const unsigned short* a0_1, const unsigned short* w0_0,
                               const unsigned short* w0_1, int K0, const unsigned short* a1_0,
                               const unsigned short* a1_1, const unsigned short* w1_0, const unsigned short* w1_1,
                               int K1, const unsigned short* hprev0, const unsigned short* hprev1,
                               const unsigned short* whh0, const unsigned short* whh1, const float* bias,
                               float* hout0, float* hout1, long hos, unsigned short* hfrag0, unsigned short* hfrag1,
                               float* c0, float* c1);
/* Attention pooling over the T timesteps of a layer run by mmego_lstm_step_bf16_fused (IMU_Net.py:77-81: softmax over t of
 * w . h_t + b, weighted sum), read from the steps' own fragment-major bf16 h_t: hf = [T][2 directions][Bp x H] (the hfrag outputs,
 * direction 0 then 1 per timestep).  The layer's step launches then need no fp32 output (hout NULL).  vec [Bn][2H] fp32 row-major,
 * attn [Bn][T] or NULL; H in {128, 256, 512} (mmego_attn_pool_frag_bf16_ok), Bp % 32 == 0. */
int mmego_attn_pool_frag_bf16_ok(int H);
int mmego_attn_pool_frag_bf16(void* stream, const unsigned short* hf, int T, int Bp, int Bn, int H, const float* w, const float* b,
                              float* vec, float* attn);
/* rows (b*T + t) of X[., C] -> T fragment-major [Bp x C] bf16 matrices (timestep t at offset t*Bp*C): a layer-0 input of
 * mmego_lstm_step_bf16_fused.  C % 16 == 0, Bp % 32 == 0. */
int mmego_cvt_bf16_frag_tm(void* stream, const float* X, long ldx, int Bn, int T, int C, unsigned short* Y, int Bp);
/* Linear(Cin <= 16, H) (+ ReLU) straight into that layout: Y[t] = bf16(act(X[b*T + t] . W^T + bias)) for rows (b*T + t) of X[., Cin]
 * (IMU_Net's fc1, Net/IMU_Net.py:53,73, as the layer-0 operand of the fused bf16 step: the fp32 activation is never stored).
 * W [H][Cin] row-major, bias [H] or NULL; H % 16 == 0, Bp % 32 == 0; rows b >= Bn are written as zeros. */
int mmego_fc_relu_bf16_frag_tm(void* stream, const float* X, long ldx, const float* W, const float* bias, int Bn, int T, int Cin,
                               int H, unsigned short* Y, int Bp, int relu);

/* ---- IMU_Net stage-1 training pieces (imu_train.hip): reference Processor/Train/Train_IMU.py:21-34,114-149 -------
 * Pointwise LSTM cell backward of one timestep, both directions: dh = dout + dh_rec (dh_rec may be NULL), reads the
 * stashed gates / cell states, writes the pre-activation gate gradients dgates_d [Bn][4H] (row stride dgs) and updates
 * dc_d [Bn][H] in place (zero it before the last timestep).  cprev_d NULL means c_{t-1} = 0. */
int mmego_lstm_cell_backward(void* stream, int ndir, int Bn, int H, const float* dout0, const float* dout1, long dos,
                             const float* dhrec0, const float* dhrec1, const float* gst0, const float* gst1,
                             const float* cst0, const float* cst1, const float* cprev0, const float* cprev1, float* dc0,
                             float* dc1, float* dgates0, float* dgates1, long dgs);
/* Stage-1 loss: sum acos(clamp((tr(R Rgt^T)-1)/2, +-(1-1e-7))) * 180/3.14159265358 + 100 * sum |t - head|_2 and its
 * gradients wrt R [F,3,3] and t [F,3] (scaled by `scale`). */
int mmego_imu_loss(void* stream, const float* R, const float* t, const float* R_gt, const float* head_gt, long F,
                   float scale, float* loss, float* dR, float* dt);
/* One step of a BiLSTM layer's backward recurrence, both directions: dh_rec = dgates_s [Bn][4H] (row stride dgs) . W_hh (given
 * transposed, wT [H][4H]) with the cell backward of the next step of the backward pass applied on the product's tiles (same
 * arithmetic as mmego_lstm_cell_backward: dh = dout + dh_rec; gst / cst / cprev: that step's forward stashes, cprev NULL at the
 * sequence start; dc updated in place; dgo: that step's gate gradients, row stride dgs).  Replaces product + cell-backward
 * launches of the reference's nn.LSTM autograd (Train_IMU.py:114-149).  Bn, H multiples of 32. */
int mmego_lstm_bwd_step(void* stream, int Bn, int H, const float* dg0, const float* dg1, long dgs, const float* wT0,
                        const float* wT1, const float* dout0, const float* dout1, long dos, const float* gst0,
                        const float* gst1, const float* cst0, const float* cst1, const float* cprev0, const float* cprev1,
                        float* dc0, float* dc1, float* dgo0, float* dgo1);
/* Backward of mmego_imu_head: (dR [F,3,3], dt [F,3]) -> dy [F,9]. */
int mmego_imu_head_backward(void* stream, const float* y, const float* dR, const float* dt, long F, float* dy);

/* ---- geometry, heads, loss, selection (geom.hip) --------------------------------------------------
 * In-place xyz <- R (xyz - t) per frame (Utils.py:284-292, quirk Q1: the caller's buffer is mutated).
 * src (may be NULL): the points are read from src instead of pts (copy + transform, one launch); src_ld = 3: src holds xyz only,
 * src_ld >= C: whole rows, the channels behind xyz are copied too.  keep (may be NULL, [F*P][C]) and feats (may be NULL: the first
 * nfeat channels, row stride ldf) receive the transformed rows as well (Upper_Net.py:392-395 makes those copies right behind the
 * transform).  C <= 8 when any row copy is asked for. */
int mmego_transform2h(void* stream, float* pts, long F, int P, int C, const float* R, const float* t, const float* src,
                      long src_ld, float* keep, float* feats, long ldf, int nfeat);
/* out = R^T in + t (transpose=1, add_t=1: Utils.py:274-281) or out = R in (its backward). */
int mmego_rotate_points(void* stream, const float* in, float* out, long F, int P, const float* R, const float* t,
                        int transpose, int add_t);
/* which=0: y[F,87] -> q[F,14,3,3], joints[F,15,3] (Upper_Net.py:122-144,354-364);
 * which=1: y[F,42] -> q[F,6,3,3], joints[F,8,3] (Lower_Net.py:12-37,125-136).  body [B,20,3]; frame n
 * uses body row n % B (quirk Q2).  Joints are in the head frame.
 * Rw [F,3,3], tw [F,3], world [F,nslots,3] (all three or none): world = Rw^T joint + tw, the head-to-world transform that follows
 * the kinematics (Upper_Net.py:362-364, Lower_Net.py:225-227; mmego_rotate_points' arithmetic) from the same launch.
 * counters / ncount / seed_ctr (0 / NULL: none): the net's once-per-training-forward tick (mmego_inc_i64) on the same launch --
 * this kernel runs once per forward, behind every reader of the dropout seed.
 * backward: Rw given -> dj is the gradient wrt the WORLD-frame joints (the head-frame gradient Rw dj is formed first). */
int mmego_head_fk_forward(void* stream, int which, const float* y, const float* body, int B, long F, float* q,
                          float* joints, const float* Rw, const float* tw, float* world, long long* counters, int ncount,
                          unsigned long long* seed_ctr);
int mmego_head_fk_backward(void* stream, int which, const float* y, const float* body, int B, long F, const float* dj,
                           float* dy, const float* Rw);
/* mmego_head_fk_forward -> mmego_l1_loss (scale, loss[2], gradient = sign) -> mmego_head_fk_backward as ONE launch.  The loss is a
 * fixed-order sum: per-workgroup partial pairs in scratch (2 * ceil(F/64) + 1 doubles; the last double's storage is a ticket that must
 * be 0 before the first call and is left 0), added in index order by the workgroup that finishes last.  map [nslots]: target joint of
 * every predicted slot; dy [F, ny]: d loss / d y.  Predictions and gradients bit-identical to the three calls; the two loss figures
 * agree to fp64 rounding of a different (but fixed) summation order (Train_Upper.py:165-182, Train_Lower.py:199-224). */
int mmego_head_fk_loss(void* stream, int which, const float* y, const float* body, int B, long F, float* q, float* joints_h,
                       const float* Rw, const float* tw, float* world, long long* counters, int ncount,
                       unsigned long long* seed_ctr, const float* target, const int* map, int ntgt, double scale, float* loss,
                       float* dy, double* scratch);
/* y[F,9] -> R[F,3,3] (eps rule of IMU_Net.py:7-18), t[F,3]. */
int mmego_imu_head(void* stream, const float* y, long F, float* R, float* t);
/* IMU_Net.fc2 (Net/IMU_Net.py:84) and mmego_imu_head in one launch: y = X [F][K] . W[9][K]^T + b (row-wise dot products, fixed
 * summation structure), then R, t as mmego_imu_head; y_out [F][9] optional.  K % 256 == 0, ldx % 4 == 0. */
int mmego_imu_fc2_head(void* stream, const float* X, long ldx, const float* W, const float* b, long F, int K, float* y_out,
                       float* R, float* t);
/* loss[0] = sum |pred - target[:, map]|, grad = scale*sign(.) (L1Loss(reduction='sum'), Train_Upper.py:53,179);
 * loss[1] = sum of the per-joint Euclidean distances (the per-minibatch accuracy log, Train_Upper.py:183-185).
 * `loss` holds TWO floats. */
int mmego_l1_loss(void* stream, const float* pred, const float* target, const int* map, int nsel, int ntgt, long F,
                  float scale, float* loss, float* grad);
/* Keep the `keep` rows with the largest column-0 key, descending, ties lowest index first; idx is int64
 * (Lower_Net.py:216-227). */
int mmego_topk_rows(void* stream, const float* pts, long F, int N, int C, int keep, float* out, long long* idx);
/* Per-frame evaluation figures of `--infer` (Processor/Test/Demo_test.py:64-69,121-123,150-163):
 * E[f,0:21] joint errors (m) of the assembled 21-joint skeleton, E[f,21:41] bone angles (deg),
 * E[f,41] mean upper-joint error, E[f,42] mean lower-joint error.  Means over frames via mmego_colsum. */
int mmego_pose_errors(void* stream, const float* upper, const float* lower, const float* target, long F, float* E);
/* Per-frame figures of the Upper stage's epoch evaluation (Processor/Train/Train_Upper.py:75-88 angle_loss, :228-240):
 * U[f,0:15] error (m) of the 15 upper joints (upper_joint_map order), U[f,15:29] angles (deg) of the 14 upper-body bones,
 * U[f,29] sum |upper - target[upper_joint_map]| of the frame (its share of L1Loss(sum)).  upper [F,15,3], target [F,21,3], U [F,30]. */
int mmego_pose_errors_upper(void* stream, const float* upper, const float* target, long F, float* U);

/* ---- eval-mode front end of Upper_Net in one launch (front.hip) ---------------------------------------
 * Per frame of N points (x [F,N,6], N a multiple of 16, <= 1024): Transform2H (Utils.py:284-292; bit-identical to mmego_transform2h,
 * written back into x -- quirk Q1 -- from x_src when given), PointNet 6-8-16-24 (Net/Upper_Net.py:242-266), concat with the first
 * four transformed columns, GlobalPointNet 28-32-48-64 and its softmax attention pooling (:270-301), all with eval-mode BatchNorm
 * folded into the k=1 convs.  vec [F,64] pooled features, attn [F,N] attention weights.  w: HOST array of 38 device pointers --
 * for PointNet conv1..3 then GlobalPointNet conv1..3: weight, bias, BatchNorm gamma, beta, running_mean, running_var; then the
 * attention Linear's weight [64] and bias [1].  Replaces transform2h + 2 x mlp3_eval + attn_pool_forward and their HBM round trips. */
int mmego_upper_front_eval(void* stream, float* x, const float* x_src, const float* R, const float* t, long F, int N,
                           const float* const* w, float eps, float* vec, float* attn);
/* A whole BiLSTM layer's recurrence for Bn <= 64 rows and H = 512 as ONE persistent launch with stationary weights (lstm_seq.hip;
 * IMU_Net's rnn_slow, Net/IMU_Net.py:61-62,82): xproj [Bn*T][xs >= 8H] rows (b*T + t) = W_ih x + b_ih of direction d at columns [4H d,
 * 4H d + 4H); out [Bn*T][os >= 2H]: h_t of direction d at columns [H d, H d + H); h_0 = c_0 = 0.  256 workgroups in 8 independent
 * groups (direction x 16 rows) that exchange h_t as (value, tag) words through xbuf.  sync: 16 unsigned words, xbuf: 2 * 8 * 16 * 512
 * 8-byte words, both zero before the first launch and private to one stream of launches; sync[9] is a sticky error flag (non-zero: a
 * bounded spin ran out -- the launch's workgroups could not all be resident -- and the results are invalid), sync[10] the launch
 * generation. */
/* mmego_lstm_seq_xcd_slots: how many such launches the current device holds at once (the 256 workgroups of a launch must be
 * co-resident): the kernel's occupancy (hipOccupancyMaxActiveBlocksPerMultiprocessor) x CU count / 256, per device -- 2 on a whole
 * MI355X, 0 on a device / partition too small for one launch.  mmego_lstm_seq_xcd_ok is 0 there too.  A caller that may have k such
 * launches in flight on different streams at once (two frozen IMU_Net forwards side by side) needs slots >= k. */
int mmego_lstm_seq_xcd_slots(void);
int mmego_lstm_seq_xcd_ok(int Bn, int H, int T);
int mmego_lstm_seq_xcd(void* stream, const float* xproj, long xs, const float* whh0, const float* whh1, const float* bhh0,
                       const float* bhh1, float* out, long os, unsigned* sync, unsigned long long* xbuf, int Bn, int H, int T);
/* The same launch with the six stages' operands (folded weights, activation tiles) rounded to bf16 and fp32 accumulation
 * (front_bf16.hip; opt-in precision mode of eval forwards).  Transform2H and its write-back stay fp32 and bit-identical. */
int mmego_upper_front_eval_bf16(void* stream, float* x, const float* x_src, const float* R, const float* t, long F, int N,
                                const float* const* w, float eps, float* vec, float* attn);

/* ---- fp32-accurate products on the bf16 matrix pipe ("split3", split3.hip) ------------------------------------
 * Opt-in mode of the frozen IMU_Net forward (Net/IMU_Net.py:58-62,76-83; IMUNet.precision = "split3"): every fp32 operand of the
 * BiLSTM products is split EXACTLY into three bf16 pieces a = a1 + a2 + a3 (round to nearest even, residuals exact) and a product
 * is the fp32-accumulated sum of the six piece products a1b1, a1b2, a2b1, a2b2, a1b3, a3b1 (nprod = 6; dropped terms <= 2^-24
 * relative) or of all nine (nprod = 9) on v_mfma_f32_32x32x16_bf16.
 * Operand layout "sfrag": block (rb, s, p) -- 32 rows x 16 k of piece p as the MFMA lanes read it, lane l = row 32 rb + l % 32,
 * k = 16 s + 8 (l / 32) .. + 8 -- is 1 KB at ((rb * K / 16 + s) * 3 + p) KB.  Activation rows are time-major (t * Bp + b). */
int mmego_split3_cvt(void* stream, const float* X, long ldx, long rows_in, int K, int tm, int Bn, int T, int Bp, long Rp,
                     unsigned short* Y);
/* mmego_split3_gemm cut into nsplit slabs of K (a long contraction with few output tiles): slab y leaves its partial product row-major
 * at ws + y * (32 Mrb) * (32 Nrb) floats -- the format mmego_slab_reduce kind 0 sums. */
int mmego_split3_gemm_slabs(void* stream, const unsigned short* A, const unsigned short* W, float* ws, int Mrb, int Nrb, int K,
                            int nprod, int wm, int nsplit);
/* out[i] = sum over y < nsplit of ws[y * n + i] (slab order): the K slabs of mmego_split3_gemm_slabs as one streaming sum.  n % 4 == 0. */
int mmego_split3_slab_sum(void* stream, const float* ws, int nsplit, long n, float* out);
/* The pieces of X^T for X [R][C] row-major (weight-gradient operands: both are read along the row axis): Y = sfrag of the [Cp x R]
 * matrix, Cp % 32 == 0 >= C, R % 16 == 0.  T > 0: column r of X^T is row r + shift of X where that row lies in the same T-row sequence
 * (rows b T + t), zero otherwise -- h_{t-1} / h_{t+1} of a layer's outputs without a shifted copy; T = 0, shift = 0: plain. */
int mmego_split3_cvt_t(void* stream, const float* X, long ldx, long R, int C, long Cp, unsigned short* Y, int shift, int T);
int mmego_split3_join(void* stream, const unsigned short* Y, long Rp, int K, float* X, long ldx);
int mmego_split3_fc_relu(void* stream, const float* X, long ldx, const float* W, const float* bias, int Bn, int T, int Cin,
                         int H, unsigned short* Y, int Bp, int relu);
int mmego_split3_gemm(void* stream, const unsigned short* A, const unsigned short* W, float* Cf, float* C, long ldc,
                      const float* bias, int Mrb, int Nrb, int K, int M, int nprod, int wm);
int mmego_split3_step(void* stream, int ndir, int Bn, int H, int first, const unsigned short* hprev0, const unsigned short* hprev1,
                      long hrb, const unsigned short* whh0, const unsigned short* whh1, const float* xpf, long mt0_0, long mt0_1,
                      float* hout0, float* hout1, long hos, unsigned short* hnext0, unsigned short* hnext1, long hnrb,
                      float* c0, float* c1, int nprod, int dbase);
/* projection for few rows (M <= 2048): C[m][d 4H + n] = A[m][:] . W_d[n][:] + bias[d 4H + n], both directions; W_d rows reordered
 * [32-unit block][gate][32 units]; K = 256 / 512 / 1024; C row-major (IMU_Net's rnn_slow: Net/IMU_Net.py:61-62,82) */
int mmego_split3_proj(void* stream, const unsigned short* A, const unsigned short* W0, const unsigned short* W1, const float* bias,
                      float* C, long ldc, int M, int H, int K, int nprod);
/* the same timestep on 16-unit workgroups (two per CU: single-direction launches of a layer's two directions run as two chains);
 * W_hh rows -- and the projection's columns -- ordered [16-unit block][gate][16 units] */
int mmego_split3_step16(void* stream, int ndir, int Bn, int H, int first, const unsigned short* hprev0, const unsigned short* hprev1,
                        long hrb, const unsigned short* whh0, const unsigned short* whh1, const float* xpf, long mt0_0, long mt0_1,
                        float* hout0, float* hout1, long hos, unsigned short* hnext0, unsigned short* hnext1, long hnrb,
                        float* c0, float* c1, int nprod, int dbase);

/* ---- anchor ("voxel") grouping of UpperNetwlocal (group.hip) -----------------------------------------
 * Per frame and per anchor of the 3x3x3 grid: indices (int64, exact, stable ties) of the 8 nearest points and
 * the gathered rows cat(anchor, xyz-anchor, features) -- square_distance / point_ball_set / AnchorGrouping of
 * Net/Upper_Net.py:10-32,54-72,100-119 (quirk Q5: points with xyz == 0 are at distance +inf).
 * xf rows hold xyz in columns 0:3 and D features after them; grouped is [F*27*8, 6+D]; dist_out may be NULL. */
int mmego_anchor_group(void* stream, const float* xf, long ldx, long F, int N, int D, const float* anchors,
                       long long* idx, float* grouped, float* dist_out);
/* Scatter the gradient of the gathered rows back to the points (accumulates into dxf, fixed order). */
int mmego_anchor_group_backward(void* stream, const float* dgrouped, const long long* idx, long F, int N, int D,
                                float* dxf, long lddx);

/* ---- the anchor ("voxel") branch of UpperNetwlocal, fused (local.hip): Net/Upper_Net.py:10-119,147-177,219-239 ---------------------
 * mmego_local_group_l1: per frame the 27 x N squared distances (reference rounding, Q5), the 8 nearest points of every anchor (int64
 * idx [F][27][8], stable ties) by wave-wide minimum rounds, the gathered rows cat(anchor, xyz - anchor, features) built in LDS and
 * multiplied there with LocalPointNet's first k=1 conv: Z1 [F*216][C1] = gathered W1^T + b1, part1[min(nwg, F)][2][64] = per-workgroup
 * (sum z, sum z^2) in fp64 (mmego_mlp_fwd_layer_n's in_part).  grouped (may be NULL) receives the gathered rows [F*216][6+D] (training:
 * the layer's weight gradient reads them); W1 NULL: grouping only.  N in {64, 128, 256}; 6 + D <= 32; C1 <= 32, C1 % 4 == 0.
 * mmego_pool8_bn_act: y = relu(BN(Z)) with the statistics finalized from mmego_mlp_fwd_layer's partials, attention score y.w + b,
 * softmax over each group's 8 rows -> attn [rows], pooled vectors -> voxT [F][64][27] (Conv3d input order); rows = F*27*8, C = 64.
 * mmego_pool8_backward: gradient of Z's activated rows dY from dvoxT [F][64][27] (y recomputed from Z and state [4][64]), gpart
 * [mmego_pool8_nblk(rows)][2][64] = (sum g, sum g xhat) with g = dY.[y > 0] (mmego_mlp_bwd_layer's g_part), awpart [nblk][128] =
 * per-workgroup partials of d(attention weight) [0:64] and d(bias) [64].
 * mmego_anchor_scatter: dxf[f][p][0:3+D] += sum over slots with idx == p of dgrouped[slot][3:6+D], slot order (deterministic). */
int mmego_local_group_l1(void* stream, const float* feats, long ldf, long F, int N, int D, const float* anchors, long long* idx,
                         float* grouped, const float* W1, const float* b1, int C1, float* Z1, long ldz1, double* part1, int nwg,
                         float* dist_out);
/* The eval-mode branch up to the pooled vectors in ONE launch: grouping, LocalPointNet's three conv + BatchNorm (running statistics,
 * folded) + ReLU stages, attention score, softmax over the 8 members, weighted sum; tab = HOST array of 20 device pointers: per layer
 * {W, b, gamma, beta, running_mean, running_var}, then the attention weight and bias.  Writes idx, attn [F*216], voxT [F][64][27]. */
int mmego_local_front_eval(void* stream, const float* feats, long ldf, long F, int N, int D, const float* anchors, long long* idx,
                           const float* const* tab, double eps, float* voxT, float* attn);
int mmego_pool8_nblk(long rows);
int mmego_pool8_bn_act(void* stream, const float* Z, long ldz, long rows, const double* part, const float* gamma, const float* beta,
                       double eps, float* rmean, float* rvar, double momentum, float* state, const float* aw_w, const float* aw_b,
                       float* voxT, float* attn);
int mmego_pool8_backward(void* stream, const float* Z, long ldz, long rows, const float* state, const float* attn, const float* dvoxT,
                         const float* aw_w, float* dY, long lddy, double* gpart, float* awpart);
int mmego_anchor_scatter(void* stream, const float* dgrouped, const long long* idx, long F, int N, int D, float* dxf, long lddx);
/* GlobalPointNet's softmax pooling over the 128 points of a frame (Upper_Net.py:285-301) fused the same way: forward = BatchNorm (from
 * mmego_mlp_fwd_layer's partials) + ReLU + score + softmax + weighted sum -> vec [F][64], attn [F*128] (the activated tensor is never
 * stored); backward = row gradients dY with y recomputed, gpart [rows/256][2][64] and awpart [rows/256][128] as mmego_pool8_backward.
 * rows = F * 128 with mmego_pool128_ok(rows) (two frames per workgroup = mlp_train's 256-row partition, rows <= 65 536). */
int mmego_pool128_ok(long rows);
int mmego_pool128_bn_act(void* stream, const float* Z, long ldz, long rows, const double* part, const float* gamma, const float* beta,
                         double eps, float* rmean, float* rvar, double momentum, float* state, const float* aw_w, const float* aw_b,
                         float* vec, float* attn);
int mmego_pool128_backward(void* stream, const float* Z, long ldz, long rows, const float* state, const float* attn, const float* vec,
                           const float* dvec, const float* aw_w, float* dY, long lddy, double* gpart, float* awpart);

/* ---- pooling / attention / graph (pool.hip) -------------------------------------------------------
 * Softmax-attention pooling over the P points of each of G groups (Upper_Net.py:285-301,163-177,
 * IMU_Net.py:79-80). */
int mmego_attn_pool_forward(void* stream, const float* X, const float* w, const float* b, long G, int P, int C,
                            float* vec, float* attn);
int mmego_attn_pool_backward(void* stream, const float* X, const float* w, const float* attn, const float* dvec, long G,
                             int P, int C, float* dX, float* pdw, float* pdb);
/* Y[g,:] = scale * sum_p X[g,p,:] and its broadcast backward (Lower_Net.py:112-115). */
int mmego_group_sum(void* stream, const float* X, long G, int P, int C, float scale, float* Y, long ldy);
int mmego_group_bcast(void* stream, const float* dY, long lddy, long G, int P, int C, float scale, float* dX,
                      int accumulate);
/* softmax(Q K^T * scale) V, 64 queries x 15 keys x 64 channels per frame (Lower_Net.py:105-109).  ldkv: row stride of K and V
 * (and of dK, dV) in floats -- K and V may be the column halves of one [rows, 128] buffer (to_k and to_v as one stacked product). */
int mmego_cross_attn_forward(void* stream, const float* Q, const float* K, const float* V, long F, float scale, float* O,
                             long ldo, float* P, long ldkv);
/* The eval-mode form (r06): osum[f][0..63] (row stride ldos) = the sum over frame f's 64 queries of the attention output -- what
 * Lower_Net.py:131-133 does with it next (gate == 1) -- added in ascending query order (the bits of mmego_group_sum2 over O); neither
 * O nor P is stored.  Q, K, V 16-byte aligned, ldkv % 4 == 0. */
int mmego_cross_attn_forward_pooled(void* stream, const float* Q, const float* K, const float* V, long F, float scale, float* osum,
                                    long ldos, long ldkv);
/* ... with the query projection of Lower_Net.py:100 inside: Q = X Wq^T + bq from the point features X [F * 64][ldx] (Wq [64][64], bq
 * [64]) is computed per frame on the matrix pipe and never stored.  X, Wq, bq, K, V 16-byte aligned, ldx % 4 == 0, ldkv % 4 == 0. */
int mmego_cross_attn_forward_pooled_q(void* stream, const float* X, long ldx, const float* Wq, const float* bq, const float* K,
                                      const float* V, long F, float scale, float* osum, long ldos, long ldkv);
int mmego_cross_attn_backward(void* stream, const float* Q, const float* K, const float* V, const float* P,
                              const float* dO, long lddo, long F, float scale, float* dQ, float* dK, float* dV, long ldkv);
/* Gradient of einsum('nkctv,kvw->nctw') wrt A (GCN.py:62).  Writes per-block partial sums
 * partial_ws[mmego_graph_dA_nblk(G)][K*V*V]; reduce them with mmego_colsum (scale = A gives the edge-importance gradient).
 * A, imp, dZ (all three or none): the einsum's input gradient dZ [G, V, K*C] from the same launch (what mmego_graph_mix
 * with backward = 1 computes, same bits).  ldz / lddz: row strides of Z / dZ in floats (column slices of wider buffers). */
int mmego_graph_dA_nblk(long G);
int mmego_graph_dA(void* stream, const float* Z, const float* dY, long G, int V, int K, int C, float* partial_ws,
                   const float* A, const float* imp, float* dZ, long ldz, long lddz);
/* ---- ST-GCN layer pieces (gcn.hip): Net/GCN.py:55-64 (graph convolution einsum), :108-122 (9x1 temporal convolution) --------
 * mmego_graph_mix: Y[f][w][c] = sum_k sum_v (A . importance)[k][v][w] X[f][v][k*C + c]  (backward = 0; X is z [F][V][K*C]), or the
 * input gradient Y[f][v][k*C + c] = sum_w (A . importance)[k][v][w] X[f][w][c] (backward = 1; X is dy [F][V][C]).
 * ldx: row stride of X in floats (X may be a column slice of a wider buffer; Y is dense).
 * stats (may be NULL; forward, F <= 1024): [C][F][3] BatchNorm partial records (V, mean, M2) of Y per frame and channel, for
 * mmego_bn_finalize -- the batch statistics of st_gcn.tcn[0] without a pass of their own over Y.
 * mmego_tconv: temporal convolution over rows (b, t, v) as an implicit GEMM, out[r][n] = bias[n] + sum_tap sum_k
 * act(X[r + (tap - taps/2) V][k]) Wp[(tap*Cout + n)*Cin + k], taps leaving the sequence contribute zero; act = ReLU(BatchNorm)
 * given by in_state [4][Cin] (mean, invstd, gamma*invstd, beta) or identity (NULL).  Wp: the conv weight re-packed by
 * mmego_tconv_pack (mode 0; the input gradient of the convolution is the same call with X = dY, Cin/Cout swapped, on the mode-1
 * pack).  act_out (may be NULL, needs in_state): [rows][Cin], receives act(X) -- what the backward pass keeps.  Cin % 4 == 0.
 * mmego_tconv_pack: W[co][ci][tap] -> [tap][co][ci] (mode 0) / [taps-1-tap][ci][co] (mode 1) / both, one behind the other (mode 2).
 * mmego_tconv_wgrad: dW[co][ci][tap] (+)= sum_r dY[r][co] Xa[r + (tap - taps/2) V][ci] (the convolution's weight gradient, no
 * unfolded operand); ws: mmego_tconv_wgrad_nsplit(...) * taps * Cout * Cin floats of partial tiles, added in a fixed order. */
int mmego_graph_mix(void* stream, const float* X, const float* A, const float* importance, float* Y, long F, int V, int K, int C,
                    int backward, float* stats, long ldx);
int mmego_tconv_pack(void* stream, const float* W, int Co, int Ci, int taps, int mode, float* Wp);
int mmego_tconv(void* stream, const float* X, long ldx, const float* in_state, const float* Wp, const float* bias, float* Y,
                long ldy, float* act_out, int B, int T, int V, int Cin, int Cout, int taps);
int mmego_tconv_wgrad_nsplit(int B, int T, int V, int Cin, int Cout, int taps);
int mmego_tconv_wgrad(void* stream, const float* dY, long lddy, const float* Xa, long ldx, float* ws, float* dW, int accumulate,
                      int B, int T, int V, int Cin, int Cout, int taps);
/* out[b][c][r] = in[b][r][c]: the (B,64,T,V)->(B,T,V,64) re-view of GCN.py:351-353 (quirk Q8). */
int mmego_transpose_batched(void* stream, const float* in, float* out, long Bn, int R, int C);
int mmego_mul(void* stream, const float* a, const float* b, float* out, long n);
int mmego_add(void* stream, const float* a, const float* b, float* out, long n);
/* The once-per-training-forward tick of a net: x[i] += 1 for the BatchNorm num_batches_tracked counters (one launch for all
 * layers; n may be 0) and, when seed_ctr is given, the next value of the dropout seed counter (mmego_lstm64_forward). */
int mmego_inc_i64(void* stream, long long* x, long n, unsigned long long* seed_ctr);

/* ---- train-mode pointwise MLP layers, fused per layer (mlp_train.hip) -------------------------------------------------
 * k=1 conv -> BatchNorm (batch statistics) -> ReLU stages of Net/Upper_Net.py:242-301,147-177 and Net/Lower_Net.py:40-72 in
 * training.  Channel widths <= 64.  nblk = mmego_mlp_train_nblk(rows) workgroups per layer launch; statistics partials are
 * nblk x 2 x 64 doubles, dW partials nblk x 4096 floats.  A BatchNorm "state" is [4][C] floats: mean, invstd, gamma*invstd, beta.
 *   mmego_mlp_fwd_layer     Z = act(X) W^T + bias, act = ReLU(BatchNorm) of the layer below finalized from in_part (NULL: identity;
 *                           workgroup 0 stores in_state and updates in_rmean / in_rvar with in_momentum); out_part receives this
 *                           layer's (sum z, sum z^2).
 *   mmego_mlp_bn_act        Y = ReLU(BatchNorm(Z)) for the last stage, statistics finalized from part (state, running stats as above).
 *   mmego_mlp_bn_bwd_reduce part = (sum g, sum g xhat), g = dY . [BatchNorm(Z) > 0]: first reduction of a block's backward chain.
 *   mmego_mlp_bwd_layer     dz from (dY, Z, state, g_part); dgamma, dbeta; dX = dz W (NULL: skipped); gprev_part = the same
 *                           two sums for the layer below (Xin = its pre-BN z, in_state its BatchNorm; NULL in_state: Xin is the
 *                           plain layer input); dW_part = per-workgroup dz^T act(Xin).
 *   mmego_mlp_dw_reduce     dW_l = fixed-order sum of the nblk partials, up to three layers per launch. */
int mmego_mlp_train_nblk(long rows);
int mmego_mlp_fwd_layer(void* stream, const float* X, long ldx, long rows, int Cin, const double* in_part, const float* in_gamma,
                        const float* in_beta, double in_eps, float* in_rmean, float* in_rvar, double in_momentum,
                        float* in_state, const float* W, const float* bias, int Cout, float* Z, long ldz, double* out_part);
int mmego_mlp_fwd_layer_n(void* stream, const float* X, long ldx, long rows, int Cin, const double* in_part, int in_nblk,
                          const float* in_gamma, const float* in_beta, double in_eps, float* in_rmean, float* in_rvar,
                          double in_momentum, float* in_state, const float* W, const float* bias, int Cout, float* Z, long ldz,
                          double* out_part);
int mmego_mlp_bn_act(void* stream, const float* Z, long ldz, long rows, int C, const double* part, const float* gamma,
                     const float* beta, double eps, float* rmean, float* rvar, double momentum, float* state, float* Y,
                     long ldy);
int mmego_mlp_bn_bwd_reduce(void* stream, const float* dY, long lddy, const float* Z, long ldz, long rows, int C,
                            const float* state, double* part);
int mmego_mlp_bwd_layer(void* stream, const float* dY, long lddy, const float* Z, long ldz, long rows, int Cout,
                        const float* state, const double* g_part, float* dgamma, float* dbeta, const float* Xin, long ldxin,
                        int Cin, const float* in_state, const float* W, float* dX, long lddx, double* gprev_part,
                        float* dW_part);
/* mmego_mlp_bwd_layer for the first stage of LocalPointNet with its input rows GATHERED on the fly through the group indices (row r =
 * cat(anchors[(r % 216) / 8], xyz - anchor, features) of point gidx[r] of frame r / 216; feats [F*N][ldf] = xyz | D features): the
 * gathered tensor of mmego_local_group_l1 need not be kept for the layer's weight gradient. */
int mmego_mlp_bwd_layer_gather(void* stream, const float* dY, long lddy, const float* Z, long ldz, long rows, int Cout,
                               const float* state, const double* g_part, float* dgamma, float* dbeta, const long long* gidx,
                               const float* feats, long ldf, const float* anchors, int N, int D, const float* W, float* dX,
                               long lddx, float* dW_part);
int mmego_mlp_dw_reduce(void* stream, long rows, int nlayers, const float* part0, float* dW0, int Cout0, int Cin0,
                        const float* part1, float* dW1, int Cout1, int Cin1, const float* part2, float* dW2, int Cout2, int Cin2);
/* mmego_mlp_dw_reduce for up to 9 layers of several chains in ONE launch (descs: n MmegoDwRed; a layer's partials are
 * mmego_mlp_train_nblk(rows) x 4096 floats).  nblk > 0 / stride > 0 describe other per-workgroup partial records summed the same way:
 * nblk records `stride` floats apart, element (m, n) of a record at m * 64 + n (e.g. the pooling kernels' attention-parameter partials:
 * Cout = 1, Cin = 64, stride 128). */
typedef struct MmegoDwRed { const float* part; float* dW; int Cout, Cin; long rows; int nblk; long stride; } MmegoDwRed;
int mmego_mlp_dw_reduce_multi(void* stream, int n, const void* descs);

/* ---- LocalVoxelNet training step (vox.hip): Net/Upper_Net.py:180-205 -------------------------------------------------------------
 * Conv3d(64, 96, k=3) over the whole 3x3x3 anchor grid (= a 1728 -> 96 map per frame), two 1x1x1 convs 96 -> 128 -> 64, a train-mode
 * BatchNorm3d + ReLU behind each, over rows = B*T frames: 4 forward + 5 backward launches, one wherever a BatchNorm needs every row's
 * statistics (mmego_vox_ok: the widths these kernels are built for, K = 1728, 2 <= rows <= 4096; other shapes take the generic
 * launches).  Forward statistics travel as (mean, M2) records [ceil(rows/16)][C] float pairs (rec*), finalized in the consumer's
 * prologue; workgroup 0 writes bn->state [4][C] = mean, invstd, a, b and the running statistics (bn: an MmegoBnRef whose rec / nrec /
 * rows_per_rec fields are ignored).  Backward sums travel as (sum g, sum g xhat) pairs per 16-row tile (prt*).
 *   vox_l1_fwd    Z[rows][C] = X[rows][K] W[C][K]^T + bias, rec
 *   vox_mid_fwd   Yin = relu(bn(Zin)) [rows][Cin] (stored for backward), Zout = Yin W[Cout][Cin]^T + bias, rec_out; (Cin, Cout) = (96, 128), (128, 64)
 *   vox_out_fwd   Y[rows][C] (row stride ldy) = relu(bn(Z))
 *   vox_bwd_sums  G = dY . [Y > 0] [rows][C], prt from Z and the forward state
 *   vox_mid_bwd   dZ = BatchNorm backward of G (sums prt, state; dgamma / dbeta assigned), Gp = (dZ W[Cout][Cin]) . [Yp > 0], prtp from Zp / statep;
 *                 (Cout, Cin) = (64, 128), (128, 96)
 *   vox_l1_bwd    dZ as above for the first layer (C = 96), dX[rows][K] = dZ W[C][K]
 *   vox_dw        dW1[C1][K] = dZ1^T X, dW2[C2][C1] = dZ2^T Y1, dW3[C3][C2] = dZ3^T Y2 (assigned)
 * The conv biases get no gradient (exactly zero in front of a batch-statistics BatchNorm). */
int mmego_vox_ok(long rows, int K, int C1, int C2, int C3);
int mmego_vox_l1_fwd(void* stream, const float* X, long ldx, long rows, int K, const float* W, const float* bias, int C, float* Z,
                     float* rec);
int mmego_vox_mid_fwd(void* stream, const float* Zin, const float* rec_in, const void* bn, long rows, int Cin, float* Yin,
                      const float* W, const float* bias, int Cout, float* Zout, float* rec_out);
int mmego_vox_out_fwd(void* stream, const float* Z, const float* rec, const void* bn, long rows, int C, float* Y, long ldy);
int mmego_vox_bwd_sums(void* stream, const float* dY, long lddy, const float* Y, long ldy, const float* Z, const float* state,
                       long rows, int C, float* G, float* prt);
int mmego_vox_mid_bwd(void* stream, const float* G, const float* Z, const float* state, const float* prt, long rows, int Cout,
                      float* dZ, float* dgamma, float* dbeta, const float* W, int Cin, const float* Yp, const float* Zp,
                      const float* statep, float* Gp, float* prtp);
int mmego_vox_l1_bwd(void* stream, const float* G, const float* Z, const float* state, const float* prt, long rows, int C, float* dZ,
                     float* dgamma, float* dbeta, const float* W, int K, float* dX, long lddx);
int mmego_vox_dw(void* stream, long rows, const float* dZ1, const float* X, long ldx, float* dW1, int C1, int K, const float* dZ2,
                 const float* Y1, float* dW2, int C2, const float* dZ3, const float* Y2, float* dW3, int C3);

/* ---- fused ST-GCN training step (gcn_fused.hip, gcn.hip): Net/GCN.py:67-147 st_gcn, :332-355 Model.extract_feature -----------------
 * Train-mode BatchNorm statistics travel between these kernels as PARTIAL RECORDS: per producer workgroup j and channel c the float
 * pair rec[j][c] = (mean_j, M2_j) over the rows the workgroup owns (rows_per_rec each, the last record ragged).  The consumer finalizes
 * them in its prologue (every workgroup, fixed order); workgroup 0 writes state [4][C] = mean, invstd, a, b and updates the running
 * statistics with torch semantics.  Replaces the colstats / bn_finalize / affine_act launches between the products. */
typedef struct MmegoBnRef {
  const float* rec; int nrec; int rows_per_rec;
  const float* gamma; const float* beta; float* running_mean; float* running_var; float momentum; float eps;
  float* state;
} MmegoBnRef;
/* One launch per st_gcn block front: [in_mode 1: x = relu(bn1(X1) + bn2(X2)), both BatchNorms from records | in_mode 0: x =
 * data_bn(X1), X1 = the frame tensor [F][V*cin], statistics computed by the kernel itself (F <= 1024)] -> xact [F*V][cin] (kept for
 * backward; may be NULL) -> zr = x W^T + bias, W [nout][cin] the stacked weight of the graph convolution's and the residual branch's
 * 1x1 convs -> mix = 1: Z [F*V][(K+1) cout] = zr, Y [F*V][cout] = einsum('nkctv,kvw->nctw', z, A . importance), recY / recR
 * [mmego_gcn_front_nrec(F)][cout]: records of Y and of the residual columns of Z (4 frames = 4 V rows per record); mix = 0: the
 * closing 1x1 conv, stored transposed outT[b][nout][T*V] (GCN.py:351-353, the re-viewed layout).  cin in {<32 with in_mode 0, 32,
 * 64, 128}; nout a multiple of 32 (<= 384 with cin <= 64). */
typedef struct MmegoGcnFront {
  const float* X1; long ld1; const float* X2; long ld2; int in_mode;
  MmegoBnRef bn1, bn2;
  float* xact;
  const float* W; const float* bias; int cin, nout;
  int mix, K, cout; const float* A; const float* importance;
  float* Z; long ldz; float* Y; float* recY; float* recR;
  float* outT; int T;
  long F; int V;
} MmegoGcnFront;
int mmego_gcn_front_nrec(long F);
int mmego_gcn_front(void* stream, const void* desc);
/* mmego_tconv in the fused training step: in_bn = MmegoBnRef of the BatchNorm (+ReLU) in front, out_rec [ceil(rows/64)][Cout] =
 * records of the output for the BatchNorm behind; mmego_tconv_bwd_stats = the input-gradient call (X = dY, gradient pack) whose
 * epilogue leaves bw_rec [ceil(rows/64)][Cout] = (sum g, sum g xhat), g = dAct . [bn(ymix) > 0], for the BatchNorm + ReLU in front of
 * the convolution's input (state [4][Cout]).  mmego_pack_multi: every weight re-layout of a step in one launch, n <= 8 entries -- kind 0:
 * mmego_tconv_pack mode 2 of W[Co][Ci][taps]; kind 1: a k=1 conv weight W[Co][Ci] (both multiples of 32) in the FRAGMENT-MAJOR order
 * mmego_gcn_front reads with one coalesced 1-KB fetch per MFMA operand group (gcn.hip, pack_multi_kernel) -- MmegoGcnFront.W then
 * points at that copy when cin >= 32. */
int mmego_tconv_train(void* stream, const float* X, long ldx, const void* in_bn, const float* Wp, const float* bias, float* Y,
                      long ldy, float* act, float* out_rec, int B, int T, int V, int Cin, int Cout, int taps);
int mmego_tconv_bwd_stats(void* stream, const float* dY, long lddy, const float* Wp, float* dAct, long ldda, const float* ymix,
                          long ldym, const float* state, float* bw_rec, int B, int T, int V, int Cin, int Cout, int taps);
/* Sequence-tiled forms of the two calls above for T*V <= 128 rows per sequence and Cin in {32, 64, 128} (mmego_tconv_seq_ok): one
 * workgroup per (sequence, 32 output columns), the sequence's rows staged in LDS once for all taps, weights Wf fragment-major per tap
 * (mmego_pack_multi kind 2: 2 * taps * Co * Ci floats, the forward image then the input-gradient image); out_rec / bw_rec are per
 * SEQUENCE: [B][Cout], T*V rows per record. */
int mmego_tconv_seq_ok(int T, int V, int Cin, int Cout);
int mmego_tconv_seq_train(void* stream, const float* X, long ldx, const void* in_bn, const float* Wf, const float* bias, float* Y,
                          long ldy, float* act, float* out_rec, int B, int T, int V, int Cin, int Cout, int taps);
int mmego_tconv_seq_bwd(void* stream, const float* dY, long lddy, const float* Wf, float* dAct, long ldda, const float* ymix,
                        long ldym, const float* state, float* bw_rec, int B, int T, int V, int Cin, int Cout, int taps);
/* Eval-mode ST-GCN block with bf16 operands and fp32 accumulation (opt-in precision mode; csrc/gcn_bf16.hip; reference
 * Net/GCN.py:55-64,67-147 with the BatchNorms frozen).  A block = mmego_gcn_mix_eval_bf16 + mmego_tconv_eval_bf16:
 *   gcn_mix: Yact = bf16(relu(bn0(einsum(conv1x1(X), A . importance)))), Rn = bn_r(residual conv1x1(X)) in fp32; the einsum is applied
 *     to X in fp32 in front of the product (exact in real arithmetic), the product's k axis is [A_0^T X | A_1^T X | A_2^T X | X].
 *     Wy / Wr: the two weights in bf16, fragment-major [Cout / 32][k steps][64 lanes][8] with lane (n, hf) of step ks holding
 *     W[32 ct + n][16 ks + 8 hf .. + 8] over the k axis above (Wy: rows of the graph conv, zero over X's columns, steps [0, ksy);
 *     Wr: the residual conv, zero elsewhere, steps [kr0, kr0 + ksr); ksy = ceil(K Cin / 16), kr0 = floor(K Cin / 16), kr0 + ksr =
 *     ceil((K + 1) Cin / 16)); biasy [V][Cout] = sum_k
 *     colsum_v((A . importance)[k])[w] b_k[c]; in_state: [4][V Cin] state of data_bn applied while loading (first block) or NULL;
 *     st0 / st_r: [4][Cout] mean, invstd, a, b.
 *   tconv: Y = relu?(bn?(bias + temporal conv of X (bf16, activated)) + res); Wp from mmego_tconv_pack_bf16 (taps * Cout * Cin bf16);
 *     one workgroup per sequence, (T + taps - 1) V rows of LDS (mmego_tconv_eval_bf16_ok). */
int mmego_gcn_mix_eval_bf16_ok(int V, int Cin, int Cout, int K);
int mmego_gcn_mix_eval_bf16(void* stream, const float* X, const float* in_state, const float* A, const float* importance,
                            const unsigned short* Wy, const unsigned short* Wr, const float* biasy, const float* biasr,
                            const float* st0, const float* st_r, unsigned short* Yact, float* Rn, long F, int V, int Cin, int Cout,
                            int K);
int mmego_tconv_eval_bf16_ok(int T, int V, int Cin, int Cout, int taps);
int mmego_tconv_eval_bf16(void* stream, const unsigned short* X, const unsigned short* Wp, const float* bias, const float* post,
                          const float* res, long ldr, float* Y, long ldy, int relu, int B, int T, int V, int Cin, int Cout, int taps);
int mmego_tconv_pack_bf16(void* stream, const float* W, int Cout, int Cin, int taps, unsigned short* Wp);
typedef struct MmegoPack { const float* W; float* Wp; int Co, Ci, taps, kind; } MmegoPack;
int mmego_pack_multi(void* stream, int n, const void* descs);
/* Backward of a block's closing pair out = relu(BN(X1) + BN(X2)) (same dY, mask = out): reduce -> rec [ceil(rows/64)][2C] (sum g, sum
 * g xhat), virtual channels [0, C) = first BatchNorm, [C, 2C) = second; apply: finalize in the prologue, dX = a (g - mean(g) - xhat
 * mean(g xhat)) for both, d(gamma) / d(beta) by workgroup 0.  st1 / st2: state [4][C].  C % 4 == 0, C <= 128. */
int mmego_gcn_bn_bwd_reduce(void* stream, const float* dY, long lddy, const float* mask, long ldm, const float* X1, long ld1,
                            const float* st1, const float* X2, long ld2, const float* st2, long rows, int C, float* rec);
int mmego_gcn_bn_bwd_apply(void* stream, const float* dY, long lddy, const float* mask, long ldm, const float* X1, long ld1,
                           const float* st1, const float* X2, long ld2, const float* st2, long rows, int C, const float* rec,
                           float* dgamma1, float* dbeta1, float* dX1, long lddx1, float* dgamma2, float* dbeta2, float* dX2,
                           long lddx2);
/* mmego_graph_dA with the backward of the BatchNorm + ReLU behind the einsum applied on load: dY0 = gradient of the activated rows,
 * Ymix = the einsum output (pre-BatchNorm), st0 = that BatchNorm's state, rec [nrec][C] = mmego_tconv_bwd_stats' records; writes
 * d(gamma), d(beta), the dA partials [mmego_graph_dA_fused_nblk(G)][K*V*V] and dZ. */
int mmego_graph_dA_fused_nblk(long G);
int mmego_graph_dA_fused(void* stream, const float* Z, long ldz, const float* dY0, const float* Ymix, const float* st0,
                         const float* rec, int nrec, float* dgamma0, float* dbeta0, long G, int V, int K, int C,
                         float* partial_ws, const float* A, const float* imp, float* dZ, long lddz);
/* ONE launch for the deferred partial-product sums of a backward pass (fixed order).  kind 0: split-K slabs ws[nsplit][M*N] ->
 * out[m*scm + n] (what mmego_gemm leaves with accumulate = 2; asum != NULL: the slab row sums ws[nsplit*M*N + k*M + m] -> asum[m]);
 * kind 1: mmego_tconv_wgrad's slabs (accumulate = 2) ws[nsplit][taps][Co][Ci] -> out[co][ci][tap], M = taps*Co, N = Ci; kind 2:
 * out[i] = scale[i] * sum_k ws[k][i], M = 1 (edge-importance gradient from mmego_graph_dA's partials).  n <= 24. */
typedef struct MmegoSlab {
  const float* ws; float* out; const float* scale; float* asum;
  int kind, nsplit, M, N, taps; long scm;
} MmegoSlab;
/* Launch-count helpers of the Lower_Net tail: two group sums / two group broadcasts / two 2-D copies per launch, and mmego_topk_rows
 * with the kept rows' first n2 columns written to a second buffer as well (same arithmetic as the single forms). */
int mmego_transform2h_pair(void* stream, float* pts, long F, int P, int C, const float* R, const float* t, float* pts2, int P2,
                           const float* src2);
int mmego_colsum2(void* stream, const float* X1, long ld1, long rows1, int C1, float* out1, const float* X2, long ld2, long rows2,
                  int C2, float* out2);
int mmego_group_sum2(void* stream, long G, const float* X1, int P1, int C1, float scale1, float* Y1, long ldy1, const float* X2, int P2,
                     int C2, float scale2, float* Y2, long ldy2);
int mmego_group_bcast2(void* stream, long G, const float* dY1, long lddy1, int P1, int C1, float scale1, float* dX1, const float* dY2,
                       long lddy2, int P2, int C2, float scale2, float* dX2);
int mmego_copy2d_pair(void* stream, const float* X1, long ldx1, float* Y1, long ldy1, long rows1, int C1, const float* X2, long ldx2,
                      float* Y2, long ldy2, long rows2, int C2);
int mmego_topk_rows2(void* stream, const float* pts, long F, int N, int C, int keep, float* out, long long* idx, float* out2, long ld2,
                     int n2);
int mmego_slab_reduce(void* stream, int n, const void* descs);
/* d(gamma)[c] = sum_r dY[r][c] xhat[r][c], d(beta)[c] = sum_r dY[r][c] of a BatchNorm whose input gradient is not needed (data_bn,
 * GCN.py:310: the skeleton input is detached, Train_Lower.py:196); state [4][C]. */
int mmego_bn_param_grads(void* stream, const float* dY, long lddy, const float* X, long ldx, const float* state, long rows, int C,
                         float* dgamma, float* dbeta);

/* ---- optimiser (optim.hip) -------------------------------------------------------------------------
 * torch.optim.Adam step (coupled L2 weight decay) over one flat buffer; state = 3 doubles on the device
 * {step, lr/(1-b1^t), sqrt(1-b2^t)}, advanced by the call itself so a captured graph replays correctly
 * (Train_Upper.py:60,182; Train_IMU.py:71-72).  skip: HOST array of nskip (<= 4) [begin, end) element ranges (multiples of 4) that
 * the update leaves untouched -- parameters that never receive a gradient (torch.optim.Adam skips grad=None: IMU_Net.fc3,
 * Net/IMU_Net.py:55, which would otherwise decay under Train_IMU's weight_decay).  NULL / 0: update everything.
 * ticket: one device int, 0 before the first call and left 0 by every call (the step count in `state` advances inside the update
 * launch: the workgroup that finishes last stores it). */
int mmego_adam_step(void* stream, float* p, const float* g, float* m, float* v, long n, double* state, double lr,
                    double beta1, double beta2, double eps, double weight_decay, const long* skip, int nskip, int* ticket);

#ifdef __cplusplus
}
#endif
#endif
