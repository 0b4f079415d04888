"""IMU_Net stage-1 training on the HIP path: stashing forward, backward through the two BiLSTM(512) stacks,
geodesic + position loss.  Reference: Net/IMU_Net.py:67-94, Processor/Train/Train_IMU.py:21-34,114-149.

The recurrent products of the backward pass run as split-K mmego_gemm calls (one per timestep and direction against a
per-layer transposed W_hh); input and weight gradients of the projections run on the large-tile kernels in their
natural operand orientation (dX = dY . W: NN, dW = dY^T . X: TN).
"""
import os

import torch

from . import blocks, hip, ops


def _s3_worth(rows, N, K):
    """A product of stage-1 training goes to the split3 kernels (IMUNet.train_precision = "split3") when its shape fits them and it is
    large enough for the operand conversions to pay (rnn_fast's 10 240-row products; rnn_slow's 512-row ones stay on the fp32 kernels)."""
    return rows % 32 == 0 and N % 32 == 0 and K % 32 == 0 and 2.0 * rows * N * K >= 8e9


def _s3_nsplit(M, N, K):
    """K slabs of a split3 weight-gradient product: 256 x 256 tiles (s3_gemm_big_kernel<., 8>, one workgroup per CU) x slabs cover the chip
    once, slabs of at least 512 k (r06; before: 256 x 128 tiles x slabs twice over the chip, slabs of at least 1024 k)."""
    tiles = ((M + 255) // 256) * ((N + 255) // 256)
    ns = int(max(1, min(256 // max(tiles, 1), K // 512, 64)))
    chunks = K // 32
    cps = (chunks + ns - 1) // ns
    return (chunks + cps - 1) // cps          # (no empty slab)


def _s3_slabs_320(M, N, K):
    """K slabs that bring a product of M x N outputs (M % 320 == 0, N % 256 == 0) to the 256 tiles of 320 x 256 that s3_gemm_big_kernel
    wants (one per CU), slabs of at least 512 k; 1: no slabs (the product has the tiles by itself, or does not fit)."""
    if M % 320 or N % 256:
        return 1
    tiles = (M // 320) * (N // 256)
    if tiles >= 256:
        return 1
    ns = int(min(256 // tiles, K // 512, 64))
    if ns < 2 or tiles * ns < 256:
        return 1
    chunks = K // 32
    cps = (chunks + ns - 1) // ns
    return (chunks + cps - 1) // cps


def _s3_pieces(ar, name, x):
    """x [rows, K] fp32 (unit column stride) -> its three bf16 pieces in the MFMA fragment order (an arena buffer per call site)."""
    rows, K = x.shape
    buf = blocks.split3_buffer(ar, name, (rows + 31) // 32 * 32, K)
    return blocks.split3_cvt(x, out=buf)


def lstm_steps_forward_stash(ar, key, lstm, x, Bn, T, split3=False):
    """Like blocks.lstm_steps_forward but keeps gate activations / cell states of every step for backward.
    split3: the input projections as fp32-accurate piece products on the bf16 matrix pipe (split3.hip) where _s3_worth says so."""
    H, L = lstm.hidden_size, lstm.num_layers
    cur = x
    out = None
    for l in range(L):
        xp = ar.get("%s.xp%d" % (key, l), (Bn * T, 8 * H))
        Wst = ops.stacked(lstm.w("weight_ih", l, 0), lstm.w("weight_ih", l, 1)) if split3 else None
        if Wst is not None and cur.stride(1) == 1 and _s3_worth(Bn * T, 8 * H, cur.shape[1]):
            bst = ops.stacked(lstm.w("bias_ih", l, 0), lstm.w("bias_ih", l, 1))
            if bst is None:             # (the two directions' biases are not neighbours in the flat buffer: put them side by side)
                bst = ar.get("%s.s3b%d" % (key, l), (8 * H,))
                ops.copy2d(lstm.w("bias_ih", l, 0).view(1, -1), bst[:4 * H].view(1, -1))
                ops.copy2d(lstm.w("bias_ih", l, 1).view(1, -1), bst[4 * H:].view(1, -1))
            # both directions' W_ih are one [8H, In] matrix in the flat buffer; the weights change every step, so their pieces are
            # made here (5-8 us) -- the activations' too (10-18 us against the ~150-260 us the product saves)
            ap = _s3_pieces(ar, "%s.s3a%d" % (key, l), cur)
            wp = _s3_pieces(ar, "%s.s3w%d" % (key, l), Wst)
            hip.call("split3_gemm", ap, wp, None, xp, xp.stride(0), bst, Bn * T // 32, 8 * H // 32, cur.shape[1], Bn * T, 6, 0)
        else:
            ops.linear_pair(cur, lstm.w("weight_ih", l, 0), lstm.w("weight_ih", l, 1), lstm.w("bias_ih", l, 0), lstm.w("bias_ih", l, 1), xp, 4 * H)
        out = ar.get("%s.out%d" % (key, l), (Bn * T, 2 * H))
        gst = ar.get("%s.gst%d" % (key, l), (2, T, Bn, 4 * H))
        cst = ar.get("%s.cst%d" % (key, l), (2, T, Bn, H))
        blocks.lstm_recurrence(ar, key, lstm, l, xp, out, Bn, T, gst=gst, cst=cst)
        cur = out
    return out


def lstm_steps_backward(ar, key, lstm, x, Bn, T, dout, G, need_dx, split3=False):
    """dout [Bn*T, 2H] (rows b*T+t) -> gradients of every LSTM weight; returns d(x) if need_dx.
    split3: the input-gradient products dX = dgates . W_ih as piece products on the bf16 matrix pipe where _s3_worth says so."""
    H, L = lstm.hidden_size, lstm.num_layers
    dev = x.device
    d_cur = dout
    zeros = ar.get("%s.zrow" % key, (Bn, H), zero=True)
    for l in range(L - 1, -1, -1):
        inp = x if l == 0 else ar.get("%s.out%d" % (key, l - 1), (Bn * T, 2 * H))
        out = ar.get("%s.out%d" % (key, l), (Bn * T, 2 * H))
        gst = ar.get("%s.gst%d" % (key, l), (2, T, Bn, 4 * H))
        cst = ar.get("%s.cst%d" % (key, l), (2, T, Bn, H))
        dg = ar.get("%s.dg" % key, (Bn * T, 8 * H))
        dc = ar.get("%s.dc" % key, (2, Bn, H), zero=True)
        dhrec = ar.get("%s.dhrec" % key, (2, Bn, H))
        wT = ar.get("%s.wT" % key, (2, H, 4 * H))
        for d in range(2):
            hip.call("transpose_batched", lstm.w("weight_hh", l, d), wT[d], 1, 4 * H, H)      # [4H,H] -> [H,4H]
        dg3 = dg.view(Bn, T, 8 * H)
        dcur3 = d_cur.view(Bn, T, 2 * H)
        fused = Bn % 32 == 0 and H % 32 == 0 and os.environ.get("MMEGO_LSTM_BWD_FUSED", "1") != "0"
        for s in range(T - 1, -1, -1):
            t0, t1 = s, T - 1 - s
            last = s == T - 1
            if fused and not last:
                # the product dh_rec = dgates_{s+1} . W_hh of both directions with THIS step's cell backward on its tiles: one launch,
                # dh_rec never leaves the CU (mmego_lstm_bwd_step; same arithmetic as the two launches below)
                p0, p1 = s + 1, T - 1 - (s + 1)          # time indices of the step before in backward order
                hip.call("lstm_bwd_step", Bn, H, dg3[:, p0, :4 * H], dg3[:, p1, 4 * H:], T * 8 * H, wT[0], wT[1],
                         dcur3[:, t0, :H], dcur3[:, t1, H:], T * 2 * H, gst[0, t0], gst[1, t1], cst[0, t0], cst[1, t1],
                         cst[0, t0 - 1] if s > 0 else None, cst[1, t1 + 1] if s > 0 else None, dc[0], dc[1],
                         dg3[:, t0, :4 * H], dg3[:, t1, 4 * H:])
                continue
            hip.call("lstm_cell_backward", 2, Bn, H, dcur3[:, t0, :H], dcur3[:, t1, H:], T * 2 * H,
                     None if last else dhrec[0], None if last else dhrec[1], gst[0, t0], gst[1, t1], cst[0, t0], cst[1, t1],
                     cst[0, t0 - 1] if s > 0 else None, cst[1, t1 + 1] if s > 0 else None, dc[0], dc[1],
                     dg3[:, t0, :4 * H], dg3[:, t1, 4 * H:], T * 8 * H)
            if s > 0 and not fused:
                # dh_{t-1} = dgates_t . W_hh for BOTH directions in one batched launch (few tiles, K = 4H: the K-quartered
                # small-GEMM kernel, in-workgroup K split, no separate reduce; measured 35 us against 42 us as a batched split-K
                # tile product and 2 x 27 us as two split-K calls): this product sits on the serial backward chain.
                # A batch = direction: rows of dg at time t0 (cols 0:4H) / t1 (cols 4H:8H); the batch stride may be negative.
                hip.call("gemm", dg.data_ptr() + 4 * (t0 * 8 * H), T * 8 * H, 1, wT, 1, 4 * H, dhrec, H, 1, None,
                         Bn, H, 4 * H, 2, (t1 - t0) * 8 * H + 4 * H, H * 4 * H, Bn * H, 0, 0, None, 1, 0, None, None)
        Gih = ops.stacked(G(lstm.w("weight_ih", l, 0)), G(lstm.w("weight_ih", l, 1))) if split3 else None
        rows, In = Bn * T, inp.shape[1]
        s3w = Gih is not None and inp.stride(1) == 1 and rows % 32 == 0 and In % 32 == 0 and _s3_worth(8 * H, In, rows)
        if not s3w:
            # weight gradients: the two directions of a product kind as ONE batched launch (ops.grad_weight_pair); the h_{t-1} operands
            # of both directions side by side in one buffer
            hp = ar.get("%s.hp2" % key, (2, Bn * T, H))
            ops.copy2d(out[:Bn * T - 1, :H], hp[0, 1:])                      # h_{t-1}: previous row of the same sequence; zero at t = 0
            ops.copy2d(zeros, hp[0].view(Bn, T * H)[:, :H])
            ops.copy2d(out[1:, H:], hp[1, :Bn * T - 1])                      # reverse direction: h fed into time t came from t+1; zero at T-1
            ops.copy2d(zeros, hp[1].view(Bn, T * H)[:, (T - 1) * H:])
        if s3w:
            # weight gradients dW = dgates^T . X: both operands are read along the ROW axis, so their pieces are those of the
            # transposes (mmego_split3_cvt_t: a coalesced column walk); the contraction is the 10 240 rows and the output has few
            # tiles, so K is cut into slabs (mmego_split3_gemm_slabs) that a streaming sum adds up (mmego_split3_slab_sum)
            dgT = blocks.split3_cvt_t(dg, out=blocks.split3_buffer(ar, "%s.s3dgT" % key, 8 * H, rows))
            inT = blocks.split3_cvt_t(inp, out=blocks.split3_buffer(ar, "%s.s3inT%d" % (key, l), In, rows))
            ns_ih = _s3_nsplit(8 * H, In, rows)
            ws = ar.get("%s.s3ws_ih" % key, (ns_ih * 8 * H * In,))
            hip.call("split3_gemm_slabs", dgT, inT, ws, 8 * H // 32, In // 32, rows, 6, 0, ns_ih)
            hip.call("split3_slab_sum", ws, ns_ih, 8 * H * In, Gih)
            ns_hh = _s3_nsplit(4 * H, H, rows)
            for d in range(2):
                # (h_{t-1} of the forward direction / h_{t+1} of the reverse one, straight from the layer's outputs: no shifted copy)
                hT = blocks.split3_cvt_t(out[:, d * H:(d + 1) * H], out=blocks.split3_buffer(ar, "%s.s3hT%d" % (key, d), H, rows),
                                         shift=-1 if d == 0 else 1, T=T)
                wsh = ar.get("%s.s3ws_hh%d" % (key, d), (ns_hh * 4 * H * H,))
                hip.call("split3_gemm_slabs", dgT[d * 4 * H // 32:], hT, wsh, 4 * H // 32, H // 32, rows, 6, 0, ns_hh)
                hip.call("split3_slab_sum", wsh, ns_hh, 4 * H * H, G(lstm.w("weight_hh", l, d)))
        else:
            ops.grad_weight_pair(dg, 4 * H, inp, G(lstm.w("weight_ih", l, 0)), G(lstm.w("weight_ih", l, 1)))
            ops.grad_weight_pair(dg, 4 * H, hp[0], G(lstm.w("weight_hh", l, 0)), G(lstm.w("weight_hh", l, 1)), X1=hp[1])
        for d in range(2):
            dgd = dg[:, d * 4 * H:(d + 1) * 4 * H]
            ops.colsum(dgd, G(lstm.w("bias_ih", l, d)), out2=G(lstm.w("bias_hh", l, d)))
        if l > 0 or need_dx:
            dinp = ar.get("%s.dx%d" % (key, l), (Bn * T, inp.shape[1]))
            Wst = ops.stacked(lstm.w("weight_ih", l, 0), lstm.w("weight_ih", l, 1))
            if split3 and Wst is not None and _s3_worth(Bn * T, inp.shape[1], 8 * H):
                # dX [rows, In] = dgates [rows, 8H] . W [8H, In]: A = the gate gradients' pieces, "W" operand = W^T [In, 8H]
                In = inp.shape[1]
                dgp = _s3_pieces(ar, "%s.s3dg" % key, dg)
                WT = ar.get("%s.s3wT%d" % (key, l), (In, 8 * H))
                hip.call("transpose_batched", Wst, WT, 1, 8 * H, In)
                wtp = _s3_pieces(ar, "%s.s3wTp%d" % (key, l), WT)
                rows = Bn * T
                ns = _s3_slabs_320(rows, In, 8 * H)
                if ns > 1 and dinp.is_contiguous():
                    # 10 240 x {512, 1024} outputs are 64 / 128 tiles of 320 x 256: with K cut into 4 / 2 slabs the product runs on
                    # s3_gemm_big_kernel (one workgroup per CU, 0.55 instead of 0.34 of the bf16 peak) and a streaming sum adds the
                    # slabs in order
                    ws = ar.get("%s.s3ws_dx%d" % (key, l), (ns * rows * In,))
                    hip.call("split3_gemm_slabs", dgp, wtp, ws, rows // 32, In // 32, 8 * H, 6, 0, ns)
                    hip.call("split3_slab_sum", ws, ns, rows * In, dinp)
                else:
                    hip.call("split3_gemm", dgp, wtp, None, dinp, dinp.stride(0), None, rows // 32, In // 32, 8 * H, rows, 6, 0)
            elif Wst is not None:
                # [W_ih ; W_ih_reverse] back to back in the flat buffer (flat_param_order): one product with K = 8H instead of a
                # product and an accumulating one.  (Against the transposed stack, both operands K-contiguous, the 10 240-row
                # products gain 2-8 % and the 512-row ones of rnn_slow lose 4x: measured, dropped.)
                if not ops.grad_input_slabs(ar, "%s.dxs%d" % (key, l), dg, Wst, dinp):
                    ops.grad_input(dg, Wst, dinp)
            else:
                ops.grad_input(dg[:, :4 * H], lstm.w("weight_ih", l, 0), dinp)
                ops.grad_input(dg[:, 4 * H:], lstm.w("weight_ih", l, 1), dinp, accumulate=True)
            d_cur = dinp
    return d_cur if need_dx else None


def _train_split3(net):
    prec = getattr(net, "train_precision", "fp32")
    if prec not in ("fp32", "split3"):
        raise ValueError("IMUNet.train_precision must be 'fp32' or 'split3', got %r" % (prec,))
    return prec == "split3"


def forward_train(net, imu):
    """IMUNet forward keeping what backward needs.  Returns (R, t)."""
    net.flat()
    ar = net.arena("train")
    B, T, S, Cin = imu.shape
    H, Bn, dev = net.hidden_n, B * T, imu.device
    x = ar.get("x", (Bn * S, Cin))
    ops.copy2d(imu.view(Bn * S, Cin), x)
    h = ar.get("fc1", (Bn * S, H))
    ops.linear(x, net.fc1.weight, net.fc1.bias, h, relu=True)
    s3 = _train_split3(net)
    fast = lstm_steps_forward_stash(ar, "fast", net.rnn_fast, h, Bn, S, split3=s3)
    pooled = ar.get("pooled", (Bn, 2 * H))
    attn = ar.get("attn", (Bn, S))
    blocks.attn_pool_forward(fast, net.attn, Bn, S, 2 * H, pooled, attn)
    slow = lstm_steps_forward_stash(ar, "slow", net.rnn_slow, pooled, B, T, split3=s3)
    y = ar.get("y", (Bn, 9))
    ops.linear(slow, net.fc2.weight, net.fc2.bias, y)
    R = torch.empty((B, T, 3, 3), dtype=torch.float32, device=dev)
    t = torch.empty((B, T, 3), dtype=torch.float32, device=dev)
    hip.call("imu_head", y, Bn, R, t)
    net._saved = (B, T, S, Cin)
    return R, t


def backward(net, dR, dt):
    ar = net.arena("train")
    B, T, S, Cin = net._saved
    H, Bn = net.hidden_n, B * T
    G = net._flat.grad
    y = ar.get("y", (Bn, 9))
    dy = ar.get("dy", (Bn, 9))
    hip.call("imu_head_backward", y, dR.contiguous(), dt.contiguous(), Bn, dy)
    slow = ar.get("slow.out%d" % (net.rnn_slow.num_layers - 1), (Bn, 2 * H))
    dslow = ar.get("dslow", (Bn, 2 * H))
    blocks.linear_backward(dy, slow, net.fc2, G, dslow)
    pooled = ar.get("pooled", (Bn, 2 * H))
    s3 = _train_split3(net)
    dpooled = lstm_steps_backward(ar, "slow", net.rnn_slow, pooled, B, T, dslow, G, True, split3=s3)
    fast = ar.get("fast.out%d" % (net.rnn_fast.num_layers - 1), (Bn * S, 2 * H))
    attn = ar.get("attn", (Bn, S))
    dfast = ar.get("dfast", (Bn * S, 2 * H))
    blocks.attn_pool_backward(ar, "pool", fast, net.attn, attn, dpooled, Bn, S, 2 * H, dfast, G)
    h = ar.get("fc1", (Bn * S, H))
    dh = lstm_steps_backward(ar, "fast", net.rnn_fast, h, Bn, S, dfast, G, True, split3=s3)
    ops.relu_mask_(dh, h)
    x = ar.get("x", (Bn * S, Cin))
    blocks.linear_backward(dh, x, net.fc1, G)
    # fc3 is never used in forward (Q7): its gradient stays zero


class ImuBridge(torch.autograd.Function):
    """One autograd node for the whole IMU_Net: forward/backward are the kernel pipelines above."""

    @staticmethod
    def forward(ctx, net, imu, *params):
        ctx.net = net
        return forward_train(net, imu)

    @staticmethod
    def backward(ctx, dR, dt):
        net = ctx.net
        dev = next(net.parameters()).device
        B, T = net._saved[0], net._saved[1]
        dR = dR if dR is not None else torch.zeros((B, T, 3, 3), device=dev)
        dt = dt if dt is not None else torch.zeros((B, T, 3), device=dev)
        backward(net, dR.to(torch.float32), dt.to(torch.float32))
        net.flat().bind_grads()
        return (None, None) + (None,) * len(net._flat.params)
