"""IMU_Net stage-1 training on the HIP path: stashing forward, backward through the two BiLSTM(512) stacks,
geodesic + position loss.  Reference: Net/IMU_Net.py:67-94, Processor/Train/Train_IMU.py:21-34,114-149.

The recurrent products of the backward pass run as split-K mmego_gemm calls (one per timestep and direction against a
per-layer transposed W_hh); input and weight gradients of the projections run on the large-tile kernels in their
natural operand orientation (dX = dY . W: NN, dW = dY^T . X: TN).
"""
import os

import torch

from . import blocks, hip, ops


def lstm_steps_forward_stash(ar, key, lstm, x, Bn, T):
    """Like blocks.lstm_steps_forward but keeps gate activations / cell states of every step for backward."""
    H, L = lstm.hidden_size, lstm.num_layers
    cur = x
    out = None
    for l in range(L):
        xp = ar.get("%s.xp%d" % (key, l), (Bn * T, 8 * H))
        ops.linear_pair(cur, lstm.w("weight_ih", l, 0), lstm.w("weight_ih", l, 1), lstm.w("bias_ih", l, 0), lstm.w("bias_ih", l, 1), xp, 4 * H)
        out = ar.get("%s.out%d" % (key, l), (Bn * T, 2 * H))
        gst = ar.get("%s.gst%d" % (key, l), (2, T, Bn, 4 * H))
        cst = ar.get("%s.cst%d" % (key, l), (2, T, Bn, H))
        blocks.lstm_recurrence(ar, key, lstm, l, xp, out, Bn, T, gst=gst, cst=cst)
        cur = out
    return out


def lstm_steps_backward(ar, key, lstm, x, Bn, T, dout, G, need_dx):
    """dout [Bn*T, 2H] (rows b*T+t) -> gradients of every LSTM weight; returns d(x) if need_dx."""
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
        # weight gradients: the two directions of a product kind as ONE batched launch (ops.grad_weight_pair); the h_{t-1} operands
        # of both directions side by side in one buffer
        hp = ar.get("%s.hp2" % key, (2, Bn * T, H))
        ops.copy2d(out[:Bn * T - 1, :H], hp[0, 1:])                      # h_{t-1}: previous row of the same sequence; zero at t = 0
        ops.copy2d(zeros, hp[0].view(Bn, T * H)[:, :H])
        ops.copy2d(out[1:, H:], hp[1, :Bn * T - 1])                      # reverse direction: h fed into time t came from t+1; zero at T-1
        ops.copy2d(zeros, hp[1].view(Bn, T * H)[:, (T - 1) * H:])
        ops.grad_weight_pair(dg, 4 * H, inp, G(lstm.w("weight_ih", l, 0)), G(lstm.w("weight_ih", l, 1)))
        ops.grad_weight_pair(dg, 4 * H, hp[0], G(lstm.w("weight_hh", l, 0)), G(lstm.w("weight_hh", l, 1)), X1=hp[1])
        for d in range(2):
            dgd = dg[:, d * 4 * H:(d + 1) * 4 * H]
            ops.colsum(dgd, G(lstm.w("bias_ih", l, d)), out2=G(lstm.w("bias_hh", l, d)))
        if l > 0 or need_dx:
            dinp = ar.get("%s.dx%d" % (key, l), (Bn * T, inp.shape[1]))
            Wst = ops.stacked(lstm.w("weight_ih", l, 0), lstm.w("weight_ih", l, 1))
            if Wst is not None:
                # [W_ih ; W_ih_reverse] back to back in the flat buffer (flat_param_order): one product with K = 8H instead of a
                # product and an accumulating one.  (Against the transposed stack, both operands K-contiguous, the 10 240-row
                # products gain 2-8 % and the 512-row ones of rnn_slow lose 4x: measured, dropped.)
                ops.grad_input(dg, Wst, dinp)
            else:
                ops.grad_input(dg[:, :4 * H], lstm.w("weight_ih", l, 0), dinp)
                ops.grad_input(dg[:, 4 * H:], lstm.w("weight_ih", l, 1), dinp, accumulate=True)
            d_cur = dinp
    return d_cur if need_dx else None


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
    fast = lstm_steps_forward_stash(ar, "fast", net.rnn_fast, h, Bn, S)
    pooled = ar.get("pooled", (Bn, 2 * H))
    attn = ar.get("attn", (Bn, S))
    blocks.attn_pool_forward(fast, net.attn, Bn, S, 2 * H, pooled, attn)
    slow = lstm_steps_forward_stash(ar, "slow", net.rnn_slow, pooled, B, T)
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
    dpooled = lstm_steps_backward(ar, "slow", net.rnn_slow, pooled, B, T, dslow, G, True)
    fast = ar.get("fast.out%d" % (net.rnn_fast.num_layers - 1), (Bn * S, 2 * H))
    attn = ar.get("attn", (Bn, S))
    dfast = ar.get("dfast", (Bn * S, 2 * H))
    blocks.attn_pool_backward(ar, "pool", fast, net.attn, attn, dpooled, Bn, S, 2 * H, dfast, G)
    h = ar.get("fc1", (Bn * S, H))
    dh = lstm_steps_backward(ar, "fast", net.rnn_fast, h, Bn, S, dfast, G, True)
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
