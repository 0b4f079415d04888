"""ST-GCN temporal convolution kernels (gcn.hip) alone: forward with BatchNorm+ReLU on load, input gradient, weight gradient,
at the training shape (B=64, T=8: 7680 rows) and at config 5's (B=2048, T=16), as replayed HIP graphs of 20 launches each.
Usage (GPU box): python scripts/bench_tconv.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmego_amd import hip  # noqa: E402

dev = torch.device("cuda:0")
V, taps = 15, 9


def timed(fn, reps=20, iters=10):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            for _ in range(reps):
                fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (reps * iters) * 1e3


for B, T in ((64, 8), (2048, 16)):
    rows = B * T * V
    for C in (32, 64, 128):
        x = torch.randn(rows, C, device=dev)
        dy = torch.randn(rows, C, device=dev)
        W = torch.randn(C, C, taps, 1, device=dev) * 0.05
        bias = torch.randn(C, device=dev)
        st = torch.stack((torch.zeros(C), torch.ones(C), torch.ones(C), torch.zeros(C))).to(dev).contiguous()
        wp = torch.empty(2, W.numel(), device=dev)
        hip.call("tconv_pack", W, C, C, taps, 2, wp)
        y, act, dx, dW = torch.empty(rows, C, device=dev), torch.empty(rows, C, device=dev), torch.empty(rows, C, device=dev), torch.empty_like(W)
        nsp = hip.lib().mmego_tconv_wgrad_nsplit(B, T, V, C, C, taps)
        ws = torch.empty(nsp * W.numel(), device=dev)
        flop = 2.0 * rows * C * C * taps
        t_pack = timed(lambda: hip.call("tconv_pack", W, C, C, taps, 2, wp))
        t_f = timed(lambda: hip.call("tconv", x, C, st, wp[0], bias, y, C, act, B, T, V, C, C, taps))
        t_e = timed(lambda: hip.call("tconv", x, C, st, wp[0], bias, y, C, None, B, T, V, C, C, taps))
        t_dx = timed(lambda: hip.call("tconv", dy, C, None, wp[1], None, dx, C, None, B, T, V, C, C, taps))
        t_dw = timed(lambda: hip.call("tconv_wgrad", dy, C, act, C, ws, dW, 0, B, T, V, C, C, taps))
        print("rows %7d C %3d: pack %5.1f us | forward (+act out) %7.1f us %5.1f TF | forward %7.1f us | input grad %7.1f us %5.1f TF | "
              "weight grad (+reduce, %d splits) %7.1f us %5.1f TF" % (rows, C, t_pack, t_f, flop / t_f * 1e-6, t_e, t_dx, flop / t_dx * 1e-6,
                                                                    nsp, t_dw, flop / t_dw * 1e-6), flush=True)
