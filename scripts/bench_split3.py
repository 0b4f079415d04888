"""The split3 kernels alone (mmego_amd/csrc/split3.hip; IMU_Net's rnn_fast products at the bench shape B=64, T=8: 512 rows x 20 samples,
H = 512): projection GEMM 10240 x 4096 x {512, 1024} on 6 / 9 piece products and both tile shapes against the native fp32 product
(gemm_tile), one recurrent step (both directions) against the fp32 step launches, and the whole IMU_Net forward in the three modes
as a replayed HIP graph."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmego_amd import blocks, hip, nets, ops  # noqa: E402

dev = torch.device("cuda:0")
hip.lib()


def timeit(fn, n=20, warm=3, fork=False):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with (ops.capture(g) if fork else torch.cuda.graph(g)):          # (ops.capture: side streams may be forked -- two-chain recurrences)
        for _ in range(n):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (5 * n) * 1e3          # us


Bn, S, H = 512, 20, 512
M, N = Bn * S, 8 * H
torch.manual_seed(0)
for K in (512, 1024):
    A = torch.randn(M, K, device=dev).relu_()
    W = torch.randn(N, K, device=dev) * 0.04
    bias = torch.randn(N, device=dev)
    Ap, Wp = blocks.split3_cvt(A), blocks.split3_cvt(W)
    Cf = torch.empty(M * N, device=dev)
    C = torch.empty(M, N, device=dev)
    flop = 2.0 * M * N * K
    us = timeit(lambda: ops.linear_pair(A, W[:4 * H], W[4 * H:], bias[:4 * H], bias[4 * H:], C, 4 * H))
    print("K=%4d native fp32 (gemm_tile, both directions in one launch): %7.1f us  %6.1f TF" % (K, us, flop / us / 1e6))
    for wm in (2, 4):
        for nprod in (6, 9):
            us = timeit(lambda: hip.call("split3_gemm", Ap, Wp, Cf, None, 0, bias, M // 32, N // 32, K, 0, nprod, wm))
            print("K=%4d split3 wm=%d nprod=%d: %7.1f us  %6.1f TF fp32-equivalent, %6.1f TF of bf16 MFMA (%.2f of 2.5 PF)"
                  % (K, wm, nprod, us, flop / us / 1e6, nprod * flop / us / 1e6, nprod * flop / us / 1e6 / 2500.0))

# one recurrent step, both directions
lstm = blocks.LstmParams(H, H, 2, dropout=0.0, bidirectional=True).to(dev)
x = torch.randn(Bn * S, H, device=dev).relu_()
ar32, ar3 = ops.Arena(dev), ops.Arena(dev)
with torch.no_grad():
    for name, fn in (("fp32 (gemm_tile + lstm_step, two chains when alone)", lambda: blocks.lstm_steps_forward(ar32, "t", lstm, x, Bn, S)),
                     ("split3 nprod=6", lambda: blocks.lstm_steps_forward_split3(ar3, "t", lstm, x, Bn, S, nprod=6)),
                     ("split3 nprod=9", lambda: blocks.lstm_steps_forward_split3(ar3, "t", lstm, x, Bn, S, nprod=9))):
        fn()
        torch.cuda.synchronize()
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record()
        torch.cuda.synchronize()
        print("BiLSTM(512) x 2 layers, 512 rows x 20 samples, eager: %-52s %8.1f us" % (name, e0.elapsed_time(e1) / 10 * 1e3))

# the recurrent step alone (layer 0 of the split path): replayed graph of the 20 steps
W3 = blocks.lstm_split3_weights(lstm)
Bp, nrb, S2 = Bn, Bn // 32, 2 * H // 16
xpf = ar3.get("t.s3xpf0", (S * Bp * 8 * H,))
O = blocks.split3_buffer(ar3, "t.s3h0", S * Bp, 2 * H)
c = ar3.get("t.c", (2, Bn, H))
o_p = O.data_ptr()
win = lambda t, d: o_p + 2 * ((t * nrb * S2 + d * (H // 16)) * 3 * 512)
for nprod in (6, 9):
    def steps():
        for s in range(S):
            t0, t1 = s, S - 1 - s
            hip.call("split3_step", 2, Bn, H, int(s == 0), win(t0 - 1, 0) if s > 0 else None, win(t1 + 1, 1) if s > 0 else None, S2 * 3,
                     W3[0][2], W3[0][3], xpf, t0 * nrb, t1 * nrb, None, None, 0, win(t0, 0), win(t1, 1), S2 * 3, c[0], c[1], nprod, 0)
    us = timeit(steps, n=2)
    print("split3 recurrent step nprod=%d (both directions per launch, 20 steps replayed): %6.2f us per timestep (%.1f TF fp32-equivalent)"
          % (nprod, us / S, 2.0 * 2 * Bn * 4 * H * H * (S - 1) / us / 1e6))

# whole IMU_Net forward, replayed graph
torch.manual_seed(1)
imu = nets.IMUNet(15, 9, 512, 2, True, 0.1).to(dev).eval()
inp = torch.randn(64, 8, 20, 15, device=dev)
for prec in ("fp32", "split3", "bf16"):
    imu.precision = prec
    for chains in (False, True):
        with torch.no_grad():
            def fwd():
                with blocks.two_chains(chains):
                    return imu(inp)
            us = timeit(fwd, n=4, fork=chains)
        print("IMU_Net forward B=64 T=8, precision=%-6s (%s, graph replay): %8.1f us"
              % (prec, "two chains per recurrence" if chains else "one launch per timestep ", us))
assert blocks.seq_xcd_errors() == 0
