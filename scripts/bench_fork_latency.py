"""r06: does the SIDE chain of a two-chain recurrence start late when the graph has been on ONE chain for a while?  (The kernel trace of a
step -- profiles/r06_timeline_sequential.txt -- shows layer 0's second chain of rnn_fast starting 5-8 timesteps (70-115 us) behind the
first, layer 1's not.)  A replayed graph of [batched projection product (~320 us on one chain) -> BiLSTM(512) recurrence as two chains],
(a) as the nets issue it, (b) with the side stream forked BEFORE the product (one trivial launch on it beside the product), so that the
branch exists -- and whatever its start costs is paid -- while the product runs."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmego_amd import blocks, hip, ops  # noqa: E402

dev = torch.device("cuda:0")
Bn, H, T = 512, 512, 20
M, N = Bn * T, 4 * H
lstm = blocks.LstmParams(H, H, 1).to(dev)
A = torch.randn(M, H, device=dev)
W = torch.randn(2 * N, H, device=dev) * 0.05
b = torch.randn(2 * N, device=dev)
xp = torch.empty(M, 2 * N, device=dev)
dummy = torch.zeros(256, device=dev)
ar = ops.Arena(dev)
o2 = ar.get("out", (Bn * T, 2 * H))


def body(early_fork, layers=2):
    for _ in range(layers):
        cur = torch.cuda.current_stream()
        if early_fork:
            side = blocks._side_stream(cur)
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                hip.call("fill", dummy, 256, 0.0)
        ops.linear_pair(A, W[:N], W[N:], b[:N], b[N:], xp, N)
        blocks.lstm_recurrence(ar, "k", lstm, 0, xp, o2, Bn, T)


for rep in range(2):
    for early in (False, True):
        with blocks.two_chains(True):
            g = torch.cuda.CUDAGraph()
            st = torch.cuda.Stream()
            st.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(st):
                body(early)
                with ops.capture(g, stream=st):
                    body(early)
        torch.cuda.synchronize()
        for _ in range(5):
            g.replay()
        ts = []
        for _ in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                g.replay()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 10 * 1e3)
        print("two x [projection + two-chain recurrence], side stream forked %s: %.1f us per replay (median of 7 x 10)" % (
            "BEFORE the product" if early else "after it (as shipped)", sorted(ts)[3]), flush=True)
