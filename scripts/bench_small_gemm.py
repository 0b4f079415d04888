"""Small products of the stage tails (weight gradients with their bias sums, input gradients, head layers) as replayed graphs:
us per call.  usage: python scripts/bench_small_gemm.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmego_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)


def timed(fn, label, n=20):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        with torch.cuda.graph(g, stream=st):
            for _ in range(n):
                fn()
    torch.cuda.synchronize()
    for _ in range(3):
        g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    print("%-60s %.2f us" % (label, e0.elapsed_time(e1) / (10 * n) * 1e3))


for rows, N, K in ((7680, 96, 3), (7680, 192, 32), (7680, 384, 64), (7680, 64, 128), (512, 87, 128), (512, 128, 128), (512, 64, 173),
                   (512, 42, 64)):
    dY, X = torch.randn(rows, N, device=dev), torch.randn(rows, K, device=dev)
    dW, db, W = torch.empty(N, K, device=dev), torch.empty(N, device=dev), torch.randn(N, K, device=dev) * 0.1
    dX = torch.empty(rows, K, device=dev)
    timed(lambda: ops.grad_weight(dY, X, dW, db=db, prefer_fused=True), "dW+db  rows %5d  N %3d  K %3d" % (rows, N, K))
    timed(lambda: ops.grad_input(dY, W, dX), "dX     rows %5d  N %3d  K %3d" % (rows, N, K))
    out = torch.empty(rows, N, device=dev)
    timed(lambda: ops.linear(X, W, db, out), "y=xW^T rows %5d  N %3d  K %3d" % (rows, N, K))
