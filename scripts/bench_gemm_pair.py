"""The GEMM launches of IMU_Net's forward exactly as the U+L step issues them: both directions' input projections of a BiLSTM
layer as one batched product, 2 x (10240 x 2048 x K), K = 512 (layer 0) and 1024 (layer 1).  Used for the PMC passes."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmego_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
M, N = 10240, 2048
for K in (512, 1024):
    A = torch.randn(M, K, device=dev)
    W = torch.randn(2 * N, K, device=dev) * 0.05
    b = torch.randn(2 * N, device=dev)
    C = torch.empty(M, 2 * N, device=dev)
    for _ in range(3):
        ops.linear_pair(A, W[:N], W[N:], b[:N], b[N:], C, N)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 25
    e0.record()
    for _ in range(n):
        ops.linear_pair(A, W[:N], W[N:], b[:N], b[N:], C, N)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    # the same launch ten times in a replayed HIP graph: no host launch cost in the figure (what the U+L step sees)
    g = torch.cuda.CUDAGraph()
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        with torch.cuda.graph(g, stream=st):
            for _ in range(10):
                ops.linear_pair(A, W[:N], W[N:], b[:N], b[N:], C, N)
    torch.cuda.synchronize()
    g.replay()
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    usg = e0.elapsed_time(e1) / 50 * 1e3
    print("pair 2 x (M%d N%d K%d): %.1f us eager, %.1f us in a replayed graph = %.1f TFLOP/s (%.3f of 157.3)"
          % (M, N, K, us, usg, 4.0 * M * N * K / usg / 1e6, 4.0 * M * N * K / usg / 1e6 / 157.3))
