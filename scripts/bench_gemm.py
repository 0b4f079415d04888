"""Micro-benchmark of mmego_gemm (NT, contiguous): us per launch and TFLOP/s for the IMU projection shapes."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmego_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
shapes = [(10240, 2048, 512), (10240, 2048, 1024), (512, 2048, 1024), (512, 256, 128), (512, 128, 128), (7680, 128, 1152)]
for M, N, K in shapes:
    A = torch.randn(M, K, device=dev)
    W = torch.randn(N, K, device=dev) * 0.05
    b = torch.randn(N, device=dev)
    C = torch.empty(M, N, device=dev)
    for _ in range(3):
        ops.linear(A, W, b, C)
    err = 0.0
    for sl in (slice(0, 256), slice(M - 256, M)):       # first rows and the tail-balanced last rows
        ref = A[sl].double() @ W.double().t() + b.double()
        err = max(err, (C[sl].double() - ref).abs().max().item())
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 50
    e0.record()
    for _ in range(n):
        ops.linear(A, W, b, C)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    print("M%d N%d K%d: %.1f us, %.1f TFLOP/s, max err %.2e" % (M, N, K, us, 2.0 * M * N * K / us / 1e6, err))
    if M == 10240:      # both directions of a BiLSTM layer as one batched product (what IMU_Net's forward launches)
        W2 = torch.cat((W, torch.randn(N, K, device=dev) * 0.05))
        b2 = torch.cat((b, torch.randn(N, device=dev)))
        C2 = torch.empty(M, 2 * N, device=dev)
        for _ in range(3):
            ops.linear_pair(A, W2[:N], W2[N:], b2[:N], b2[N:], C2, N)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(n):
            ops.linear_pair(A, W2[:N], W2[N:], b2[:N], b2[N:], C2, N)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / n * 1e3
        ref2 = A[:128].double() @ W2.double().t() + b2.double()
        print("   pair 2 x (M%d N%d K%d): %.1f us, %.1f TFLOP/s, max err %.2e" % (M, N, K, us, 4.0 * M * N * K / us / 1e6,
                                                                               (C2[:128].double() - ref2).abs().max().item()))
