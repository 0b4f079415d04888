"""r06: the fp32 projection product of IMU_Net's BiLSTM layers (both directions batched: 2 x (10240 x 2048 x K)) on the 320 x 256-tile
LDS-DMA kernel (gemm_tile_big_kernel) -- run this script twice, with MMEGO_GEMM_BIG=0 (the 128 x 128 persistent walk) and without: result
against float64 (both must be fp32 products), checksum of C for a cross-process comparison, time in a replayed graph."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmego_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
M, N = 10240, 2048
g_ = torch.Generator().manual_seed(4)
for K in (512, 1024):
    A = torch.randn(M, K, generator=g_).to(dev)
    W = (torch.randn(2 * N, K, generator=g_) * 0.05).to(dev)
    b = torch.randn(2 * N, generator=g_).to(dev)
    C = torch.full((M, 2 * N), float("nan"), device=dev)
    run = lambda: ops.linear_pair(A, W[:N], W[N:], b[:N], b[N:], C, N)
    run()
    torch.cuda.synchronize()
    rows = torch.tensor([0, 1, 31, 32, 159, 160, 319, 320, 5000, 10239], device=dev)
    ref = A[rows].double() @ W.double().t() + b.double()
    err = float((C[rows].double() - ref).abs().max())
    assert torch.isfinite(C).all()
    g = torch.cuda.CUDAGraph()
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        with torch.cuda.graph(g, stream=st):
            for _ in range(10):
                run()
    torch.cuda.synchronize()
    g.replay()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 50 * 1e3)
    us = sorted(ts)[2]
    print("MMEGO_GEMM_BIG=%s  2 x (M%d N%d K%d): %.1f us per launch in a replayed graph = %.1f TFLOP/s (%.3f of 157.3); max |C - float64| on 10 rows %.3g; "
          "sum(C) %.6f sum(|C|) %.4f" % (os.environ.get("MMEGO_GEMM_BIG", "1"), M, N, K, us, 4.0 * M * N * K / us / 1e6, 4.0 * M * N * K / us / 1e6 / 157.3,
                                      err, float(C.double().sum()), float(C.double().abs().sum())), flush=True)
