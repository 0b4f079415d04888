"""r06: the BiLSTM weight-gradient products of stage-1 IMU_Net training (ops.grad_weight_pair: dW = dgates^T X for both directions as one
batched TN product, 2 x (2048 x {512, 1024}) outputs over K = 10 240 rows) by split-K factor: us per launch in a replayed graph incl. the
slab sum, result against float64.  usage: python scripts/bench_wgrad_split.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmego_amd import hip, ops
dev = torch.device("cuda:0")
hip.lib()
rows, ncol = 10240, 2048
g = torch.Generator().manual_seed(2)
dY = (torch.randn(rows, 2 * ncol, generator=g) * 0.1).to(dev)
for K in (512, 1024):
    X2 = torch.randn(2, rows, K, generator=g).to(dev)
    W = torch.zeros(2, ncol, K, device=dev)
    ref = torch.stack([dY[:, d * ncol:(d + 1) * ncol][:, :64].double().t() @ X2[d].double() for d in range(2)]).cpu()
    for nsplit in (1, 2, 4, 8):
        ws = ops.scratch(dev, nsplit * 2 * ncol * K) if nsplit > 1 else None
        xdist = (X2[1].data_ptr() - X2[0].data_ptr()) // 4
        dist = (W[1].data_ptr() - W[0].data_ptr()) // 4
        run = lambda: hip.call("gemm", dY, 1, dY.stride(0), X2[0], X2[0].stride(0), 1, W[0], K, 1, None, ncol, K, rows, 2, ncol, xdist, dist, 0, 0, ws, nsplit, 0, None, None)
        for _ in range(3): run()
        torch.cuda.synchronize()
        err = float((W[:, :64].double().cpu() - ref).abs().max())
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            for _ in range(10): run()
        gr.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): gr.replay()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 50 * 1e3
        print("dW pair 2 x (%d x %d), K = %d rows, nsplit %d: %.1f us per launch (%.1f TFLOP/s), max |dW - float64| on 64 rows %.2e" % (ncol, K, rows, nsplit, us, 2.0 * 2 * ncol * K * rows / us / 1e6, err))
