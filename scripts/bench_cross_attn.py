"""mmego_cross_attn_forward alone (pool.hip) at the U+L step's shape (512 frames) and config 5's (32 768 frames): us per launch in a
replayed graph, GB/s on the algorithmic bytes (Q + K + V read, O + P written), a checksum for cross-build comparison.
usage: python scripts/bench_cross_attn.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmego_amd import hip
dev = torch.device("cuda:0")
hip.lib()
for F in (512, 4096, 32768):
    g = torch.Generator().manual_seed(1)
    Q = torch.randn(F * 64, 64, generator=g).to(dev)
    KV = torch.randn(F * 15, 128, generator=g).to(dev)
    both = torch.zeros(F * 64, 128, device=dev)
    P = torch.zeros(F * 64, 15, device=dev)
    run = lambda: hip.call("cross_attn_forward", Q, KV, KV[:, 64:], F, 0.125, both[:, 64:], 128, P, 128)
    for _ in range(3): run()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(10): run()
    gr.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): gr.replay()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    nbytes = F * (64 * 64 * 4 * 2 + 15 * 64 * 4 * 2 + 64 * 15 * 4)
    print("cross_attn_forward F=%6d: %.1f us per launch, %.0f GB/s algorithmic; sum |O| %.6f sum P %.3f" % (F, us, nbytes / us / 1e3, float(both.double().abs().sum()), float(P.double().sum())))
