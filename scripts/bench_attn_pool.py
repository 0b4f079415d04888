"""Attention-pooling forward (softmax over the P rows of a group, weighted row sum) at the PointNet and IMU_Net shapes:
us per launch and achieved HBM rate."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmego_amd import hip  # noqa: E402

dev = torch.device("cuda:0")
hip.lib()
for G, P, C in ((512, 128, 64), (512, 20, 1024), (32768, 256, 64), (32768, 20, 1024)):
    X = torch.randn(G * P, C, device=dev)
    w = torch.randn(C, device=dev) * 0.1
    b = torch.zeros(1, device=dev)
    vec = torch.empty(G, C, device=dev)
    attn = torch.empty(G, P, device=dev)
    def call():
        hip.call("attn_pool_forward", X, w, b, G, P, C, vec, attn)
    for _ in range(3):
        call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    e0.record()
    for _ in range(n):
        call()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    sc = (X.view(G, P, C) @ w + b).softmax(dim=1)
    ref = (X.view(G, P, C) * sc.unsqueeze(-1)).sum(1)
    print("G=%d P=%d C=%d: %.1f us, %.2f TB/s, max err vs torch %.1e" % (G, P, C, us, X.numel() * 4 / us / 1e6, (vec - ref).abs().max().item()))
