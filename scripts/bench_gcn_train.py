"""The fused ST-GCN training kernels alone (Lower_Net's KeyEncoder at the bench shape B=64 T=8): GPU time per entry point of one
forward + backward of the GCN part (event pairs around every launch, eager), for the fused path and -- nets._GCN_FUSED = False -- the
launch chain.  (The MMEGO_GCN_DBG phase mask of r04 -- timing by elimination -- was removed from the kernel in r05: git history.)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmego_amd import hip, nets  # noqa: E402

dev = torch.device("cuda:0")
B, T = 64, 8
torch.manual_seed(3)
net = nets.LowerNet(64).to(dev).train()
net.flat()
ar = net.arena("train")
up = torch.randn(B * T, 45, device=dev)
ar.get("up", (B * T, 45)).copy_(up)
G = net._flat.grad
dk = torch.randn(B * T * 15, 64, device=dev)


def body():
    net._gcn_forward(ar, ar.get("up", (B * T, 45)), B, T, True)
    if net._gcn_was_fused:
        from mmego_amd import ops
        slabs = ops.SlabList()
        net._gcn_backward_fused(ar, dk, B, T, G, slabs)
        slabs.run()
    else:
        net._gcn_backward(ar, dk, B, T, G)


rec = {}
orig = hip.call


def timed(name, *a):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); orig(name, *a); e1.record()
    rec.setdefault(name, []).append((e0, e1))


for _ in range(3):
    body()
torch.cuda.synchronize()
hip.call = timed
iters = 5
for _ in range(iters):
    body()
torch.cuda.synchronize()
hip.call = orig
tot = 0.0
for n, v in sorted(rec.items(), key=lambda kv: -sum(a.elapsed_time(b) for a, b in kv[1])):
    us = [a.elapsed_time(b) * 1e3 for a, b in v]
    per = len(us) // iters
    tot += sum(us) / iters
    print("%-22s %2d launches  %7.1f us   each: %s" % (n, per, sum(us) / iters, " ".join("%.1f" % (sum(us[k::per]) / iters) for k in range(per))))
# every distinct launch of the body on its own: 20 back-to-back repeats in one replayed graph (the launches are idempotent), i.e. the
# kernel's duration + one graph-node dispatch, without the host launch latency an eager event pair carries
calls = []
hip.call = lambda name, *a: (calls.append((name, a)), orig(name, *a))[1]
body()
torch.cuda.synchronize()
hip.call = orig
print("per launch, 20 repeats in a replayed graph:")
for name, a in calls:
    gk = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gk):
        for _ in range(20):
            orig(name, *a)
    gk.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        gk.replay()
    e1.record()
    torch.cuda.synchronize()
    print("   %-20s %6.1f us" % (name, e0.elapsed_time(e1) / 100 * 1e3))
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    body()
for _ in range(3):
    g.replay()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50):
    g.replay()
e1.record()
torch.cuda.synchronize()
print("fused=%s: sum of launches %.1f us; replayed graph %.1f us per forward+backward" %
      (net._gcn_was_fused, tot, e0.elapsed_time(e1) / 50 * 1e3))
