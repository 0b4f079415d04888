"""How much does one tiny kernel cost inside a replayed HIP graph vs eagerly? (calibrates the value of kernel fusion)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmego_amd import hip  # noqa: E402

dev = torch.device("cuda:0")
x = torch.zeros(256, device=dev)
y = torch.zeros(256, device=dev)
N = 400


def body():
    for _ in range(N):
        hip.call("add", x, y, x, 256)


body()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    body()
torch.cuda.synchronize()
print("eager: %.2f us per tiny kernel" % ((time.perf_counter() - t0) / 5 / N * 1e6))
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    body()
g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    g.replay()
torch.cuda.synchronize()
print("graph: %.2f us per tiny kernel" % ((time.perf_counter() - t0) / 20 / N * 1e6))
