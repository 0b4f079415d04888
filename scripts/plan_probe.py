"""Probe of plan.StepPlan against a one-graph capture: (a) one BiLSTM layer's two-chain recurrence alone, (b) the U+L step;
host enqueue time and device time per replay.  usage: python scripts/plan_probe.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from mmego_amd import blocks, hip, ops  # noqa: E402
from mmego_amd.plan import StepPlan  # noqa: E402
from mmego_amd.train_step import ConcurrentStages, StageStep  # noqa: E402

dev = torch.device("cuda:0")
hip.lib()


def timeit(replay, n=20):
    for _ in range(3):
        replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        replay()
    e1.record()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    return (t1 - t0) / n * 1e3, e0.elapsed_time(e1) / n


Bn, H, T = 512, 512, 20
lstm = blocks.LstmParams(H, H, 1).to(dev)
xp = torch.randn(Bn * T, 8 * H, device=dev) * 0.1
ar = ops.Arena(dev)
out = ar.get("out", (Bn * T, 2 * H))


def rec():
    with blocks.two_chains(True):
        blocks.lstm_recurrence(ar, "k", lstm, 0, xp, out, Bn, T)


rec()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
st = torch.cuda.Stream()
st.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(st):
    with ops.capture(g, stream=st):
        rec()
plan = StepPlan().record(rec).build()
print("recurrence:", plan.describe())
h, d = timeit(g.replay)
print("  one graph, two branches : host %.3f ms, device %.1f us per layer = %.2f us per timestep" % (h, d * 1e3, d * 1e3 / T))
h, d = timeit(plan.replay)
print("  StepPlan (graph per chain): host %.3f ms, device %.1f us per layer = %.2f us per timestep" % (h, d * 1e3, d * 1e3 / T))
h, d = timeit(plan.run_eagerly)
print("  same plan, eager launches : host %.3f ms, device %.1f us per layer = %.2f us per timestep" % (h, d * 1e3, d * 1e3 / T))

