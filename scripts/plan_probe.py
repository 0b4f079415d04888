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

# the U+L step: ConcurrentStages against OverlappedStages (no join at the end of a step), device time per step over 60 steps
from mmego_amd.train_step import OverlappedStages  # noqa: E402


def engine(cls):
    imu, upper, lower, upper_frozen = bench.build_hip_models(dev)
    imu_l = bench.clone_imu(imu, dev)
    x, imu_in, body, target = bench.synth_batch(1234, dev)
    su = StageStep("upper", upper, imu, lr=3e-5)
    sl = StageStep("lower", lower, imu_l, upper_frozen=upper_frozen, lr=3e-5)
    su.bind(x, imu_in, body, target); sl.bind(x, imu_in, body, target)
    eng = cls([su, sl])
    eng.prepare()
    return eng


for cls in (ConcurrentStages, OverlappedStages, ConcurrentStages, OverlappedStages):
    eng = engine(cls)
    for _ in range(5):
        eng.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(60):
        eng.step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("%s: host %.3f ms, %.3f ms per step" % (cls.__name__, (t1 - t0) / 60 * 1e3, (t2 - t0) / 60 * 1e3))
    del eng
