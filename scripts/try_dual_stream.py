"""Experiment: run the Upper-stage and Lower-stage bodies (independent programs in the reference) as two concurrent
branches of one HIP graph.  Prints ms per U+L step for the sequential and the concurrent arrangement."""
import os
import sys
import time

import faulthandler

import torch

faulthandler.enable()

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from mmego_amd import hip, nets  # noqa: E402
from mmego_amd.train_step import StageStep  # noqa: E402

dev = torch.device("cuda:0")
hip.lib()
imu, upper, lower, upper_frozen = bench.build_hip_models(dev)
imu2 = nets.IMUNet(15, 9, 512, 2, True, 0.1)
imu2.load_state_dict(imu.state_dict())
imu2 = imu2.to(dev).eval()
x, imu_in, body, target = bench.synth_batch(1234, dev)
su = StageStep("upper", upper, imu, lr=3e-5, use_graph=False)
sl = StageStep("lower", lower, imu2, upper_frozen=upper_frozen, lr=3e-5, use_graph=False)
su.bind(x, imu_in, body, target)
sl.bind(x, imu_in, body, target)
side = torch.cuda.Stream()


def both(concurrent):
    if not concurrent:
        su._body()
        sl._body()
        return
    main = torch.cuda.current_stream()
    side.wait_stream(main)
    with torch.cuda.stream(side):
        sl._body()
    su._body()
    main.wait_stream(side)


for concurrent in (False, True):
    both(concurrent)
    torch.cuda.synchronize()
    print('eager ok', concurrent, su.loss.item(), sl.loss.item(), flush=True)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        both(concurrent)

    def step():
        g.replay()
        su.opt.step()
        sl.opt.step()
    for _ in range(10):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 100
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    print("concurrent=%s: %.3f ms per U+L step, losses %.3f %.3f" % (concurrent, (time.perf_counter() - t0) / n * 1e3, su.loss.item(), sl.loss.item()))
