"""IMU_Net forwards in one precision mode, eagerly, for `rocprofv3 --kernel-trace --stats` (per-kernel durations of the split3 /
fp32 / bf16 forward at B=64, T=8).  usage: s3_forward_trace.py [fp32|split3|bf16] [n]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmego_amd import blocks, hip, nets  # noqa: E402

prec = sys.argv[1] if len(sys.argv) > 1 else "split3"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device("cuda:0")
hip.lib()
torch.manual_seed(1)
imu = nets.IMUNet(15, 9, 512, 2, True, 0.1).to(dev).eval()
imu.precision = prec
inp = torch.randn(64, 8, 20, 15, device=dev)
with torch.no_grad(), blocks.two_chains(False):
    for _ in range(n):
        imu(inp)
torch.cuda.synchronize()
print("done", prec, n)
