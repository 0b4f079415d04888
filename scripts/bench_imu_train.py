"""Stage-1 (IMU_Net) training step on the HIP path: ms per step at the reference's batch (B=20 sequences x T=20 frames x 20
samples) and at the bench batch (B=64, T=8), plus the per-entry-point split of one step."""
import os
import sys
import time
import collections

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmego_amd import hip, imu_train, nets  # noqa: E402
from mmego_amd.params import FusedAdam  # noqa: E402

dev = torch.device("cuda:0")
hip.lib()
torch.manual_seed(0)
net = nets.IMUNet(15, 9, 512, 2, True, 0).to(dev).train()
opt = FusedAdam(net.flat(), lr=1e-4, weight_decay=0.001)
loss = torch.zeros(1, device=dev)
for B, T in ((20, 20), (64, 8)):
    imu = torch.randn(B, T, 20, 15, device=dev)
    Rg = torch.linalg.qr(torch.randn(B, T, 3, 3, device=dev))[0].contiguous()
    head = torch.randn(B, T, 3, device=dev)

    def step():
        with torch.no_grad():
            R, t = imu_train.forward_train(net, imu)
            dR, dt = torch.empty_like(R), torch.empty_like(t)
            hip.call("imu_loss", R.contiguous(), t.contiguous(), Rg, head, B * T, 1.0, loss, dR, dt)
            imu_train.backward(net, dR, dt)
        opt.step()
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 10
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    print("B=%d T=%d: %.2f ms per stage-1 step (eager)" % (B, T, (time.perf_counter() - t0) / n * 1e3))
    from mmego_amd.train_step import ImuStep
    st = ImuStep(net, lr=1e-4, use_graph=True)
    st.opt = opt
    tgt = torch.zeros(B, T, 21, 3, device=dev)
    tgt[:, :, 20] = head
    st.bind(imu, Rg, tgt)
    for _ in range(3):
        st.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        st.step()
    torch.cuda.synchronize()
    print("B=%d T=%d: %.2f ms per stage-1 step (HIP graph)" % (B, T, (time.perf_counter() - t0) / 20 * 1e3))
    # per entry point
    rec = collections.defaultdict(list)
    orig = hip.call

    def timed(name, *a):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); orig(name, *a); e1.record()
        key = name
        if name == "gemm":      # (A, sam, sak, B, sbk, sbn, C, scm, scn, bias, M, N, K, nbatch, ..., nsplit)
            key = "gemm M%d N%d K%d %s%s split%d" % (a[10], a[11], a[12], "k" if a[2] == 1 else "m", "k" if a[4] == 1 else "n", a[20])
        rec[key].append((e0, e1))
    hip.call = timed
    step()
    torch.cuda.synchronize()
    hip.call = orig
    tot = {k: (sum(a.elapsed_time(b) for a, b in v), len(v)) for k, v in rec.items()}
    for k, (ms, cnt) in sorted(tot.items(), key=lambda kv: -kv[1][0])[:14]:
        print("   %-40s %4d launches %8.3f ms  (%.1f us each)" % (k, cnt, ms, ms / cnt * 1e3))
