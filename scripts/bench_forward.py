"""Forward-only throughput (eval mode, fp32) of BASELINE.json's configs 2 and 5:
  config 2: Upper_Net forward, B=64, T=8, N=128
  config 5 shape: Upper_Net + Lower_Net forward, B=2048, T=16, N=256 (8.4 M points; fp32 here, the bf16 variant is not built)
Prints ms per forward, frames/s and the per-entry-point split of one forward."""
import collections
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmego_amd import hip, nets  # noqa: E402

dev = torch.device("cuda:0")
hip.lib()
torch.manual_seed(0)
upper, lower = nets.UpperNet().to(dev).eval(), nets.LowerNet(64).to(dev).eval()


def batch(B, T, N):
    g = torch.Generator().manual_seed(1)
    x = torch.zeros(B, T, N, 6)
    xyz = torch.randn(B, T, N, 3, generator=g) * torch.tensor([0.41, 0.30, 0.38]) + torch.tensor([0.84, 0.05, 0.18])
    x[..., :3] = xyz
    x[..., 3] = xyz.norm(dim=-1)
    x[..., 4] = torch.randn(B, T, N, generator=g) * 0.41
    x[..., 5] = torch.rand(B, T, N, generator=g) * 36 + 10
    dead = torch.rand(B, T, N, generator=g) < 0.4
    dead[:, :, :16] = False
    x[dead] = 0
    R = torch.linalg.qr(torch.randn(B, T, 3, 3, generator=g))[0].contiguous()
    t = torch.randn(B, T, 3, generator=g) * 0.1 + torch.tensor([0.8, 0.0, 0.9])
    body = torch.randn(B, 20, 3, generator=g) * 0.2
    return [v.to(dev) for v in (x, R, t, body)]


for name, (B, T, N), with_lower in (("config 2: Upper_Net fwd B=64 T=8 N=128", (64, 8, 128), False),
                                    ("config 5 shape: Upper+Lower fwd B=2048 T=16 N=256 (fp32)", (2048, 16, 256), True)):
    x0, R, t, body = batch(B, T, N)
    h0 = torch.zeros(6, B, 64, device=dev)

    def fwd():
        with torch.no_grad():
            x = x0.clone()
            up = upper(x, h0, h0.clone(), body, R, t)[0]
            if with_lower:
                return lower(up, x, None, None, None, None, body, R, t)[0]
            return up
    for _ in range(2):
        fwd()
    torch.cuda.synchronize()
    n = 5
    t0 = time.perf_counter()
    for _ in range(n):
        fwd()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    print("%s: %.2f ms per forward, %.0f frames/s" % (name, ms, B * T / ms * 1e3))
    rec = collections.defaultdict(list)
    orig = hip.call

    def timed(nm, *a):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); orig(nm, *a); e1.record()
        rec[nm].append((e0, e1))
    hip.call = timed
    fwd()
    torch.cuda.synchronize()
    hip.call = orig
    tot = {k: (sum(a.elapsed_time(b) for a, b in v), len(v)) for k, v in rec.items()}
    for k, (m, cnt) in sorted(tot.items(), key=lambda kv: -kv[1][0])[:8]:
        print("   %-22s %4d launches %9.3f ms" % (k, cnt, m))
