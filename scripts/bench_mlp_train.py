"""Train-mode three-stage pointwise MLP (conv k=1 -> BatchNorm -> ReLU x 3), forward + backward: the per-layer fused kernels of
mlp_train.hip against the generic launch chain (what wider layers take), in a replayed HIP graph; plus per-kernel event times."""
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmego_amd import blocks, hip, nets, ops  # noqa: E402
from mmego_amd.params import FlatParams  # noqa: E402

dev = torch.device("cuda:0")
_was = blocks._mlp3_fused_train
hip.lib()
for name, mod, rows in (("PointNet 6-8-16-24", nets.PointNet(), 65536), ("GlobalPointNet 28-32-48-64", nets.GlobalPointNet(), 65536),
                        ("BasePointNet 6-16-32-61", nets.BasePointNet(64), 32768)):
    mod = mod.to(dev).train()
    flat = FlatParams(mod).ensure()
    G = flat.grad
    cin, cout = mod.conv1.weight.shape[1], mod.conv3.weight.shape[0]
    x = torch.randn(rows, cin, device=dev)
    dy = torch.randn(rows, cout, device=dev)
    for fused in (False, True):
        blocks._mlp3_fused_train = _was if fused else (lambda mod, x: False)
        ar = ops.Arena(dev)
        y = ar.get("y3", (rows, cout))

        def run():
            blocks.mlp3_forward(ar, "m", mod, x, y, True)
            blocks.mlp3_backward(ar, "m", mod, x, y, dy, G, True)
        run()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        st = torch.cuda.Stream()
        st.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(st):
            run()
            with torch.cuda.graph(g, stream=st):
                run()
        torch.cuda.synchronize()
        for _ in range(3):
            g.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        rec = collections.defaultdict(list)
        orig = hip.call

        def timed(nm, *a):
            a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a0.record(); orig(nm, *a); a1.record()
            rec[nm].append((a0, a1))
        hip.call = timed
        run()
        torch.cuda.synchronize()
        hip.call = orig
        parts = ", ".join("%s x%d %.1f" % (k, len(v), sum(a.elapsed_time(b) for a, b in v) * 1e3 / len(v)) for k, v in rec.items())
        print("%-28s rows %6d %-8s: %7.1f us fwd+bwd (graph), %2d launches | eager us each: %s"
              % (name, rows, "fused" if fused else "unfused", e0.elapsed_time(e1) / 20 * 1e3, sum(len(v) for v in rec.values()), parts))
