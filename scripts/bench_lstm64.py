import os, sys, torch
sys.path.insert(0, '/root/repo')
from mmego_amd import hip
dev = torch.device('cuda:0')
B = 64
for T in (8, 16):
    for stash in (0, 1):
        xp = torch.randn(B * T, 512, device=dev)
        w = [torch.randn(256, 64, device=dev) * 0.1 for _ in range(2)]
        b = [torch.randn(256, device=dev) * 0.1 for _ in range(2)]
        out = torch.empty(B * T, 128, device=dev)
        hn = torch.empty(2, B, 64, device=dev); cn = torch.empty(2, B, 64, device=dev)
        gates = torch.empty(2, T, B, 256, device=dev); cst = torch.empty(2, T, B, 64, device=dev); hp = torch.empty(2, B * T, 64, device=dev)
        st = (gates[0], gates[1], cst[0], cst[1], hp[0], hp[1]) if stash else (None,) * 6
        def call():
            hip.call("lstm64_forward", B, T, xp, xp[:, 256:], 512, w[0], w[1], b[0], b[1], None, None, None, None, out, 128, hn[0], hn[1], cn[0], cn[1], *st, None, None, 0.0, None, 0)
        for _ in range(5): call()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100): call()
        e1.record(); torch.cuda.synchronize()
        print("T=%d stash=%d: %.1f us" % (T, stash, e0.elapsed_time(e1) * 10))
