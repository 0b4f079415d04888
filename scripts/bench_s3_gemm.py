"""r06: mmego_split3_gemm, the 320 x 256-tile LDS-DMA kernel (wm = 10 / the library's choice) against the 256 x 128 register-staged one
(wm = 4) at rnn_fast's projection shapes: bit-identical results (same k order, same piece-product order), time per launch by event
pairs, bf16-MFMA fraction of 2.5 PFLOP/s."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mmego_amd import blocks, hip  # noqa: E402

dev = torch.device("cuda:0")
hip.lib()
g = torch.Generator().manual_seed(1)
for M, N, K in ((10240, 4096, 512), (10240, 4096, 1024), (640, 512, 256)):
    A = torch.randn(M, K, generator=g).to(dev)
    W = (torch.randn(N, K, generator=g) * 0.05).to(dev)
    bias = torch.randn(N, generator=g).to(dev)
    Ap, Wp = blocks.split3_cvt(A), blocks.split3_cvt(W)
    outs = {}
    for wm in (4, 10, 0):
        Cf = torch.zeros(M * N, device=dev)
        C = torch.zeros(M, N, device=dev)
        both = len(sys.argv) > 1 and sys.argv[1] == "both"      # default: the tile-major output alone, as the IMU_Net forward asks for it
        call = lambda: hip.call("split3_gemm", Ap, Wp, Cf, C if both else None, N, bias, M // 32, N // 32, K, M, 6, wm)
        for _ in range(3):
            call()
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(21)]
        ev[0].record()
        for i in range(20):
            call()
            ev[i + 1].record()
        torch.cuda.synchronize()
        ts = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(20))
        us = ts[len(ts) // 2] * 1e3
        outs[wm] = (Cf.clone(), C.clone())
        print("M %5d N %4d K %4d wm %2d: %7.1f us per launch (median of 20, back to back), %.3f of 2.5 PFLOP/s on the 6 piece products" % (
            M, N, K, wm, us, 6 * 2.0 * M * N * K / (us * 1e-6) / 2.5e15), flush=True)
    ref = (A.double() @ W.double().t() + bias.double()).float()
    print("   wm 10 == wm 4 bit for bit: tile-major %s, row-major %s; wm 0 == wm 10: %s; max |C - float64| %.3g" % (
        torch.equal(outs[10][0], outs[4][0]), torch.equal(outs[10][1], outs[4][1]), torch.equal(outs[0][1], outs[10][1]),
        float((outs[10][1] - ref).abs().max())), flush=True)
