"""BASELINE config 5 ("large-batch stress: B=2048 T=16 N=256, bf16 forward / fp32 accumulate") and the bf16 kernels alone.

  kernels   the bf16 projection GEMM and the bf16 recurrent step at the config-3 and config-5 shapes, TFLOP/s against the
            dense bf16 MFMA peak (2.5 PFLOP/s), next to the fp32 kernels at the same shapes (peak 157.3 TFLOP/s)
  imu       IMU_Net forward, fp32 vs bf16 mode, at B=64 T=8 and B=2048 T=16
  config5   the whole config-5 forward IMU_Net -> Upper_Net -> Lower_Net (bf16 mode for the IMU BiLSTM products, fp32
            elsewhere), with its algorithmic FLOPs
usage: python scripts/bench_config5.py [kernels] [imu] [config5] [--quick] [--trace]
  (no section = all three; --quick skips the B=2048 shapes; --trace = config5 in bf16 only, 1 warm-up + 2 forwards: the
  input for rocprofv3)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmego_amd import blocks, hip, nets, ops  # noqa: E402

dev = torch.device("cuda:0")
hip.lib()
args = [a for a in sys.argv[1:] if not a.startswith("--")]
quick, trace = "--quick" in sys.argv, "--trace" in sys.argv
sections = ["config5"] if trace else (args or ["kernels", "imu", "config5"])
PEAK_BF16, PEAK_F32 = 2500.0, 157.3
H = 512


def timeit(fn, n=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def timeit_graph(fn, inner=20, n=10):
    """ms per fn() with `inner` back-to-back calls captured in one HIP graph (no host launch cost in the figure)."""
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
        with torch.cuda.graph(g, stream=s):
            for _ in range(inner):
                fn()
    torch.cuda.synchronize()
    return timeit(g.replay, n=n, warm=2) / inner


def kernels():
    print("== kernels ==")
    for M, N, K in ((10240, 8192, 512), (10240, 8192, 1024)) + (() if quick else ((655360, 8192, 512), (655360, 8192, 1024))):
        A = torch.randn(M, K, device=dev)
        W = torch.randn(N, K, device=dev) / K ** 0.5
        b = torch.randn(N, device=dev)
        Ab = blocks.cvt_bf16(A, torch.empty((M, K), dtype=torch.bfloat16, device=dev))
        Wb = blocks.cvt_bf16(W, torch.empty((N, K), dtype=torch.bfloat16, device=dev))
        C = torch.empty((M, N), device=dev)
        ms = timeit(lambda: hip.call("gemm_bf16", Ab, K, Wb, K, None, 0, None, 0, C, b, M, N, K, 0), n=10)     # tile-major output
        tf = 2.0 * M * N * K / ms / 1e9
        ms32 = timeit(lambda: ops.linear_pair(A, W[:N // 2], W[N // 2:], b[:N // 2], b[N // 2:], C, N // 2), n=5)
        tf32 = 2.0 * M * N * K / ms32 / 1e9
        out_gb = 4.0 * M * N / 1e9
        print("projection M=%d N=%d K=%d: bf16 %.3f ms %.0f TFLOP/s (%.2f of bf16 peak; output alone %.2f GB = %.2f TB/s) | "
              "fp32 %.3f ms %.0f TFLOP/s (%.2f of fp32 peak)"
              % (M, N, K, ms, tf, tf / PEAK_BF16, out_gb, out_gb / ms, ms32, tf32, tf32 / PEAK_F32))
        del A, W, Ab, Wb, C

    for Bn in (512,) + (() if quick else (32768,)):
        T = 4
        lstm = blocks.LstmParams(H, H, 1).to(dev)
        Wc = blocks.lstm_bf16_weights(lstm)[0]
        xp = torch.randn(Bn * T, 8 * H, device=dev)             # (the bf16 step reads it as T tile-major [Bn, 8H] slabs)
        out = torch.zeros(Bn * T, 2 * H, device=dev)
        outb = torch.zeros(Bn * T, 2 * H, dtype=torch.bfloat16, device=dev)
        c = torch.zeros(2, Bn, H, device=dev)
        hf = torch.zeros(2, 2, Bn, H, dtype=torch.bfloat16, device=dev)
        xs, os_ = T * 8 * H, T * 2 * H

        def step_bf(first=0):
            hip.call("lstm_step_bf16", 2, Bn, H, first, hf[0, 0], hf[0, 1], Wc[2], Wc[3], xp, 1 * (Bn // 32), 2 * (Bn // 32),
                     out.data_ptr() + 4 * 2 * H, out.data_ptr() + 4 * 3 * H, os_,
                     outb.data_ptr() + 2 * 2 * H, outb.data_ptr() + 2 * 3 * H, os_, hf[1, 0], hf[1, 1], c[0], c[1])

        w0, w1 = lstm.w("weight_hh", 0, 0), lstm.w("weight_hh", 0, 1)
        b0, b1 = lstm.w("bias_hh", 0, 0), lstm.w("bias_hh", 0, 1)

        def step_f32():
            hip.call("lstm_step", 2, Bn, H, 0, out.data_ptr(), out.data_ptr() + 4 * H, os_, w0, w1, b0, b1,
                     xp.data_ptr(), xp.data_ptr() + 16 * H, xs, out.data_ptr() + 4 * 2 * H, out.data_ptr() + 4 * 3 * H, os_,
                     c[0], c[1], None, None, None, None)
        fl = 2.0 * 2 * Bn * H * 4 * H
        # streamed per step and direction: xproj 4H fp32, c read + write, h_t fp32 + 2 x bf16, h_{t-1} bf16 (W_hh stays in L2)
        gb = 2.0 * Bn * H * (4 * 4 + 4 + 4 + 4 + 2 + 2 + 2) / 1e9
        ms, ms32 = timeit_graph(step_bf), timeit_graph(step_f32)
        ms0 = timeit_graph(lambda: step_bf(1))
        print("recurrent step Bn=%d H=%d: bf16 %.1f us %.0f TFLOP/s (%.2f of bf16 peak; %.3f GB streamed = %.2f TB/s, %.2f of "
              "8 TB/s) | fp32 %.1f us %.0f TFLOP/s (%.2f of fp32 peak)"
              % (Bn, H, ms * 1e3, fl / ms / 1e9, fl / ms / 1e9 / PEAK_BF16, gb, gb / ms, gb / ms / 8.0,
                 ms32 * 1e3, fl / ms32 / 1e9, fl / ms32 / 1e9 / PEAK_F32))
        print("   (bf16 step without the product -- first timestep: launch + xproj + cell + stores: %.1f us)" % (ms0 * 1e3))
        del xp, out, outb, c, hf


def imu_flops(B, T):
    Bn = B * T
    return 2.0 * (Bn * 20) * (15 * H + 8 * H * H + 8 * H * 2 * H + 2 * 8 * H * H) + 2.0 * Bn * (8 * H * 2 * H * 2 + 2 * 8 * H * H)


def imu(imu_net):
    print("== IMU_Net forward ==")
    for B, T in ((64, 8),) + (() if quick else ((2048, 16),)):
        x = torch.randn(B, T, 20, 15, device=dev)
        res = {}
        for prec in ("fp32", "bf16"):
            imu_net.precision = prec
            with torch.no_grad():
                ms = timeit_graph(lambda: imu_net(x), inner=2, n=5)
                res[prec] = [v.clone() for v in imu_net(x)]
            print("IMU_Net forward B=%d T=%d %s: %.3f ms, %.0f frames/s, %.1f TFLOP/s"
                  % (B, T, prec, ms, B * T / ms * 1e3, imu_flops(B, T) / ms / 1e9))
        print("   max |R_bf16 - R_fp32| = %.2e, max |t_bf16 - t_fp32| = %.2e"
              % (float((res["fp32"][0] - res["bf16"][0]).abs().max()), float((res["fp32"][1] - res["bf16"][1]).abs().max())))
        for ar in imu_net._arenas.values():
            ar.bufs.clear()
        torch.cuda.empty_cache()


def config5(imu_net):
    print("== config 5: IMU_Net -> Upper_Net -> Lower_Net forward, B=2048 T=16 N=256 ==")
    B, T, N = 2048, 16, 256
    upper, lower = nets.UpperNet().to(dev).eval(), nets.LowerNet(64).to(dev).eval()
    g = torch.Generator().manual_seed(1)
    x0 = torch.zeros(B, T, N, 6)
    xyz = torch.randn(B, T, N, 3, generator=g) * torch.tensor([0.41, 0.30, 0.38]) + torch.tensor([0.84, 0.05, 0.18])
    x0[..., :3] = xyz
    x0[..., 3] = xyz.norm(dim=-1)
    x0[..., 4] = torch.randn(B, T, N, generator=g) * 0.41
    x0[..., 5] = torch.rand(B, T, N, generator=g) * 36 + 10
    dead = torch.rand(B, T, N, generator=g) < 0.4
    dead[:, :, :16] = False
    x0[dead] = 0
    x0 = x0.to(dev)
    body = (torch.randn(B, 20, 3, generator=g) * 0.2).to(dev)
    x_imu = torch.randn(B, T, 20, 15, generator=g).to(dev)
    h0 = torch.zeros(6, B, 64, device=dev)
    # 222.48 MFLOP (IMU) + 1.075 (Upper) + 4.632 (Lower) per frame (N=128 figures; the point-wise MLP part of
    # Upper/Lower doubles at N=256 and is < 3 % of the total either way)
    flops = imu_flops(B, T) + B * T * 2.0 * (1.075e6 + 4.632e6)
    for prec in (("bf16",) if trace else ("fp32", "bf16")):
        imu_net.precision = upper.precision = lower.precision = prec

        def fwd():
            with torch.no_grad():
                x = x0.clone()
                R, t = imu_net(x_imu)
                up = upper(x, h0, h0.clone(), body, R, t)[0]
                return lower(up, x, None, None, None, None, body, R, t)[0]
        ms = timeit(fwd, n=2 if trace else 3, warm=1)
        print("config 5 forward, precision %s on all three nets: %.1f ms, %.0f frames/s, %.0f TFLOP/s algorithmic (%.3f of the %s MFMA peak)"
              % (prec, ms, B * T / ms * 1e3, flops / ms / 1e9, flops / ms / 1e9 / (PEAK_BF16 if prec == "bf16" else PEAK_F32),
                 prec))
        torch.cuda.synchronize()


if "kernels" in sections:
    kernels()
if "imu" in sections or "config5" in sections:
    torch.manual_seed(0)
    net = nets.IMUNet(15, 9, H, 2).to(dev).eval()
    if "imu" in sections:
        imu(net)
    if "config5" in sections and not quick:
        config5(net)
