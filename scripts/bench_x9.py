"""fp32 projection GEMM: native fp32 MFMA kernel vs the bf16x9 split kernel (x9.hip) at the IMU_Net shapes; error of both vs fp64."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmego_amd import hip, ops  # noqa: E402

dev = torch.device("cuda:0")
hip.lib()


def timeit(fn, n=10, warm=3):
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


def split(x):
    y = torch.empty((x.shape[0], x.shape[1] // 8, 3, 8), dtype=torch.bfloat16, device=dev)
    hip.call("x9_split", x, x.stride(0), x.shape[0], x.shape[1], y)
    return y


for M, N, K in ((10240, 8192, 512), (10240, 8192, 1024)):
    A = torch.relu(torch.randn(M, K, device=dev))
    W = torch.randn(N, K, device=dev) / K ** 0.5
    b = torch.randn(N, device=dev)
    C = torch.empty((M, N), device=dev)
    C9 = torch.empty((M, N), device=dev)
    A9, W9 = split(A), split(W)
    ms_split = timeit(lambda: split(A))
    ms9 = timeit(lambda: hip.call("x9_gemm", A9, W9, C9, N, None, b, M, N, K, 0))
    ms9f = timeit(lambda: hip.call("x9_gemm", A9, W9, None, 0, C9, b, M, N, K, 0))
    hip.call("x9_gemm", A9, W9, C9, N, None, b, M, N, K, 0)
    ms32 = timeit(lambda: ops.linear_pair(A, W[:N // 2], W[N // 2:], b[:N // 2], b[N // 2:], C, N // 2), n=5)
    want = A[:256].double() @ W.double().T + b.double()
    fl = 2.0 * M * N * K
    print("M=%d N=%d K=%d: x9 %.3f ms (%.0f TFLOP/s fp32-equivalent; tile-major output %.3f ms; split of A %.3f ms) | native fp32 %.3f ms "
          "(%.0f TFLOP/s) | max err vs fp64: x9 %.2e, native %.2e"
          % (M, N, K, ms9, fl / ms9 / 1e9, ms9f, ms_split, ms32, fl / ms32 / 1e9,
             (C9[:256].double() - want).abs().max().item(), (C[:256].double() - want).abs().max().item()))

# recurrent step: native fp32 (lstm_step_dma2) vs split products (lstm_step_x9), 20 dependent launches per replayed HIP graph
H, Bn, T = 512, 512, 4
w = [torch.randn(4 * H, H, device=dev) * 0.04 for _ in range(2)]
bb = [torch.randn(4 * H, device=dev) * 0.04 for _ in range(2)]
xp = torch.randn(Bn * T, 8 * H, device=dev)
out = torch.zeros(Bn * T, 2 * H, device=dev)
c = torch.zeros(2, Bn, H, device=dev)
hf = torch.zeros(2, 2, Bn * H * 3, dtype=torch.bfloat16, device=dev)
hs = torch.zeros(Bn, 2 * H // 8, 3, 8, dtype=torch.bfloat16, device=dev)
wf = [split(w[d]).view(4, H // 32, 32, H // 16, 2, 3, 8).permute(1, 0, 3, 5, 4, 2, 6).contiguous() for d in range(2)]
xs, os_ = T * 8 * H, T * 2 * H


def step_x9(first=0):
    hip.call("lstm_step_x9", 2, Bn, H, first, hf[0, 0], hf[0, 1], wf[0], wf[1], xp, Bn // 32, 2 * (Bn // 32),
             out.data_ptr() + 4 * 2 * H, out.data_ptr() + 4 * 3 * H, os_, hs, hs.data_ptr() + 2 * (H // 8) * 24, 2 * H,
             hf[1, 0], hf[1, 1], c[0], c[1])


def step_f32():
    hip.call("lstm_step", 2, Bn, H, 0, out.data_ptr(), out.data_ptr() + 4 * H, os_, w[0], w[1], bb[0], bb[1],
             xp.data_ptr(), xp.data_ptr() + 16 * H, xs, out.data_ptr() + 4 * 2 * H, out.data_ptr() + 4 * 3 * H, os_,
             c[0], c[1], None, None, None, None)


def timeit_graph(fn, inner=20, n=10):
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


fl = 2.0 * 2 * Bn * H * 4 * H
m9, m32, m0 = timeit_graph(step_x9), timeit_graph(step_f32), timeit_graph(lambda: step_x9(1))
print("recurrent step Bn=512 H=512: x9 %.1f us (%.0f TFLOP/s fp32-equivalent; %.1f us without the product) | native fp32 %.1f us (%.0f TFLOP/s)"
      % (m9 * 1e3, fl / m9 / 1e9, m0 * 1e3, m32 * 1e3, fl / m32 / 1e9))
