"""The fused projection + recurrence step of the bf16 mode (bf16.hip, mmego_lstm_step_bf16_fused) alone, at the config-5 shape
(Bn = 32768 rows, H = 512): layer 0 (K = 512 + 512) and layer 1 (K = 1024 + 512), both directions per launch.
usage: python scripts/bench_fused_step.py [Bn]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmego_amd import blocks, hip  # noqa: E402

dev = torch.device("cuda:0")
hip.lib()
H = 512
Bn = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
torch.manual_seed(0)
lstm = blocks.LstmParams(H, H, 2).to(dev)
W = blocks.lstm_bf16_weights_fused(lstm)
Bp = (Bn + 31) // 32 * 32
g = torch.Generator(device="cuda").manual_seed(1)
mk = lambda: (torch.randn(Bp * H, device=dev, generator=g) * 0.5).to(torch.bfloat16)
xa, xb, ha, hb = mk(), mk(), mk(), mk()
pa, pb = mk(), mk()
hf = torch.zeros(2, Bp * H, dtype=torch.bfloat16, device=dev)
c = torch.randn(2, Bn, H, device=dev)
out = torch.zeros(Bn, 2 * H, device=dev)


def step(l, first=0):
    segs, whh, bias = W[l]
    if l == 0:
        a = (xa, xb, segs[0][0], segs[1][0], H, None, None, None, None, 0)
        nseg = 1
    else:
        a = (xa, xb, segs[0][0], segs[1][0], H, pa, pb, segs[0][1], segs[1][1], H)
        nseg = 2
    hip.call("lstm_step_bf16_fused", 2, Bn, H, first, nseg, *a, None if first else ha, None if first else hb, whh[0], whh[1], bias,
             out.data_ptr(), out.data_ptr() + 4 * H, 2 * H, hf[0], hf[1], c[0], c[1])


def timeit_graph(fn, inner=10, n=5):
    fn()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
        with torch.cuda.graph(gr, stream=s):
            for _ in range(inner):
                fn()
    torch.cuda.synchronize()
    for _ in range(2):
        gr.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        gr.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n / inner


for l, K in ((0, 2 * H), (1, 3 * H)):
    ms = timeit_graph(lambda: step(l))
    fl = 2.0 * 2 * Bn * 4 * H * K
    print("fused step layer %d  Bn=%d K=%d: %.1f us  %.0f TFLOP/s (%.3f of 2.5 PFLOP/s)" % (l, Bn, K, ms * 1e3, fl / ms / 1e9, fl / ms / 1e9 / 2500.0))
ms = timeit_graph(lambda: step(1, 1))
print("first timestep of layer 1 (K=%d, no h segment): %.1f us" % (2 * H, ms * 1e3))
